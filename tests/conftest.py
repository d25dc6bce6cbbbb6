import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; see oracle/gp_oracle.c)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ctx():
    """libgpmi context on cuda:0 -- fails loudly (no fallback) when the HIP path is unavailable."""
    import gp_amd
    return gp_amd.default_context(0)


@pytest.fixture(scope="session")
def golden():
    import json
    g = {}
    for name in ("gp_derivs", "kat", "ch2"):
        with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as f:
            g[name] = json.load(f)
    return g
