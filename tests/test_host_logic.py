"""CPU: host-side logic of the mirror modules (argument conventions, sharding, arg-max)."""
import math

import numpy as np
import pytest


def test_phi_accepts_list_vector_dict():
    from gp_amd.kernels import _phi
    for phi in ([1.5, 0.3], (1.5, 0.3), np.array([1.5, 0.3]), {"alpha": 1.5, "l": 0.3}, [np.array([1.5]), 0.3]):
        a, l = _phi(phi)
        assert a == 1.5 and l.tolist() == [0.3]
    a, l = _phi([2.0, [0.1, 0.2, 0.3]])
    assert l.tolist() == [0.1, 0.2, 0.3]
    with pytest.raises(ValueError):
        _phi([1.0])


def test_shard_indices_cover_grid_once():
    from gp_amd.grid import shard_indices
    for G in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            allidx = np.concatenate([shard_indices(G, r, world) for r in range(world)])
            assert sorted(allidx.tolist()) == list(range(G))
    assert shard_indices(64, 3, 8).tolist() == list(range(3, 64, 8))


def test_stan_lp_matches_oracle(orc):
    from gp_amd.stan_models import stan_lp
    for args in [(-3.5, 7.25, 1.2, 0.8, 0.3), (10.0, 0.5, 0.4, 2.5, 1.1)]:
        assert stan_lp(*args) == pytest.approx(orc.stan_lp(*args), rel=1e-15)


def test_get_ml_from_grid_argmax():
    from gp_amd.stan_models import get_ml_from_grid
    v = np.array([[1.0, np.nan, 3.0], [7.0, -np.inf, 2.0]])
    r = get_ml_from_grid(v, 1.0, [0.1, 0.2], [0.5, 0.6, 0.7])
    assert r == {"alpha": 1.0, "rho": 0.2, "sigma": 0.5}


def test_kind_names():
    from gp_amd import _lib
    assert _lib.KINDS == ("QQ", "QR", "RQ", "RR", "QT", "TQ", "RT", "TR", "TT")
    assert _lib._kind("TT") == 8 and _lib._kind(3) == 3


def test_committed_bench_line_keeps_the_contract():
    """The bench line the GPU box produced for this code (profiles/r02_bench_c3.json) carries every key the
    driver's contract and the tier's measurement section name, with consistent values."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.loads(open(os.path.join(root, "profiles", "r02_bench_c3.json")).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "c4"):
        assert k in d, k
    assert d["metric"] == "gp_logml_evals_per_sec_N16384_D3" and d["unit"] == "evals/s" and d["dtype"] == "f64"
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["achieved"] <= r["peak"] and r["traffic"] is not None and r["traffic_source"].startswith("profiles/r02_pmc_bench_c3_n16384")
    # whole-evaluation flops cannot exceed the peak either
    assert 16384 ** 3 / 3.0 / (d["ms_per_step"] * 1e-3) / 1e12 <= r["peak"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1
    c4 = d["c4"]
    assert c4["grid_points"] == 64 and c4["sharded_results_bit_identical_to_one_rank"] is True and c4["results_ok"]
