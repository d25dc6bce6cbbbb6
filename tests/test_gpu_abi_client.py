"""GPU: the C ABI used from plain C (tests/abi_client.c) -- no Python, no torch in that process.
This is the binding surface the reference's R .Call() shim (r/gpmi_shim.c) sits on."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client(tmp_path):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc on this box")
    lib = os.path.join(ROOT, "gp_amd", "csrc")
    assert os.path.exists(os.path.join(lib, "libgpmi.so")), "libgpmi.so must be built in-tree"
    exe = str(tmp_path / "abi_client")
    subprocess.check_call([gcc, "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_client.c"),
                           "-L" + lib, "-lgpmi", "-lm", "-Wl,-rpath," + lib, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "abi_client: ok" in r.stdout, r.stdout + r.stderr
