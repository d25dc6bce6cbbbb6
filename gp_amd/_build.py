"""Build libgpmi.so (HIP, gfx950) in-tree with hipcc.  No torch involvement: the
library is a plain C-ABI shared object (include/gpmi.h)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["se_kernels.hip", "chol_kernels.hip", "gpmi_api.hip"]
LIB = os.path.join(CSRC, "libgpmi.so")


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libgpmi cannot be built (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        os.path.join(CSRC, "gpmi_internal.h"),
        os.path.join(CSRC, "factor16.h"),
        os.path.join(HERE, "..", "include", "gpmi.h"),
    ]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++20", "-fPIC", "-shared",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
