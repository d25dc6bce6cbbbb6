"""Build libgpmi.so (HIP, gfx950) in-tree with hipcc.  No torch involvement: the
library is a plain C-ABI shared object (include/gpmi.h)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["se_kernels.hip", "chol_kernels.hip", "gpmi_api.hip"]
LIB = os.path.join(CSRC, "libgpmi.so")
PROBES_LIB = os.path.join(CSRC, "libgpmi_probes.so")  # -DGPMI_PROBES: tools/ only, never loaded by the product


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libgpmi cannot be built (there is no CPU fallback)")


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        os.path.join(CSRC, "gpmi_internal.h"),
        os.path.join(CSRC, "factor16.h"),
        os.path.join(HERE, "..", "include", "gpmi.h"),
    ]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, probes=False):
    lib = PROBES_LIB if probes else LIB
    if not force and not needs_build(lib):
        return lib
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++20", "-fPIC", "-shared",
           "-fvisibility=hidden"] + (["-DGPMI_PROBES"] if probes else []) + ["-o", lib] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, probes="--probes" in sys.argv[1:]))
