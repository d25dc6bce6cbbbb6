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
        os.path.join(CSRC, "se_device.h"),
        os.path.join(HERE, "..", "include", "gpmi.h"),
    ]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, probes=False):
    lib = PROBES_LIB if probes else LIB
    if not force and not needs_build(lib):
        return lib
    # the translation units are compiled side by side (chol_kernels.hip alone takes minutes: every one-workgroup kernel
    # inlines the whole diagonal-block body and the tile functions), then linked
    flags = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++20", "-fPIC", "-fvisibility=hidden"] + (
        ["-DGPMI_PROBES"] if probes else [])
    objdir = os.path.join(CSRC, "build", "probes" if probes else "product")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((cmd, obj, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, obj, proc in jobs:
        if proc.wait() != 0:
            for _, _, other in jobs:
                if other.poll() is None:
                    other.wait()
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = [hipcc()] + flags + ["-shared", "-o", lib] + [obj for _, obj, _ in jobs]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=CSRC)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, probes="--probes" in sys.argv[1:]))
