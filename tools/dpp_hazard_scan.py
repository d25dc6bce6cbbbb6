"""Scan the gfx950 code of the Cholesky kernels for the hazard factor16's hand-scheduled DPP chain must never meet:
a VALU instruction that writes a VGPR within two wait states in front of a DPP instruction that READS that VGPR as its
broadcast source.  The hardware does not interlock that case, the compiler's hazard recogniser cannot see into the
`asm` statements of factor16.h, and the chain carries its own `s_nop` only where the source listing needs one -- so a
register-allocator copy (v_accvgpr_read, a spill reload by v_readlane ... v_mov) placed in front of a DPP read silently
produces wrong numbers (seen in round 3 when a kernel was given 512 registers: wrong logml from n = 6 on).

  python tools/dpp_hazard_scan.py            disassembles the code objects of the BUILT gp_amd/csrc/libgpmi.so (llvm-objdump, seconds)
  python tools/dpp_hazard_scan.py FILE       scans FILE: a library (.so), an `hipcc -S` listing or an llvm-objdump listing
Exit status 1 when a suspect is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    m = re.match(r"-?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _instructions(path):
    """[(kernel name, [instruction text, ...]), ...] of an `hipcc -S` listing or an llvm-objdump -d listing"""
    lines = open(path).read().split("\n")
    out = []
    cur = None
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <(_Z[^>]*)>:", l)            # objdump: kernel entry
        if m is None and l.startswith("_Z") and "@" not in l and l.rstrip().endswith(":"):
            m = re.match(r"^(_Z\S*):", l)                      # -S listing: kernel label
        if m:
            cur = (m.group(1), [])
            out.append(cur)
            continue
        if cur is None:
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":") or re.match(r"^[0-9a-f]+ <", t):
            continue
        t = t.split("//")[0].split(";")[0].strip()
        if t:
            cur[1].append(t)
    return out


def scan(path, verbose=True):
    total = suspects = 0
    for name, ins in _instructions(path):
        ndpp = bad = 0
        for i, t in enumerate(ins):
            if "_dpp" not in t:
                continue
            ndpp += 1
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            src = regs(ops[1].split()[0])
            ws, j = 0, i - 1
            while j >= 0 and ws < 2:   # walk back two wait states: s_nop N counts N + 1, anything else 1
                p = ins[j]
                if p.startswith("s_nop"):
                    ws += int(p.split()[1]) + 1
                else:
                    if p.startswith("v_") and not p.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                        if regs(p.split(None, 1)[1].split(",")[0].strip()) & src:
                            bad += 1
                            if verbose:
                                print("  SUSPECT in %s:\n    %s\n    %s" % (name[:60], p, t))
                            break
                    ws += 1
                j -= 1
        if ndpp and verbose:
            print("%-70s %4d DPP instructions, %d suspects" % (name[:70], ndpp, bad))
        total += ndpp
        suspects += bad
    return total, suspects


def disassemble(lib, workdir):
    """llvm-objdump listings of the gfx950 code objects bundled in a built library (copied to workdir first: the
    unbundler writes next to its input)"""
    import shutil
    objdump = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-objdump")
    local = os.path.join(workdir, os.path.basename(lib))
    shutil.copy(lib, local)
    subprocess.check_call([objdump, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    outs = []
    for f in sorted(os.listdir(workdir)):
        if f.startswith(os.path.basename(lib) + ".") and "amdgcn" in f:
            dis = os.path.join(workdir, f + ".dis")
            with open(dis, "w") as fh:
                subprocess.check_call([objdump, "-d", os.path.join(workdir, f)], stdout=fh)
            outs.append(dis)
    return outs


def compile_asm(out):
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++20", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "gp_amd", "csrc", "chol_kernels.hip"), "-o", out], stderr=subprocess.DEVNULL)


if __name__ == "__main__":
    target = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gp_amd", "csrc", "libgpmi.so")
    paths = disassemble(target, tempfile.mkdtemp()) if target.endswith(".so") else [target]
    n = s = 0
    for path in paths:
        a, b = scan(path)
        n += a
        s += b
    print("%d DPP instructions, %d hazard suspects" % (n, s))
    sys.exit(1 if s or not n else 0)
