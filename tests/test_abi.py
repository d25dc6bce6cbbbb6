"""CPU: the C-ABI library loads and exports every symbol include/gpmi.h declares; without a
GPU the product path fails loudly (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(probes=False):
    src = open(os.path.join(ROOT, "include", "gpmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    m = re.search(r"#ifdef GPMI_PROBES(.*?)#endif", src, flags=re.S)
    assert m, "probe section not found"
    if probes:
        src = m.group(1)
    else:
        src = src.replace(m.group(0), "")
    return sorted(set(re.findall(r"\b(gpmi_[A-Za-z0-9_]+)\s*\(", src)))


def test_build_and_exports():
    from gp_amd import _build, _lib
    lib = _build.build()
    assert os.path.exists(lib)
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    exported = set(re.findall(r" T (gpmi_[A-Za-z0-9_]+)", out))
    declared = _header_functions()
    assert declared, "header parse found nothing"
    missing = [f for f in declared if f not in exported]
    assert not missing, missing
    # ... and nothing else: no probe entry point, no internal launcher (-fvisibility=hidden)
    assert sorted(exported) == declared, sorted(exported ^ set(declared))
    assert not re.findall(r" T (launch_|_Z\w*launch_|_Z\w*gemm8|_Z\w*gemm9)", out)
    assert sorted(_lib.SYMBOLS) == declared  # python binding covers the whole header
    h = _lib.load()
    assert h.gpmi_version() == 302
    # gfx950 code object is embedded
    assert b"gfx950" in open(lib, "rb").read()


def test_probe_build_is_a_separate_library():
    # tools/ run on libgpmi_probes.so (-DGPMI_PROBES): the product ABI plus the gpmi_probe_* entry points
    from gp_amd import _build, _lib
    lib = _build.build(probes=True)
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    exported = set(re.findall(r" T (gpmi_[A-Za-z0-9_]+)", out))
    assert sorted(exported) == sorted(_header_functions() + _header_functions(probes=True))
    assert sorted(_lib.PROBE_SYMBOLS) == _header_functions(probes=True)
    code = open(lib, "rb").read()
    assert b"k_gemm9" in code and b"k_gemm9" not in open(_build.LIB, "rb").read()


def test_signatures_are_plain_c():
    src = open(os.path.join(ROOT, "include", "gpmi.h")).read()
    assert 'extern "C"' in src
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert "torch" not in code.lower() and "std::" not in code and "&" not in code


def test_no_cpu_fallback_without_device():
    import gp_amd
    from gp_amd import _lib
    if gp_amd.device_count() > 0:
        pytest.skip("a GPU is visible; covered by the gpu tests")
    with pytest.raises(gp_amd.GpmiError) as e:
        gp_amd.Context(0)
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)
    from gp_amd import kernels
    with pytest.raises(gp_amd.GpmiError):
        kernels.QQ([0.0, 1.0], [0.0, 1.0], [1.0, 1.0])


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)


def test_only_tests_smoke_and_cpu_baseline_touch_the_oracle():
    # tools/, r/ and the launcher part of bench.py never import, link or execute oracle/; bench.py does so only inside
    # its cpu_baseline / cpu_lapack legs (function-local imports), __graft_entry__ only in smoke() and build()
    import re
    for d in ("tools", "r"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".sh", ".c", ".R", ".hip")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"(from|import)\s+oracle|oracle/|liboracle|libgporacle", txt), os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    for m in re.finditer(r"^(\s*)from oracle import", bench, flags=re.M):
        assert len(m.group(1)) >= 4, "bench.py imports the oracle at module level"
    body = bench[bench.index("def main()"):]
    assert "from oracle" not in body and "orc." not in body.split("def main()")[1].split("cpu_baseline(")[0]


def test_null_context_is_an_error_not_a_crash():
    from gp_amd import _lib
    h = _lib.load()
    assert h.gpmi_sync(None) == -1
    assert b"NULL" in h.gpmi_last_error()


def test_r_shim_parses_and_matches_the_r_wrappers():
    """r/gpmi_shim.c is what `.Call` binds (it replaces covariance.cpp:6-9's Rcpp export); there is no R in the
    image, so it is parsed and type-checked by gcc against tests/r_api/ -- declarations of the R API it uses,
    nothing of R's behaviour -- and every .Call("name", ...) of r/gpmi.R is matched, by name and number of
    arguments, to a `SEXP name(SEXP, ...)` of the shim."""
    shim = os.path.join(ROOT, "r", "gpmi_shim.c")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                        "-I" + os.path.join(ROOT, "tests", "r_api"), "-I" + os.path.join(ROOT, "include"), shim],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = open(shim).read()
    defs = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
            for m in re.finditer(r"^SEXP (gpmi_R_\w+)\(([^)]*)\)", src, flags=re.M)}
    assert len(defs) >= 19
    rsrc = open(os.path.join(ROOT, "r", "gpmi.R")).read()
    calls = []
    for m in re.finditer(r'\.Call\("(gpmi_R_\w+)"', rsrc):
        # count top-level commas of the call's argument list
        i = m.end(); depth = 1; commas = 0
        while depth:
            ch = rsrc[i]
            depth += ch in "([{"
            depth -= ch in ")]}"
            commas += (ch == "," and depth == 1)
            i += 1
        calls.append((m.group(1), commas))
    assert len(calls) >= 19
    for name, nargs in calls:
        assert name in defs, name
        assert defs[name] == nargs, (name, defs[name], nargs)
    # every ABI function the shim calls is declared in include/gpmi.h
    declared = set(_header_functions())
    used = set(re.findall(r"\b(gpmi_(?!R_)[a-z_A-Z0-9]+)\s*\(", src)) - {"gpmi_ctx", "gpmi_seq"}
    assert used <= declared, used - declared


def test_no_valu_write_in_front_of_a_dpp_read_in_the_built_kernels():
    """factor16's DPP chain (hand-written `asm`, its own hazard spacing) as the compiler actually laid it out in every
    kernel of the BUILT library that embeds it: no VALU write of a register within two wait states in front of a DPP
    read of it (the hardware does not interlock that; a register-allocator copy there gave wrong numbers in round 3).
    Disassembles the code objects of libgpmi.so (llvm-objdump) and scans them with tools/dpp_hazard_scan.py."""
    import importlib.util
    import tempfile
    from gp_amd import _build
    lib = _build.build()
    spec = importlib.util.spec_from_file_location("dpp_hazard_scan", os.path.join(ROOT, "tools", "dpp_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n = suspects = 0
    for path in mod.disassemble(lib, tempfile.mkdtemp()):
        a, b = mod.scan(path, verbose=False)
        n += a
        suspects += b
    assert n >= 10 * 352, n         # the chain is in the diagonal kernel, both update kernels and every one-workgroup kernel
    assert suspects == 0
