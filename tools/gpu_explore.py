"""GPU exploration helper (not part of the product): per-size, per-option timings of one
logml evaluation with the library's own stage timers.
usage: gpu_explore.py n=4096,16384 nbo=256,512 la=0,1 gv=0 res=0,16 mm=0 order=0 probe=1"""
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd  # noqa: E402
from gp_amd.synth import synth  # noqa: E402

opts = {"n": "16384", "nbo": "256", "la": "1", "gv": "0", "res": "0", "mm": "0", "order": "0", "probe": "0", "stg": "131076", "ra": "1"}
for a in sys.argv[1:]:
    k, v = a.split("=")
    opts[k] = v
L = lambda k: [int(x) for x in opts[k].split(",")]
ctx = gp_amd.Context(0)
if int(opts["probe"]):
    print("mfma f64 16x16x4 peak probe: %.2f TFLOP/s at %.0f MHz" % ctx.probe_mfma_peak(40000), flush=True)
ctx.set_option("timing", 1)
for n in L("n"):
    X, y = synth(n, 3)
    for nbo, la, gv, res, mm, order in itertools.product(L("nbo"), L("la"), L("gv"), L("res"), L("mm"), L("order")):
        ctx.set_option("rect_auto", int(opts["ra"]))
        ctx.set_option("stagger", int(opts["stg"]))
        ctx.set_option("gemm_variant", gv)
        ctx.set_option("syrk_order", order)
        ctx.set_option("nb_outer", nbo)
        ctx.set_option("lookahead", la)
        ctx.set_option("cu_mask_mode", mm)
        ctx.set_option("cu_reserve", res)
        ctx.logml(X, y, 1.0, [0.3], 0.1)
        best = None
        for rep in range(3):
            r = ctx.logml(X, y, 1.0, [0.3], 0.1)
            ms = ctx.last_timing()
            if best is None or ms[1] < best[1]:
                best = ms.copy()
        ctx.set_option("kernel_timing", 1)
        ctx.kernel_timing(True)
        ctx.logml(X, y, 1.0, [0.3], 0.1)
        kt = ctx.kernel_timing(True)
        ctx.set_option("kernel_timing", 0)
        chol_tf = n ** 3 / 3.0 / (best[1] * 1e-3) / 1e12
        syrk_tf = kt["syrk"][2] / (kt["syrk"][1] * 1e-3) / 1e12 if kt["syrk"][1] > 0 else 0
        print("N=%6d nbo=%4d la=%d gv=%d res=%3d mm=%d ord=%d | build %.3f chol %.3f ms (%.1f TF) fin %.3f | syrk %.3f ms (%.1f TF, %d) logml %.10g"
              % (n, nbo, la, gv, res, mm, order, best[0], best[1], chol_tf, best[2], kt["syrk"][1], syrk_tf, kt["syrk"][0], r[0]), flush=True)
