"""GPU exploration helper (not part of the product): MFMA f64 peak probe and per-size,
per-option timings of one logml evaluation with the library's own stage timers."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd  # noqa: E402
from gp_amd.synth import synth  # noqa: E402

ctx = gp_amd.Context(0)
print("mfma f64 16x16x4 peak probe: %.2f TFLOP/s" % ctx.probe_mfma_peak(20000), flush=True)
print("mfma f64 16x16x4 peak probe: %.2f TFLOP/s" % ctx.probe_mfma_peak(40000), flush=True)
sizes = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "8192", "16384"])]
nbos = [int(s) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["256", "512"])]
ctx.set_option("timing", 1)
for n in sizes:
    X, y = synth(n, 3)
    for nbo in nbos:
        ctx.set_option("nb_outer", nbo)
        ctx.logml(X, y, 1.0, [0.3], 0.1)
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            r = ctx.logml(X, y, 1.0, [0.3], 0.1)
            dt = time.perf_counter() - t0
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
        print("N=%6d nbo=%4d build %.3f ms  chol %.3f ms (%.1f TF)  fin %.3f ms | syrk %.3f ms (%.1f TF, %d launches) logml %.10g"
              % (n, nbo, best[0], best[1], chol_tf, best[2], kt["syrk"][1], syrk_tf, kt["syrk"][0], r[0]), flush=True)
