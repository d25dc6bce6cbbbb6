"""Large-N robustness: logml at N = 32768 and a ragged N, two blockings must agree (no oracle at this size)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
for n in (32768, 40001):
    X, y = synth(n, 3)
    res = []
    for nbo in (0, 512):
        ctx.set_option("nb_outer", nbo)
        t0 = time.perf_counter(); v = ctx.logml(X, y, 1.0, [0.3], 0.1); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); v = ctx.logml(X, y, 1.0, [0.3], 0.1); dt = time.perf_counter() - t0
        res.append(v[0]); print("n=%d nb_outer=%d logml=%.9f  %.1f ms  (%.1f TFLOP/s)" % (n, nbo, v[0], dt * 1e3, n ** 3 / 3 / dt / 1e12), flush=True)
    print("  rel diff %.2e" % (abs(res[0] - res[1]) / abs(res[0])))
