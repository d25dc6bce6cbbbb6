"""One look-ahead evaluation (after a warm-up one) for tracing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
n = 16384
res = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = gp_amd.Context(0); ctx.reserve(n); ctx.set_option("grid_lanes", 1)
ctx.set_option("cu_reserve", res); ctx.set_option("lookahead", 1)
ctx.set_option("syrk_persist", int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx.set_option("diag_waves", int(sys.argv[3]) if len(sys.argv) > 3 else 5)
X, y = synth(n, 3)
for rep in range(2):
    t0 = time.perf_counter(); v = ctx.logml(X, y, 1.0, [0.3], 0.1); print(v[0], time.perf_counter() - t0)
