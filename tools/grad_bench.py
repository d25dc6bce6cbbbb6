"""Value + gradient (gpmi_logml_grad): one at a time and four at once on the lanes (rstan's four chains)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
for n in (4096, 8192, 16384):
    X, y = synth(n, 3)
    ctx.logml_grad(X, y, 1.0, [0.3], 0.1)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.logml_grad(X, y, 1.0, [0.3], 0.1)
    t1 = (time.perf_counter() - t0) / 3
    G = 8
    a = np.ones(G); r = 0.3 * (1 + 0.01 * np.arange(G)); s = 0.1 * np.ones(G)
    ctx.logml_grad_grid(X, y, a, r, s)
    t0 = time.perf_counter()
    ctx.logml_grad_grid(X, y, a, r, s)
    t4 = (time.perf_counter() - t0) / G
    t0 = time.perf_counter(); ctx.logml(X, y, 1.0, [0.3], 0.1); tv = time.perf_counter() - t0
    print("n=%5d: value %.2f ms; value + gradient %.2f ms one at a time, %.2f ms per point on the lanes (%.0f%%)" % (n, tv * 1e3, t1 * 1e3, t4 * 1e3, 100 * t4 / t1), flush=True)
