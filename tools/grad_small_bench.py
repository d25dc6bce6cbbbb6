"""Value + gradient (gpmi_logml_grad) at the sizes the reference's Stan fits run (N = 21 ... 1438): what one leapfrog step costs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [21, 64, 128, 199, 256, 512, 1000, 1438]:
    X, y = synth(n, 1)
    ctx.logml_grad(X, y, 1.0, [0.3], 0.1)
    reps = 100 if n <= 256 else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.logml_grad(X, y, 1.0, [0.3], 0.1)
    tg = (time.perf_counter() - t0) / reps
    ctx.logml(X, y, 1.0, [0.3], 0.1)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.logml(X, y, 1.0, [0.3], 0.1)
    tv = (time.perf_counter() - t0) / reps
    G = 4
    a = np.ones(G); r = 0.3 * (1 + 0.01 * np.arange(G)); s = 0.1 * np.ones(G)
    ctx.logml_grad_grid(X, y, a, r, s)
    t0 = time.perf_counter()
    for _ in range(reps // 4):
        ctx.logml_grad_grid(X, y, a, r, s)
    t4 = (time.perf_counter() - t0) / (reps // 4)
    print("n=%5d: value %8.1f us   value + gradient %8.1f us   four chains at once %8.1f us (%.1f us per chain)" % (n, tv * 1e6, tg * 1e6, t4 * 1e6, t4 * 1e6 / 4), flush=True)
