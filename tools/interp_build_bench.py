"""Interpolation-table build (gpmi_interp_build; test_interpolate.R:9-19): P factorisations + tangents,
one after another (grid_lanes = 1: what round 1 did) vs on the grid lanes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from scipy.stats import gamma
ctx = gp_amd.Context(0)
lp = np.linspace(gamma.ppf(0.05, 4.0, scale=0.25), gamma.ppf(0.95, 4.0, scale=0.25), 10)
for n in (1024, 4096, 8192):
    x = np.linspace(0.0, 0.35 * n, n)   # ~3 points per (smallest) length-scale: well conditioned with the 1e-10 jitter
    for lanes in (1, 2, 4):
        ctx.set_option("grid_lanes", lanes)
        ctx.interp_build(x, lp)
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.interp_build(x, lp)
        dt = (time.perf_counter() - t0) / 3
        print("n=%5d P=10 lanes=%d: table build %8.2f ms (%.2f ms per entry)" % (n, lanes, 1e3 * dt, 1e2 * dt), flush=True)
    ctx.interp_free()
