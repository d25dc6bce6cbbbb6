"""One interpolation-table build under a kernel trace:
     rocprofv3 --kernel-trace -d DIR -- python3 tools/interp_trace_run.py N LANES
   then  python3 tools/queue_overlap.py DIR  for the per-queue picture."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd
from scipy.stats import gamma
n = int(sys.argv[1]); lanes = int(sys.argv[2])
ctx = gp_amd.Context(0)
lp = np.linspace(gamma.ppf(0.05, 4.0, scale=0.25), gamma.ppf(0.95, 4.0, scale=0.25), 10)
x = np.linspace(0.0, 0.35 * n, n)
ctx.set_option("grid_lanes", lanes)
ctx.interp_build(x, lp)   # allocations, stream calibration
import time
time.sleep(0.05)          # an idle gap marks the start of the traced build
t0 = time.perf_counter()
ctx.interp_build(x, lp)
print("n=%d lanes=%d: %.2f ms" % (n, lanes, 1e3 * (time.perf_counter() - t0)))
