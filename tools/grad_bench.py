"""Time of gpmi_logml_grad (value + gradient) next to gpmi_logml (value only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
for n in (4096, 8192, 16384):
    X, y = synth(n, 3)
    for rep in range(2):
        t0 = time.perf_counter(); v = ctx.logml(X, y, 1.0, [0.3], 0.1); t1 = time.perf_counter()
        out, g = ctx.logml_grad(X, y, 1.0, [0.3], 0.1); t2 = time.perf_counter()
    print("n=%6d  logml %.1f ms   logml+grad %.1f ms  (%.1f Cholesky-equivalents on top)  grad=%s" %
          (n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), (t2 - t1) / (t1 - t0) - 1.0, np.array2string(g, precision=4)), flush=True)
