"""Large-N robustness beyond 2^31 matrix elements (N > 46341), where an int32 element index would wrap.
No oracle and no LAPACK at these sizes, so the check is a decomposition the domain offers: two clusters
of points so far apart that every cross-covariance underflows to exactly 0 make K block diagonal, hence
logml(all N points) = logml(first cluster) + logml(second cluster), each of which is a safe-size problem
(N/2 < 46341).  The second cluster's block lives entirely past element 2^31 of the matrix."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [50001, 65536]
for n in sizes:
    X, y = synth(n, 3)
    X = np.asfortranarray(X)
    h = n // 2 + 3  # ragged split
    X[h:, 0] += 1000.0
    t0 = time.perf_counter(); full = ctx.logml(X, y, 1.0, [0.3], 0.1); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); full = ctx.logml(X, y, 1.0, [0.3], 0.1); dt = time.perf_counter() - t0
    a = ctx.logml(np.asfortranarray(X[:h]), y[:h], 1.0, [0.3], 0.1)
    b = ctx.logml(np.asfortranarray(X[h:]), y[h:], 1.0, [0.3], 0.1)
    rel = abs(full[0] - (a[0] + b[0])) / abs(full[0])
    print("n=%d logml=%.9f  parts %.9f + %.9f  rel diff %.2e   %.1f ms (%.1f TFLOP/s)"
          % (n, full[0], a[0], b[0], rel, dt * 1e3, n ** 3 / 3 / dt / 1e12), flush=True)
    assert np.isfinite(full[0]) and rel <= 1e-10, (n, full, a, b)
print("ok")
