"""Random-size parity sweep (one-off robustness run, not a test): logml at random N (ragged, around the block and
outer-block thresholds) and D against numpy + LAPACK, relative error printed; exits non-zero above 1e-9."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd
import scipy.linalg as sla
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
sizes = sorted(set([int(v) for v in rng.integers(129, 6200, 30)] + [127, 128, 129, 255, 256, 257, 383, 385, 3584, 4608, 5000]))
worst = 0.0
for n in sizes:
    D = int(rng.integers(1, 4))
    X, y = synth(n, D)
    rho, sig = float(rng.uniform(0.15, 0.6)), float(rng.uniform(0.05, 0.3))
    d2 = np.zeros((n, n))
    for d in range(D):
        d2 += (X[:, d][:, None] - X[:, d][None, :]) ** 2
    K = np.exp(-0.5 * d2 / rho ** 2)
    K[np.diag_indices(n)] = 1.0 + sig * sig
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    z = sla.solve_triangular(L, y, lower=True, check_finite=False)
    want = -0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * n * math.log(2 * math.pi)
    got = ctx.logml(X, y, 1.0, [rho], sig)[0]
    rel = abs(got - want) / abs(want)
    worst = max(worst, rel)
    print("n=%5d D=%d rho=%.3f sigma=%.3f  logml %.9f  rel err %.2e" % (n, D, rho, sig, got, rel), flush=True)
print("worst %.2e" % worst)
sys.exit(0 if worst <= 1e-9 else 1)
