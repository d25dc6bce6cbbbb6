import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gp_amd
ctx = gp_amd.Context(0)
ctx.set_option("diag_waves", 4)
rng = np.random.default_rng(1)
for n in (5, 16, 17, 100, 128, 129, 300, 1000):
    B = rng.standard_normal((n, n)); A = B @ B.T + n * np.eye(n)
    L = ctx.potrf(A); Lr = np.linalg.cholesky(A)
    print(n, np.abs(L - Lr).max() / np.abs(Lr).max())
for k in (1, 16, 17, 128, 129, 255, 256, 257, 400):
    n = k + 37; A = np.eye(n) * 2.0; A[k - 1, k - 1] = -1.0
    try:
        ctx.potrf(A); print("no error?!", k)
    except gp_amd.NotPositiveDefinite as e:
        assert e.order == k, (k, e.order)
print("pivots ok")
