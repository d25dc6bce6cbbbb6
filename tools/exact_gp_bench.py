import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import gp_amd
ctx = gp_amd.Context(0)
for n in (30, 100, 256, 300, 1000):
    x = np.linspace(0, 10, n).reshape(-1, 1); z = np.random.default_rng(0).standard_normal(n)
    ctx.exact_gp_f(x, 1.0, [0.4], z, 1e-6)
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.exact_gp_f(x, 1.0, [0.4], z, 1e-6)
    d1 = (time.perf_counter() - t0) / 100
    t0 = time.perf_counter()
    for _ in range(20):
        K = ctx.se_cov(x, None, 1.0, [0.4], diag_add=1e-6); f = ctx.trmv_lower(ctx.potrf(K), z)
    d2 = (time.perf_counter() - t0) / 20
    print("n=%4d: gpmi_exact_gp_f %7.1f us per call; se_cov + potrf + trmv through host matrices %8.1f us" % (n, d1 * 1e6, d2 * 1e6), flush=True)
