import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import gp_amd
from scipy.stats import gamma
ctx = gp_amd.Context(0)
lp = np.linspace(gamma.ppf(0.05, 4.0, scale=0.25), gamma.ppf(0.95, 4.0, scale=0.25), 10)
for n in (50, 100, 200, 400):
    x = np.linspace(0.0, 10.0, n)
    ctx.interp_build(x, lp)
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.interp_build(x, lp)
    dt = (time.perf_counter() - t0) / 10
    L, dL = ctx.rbf_cov_chol(x, lp[3])
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.rbf_cov_chol(x, lp[3])
    d1 = (time.perf_counter() - t0) / 20
    z = np.random.default_rng(0).standard_normal(n)
    ctx.approx_Lz(0.9, z)
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.approx_Lz(0.9, z)
    d2 = (time.perf_counter() - t0) / 100
    print("n=%4d: table build P=10 %8.1f us; one rbf_cov_chol (host call) %8.1f us; approx_Lz (host call) %7.1f us" % (n, dt*1e6, d1*1e6, d2*1e6), flush=True)
