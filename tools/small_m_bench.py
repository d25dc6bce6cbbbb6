"""gpmi_gp_condition (p_Xn / p_dotXn / sample_derivs moments) at the sizes the reference's drivers use (R/tests.R
N = 21, pendulum_fit*.R 79 .. 199): the augmented partial factorisation in ONE workgroup (k_potrf_small) against
the multi-launch chain; host-buffer calls, so both include the same PCIe traffic."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd
ctx = gp_amd.Context(0)
for n, m in ((21, 21), (79, 40), (79, 79), (100, 100), (128, 128), (199, 100), (199, 199), (256, 256)):
    t = np.linspace(0, n / 10.0, n); ts = np.linspace(0, n / 10.0, m) + 0.01
    y = np.sin(t)
    res = []
    for gc, sm in ((1024, 1024), (0, 1024), (0, 0)):   # fused one-launch call; chain with the one-workgroup factorisation; chain
        ctx.set_option("small_gc", gc)
        ctx.set_option("small_m", sm)
        ctx.gp_condition(t, ts, y, 1.0, 1.0, 0.01, 1e-8, "QQ", "RQ", "RR")
        t0 = time.perf_counter()
        for _ in range(100):
            ctx.gp_condition(t, ts, y, 1.0, 1.0, 0.01, 1e-8, "QQ", "RQ", "RR")
        res.append((time.perf_counter() - t0) / 100 * 1e6)
    print("n=%4d m=%4d (M = %4d rows): ONE launch (k_gp_condition_small) %7.1f us; launch chain with the factorisation in one workgroup %7.1f us, "
          "with the blocked factorisation %7.1f us per gp_condition call" % (n, m, n + m + 1, res[0], res[1], res[2]), flush=True)
