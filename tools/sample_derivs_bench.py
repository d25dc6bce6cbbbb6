"""The reference's derivative-imputation loop (pendulum_fit.R:261-268: mclapply(s_list[1:100], sample_derivs_both_states,
mc.cores = 2): 2 x 100 draws, each with its own (l, a, sy) and noisy series, at N = 199): B draws through
gpmi_sample_derivs_batch -- one workgroup per draw against the launch chains on the four lanes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd
ctx = gp_amd.Context(0)
cases = [(25, 25), (79, 79), (199, 199), (256, 256)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for n, m in cases:
    rng = np.random.default_rng(n)
    t = np.sort(rng.uniform(0, n / 10.0, n)); ts = t[:m].copy() if m <= n else np.sort(rng.uniform(0, n / 10.0, m))
    for B in (1, 4, 16, 64, 200):
        Y = np.sin(t)[:, None] + 0.05 * rng.standard_normal((n, B)); Z = rng.standard_normal((m, B))
        P = np.column_stack([1.0 + 0.1 * rng.random(B), 1.0 + 0.1 * rng.random(B), 0.05 + 0.02 * rng.random(B)])
        res = {}
        for name, sd in (("one workgroup per draw", 1024), ("lanes", 0)):
            ctx.set_option("small_sd", sd); ctx.set_option("small_sdb", 0)
            ctx.sample_derivs_batch(t, ts, Y, P, 1e-8, Z)
            best = 1e9
            for r in range(3):
                t0 = time.perf_counter()
                d, mu, info = ctx.sample_derivs_batch(t, ts, Y, P, 1e-8, Z)
                best = min(best, time.perf_counter() - t0)
            res[name] = (best, d, info)
        a, b = res["one workgroup per draw"], res["lanes"]
        ok = (a[2] == 0) & (b[2] == 0)
        rel = np.max(np.abs(a[1][:, ok] - b[1][:, ok])) / np.max(np.abs(b[1][:, ok])) if ok.any() else float("nan")
        print("n=%4d m=%4d B=%4d  one workgroup per draw %9.1f us (%7.1f us/draw)   lanes %9.1f us (%7.1f us/draw)   max rel diff %.1e  not-PD %d/%d"
              % (n, m, B, a[0] * 1e6, a[0] * 1e6 / B, b[0] * 1e6, b[0] * 1e6 / B, rel, int((a[2] != 0).sum()), int((b[2] != 0).sum())), flush=True)
