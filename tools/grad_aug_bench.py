import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
for n in (1438, 2048, 2560, 3072, 3584, 4096):
    X, y = synth(n, 1)
    res = []
    for aug in (8192, 0):
        ctx.set_option("grad_aug_n", aug); ctx.set_option("grad_aug_ng", aug)
        ctx.logml_grad(X, y, 1.0, [0.3], 0.1)
        t0 = time.perf_counter()
        for _ in range(10):
            o, g = ctx.logml_grad(X, y, 1.0, [0.3], 0.1)
        t1 = (time.perf_counter() - t0) / 10
        G = 4
        a = np.ones(G); r = 0.3 * (1 + 0.01 * np.arange(G)); s = 0.1 * np.ones(G)
        ctx.logml_grad_grid(X, y, a, r, s)
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.logml_grad_grid(X, y, a, r, s)
        t4 = (time.perf_counter() - t0) / 3
        res.append((t1, t4, g))
    print("n=%5d: augmented %8.1f us / four chains %8.1f us;  three chains of launches %8.1f us / %8.1f us;  grad rel diff %.1e"
          % (n, res[0][0] * 1e6, res[0][1] * 1e6, res[1][0] * 1e6, res[1][1] * 1e6, np.max(np.abs(res[0][2] - res[1][2])) / np.max(np.abs(res[1][2]))), flush=True)
