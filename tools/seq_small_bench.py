import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import gp_amd
ctx = gp_amd.Context(0)
for n in (21, 100, 256, 1000):
    rng = np.random.default_rng(n)
    X = np.sort(rng.uniform(0, 4, n)).reshape(-1, 1)
    Kn = np.asfortranarray(0.05 * np.eye(n)); mn = np.sin(X[:, 0])
    t0 = time.perf_counter(); s = ctx.seq_sampler(X, mn, Kn, 1.2, [0.5], 1e-6, 64); t1 = time.perf_counter()
    s.close(); t0 = time.perf_counter(); s = ctx.seq_sampler(X, mn, Kn, 1.2, [0.5], 1e-6, 64); t1 = time.perf_counter()
    ts = []
    for k in range(30):
        q = time.perf_counter(); mu, v = s.step(np.array([0.1 + 0.12 * k])); s.commit(mu); ts.append(time.perf_counter() - q)
    print("n=%4d create %.1f us, step+commit %.1f us (median of 30)" % (n, (t1 - t0) * 1e6, np.median(ts) * 1e6), flush=True)
    s.close()
