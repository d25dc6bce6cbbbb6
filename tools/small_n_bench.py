"""Small-N regime (the sizes the reference's own scripts run: N = 21 ... 1000): latency of one host-buffer
evaluation (gpmi_logml: inputs in, 3 doubles back, blocking), of one device-resident evaluation, and the
per-evaluation time of a 64-point grid -- through the one-workgroup kernels (default for n <= 256) and through
the blocked multi-launch path (small_n = 0) on the same inputs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd
from gp_amd.synth import synth
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
ctx.reserve(2048)
sizes = [int(a) for a in sys.argv[1:]] or [21, 64, 79, 128, 129, 199, 256, 512, 1000, 2048]
for n in sizes:
    X, y = synth(n, 3)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
    G = 64
    out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
    rho = 0.3 * (1.0 + 0.01 * (np.arange(G) % 16)); sig = 0.1 * np.ones(G)
    row = {}
    for small in ((256, 0) if n <= 256 else (0,)):
        ctx.set_option("small_n", small)
        ctx.set_option("small_n1", small)
        ctx.set_option("small_m", 640 if small else 0)
        ctx.logml(X, y, 1.0, [0.3], 0.1)
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.logml(X, y, 1.0, [0.3], 0.1)
        host = (time.perf_counter() - t0) / 200
        # device-resident single evaluations, back to back (launch pipeline full): device time per evaluation
        for r in range(2):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(200):
                ctx.logml_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), 1.0, [0.3], 0.1, 0.0, out.data_ptr(), info.data_ptr())
            ctx.sync()
            one = (time.perf_counter() - t0) / 200
        best = 1e9
        for r in range(6):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(G), rho, sig, 0.0, out.data_ptr(), info.data_ptr())
            ctx.sync()
            if r:
                best = min(best, (time.perf_counter() - t0) / G)
        row[small] = (host, one, best)
    ctx.set_option("small_n", 256); ctx.set_option("small_n1", 128); ctx.set_option("small_m", 160)
    for small, (host, one, best) in row.items():
        print("n=%5d  %-14s host-buffer call %7.1f us   device-resident %7.1f us/eval   grid of 64: %6.2f us/eval"
              % (n, "one workgroup" if small else "blocked path", host * 1e6, one * 1e6, best * 1e6), flush=True)
