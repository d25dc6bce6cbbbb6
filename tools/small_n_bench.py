"""Small-N regime (the sizes the reference's own scripts run: N = 21 ... 1000): latency of one host-buffer
evaluation (gpmi_logml: upload, build, factor, 3 doubles back) and throughput of a 64-point grid."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")
import gp_amd
from gp_amd.synth import synth
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
ctx.reserve(2048)
for n in (21, 64, 128, 256, 512, 1000, 2048):
    X, y = synth(n, 3)
    ctx.logml(X, y, 1.0, [0.3], 0.1)
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.logml(X, y, 1.0, [0.3], 0.1)
    host = (time.perf_counter() - t0) / 50
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
    G = 64
    out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
    rho = 0.3 * (1.0 + 0.01 * (np.arange(G) % 16)); sig = 0.1 * np.ones(G)
    res = {}
    for lanes in (1, 0):
        ctx.set_option("grid_lanes", lanes)
        best = 1e9
        for r in range(4):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(G), rho, sig, 0.0, out.data_ptr(), info.data_ptr())
            torch.cuda.synchronize(dev)
            if r:
                best = min(best, (time.perf_counter() - t0) / G)
        res[lanes] = best
    print("n=%5d  host-buffer call %7.1f us   grid of 64: %7.1f us/eval one at a time, %7.1f us/eval on lanes"
          % (n, host * 1e6, res[1] * 1e6, res[0] * 1e6), flush=True)
