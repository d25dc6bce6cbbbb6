import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
G = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = gp_amd.Context(0)
if len(sys.argv) > 4:
    ctx.set_option("lookahead", int(sys.argv[4]))
if len(sys.argv) > 5:
    ctx.set_option("nb_outer", int(sys.argv[5]))
X, y = synth(n, 3)
dev = torch.device("cuda:0")
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
rho = 0.3 * (1 + 0.01 * np.arange(G)); sig = 0.1 * np.ones(G)
for lanes in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1", "2", "3"])]:
    ctx.set_option("grid_lanes", lanes)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(G), rho, sig, 0.0, out.data_ptr(), info.data_ptr())
        ctx.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("N=%d lanes=%d: %.2f ms/eval (%.1f evals/s) logml0=%.8g" % (n, lanes, 1e3 * dt / G, G / dt, out[0, 0].item()), flush=True)
