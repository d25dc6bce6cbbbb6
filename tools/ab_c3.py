"""c3 throughput (24-point grid on the lanes) and one-at-a-time time of whatever libgpmi.so is in place -- tolerant of an older
library (symbols added later are skipped): the measuring half of a same-box A/B (tools/ab_libs.sh swaps the libraries)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
from gp_amd import _lib
probe = ctypes.CDLL(_lib.LIB_PATH)
_lib.SYMBOLS = tuple(s for s in _lib.SYMBOLS if hasattr(probe, s))
import gp_amd
from gp_amd.synth import synth
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
X, y = synth(n, 3)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
G = 24
out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
rho = 0.3 * (1.0 + 0.01 * (np.arange(G) % 16)); sig = 0.1 * np.ones(G)
res = []
for lanes, g in ((0, G), (1, 4)):
    ctx.set_option("grid_lanes", lanes)
    best = 1e9
    for r in range(4):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(g), rho[:g], sig[:g], 0.0, out.data_ptr(), info.data_ptr())
        ctx.sync()
        if r:
            best = min(best, (time.perf_counter() - t0) / g)
    res.append(best)
print("n=%d: %.3f ms per evaluation on the lanes (%.2f evals/s), %.3f ms one at a time" % (n, res[0] * 1e3, 1.0 / res[0], res[1] * 1e3), flush=True)
