"""Single-evaluation look-ahead check: stream topology + time per evaluation with / without look-ahead."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
ctx.reserve(n)
ctx.set_option("grid_lanes", 1)
persist = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx.set_option("syrk_persist", persist)
dw = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx.set_option("diag_waves", dw)
if len(sys.argv) > 2 and sys.argv[2] == "tstream":
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
X, y = synth(n, 3)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
out = torch.zeros((8, 3), dtype=torch.float64, device=dev); info = torch.zeros(8, dtype=torch.int32, device=dev)
for la in (0, 1):
    for res in (8, 0) if la else (8,):
        ctx.set_option("cu_reserve", res)
        ctx.set_option("lookahead", la)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(4), 0.3 * np.ones(4), 0.1 * np.ones(4), 0.0, out.data_ptr(), info.data_ptr())
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("dw=%d " % dw, end=""); print("n=%d persist=%d lookahead=%d cu_reserve=%d: %.2f ms/eval" % (n, persist, la, res, 1e3 * dt / 4), flush=True)
