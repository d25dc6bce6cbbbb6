"""A few one-at-a-time evaluations with given options (name:value,...) -- the subject of a rocprofv3
kernel trace:  rocprofv3 --kernel-trace -d DIR -- python3 tools/la_run.py 16384 lookahead:1"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = gp_amd.Context(0); ctx.reserve(n)
for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 else []):
    k, v = kv.split(":"); ctx.set_option(k, int(v))
ctx.set_option("grid_lanes", 1)
X, y = synth(n, 3)
dev = torch.device("cuda:0")
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
out = torch.zeros((3, 3), dtype=torch.float64, device=dev); info = torch.zeros(3, dtype=torch.int32, device=dev)
for r in range(3):
    ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(1), 0.3 * np.ones(1), 0.1 * np.ones(1), 0.0, out[r].data_ptr(), info[r:].data_ptr())
    torch.cuda.synchronize()
print(out[:, 0].cpu().numpy())
