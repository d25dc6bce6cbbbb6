"""SE covariance build inside full evaluations (N=16384, D=3): non-temporal vs cached stores,
alternating, per-launch HIP-event timing of the library (kernel_timing)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
n = 16384
ctx = gp_amd.Context(0); ctx.reserve(n)
X, y = synth(n, 3)
dev = torch.device("cuda:0")
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
out = torch.zeros((4, 3), dtype=torch.float64, device=dev); info = torch.zeros(4, dtype=torch.int32, device=dev)
ctx.set_option("grid_lanes", 1); ctx.set_option("kernel_timing", 1)
for rep in range(4):
    for nt in (0, 1):
        ctx.set_option("se_nt", nt)
        ctx.kernel_timing(reset=True)
        ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(4), 0.3 * np.ones(4), 0.1 * np.ones(4), 0.0, out.data_ptr(), info.data_ptr())
        torch.cuda.synchronize()
        k = ctx.kernel_timing(reset=True)["build"]
        print("nt=%d  %.4f ms  %.0f GB/s" % (nt, k[1] / k[0], k[2] / k[1] / 1e6), flush=True)
