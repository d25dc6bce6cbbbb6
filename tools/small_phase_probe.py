"""Where a one-workgroup small-N evaluation spends its cycles (probe build: s_memtime stamps at the phase
boundaries of k_logml_small, thread 0)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPMI_USE_PROBES"] = "1"
import gp_amd
from gp_amd.synth import synth
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
ctx.set_option("small_n1", 256)
for n in [int(a) for a in sys.argv[1:]] or [21, 64, 128, 199, 256]:
    X, y = synth(n, 3)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
    out = torch.zeros(3, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(3):
        ctx.logml_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), 1.0, [0.3], 0.1, 0.0, out.data_ptr(), info.data_ptr())
    ctx.sync(); ctx.probe_small()
    R = 50
    for _ in range(R):
        ctx.logml_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), 1.0, [0.3], 0.1, 0.0, out.data_ptr(), info.data_ptr())
    ctx.sync()
    p = ctx.probe_small()
    c = p / max(p[3], 1)
    print("n=%4d  cycles per evaluation: build %7.0f  diagonal blocks %7.0f  rows below %7.0f  trailing tiles %7.0f  finalize %7.0f  (sum %.1f us at 2.4 GHz)"
          % (n, c[0], c[1], c[2], c[4], c[5], (c[0] + c[1] + c[2] + c[4] + c[5]) / 2400.0), flush=True)
