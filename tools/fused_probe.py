"""Where block 0 of the fused in-block launches (sub-tile of the next diagonal block + its factorisation)
spends its cycles: sub-tile product, wait for the two sibling sub-tiles, diagonal-block body."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")
import gp_amd
from gp_amd.synth import synth
for n in (8192, 16384):
    ctx = gp_amd.Context(0); ctx.reserve(n); ctx.set_option("grid_lanes", 1)
    X, y = synth(n, 3)
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
    out = torch.zeros((3, 3), dtype=torch.float64, device=dev); info = torch.zeros(3, dtype=torch.int32, device=dev)
    for r in range(2):
        ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(1), 0.3 * np.ones(1), 0.1 * np.ones(1), 0.0, out.data_ptr(), info.data_ptr())
        ctx.sync(); torch.cuda.synchronize()
        f = ctx.probe_fused()
    print("n=%d: %d fused launches (avg K %.0f): sub-tile %.0f cycles, sibling wait %.0f, diagonal body %.0f  (2.3 GHz: %.1f / %.1f / %.1f us)"
          % (n, f[3], f[4] / f[3], f[0] / f[3], f[1] / f[3], f[2] / f[3], f[0] / f[3] / 2300, f[1] / f[3] / 2300, f[2] / f[3] / 2300))
    ctx.close()
