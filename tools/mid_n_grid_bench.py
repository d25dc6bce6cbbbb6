"""Grids at the mid sizes (n = 300 ... 1438; the reference's westbrook.R runs N = 1438): per-evaluation time of a G-point grid
through the one-workgroup-per-point kernel (every CU a problem of its own) against the four lanes of the blocked path."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd
from gp_amd.synth import synth
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [300, 384, 512, 640, 768, 1024]
for n in sizes:
    X, y = synth(n, 3)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
    for G in (8, 16, 32, 64, 128, 256, 512):
        out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
        rho = 0.3 * (1.0 + 0.01 * (np.arange(G) % 16)); sig = 0.1 * np.ones(G)
        res = {}
        for name, force in (("one workgroup per point", True), ("lanes", False)):
            if name == "lanes" and G > 64:
                continue
            for k, v in (("small_n2", 1024 if force else 0), ("small_g2", 0)):
                ctx.set_option(k, v)
            best = 1e9
            for r in range(4):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(G), rho, sig, 0.0, out.data_ptr(), info.data_ptr())
                ctx.sync()
                if r:
                    best = min(best, (time.perf_counter() - t0))
            res[name] = (best, float(out[G // 2, 0]))
        a = res["one workgroup per point"]
        b = res.get("lanes")
        print("n=%5d G=%4d  one workgroup per point %9.1f us (%7.2f us/eval)%s" % (n, G, a[0] * 1e6, a[0] * 1e6 / G,
              "   lanes %9.1f us (%7.2f us/eval)  rel diff %.1e" % (b[0] * 1e6, b[0] * 1e6 / G, abs(a[1] - b[1]) / abs(b[1])) if b else ""), flush=True)
