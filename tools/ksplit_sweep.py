import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import gp_amd
from gp_amd.synth import synth
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
for n in (2048, 3072, 4096, 8192):
    X, y = synth(n, 3)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
    G = 8
    out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
    rho = 0.3 * np.ones(G); sig = 0.1 * np.ones(G)
    for km in (100, 0, 200, 300, 400, 512):
        ctx.set_option("ksplit", km)
        res = []
        for lanes in (1, 0):
            ctx.set_option("grid_lanes", lanes)
            best = 1e9
            for r in range(4):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(G), rho, sig, 0.0, out.data_ptr(), info.data_ptr())
                ctx.sync()
                if r: best = min(best, (time.perf_counter() - t0) / G)
            res.append(best)
        print("n=%5d ksplit_max %4d: one at a time %7.3f ms, lanes %7.3f ms   %.9f" % (n, km, res[0]*1e3, res[1]*1e3, float(out[0,0])), flush=True)
