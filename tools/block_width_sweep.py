"""Outer-block width policy (nb_thr1024 / nb_thr512 / nb_thr256: the order of the matrix still to update from which a
block is 1024 / 512 / 256 columns wide) for ONE evaluation at a time and for the grid on the lanes: the panel chain of a
single evaluation wants narrow blocks at the end, concurrent lanes hide the chain and want wide ones (fewer C epilogues)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_amd
from gp_amd.synth import synth
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [8192, 16384]
POL = [(8192, 4608, 3584), (6144, 3072, 2048), (4096, 2048, 1024), (3072, 2048, 1024), (2048, 1024, 512), (4096, 4096, 1024), (2048, 2048, 2048), (1024, 1024, 1024)]
for n in sizes:
    X, y = synth(n, 3)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
    G = 16
    out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
    rho = 0.3 * (1.0 + 0.01 * (np.arange(G) % 16)); sig = 0.1 * np.ones(G)
    for pol in POL:
        for k, v in zip(("nb_thr1024", "nb_thr512", "nb_thr256"), pol):
            ctx.set_option(k, v)
        res = []
        for lanes in (1, 0):
            ctx.set_option("grid_lanes", lanes)
            best = 1e9
            for r in range(4):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(G), rho, sig, 0.0, out.data_ptr(), info.data_ptr())
                ctx.sync()
                if r:
                    best = min(best, (time.perf_counter() - t0) / G)
            res.append(best)
        print("n=%5d thresholds %-20s one at a time %7.3f ms   on the lanes %7.3f ms per evaluation   logml %.9f"
              % (n, pol, res[0] * 1e3, res[1] * 1e3, float(out[0, 0])), flush=True)
