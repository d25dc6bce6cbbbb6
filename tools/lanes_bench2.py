import os, sys, time, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
opts = {"n": "16384", "G": "12", "lanes": "4", "nbo": "512", "order": "0", "stg": "131076", "la": "0", "res": "8", "lanela": "0", "reserve": "0"}
for a in sys.argv[1:]:
    k, v = a.split("="); opts[k] = v
L = lambda k: [int(x) for x in opts[k].split(",")]
n = int(opts["n"]); G = int(opts["G"])
ctx = gp_amd.Context(0)
if int(opts.get("reserve", "0")):
    ctx.reserve(n)
ctx.set_option("cu_reserve", int(opts["res"]))
ctx.set_option("lane_lookahead", int(opts["lanela"]))
X, y = synth(n, 3)
dev = torch.device("cuda:0")
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
rho = 0.3 * (1 + 0.01 * np.arange(G)); sig = 0.1 * np.ones(G)
for lanes, nbo, order, stg, la in itertools.product(L("lanes"), L("nbo"), L("order"), L("stg"), L("la")):
    ctx.set_option("grid_lanes", lanes); ctx.set_option("nb_outer", nbo); ctx.set_option("syrk_order", order)
    ctx.set_option("stagger", stg); ctx.set_option("lookahead", la)
    best = 1e9
    for rep in range(3):
        if int(opts.get("warm4", "0")):
            ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(4), rho[:4], sig[:4], 0.0, out.data_ptr(), info.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(G), rho, sig, 0.0, out.data_ptr(), info.data_ptr())
        ctx.sync(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("N=%d lanes=%d nbo=%d order=%d stg=%d la=%d: %.2f ms/eval (%.1f evals/s)" % (n, lanes, nbo, order, stg, la, 1e3 * best / G, G / best), flush=True)
