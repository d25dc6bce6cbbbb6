"""A/B helper: ms per evaluation of the logml path (grid entry point) for several N, one at a time
(lanes=1) and with the default lanes, under option settings given as name:value[,name:value...]
groups.   python tools/ab_bench.py "nb_adapt:0" "nb_adapt:1" [ns=4096,8192,16384] [reps=3]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
groups = [a for a in sys.argv[1:] if "=" not in a] or ["none:0"]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
ns = [int(v) for v in kw.get("ns", "4096,8192,16384").split(",")]
reps = int(kw.get("reps", "3"))
lanes_list = [int(v) for v in kw.get("lanes", "1,0").split(",")]
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
ctx.reserve(max(ns))
for n in ns:
    X, y = synth(n, 3)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
    G = 12 if n >= 12288 else 32
    out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
    rho = 0.3 * (1.0 + 0.01 * (np.arange(G) % 16)); sig = 0.1 * np.ones(G)
    for lanes in lanes_list:
        for grp in groups:
            for kv in grp.split(","):
                k_, v_ = kv.split(":")
                if k_ != "none":
                    ctx.set_option(k_, int(v_))
            ctx.set_option("grid_lanes", lanes)
            best = 1e9
            for r in range(reps + 1):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(G), rho, sig, 0.0, out.data_ptr(), info.data_ptr())
                torch.cuda.synchronize(dev)
                if r:
                    best = min(best, (time.perf_counter() - t0) / G)
            print("n=%6d lanes=%d %-28s %8.3f ms/eval  logml0=%.9f" % (n, lanes, grp, best * 1e3, out[0, 0].item()), flush=True)
