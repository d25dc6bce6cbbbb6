"""Bisect helper: the bench.py c3 timed region, minimal, with switches."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
opts = {"sync": "0", "rho": "bench", "setopts": "0", "warm": "4", "G": "12", "reps": "1"}
for a in sys.argv[1:]:
    k, v = a.split("="); opts[k] = v
n = 16384; G = int(opts["G"]); warm = int(opts["warm"])
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
ctx.reserve(n)
for kv in opts["setopts"].split(","):
    if ":" in kv:
        k_, v_ = kv.split(":"); ctx.set_option(k_, int(v_))
X, y = synth(n, 3)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
npts = G + warm
out = torch.zeros((npts, 3), dtype=torch.float64, device=dev); info = torch.zeros(npts, dtype=torch.int32, device=dev)
if opts["rho"] == "bench":
    rho = 0.3 * (1.0 + 0.01 * (np.arange(npts) % 16))
else:
    rho = np.concatenate([0.3 * np.ones(warm), 0.3 * (1 + 0.01 * np.arange(G))])
sig = 0.1 * np.ones(npts)
def run(lo, hi):
    ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(hi - lo), rho[lo:hi], sig[lo:hi], 0.0, out[lo].data_ptr(), info[lo:].data_ptr())
for rep in range(int(opts["reps"])):
    if warm:
        run(0, warm)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(warm, npts)
    if int(opts["sync"]):
        ctx.sync()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    print("%s: %.2f ms/eval" % (opts, 1e3 * dt / G), flush=True)
if int(opts.get("check", "0")):
    res = out.cpu().numpy()
    ctx.set_option("grid_lanes", 1); ctx.set_option("sched", 0)
    for g in (warm, warm + 5, npts - 1):
        v = ctx.logml(X, y, 1.0, [rho[g]], sig[g])
        print("check g=%d grid=%.10f single=%.10f %s" % (g, res[g, 0], v[0], "OK" if res[g, 0] == v[0] else "DIFF"))
