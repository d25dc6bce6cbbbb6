"""Expected strong scaling of the c4 grid (64 points, N=8192) from one GPU: time of the whole grid
vs the 8-point share one rank of an 8-GPU job gets (same call, no communication)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
n = 8192
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0); ctx.reserve(n)
for kv in sys.argv[1:]:
    k_, v_ = kv.split(":"); ctx.set_option(k_, int(v_))
X, y = synth(n, 3)
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
R, S = np.meshgrid(np.geomspace(0.1, 1.0, 8), np.geomspace(0.05, 0.5, 8), indexing="ij")
rho, sig = R.ravel(), S.ravel()
out = torch.zeros((64, 3), dtype=torch.float64, device=dev); info = torch.zeros(64, dtype=torch.int32, device=dev)
def run(idx):
    ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(len(idx)), rho[idx], sig[idx], 0.0, out.data_ptr(), info.data_ptr())
res = {}
for name, idx in (("64", np.arange(64)), ("32", np.arange(0, 64, 2)), ("16", np.arange(0, 64, 4)), ("8", np.arange(0, 64, 8))):
    best = 1e9
    for r in range(4):
        torch.cuda.synchronize(dev); t0 = time.perf_counter(); run(idx); torch.cuda.synchronize(dev)
        if r: best = min(best, time.perf_counter() - t0)
    res[name] = best
    print("%2s points: %.2f ms (%.3f ms/point)" % (name, best * 1e3, best * 1e3 / len(idx)), flush=True)
print("expected speed-up at 2/4/8 GPUs (no comm): %.2f %.2f %.2f" % (res["64"] / res["32"], res["64"] / res["16"], res["64"] / res["8"]))
