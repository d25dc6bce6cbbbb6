"""Does the leading dimension of the workspace matter for the trailing update?  Stand-alone SYRK launches
(m = 14336, K = 1024) with forced leading dimensions, and whole evaluations at N = 16384 with extra padding."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
os.environ["GPMI_USE_PROBES"] = "1"
import numpy as np, torch, gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
ctx.probe_syrk(14336, 1024, 2)
ms, tf = ctx.probe_syrk(14336, 1024, 5)
n = 16384
ctx.reserve(n); ctx.set_option("grid_lanes", 1)
X, y = synth(n, 3); dev = torch.device("cuda:0")
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
out = torch.zeros((3, 3), dtype=torch.float64, device=dev); info = torch.zeros(3, dtype=torch.int32, device=dev)
best = 1e9
for r in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(1), 0.3 * np.ones(1), 0.1 * np.ones(1), 0.0, out.data_ptr(), info.data_ptr())
    ctx.sync(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
ctx.set_option("grid_lanes", 4)
G = 8; out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
b4 = 1e9
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(G), 0.3 * np.ones(G), 0.1 * np.ones(G), 0.0, out.data_ptr(), info.data_ptr())
    ctx.sync(); torch.cuda.synchronize(); b4 = min(b4, (time.perf_counter() - t0) / G)
print("%%s: stand-alone %%.3f ms %%.1f TF | evaluation %%.2f ms | 4 lanes %%.2f ms/eval" %% (os.environ.get("TAG"), ms, tf, best * 1e3, b4 * 1e3), flush=True)
''' % ROOT
for tag, env in (("natural ld", {}), ("GPMI_LD=16416", {"GPMI_LD": "16416"}), ("GPMI_LD=16400", {"GPMI_LD": "16400"}), ("GPMI_LD=16448", {"GPMI_LD": "16448"}),
                 ("GPMI_LD=16512", {"GPMI_LD": "16512"}), ("GPMI_LD=16640", {"GPMI_LD": "16640"}), ("GPMI_LD=17408", {"GPMI_LD": "17408"}), ("GPMI_LD=16392", {"GPMI_LD": "16392"})):
    e = dict(os.environ, TAG=tag, **env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=280)
    print((r.stdout.strip() or r.stderr[-400:]), flush=True)
