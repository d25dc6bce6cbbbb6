"""Shader clock and cycles per SYRK workgroup: inside a full evaluation vs stand-alone launches of the
same kernel (why is the in-situ trailing update ~9 % slower per tile than gpmi_probe_syrk?)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")
import gp_amd
from gp_amd.synth import synth
n = 16384
ctx = gp_amd.Context(0); ctx.reserve(n)
ctx.set_option("grid_lanes", 1)
X, y = synth(n, 3)
dev = torch.device("cuda:0")
dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
out = torch.zeros((3, 3), dtype=torch.float64, device=dev); info = torch.zeros(3, dtype=torch.int32, device=dev)
def ev():
    ctx.logml_grid_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), np.ones(1), 0.3 * np.ones(1), 0.1 * np.ones(1), 0.0, out.data_ptr(), info.data_ptr())
    ctx.sync(); torch.cuda.synchronize()
for opts in ("", "nb_outer:16384"):
    ev(); ctx.probe_clock(True)
    t0 = time.perf_counter(); ev(); dt = time.perf_counter() - t0
    print("in-situ evaluation %.2f ms: clock %.0f MHz, %.0f cycles per SYRK workgroup, %d workgroups" % ((dt * 1e3,) + ctx.probe_clock(True)), flush=True)
    break
for m, k in ((15360, 1024), (14336, 1024), (8192, 1024), (14336, 512)):
    ctx.probe_syrk(m, k, 2); ctx.probe_clock(True)
    ms, tf = ctx.probe_syrk(m, k, 5)
    print("stand-alone m=%d k=%d: %.3f ms %.1f TF: clock %.0f MHz, %.0f cycles per workgroup, %d workgroups" % ((m, k, ms, tf) + ctx.probe_clock(True)), flush=True)
print("mfma issue probe: %.1f TF at %.0f MHz" % ctx.probe_mfma_peak(40000))
ev(); ctx.probe_clock(True); ev()
print("in-situ again: clock %.0f MHz, %.0f cycles per SYRK workgroup, %d workgroups" % ctx.probe_clock(True))
