"""Covariance-build kernels alone (HIP events around back-to-back launches, output resident in HBM):
k_se_cov<D> for D <= 8, the LDS-tiled k_se_cov_big for D > 8 (R's QQard takes any D, R/kernels.R:11-19),
lower triangle of an N x N matrix as the factorisation consumes it, and the joint [y, y'] build of c5.
GB/s = algorithmic bytes (8 B per stored element) / time; VALU instructions per element are static
counts from the ISA."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
from gp_amd._lib import LOWER
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
s = torch.cuda.Stream(dev); ctx.set_stream(s.cuda_stream)
n = 16384
K = torch.empty((n, n + 32), dtype=torch.float64, device=dev)   # column-major n x n with ld = n + 32
ld = n + 32
def timed(fn, reps=8):
    with torch.cuda.stream(s):
        fn(); fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            fn()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for D in (1, 3, 8, 9, 16, 32, 64):
    X, _ = synth(n, D)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    ell = [0.3 * np.sqrt(D / 3.0)]
    for flags, name, nbytes in ((LOWER, "lower", 4.0 * n * (n + 1)), (0, "full ", 8.0 * n * n)):
        ms = timed(lambda: ctx.se_cov_dev(dX.data_ptr(), n, n, 0, n, n, D, 1.0, ell, 0.01, flags, K.data_ptr(), ld))
        print("se_cov D=%2d %s N=%d: %.4f ms  %6.0f GB/s (%.0f%% of 8 TB/s)" % (D, name, n, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80.0), flush=True)
# joint build through the c5 evaluation's own timer (kernel_timing brackets the build launch)
nj = 8192
t = torch.from_numpy(np.linspace(0, 10, nj)).to(dev); yy = torch.from_numpy(np.concatenate([np.sin(np.linspace(0, 10, nj)), np.cos(np.linspace(0, 10, nj))])).to(dev)
out = torch.zeros(3, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
ctx.set_option("kernel_timing", 1)
for rep in range(3):
    ctx.kernel_timing(reset=True)
    with torch.cuda.stream(s):
        ctx.joint_logml_dev(t.data_ptr(), nj, yy.data_ptr(), 1.0, 0.5, 0.1, 1e-6, out.data_ptr(), info.data_ptr())
    torch.cuda.synchronize()
    k = ctx.kernel_timing(reset=True)["build"]
    print("joint_cov n=%d (order %d, lower): %.4f ms  %.0f GB/s" % (nj, 2 * nj, k[1] / k[0], k[2] / k[1] / 1e6), flush=True)
