"""approx_Lz throughput: fused Hermite blend x vector on a device-resident table (HBM-bound)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
ctx = gp_amd.Context(0)
dev = torch.device("cuda:0")
for n in (2048, 4096, 8192):
    rng = np.random.default_rng(0)
    lp = np.array([0.5, 1.0, 1.5])
    Ls = [np.tril(rng.standard_normal((n, n))) for _ in lp]
    ctx.interp_load(lp, Ls, Ls)
    z = torch.from_numpy(rng.standard_normal(n)).to(dev); f = torch.zeros(n, dtype=torch.float64, device=dev)
    for _ in range(3):
        ctx.approx_Lz_dev(0.8, z.data_ptr(), f.data_ptr())
    ctx.sync(); torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.approx_Lz_dev(0.8, z.data_ptr(), f.data_ptr())
    ctx.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    alg = 4 * 8 * n * (n + 1) / 2
    print("n=%5d  %8.1f us  %7.1f GB/s algorithmic (4 lower triangles)" % (n, 1e6 * dt, alg / dt / 1e9), flush=True)
    t0 = time.perf_counter(); L = ctx.approx_L(0.8); dt = time.perf_counter() - t0
    print("         approx_L to host: %.1f ms" % (1e3 * dt))
    ctx.interp_free()
