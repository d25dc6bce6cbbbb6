"""HBM write ceiling next to the covariance build: torch fill / copy of 2 GiB vs k_se_cov (events)."""
import torch, time
n = 16384
a = torch.empty(n * n, dtype=torch.float64, device="cuda")
b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.5)); print("fill 2.147 GB: %.3f ms = %.0f GB/s" % (ms, a.numel() * 8 / ms / 1e6))
ms = t(lambda: a.zero_()); print("memset 2.147 GB: %.3f ms = %.0f GB/s" % (ms, a.numel() * 8 / ms / 1e6))
ms = t(lambda: b.copy_(a)); print("copy 2.147 GB: %.3f ms = %.0f GB/s (read+write)" % (ms, 2 * a.numel() * 8 / ms / 1e6))
