"""Does replaying one evaluation as a HIP graph (captured through torch) shorten the launch gaps of
the panel chain?  Direct enqueue vs graph replay, one evaluation at a time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
dev = torch.device("cuda:0")
ctx = gp_amd.Context(0)
for n in (4096, 8192, 16384):
    ctx.reserve(n)
    X, y = synth(n, 3)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev); dy = torch.from_numpy(y).to(dev)
    out = torch.zeros(3, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream(dev); ctx.set_stream(s.cuda_stream)
    def run():
        ctx.logml_dev(dX.data_ptr(), n, n, 3, dy.data_ptr(), 1.0, [0.3], 0.1, 0.0, out.data_ptr(), info.data_ptr())
    def timeit(f, reps=8):
        f(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best * 1e3
    with torch.cuda.stream(s):
        t_direct = timeit(run)
        ref = out.clone()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
                run()
            t_graph = timeit(g.replay)
            print("n=%6d direct %.3f ms  graph replay %.3f ms  same result: %s" % (n, t_direct, t_graph, bool(torch.equal(ref, out))), flush=True)
        except Exception as e:
            print("n=%6d direct %.3f ms  graph capture failed: %r" % (n, t_direct, e), flush=True)
            break
