"""Where the one-workgroup value + gradient kernel spends its cycles (probe build: s_memtime stamps, thread 0 of block 0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPMI_USE_PROBES"] = "1"
import gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [21, 64, 128, 199, 256]:
    X, y = synth(n, 1)
    for _ in range(3):
        ctx.logml_grad(X, y, 1.0, [0.3], 0.1)
    ctx.probe_fused()
    for _ in range(50):
        ctx.logml_grad(X, y, 1.0, [0.3], 0.1)
    p = np.asarray(ctx.probe_fused(), dtype=float)
    c = p / max(p[3], 1)
    print("n=%4d  cycles: build + identity %7.0f  diagonal blocks %7.0f  strips (rows below + U) %7.0f  tiles (trailing + U) %7.0f  "
          "value + a = U z %7.0f  K^-1 tiles + contraction %7.0f  trees %7.0f  (sum %.1f us at 2.4 GHz)"
          % (n, c[0], c[1], c[2], c[4], c[5], c[6], c[7], (c[0] + c[1] + c[2] + c[4] + c[5] + c[6] + c[7]) / 2400.0), flush=True)
