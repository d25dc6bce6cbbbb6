"""Where a 128-pivot diagonal-block body (potrf_diag4_body) spends its cycles inside a real evaluation (probe build)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPMI_USE_PROBES"] = "1"
import gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [2048, 16384]:
    X, y = synth(n, 3)
    ctx.logml(X, y, 1.0, [0.3], 0.1); ctx.probe_body(); ctx.probe_fused()
    ctx.logml(X, y, 1.0, [0.3], 0.1)
    b = ctx.probe_body(); f = ctx.probe_fused()
    k = max(b[5], 1)
    print("n=%5d: %d bodies; per body: loads %6.0f  loop %6.0f  stores %6.0f cycles (sum %.1f us at 2.3 GHz); factor wave: factor16 %6.0f, waiting %6.0f"
          % (n, b[5], b[0] / k, b[1] / k, b[2] / k, (b[0] + b[1] + b[2]) / k / 2300.0, b[3] / k, b[4] / k))
    if f[3] > 0:
        print("         fused launches, block 0: sub-tile %6.0f  wait %6.0f  body %6.0f cycles per launch" % (f[0] / f[3], f[1] / f[3], f[2] / f[3]))
    if f[7] > 0:
        print("         of these the %d K = 128 leaf launches (body on the critical path): sub-tile %6.0f  body %6.0f cycles" % (f[7], f[6] / f[7], f[5] / f[7]))
