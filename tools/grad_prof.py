import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd.synth import synth
ctx = gp_amd.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
X, y = synth(n, 3)
for rep in range(3):
    out, g = ctx.logml_grad(X, y, 1.0, [0.3], 0.1)
print(out, g)
