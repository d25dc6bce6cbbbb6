"""Sequential sampler (gpmi_seq_*): set-up and per-step wall time, and the error of the
R/tests.R:78 scenario against the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd import synth, ode_gp
from oracle import oracle as orc

ctx = gp_amd.default_context(0)
t = np.linspace(-2, 2, 21); f = np.exp(t)
p = ode_gp.p_dotXn(t, f, [1.0, 1.0], 0.05, joint=True, ctx=ctx)
ps = ode_gp.p_Xn(t, f, [1.0, 1.0], 0.05, joint=True, ctx=ctx)
X = ps["condMean"]
a = ode_gp.create_p_dotXnS([X], p["condMean"], p["condVar"], [1.0, 1.0], ctx=ctx)
b = orc.create_p_dotXnS([X], p["condMean"], p["condVar"], 1.0, 1.0)
for xs, z in zip([0.6, 1.0, 0.5, 0.1, 0.2, 1.2], [0.5, -0.3, 1.2, -0.8, 0.1, 0.9]):
    ra = a([xs], z=z); rb = b([xs], z)
    print("xs %.1f mu %.12f err %.2e  var %.3e err %.2e" % (xs, ra["mu"], ra["mu"] - rb["mu"], ra["sigma"], ra["sigma"] - rb["sigma"]))
for n in (4096, 16384):
    Xb, y = synth.synth(n, 3)
    Kn = np.asfortranarray(0.05 * np.eye(n)); mn = 0.5 * y
    t0 = time.time(); s = ctx.seq_sampler(Xb, mn, Kn, 1.2, [0.05, 0.06, 0.04], 1e-6, 64); t1 = time.time()
    s.close(); t0 = time.time(); s = ctx.seq_sampler(Xb, mn, Kn, 1.2, [0.05, 0.06, 0.04], 1e-6, 64); t1 = time.time()
    ts = []
    for k in range(20):
        q = time.time(); mu, v = s.step(Xb[k] + 0.003); s.commit(mu); ts.append(time.time() - q)
    print("n=%d create %.1f ms (incl. %.0f MB upload of Kn), step %.2f ms (median of 20)" % (n, (t1 - t0) * 1e3, n * n * 8 / 1e6, np.median(ts) * 1e3))
    s.close()
