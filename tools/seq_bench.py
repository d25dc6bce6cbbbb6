"""Sequential sampler (gpmi_seq_*): set-up and per-step wall time.  (Parity of the R/tests.R:78 scenario against the
CPU restatement is a test: tests/test_gpu_seq.py::test_seq_sampler_reference_scenario.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
from gp_amd import synth

ctx = gp_amd.default_context(0)
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
