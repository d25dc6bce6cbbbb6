"""Posterior (p_dotXn / sample_derivs moments) wall time through the host-buffer entry point."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
ctx = gp_amd.Context(0)
for n in (2048, 4096, 8192):
    t = np.linspace(0, 0.01 * n, n); y = np.sin(t)
    for rep in range(2):
        t0 = time.perf_counter()
        mn, Kn = ctx.gp_condition(t, t, y, 1.0, 0.5, 0.01, 1e-8, "QQ", "RQ", "RR")
        dt = time.perf_counter() - t0
    print("n=%5d (factor order %d, Schur block %d): %.1f ms incl. %.0f MB of Kn over PCIe; mean|mn|=%.3f" % (n, 2 * n, n, dt * 1e3, n * n * 8 / 1e6, np.abs(mn).mean()), flush=True)
