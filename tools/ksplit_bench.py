"""SYRK launch time with and without the K-split of tail-round tiles (option ksplit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
ctx = gp_amd.Context(0)
ctx.set_option("stagger", (2 << 16) | 4)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tot = {0: 0.0, 1: 0.0}
for m in range(15360, 0, -K):
    r = {}
    for ks in (0, 1):
        ctx.set_option("ksplit", ks)
        ctx.probe_syrk(m, K, 2)
        t, tf = ctx.probe_syrk(m, K, 6)
        r[ks] = (t, tf); tot[ks] += t
    T = (m + 127) // 128; nt = T * (T + 1) // 2
    print("m=%6d tiles=%5d rounds=%6.2f  off %7.3f ms %6.2f TF   on %7.3f ms %6.2f TF" % (m, nt, nt / 512.0, r[0][0], r[0][1], r[1][0], r[1][1]), flush=True)
print("sum off %.3f ms, on %.3f ms" % (tot[0], tot[1]))
