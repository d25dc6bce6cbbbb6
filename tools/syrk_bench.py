"""Stand-alone SYRK (trailing update) sweep: variant x K x m."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPMI_USE_PROBES", "1")  # tools run on the probe build (libgpmi_probes.so)
import gp_amd
ctx = gp_amd.Context(0)
ms = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["16128", "8192", "4096"])]
ks = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["128", "256", "512", "1024"])]
gvs = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1"])]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx.set_option("syrk_order", int(sys.argv[5]) if len(sys.argv) > 5 else 0)
staggers = [int(x) for x in (sys.argv[6].split(",") if len(sys.argv) > 6 else ["0"])]
for gv, stg in [(g, s_) for g in gvs for s_ in staggers]:
    ctx.set_option("gemm_variant", gv)
    ctx.set_option("stagger", stg)
    for m in ms:
        for k in ks:
            t, tf = ctx.probe_syrk(m, k, reps)
            print("stg=%6d gv=%d m=%6d k=%5d  %9.3f ms  %6.2f TFLOP/s" % (stg, gv, m, k, t, tf), flush=True)
