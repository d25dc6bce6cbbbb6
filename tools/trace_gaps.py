"""Summarise a rocprofv3 kernel trace of a grid run: per-queue busy time, union of the
trailing-update (SYRK) intervals, gaps between consecutive SYRKs."""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    for k in ("k_gemm_nt<1", "k_gemm_nt<0", "k_potrf_diag", "k_trsm_panel", "k_se_cov", "k_logml", "k_set_row", "k_cal"):
        if k in nm:
            return k
    return nm[:24]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
# timed region: from the last big idle gap (> 5 ms) onwards
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 5e6:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]; t1 = max(e[1] for e in ev)
print("region: %d kernels, span %.2f ms" % (len(ev), (t1 - t0) / 1e6))
byq = {}
for s, e, n, q in ev:
    byq.setdefault(q, []).append((s, e, n))
for q, l in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in l)
    kinds = {}
    for s, e, n in l:
        kinds[n] = kinds.get(n, 0) + 1
    print("queue %s: %5d kernels, busy %.2f ms  %s" % (q, len(l), busy / 1e6, kinds))
sy = sorted((s, e) for s, e, n, q in ev if n == "k_gemm_nt<1")
u = 0; cs, ce = sy[0]
gaps = []
for s, e in sy[1:]:
    if s > ce:
        u += ce - cs; gaps.append((s - ce, ce)); cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
print("SYRK union busy %.2f ms of %.2f (%.1f%%); sum of durations %.2f ms" % (u / 1e6, (t1 - t0) / 1e6, 100.0 * u / (t1 - t0), sum(e - s for s, e in sy) / 1e6))
gaps.sort(reverse=True)
print("largest SYRK-idle gaps (us @ offset ms):", ["%.0f@%.1f" % (g / 1e3, (at - t0) / 1e6) for g, at in gaps[:12]])
print("total idle in gaps %.2f ms over %d gaps" % (sum(g for g, _ in gaps) / 1e6, len(gaps)))
