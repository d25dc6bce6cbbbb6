"""Timeline of the LAST evaluation in a rocprofv3 kernel trace: per outer block, when the bulk update
(SYRK on the bulk queue), the block-column update and the panel kernels ran, and how much of the
panel phase was covered by a running bulk update."""
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
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in rows]
# last evaluation: from the last k_se_cov on
last = max(i for i, e in enumerate(ev) if e[2] == "k_se_cov")
ev = ev[last:]
t0 = ev[0][0]; t1 = max(e[1] for e in ev)
print("evaluation span %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(ev)))
byq = {}
for s, e, n, q, g in ev:
    byq.setdefault(q, []).append((s, e, n, g))
for q, l in sorted(byq.items()):
    kinds = {}
    for s, e, n, g in l:
        k = kinds.setdefault(n, [0, 0]); k[0] += 1; k[1] += e - s
    print("queue %s: busy %.2f ms " % (q, sum(e - s for s, e, _, _ in l) / 1e6), {k: (v[0], round(v[1] / 1e6, 2)) for k, v in kinds.items()})
mode = sys.argv[2] if len(sys.argv) > 2 else "syrk"
if mode == "syrk":
    for s, e, n, q, g in ev:
        if n == "k_gemm_nt<1":
            print("%9.1f -> %9.1f  %8.1f us  q=%s grid=%d" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, g))
else:
    lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0; hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
    for s, e, n, q, g in ev:
        if lo <= (s - t0) / 1e3 <= hi:
            print("%9.1f -> %9.1f  %8.1f us  q=%s grid=%-6d %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, g, n))
