"""Print a per-kernel timeline (start offset, duration, stream/queue) from a rocprofv3 kernel trace CSV."""
import csv, glob, os, sys
d = sys.argv[1]; lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0; n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print(rows[0].keys())
for r in rows[lo:lo + n]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    nm = r["Kernel_Name"]
    for k in ("k_gemm_nt", "k_gemm8", "k_gemm9", "k_potrf_diag", "k_trsm_panel", "k_se_cov", "k_logml", "k_set_row"):
        if k in nm:
            nm = k + nm.split(k)[1][:8]
            break
    print("%10.1f %10.1f %8.1f us  q=%s grid=%s  %s" % (s, e, e - s, r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), nm[:40]))
