"""Per-kernel means of rocprofv3 --pmc counters (+ dispatch durations from the kernel trace of the
same runs) -> JSON.  HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE reads
exactly half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, sys, collections
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k, cs in acc.items():
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e["launches_per_pass"] = len(next(iter(cs.values())))
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
    if k in dur:
        e["avg_us_profiled"] = sum(dur[k]) / len(dur[k])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
        # busy cycles summed over 1024 SIMDs / (per-XCD active cycles * 1024)
        e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in res.items():
    if "gemm" in k or "se_cov" in k or "joint_cov" in k or "potrf" in k or "trsm" in k:
        print(k[:60], {x: ("%.4g" % y) for x, y in e.items() if x in ("hbm_bytes_per_launch", "mfma_util", "avg_us_profiled", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")})
