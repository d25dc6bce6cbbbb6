"""Timeline of ONE evaluation's launches from a rocprofv3 kernel trace of
     rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --grid-lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-c4
   per outer block: duration of the panel phase (everything between two multi-round SYRK launches), its launches by kind,
   gaps between launches; then the first block's launches one by one."""
import csv, glob, os, re, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", nm)[:20]
def wgs(r):   # workgroups of a launch (grid sizes are in work-items)
    g = 1
    for ax in ("X", "Y", "Z"):
        g *= max(1, int(r.get("Grid_Size_" + ax, "1")) // max(1, int(r.get("Workgroup_Size_" + ax, "1"))))
    return g
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), wgs(r)) for r in rows]
# the last evaluation of the largest size: from its covariance build to its finalize kernel
big = max(e[3] for e in ev if e[2].startswith("k_se_cov") or e[2].startswith("k_joint_cov"))
st = max(i for i, e in enumerate(ev) if (e[2].startswith("k_se_cov") or e[2].startswith("k_joint_cov")) and e[3] == big)
en = min(i for i, e in enumerate(ev) if i > st and e[2].startswith("k_logml_finalize"))
ev = ev[st:en + 1]
t0 = ev[0][0]
print("evaluation: %d launches, %.3f ms" % (len(ev), (ev[-1][1] - t0) / 1e6))
blocks = []; cur = []
for e in ev[1:]:
    if e[2].startswith("k_gemm_nt<1>") and e[3] > 600:   # multi-round trailing update: ends a panel phase
        blocks.append((cur, e)); cur = []
    else:
        cur.append(e)
tot_panel = tot_gap = 0.0
for bi, (pan, syrk) in enumerate(blocks):
    if not pan:
        continue
    dur = (pan[-1][1] - pan[0][0]) / 1e3
    busy = sum(e[1] - e[0] for e in pan) / 1e3
    kinds = {}
    for e in pan:
        k = kinds.setdefault(e[2], [0, 0.0]); k[0] += 1; k[1] += (e[1] - e[0]) / 1e3
    tot_panel += dur; tot_gap += dur - busy
    print("block %2d: panel phase %7.1f us (%d launches, %5.1f us in gaps)  %s   | SYRK %7.1f us, %d workgroups" % (
        bi, dur, len(pan), dur - busy, "  ".join("%s x%d %.0f" % (k, v[0], v[1]) for k, v in sorted(kinds.items())), (syrk[1] - syrk[0]) / 1e3, syrk[3]))
print("panel phases (blocks followed by a multi-round update): %.2f ms, of which gaps %.2f ms" % (tot_panel / 1e3, tot_gap / 1e3))
print("tail (after the last multi-round update): %d launches, %.2f ms" % (len(cur), (cur[-1][1] - cur[0][0]) / 1e6 if cur else 0.0))
print("first block, launch by launch:")
for e in blocks[0][0]:
    print("  %9.1f  %7.1f us  %-20s %5d workgroups" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
