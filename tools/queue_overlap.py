"""Per-queue picture of a rocprofv3 kernel trace (the region behind the last idle gap > 20 ms): kernels and busy
time per hardware queue, time by kernel name, and how many queues are busy at once (histogram over time)."""
import csv, glob, os, re, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", nm)[:28]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]) + " g" + r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Queue_Id", "?")) for r in rows]
cut = 0
run_end = ev[0][1]
for i in range(1, len(ev)):
    if ev[i][0] - run_end > 20e6:
        cut = i
    run_end = max(run_end, ev[i][1])
ev = ev[cut:]
t0 = ev[0][0]; t1 = max(e[1] for e in ev)
span = (t1 - t0) / 1e6
print("region: %d kernels, span %.2f ms" % (len(ev), span))
byq, byk = {}, {}
for s, e, n, q in ev:
    byq.setdefault(q, [0, 0.0]); byq[q][0] += 1; byq[q][1] += (e - s) / 1e6
    byk.setdefault(n, [0, 0.0]); byk[n][0] += 1; byk[n][1] += (e - s) / 1e6
for q, (c, b) in sorted(byq.items()):
    print("queue %-4s %5d kernels, busy %8.2f ms (%.0f%% of the span)" % (q, c, b, 100 * b / span))
print("sum of kernel durations %.2f ms = %.2f x the span" % (sum(b for _, b in byq.values()), sum(b for _, b in byq.values()) / span))
for n, (c, b) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-40s %5d launches %8.2f ms  avg %7.1f us" % (n, c, b, 1e3 * b / c))
# concurrency histogram
pts = sorted([(s, 1) for s, e, _, _ in ev] + [(e, -1) for s, e, _, _ in ev])
hist = {}; cur = 0; last = t0
for t, dlt in pts:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += dlt
print("kernels in flight at once: " + ", ".join("%d: %.1f%%" % (k, 100.0 * v / (t1 - t0)) for k, v in sorted(hist.items())))

# per queue: idle time between consecutive kernels of the SAME queue (dependency / host-bound gaps)
lastq = {}; gapq = {}
for s_, e_, n_, q_ in ev:
    if q_ in lastq:
        gapq.setdefault(q_, []).append(max(0, s_ - lastq[q_]))
    lastq[q_] = e_
for q_, g in sorted(gapq.items()):
    g.sort()
    print("queue %-4s gaps between its kernels: median %.1f us, mean %.1f us, total %.2f ms" % (q_, g[len(g) // 2] / 1e3, sum(g) / len(g) / 1e3, sum(g) / 1e6))
if len(sys.argv) > 2:   # timeline window: first kernel index, count
    lo = int(sys.argv[2]); cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    for s_, e_, n_, q_ in ev[lo:lo + cnt]:
        print("%10.1f %8.1f us  q=%-3s %s" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, q_, n_))
# coarse activity map: per queue, busy fraction in bins of span / 60 (# >= 75 %, + >= 25 %, . > 0, blank idle)
nb = 60; bw = (t1 - t0) / nb
for q_ in sorted(byq):
    busy = [0.0] * nb
    for s_, e_, n_, qq in ev:
        if qq != q_:
            continue
        b0 = int((s_ - t0) / bw); b1 = min(nb - 1, int((e_ - t0) / bw))
        for b in range(b0, b1 + 1):
            lo_, hi_ = t0 + b * bw, t0 + (b + 1) * bw
            busy[b] += max(0.0, min(e_, hi_) - max(s_, lo_))
    print("queue %-4s |%s|" % (q_, "".join("#" if x >= 0.75 * bw else "+" if x >= 0.25 * bw else "." if x > 0 else " " for x in busy)))
