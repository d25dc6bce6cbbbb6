"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a small text summary
suitable for profiles/ (tracked)."""
import csv
import glob
import os
import sys


def main(d, out, title):
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not stats:
        sys.exit("no kernel_stats.csv under " + d)
    rows = list(csv.DictReader(open(stats[0])))
    with open(out, "w") as f:
        f.write("# %s\n# source: rocprofv3 --kernel-trace --stats (%s)\n" % (title, os.path.basename(stats[0])))
        f.write("%-72s %8s %12s %12s %12s %12s %8s\n" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
        for r in rows:
            f.write("%-72s %8s %12.3f %12.2f %12.2f %12.2f %8.2f\n" % (
                r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "kernel stats")
