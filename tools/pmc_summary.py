"""Aggregate rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "k_fill" in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
