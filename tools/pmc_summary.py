"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name (sum over dispatches)."""
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, {c: (f"{v:.4g}", n[(k, c)]) for c, v in d.items()})
