"""Summarise a rocprofv3 --kernel-trace csv per (kernel, grid size): calls, total ms, average us."""
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[(k, g)]
    a[0] += 1; a[1] += d
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (k, g), (n, t) in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{k:30s} grid {g:>9s} calls {n:5d} total {t/1e3:9.1f} ms avg {t/n:9.1f} us")
