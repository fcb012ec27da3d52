import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    if not (k.startswith("k_fwd<8, 10, 6") or k.startswith("k_tracew")): continue
    v.sort(); n = len(v)
    small = [x for x in v if x < 100]
    print(f"{k:36s} n {n:6d}  launches under 100 us: {len(small):6d} mean {sum(small)/max(len(small),1):7.1f} us  p50 {v[n//2]:8.1f}  mean {sum(v)/n:8.1f}")
    qs = [v[int(n*q)] for q in (0.05,0.1,0.2,0.3,0.4,0.45,0.5,0.55,0.6)]
    print("   quantiles 5..60 %:", [round(x,1) for x in qs])
