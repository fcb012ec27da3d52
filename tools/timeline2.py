#!/usr/bin/env python3
"""Development: how the chunk streams overlap, from a rocprofv3 --kernel-trace CSV.
Prints (second half of the trace): time by number of k_fwd launches resident at once; time by the set of kernel classes
resident; average k_fwd launch duration by what ran beside it.   usage: timeline2.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def cls(n):
    n = n.replace("void ", "")
    for k in ("k_fwd_wide", "k_fwd", "k_tracew", "k_tracet", "k_trace", "k_addaln", "k_rows_sub", "k_rows", "k_prune_lcc", "k_topo", "k_resolve", "k_addw", "k_avg", "k_build"):
        if n.startswith(k): return k
    return "other"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + (t1 - t0) * 0.5
ev = [e for e in ev if e[0] >= cut]
pts = []
for i, (s, e, n, q) in enumerate(ev):
    pts.append((s, 1, i)); pts.append((e, -1, i))
pts.sort()
live = set()
by_nfwd = collections.Counter(); by_set = collections.Counter()
fwd_share = collections.defaultdict(lambda: [0.0, 0.0])        # per fwd launch: time alone among fwd / with tails
last = pts[0][0]
for t, d, i in pts:
    dt = t - last
    if dt > 0 and live:
        names = [ev[j][2] for j in live]
        nf = sum(1 for x in names if x == "k_fwd")
        by_nfwd[nf] += dt
        by_set["+".join(sorted(set(names)))] += dt
    elif dt > 0:
        by_nfwd[-1] += dt
    if d > 0: live.add(i)
    else: live.discard(i)
    last = t
wall = pts[-1][0] - pts[0][0]
print(f"wall {wall/1e6:.1f} ms")
print("time by number of k_fwd launches resident (-1 = nothing running):")
for k in sorted(by_nfwd): print(f"   {k:2d}: {by_nfwd[k]/1e6:8.1f} ms  {100*by_nfwd[k]/wall:5.1f} %")
print("time by resident kernel classes (top 16):")
for k, v in by_set.most_common(16): print(f"   {v/1e6:8.1f} ms {100*v/wall:5.1f} %  {k}")
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in ev:
    a = agg[n]; a[0] += 1; a[1] += (e - s) / 1e6
print("launches:")
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1]): print(f"   {n:12s} {c:6d} launches {d:9.1f} ms sum {1e3*d/c:8.1f} us avg")
qs = collections.Counter(q for _, _, _, q in ev)
print("queues:", dict(qs))
