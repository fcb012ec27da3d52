#!/usr/bin/env python3
"""Development: where a chunk stream's wall time goes, from a rocprofv3 --kernel-trace CSV -- per queue the time inside each kernel class and
the gaps between consecutive kernels of the queue (middle 60 % of the trace).   usage: stream_path.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def cls(n):
    n = n.replace("void ", "")
    for k in ("k_fwd_wide", "k_fwd_dt", "k_fwd", "k_tracew", "k_trace", "k_addaln", "k_rows_sub", "k_prune_lcc", "k_topo", "k_resolve", "k_addw", "k_avg", "k_init", "k_lag", "k_finish"):
        if n.startswith(k): return k
    return "other"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo, hi = t0 + (t1 - t0) * 0.2, t0 + (t1 - t0) * 0.8
byq = collections.defaultdict(list)
for e in ev:
    if lo <= e[0] <= hi: byq[e[3]].append(e)
tot = collections.Counter(); cnt = collections.Counter()
gap_after = collections.Counter()
wall = 0.0
for q, L in byq.items():
    if len(L) < 50: continue
    wall += L[-1][1] - L[0][0]
    for a, b in zip(L, L[1:]):
        tot[a[2]] += a[1] - a[0]; cnt[a[2]] += 1
        g = max(0, b[0] - a[1])
        tot["gap"] += g; gap_after[a[2]] += g
print(f"{len(byq)} queues, summed stream wall {wall/1e6:.1f} ms")
for k, v in tot.most_common():
    print(f"   {k:12s} {v/1e6:9.1f} ms {100*v/wall:5.1f} %   {cnt[k]:7d} launches  avg {v/max(cnt[k],1)/1e3:8.1f} us")
print("gaps by the kernel in front of them:", {k: round(v / 1e6, 1) for k, v in gap_after.most_common(8)})
