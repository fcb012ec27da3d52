#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: wall, union-busy, time with k_fwd resident, time with only
latency-bound kernels resident, idle gaps.  usage: timeline.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows]
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + (t1 - t0) * skip                     # drop warmup part
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, n in ev:
    f = 1 if "k_fwd" in n else 0
    pts.append((s, 1, f)); pts.append((e, -1, -f))
pts.sort()
busy = fwd = other_only = 0
na = nf = 0
last = pts[0][0]
for t, d, f in pts:
    dt = t - last
    if na > 0: busy += dt
    if nf > 0: fwd += dt
    elif na > 0: other_only += dt
    na += d; nf += f; last = t
wall = t1 - t0
print(f"wall {wall/1e6:.1f} ms  busy {busy/1e6:.1f}  k_fwd resident {fwd/1e6:.1f}  only-others {other_only/1e6:.1f}  idle {(wall-busy)/1e6:.1f}")
agg = {}
for s, e, n in ev:
    a = agg.setdefault(n[:40], [0, 0]); a[0] += e - s; a[1] += 1
for n, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:14]:
    print(f"  {n:40s} {d/1e6:9.1f} ms  {c:6d} launches  {d/c/1e3:8.1f} us avg")
