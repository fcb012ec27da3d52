#!/usr/bin/env python3
"""Kernel resource table from hipcc -Rpass-analysis=kernel-resource-usage (stdin or file): name, VGPRs, SGPR spills, occupancy, LDS."""
import re, subprocess, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
cur = None; rows = []
for line in txt.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    s = m.group(1).strip()
    if s.startswith("Function Name:") or s.startswith("Name:"):
        cur = {"name": s.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in s:
        k, v = s.split(":", 1); cur[k.strip()] = v.strip()
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    print(f"{d:42s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr_spill {r.get('SGPRs Spill','?'):>4s} vgpr_spill {r.get('VGPRs Spill','?'):>3s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?'):>6s} scratch {r.get('ScratchSize [bytes/lane]','?')}")
