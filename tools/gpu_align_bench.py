"""Throughput of the overlap aligner: n overlaps of ~L x L bases at the given divergence."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vechat_amd.align import align_pairs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
div = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
rng = random.Random(3)
def mut(s):
    out = bytearray()
    for c in s:
        r = rng.random()
        if r < div * 0.3: continue
        if r < div * 0.7: out.append(rng.choice(b"ACGT"))
        out.append(rng.choice(b"ACGT") if div * 0.7 <= r < div else c)
    return bytes(out)
base = [bytes(rng.choice(b"ACGT") for _ in range(L)) for _ in range(16)]
pairs = [(mut(base[k % 16]), base[k % 16]) for k in range(n)]
cells = sum(len(q) * len(t) for q, t in pairs)
align_pairs(pairs[:64])
for rep in range(2):
    t0 = time.time(); cg, d = align_pairs(pairs); dt = time.time() - t0
    print(f"rep {rep}: {n} overlaps {L}x{L} div {div}: {dt:.3f}s = {n/dt:.0f} overlaps/s, {cells/dt/1e12:.2f} TCUPS (wall, incl. H2D/D2H and CIGAR text), mean distance {sum(d)/n:.0f}", flush=True)
