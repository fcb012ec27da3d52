"""Lifecycle check: repeated create / submit (changing shapes) / run / destroy must not leak device memory and must
keep giving oracle-identical bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vechat_amd import capi
from vechat_amd.engine import HipContext
import oracle_api as oa

def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20

base = free_mb()
shapes = [(100, 5, 6), (500, 20, 10), (250, 40, 4), (800, 12, 3), (500, 20, 10)]
hist = []
for rep in range(10):
    c = HipContext(device=0, n_streams=1 + rep % 3, chunk_windows=[0, 4, 7, 0][rep % 4])
    for k, (L, D, n) in enumerate(shapes):
        b = capi.synth_batch(capi.synth_cfg(100 + rep * 10 + k, L, D, frac_partial=0.2), 0, n)
        cons, st = c.consensus(b)
        ref, pol, _ = oa.oracle_run(b, c.params)
        assert cons == ref and [int(x) == 0 for x in st] == [bool(p) for p in pol], (rep, k)
        c.run(); c.sync()                                 # a second run of the same batch
        cons2, _ = c.collect()
        assert cons2 == ref
    c.close()
    hist.append(free_mb())
    print(f"rep {rep}: free {hist[-1]:.0f} MB (start {base:.0f})", flush=True)
# the runtime keeps a few hundred MB of its own after first use; what must not happen is a steady per-context loss
assert max(hist[4:]) - min(hist[4:]) < 16, "device memory leaked"
print("lifecycle ok")
