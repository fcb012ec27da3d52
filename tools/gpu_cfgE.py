"""BASELINE config E shape: ONT-profile 1 kb windows x 128 reads, parity on a few windows + throughput."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vechat_amd import capi
from vechat_amd.engine import HipContext
import oracle_api as oa
cfg = capi.synth_cfg(1005, 1000, 128, profile=capi.ONT)
b = capi.synth_batch(cfg, 0, 3)
ctx = HipContext(device=0, profile=1)
cons, st = ctx.consensus(b)
ref, pol, ost = oa.oracle_run(b, ctx.params)
print("parity", [c == r for c, r in zip(cons, ref)], [int(x) for x in st], ctx.errinfo(), "maxN", ost.max_nodes, "maxE", ost.max_edges, ctx.stats()["max_nodes"], ctx.stats()["max_edges"], flush=True)
ctx.close()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
b = capi.synth_batch(cfg, 0, n)
ctx = HipContext(device=0, profile=1)
ctx.submit(b)
for rep in range(2):
    t0 = time.time(); ctx.run(); ctx.sync(); dt = time.time() - t0
    s = ctx.stats()
    print(f"rep {rep}: {n/dt:.1f} win/s cells={s['cells']:.3e} GCUPS={s['cells']/dt/1e9:.1f} NC={s['max_nodes']} EC={s['max_edges']} CW={s['chunk_windows']}x{s['n_streams']} ms={ {k: round(v['ms'],1) for k,v in s['kernels'].items()} }", flush=True)
cons, st = ctx.collect()
import collections
print("status", collections.Counter(int(x) for x in st), [e for e in ctx.errinfo() if e != (0, 0)][:4])
