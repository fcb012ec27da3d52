import os, sys, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vechat_amd import capi
from vechat_amd.engine import HipContext
import oracle_api as oa
parts = [capi.synth_batch(capi.synth_cfg(50 + i, L, D, frac_partial=fp), 0, 2)
         for i, (L, D, fp) in enumerate([(80, 1, 0), (300, 30, 0.2), (64, 2, 0), (150, 3, 0.5), (500, 9, 0)])]
wins, fl = [], []
for p in parts:
    for w in range(p.n_windows):
        wins.append(p.window(w)); fl.append(int(p.win_fasta[w]))
batch = capi.Batch.from_windows(wins, fl, presorted=True)
for streams in (1, 2, 4):
    ctx = HipContext(device=0, n_streams=streams)
    print("streams", streams, "submit", flush=True)
    ctx.submit(batch); print(ctx.stats()["chunk_windows"], flush=True)
    ctx.run(); print("run ok", flush=True)
    ctx.sync(); print("sync ok", flush=True)
    cons, st = ctx.collect()
    ref, pol, _ = oa.oracle_run(batch, ctx.params)
    print([int(x) for x in st], [c == r for c, r in zip(cons, ref)], flush=True)
    ctx.close()
