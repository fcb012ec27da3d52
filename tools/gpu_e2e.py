"""PCIe-inclusive rate: host batches in, consensus bytes out, with two contexts on two host threads so that the H2D
copy and validation of one batch (vc_submit) overlap the kernels of the other.  usage: gpu_e2e.py [windows_per_batch] [batches]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vechat_amd import capi
from vechat_amd.engine import HipContext
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = capi.synth_cfg(1002, 500, 64)
batches = [capi.synth_batch(cfg, k * n, n) for k in range(2)]          # two distinct host batches, reused
def worker(ctx, batch, reps, out):
    for _ in range(reps):
        ctx.submit(batch); ctx.run(); ctx.sync()
        cons, st = ctx.collect()
        out.append(sum(len(x) for x in cons))
for mode in ("one context, serial", "two contexts, overlapped"):
    ctxs = [HipContext(device=0) for _ in range(1 if mode.startswith("one") else 2)]
    for c, b in zip(ctxs, batches):
        c.submit(b); c.run(); c.sync()                                   # warm-up (allocations)
    outs = [[] for _ in ctxs]
    t0 = time.time()
    th = [threading.Thread(target=worker, args=(c, batches[i], nb // len(ctxs), outs[i])) for i, c in enumerate(ctxs)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.time() - t0
    done = sum(len(o) for o in outs) * n
    print(f"{mode}: {done} windows in {dt:.2f}s = {done/dt:.0f} windows/s end to end (submit + run + collect)", flush=True)
    for c in ctxs: c.close()
