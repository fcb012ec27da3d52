"""development: where does the host-to-host path spend its time?  one context in series vs two contexts on two threads"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vechat_amd import capi
from vechat_amd.engine import HipContext
n, B = 65536, 16384
batch = capi.synth_batch(capi.synth_cfg(1002, 500, 64), 0, n)
parts = [batch.slice(lo, lo + B) for lo in range(0, n, B)]
free_b, _ = torch.cuda.mem_get_info(0)
def series(ctx, idx, tag):
    for i in idx:
        t0 = time.perf_counter(); ctx.submit(parts[i]); t1 = time.perf_counter(); ctx.run(); t2 = time.perf_counter(); ctx.sync(); t3 = time.perf_counter(); ctx.collect(); t4 = time.perf_counter()
        print(f"{tag} batch {i}: submit {t1-t0:.3f} run {t2-t1:.3f} sync {t3-t2:.3f} collect {t4-t3:.3f}", flush=True)
c0 = HipContext(device=0, scratch_bytes=int(0.42 * free_b)); c1 = HipContext(device=0, scratch_bytes=int(0.42 * free_b))
series(c0, [0], "warm0"); series(c1, [1], "warm1")
t0 = time.perf_counter(); series(c0, range(4), "one"); dt = time.perf_counter() - t0
print(f"one context in series: {n/dt:.0f} windows/s")
t0 = time.perf_counter()
th = [threading.Thread(target=series, args=(c, range(k, 4, 2), f"thr{k}")) for k, c in enumerate((c0, c1))]
[t.start() for t in th]; [t.join() for t in th]
dt = time.perf_counter() - t0
print(f"two contexts, two threads: {n/dt:.0f} windows/s")
