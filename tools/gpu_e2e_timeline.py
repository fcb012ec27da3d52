"""development: where the host-to-host rate of one context loses against the resident rate -- per batch: seconds in vc_submit, vc_run and vc_collect
(a collect that returns at once means the device ran dry).   python tools/gpu_e2e_timeline.py [windows] [batch] [first_batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vechat_amd import capi
from vechat_amd.engine import HipContext

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
first = int(sys.argv[3]) if len(sys.argv) > 3 else bsz
b = capi.synth_batch(capi.synth_cfg(1002, 500, 64), 0, n)
sizes = [min(first, n)]
rem = (n - sizes[0]) % bsz
if rem:
    sizes.append(rem)                      # the odd remainder early (it runs beside full batches), full batches to the end
sizes += [bsz] * ((n - sum(sizes)) // bsz)
cuts = [0]
for k in sizes:
    cuts.append(cuts[-1] + k)
parts = [b.slice(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]
ctx = HipContext(device=0)
if os.environ.get("VC_WARM", "1") != "0":          # VC_WARM=0: rep 0 is a cold start (device start-up, workspaces made under the first batches)
    ctx.submit(parts[-1]); ctx.run(); ctx.submit(parts[0]); ctx.run(); ctx.collect(); ctx.collect()
for rep in range(2):
    ev = []
    t0 = time.perf_counter()
    def T(): return time.perf_counter() - t0
    a = T(); ctx.submit(parts[0]); s = T(); ctx.run(); r = T()
    ev.append((0, parts[0].n_windows, s - a, r - s, 0.0))
    for i in range(1, len(parts)):
        a = T(); ctx.submit(parts[i]); s = T(); ctx.run(); r = T(); ctx.collect(); c = T()
        ev.append((i, parts[i].n_windows, s - a, r - s, c - r))
    a = T(); ctx.collect(); c = T()
    ev.append((-1, 0, 0, 0, c - a))
    dt = T()
    print(f"rep {rep}: {n / dt:.0f} windows/s  ({dt:.3f} s; batches {len(parts)} of {bsz}, first {first})")
    for e in ev:
        print("   batch %3d n=%6d submit %.3f run %.3f collect-wait %.3f" % e)
# the same windows resident
ctx.close()
ctx = HipContext(device=0)
ctx.submit(b); ctx.run(); ctx.sync()
t0 = time.perf_counter(); ctx.run(); ctx.sync(); dt = time.perf_counter() - t0
print(f"resident: {n / dt:.0f} windows/s")
ctx.close()
