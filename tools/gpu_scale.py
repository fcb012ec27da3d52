"""Throughput probe: n windows of one config through the HIP path, per-kernel event times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vechat_amd import capi
from vechat_amd.engine import HipContext
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = int(sys.argv[3]) if len(sys.argv) > 3 else 500
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
streams = int(sys.argv[5]) if len(sys.argv) > 5 else 0
fp = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
t0 = time.time()
b = capi.synth_batch(capi.synth_cfg(1002, L, D, frac_partial=fp), 0, n)
print(f"generated {n} windows in {time.time()-t0:.1f}s, {b.bases.size/1e6:.1f} MB bases", flush=True)
ctx = HipContext(device=0, profile=1, chunk_windows=chunk, n_streams=streams, num_prune=int(os.environ.get('VC_NUM_PRUNE', '3')),
                 scratch_bytes=int(float(os.environ.get('VC_SCRATCH_GB', '0')) * (1 << 30)))
t0 = time.time(); ctx.submit(b); ts = time.time() - t0
print(f"vc_submit (validation + H2D of {2*b.bases.size/1e6:.0f} MB): {ts:.3f}s = {n/ts:.0f} win/s", flush=True)
acc = {"cells": 0, "dp_rows": 0, "trace_steps": 0, "command": "python tools/gpu_scale.py " + " ".join(sys.argv[1:])}
for rep in range(2):
    t0 = time.time(); ctx.run(); ctx.sync(); t1 = time.time()
    s = ctx.stats()
    for k in ("cells", "dp_rows", "trace_steps"):
        acc[k] += s[k]
    km = {k: round(v['ms'], 1) for k, v in s['kernels'].items()}
    print(f"rep {rep}: {n/(t1-t0):.1f} win/s ({t1-t0:.3f}s) cells={s['cells']:.3e} GCUPS={s['cells']/(t1-t0)/1e9:.1f} rows={s['dp_rows']:.3e} far={s['far_row_reads']} redo={s['band_redo']} trace steps/spec/rounds={s['trace_steps']}/{s['trace_spec']}/{s['trace_rounds']} NC={s['max_nodes']} EC={s['max_edges']} CW={s['chunk_windows']}x{s['n_streams']} dev={s.get('device_bytes', 0)/2**30:.1f}GiB ms={km}", flush=True)
cons, status = ctx.collect()
import collections
print("status histogram", collections.Counter(int(x) for x in status), "errinfo sample", [e for e in ctx.errinfo() if e != (0, 0)][:5])
if os.environ.get("VC_STATS_JSON"):
    import json
    json.dump(acc, open(os.environ["VC_STATS_JSON"], "w"))
