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
b = capi.synth_batch(capi.synth_cfg(int(os.environ.get("VC_SEED", "1002")), L, D, frac_partial=fp, profile=capi.ONT if os.environ.get("VC_PROFILE") == "ont" else capi.PACBIO), 0, n)
print(f"generated {n} windows in {time.time()-t0:.1f}s, {b.bases.size/1e6:.1f} MB bases", flush=True)
_sc = [int(x) for x in os.environ['VC_SCORES'].split(',')] if os.environ.get('VC_SCORES') else None      # e.g. VC_SCORES=5,-4,-8: scores whose rows stay raw int16
ctx = HipContext(device=0, profile=1, chunk_windows=chunk, n_streams=streams, num_prune=int(os.environ.get('VC_NUM_PRUNE', '3')), **(dict(match=_sc[0], mismatch=_sc[1], gap=_sc[2]) if _sc else {}),
                 scratch_bytes=int(float(os.environ.get('VC_SCRATCH_GB', '0')) * (1 << 30)))
t0 = time.time(); ctx.submit(b); ts = time.time() - t0
print(f"vc_submit (validation + H2D of {2*b.bases.size/1e6:.0f} MB): {ts:.3f}s = {n/ts:.0f} win/s", flush=True)
acc = {"cells": 0, "dp_rows": 0, "trace_steps": 0, "windows": 0, "command": "python tools/gpu_scale.py " + " ".join(sys.argv[1:])}
for rep in range(2):
    t0 = time.time(); ctx.run(); ctx.sync(); t1 = time.time()
    s = ctx.stats()
    for k in ("cells", "dp_rows", "trace_steps"):
        acc[k] += s[k]
    acc["windows"] += n
    km = {k: round(v['ms'], 1) for k, v in s['kernels'].items()}
    print(f"rep {rep}: {n/(t1-t0):.1f} win/s ({t1-t0:.3f}s) cells={s['cells']:.3e} GCUPS={s['cells']/(t1-t0)/1e9:.1f} rows={s['dp_rows']:.3e} far={s['far_row_reads']} redo={s['band_redo']} trace steps/spec/rounds={s['trace_steps']}/{s['trace_spec']}/{s['trace_rounds']} NC={s['max_nodes']} EC={s['max_edges']} CW={s['chunk_windows']}x{s['n_streams']} dev={s.get('device_bytes', 0)/2**30:.1f}GiB ms={km}", flush=True)
if os.environ.get("VC_PIPE") == "1":
    import ctypes as C
    pp = (C.c_uint64 * 144)()
    ctx.lib.vc_debug_pipe_prof.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    ctx.lib.vc_debug_pipe_prof(ctx.h, pp)
    v = list(pp)
    ms = lambda x: x / 1e5          # ticks of 100 MHz -> ms
    if v[5] and v[11] and v[4] and v[10]:
        print(f"pipe F waves={v[5]} items={v[4]}: per wave wait {ms(v[0])/v[5]:.1f} add {ms(v[1])/v[5]:.1f} fwd {ms(v[2])/v[5]:.1f} hand {ms(v[3])/v[5]:.1f} ms; per item add {ms(v[1])/v[4]*1e3:.0f} fwd {ms(v[2])/v[4]*1e3:.0f} hand {ms(v[3])/v[4]*1e3:.0f} us", flush=True)
        print(f"pipe T waves={v[11]} rounds={v[9]} items={v[10]} ties={v[12]}: per wave wait {ms(v[6])/v[11]:.1f} walk {ms(v[7])/v[11]:.1f} hand {ms(v[8])/v[11]:.1f} ms; per round walk {ms(v[7])/v[9]*1e3:.0f} hand {ms(v[8])/v[9]*1e3:.0f} us; items per round {v[10]/v[9]:.2f}", flush=True)
    if os.environ.get("VC_PIPE") == "1":
        tl = lambda k: [round(x / 1e5) for x in v[16 + 32 * k: 16 + 32 * k + 32]]
        print(f"tie resolution: {ms(v[15])/max(v[12],1)*1e3:.0f} us per tie, {ms(v[15])/max(v[11],1):.1f} ms per T wave")
        print(f"pickup latency (publish -> consumer has it): F {ms(v[13])/max(v[4],1)*1e3:.0f} us per item, T {ms(v[14])/max(v[10],1)*1e3:.0f} us per item")
        print("timeline (20 ms buckets since wave start, summed over waves and chunks)")
        print("  F wait ms ", tl(0)); print("  F busy ms ", tl(1))
        print("  T rounds  ", v[16 + 64: 16 + 96]); print("  T items   ", v[16 + 96: 16 + 128])
cons, status = ctx.collect()
import collections
print("status histogram", collections.Counter(int(x) for x in status), "errinfo sample", [e for e in ctx.errinfo() if e != (0, 0)][:5])
if os.environ.get("VC_STATS_JSON"):
    import json
    json.dump(acc, open(os.environ["VC_STATS_JSON"], "w"))
