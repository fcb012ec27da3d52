"""development: one small batch through the persistent build pipeline, then the pipeline's hand-over words"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vechat_amd import capi
from vechat_amd.engine import HipContext
import oracle_api as oa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
b = capi.synth_batch(capi.synth_cfg(1001, 500, D), 0, n)
print("layer-1 lengths", [int(b.seq_off[int(b.win_seq_off[w]) + 2] - b.seq_off[int(b.win_seq_off[w]) + 1]) for w in range(n)])
ctx = HipContext(device=0)
try:
    cons, status = ctx.consensus(b)
    print("status", [int(x) for x in status], "errinfo", ctx.errinfo())
    ref, pol, st = oa.oracle_run(b, ctx.params)
    print("mismatches", sum(1 for w in range(n) if cons[w] != ref[w]))
except Exception as e:
    print("EXC", e)
out = (C.c_uint32 * (16 + 4 * n))()
ctx.lib.vc_debug_pipe_state.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
rc = ctx.lib.vc_debug_pipe_state(ctx.h, out, n)
o = list(out)
print("rc", rc, "ctl fq_res/head tq_res/head rq_res/head active done finished abort:", o[:10])
for i in range(n):
    l, e, t, p = o[10 + 4 * i: 14 + 4 * i]
    print(f"  window {i}: layer {l} job_end row {e >> 16} col {e & 0xFFFF} type {t} npairs {p}")
