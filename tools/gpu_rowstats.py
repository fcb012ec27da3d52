"""Development: statistics of the DP rows k_fwd walks in the build phase -- predecessor distances, rows a later row reads back
from farther than one row, hit rates of ring designs.   usage: gpu_rowstats.py [windows=16]"""
import ctypes as C, os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vechat_amd import capi
from vechat_amd.engine import HipContext
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = capi.synth_batch(capi.synth_cfg(1002, 500, 64), 0, n)
ctx = HipContext(device=0, n_streams=1, chunk_windows=n)
ctx.submit(batch)
f = ctx.lib.vc_debug_rows
f.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
for layer in (4, 16, 32, 48, 62):
    ctx.lib.vc_debug_stop_after(ctx.h, 1, layer)
    ctx.run(); ctx.sync()
    hist = collections.Counter(); npred = collections.Counter(); rtype = collections.Counter()
    rows_tot = keep_tot = 0
    hits = {k: [0, 0] for k in (4, 5, 6, 8, 10, 12)}           # plain ring of k rows: [hits, reads]
    khits = {k: [0, 0] for k in (3, 4, 5, 6, 8)}               # ring of k KEPT rows
    for w in range(n):
        buf = np.zeros(4 * 8192, np.uint32); nr = C.c_uint32(0)
        rc = f(ctx.h, w, buf.ctypes.data_as(C.POINTER(C.c_uint32)), 8192, C.byref(nr))
        assert rc == 0, ctx.lib.vc_last_error(ctx.h)
        R = nr.value
        rec = buf[:4 * R].reshape(R, 4)
        keep = np.zeros(R + 1, bool)
        preds = []
        for r in range(R):
            x = int(rec[r, 0]); fl = (x >> 8) & 0xFF; k = (x >> 16) & 0xFF
            if fl & 4: preds.append([]); continue             # overflow list: skip
            d = [int(rec[r, 1]) & 0xFFFF, int(rec[r, 1]) >> 16, int(rec[r, 2]) & 0xFFFF, int(rec[r, 2]) >> 16, int(rec[r, 3]) & 0xFFFF, int(rec[r, 3]) >> 16][:k]
            d = [v for v in d if v <= r]                        # drop the virtual row
            preds.append(d); npred[len(d)] += 1
            rtype[('prev' if 1 in d else 'noprev', len([v for v in d if v != 1]), 'sink' if fl & 1 else '')] += 1
            for v in d:
                hist[min(v, 20)] += 1
                if v >= 2: keep[r - v] = True
        kix = np.concatenate([[0], np.cumsum(keep[:R])])      # kept rows before row r
        rows_tot += R; keep_tot += int(keep[:R].sum())
        for r in range(R):
            for v in preds[r]:
                if v < 2: continue
                for k in hits: hits[k][1] += 1; hits[k][0] += v <= k
                between = kix[r] - kix[r - v]                 # kept rows in [r-v, r): the pred itself is the oldest of them
                for k in khits: khits[k][1] += 1; khits[k][0] += between <= k
    tot = sum(hist.values())
    print(f"layer {layer}: rows/window {rows_tot / n:.0f}  kept rows {100 * keep_tot / rows_tot:.1f} %  preds/row {tot / rows_tot:.2f}  non-adjacent reads/row {sum(v for k, v in hist.items() if k >= 2) / rows_tot:.3f}")
    print("   distance histogram %:", {k: round(100 * v / tot, 1) for k, v in sorted(hist.items())})
    print("   row types % (row above a predecessor?, other predecessors, sink):", {k: round(100 * v / rows_tot, 1) for k, v in sorted(rtype.items(), key=lambda kv: -kv[1])[:12]})
    print("   in-degree %:", {k: round(100 * v / rows_tot, 1) for k, v in sorted(npred.items())})
    print("   plain ring hit % of non-adjacent reads:", {k: round(100 * a / max(b, 1), 2) for k, (a, b) in hits.items()})
    print("   kept-row ring hit %:", {k: round(100 * a / max(b, 1), 2) for k, (a, b) in khits.items()}, flush=True)
