"""Quick parity probe on the GPU box: HIP path vs the oracle on a few synthetic configurations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vechat_amd import capi
from vechat_amd.engine import HipContext
import oracle_api as oa

cases = [
    ("tiny 60x4", capi.synth_cfg(1, 60, 4), 4),
    ("tiny 60x4 fasta", capi.synth_cfg(2, 60, 5, fastq=0, backbone_fastq=0), 4),
    ("200x12 partial", capi.synth_cfg(3, 200, 12, frac_partial=0.3), 8),
    ("500x32", capi.synth_cfg(1001, 500, 32), 6),
    ("500x64", capi.synth_cfg(1002, 500, 64), 6),
    ("500x40 2hap partial", capi.synth_cfg(13, 500, 40, n_haplotypes=2, snp_rate=0.02, frac_partial=0.2), 6),
]
only = sys.argv[1:]
allok = True
for name, cfg, n in cases:
    if only and not any(o in name for o in only):
        continue
    b = capi.synth_batch(cfg, 0, n)
    ctx = HipContext(device=0, profile=1)
    t0 = time.time()
    try:
        cons, status = ctx.consensus(b)
    except Exception as e:
        print(name, "EXCEPTION", e); allok = False; continue
    t1 = time.time()
    ref, pol, st = oa.oracle_run(b, ctx.params)
    s = ctx.stats()
    nbad = 0
    ei = ctx.errinfo()
    for w in range(n):
        ok = cons[w] == ref[w] and int(status[w]) == (0 if pol[w] else 1)
        if not ok:
            nbad += 1
            # first difference
            d = next((i for i, (x, y) in enumerate(zip(cons[w], ref[w])) if x != y), min(len(cons[w]), len(ref[w])))
            print(f"   window {w}: status={int(status[w])} len hip={len(cons[w])} oracle={len(ref[w])} first diff at {d} errinfo={ei[w]}")
    allok = allok and nbad == 0
    km = {k: round(v['ms'], 2) for k, v in s['kernels'].items()}
    print(f"{name}: {n} windows, mismatches={nbad}, hip {t1-t0:.2f}s cells hip={s['cells']} oracle={st.cells} NC={s['max_nodes']} EC={s['max_edges']} ms={km}")
    ctx.close()
print("ALL OK" if allok else "MISMATCHES")
