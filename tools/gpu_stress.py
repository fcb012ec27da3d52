"""Randomised parity sweep: many small batches of random shape through the HIP path and the oracle.
usage: gpu_stress.py [n_cases] [seed]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vechat_amd import capi
from vechat_amd.engine import HipContext
import oracle_api as oa

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
unsupported = 0
t0 = time.time()
ctxs = {}
for case in range(n_cases):
    L = rng.choice([60, 120, 250, 400, 500, 500, 640, 800, 1000])
    D = rng.choice([3, 5, 8, 12, 20, 32, 48])
    n = rng.choice([4, 8, 16])
    mode = rng.choice([0, 0, 0, 1])
    kw = dict(frac_partial=rng.choice([0, 0, 0.2, 0.5]), n_haplotypes=rng.choice([1, 1, 2, 3]), snp_rate=rng.choice([0.005, 0.02]),
              fastq=rng.choice([0, 1, 1]), backbone_fastq=rng.choice([0, 1, 1]), profile=rng.choice([capi.PACBIO, capi.ONT]))
    if kw["fastq"] == 0 and kw["backbone_fastq"] == 1:
        pass
    pk = dict(mode=mode, num_prune=rng.choice([1, 2, 3, 3, 4]), min_confidence=rng.choice([0.2, 0.2, 0.1, 0.3]),
              min_support=rng.choice([0.2, 0.2, 0.15]), trim=rng.choice([0, 1]))
    if rng.random() < 0.15:
        pk.update(match=5, mismatch=-4, gap=-8)            # raw (unpacked) rows
    key = tuple(sorted(pk.items()))
    if key not in ctxs:
        ctxs[key] = HipContext(device=0, **pk)
    c = ctxs[key]
    seed = rng.randrange(1, 1 << 30)
    batch = capi.synth_batch(capi.synth_cfg(seed, L, D, **kw), 0, n)
    cons, status = c.consensus(batch)
    ref, pol, st = oa.oracle_run(batch, c.params)
    # windows the device path declines (e.g. the reference's int32 score rule, reported VC_WIN_UNSUPPORTED) are counted, not compared
    uns = [w for w in range(n) if int(status[w]) == capi.VC_WIN_UNSUPPORTED]
    unsupported += len(uns)
    mism = sum(1 for w in range(n) if w not in uns and (cons[w] != ref[w] or (int(status[w]) == capi.VC_WIN_OK) != bool(pol[w])))
    retried = bool(uns) or any(e != (0, 0) for e in c.errinfo())          # an overflow retry reruns part of the batch: work counters differ
    if mism or (not retried and c.stats()["cells"] != st.cells):
        bad += 1
        print(f"CASE {case} MISMATCH: seed={seed} L={L} D={D} n={n} kw={kw} pk={pk} mism={mism} status={[int(x) for x in status]} err={[e for e in c.errinfo() if e != (0, 0)][:3]}", flush=True)
print(f"{n_cases} cases, {bad} bad, {unsupported} windows reported unsupported, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
