"""Generates tests/golden/two_rounds.json.gz: the two-round VeChat run (scripts/vechat:283-397) as a parity fixture for row N4.
Runs ONLY in the build container: it reads /root/reference/example/ref.fa (DATA) and takes every per-window consensus from the
REAL reference compiled in place (oracle/_ref: window.cpp + spoa, both overloads).

  python tests/golden/make_two_rounds.py

Reads from two haplotypes of a genome segment, both strands, full-length and partial.  The overlapper is external in the reference
too (minimap2 | awk | fpa), so the fixture carries its output: all-vs-all overlaps with CIGARs (global unit-cost alignments computed
here) for round 1, and -- computed on the round-1 result -- for round 2.  Round 1 is the haplotype-aware pass
(`-f -p -d 0.2 -s 0.2`, fragment correction: the reads are their own targets), round 2 the linear pass (`-f`) on the corrected reads.
Expected texts: window assembly and stitching by this repository's builder (two restatements agree, tests/test_windows.py),
consensus of every window by the reference.  tests/test_driver.py::test_two_rounds_on_the_device_match_the_reference runs
`vechat_amd.driver` through the real polisher on the device and requires the identical final FASTA.
"""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

from vechat_amd import capi, seqio  # noqa: E402
from vechat_amd.windows import WindowBuilder  # noqa: E402
import oracle_api as oa  # noqa: E402
import windows_ref as wr  # noqa: E402
from make_plumbing import mutate, nw_cigar, read_contig  # noqa: E402


def overlaps_paf(reads, spans, min_len=500):
    """all-vs-all: reads = [(name, stored sequence, reverse flag)], spans = name -> (a, b) on the truth; PAF lines with cg:Z:"""
    lines = []
    for qn, qs, qf in reads:
        for tn, ts, tf in reads:
            if qn == tn:
                continue
            (qa, qb), (ta, tb) = spans[qn], spans[tn]
            lo, hi = max(qa, ta), min(qb, tb)
            if hi - lo < min_len:
                continue

            def stored(seq, flag, a, b, x0, x1):             # truth interval [x0, x1) -> interval of the stored sequence
                p0 = int(round((x0 - a) * len(seq) / (b - a))); p1 = int(round((x1 - a) * len(seq) / (b - a)))
                return (len(seq) - p1, len(seq) - p0) if flag else (p0, p1)
            q0, q1 = stored(qs, qf, qa, qb, lo, hi)
            t0, t1 = stored(ts, tf, ta, tb, lo, hi)
            strand = qf != tf
            qpiece = wr.revcomp(qs)[len(qs) - q1:len(qs) - q0] if strand else qs[q0:q1]
            cg = nw_cigar(qpiece, ts[t0:t1])
            lines.append(f"{qn}\t{len(qs)}\t{q0}\t{q1}\t{'-' if strand else '+'}\t{tn}\t{len(ts)}\t{t0}\t{t1}\t{min(q1 - q0, t1 - t0)}\t{max(q1 - q0, t1 - t0)}\t60\tcg:Z:{cg}")
    return "\n".join(lines) + "\n"


def reference_round(recs, paf_text, tmp, mode, include_unpolished):
    """One polisher invocation as the reference would compute it: our window builder, the reference's window.cpp per window."""
    p = os.path.join(tmp, "o.paf")
    open(p, "w").write(paf_text)
    ovl = seqio.read_overlaps(p)
    wb = WindowBuilder(500, 10.0)
    kept, wtype = seqio.load_polisher_input(wb, recs, recs, ovl, 0.3)
    batch, ids = wb.build()
    prm = capi.default_params(mode=mode, min_confidence=0.2 if mode == 0 else 0.22, min_support=0.2 if mode == 0 else 0.19, num_prune=3, trim=1,
                              window_type=wtype)
    cons, status = [], []
    for w in range(batch.n_windows):
        c, ok = oa.ref_window(batch, w, prm)
        cons.append(c); status.append(capi.VC_WIN_OK if ok else capi.VC_WIN_UNPOLISHED)
    text = b"".join(b">" + n.encode() + b"\n" + d + b"\n" for n, d in wb.stitch(cons, status, drop_unpolished=not include_unpolished, fragment_correction=True))
    wb.close()
    return text.decode(), batch.n_windows, kept


def main():
    import tempfile
    rng = random.Random(20260930)
    genome = read_contig("/root/reference/example/ref.fa")
    L = 1700
    hapA = genome[90000:90000 + L]
    hapB = bytes(rng.choice([x for x in b"ACGT" if x != c]) if rng.random() < 0.012 else c for c in hapA)
    reads, spans, fastq = [], {}, ""
    for r in range(12):
        hap = hapA if r % 2 == 0 else hapB
        a, b = (0, L) if r % 3 else (rng.randrange(0, 300), rng.randrange(L - 300, L))
        fwd = mutate(rng, hap[a:b], 0.10)
        flag = rng.random() < 0.5
        data = wr.revcomp(fwd) if flag else fwd
        qual = bytes(rng.randint(33 + 6, 33 + 26) for _ in data)
        name = f"read{r}"
        reads.append((name, data, flag)); spans[name] = (a, b)
        fastq += f"@{name}\n{data.decode()}\n+\n{qual.decode()}\n"
    with tempfile.TemporaryDirectory() as tmp:
        fq = os.path.join(tmp, "reads.fastq")
        open(fq, "w").write(fastq)
        recs1 = seqio.read_sequences(fq)
        paf1 = overlaps_paf(reads, spans)
        text1, nw1, kept1 = reference_round(recs1, paf1, tmp, 0, False)
        fa1 = os.path.join(tmp, "r1.fa")
        open(fa1, "w").write(text1)
        recs2 = seqio.read_sequences(fa1)
        flags = {n: f for n, _, f in reads}
        reads2 = [(n, d, flags[n[:-1]]) for n, d, _ in recs2]                # names carry one 'r' per round
        spans2 = {n: spans[n[:-1]] for n, _, _ in reads2}
        paf2 = overlaps_paf(reads2, spans2)
        text2, nw2, kept2 = reference_round(recs2, paf2, tmp, 1, False)
    fx = dict(reads_fastq=fastq, round1_paf=paf1, round1_fasta=text1, round2_paf=paf2, round2_fasta=text2,
              note="round 1: -f -p -d 0.2 -s 0.2; round 2: -f; window consensus by oracle/_ref (the reference's window.cpp)")
    out = os.path.join(HERE, "two_rounds.json.gz")
    with gzip.open(out, "wt", compresslevel=9) as f:
        json.dump(fx, f)
    print(out, os.path.getsize(out), "bytes; round 1:", nw1, "windows,", kept1, "overlaps,", text1.count(">"), "reads out; round 2:", nw2, "windows,",
          kept2, "overlaps,", text2.count(">"), "reads out")


if __name__ == "__main__":
    main()
