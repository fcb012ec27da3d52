"""Generates tests/golden/plumbing.json.gz: the stand-in for BASELINE config A (example/reads.fq.gz is not
shipped with the reference; SURVEY 8(d) asks for reads simulated from example/ref.fa with our own overlap /
window builder).  Runs ONLY in the build container: it reads /root/reference/example/ref.fa (DATA) and takes
the expected per-window results from the REAL reference compiled in place (oracle/_ref).

  python tests/golden/make_plumbing.py

Two targets (noisy reads of a genome segment, one FASTQ, one FASTA) and reads from two haplotypes on both
strands; read-vs-target alignments are global edit-distance alignments computed here (numpy DP) and handed to
the window builder as CIGAR strings, the way a SAM file would.
"""
import gzip
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from vechat_amd import capi  # noqa: E402
from vechat_amd.windows import WindowBuilder  # noqa: E402
import oracle_api as oa  # noqa: E402
import windows_ref as wr  # noqa: E402

W = 500


def read_contig(path):
    seq = []
    for line in open(path):
        if line.startswith(">"):
            if seq:
                break
            continue
        seq.append(line.strip())
    return "".join(seq).upper().encode()


def mutate(rng, s, rate, ins=0.4, dele=0.3):
    out = bytearray()
    for c in s:
        r = rng.random()
        if r < rate * dele:
            continue
        if r < rate * (dele + ins):
            out.append(rng.choice(b"ACGT"))
            out.append(c)
        elif r < rate:
            out.append(rng.choice([x for x in b"ACGT" if x != c]))
        else:
            out.append(c)
    return bytes(out)


def nw_cigar(q, t):
    """global unit-cost alignment of q (read) against t (target) -> CIGAR with M/I/D"""
    n, m = len(q), len(t)
    qa, ta = np.frombuffer(q, np.uint8), np.frombuffer(t, np.uint8)
    D = np.zeros((n + 1, m + 1), np.int32)
    D[0] = np.arange(m + 1)
    idx = np.arange(m + 1)
    for i in range(1, n + 1):
        sub = D[i - 1, :-1] + (ta != qa[i - 1])
        up = D[i - 1, 1:] + 1
        row = np.empty(m + 1, np.int32)
        row[0] = i
        row[1:] = np.minimum(sub, up)
        row = np.minimum.accumulate(row - idx) + idx          # horizontal gaps
        D[i] = row
    i, j, ops = n, m, []
    while i > 0 or j > 0:
        if i > 0 and j > 0 and D[i, j] == D[i - 1, j - 1] + (q[i - 1] != t[j - 1]):
            ops.append("M"); i -= 1; j -= 1
        elif i > 0 and D[i, j] == D[i - 1, j] + 1:
            ops.append("I"); i -= 1
        else:
            ops.append("D"); j -= 1
    ops.reverse()
    cig, run = "", 1
    for a, b in zip(ops, ops[1:] + ["$"]):
        if a == b:
            run += 1
        else:
            cig += f"{run}{a}"; run = 1
    return cig


def main():
    rng = random.Random(20240929)
    genome = read_contig("/root/reference/example/ref.fa")
    seqs, overlaps = [], []
    truths = []
    for t, (start, length, fastq) in enumerate([(40000, 1700, True), (150000, 1300, False)]):
        hapA = genome[start:start + length]
        hapB = bytes(rng.choice([x for x in b"ACGT" if x != c]) if rng.random() < 0.012 else c for c in hapA)
        target = mutate(rng, hapA, 0.12)
        qual = bytes(rng.randint(33 + 5, 33 + 24) for _ in target) if fastq else None
        seqs.append((f"target{t}", target, qual))
        truths.append((hapA, hapB, target))
    nt = len(seqs)
    for t, (hapA, hapB, target) in enumerate(truths):
        for r in range(28):
            hap = hapA if rng.random() < 0.55 else hapB
            if rng.random() < 0.5:
                a, b = 0, len(hap)
            else:
                a = rng.randrange(0, len(hap) - 450)
                b = rng.randint(a + 400, len(hap))
            piece = mutate(rng, hap[a:b], 0.15)
            # target interval that corresponds to truth[a:b): align the truth piece ends by proportion, then let the
            # global alignment absorb the slack as leading / trailing indels
            ta = int(round(a * len(target) / len(hap)))
            tb = int(round(b * len(target) / len(hap)))
            cigar = nw_cigar(piece, target[ta:tb])
            pre, post = rng.randint(0, 30), rng.randint(0, 30)
            fwd = bytes(rng.choice(b"ACGT") for _ in range(pre)) + piece + bytes(rng.choice(b"ACGT") for _ in range(post))
            strand = rng.random() < 0.5
            data = wr.revcomp(fwd) if strand else fwd
            qb, qe = (post, post + len(piece)) if strand else (pre, pre + len(piece))
            qual = bytes(rng.randint(33 + 5, 33 + 24) for _ in data) if rng.random() < 0.8 else None
            seqs.append((f"read{t}_{r}", data, qual))
            overlaps.append((len(seqs) - 1, t, int(strand), qb, qe, len(data), ta, tb, cigar))

    # assembly edge cases (src/polisher.cpp:408-459, src/overlap.cpp:222-292), crafted rather than left to chance: overlaps that
    # begin / end exactly on a window boundary, a reverse-strand read that ends with the target, a span that leaves less than
    # 2 % of a window in its last window (that layer is dropped, :416), a low-quality read (mean quality < 10: dropped, :420-434),
    # a read that covers one window only
    for t, (hapA, hapB, target) in enumerate(truths):
        def crafted(tag, ta, tb, strand, qual_lo=5, qual_hi=24, has_qual=True):
            a = int(round(ta * len(hapA) / len(target))); b = int(round(tb * len(hapA) / len(target)))
            piece = mutate(rng, hapA[a:b], 0.15)
            cigar = nw_cigar(piece, target[ta:tb])
            data = wr.revcomp(piece) if strand else piece
            qual = bytes(rng.randint(33 + qual_lo, 33 + qual_hi) for _ in data) if has_qual else None
            seqs.append((f"edge{t}_{tag}", data, qual))
            overlaps.append((len(seqs) - 1, t, int(strand), 0, len(piece), len(data), ta, tb, cigar))
        crafted("on_boundaries", W, 2 * W, False)                        # exactly one whole window
        crafted("ends_on_boundary", 130, W, True)                        # reverse strand, last aligned base is the window's last
        crafted("starts_on_boundary", W, W + 333, False)
        crafted("rc_to_target_end", len(target) - 620, len(target), True)
        crafted("sliver", 40, W + 6, False)                              # 6 bases into the second window: < 2 % of W, dropped there
        crafted("low_quality", 100, 2 * W + 100, False, 1, 7)            # mean quality ~4 < 10
        crafted("one_window_fasta", W + 20, 2 * W - 20, True, has_qual=False)

    wb = WindowBuilder(W, 10.0)
    for name, d, q in seqs:
        wb.add_sequence(name, d, q)
    wb.set_targets(nt)
    for o in overlaps:
        wb.add_overlap(*o)
    batch, ids = wb.build()
    expected = {}
    for key, mode in (("hap", 0), ("linear", 1)):
        p = capi.default_params(mode=mode)
        cons, pol = [], []
        for w in range(batch.n_windows):
            c, ok = oa.ref_window(batch, w, p)
            o, okp, _ = oa.oracle_run(batch, p, w, w + 1)
            assert o[0] == c and bool(okp[0]) == bool(ok), (key, w)            # our oracle agrees with the reference here too
            cons.append(c.decode()); pol.append(bool(ok))
        st = wb.stitch([c.encode() for c in cons], [capi.VC_WIN_OK if x else capi.VC_WIN_UNPOLISHED for x in pol])
        expected[key] = dict(consensus=cons, polished=pol, stitched=[[n, d.decode()] for n, d in st])
    fx = dict(window_length=W, quality_threshold=10.0, n_targets=nt,
              sequences=[[n, d.decode(), None if q is None else q.decode()] for n, d, q in seqs],
              overlaps=[list(o) for o in overlaps],
              windows=[list(x) for x in ids], layers_per_window=[int(x) - 1 for x in np.diff(batch.win_seq_off)],
              expected=expected)
    out = os.path.join(HERE, "plumbing.json.gz")
    with gzip.open(out, "wt", compresslevel=9) as f:
        json.dump(fx, f)
    print(out, os.path.getsize(out), "bytes;", batch.n_windows, "windows, layers", fx["layers_per_window"],
          "polished", expected["hap"]["polished"], [len(c) for c in expected["hap"]["consensus"]])


if __name__ == "__main__":
    main()
