"""GPU suite for the overlap aligner.  Parity with edlib cannot be pinned (not vendored, and an
optimal path is not unique); what is checked against a CPU DP is what any consumer relies on: the CIGAR is a valid
global alignment of exactly these two sequences and its cost is the unit-cost edit distance."""
import random
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def edit_distance(q, t):
    qa, ta = np.frombuffer(q, np.uint8), np.frombuffer(t, np.uint8)
    m = len(t)
    idx = np.arange(m + 1)
    row = idx.astype(np.int64)
    for i in range(1, len(q) + 1):
        new = np.empty(m + 1, np.int64)
        new[0] = i
        if m:
            new[1:] = np.minimum(row[:-1] + (ta != qa[i - 1]), row[1:] + 1)
        row = np.minimum.accumulate(new - idx) + idx
    return int(row[m])


def cigar_cost(cigar, q, t):
    """walks the CIGAR over both sequences; returns its cost, or None if it does not consume them exactly"""
    i = j = cost = 0
    for num, op in re.findall(r"(\d+)([MID])", cigar):
        n = int(num)
        if op == "M":
            if i + n > len(q) or j + n > len(t):
                return None
            cost += sum(1 for k in range(n) if q[i + k] != t[j + k]); i += n; j += n
        elif op == "I":
            cost += n; i += n
        else:
            cost += n; j += n
    return cost if (i, j) == (len(q), len(t)) and re.fullmatch(r"(\d+[MID])*", cigar) else None


def _mut(rng, s, rate):
    out = bytearray()
    for c in s:
        r = rng.random()
        if r < rate * 0.3:
            continue
        if r < rate * 0.7:
            out.append(rng.choice(b"ACGT"))
        out.append(rng.choice(b"ACGT") if rate * 0.7 <= r < rate else c)
    return bytes(out)


def test_cigars_are_optimal_global_alignments(built):
    from vechat_amd.align import align_pairs
    rng = random.Random(11)
    pairs = [(b"A", b"A"), (b"A", b"C"), (b"ACGT", b"A"), (b"A", b"ACGTACGT"), (b"ACGTTGCA", b"ACGTTGCA"),
             (b"", b"ACG"), (b"ACG", b""), (b"", b"")]
    for L in (7, 33, 64, 65, 500, 2047, 2048, 2049, 3000, 4500):
        t = bytes(rng.choice(b"ACGT") for _ in range(L))
        pairs.append((_mut(rng, t, 0.25) or b"A", t))
        pairs.append((t, _mut(rng, t, 0.1) or b"A"))
    pairs.append((bytes(rng.choice(b"ACGT") for _ in range(300)), bytes(rng.choice(b"ACGT") for _ in range(2500))))   # unrelated
    pairs.append((b"ACGTNNACGT" * 30, b"ACGTACGT" * 40))
    rng2 = random.Random(5)
    long_t = bytes(rng2.choice(b"ACGT") for _ in range(21000))                # beyond int16 as an absolute score: relative scores
    long_q = _mut(rng2, long_t, 0.2)
    cg_big, d_big = align_pairs([pairs[0], (long_q, long_t), pairs[1]])
    assert d_big[0] == 0 and d_big[2] == 1 and cg_big[0] == "1M"
    assert d_big[1] == edit_distance(long_q, long_t) and cigar_cost(cg_big[1], long_q, long_t) == d_big[1]
    cigars, dist = align_pairs(pairs)
    for (q, t), cg, d in zip(pairs, cigars, dist):
        assert d == edit_distance(q, t), (len(q), len(t))
        assert cigar_cost(cg, q, t) == d, (len(q), len(t), cg[:60])


@pytest.mark.parametrize("dist_path", [False, True])
def test_paf_without_cigar_end_to_end(built, tmp_path, capsys, monkeypatch, dist_path):
    """The VeChat driver's own input shape: PAF from minimap2 without cg tags.  The command line aligns the
    overlaps on the device first; the corrected reads must come out polished and close to the truth."""
    import fixtures
    from test_seqio import write_inputs
    from vechat_amd import polish
    if dist_path:                                   # one-process-per-GPU path (sharded alignment, RCCL exchange), single rank
        monkeypatch.setenv("VC_FORCE_DIST", "1")
        monkeypatch.setenv("MASTER_PORT", "29549")
    fx, wb = fixtures.load_plumbing()
    wb.close()
    rp, op, tp = write_inputs(fx, tmp_path, sam=False)
    txt = "\n".join(ln.split("\tcg:Z:")[0] for ln in open(op).read().strip().split("\n")) + "\n"
    open(op, "w").write(txt)
    assert polish.main([str(rp), str(op), str(tp), "-p", "-d", "0.2", "-s", "0.2"]) == 0
    out = capsys.readouterr().out.strip().split("\n")
    got = {out[i][1:].split()[0]: out[i + 1] for i in range(0, len(out), 2)}
    exp = {n.split()[0]: d for n, d in fx["expected"]["hap"]["stitched"]}
    assert set(got) == set(exp)
    for name in exp:                       # another optimal alignment moves a few window boundaries: near-identical, not identical
        assert abs(len(got[name]) - len(exp[name])) < 0.05 * len(exp[name])
        assert edit_distance(got[name].encode(), exp[name].encode()) < 0.05 * len(exp[name])
