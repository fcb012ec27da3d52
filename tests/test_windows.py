"""CPU suite for the window assembly / stitching component: the C++ implementation behind
the C ABI against an independent Python restatement, on seeded random overlaps and on the edge cases the
reference's rules create (2 % length filter, mean-quality filter, single-column layers, reverse strand,
the dummy-quality window flag)."""
import ctypes as C
import random

import numpy as np
import pytest

import windows_ref as wr
from vechat_amd import capi
from vechat_amd.windows import WindowBuilder


def _rank(host):
    def f(begins):
        n = len(begins)
        out = (C.c_uint32 * n)()
        host.vc_rank_layers((C.c_uint32 * n)(*begins), n, out)
        return list(out)
    return f


def _mutate(rng, s, rate, alpha=b"ACGT"):
    """-> (read bytes, cigar of read vs s)"""
    out, ops = bytearray(), []
    for c in s:
        r = rng.random()
        if r < rate * 0.3:
            ops.append("D")
        elif r < rate * 0.7:
            out.append(rng.choice(alpha)); out.append(c); ops.append("I"); ops.append("M")
        else:
            out.append(rng.choice(alpha) if r < rate else c); ops.append("M")
    cig, run = "", 1
    for a, b in zip(ops, ops[1:] + ["$"]):
        if a == b:
            run += 1
        else:
            cig += f"{run}{a}"; run = 1
    return bytes(out), cig


def _case(seed, W, fastq_targets, fastq_reads):
    rng = random.Random(seed)
    flank = b"ACGTNRYKMSW" if seed % 3 == 1 else b"ACGT"      # symbols the reverse complement leaves as they are
    seqs, ovl = [], []
    nt = rng.randint(1, 3)
    for t in range(nt):
        L = rng.randint(W // 2, 4 * W + 37)
        d = bytes(rng.choice(b"ACGT") for _ in range(L))
        q = bytes(rng.randint(33, 70) for _ in range(L)) if fastq_targets else None
        if q is not None and t == 0:
            q = q[:-(L % W or W)] + b"!" * (L % W or W)        # last window of target 0: all '!' (flag quirk)
        seqs.append((f"t{t}", d, q))
    for r in range(rng.randint(3, 12)):
        t = rng.randrange(nt)
        td = seqs[t][1]
        tb = rng.randrange(0, max(1, len(td) - 5)); te = rng.randint(tb + 1, len(td))
        piece, cig = _mutate(rng, td[tb:te], 0.15, flank)
        while cig and cig[-1] == "D" or (cig and cig.split("M")[0].endswith("D")):   # keep the record simple: start/end on M or I
            break
        if not piece:
            continue
        pre, post = rng.randint(0, 20), rng.randint(0, 20)
        fwd = bytes(rng.choice(flank) for _ in range(pre)) + piece + bytes(rng.choice(flank) for _ in range(post))
        strand = rng.random() < 0.5
        data = wr.revcomp(fwd) if strand else fwd
        ql = len(data)
        qb, qe = (post, post + len(piece)) if strand else (pre, pre + len(piece))
        q = bytes(rng.randint(33 + 2, 33 + 30) for _ in range(ql)) if fastq_reads and rng.random() < 0.8 else None
        seqs.append((f"r{r}", data, q))
        ovl.append((len(seqs) - 1, t, int(strand), qb, qe, ql, tb, te, cig))
    return seqs, nt, ovl


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("W", [20, 50])
def test_builder_matches_the_python_restatement(built, seed, W):
    host = capi.load_host()
    seqs, nt, ovl = _case(seed * 7 + W, W, fastq_targets=seed % 2 == 0, fastq_reads=seed % 3 != 0)
    # the builder wants targets first
    wb = WindowBuilder(W, 10.0)
    for name, d, q in seqs[:nt]:
        wb.add_sequence(name, d, q)
    for name, d, q in seqs[nt:]:
        wb.add_sequence(name, d, q)
    wb.set_targets(nt)
    for k, o in enumerate(ovl):
        assert wb.add_overlap(*o) == k
        assert wb.breaking_points(k) == wr.breaking_points(o[8], o[2], o[3], o[4], o[5], o[6], o[7], W)
    batch, ids = wb.build()
    wins, cov = wr.build_windows(seqs, nt, ovl, W, 10.0, _rank(host))
    assert batch.n_windows == len(wins) and ids == [(w["target"], w["rank"]) for w in wins]
    for w, ref in enumerate(wins):
        s, q, b, e = batch.window(w)
        assert int(batch.win_fasta[w]) == int(ref["fasta"])
        exp = [(ref["backbone"], ref["backbone_quality"], 0, 0)] + ref["layers"]
        exp = [exp[i] for i in ref["order"]]
        assert len(s) == len(exp)
        for k in range(len(s)):
            assert s[k] == exp[k][0] and q[k] == exp[k][1] and (b[k], e[k]) == (exp[k][2], exp[k][3])
        assert [int(x) for x in batch.seq_orig[batch.win_seq_off[w]:batch.win_seq_off[w + 1]]] == ref["order"]
    # the same build in two steps (vc_wb_build_begin + vc_wb_build_fill in ragged pieces): every slice complete once its windows are filled
    sb, sids, fill = wb.build_streaming()
    assert sids == ids and sb.n_windows == batch.n_windows
    cuts = sorted({0, batch.n_windows} | {random.Random(seed + 99).randrange(0, batch.n_windows + 1) for _ in range(3)})
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        fill(lo, hi)
        got, want = sb.slice(lo, hi), batch.slice(lo, hi)
        for k in ("win_seq_off", "seq_off", "seq_begin", "seq_end", "seq_has_qual", "bases", "quals", "win_fasta"):
            assert np.array_equal(getattr(got, k), getattr(want, k)), (k, lo, hi)
    for k in ("win_seq_off", "seq_off", "seq_begin", "seq_end", "seq_has_qual", "bases", "quals", "win_fasta", "seq_orig"):
        assert np.array_equal(getattr(sb, k), getattr(batch, k)), k
    # stitching with made-up window results
    rng = random.Random(seed)
    cons = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 9))) for _ in wins]
    status = [rng.choice([capi.VC_WIN_OK, capi.VC_WIN_OK, capi.VC_WIN_UNPOLISHED]) for _ in wins]
    names = [s[0] for s in seqs]
    for drop in (True, False):
        got = wb.stitch(cons, status, drop_unpolished=drop, fragment_correction=True)
        assert got == wr.stitch(wins, cov, names, cons, [st == capi.VC_WIN_OK for st in status], drop, True)
    wb.close()


def test_layer_filters_and_flag_quirk(built):
    W = 100
    wb = WindowBuilder(W, 10.0)
    t = bytes(b"ACGT"[i % 4] for i in range(250))
    wb.add_sequence("t", t, None)                      # FASTA target: windows 100, 100, 50
    wb.add_sequence("short", t[10:11], None)           # 1 base < 2 % of W -> dropped
    wb.add_sequence("lowq", t[0:60], b"#" * 60)        # mean quality 2 -> dropped
    wb.add_sequence("ok", t[100:250], b"5" * 150)      # spans windows 1 and 2
    wb.set_targets(1)
    wb.add_overlap(1, 0, 0, 0, 1, 1, 10, 11, "1M")
    wb.add_overlap(2, 0, 0, 0, 60, 60, 0, 60, "60M")
    wb.add_overlap(3, 0, 0, 0, 150, 150, 100, 250, "150M")
    batch, ids = wb.build()
    assert ids == [(0, 0), (0, 1), (0, 2)]
    assert [int(x) for x in np.diff(batch.win_seq_off)] == [1, 2, 2]
    assert [int(x) for x in batch.win_fasta] == [1, 1, 0]          # the short last window loses the flag (window.cpp:223)
    s, q, b, e = batch.window(2)
    assert s[1] == t[200:250] and (b[1], e[1]) == (0, 49)
    with pytest.raises(ValueError):
        wb.add_overlap(3, 0, 0, 0, 150, 149, 100, 250, "150M")    # length mismatch is fatal in the reference too
    with pytest.raises(ValueError):
        wb.stitch([b"A", b"C", b"G"], [0, 2, 0])                  # a window without a result cannot be stitched
    out = wb.stitch([b"AA", b"", b"G"], [1, 1, 1], drop_unpolished=True)
    assert out == []                                              # polished ratio 0 -> dropped
    out = wb.stitch([b"AA", b"", b"G"], [1, 0, 1], drop_unpolished=True, fragment_correction=False)
    assert out == [("t LN:i:3 RC:i:3 XC:f:0.333333", b"AAG")]
    wb.close()


@pytest.mark.parametrize("mode,key", [(0, "hap"), (1, "linear")])
def test_plumbing_fixture_through_the_oracle(built, mode, key):
    """Config A stand-in: reads simulated from example/ref.fa, assembled into windows by the builder; the oracle
    must reproduce what the real reference produced for every window, and stitching must give the stored reads."""
    import fixtures
    import oracle_api as oa
    fx, wb = fixtures.load_plumbing()
    batch, ids = wb.build()
    assert [list(x) for x in ids] == fx["windows"]
    assert [int(x) - 1 for x in np.diff(batch.win_seq_off)] == fx["layers_per_window"]
    exp = fx["expected"][key]
    cons, pol, _ = oa.oracle_run(batch, capi.default_params(mode=mode))
    assert [c.decode() for c in cons] == exp["consensus"] and [bool(x) for x in pol] == exp["polished"]
    st = wb.stitch(cons, [capi.VC_WIN_OK if x else capi.VC_WIN_UNPOLISHED for x in pol])
    assert [[n, d.decode()] for n, d in st] == exp["stitched"]
    wb.close()


def test_cigar_that_overruns_the_read_is_rejected(built):
    """A CIGAR consuming more query / target than the overlap's spans must be refused when the overlap is added (it would
    put breaking points past the end of the read)."""
    import ctypes as C
    from vechat_amd import capi
    lib = capi.load_host()
    wb = lib.vc_wb_create(500, 10.0)
    t = b"ACGT" * 300
    r = b"ACGT" * 50
    assert lib.vc_wb_add_sequence(wb, b"t", t, len(t), None) == 0
    assert lib.vc_wb_add_sequence(wb, b"r", r, len(r), None) == 1
    assert lib.vc_wb_set_targets(wb, 1) == 0
    assert lib.vc_wb_add_overlap(wb, 1, 0, 0, 0, 200, 200, 0, 200, b"200M") == 0
    assert lib.vc_wb_add_overlap(wb, 1, 0, 0, 0, 200, 200, 0, 1000, b"1000M") != 0
    assert b"CIGAR" in lib.vc_wb_last_error(wb)
    assert lib.vc_wb_add_overlap(wb, 1, 0, 0, 0, 200, 200, 0, 200, b"150M") != 0
    lib.vc_wb_destroy(wb)
