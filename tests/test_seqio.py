"""CPU suite for the file-format layer: the plumbing fixture written out as FASTQ/FASTA + SAM
and + PAF(cg), read back, must give the very batch the fixture's in-memory overlaps give."""
import gzip

import numpy as np
import pytest

import fixtures
from vechat_amd import seqio
from vechat_amd.windows import WindowBuilder


def write_inputs(fx, d, sam=True):
    nt = fx["n_targets"]
    seqs = fx["sequences"]

    def rec(f, name, data, qual, extra=""):
        f.write(f"@{name}{extra}\n{data}\n+\n{qual}\n" if qual is not None else f">{name}{extra}\n{data}\n")
    tp, rp = d / "targets.fastq", d / "reads.fastq.gz"
    with open(tp, "w") as f:
        for n, s, q in seqs[:nt]:
            rec(f, n, s.lower() if n.endswith("1") else s, q if q is not None else "!" * len(s), " some description")
    with gzip.open(rp, "wt") as f:
        for n, s, q in seqs[nt:]:
            rec(f, n, s, q if q is not None else "!" * len(s))
    op = d / ("ovl.sam" if sam else "ovl.paf")
    with open(op, "w") as f:
        if sam:
            f.write("@HD\tVN:1.6\n")
        for q_id, t_id, strand, qb, qe, ql, tb, te, cigar in fx["overlaps"]:
            qn, tn = seqs[q_id][0], seqs[t_id][0]
            if sam:
                lead, trail = (ql - qe, qb) if strand else (qb, ql - qe)      # clips in the orientation of the alignment
                cg = (f"{lead}S" if lead else "") + cigar + (f"{trail}S" if trail else "")
                f.write(f"{qn}\t{16 if strand else 0}\t{tn}\t{tb + 1}\t60\t{cg}\t*\t0\t0\t*\t*\n")
            else:
                f.write(f"{qn}\t{ql}\t{qb}\t{qe}\t{'-' if strand else '+'}\t{tn}\t{len(seqs[t_id][1])}\t{tb}\t{te}\t0\t0\t60\ttp:A:P\tcg:Z:{cigar}\n")
        if sam:
            f.write(f"{seqs[nt][0]}\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n")          # an unmapped record is skipped
    return rp, op, tp


@pytest.mark.parametrize("sam", [True, False])
def test_files_round_trip_into_the_same_batch(built, tmp_path, sam):
    fx, wb0 = fixtures.load_plumbing()
    b0, ids0 = wb0.build()
    rp, op, tp = write_inputs(fx, tmp_path, sam)
    targets, reads, ovl = seqio.read_sequences(tp), seqio.read_sequences(rp), seqio.read_overlaps(op)
    assert targets[0][0] == fx["sequences"][0][0] and targets[1][2] is None          # description cut, all-'!' quality dropped
    assert targets[1][1] == fx["sequences"][1][1].encode()                           # lower case input is upper-cased
    wb = WindowBuilder(fx["window_length"], fx["quality_threshold"])
    kept, wtype = seqio.load_polisher_input(wb, targets, reads, ovl)
    assert kept == len(fx["overlaps"]) and wtype == 1
    b1, ids1 = wb.build()
    assert ids1 == ids0
    for k in ("win_seq_off", "seq_off", "seq_begin", "seq_end", "seq_has_qual", "bases", "quals", "win_fasta"):
        assert np.array_equal(getattr(b0, k), getattr(b1, k)), k
    wb.close(); wb0.close()


def test_filters(built, tmp_path):
    t = [("t", b"ACGT" * 30, None)]
    r = [("t", b"ACGT" * 30, None), ("r", b"ACGT" * 10, None)]
    o = [seqio.Overlap(q_name="t", t_name="t", strand=False, q_begin=0, q_end=120, q_length=120, t_begin=0, t_end=120, cigar="120M", error=0.0, length=120),
         seqio.Overlap(q_name="r", t_name="t", strand=False, q_begin=0, q_end=40, q_length=40, t_begin=0, t_end=100, cigar="40M", error=0.6, length=100),
         seqio.Overlap(q_name="r", t_name="t", strand=False, q_begin=0, q_end=40, q_length=40, t_begin=4, t_end=44, cigar="40M", error=0.0, length=40),
         seqio.Overlap(q_name="zz", t_name="t", strand=False, q_begin=0, q_end=4, q_length=4, t_begin=0, t_end=4, cigar="4M", error=0.0, length=4)]
    wb = WindowBuilder(50, 10.0)
    kept, wtype = seqio.load_polisher_input(wb, t, r, o)
    assert kept == 1 and wtype == 0                       # self overlap, high error and unknown read are dropped
    wb.close()
    with pytest.raises(ValueError):
        seqio.read_overlaps(tmp_path / "x.txt")
    (tmp_path / "x.mhap").write_text("2 1 0.1 10 0 0 40 40 1 4 44 120\n")
    mh = seqio.read_overlaps(tmp_path / "x.mhap")
    assert (mh[0].q_name, mh[0].t_name, mh[0].strand, mh[0].cigar) == ("#1", "#0", True, None)
    seqio._resolve_indices(t, r, mh)
    assert (mh[0].q_name, mh[0].t_name) == ("r", "t")
    (tmp_path / "nocg.paf").write_text("r\t40\t0\t40\t+\tt\t120\t4\t44\t40\t40\t60\n")
    o = seqio.read_overlaps(tmp_path / "nocg.paf")
    assert o[0].cigar is None                                  # to be aligned on the device (seqio.align_missing)
    wb = WindowBuilder(50, 10.0)
    with pytest.raises(ValueError):
        seqio.load_polisher_input(wb, t, r, o)
    wb.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_sequence_ingest_matches_the_reference_parser(tmp_path, seed):
    """read_sequences against the reference's own ingest (its vendored bioparser feeding racon::Sequence, compiled in
    place by oracle/Makefile `seqparse`): names, upper-casing, dropped all-'!' quality, multi-line FASTA, .gz, IUPAC and
    other symbols; and the reverse complement / reverse quality rule (src/sequence.cpp:50-83) the window builder restates."""
    import random

    import oracle_api as oa
    import windows_ref

    if not oa.have_seqparse():
        pytest.skip("oracle/_ref/libvcseq.so not built (needs /root/reference)")
    rng = random.Random(seed)

    def rs(n, alpha="ACGTacgtNnRYKMSWBDHVUu-*."):
        return "".join(rng.choice(alpha) for _ in range(n))

    fa = ""
    for i in range(12):
        body = rs(rng.choice([1, 5, 70, 71, 500, 3000]))
        width = rng.choice([len(body), 60, 70])
        fa += f">t{i}" + rng.choice(["", " some description", "\ttabbed text"]) + "\n"
        fa += "".join(body[k:k + width] + "\n" for k in range(0, len(body), width))
    fq = ""
    for i in range(12):
        n = rng.choice([1, 2, 150, 2000])
        q = "!" * n if i % 4 == 2 else "".join(chr(33 + rng.randint(0, 60)) for _ in range(n))
        fq += f"@q{i}" + rng.choice(["", " len=%d" % n]) + f"\n{rs(n)}\n+\n{q}\n"
    files = []
    for name, text, is_fq in (("a.fasta", fa, 0), ("a.fastq", fq, 1)):
        p = tmp_path / name
        p.write_text(text)
        pz = tmp_path / (name + ".gz")
        with gzip.open(pz, "wt") as f:
            f.write(text)
        files += [(p, is_fq), (pz, is_fq)]
    for p, is_fq in files:
        ref = oa.ref_parse_sequences(p, is_fq)
        mine = seqio.read_sequences(p)
        assert len(ref) == len(mine) == 12
        for (rn, rd, rq, rrc, rrq), (n, d, q) in zip(ref, mine):
            assert (rn, rd, rq) == (n, d, q)
            assert rrc == windows_ref.revcomp(d)
            assert rrq == (q[::-1] if q is not None else None)


def test_cigar_scan_fast_and_plain_paths_agree():
    """The vectorised CIGAR scan of the SAM constructor (lengths and letters as two arrays) against the plain scan over
    (count, letter) pairs it replaced (src/overlap.cpp:44-110), on random and on odd strings."""
    import random
    from vechat_amd import seqio
    rnd = random.Random(5)

    def plain(cigar, flag, pos):
        ops = seqio._CIG.findall(cigar)
        q_begin = int(ops[0][0]) if ops and ops[0][1] in b"SH" else 0
        q_aln = sum(int(k) for k, o in ops if o in b"M=XI")
        t_aln = sum(int(k) for k, o in ops if o in b"M=XDN")
        clip = sum(int(k) for k, o in ops if o in b"SH")
        strand = bool(flag & 0x10)
        q_end, q_length = q_begin + q_aln, clip + q_aln
        if strand:
            q_begin, q_end = q_length - q_end, q_length - q_begin
        return q_begin, q_end, q_length, pos - 1, pos - 1 + t_aln, max(q_aln, t_aln)

    cases = [b"10M", b"5S10M2I3D4H", b"3H7M", b"12=3X1I", b"10", b"M10", b"3MM", b"12Q", b"1M2", b"7N3M", b"2P5M"]
    for _ in range(300):
        n = rnd.randint(1, 400)
        body = b"".join(b"%d%s" % (rnd.randint(1, 5000), rnd.choice([b"M", b"I", b"D", b"=", b"X", b"N"])) for _ in range(n))
        cases.append(rnd.choice([b"", b"12S", b"3H"]) + body + rnd.choice([b"", b"9S", b"40H"]))
    for cg in cases:
        for flag in (0, 16):
            o = seqio._sam_overlap("q", flag, "t", 101, cg)
            assert (o.q_begin, o.q_end, o.q_length, o.t_begin, o.t_end, o.length) == plain(cg, flag, 101), cg


# ---- the C++ readers behind the C ABI (vechat_amd/csrc/vc_io.cpp) against the Python restatement above, format by format
def _same_overlaps(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        for k in seqio.Overlap.__slots__:
            assert getattr(x, k) == getattr(y, k), (k, getattr(x, k), getattr(y, k))


@pytest.mark.parametrize("sam", [True, False])
def test_native_readers_give_the_same_batch(built, tmp_path, sam):
    fx, wb0 = fixtures.load_plumbing()
    b0, ids0 = wb0.build()
    rp, op, tp = write_inputs(fx, tmp_path, sam)
    targets, reads, ovl = seqio.NativeSequences(tp), seqio.NativeSequences(rp), seqio.NativeOverlaps(op)
    assert targets.records() == seqio.read_sequences(tp) and reads.records() == seqio.read_sequences(rp)
    assert targets.index() == seqio.sequence_index(tp)
    keep = {n for n, _, _ in seqio.read_sequences(rp)[::3]}
    assert seqio.NativeSequences(rp, keep=keep).records() == seqio.read_sequences(rp, keep)
    assert seqio.NativeSequences(rp, names_only=True).index() == seqio.sequence_index(rp)
    _same_overlaps(ovl.records(), seqio.read_overlaps(op))
    wb = WindowBuilder(fx["window_length"], fx["quality_threshold"])
    kept, wtype = seqio.load_polisher_input_native(wb, targets, reads, ovl)
    assert kept == len(fx["overlaps"]) and wtype == 1
    b1, ids1 = wb.build()
    assert ids1 == ids0
    for k in ("win_seq_off", "seq_off", "seq_begin", "seq_end", "seq_has_qual", "bases", "quals", "win_fasta"):
        assert np.array_equal(getattr(b0, k), getattr(b1, k)), k
    wb.close(); wb0.close()


def test_overlaps_without_cigar_native_and_python_pick_the_same_pairs(built, tmp_path, monkeypatch):
    """align_missing_native (library-held records, sequences sliced from the readers' buffers, pairs sent in batches) hands the aligner
    the pairs align_missing hands it, and files the answers under the same records -- the aligner itself replaced by a stand-in
    that answers with a digest of the pair (the device aligner has its own GPU tests, tests/test_align.py)."""
    import zlib
    from vechat_amd import align
    fx, wb = fixtures.load_plumbing()
    wb.close()
    rp, op, tp = write_inputs(fx, tmp_path, sam=False)
    lines = [ln.split("\tcg:Z:") for ln in open(op).read().strip().split("\n")]
    lines = [a if k % 5 == 0 else a + "\tcg:Z:" + cg for k, (a, cg) in enumerate(lines)]          # every fifth record keeps no CIGAR ...
    c = lines[0].split("\t"); lines.append("\t".join(["nobody"] + c[1:]))                      # ... one names an unknown read,
    lines.append("\t".join(c[:5] + [c[0]] + c[6:]))                                           # one is a self-overlap
    open(op, "w").write("\n".join(lines) + "\n")
    seen = []

    def fake(pairs, device=0, lib=None):
        seen.append(len(pairs))
        return ["%dM" % (zlib.crc32(q + b"|" + t) % 1000 + 1) for q, t in pairs], [(-1 if len(q) % 7 == 0 else len(q)) for q, _ in pairs]
    monkeypatch.setattr(align, "align_pairs", fake)
    targets, reads, ovl = seqio.read_sequences(tp), seqio.read_sequences(rp), seqio.read_overlaps(op)
    n_py = seqio.align_missing(targets, reads, ovl, 0.3)
    n_missing = sum(seen)
    seen.clear()
    nt, nr, no = seqio.NativeSequences(tp), seqio.NativeSequences(rp), seqio.NativeOverlaps(op)
    n_nat = seqio.align_missing_native(nt, nr, no, 0.3, batch_pairs=3)
    assert n_nat == n_py and sum(seen) == n_missing and max(seen) <= 3 and len(seen) > 1
    got = no.records()
    for a, b in zip(ovl, got):
        if a.cigar is not None:                  # (records the Python path leaves untouched are marked "" natively: load drops both)
            assert b.cigar == a.cigar
        else:
            assert b.cigar in (None, "")
    assert sum(1 for o in got if o.cigar) == sum(1 for o in ovl if o.cigar)


def test_native_readers_odd_inputs_and_filters(built, tmp_path):
    import random
    rnd = random.Random(9)
    # SAM records with odd CIGAR strings, both strands; MHAP; PAF without cg
    cases = ["10M", "5S10M2I3D4H", "3H7M", "12=3X1I", "3MM", "1M2", "7N3M", "2P5M", "4H3S9M1S"]
    for _ in range(60):
        body = "".join("%d%s" % (rnd.randint(1, 5000), rnd.choice("MID=XN")) for _ in range(rnd.randint(1, 300)))
        cases.append(rnd.choice(["", "12S", "3H"]) + body + rnd.choice(["", "9S", "40H"]))
    with open(tmp_path / "o.sam", "w") as f:
        f.write("@SQ\tSN:t\tLN:9\n\n")
        for i, cg in enumerate(cases):
            f.write(f"q{i}\t{16 * (i % 2)}\tt{i % 3}\t{101 + i}\t60\t{cg}\t*\t0\t0\t*\t*\n")
    _same_overlaps(seqio.NativeOverlaps(tmp_path / "o.sam").records(), seqio.read_overlaps(tmp_path / "o.sam"))
    (tmp_path / "x.mhap").write_text("2 1 0.1 10 0 0 40 40 1 4 44 120\n1 1 0.2 9 1 3 9 40 1 0 7 120\n")
    _same_overlaps(seqio.NativeOverlaps(tmp_path / "x.mhap").records(), seqio.read_overlaps(tmp_path / "x.mhap"))
    (tmp_path / "nocg.paf").write_text("r\t40\t0\t40\t+\tt\t120\t4\t44\t40\t40\t60\nr\t40\t0\t40\t-\tt\t120\t4\t44\t40\t40\t60\tcg:Z:40M\n")
    _same_overlaps(seqio.NativeOverlaps(tmp_path / "nocg.paf").records(), seqio.read_overlaps(tmp_path / "nocg.paf"))
    with pytest.raises(ValueError):
        seqio.NativeOverlaps(tmp_path / "x.txt")
    with pytest.raises(ValueError):
        seqio.NativeSequences(tmp_path / "missing.fa")
    # coordinates out of order or beyond 32 bits: both readers refuse the record and say so (they used to wrap around / go negative)
    bad = {"a.paf": "r\t40\t30\t10\t+\tt\t120\t4\t44\t40\t40\t60\n", "b.paf": "r\t40\t0\t40\t+\tt\t120\t44\t4\t40\t40\t60\n",
           "c.paf": "r\t40\t0\t5000000000\t+\tt\t120\t4\t44\t40\t40\t60\n", "d.sam": "q\t0\tt\t0\t60\t10M\t*\t0\t0\t*\t*\n",
           "e.mhap": "1 1 0.1 10 0 40 0 40 1 4 44 120\n", "f.mhap": "0 1 0.1 10 0 0 40 40 1 4 44 120\n"}
    for name, text in bad.items():
        (tmp_path / name).write_text(text)
        with pytest.raises(ValueError, match="malformed"):
            seqio.NativeOverlaps(tmp_path / name)
        with pytest.raises(ValueError, match="malformed"):
            seqio.read_overlaps(tmp_path / name)
    # the filters of Polisher::initialize, and a read that is also a target
    (tmp_path / "t.fa").write_text(">t\n" + "ACGT" * 30 + "\n")
    (tmp_path / "r.fa").write_text(">t\n" + "ACGT" * 30 + "\n>r\n" + "ACGT" * 10 + "\n")
    (tmp_path / "f.paf").write_text("t\t120\t0\t120\t+\tt\t120\t0\t120\t0\t0\t60\tcg:Z:120M\n"
                                    "r\t40\t0\t40\t+\tt\t120\t0\t100\t0\t0\t60\tcg:Z:40M\n"
                                    "r\t40\t0\t40\t+\tt\t120\t4\t44\t0\t0\t60\tcg:Z:40M\n"
                                    "zz\t4\t0\t4\t+\tt\t120\t0\t4\t0\t0\t60\tcg:Z:4M\n")
    wb = WindowBuilder(50, 10.0)
    kept, wtype = seqio.load_polisher_input_native(wb, seqio.NativeSequences(tmp_path / "t.fa"), seqio.NativeSequences(tmp_path / "r.fa"),
                                                  seqio.NativeOverlaps(tmp_path / "f.paf"))
    assert kept == 1 and wtype == 0
    wb.close()
    wb = WindowBuilder(50, 10.0)
    with pytest.raises(ValueError, match="CIGAR"):
        seqio.load_polisher_input_native(wb, seqio.NativeSequences(tmp_path / "t.fa"), seqio.NativeSequences(tmp_path / "r.fa"),
                                         seqio.NativeOverlaps(tmp_path / "nocg.paf"))
    wb.close()


def test_native_sequence_reader_large_file_in_pieces(built, tmp_path):
    """Plain files beyond 4 MB are cut at record boundaries and parsed by several threads: same records as one thread, in order."""
    import random
    rnd = random.Random(3)
    with open(tmp_path / "big.fastq", "w") as f, open(tmp_path / "big.fasta", "w") as g:
        for i in range(9000):
            n = rnd.choice([90, 500, 1200])
            s = "".join(rnd.choice("ACGTacgtN") for _ in range(n))
            q = "".join(chr(33 + rnd.randint(0, 50)) for _ in range(n)).replace("+", "@") if i % 7 else "!" * n
            f.write(f"@r{i} d\n{s}\n+\n{q}\n")
            g.write(f">r{i}\n" + "".join(s[k:k + 70] + "\n" for k in range(0, n, 70)) + ("\n" if i % 50 == 0 else ""))
    # a file that defeats the mid-file guess of a record start (every quality line opens with '@' and every sequence with '+', so a
    # quality line looks like a header): the pieces do not chain and one thread walks the file
    with open(tmp_path / "odd.fastq", "w") as f:
        for i in range(9000):
            n = rnd.choice([300, 800])
            f.write(f"@o{i}\n+{''.join(rnd.choice('ACGT') for _ in range(n - 1))}\n+\n@{''.join(chr(34 + rnd.randint(0, 40)) for _ in range(n - 1))}\n")
    for name in ("big.fastq", "big.fasta", "odd.fastq"):
        assert seqio.NativeSequences(tmp_path / name).records() == seqio.read_sequences(tmp_path / name)
    assert seqio.NativeSequences(tmp_path / "big.fastq", names_only=True).index() == seqio.sequence_index(tmp_path / "big.fastq")


@pytest.mark.parametrize("seed", [1])
def test_native_sequence_ingest_matches_the_reference_parser(tmp_path, seed):
    """the C++ reader against the reference's own bioparser + racon::Sequence, as test_sequence_ingest_matches_the_reference_parser does"""
    import random

    import oracle_api as oa
    if not oa.have_seqparse():
        pytest.skip("oracle/_ref/libvcseq.so not built (needs /root/reference)")
    rng = random.Random(seed)
    rs = lambda n: "".join(rng.choice("ACGTacgtNnRYKMSWBDHVUu-*.") for _ in range(n))
    fa, fq = "", ""
    for i in range(12):
        body = rs(rng.choice([1, 5, 70, 71, 500, 3000]))
        width = rng.choice([len(body), 60, 70])
        fa += f">t{i}" + rng.choice(["", " some description", "\ttabbed text"]) + "\n" + "".join(body[k:k + width] + "\n" for k in range(0, len(body), width))
        n = rng.choice([1, 2, 150, 2000])
        q = "!" * n if i % 4 == 2 else "".join(chr(33 + rng.randint(0, 60)) for _ in range(n))
        fq += f"@q{i}" + rng.choice(["", " len=%d" % n]) + f"\n{rs(n)}\n+\n{q}\n"
    for name, text, is_fq in (("a.fasta", fa, 0), ("a.fastq", fq, 1)):
        p = tmp_path / name
        p.write_text(text)
        pz = tmp_path / (name + ".gz")
        with gzip.open(pz, "wt") as f:
            f.write(text)
        for path in (p, pz):
            ref = oa.ref_parse_sequences(path, is_fq)
            mine = seqio.NativeSequences(path).records()
            assert [(n, d, q) for n, d, q, _, _ in ref] == mine
