"""File formats either side of the hot path, restated from the reference's use of
bioparser and its Sequence / Overlap constructors:

  read_sequences  <- src/sequence.cpp:19-42 (upper-casing; an all-'!' quality string counts as no quality),
                     names cut at the first whitespace as bioparser does
  read_sam        <- src/overlap.cpp:44-110 (SAM constructor: unmapped flag, strand, clips, lengths, error)
  read_paf        <- src/overlap.cpp:29-42  (PAF constructor); a `cg:Z:` tag supplies the CIGAR -- without it the
                     record keeps cigar=None and align_missing() aligns it on the device (vechat_amd/align.py; the
                     reference uses edlib there, overlap.cpp:205-220)
  read_mhap       <- src/overlap.cpp:14-27  (MHAP constructor: 1-based file positions, strand = a_rc ^ b_rc); no CIGAR
  load_polisher_input <- src/polisher.cpp:207-352 (reads that are also targets share one record, self-overlaps
                     and overlaps above the error threshold are dropped, window type from the mean read length)
read_sequences is pinned against the reference's own bioparser + racon::Sequence (oracle/ref_seqparse.cpp, built in
place; tests/test_seqio.py).  The overlap readers are unpinned restatements (src/overlap.cpp needs edlib.h, which is
not in the tree).
"""
import ctypes as C
import gzip
import os
import re

import numpy as np


def _open(path):
    return gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")


def _records(path):
    """FASTA / FASTQ records of a file, one at a time (the file is streamed, not held)."""
    with _open(path) as f:
        pending = None                                   # a header line already read while collecting a FASTA record
        lineno = 0
        while True:
            ln = pending if pending is not None else f.readline()
            pending = None
            if not ln:
                return
            lineno += 1
            ln = ln.rstrip(b"\n")
            if not ln:
                continue
            if ln[:1] == b">":
                name = ln[1:].split()[0] if ln[1:].split() else b""
                parts = []
                while True:
                    nx = f.readline()
                    if not nx:
                        break
                    if nx[:1] == b">":
                        pending = nx
                        break
                    parts.append(nx.strip())
                yield name.decode(), b"".join(parts).upper(), None
            elif ln[:1] == b"@":
                name = ln[1:].split()[0] if ln[1:].split() else b""
                data = f.readline().strip().upper()
                f.readline()
                qual = f.readline().strip()
                if len(qual) != len(data):
                    raise ValueError(f"{path}: quality length differs from sequence length for {name.decode()}")
                # an all-'!' quality string counts as none (src/sequence.cpp:19-42 sums c - '!'); the byte-wise sum is only
                # needed when a byte below '!' could cancel others
                if qual.count(b"!") == len(qual) or (qual and min(qual) < 33 and sum(c - 33 for c in qual) == 0):
                    qual = None
                yield name.decode(), data, qual
            else:
                raise ValueError(f"{path}: unrecognised record at line {lineno}")


def read_sequences(path, keep=None):
    """FASTA or FASTQ (optionally .gz) -> [(name, data, quality|None)].  keep: a set of names -- the other records are
    passed over without being held (a rank of a multi-GPU run loads only the sequences of its own targets)."""
    return [r for r in _records(path) if keep is None or r[0] in keep]


def sequence_index(path):
    """[(name, length)] of every record, in file order, without keeping the data."""
    return [(n, len(d)) for n, d, _ in _records(path)]


_CIG = re.compile(rb"(\d+)([MIDNSHP=X])")
_CIG_TO_SPACE = bytes.maketrans(b"MIDNSHP=X", b" " * 9)
_CIG_OK = np.zeros(256, bool); _CIG_OK[list(b"MIDNSHP=X")] = True
_CIG_Q, _CIG_T, _CIG_CLIP = np.zeros(256, bool), np.zeros(256, bool), np.zeros(256, bool)
_CIG_Q[list(b"M=XI")] = True; _CIG_T[list(b"M=XDN")] = True; _CIG_CLIP[list(b"SH")] = True


class Overlap:
    __slots__ = ("q_name", "t_name", "strand", "q_begin", "q_end", "q_length", "t_begin", "t_end", "cigar", "error", "length")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def _check_spans(fmt, q_begin, q_end, q_length, t_begin, t_end):
    """What both readers refuse (vc_io.cpp: parse_overlap_lines has the same test): coordinates out of order or beyond 32 bits.
    The reference's Overlap constructors do not look (src/overlap.cpp:14-110); such a record would only surface much later."""
    if not (0 <= q_begin <= q_end and 0 <= t_begin <= t_end and max(q_end, q_length, t_end) < (1 << 32)):
        raise ValueError(f"malformed {fmt} record: coordinates out of order or beyond 32 bits "
                         f"(query {q_begin}..{q_end} of {q_length}, target {t_begin}..{t_end})")


def _sam_overlap(q_name, flag, t_name, pos, cigar):
    if flag & 0x4:
        return None
    if len(cigar) < 2:
        raise ValueError("missing alignment from SAM object")
    # lengths and letters of the CIGAR as two arrays (a 10 kb read has thousands of operations: no Python loop over them)
    lets = np.frombuffer(cigar.translate(None, b"0123456789"), dtype=np.uint8)
    n = None
    if len(lets) and cigar[:1].isdigit() and not cigar[-1:].isdigit() and _CIG_OK[lets].all():
        n = np.fromstring(cigar.translate(_CIG_TO_SPACE).decode("ascii", "replace"), dtype=np.int64, sep=" ")
        if len(n) != len(lets):                 # two letters in a row, or the like
            n = None
    if n is not None:
        is_q, is_t, is_clip = _CIG_Q[lets], _CIG_T[lets], _CIG_CLIP[lets]
        q_begin = int(n[0]) if is_clip[0] else 0
        q_aln, t_aln, clip = int(n[is_q].sum()), int(n[is_t].sum()), int(n[is_clip].sum())
    else:                                       # anything unusual: the plain scan over (count, letter) pairs
        ops = _CIG.findall(cigar)
        q_begin = int(ops[0][0]) if ops and ops[0][1] in b"SH" else 0
        q_aln = sum(int(k) for k, o in ops if o in b"M=XI")
        t_aln = sum(int(k) for k, o in ops if o in b"M=XDN")
        clip = sum(int(k) for k, o in ops if o in b"SH")
    strand = bool(flag & 0x10)
    q_end = q_begin + q_aln
    q_length = clip + q_aln
    if strand:
        q_begin, q_end = q_length - q_end, q_length - q_begin
    t_begin = pos - 1
    t_end = t_begin + t_aln
    length = max(q_aln, t_aln)
    _check_spans("SAM", q_begin, q_end, q_length, t_begin, t_end)
    return Overlap(q_name=q_name, t_name=t_name, strand=strand, q_begin=q_begin, q_end=q_end, q_length=q_length,
                   t_begin=t_begin, t_end=t_end, cigar=cigar.decode(), length=length,
                   error=1 - min(q_aln, t_aln) / float(length) if length else 1.0)


def read_sam(path):
    out = []
    with _open(path) as f:
        for ln in f:
            if not ln.strip() or ln[:1] == b"@":
                continue
            c = ln.rstrip(b"\n").split(b"\t")
            o = _sam_overlap(c[0].decode(), int(c[1]), c[2].decode(), int(c[3]), c[5])
            if o is not None:
                out.append(o)
    return out


def read_paf(path):
    out = []
    with _open(path) as f:
        for ln in f:
            if not ln.strip():
                continue
            c = ln.rstrip(b"\n").split(b"\t")
            qb, qe, tb, te = int(c[2]), int(c[3]), int(c[7]), int(c[8])
            cg = [x[5:] for x in c[12:] if x.startswith(b"cg:Z:")]
            _check_spans("PAF", qb, qe, int(c[1]), tb, te)
            length = max(qe - qb, te - tb)
            out.append(Overlap(q_name=c[0].decode(), t_name=c[5].decode(), strand=c[4] == b"-", q_begin=qb, q_end=qe,
                               q_length=int(c[1]), t_begin=tb, t_end=te, cigar=cg[0].decode() if cg else None, length=length,
                               error=1 - min(qe - qb, te - tb) / float(length) if length else 1.0))
    return out


def read_mhap(path):
    """MHAP: `a_id b_id error minmers a_rc a_begin a_end a_length b_rc b_begin b_end b_length`; ids are 1-based positions in
    the reads / targets files (overlap.cpp:14-27), kept here as "#<index>" names that load_polisher_input resolves."""
    out = []
    with _open(path) as f:
        for ln in f:
            c = ln.split()
            if not c:
                continue
            a_rc, ab, ae, al, b_rc, bb, be = int(c[4]), int(c[5]), int(c[6]), int(c[7]), int(c[8]), int(c[9]), int(c[10])
            _check_spans("MHAP", ab, ae, al, bb, be)
            if int(c[0]) < 1 or int(c[1]) < 1:
                raise ValueError("malformed MHAP record: sequence ids start at 1")
            length = max(ae - ab, be - bb)
            out.append(Overlap(q_name=f"#{int(c[0]) - 1}", t_name=f"#{int(c[1]) - 1}", strand=bool(a_rc ^ b_rc), q_begin=ab, q_end=ae,
                               q_length=al, t_begin=bb, t_end=be, cigar=None, length=length,
                               error=1 - min(ae - ab, be - bb) / float(length) if length else 1.0))
    return out


def read_overlaps(path):
    p = str(path)
    if p.endswith((".mhap", ".mhap.gz")):
        return read_mhap(path)
    if p.endswith((".sam", ".sam.gz")):
        return read_sam(path)
    if p.endswith((".paf", ".paf.gz")):
        return read_paf(path)
    raise ValueError(f"{path}: unsupported overlap format (valid extensions: .mhap, .mhap.gz, .paf, .paf.gz, .sam, .sam.gz)")


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _resolve_indices(targets, reads, overlaps):
    """MHAP records name sequences by file position ("#k"): turn them into names (overlap.cpp:129-166, id_to_id)."""
    for o in overlaps:
        if o.q_name.startswith("#") and o.q_name[1:].isdigit() and int(o.q_name[1:]) < len(reads):
            o.q_name = reads[int(o.q_name[1:])][0]
        if o.t_name.startswith("#") and o.t_name[1:].isdigit() and int(o.t_name[1:]) < len(targets):
            o.t_name = targets[int(o.t_name[1:])][0]


def align_missing(targets, reads, overlaps, error_threshold=0.3, device=0, shard=None):
    """Give every overlap without a CIGAR one (overlap.cpp:179-203: the aligned pieces are q[q_begin:q_end], reverse
    complemented for strand '-', and t[t_begin:t_end]).  Overlaps that load_polisher_input would drop are skipped.
    shard=(rank, world, exchange): this rank aligns its contiguous share and `exchange(list of bytes)` returns
    everybody's results in order (one process per GPU)."""
    from .align import align_pairs
    _resolve_indices(targets, reads, overlaps)
    seq = {n: d for n, d, _ in reads}
    tgt = {n: d for n, d, _ in targets}
    todo = [o for o in overlaps if o.cigar is None and o.q_name in seq and o.t_name in tgt and o.error <= error_threshold
            and o.q_name != o.t_name]
    lo, hi = 0, len(todo)
    if shard is not None:
        base, rem = divmod(len(todo), shard[1])
        lo = shard[0] * base + min(shard[0], rem)
        hi = lo + base + (1 if shard[0] < rem else 0)
    pairs = []
    for o in todo[lo:hi]:
        q = seq[o.q_name][o.q_begin:o.q_end]
        pairs.append((q.translate(_COMP)[::-1] if o.strand else q, tgt[o.t_name][o.t_begin:o.t_end]))
    cigars, dist = align_pairs(pairs, device=device)
    mine = [(cg if d >= 0 else "!").encode() for cg, d in zip(cigars, dist)]            # "!": beyond the aligner's envelope
    everyone = mine if shard is None else shard[2](mine)
    n_ok = 0
    for o, cg in zip(todo, everyone):
        if cg == b"!":                    # drop the overlap rather than guess
            o.error = 2.0
            o.cigar = ""
        else:
            o.cigar = cg.decode()
            n_ok += 1
    return n_ok


def load_polisher_input(builder, targets, reads, overlaps, error_threshold=0.3, allow_empty=False):
    """Feed a WindowBuilder the way Polisher::initialize fills its tables (fragment-correction mode, -f).
    Returns (number of overlaps kept, window_type): window_type 0 = NGS (mean read length <= 1000), 1 = TGS.
    allow_empty: one rank's share of a multi-GPU run may keep no overlap (or no read); its targets still get their
    windows, like every target does in the reference (polisher.cpp:389-411)."""
    if not targets:
        raise ValueError("empty target sequences set")
    if not reads and not allow_empty:
        raise ValueError("empty sequences set")
    _resolve_indices(targets, reads, overlaps)
    t_id, q_id = {}, {}
    for name, data, qual in targets:
        t_id[name] = builder.add_sequence(name, data, qual)
    total = 0
    for name, data, qual in reads:
        total += len(data)
        if name in t_id:                       # a read that is also a target shares its record
            tn, td, tq = targets[t_id[name]]
            if len(td) != len(data) or len(tq or b"") != len(qual or b""):
                raise ValueError(f"duplicate sequence {name} with unequal data")
            q_id[name] = t_id[name]
        else:
            q_id[name] = builder.add_sequence(name, data, qual)
    builder.set_targets(len(targets))
    kept = 0
    for o in overlaps:
        if o.q_name not in q_id or o.t_name not in t_id:
            continue
        q, t = q_id[o.q_name], t_id[o.t_name]
        if o.error > error_threshold or q == t:
            continue
        if o.cigar is None:
            raise ValueError("overlap without a CIGAR: run align_missing() first")
        builder.add_overlap(q, t, o.strand, o.q_begin, o.q_end, o.q_length, o.t_begin, o.t_end, o.cigar)
        kept += 1
    if kept == 0 and not allow_empty:
        raise ValueError("empty overlap set")
    return kept, 0 if total / float(max(len(reads), 1)) <= 1000 else 1


# ---------------------------------------------------------------------------------------------------------------------------
# The same layer in C++ behind the C ABI (vechat_amd/csrc/vc_io.cpp: vc_io_read_sequences / vc_io_read_overlaps / vc_io_load):
# what `python -m vechat_amd.polish` uses.  The Python functions above stay as the independent restatement the tests compare
# it with (and serve the multi-rank path, which plans on names and lengths).  VC_PY_PARSERS=1 selects them everywhere.
# ---------------------------------------------------------------------------------------------------------------------------
def native_parsers():
    return os.environ.get("VC_PY_PARSERS") != "1"


class NativeSequences:
    """The records of one FASTA / FASTQ(.gz) file, held by the library (vc_seqset)."""

    def __init__(self, path, keep=None, names_only=False, lib=None, keep_blob=None):
        """keep: a set of names -- the other records are passed over; keep_blob: the same as '\\n'-separated bytes (vc_io_rank_names)."""
        from . import capi
        self.lib = lib or capi.load_host()
        kn = keep_blob if keep_blob is not None else (None if keep is None else "\n".join(sorted(keep)).encode())
        self.h = self.lib.vc_io_read_sequences(os.fsencode(str(path)), kn, 1 if names_only else 0)
        err = self.lib.vc_seqset_error(self.h)
        if err:
            msg = err.decode()
            self.close()
            raise ValueError(msg)
        self.n = int(self.lib.vc_seqset_size(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.vc_seqset_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __len__(self):
        return self.n

    def _arr(self, fn, k):
        return np.ctypeslib.as_array(fn(self.h), shape=(max(k, 1),))[:k]

    @property
    def lengths(self):
        return self._arr(self.lib.vc_seqset_lengths, self.n).copy()

    def names(self):
        off = self._arr(self.lib.vc_seqset_name_off, self.n + 1)
        blob = C.string_at(self.lib.vc_seqset_names(self.h), int(off[-1])) if self.n else b""
        return [blob[int(off[i]):int(off[i + 1])].decode() for i in range(self.n)]

    def index(self):
        return list(zip(self.names(), (int(x) for x in self.lengths)))

    def records(self):
        """[(name, data, quality|None)] -- the Python readers' shape (tests, small inputs)."""
        names = self.names()
        off = self._arr(self.lib.vc_seqset_data_off, self.n + 1)
        tot = int(off[-1]) if self.n else 0
        data = C.string_at(self.lib.vc_seqset_data(self.h), tot) if tot else b""
        qp = self.lib.vc_seqset_qual(self.h)
        qual = C.string_at(qp, tot) if (qp and tot) else None
        hq = self._arr(self.lib.vc_seqset_has_qual, self.n)
        return [(names[i], data[int(off[i]):int(off[i + 1])], qual[int(off[i]):int(off[i + 1])] if (qual is not None and hq[i]) else None)
                for i in range(self.n)]


class NativeOverlaps:
    """The records of one MHAP / PAF / SAM(.gz) file, held by the library (vc_ovlset)."""

    def __init__(self, path, lib=None):
        from . import capi
        self.lib = lib or capi.load_host()
        self.h = self.lib.vc_io_read_overlaps(os.fsencode(str(path)))
        err = self.lib.vc_ovlset_error(self.h)
        if err:
            msg = err.decode()
            self.close()
            raise ValueError(msg)
        self.n = int(self.lib.vc_ovlset_size(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.vc_ovlset_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __len__(self):
        return self.n

    def get(self, i):
        from . import capi
        r = capi.VcOverlapRec()
        if self.lib.vc_ovlset_get(self.h, i, C.byref(r)) != 0:
            raise IndexError(i)
        if r.by_index:
            qn, tn = f"#{r.q_index}", f"#{r.t_index}"
        else:
            qn, tn = C.string_at(r.q_name, r.q_name_len).decode(), C.string_at(r.t_name, r.t_name_len).decode()
        return Overlap(q_name=qn, t_name=tn, strand=bool(r.strand), q_begin=r.q_begin, q_end=r.q_end, q_length=r.q_length, t_begin=r.t_begin,
                       t_end=r.t_end, cigar=None if r.cigar is None else r.cigar.decode(), error=2.0 if r.dropped else r.error, length=r.length)

    def records(self):
        return [self.get(i) for i in range(self.n)]

    def set_cigar(self, i, cigar):
        self.lib.vc_ovlset_set_cigar(self.h, i, None if cigar is None else cigar.encode())


class _OverlapRecRaw(C.Structure):
    """vc_overlap_rec with the CIGAR as a bare pointer: looking at a record does not copy its CIGAR into a Python string."""
    _fields_ = [
        ("q_name", C.c_void_p), ("q_name_len", C.c_uint32), ("t_name", C.c_void_p), ("t_name_len", C.c_uint32),
        ("by_index", C.c_uint8), ("q_index", C.c_uint32), ("t_index", C.c_uint32), ("strand", C.c_uint8),
        ("q_begin", C.c_uint32), ("q_end", C.c_uint32), ("q_length", C.c_uint32), ("t_begin", C.c_uint32), ("t_end", C.c_uint32),
        ("length", C.c_uint32), ("error", C.c_double), ("cigar", C.c_void_p), ("dropped", C.c_uint8),
    ]


def _seq_view(s):
    """(names, offsets, bases) of a NativeSequences: the bases as a view into the library's buffer, not a copy"""
    off = s._arr(s.lib.vc_seqset_data_off, s.n + 1)
    tot = int(off[-1]) if s.n else 0
    data = np.frombuffer((C.c_uint8 * tot).from_address(s.lib.vc_seqset_data(s.h)), np.uint8) if tot else np.zeros(0, np.uint8)
    return s.names(), off, data


def align_missing_native(targets, reads, overlaps, error_threshold=0.3, device=0, batch_pairs=1 << 16):
    """align_missing() for the library-held records: every overlap without a CIGAR that load would keep is aligned on the device.
    The sequences stay where the readers put them (views, sliced per pair); only the records without a CIGAR are looked at beyond
    their header, and the pairs go to the device `batch_pairs` at a time, so the host holds one batch of pieces, not a second copy
    of the input."""
    from .align import align_pairs
    lib, n = overlaps.lib, len(overlaps)
    r = _OverlapRecRaw()
    get = lib.vc_ovlset_get
    rp = C.cast(C.byref(r), get.argtypes[2])
    views = None
    pend, n_ok, any_missing = [], 0, False

    def flush():
        nonlocal n_ok
        if not pend:
            return
        _, roff, rdata = views[1]
        _, toff, tdata = views[0]
        pairs = []
        for _, qi, ti, qb, qe, tb, te, strand in pend:
            q = rdata[int(roff[qi]) + qb:int(roff[qi]) + qe].tobytes()
            pairs.append((q.translate(_COMP)[::-1] if strand else q, tdata[int(toff[ti]) + tb:int(toff[ti]) + te].tobytes()))
        cigars, dist = align_pairs(pairs, device=device)
        for (i, *_), cg, d in zip(pend, cigars, dist):
            overlaps.set_cigar(i, cg if d >= 0 else None)      # beyond the aligner's envelope: dropped rather than guessed
            n_ok += d >= 0
        pend.clear()

    for i in range(n):
        if get(overlaps.h, i, rp) != 0:
            raise IndexError(i)
        if r.cigar:
            continue
        if views is None:                                      # first record without a CIGAR: names -> position (the last record of a name wins,
            tv, rv = _seq_view(targets), _seq_view(reads)      # as in the dictionaries of align_missing)
            views = (tv, rv, {nm: k for k, nm in enumerate(tv[0])}, {nm: k for k, nm in enumerate(rv[0])})
        (tnames, toff, _), (rnames, roff, _), tmap, rmap = views
        if r.by_index:                                         # MHAP: file positions (overlap.cpp:129-166)
            qn = rnames[r.q_index] if r.q_index < len(rnames) else None
            tn = tnames[r.t_index] if r.t_index < len(tnames) else None
        else:
            qn, tn = C.string_at(r.q_name, r.q_name_len).decode(), C.string_at(r.t_name, r.t_name_len).decode()
        qi, ti = rmap.get(qn), tmap.get(tn)
        err = 2.0 if r.dropped else r.error
        if qi is None or ti is None or not err <= error_threshold or qn == tn:
            overlaps.set_cigar(i, "")                          # load filters it out anyway (unknown names, error, self-overlap)
            continue
        ql, tl = int(roff[qi + 1] - roff[qi]), int(toff[ti + 1] - toff[ti])
        pend.append((i, qi, ti, min(r.q_begin, ql), min(r.q_end, ql), min(r.t_begin, tl), min(r.t_end, tl), bool(r.strand)))
        if len(pend) >= batch_pairs:
            flush()
    flush()
    return n_ok


def load_polisher_input_native(builder, targets, reads, overlaps, error_threshold=0.3, allow_empty=False):
    """load_polisher_input() in the library (vc_io_load): (overlaps kept, window type)."""
    wt = C.c_int(0)
    err = C.create_string_buffer(512)
    kept = builder.lib.vc_io_load(builder.h, targets.h, reads.h, overlaps.h, error_threshold, 1 if allow_empty else 0, C.byref(wt), err, 512)
    if kept < 0:
        raise ValueError(err.value.decode())
    builder.n_overlaps += int(kept)
    return int(kept), int(wt.value)


def read_inputs_native(sequences, overlaps, targets):
    """The three input files at once (the readers release the interpreter lock; each cuts its file into pieces for its own threads):
    -> (NativeSequences reads, NativeOverlaps, NativeSequences targets)"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(3) as ex:
        fo = ex.submit(NativeOverlaps, overlaps)
        ft = ex.submit(NativeSequences, targets)
        fr = ex.submit(NativeSequences, sequences)
        return fr.result(), fo.result(), ft.result()
