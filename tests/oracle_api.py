"""TEST INFRASTRUCTURE: ctypes access to the oracle (oracle/liboracle.so, our plain-C restatement)
and, when present, to the real reference built in place (oracle/_ref/libvcref_*.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from vechat_amd.capi import Batch, VcBatch, VcParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class VcoParams(C.Structure):
    _fields_ = [
        ("match", C.c_int32), ("mismatch", C.c_int32), ("gap", C.c_int32),
        ("sw_match", C.c_int32), ("sw_mismatch", C.c_int32), ("sw_gap", C.c_int32),
        ("min_confidence", C.c_double), ("min_support", C.c_double),
        ("num_prune", C.c_uint32),
        ("mode", C.c_int32), ("trim", C.c_int32), ("window_type", C.c_int32),
    ]


class VcoStats(C.Structure):
    _fields_ = [("cells", C.c_uint64), ("alignments", C.c_uint64), ("max_nodes", C.c_uint64), ("max_edges", C.c_uint64)]


def vco_params(p: VcParams):
    return VcoParams(p.match, p.mismatch, p.gap, p.sw_match, p.sw_mismatch, p.sw_gap,
                     p.min_confidence, p.min_support, p.num_prune, p.mode, p.trim, p.window_type)


_oracle = None
_ref = {}


def load_oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        lib = C.CDLL(path)
        lib.vco_run.argtypes = [C.POINTER(VcBatch), C.POINTER(VcoParams), C.c_uint32, C.c_uint32,
                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.c_uint64,
                                C.POINTER(C.c_uint8), C.POINTER(VcoStats)]
        lib.vco_run.restype = C.c_int
        lib.vco_spoa_consensus.restype = C.c_int
        lib.vco_spoa_align_probe.restype = C.c_int
        lib.vco_weight_lut.argtypes = [C.POINTER(C.c_uint32)]
        lib.vco_window_stages.argtypes = [C.POINTER(VcBatch), C.POINTER(VcoParams), C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32,
                                          C.POINTER(C.c_uint32)]
        lib.vco_window_stages.restype = C.c_int
        _oracle = lib
    return _oracle


def oracle_stages(batch: Batch, params: VcParams, w=0):
    """Per-stage digests of window w (haplotype overload): list of [kind, index, nodes, edges, hash, hash, pairs, hash]."""
    lib = load_oracle()
    vb = batch.as_struct()
    vp = vco_params(params)
    cap = 3 * int(batch.win_seq_off[w + 1] - batch.win_seq_off[w]) + 64
    rec = np.zeros(8 * cap, np.uint64)
    n = C.c_uint32(0)
    rc = lib.vco_window_stages(C.byref(vb), C.byref(vp), w, rec.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"vco_window_stages failed: {rc}")
    return [[int(x) for x in rec[8 * i:8 * i + 8]] for i in range(n.value)]


def have_ref(kind="sse41"):
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", f"libvcref_{kind}.so"))


def load_ref(kind="sse41"):
    if kind not in _ref:
        lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", f"libvcref_{kind}.so"))
        lib.vcref_window.restype = C.c_int
        lib.vcref_spoa_consensus.restype = C.c_int
        lib.vcref_spoa_align_probe.restype = C.c_int
        _ref[kind] = lib
    return _ref[kind]


def oracle_run(batch: Batch, params: VcParams, w0=0, w1=None):
    """Oracle over windows [w0,w1): returns (list of consensus bytes, polished flags, stats)."""
    lib = load_oracle()
    w1 = batch.n_windows if w1 is None else w1
    vb = batch.as_struct()
    vp = vco_params(params)
    cap = int(batch.bases.size) + 4096 * (w1 - w0) + 1024
    cons = np.zeros(cap, np.uint8)
    off = np.zeros(batch.n_windows + 1, np.uint64)
    pol = np.zeros(batch.n_windows, np.uint8)
    st = VcoStats()
    rc = lib.vco_run(C.byref(vb), C.byref(vp), w0, w1, off.ctypes.data_as(C.POINTER(C.c_uint64)),
                     cons.ctypes.data_as(C.POINTER(C.c_uint8)), cap,
                     pol.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"vco_run rc={rc}")
    out = [cons[int(off[w]):int(off[w + 1])].tobytes() for w in range(w0, w1)]
    return out, pol[w0:w1].copy(), st


def ref_window(batch: Batch, w, params: VcParams, kind="sse41"):
    """One window through the REAL reference.  Layers are handed over in their original
    add_layer() order (batch.seq_orig) because the reference sorts internally."""
    lib = load_ref(kind)
    seqs, quals, b, e = batch.window(w)
    s0 = int(batch.win_seq_off[w])
    n = len(seqs)
    orig = [int(x) for x in batch.seq_orig[s0:s0 + n]] if batch.seq_orig is not None else list(range(n))
    inv = [0] * n
    for k, o in enumerate(orig):
        inv[o] = k
    order = [inv[i] for i in range(1, n)]         # stored positions of add_layer index 1..n-1
    L = len(seqs[0])
    # backbone quality must be NUL-terminated where the reference's C-string compare stops:
    # FASTA-style windows get exactly L '!' (if_fasta true), others the real quality.
    bq = quals[0]
    if bool(batch.win_fasta[w]) != (bq == b"!" * L):
        # short last window of a FASTA target: dummy string longer than L (polisher.cpp:181,399)
        bq = bq + b"!" * 8 if not batch.win_fasta[w] and bq == b"!" * L else bq
    nl = n - 1
    SeqArr = C.c_char_p * max(nl, 1)
    U32 = C.c_uint32 * max(nl, 1)
    sa = SeqArr(*[seqs[k] for k in order]) if nl else SeqArr()
    qa = SeqArr(*[quals[k] for k in order]) if nl else SeqArr()
    la = U32(*[len(seqs[k]) for k in order]) if nl else U32()
    ba = U32(*[b[k] for k in order]) if nl else U32()
    ea = U32(*[e[k] for k in order]) if nl else U32()
    cap = sum(len(s) for s in seqs) + 4096
    out = C.create_string_buffer(cap)
    out_len = C.c_uint32(0)
    pol = C.c_int(0)
    rc = lib.vcref_window(C.c_char_p(seqs[0]), C.c_uint32(L), C.c_char_p(bq), C.c_uint32(nl), sa, la, qa, ba, ea,
                          C.c_int(params.mode), C.c_int(params.window_type), C.c_int(params.trim),
                          C.c_int(params.match), C.c_int(params.mismatch), C.c_int(params.gap),
                          C.c_double(params.min_confidence), C.c_double(params.min_support),
                          C.c_uint32(params.num_prune), out, C.c_uint32(cap), C.byref(out_len), C.byref(pol))
    if rc != 0:
        raise RuntimeError(f"vcref_window rc={rc}")
    return out.raw[:out_len.value], pol.value


def _seq_arrays(seqs, quals):
    n = len(seqs)
    SA = C.c_char_p * n
    sa = SA(*seqs)
    la = (C.c_uint32 * n)(*[len(s) for s in seqs])
    qa = SA(*quals) if quals is not None else None
    return n, sa, la, qa


def spoa_consensus(lib, prefix, seqs, quals, atype, m, n_, g):
    n, sa, la, qa = _seq_arrays(seqs, quals)
    cap = max(len(s) for s in seqs) * 4 + 1024
    out = C.create_string_buffer(cap)
    out_len = C.c_uint32(0)
    fn = getattr(lib, prefix + "_spoa_consensus")
    rc = fn(C.c_uint32(n), sa, la, qa, C.c_int(atype), C.c_int(m), C.c_int(n_), C.c_int(g), out,
            C.c_uint32(cap), C.byref(out_len))
    if rc != 0:
        raise RuntimeError(f"{prefix}_spoa_consensus rc={rc}")
    return out.raw[:out_len.value]


def spoa_align_probe(lib, prefix, seqs, quals, build_type, m, n_, g, query, query_type):
    n, sa, la, qa = _seq_arrays(seqs, quals)
    pcap = 4 * (sum(len(s) for s in seqs) + len(query)) + 64
    rcap = sum(len(s) for s in seqs) + 64
    pairs = (C.c_int32 * (2 * pcap))()
    rank = (C.c_uint32 * rcap)()
    npairs = C.c_uint32(0)
    nnodes = C.c_uint32(0)
    fn = getattr(lib, prefix + "_spoa_align_probe")
    rc = fn(C.c_uint32(n), sa, la, qa, C.c_int(build_type), C.c_int(m), C.c_int(n_), C.c_int(g),
            C.c_char_p(query), C.c_uint32(len(query)), C.c_int(query_type), pairs, C.c_uint32(pcap),
            C.byref(npairs), rank, C.c_uint32(rcap), C.byref(nnodes))
    if rc != 0:
        raise RuntimeError(f"{prefix}_spoa_align_probe rc={rc}")
    return list(pairs[:2 * npairs.value]), list(rank[:nnodes.value])


def have_adapter():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libvcadapter.so"))


def adapter_run(batch: Batch, params: VcParams):
    """oracle/ref_adapter.cpp: the reference's own racon::Window objects, once through Window::generate_consensus on the CPU
    and once through a racon::CUDABatchProcessor-named batch class backed by libvechat_hip.so.  -> number of differing windows."""
    lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libvcadapter.so"))
    lib.vcadapter_run.restype = C.c_int
    nw = batch.n_windows
    bbs, bqs, bls, off, seqs, lens, quals, begins, ends = [], [], [], [0], [], [], [], [], []
    for w in range(nw):
        s, q, b, e = batch.window(w)
        s0 = int(batch.win_seq_off[w])
        n = len(s)
        orig = [int(x) for x in batch.seq_orig[s0:s0 + n]] if batch.seq_orig is not None else list(range(n))
        inv = [0] * n
        for k, o in enumerate(orig):
            inv[o] = k
        L = len(s[0])
        bq = q[0]
        if not batch.win_fasta[w] and bq == b"!" * L:
            bq = bq + b"!" * 8                       # short last window of a FASTA target: the shared dummy string is longer than L
        bbs.append(s[0]); bqs.append(bq); bls.append(L)
        for i in range(1, n):
            k = inv[i]
            seqs.append(s[k]); lens.append(len(s[k])); quals.append(q[k]); begins.append(b[k]); ends.append(e[k])
        off.append(len(seqs))
    nl = max(len(seqs), 1)
    CP = C.c_char_p
    err = C.create_string_buffer(512)
    rc = lib.vcadapter_run(C.c_uint32(nw), (CP * nw)(*bbs), (C.c_uint32 * nw)(*bls), (CP * nw)(*bqs),
                           (C.c_uint32 * (nw + 1))(*off), (CP * nl)(*seqs), (C.c_uint32 * nl)(*lens), (CP * nl)(*quals),
                           (C.c_uint32 * nl)(*begins), (C.c_uint32 * nl)(*ends), C.c_int(params.mode), C.c_int(params.trim),
                           C.c_int(params.match), C.c_int(params.mismatch), C.c_int(params.gap),
                           C.c_double(params.min_confidence), C.c_double(params.min_support), C.c_uint32(params.num_prune),
                           err, C.c_uint32(512))
    if rc < 0:
        raise RuntimeError(f"vcadapter_run: {err.value.decode()}")
    return rc


def adapter_polish(batch: Batch, ids, target_names, target_cov, params: VcParams, cpu_only=False, batch_windows=0,
                   max_nodes=0, max_edges=0, drop_unpolished=True, fragment=True):
    """oracle/ref_adapter.cpp:vcadapter_polish -- the loop of CUDAPolisher::polish over real racon::Window objects of several
    targets (batch fill, generateConsensus, CPU path for windows the device did not take, stitching with tags).
    -> (FASTA text, number of windows that took the CPU path)."""
    lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libvcadapter.so"))
    lib.vcadapter_polish.restype = C.c_long
    nw = batch.n_windows
    bbs, bqs, bls, off, seqs, lens, quals, begins, ends = [], [], [], [0], [], [], [], [], []
    for w in range(nw):
        s, q, b, e = batch.window(w)
        s0 = int(batch.win_seq_off[w])
        n = len(s)
        orig = [int(x) for x in batch.seq_orig[s0:s0 + n]] if batch.seq_orig is not None else list(range(n))
        inv = [0] * n
        for k, o in enumerate(orig):
            inv[o] = k
        L = len(s[0])
        bq = q[0]
        if not batch.win_fasta[w] and bq == b"!" * L:
            bq = bq + b"!" * 8
        bbs.append(s[0]); bqs.append(bq); bls.append(L)
        for i in range(1, n):
            k = inv[i]
            seqs.append(s[k]); lens.append(len(s[k])); quals.append(q[k]); begins.append(b[k]); ends.append(e[k])
        off.append(len(seqs))
    nl, nt = max(len(seqs), 1), len(target_names)
    CP, U = C.c_char_p, C.c_uint32
    cap = int(batch.bases.size) + 4096 * nw + 65536
    out = C.create_string_buffer(cap)
    err = C.create_string_buffer(512)
    ncpu = U(0)
    rc = lib.vcadapter_polish(U(nw), (U * nw)(*[i for i, _ in ids]), (U * nw)(*[r for _, r in ids]), (CP * nw)(*bbs), (U * nw)(*bls),
                              (CP * nw)(*bqs), (U * (nw + 1))(*off), (CP * nl)(*seqs), (U * nl)(*lens), (CP * nl)(*quals),
                              (U * nl)(*begins), (U * nl)(*ends), U(nt), (CP * nt)(*[t.encode() for t in target_names]),
                              (U * nt)(*target_cov), C.c_int(1 if params.mode == 0 else 0), C.c_int(params.trim), C.c_int(int(fragment)),
                              C.c_int(int(drop_unpolished)), U(batch_windows), C.c_int(int(cpu_only)), U(max_nodes), U(max_edges),
                              out, C.c_uint64(cap), C.byref(ncpu), err, U(512))
    if rc < 0:
        raise RuntimeError(f"vcadapter_polish: {rc} {err.value.decode()}")
    return out.raw[:rc].decode(), int(ncpu.value)


def have_seqparse():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libvcseq.so"))


def ref_parse_sequences(path, fastq):
    """oracle/ref_seqparse.cpp: the reference's bioparser + racon::Sequence on a file
    -> [(name, data, quality|None, reverse complement, reverse quality|None)]"""
    lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libvcseq.so"))
    lib.vcref_parse_sequences.restype = C.c_long
    lib.vcref_parse_sequences.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_long]
    need = lib.vcref_parse_sequences(str(path).encode(), int(fastq), None, 0)
    if need < 0:
        raise RuntimeError("reference parser threw")
    buf = C.create_string_buffer(need + 1)
    lib.vcref_parse_sequences(str(path).encode(), int(fastq), buf, need)
    out = []
    for ln in buf.raw[:need].split(b"\n"):
        if not ln:
            continue
        name, data, q, rc, rq = ln.split(b"\t")
        out.append((name.decode(), data, None if q == b"*" else q, rc, None if rq == b"*" else rq))
    return out
