"""ctypes binding of the C ABI declared in include/vechat_hip.h.

This is plumbing: the product is libvechat_hip.so (HIP kernels + C ABI).  There is deliberately
no CPU fallback here -- if the HIP library is missing or no gfx950 device is visible the calls
raise.  libvechat_host.so (host helpers only, no device code) serves the CPU-only tests.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

VC_WIN_OK, VC_WIN_UNPOLISHED, VC_WIN_OVERFLOW, VC_WIN_UNSUPPORTED, VC_WIN_INVALID = range(5)
VC_OK, VC_ERR_ARG, VC_ERR_HIP, VC_ERR_NO_DEVICE, VC_ERR_STATE, VC_ERR_CAPACITY = 0, -1, -2, -3, -4, -5


class VcParams(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("match", C.c_int32), ("mismatch", C.c_int32), ("gap", C.c_int32),
        ("sw_match", C.c_int32), ("sw_mismatch", C.c_int32), ("sw_gap", C.c_int32),
        ("min_confidence", C.c_double), ("min_support", C.c_double),
        ("num_prune", C.c_uint32),
        ("mode", C.c_int32), ("trim", C.c_int32), ("window_type", C.c_int32),
        ("max_nodes", C.c_uint32), ("max_edges", C.c_uint32), ("chunk_windows", C.c_uint32),
        ("scratch_bytes", C.c_uint64),
        ("profile", C.c_int32),
        ("n_streams", C.c_uint32),
    ]


class VcBatch(C.Structure):
    _fields_ = [
        ("n_windows", C.c_uint32),
        ("win_seq_off", C.POINTER(C.c_uint32)),
        ("seq_off", C.POINTER(C.c_uint64)),
        ("seq_begin", C.POINTER(C.c_uint32)),
        ("seq_end", C.POINTER(C.c_uint32)),
        ("seq_has_qual", C.POINTER(C.c_uint8)),
        ("bases", C.POINTER(C.c_uint8)),
        ("quals", C.POINTER(C.c_uint8)),
        ("win_fasta", C.POINTER(C.c_uint8)),
    ]


class VcResult(C.Structure):
    _fields_ = [
        ("cons_off", C.POINTER(C.c_uint64)),
        ("cons", C.POINTER(C.c_uint8)),
        ("cons_cap", C.c_uint64),
        ("status", C.POINTER(C.c_uint8)),
    ]


class VcStats(C.Structure):
    _fields_ = [
        ("cells", C.c_uint64), ("alignments", C.c_uint64), ("dp_rows", C.c_uint64),
        ("far_row_reads", C.c_uint64), ("trace_steps", C.c_uint64), ("trace_spec", C.c_uint64), ("trace_rounds", C.c_uint64),
        ("n_classes", C.c_uint32),
        ("ms", C.c_double * 16),
        ("launches", C.c_uint64 * 16),
        ("names", (C.c_char * 24) * 16),
        ("max_nodes", C.c_uint32), ("max_edges", C.c_uint32), ("chunk_windows", C.c_uint32), ("n_streams", C.c_uint32),
        ("busy_ms", C.c_double * 16),
        ("band_redo", C.c_uint64), ("device_bytes", C.c_uint64),
        ("fwd_shader_cycles", C.c_uint64), ("fwd_wall_ticks", C.c_uint64),
    ]


class VcOverlapRec(C.Structure):
    _fields_ = [
        ("q_name", C.c_void_p), ("q_name_len", C.c_uint32), ("t_name", C.c_void_p), ("t_name_len", C.c_uint32),
        ("by_index", C.c_uint8), ("q_index", C.c_uint32), ("t_index", C.c_uint32), ("strand", C.c_uint8),
        ("q_begin", C.c_uint32), ("q_end", C.c_uint32), ("q_length", C.c_uint32), ("t_begin", C.c_uint32), ("t_end", C.c_uint32),
        ("length", C.c_uint32), ("error", C.c_double), ("cigar", C.c_char_p), ("dropped", C.c_uint8),
    ]


class VcSynthCfg(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("backbone_len", C.c_uint32), ("n_layers", C.c_uint32),
        ("error_rate", C.c_double),
        ("frac_ins", C.c_double), ("frac_del", C.c_double), ("frac_sub", C.c_double),
        ("frac_partial", C.c_double),
        ("fastq", C.c_int32), ("backbone_fastq", C.c_int32),
        ("n_haplotypes", C.c_int32),
        ("snp_rate", C.c_double),
    ]


def default_params(**kw):
    """Defaults the Python driver ends up with (scripts/vechat:70-72: -d 0.2 -s 0.2; main.cpp:46-61)."""
    p = VcParams(device=0, match=3, mismatch=-5, gap=-4, sw_match=3, sw_mismatch=-5, sw_gap=-4,
                 min_confidence=0.2, min_support=0.2, num_prune=3, mode=0, trim=1, window_type=1,
                 max_nodes=0, max_edges=0, chunk_windows=0, scratch_bytes=0, profile=0, n_streams=0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class Batch:
    """A packed window batch held in numpy arrays (layout of struct vc_batch)."""

    def __init__(self, win_seq_off, seq_off, seq_begin, seq_end, seq_has_qual, bases, quals,
                 win_fasta, seq_orig=None):
        self.win_seq_off = np.ascontiguousarray(win_seq_off, dtype=np.uint32)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        self.seq_begin = np.ascontiguousarray(seq_begin, dtype=np.uint32)
        self.seq_end = np.ascontiguousarray(seq_end, dtype=np.uint32)
        self.seq_has_qual = np.ascontiguousarray(seq_has_qual, dtype=np.uint8)
        self.bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.quals = np.ascontiguousarray(quals, dtype=np.uint8)
        self.win_fasta = np.ascontiguousarray(win_fasta, dtype=np.uint8)
        self.seq_orig = None if seq_orig is None else np.ascontiguousarray(seq_orig, dtype=np.uint32)
        if self.bases.size == 0:
            self.bases = np.zeros(1, np.uint8)
            self.quals = np.zeros(1, np.uint8)

    @property
    def n_windows(self):
        return int(self.win_fasta.size)

    @property
    def n_seqs(self):
        return int(self.seq_begin.size)

    def as_struct(self, cls=VcBatch):
        return cls(self.n_windows, _ptr(self.win_seq_off, C.c_uint32), _ptr(self.seq_off, C.c_uint64),
                   _ptr(self.seq_begin, C.c_uint32), _ptr(self.seq_end, C.c_uint32),
                   _ptr(self.seq_has_qual, C.c_uint8), _ptr(self.bases, C.c_uint8),
                   _ptr(self.quals, C.c_uint8), _ptr(self.win_fasta, C.c_uint8))

    def window(self, w):
        """(seqs, quals|None, begins, ends) of window w in stored (rank) order, as bytes."""
        s0, s1 = int(self.win_seq_off[w]), int(self.win_seq_off[w + 1])
        seqs, quals, b, e = [], [], [], []
        for s in range(s0, s1):
            o0, o1 = int(self.seq_off[s]), int(self.seq_off[s + 1])
            seqs.append(self.bases[o0:o1].tobytes())
            quals.append(self.quals[o0:o1].tobytes() if self.seq_has_qual[s] else None)
            b.append(int(self.seq_begin[s]))
            e.append(int(self.seq_end[s]))
        return seqs, quals, b, e

    def slice(self, lo, hi):
        """Windows [lo, hi) as a new Batch (contiguous range: offsets rebased, no per-window work)."""
        s0, s1 = int(self.win_seq_off[lo]), int(self.win_seq_off[hi])
        b0, b1 = int(self.seq_off[s0]), int(self.seq_off[s1])
        return Batch(self.win_seq_off[lo:hi + 1] - np.uint32(s0), self.seq_off[s0:s1 + 1] - np.uint64(b0),
                     self.seq_begin[s0:s1], self.seq_end[s0:s1], self.seq_has_qual[s0:s1], self.bases[b0:b1],
                     self.quals[b0:b1], self.win_fasta[lo:hi], None if self.seq_orig is None else self.seq_orig[s0:s1])

    def select(self, idx):
        """A new Batch holding windows idx (in that order)."""
        return Batch.from_windows([self.window(w) for w in idx], [int(self.win_fasta[w]) for w in idx],
                                  presorted=True)

    @staticmethod
    def from_windows(windows, fasta_flags, presorted=False, host=None):
        """windows: list of (seqs, quals|None, begins, ends); sequence 0 is the backbone.
        Unless presorted, layers are put into the reference's rank order (vc_rank_layers)."""
        wso, so, sb, se, hq, bases, quals, orig = [0], [0], [], [], [], [], [], []
        for seqs, qs, b, e in windows:
            n = len(seqs)
            order = list(range(n))
            if not presorted:
                host = host or load_host()
                rk = (C.c_uint32 * n)()
                host.vc_rank_layers((C.c_uint32 * n)(*b), n, rk)
                order = list(rk)
            for i in order:
                bases.append(seqs[i])
                quals.append(qs[i] if qs[i] is not None else b"!" * len(seqs[i]))
                so.append(so[-1] + len(seqs[i]))
                sb.append(b[i]); se.append(e[i]); hq.append(0 if qs[i] is None else 1)
                orig.append(i)
            wso.append(len(sb))
        return Batch(np.array(wso), np.array(so, dtype=np.uint64), np.array(sb), np.array(se), np.array(hq),
                     np.frombuffer(b"".join(bases), dtype=np.uint8), np.frombuffer(b"".join(quals), dtype=np.uint8),
                     np.array(fasta_flags), np.array(orig))


_host = None
_hip = None


def _declare_host(lib):
    lib.vc_rank_layers.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.vc_rank_layers.restype = None
    lib.vc_backbone_is_fasta.argtypes = [C.c_char_p, C.c_uint32]
    lib.vc_backbone_is_fasta.restype = C.c_int
    lib.vc_weight_lut.argtypes = [C.POINTER(C.c_uint32)]
    lib.vc_weight_lut.restype = None
    vp = C.c_void_p
    u32 = C.c_uint32
    lib.vc_wb_create.argtypes = [u32, C.c_double]; lib.vc_wb_create.restype = vp
    lib.vc_wb_destroy.argtypes = [vp]; lib.vc_wb_destroy.restype = None
    lib.vc_wb_last_error.argtypes = [vp]; lib.vc_wb_last_error.restype = C.c_char_p
    lib.vc_wb_add_sequence.argtypes = [vp, C.c_char_p, C.c_char_p, u32, C.c_char_p]; lib.vc_wb_add_sequence.restype = C.c_int
    lib.vc_wb_set_targets.argtypes = [vp, u32]; lib.vc_wb_set_targets.restype = C.c_int
    lib.vc_wb_add_overlap.argtypes = [vp, u32, u32, C.c_int, u32, u32, u32, u32, u32, C.c_char_p]; lib.vc_wb_add_overlap.restype = C.c_int
    lib.vc_wb_n_breaking_points.argtypes = [vp, u32]; lib.vc_wb_n_breaking_points.restype = u32
    lib.vc_wb_breaking_points.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32)]; lib.vc_wb_breaking_points.restype = None
    lib.vc_wb_build.argtypes = [vp, C.POINTER(VcBatch)]; lib.vc_wb_build.restype = C.c_int
    lib.vc_wb_build_begin.argtypes = [vp, C.POINTER(VcBatch)]; lib.vc_wb_build_begin.restype = C.c_int
    lib.vc_wb_build_fill.argtypes = [vp, C.c_uint32, C.c_uint32]; lib.vc_wb_build_fill.restype = C.c_int
    lib.vc_wb_seq_orig.argtypes = [vp]; lib.vc_wb_seq_orig.restype = C.POINTER(u32)
    lib.vc_wb_n_windows.argtypes = [vp]; lib.vc_wb_n_windows.restype = u32
    lib.vc_wb_window_target.argtypes = [vp, u32]; lib.vc_wb_window_target.restype = u32
    lib.vc_wb_window_rank.argtypes = [vp, u32]; lib.vc_wb_window_rank.restype = u32
    lib.vc_wb_window_ids.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]; lib.vc_wb_window_ids.restype = None
    lib.vc_wb_stitch.argtypes = [vp, C.POINTER(VcResult), C.c_int, C.c_int]; lib.vc_wb_stitch.restype = C.c_int
    lib.vc_wb_n_polished.argtypes = [vp]; lib.vc_wb_n_polished.restype = u32
    lib.vc_wb_polished_name.argtypes = [vp, u32]; lib.vc_wb_polished_name.restype = C.c_char_p
    lib.vc_wb_polished_data.argtypes = [vp, u32, C.POINTER(C.c_uint64)]; lib.vc_wb_polished_data.restype = C.POINTER(C.c_char)
    u64p = C.POINTER(C.c_uint64)
    lib.vc_io_read_sequences.argtypes = [C.c_char_p, C.c_char_p, C.c_int]; lib.vc_io_read_sequences.restype = vp
    lib.vc_seqset_free.argtypes = [vp]; lib.vc_seqset_free.restype = None
    lib.vc_seqset_error.argtypes = [vp]; lib.vc_seqset_error.restype = C.c_char_p
    lib.vc_seqset_size.argtypes = [vp]; lib.vc_seqset_size.restype = C.c_uint64
    for name, rt in (("vc_seqset_name_off", u64p), ("vc_seqset_data_off", u64p), ("vc_seqset_lengths", u64p), ("vc_seqset_names", C.c_void_p),
                     ("vc_seqset_data", C.c_void_p), ("vc_seqset_qual", C.c_void_p), ("vc_seqset_has_qual", C.POINTER(C.c_uint8))):
        getattr(lib, name).argtypes = [vp]; getattr(lib, name).restype = rt
    lib.vc_io_read_overlaps.argtypes = [C.c_char_p]; lib.vc_io_read_overlaps.restype = vp
    lib.vc_ovlset_free.argtypes = [vp]; lib.vc_ovlset_free.restype = None
    lib.vc_ovlset_error.argtypes = [vp]; lib.vc_ovlset_error.restype = C.c_char_p
    lib.vc_ovlset_size.argtypes = [vp]; lib.vc_ovlset_size.restype = C.c_uint64
    lib.vc_ovlset_get.argtypes = [vp, C.c_uint64, C.POINTER(VcOverlapRec)]; lib.vc_ovlset_get.restype = C.c_int
    lib.vc_ovlset_set_cigar.argtypes = [vp, C.c_uint64, C.c_char_p]; lib.vc_ovlset_set_cigar.restype = C.c_int
    lib.vc_io_load.argtypes = [vp, vp, vp, vp, C.c_double, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_uint64]; lib.vc_io_load.restype = C.c_int64
    lib.vc_io_target_cost.argtypes = [vp, vp, C.POINTER(C.c_double)]; lib.vc_io_target_cost.restype = C.c_int
    lib.vc_io_rank_names.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, u64p]; lib.vc_io_rank_names.restype = C.c_void_p
    lib.vc_io_free.argtypes = [vp]; lib.vc_io_free.restype = None
    lib.vc_synth_generate.argtypes = [C.POINTER(VcSynthCfg), C.c_uint64, C.c_uint32, C.c_uint32]
    lib.vc_synth_generate.restype = C.c_void_p
    lib.vc_synth_batch.argtypes = [C.c_void_p, C.POINTER(VcBatch)]
    lib.vc_synth_batch.restype = None
    lib.vc_synth_n_seqs.argtypes = [C.c_void_p]
    lib.vc_synth_n_seqs.restype = C.c_uint64
    lib.vc_synth_n_bytes.argtypes = [C.c_void_p]
    lib.vc_synth_n_bytes.restype = C.c_uint64
    lib.vc_synth_orig_index.argtypes = [C.c_void_p]
    lib.vc_synth_orig_index.restype = C.POINTER(C.c_uint32)
    lib.vc_synth_free.argtypes = [C.c_void_p]
    lib.vc_synth_free.restype = None
    return lib


def load_host():
    """Host helpers only (rank sort, fasta flag, LUT, synthetic generator)."""
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libvechat_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _host = _declare_host(C.CDLL(path))
    return _host


def load_hip():
    """The product library.  Raises if it has not been built; never falls back to CPU code."""
    global _hip
    if _hip is None:
        path = os.environ.get("VECHAT_HIP_LIB") or os.path.join(LIB_DIR, "libvechat_hip.so")   # override: development A/B builds
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: the HIP extension is required (no CPU fallback); "
                               "run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = _declare_host(C.CDLL(path))
        vp = C.c_void_p
        lib.vc_create.argtypes = [C.POINTER(vp), C.POINTER(VcParams)]
        lib.vc_create.restype = C.c_int
        lib.vc_destroy.argtypes = [vp]
        lib.vc_destroy.restype = None
        lib.vc_last_error.argtypes = [vp]
        lib.vc_last_error.restype = C.c_char_p
        lib.vc_submit.argtypes = [vp, C.POINTER(VcBatch)]
        lib.vc_submit.restype = C.c_int
        for name in ("vc_run", "vc_sync"):
            getattr(lib, name).argtypes = [vp]
            getattr(lib, name).restype = C.c_int
        lib.vc_result_size.argtypes = [vp, C.POINTER(C.c_uint64)]
        lib.vc_result_size.restype = C.c_int
        lib.vc_result_windows.argtypes = [vp, C.POINTER(C.c_uint32)]
        lib.vc_result_windows.restype = C.c_int
        lib.vc_collect.argtypes = [vp, C.POINTER(VcResult)]
        lib.vc_collect.restype = C.c_int
        lib.vc_collect_device.argtypes = [vp, vp, C.c_uint64, vp, vp]
        lib.vc_collect_device.restype = C.c_int
        lib.vc_get_stats.argtypes = [vp, C.POINTER(VcStats)]
        lib.vc_get_stats.restype = C.c_int
        lib.vc_debug_errinfo.argtypes = [vp, C.POINTER(C.c_uint32)]
        lib.vc_debug_errinfo.restype = C.c_int
        lib.vc_debug_stop_after.argtypes = [vp, C.c_uint32, C.c_uint32]
        lib.vc_debug_stop_after.restype = C.c_int
        lib.vc_debug_stage_digest.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]
        lib.vc_debug_stage_digest.restype = C.c_int
        lib.vc_set_profile.argtypes = [vp, C.c_int]
        lib.vc_set_profile.restype = C.c_int
        lib.vc_set_pipeline.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32]
        lib.vc_set_pipeline.restype = C.c_int
        try:                                  # (development: an A/B variant library built before an entry point was added still loads)
            lib.vc_set_polish_params.argtypes = [vp, C.POINTER(VcParams)]
            lib.vc_set_polish_params.restype = C.c_int
        except AttributeError:
            pass
        lib.vc_has_experiments.argtypes = []
        lib.vc_has_experiments.restype = C.c_int
        lib.vc_reserve.argtypes = [vp, C.c_uint64]; lib.vc_reserve.restype = C.c_int
        lib.vc_release.argtypes = [vp]; lib.vc_release.restype = C.c_int
        lib.vc_set_window_type.argtypes = [vp, C.c_int]; lib.vc_set_window_type.restype = C.c_int
        lib.vc_stream.argtypes = [vp]
        lib.vc_stream.restype = vp
        _hip = lib
    return _hip


PACBIO = dict(error_rate=0.15, frac_ins=0.40, frac_del=0.30, frac_sub=0.30)
ONT = dict(error_rate=0.10, frac_ins=0.25, frac_del=0.45, frac_sub=0.30)


def synth_cfg(seed, backbone_len, n_layers, profile=PACBIO, frac_partial=0.0, fastq=1,
              backbone_fastq=1, n_haplotypes=1, snp_rate=0.01):
    return VcSynthCfg(seed=seed, backbone_len=backbone_len, n_layers=n_layers,
                      error_rate=profile["error_rate"], frac_ins=profile["frac_ins"],
                      frac_del=profile["frac_del"], frac_sub=profile["frac_sub"],
                      frac_partial=frac_partial, fastq=fastq, backbone_fastq=backbone_fastq,
                      n_haplotypes=n_haplotypes, snp_rate=snp_rate)


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota, shared out over the ranks
    of a one-process-per-GPU launch on this node (LOCAL_WORLD_SIZE): eight ranks must not start eight full-size pools."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        local_world = 1
    return max(1, n // max(1, local_world))


def synth_batch(cfg, first, n, n_threads=0, lib=None):
    """Generate windows [first, first+n) of the synthetic stream as a Batch (rank-ordered layers)."""
    lib = lib or load_host()
    n_threads = n_threads or min(usable_cores(), 32)
    h = lib.vc_synth_generate(C.byref(cfg), first, n, n_threads)
    if not h:
        raise RuntimeError("vc_synth_generate failed")
    try:
        vb = VcBatch()
        lib.vc_synth_batch(h, C.byref(vb))
        ns = lib.vc_synth_n_seqs(h)
        nb = lib.vc_synth_n_bytes(h)
        arr = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(int(k),)).astype(dt, copy=True)
        return Batch(arr(vb.win_seq_off, n + 1, np.uint32), arr(vb.seq_off, ns + 1, np.uint64),
                     arr(vb.seq_begin, ns, np.uint32), arr(vb.seq_end, ns, np.uint32),
                     arr(vb.seq_has_qual, ns, np.uint8), arr(vb.bases, max(nb, 1), np.uint8)[:nb] if nb else np.zeros(0, np.uint8),
                     arr(vb.quals, max(nb, 1), np.uint8)[:nb] if nb else np.zeros(0, np.uint8),
                     arr(vb.win_fasta, n, np.uint8), arr(lib.vc_synth_orig_index(h), ns, np.uint32))
    finally:
        lib.vc_synth_free(h)
