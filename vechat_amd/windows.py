"""Window assembly and stitching over the C ABI of vc_windows.cpp: the host-side
mirror of the part of racon's Polisher that turns overlaps into windows (src/polisher.cpp:389-462) and
window results back into corrected sequences (src/polisher.cpp:520-547)."""
import ctypes as C

import numpy as np

from . import capi


class WindowBuilder:
    def __init__(self, window_length=500, quality_threshold=10.0, lib=None):
        self.lib = lib or capi.load_host()
        self.h = self.lib.vc_wb_create(window_length, quality_threshold)
        if not self.h:
            raise ValueError("window_length must be positive")
        self.n_overlaps = 0

    def close(self):
        if self.h:
            self.lib.vc_wb_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise ValueError(self.lib.vc_wb_last_error(self.h).decode())

    def add_sequence(self, name, data, quality=None):
        """Targets first (then set_targets), reads after; returns the sequence id."""
        i = self.lib.vc_wb_add_sequence(self.h, name.encode() if isinstance(name, str) else name, data, len(data), quality)
        if i < 0:
            raise ValueError("empty sequence")
        return i

    def set_targets(self, n):
        self._check(self.lib.vc_wb_set_targets(self.h, n))

    def add_overlap(self, q_id, t_id, strand, q_begin, q_end, q_length, t_begin, t_end, cigar):
        self._check(self.lib.vc_wb_add_overlap(self.h, q_id, t_id, int(strand), q_begin, q_end, q_length, t_begin, t_end,
                                               cigar.encode() if isinstance(cigar, str) else cigar))
        self.n_overlaps += 1
        return self.n_overlaps - 1

    def breaking_points(self, overlap):
        n = self.lib.vc_wb_n_breaking_points(self.h, overlap)
        t = (C.c_uint32 * max(n, 1))()
        q = (C.c_uint32 * max(n, 1))()
        self.lib.vc_wb_breaking_points(self.h, overlap, t, q)
        return [(int(t[i]), int(q[i])) for i in range(n)]

    def build_streaming(self):
        """-> (batch, ids, fill): the batch laid out but not yet written (vc_wb_build_begin; arrays are views of the builder's buffers),
        and fill(lo, hi), which writes the windows [lo, hi) (vc_wb_build_fill).  A slice may be taken and submitted once its windows are
        filled -- HipContext.consensus_batched(batch, fill=fill) fills slice i + 1 while the device works on slice i."""
        batch, ids = self.build(copy=False, _begin_only=True)
        return batch, ids, lambda lo, hi: self._check(self.lib.vc_wb_build_fill(self.h, int(lo), int(hi)))

    def build(self, copy=True, _begin_only=False):
        """-> capi.Batch of every window of every target, plus (target, rank) per window.  copy=False: the batch's arrays are
        views of the builder's buffers (valid until the next build / close) -- half a gigabyte not copied for a large input."""
        vb = capi.VcBatch()
        self._check((self.lib.vc_wb_build_begin if _begin_only else self.lib.vc_wb_build)(self.h, C.byref(vb)))
        n = int(vb.n_windows)

        def arr(p, k, dt):
            a = np.ctypeslib.as_array(p, shape=(max(int(k), 1),))[:int(k)]
            return a.astype(dt, copy=True) if copy else a
        wso = arr(vb.win_seq_off, n + 1, np.uint32)
        ns = int(wso[-1])
        so = arr(vb.seq_off, ns + 1, np.uint64)
        nb = int(so[-1])
        batch = capi.Batch(wso, so, arr(vb.seq_begin, ns, np.uint32), arr(vb.seq_end, ns, np.uint32),
                           arr(vb.seq_has_qual, ns, np.uint8), arr(vb.bases, nb, np.uint8), arr(vb.quals, nb, np.uint8),
                           arr(vb.win_fasta, n, np.uint8), arr(self.lib.vc_wb_seq_orig(self.h), ns, np.uint32))
        tg, rk = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        self.lib.vc_wb_window_ids(self.h, tg.ctypes.data_as(C.POINTER(C.c_uint32)), rk.ctypes.data_as(C.POINTER(C.c_uint32)))
        ids = list(zip(tg[:n].tolist(), rk[:n].tolist()))
        return batch, ids

    def stitch(self, consensus, status, drop_unpolished=True, fragment_correction=True):
        """consensus: list of bytes per window, status: per-window VC_WIN_*; -> [(name_with_tags, data)]"""
        off = np.zeros(len(consensus) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in consensus])
        blob = np.frombuffer(b"".join(consensus) + b"\0", dtype=np.uint8).copy()
        st = np.ascontiguousarray(status, dtype=np.uint8)
        res = capi.VcResult(off.ctypes.data_as(C.POINTER(C.c_uint64)), blob.ctypes.data_as(C.POINTER(C.c_uint8)), blob.size,
                            st.ctypes.data_as(C.POINTER(C.c_uint8)))
        self._check(self.lib.vc_wb_stitch(self.h, C.byref(res), int(drop_unpolished), int(fragment_correction)))
        out = []
        for i in range(self.lib.vc_wb_n_polished(self.h)):
            ln = C.c_uint64()
            p = self.lib.vc_wb_polished_data(self.h, i, C.byref(ln))
            out.append((self.lib.vc_wb_polished_name(self.h, i).decode(), C.string_at(p, ln.value)))
        return out
