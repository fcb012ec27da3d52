"""Overlap alignment on the device over the C ABI `vc_align`: global unit-cost alignment
with path for overlaps that come without a CIGAR (the reference calls edlib there, src/overlap.cpp:205-220)."""
import ctypes as C

import numpy as np

from . import capi


class VcAlignBatch(C.Structure):
    _fields_ = [("n", C.c_uint32), ("q_off", C.POINTER(C.c_uint64)), ("q", C.POINTER(C.c_uint8)),
                ("t_off", C.POINTER(C.c_uint64)), ("t", C.POINTER(C.c_uint8))]


def align_pairs(pairs, device=0, lib=None):
    """pairs: [(query bytes, target bytes)] -> ([cigar str], [edit distance])"""
    lib = lib or capi.load_hip()
    lib.vc_align.argtypes = [C.c_int, C.POINTER(VcAlignBatch), C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    lib.vc_align.restype = C.c_int
    lib.vc_align_last_error.restype = C.c_char_p
    n = len(pairs)
    if n == 0:
        return [], []
    qo = np.zeros(n + 1, np.uint64); to = np.zeros(n + 1, np.uint64)
    qo[1:] = np.cumsum([len(q) for q, _ in pairs]); to[1:] = np.cumsum([len(t) for _, t in pairs])
    qb = np.frombuffer(b"".join(q for q, _ in pairs) + b"\0", np.uint8).copy()
    tb = np.frombuffer(b"".join(t for _, t in pairs) + b"\0", np.uint8).copy()
    b = VcAlignBatch(n, qo.ctypes.data_as(C.POINTER(C.c_uint64)), qb.ctypes.data_as(C.POINTER(C.c_uint8)),
                     to.ctypes.data_as(C.POINTER(C.c_uint64)), tb.ctypes.data_as(C.POINTER(C.c_uint8)))
    cap = int(qo[-1] + to[-1]) * 6 + 16 * n + 64
    buf = C.create_string_buffer(cap)
    off = np.zeros(n + 1, np.uint64)
    dist = np.zeros(n, np.int32)
    rc = lib.vc_align(device, C.byref(b), buf, cap, off.ctypes.data_as(C.POINTER(C.c_uint64)), dist.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise RuntimeError(f"vc_align failed ({rc}): {lib.vc_align_last_error().decode()}")
    raw = buf.raw
    return [raw[int(off[k]):int(off[k + 1]) - 1].decode() for k in range(n)], [int(x) for x in dist]


def release(lib=None):
    """vc_align_release: the aligner's (large) matrix buffer goes back to the device."""
    lib = lib or capi.load_hip()
    lib.vc_align_release.restype = None
    lib.vc_align_release()
