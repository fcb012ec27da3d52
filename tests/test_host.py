"""CPU suite, part 2: host-side logic of the product and the C-ABI surface (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vechat_amd import capi, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "vechat_hip.h")).read()
    names = set(re.findall(r"\b(vc_[a-z_]+)\s*\(", hdr))
    assert {"vc_create", "vc_submit", "vc_run", "vc_collect", "vc_rank_layers"} <= names
    lib = C.CDLL(os.path.join(capi.LIB_DIR, "libvechat_hip.so"))
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    # ... and the other way round: nothing is exported that a binder cannot find in the header
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(capi.LIB_DIR, "libvechat_hip.so")], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-2] in ("T", "t") and l.split()[-1].startswith("vc_")}
    undeclared = sorted(n for n in exported if n not in names and not n.startswith("vc_debug_fwd_lab"))
    assert not undeclared, undeclared


def test_no_silent_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.VcError) as ei:
        engine.HipContext(device=0)
    assert "no CPU fallback" in str(ei.value) or "no HIP device" in str(ei.value)


def test_rank_layers_is_a_permutation_sorted_by_begin(built):
    host = capi.load_host()
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 17, 65, 200):
        begins = rng.integers(0, 6, size=n).astype(np.uint32)
        begins[0] = 0
        rk = np.zeros(n, np.uint32)
        host.vc_rank_layers(begins.ctypes.data_as(C.POINTER(C.c_uint32)), n, rk.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert rk[0] == 0 and sorted(rk.tolist()) == list(range(n))
        assert all(begins[rk[i]] <= begins[rk[i + 1]] for i in range(1, n - 1))


def test_backbone_is_fasta_follows_the_cstring_compare(built):
    host = capi.load_host()
    assert host.vc_backbone_is_fasta(b"!" * 50, 50) == 1
    assert host.vc_backbone_is_fasta(b"!" * 80, 50) == 0      # short last window of a FASTA target
    assert host.vc_backbone_is_fasta(b"!" * 49 + b"5", 50) == 0


def test_weight_lut_lands_on_the_reference_values(built):
    host = capi.load_host()
    lut = (C.c_uint32 * 256)()
    host.vc_weight_lut(lut)
    # Q0 -> 0, Q10 -> 900, Q20 -> 990, Q30 -> 999 (a 1-ulp different pow would flip them)
    assert [lut[33], lut[43], lut[53], lut[63]] == [0, 900, 990, 999]
    assert all(lut[c] <= 999 for c in range(33, 127))


def test_synthetic_stream_is_deterministic_and_thread_independent(built):
    cfg = capi.synth_cfg(77, 120, 9, frac_partial=0.3)
    a = capi.synth_batch(cfg, 10, 6, n_threads=1)
    b = capi.synth_batch(cfg, 10, 6, n_threads=3)
    c = capi.synth_batch(cfg, 12, 2, n_threads=1)
    assert a.bases.tobytes() == b.bases.tobytes() and (a.seq_begin == b.seq_begin).all()
    s0 = int(a.win_seq_off[2]); o0 = int(a.seq_off[s0])
    assert a.bases[o0:o0 + c.bases.size].tobytes() == c.bases.tobytes()
    assert int(a.win_seq_off[-1]) == 6 * 10


def test_window_mirror_validates_like_the_reference(built):
    w = engine.create_window(0, 0, 1, b"ACGT" * 5, b"!" * 20)
    w.add_layer(b"", None, 0, 5)                 # silently ignored (window.cpp:50-53)
    w.add_layer(b"ACGT", None, 3, 3)
    assert len(w.sequences) == 1
    with pytest.raises(ValueError):
        w.add_layer(b"ACGT", b"!!", 0, 5)
    with pytest.raises(ValueError):
        w.add_layer(b"ACGT", None, 6, 5)
    with pytest.raises(ValueError):
        engine.create_window(0, 0, 1, b"", b"")


def test_batch_slice_equals_select(built):
    b = capi.synth_batch(capi.synth_cfg(9, 80, 5, frac_partial=0.3), 0, 7)
    for lo, hi in ((0, 7), (2, 5), (6, 7), (3, 3)):
        x, y = b.slice(lo, hi), b.select(list(range(lo, hi))) if hi > lo else None
        assert x.n_windows == hi - lo
        if y is not None:
            for k in ("win_seq_off", "seq_off", "seq_begin", "seq_end", "seq_has_qual", "bases", "quals", "win_fasta"):
                assert np.array_equal(getattr(x, k), getattr(y, k)), k
