"""GPU suite (-m gpu): the HIP path, driven through the C ABI, against the oracle and the golden
fixtures.  Bar: bit-exact consensus bytes and status for every window."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import fixtures
import oracle_api as oa
from vechat_amd import capi
from vechat_amd.engine import HipBatchProcessor, HipContext, VcError, create_window

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx(built):
    c = HipContext(device=0)
    yield c
    c.close()


def _check(ctx, batch, label=""):
    cons, status = ctx.consensus(batch)
    ref, pol, st = oa.oracle_run(batch, ctx.params)
    for w in range(batch.n_windows):
        assert int(status[w]) == (capi.VC_WIN_OK if pol[w] else capi.VC_WIN_UNPOLISHED), (label, w, int(status[w]), ctx.errinfo()[w])
        assert cons[w] == ref[w], (label, w, len(cons[w]), len(ref[w]))
    return st


@pytest.mark.parametrize("mode,key", [(0, "hap"), (1, "linear")])
def test_golden_windows_through_the_c_abi(built, mode, key):
    gold = fixtures.load_windows()
    batch = fixtures.fixture_batch(gold["windows"])
    c = HipContext(device=0, mode=mode)
    cons, status = c.consensus(batch)
    c.close()
    for w, win in enumerate(gold["windows"]):
        exp = win["expected"][key]
        assert cons[w].decode() == exp["consensus"], win["name"]
        assert (int(status[w]) == capi.VC_WIN_OK) == exp["polished"], win["name"]
        assert int(status[w]) in (capi.VC_WIN_OK, capi.VC_WIN_UNPOLISHED)


@pytest.mark.parametrize("seed,L,D,n,kw", [
    (1, 60, 4, 8, {}),
    (2, 60, 5, 8, dict(fastq=0, backbone_fastq=0)),
    (3, 200, 12, 16, dict(frac_partial=0.3)),
    (1001, 500, 32, 8, {}),                                   # BASELINE config B shape
    (1002, 500, 64, 6, {}),                                   # BASELINE config C shape
    (13, 500, 40, 6, dict(n_haplotypes=2, snp_rate=0.02, frac_partial=0.2)),
    (1005, 1000, 48, 2, dict(profile=capi.ONT)),              # ONT-profile 1 kb windows (config E shape, shallower)
    (17, 250, 20, 8, dict(fastq=0, backbone_fastq=1, frac_partial=0.25)),
])
def test_parity_with_oracle_on_seeded_windows(ctx, seed, L, D, n, kw):
    batch = capi.synth_batch(capi.synth_cfg(seed, L, D, **kw), 0, n)
    st = _check(ctx, batch, f"seed{seed}")
    assert ctx.stats()["cells"] == st.cells                   # same DP work counted on both sides


@pytest.mark.parametrize("seed,L,D,n,kw", [
    (1002, 500, 64, 4, {}),
    (23, 300, 25, 8, dict(frac_partial=0.3, fastq=0, backbone_fastq=0)),
    (24, 200, 12, 8, dict(n_haplotypes=2, snp_rate=0.03)),
])
def test_racon_linear_overload_parity(built, seed, L, D, n, kw):
    """Round 2 of the driver: Window::generate_consensus(engine, trim) (window.cpp:74-174)."""
    batch = capi.synth_batch(capi.synth_cfg(seed, L, D, **kw), 0, n)
    for trim in (1, 0):
        c = HipContext(device=0, mode=1, trim=trim)
        _check(c, batch, f"linear seed{seed} trim{trim}")
        c.close()


@pytest.mark.parametrize("mode,key", [(0, "hap"), (1, "linear")])
def test_plumbing_reads_to_corrected_sequences(built, mode, key):
    """BASELINE config A stand-in, end to end on the device: overlaps (CIGAR) -> windows -> HIP consensus ->
    stitched corrected reads, against what the real reference produced for the same windows."""
    fx, wb = fixtures.load_plumbing()
    batch, ids = wb.build()
    c = HipContext(device=0, mode=mode)
    cons, status = c.consensus(batch)
    c.close()
    exp = fx["expected"][key]
    assert [x.decode() for x in cons] == exp["consensus"]
    assert [int(s) == capi.VC_WIN_OK for s in status] == exp["polished"]
    st = wb.stitch(cons, status)
    assert [[n, d.decode()] for n, d in st] == exp["stitched"]
    wb.close()


@pytest.mark.parametrize("flags,key", [(["-p", "-d", "0.2", "-s", "0.2"], "hap"), ([], "linear")])
def test_command_line_from_files(built, tmp_path, capsys, flags, key):
    """reads.fastq.gz + overlaps.sam + targets.fastq -> corrected FASTA through `python -m vechat_amd.polish`."""
    from test_seqio import write_inputs
    from vechat_amd import polish
    fx, wb = fixtures.load_plumbing()
    wb.close()
    rp, op, tp = write_inputs(fx, tmp_path, sam=True)
    assert polish.main([str(rp), str(op), str(tp)] + flags) == 0
    out = capsys.readouterr().out.strip().split("\n")
    got = [[out[i][1:], out[i + 1]] for i in range(0, len(out), 2)]
    assert got == fx["expected"][key]["stitched"]


def test_command_line_distributed_path(built, tmp_path, capsys, monkeypatch):
    """The one-process-per-GPU path of the command line (cost-balanced window range, RCCL gather, rank 0 stitches),
    exercised with a single rank."""
    from test_seqio import write_inputs
    from vechat_amd import polish
    monkeypatch.setenv("VC_FORCE_DIST", "1")
    monkeypatch.setenv("MASTER_PORT", "29547")
    fx, wb = fixtures.load_plumbing()
    wb.close()
    rp, op, tp = write_inputs(fx, tmp_path, sam=True)
    assert polish.main([str(rp), str(op), str(tp), "-p", "-d", "0.2", "-s", "0.2"]) == 0
    out = capsys.readouterr().out.strip().split("\n")
    assert [[out[i][1:], out[i + 1]] for i in range(0, len(out), 2)] == fx["expected"]["hap"]["stitched"]


@pytest.mark.parametrize("mode", [0, 1])
def test_reference_windows_through_the_compiled_adapter(built, mode):
    """The drop-in boundary compiled against the reference's own headers (oracle/ref_adapter.cpp): real racon::Window
    objects filled by createWindow / add_layer, Window::generate_consensus on the CPU vs a racon::CUDABatchProcessor-named
    batch class over libvechat_hip.so.  Built only where /root/reference exists; travels to the GPU box prebuilt."""
    if not oa.have_adapter():
        pytest.skip("oracle/_ref/libvcadapter.so not built (no reference tree at build time)")
    p = capi.default_params(mode=mode)
    gold = fixtures.fixture_batch(fixtures.load_windows()["windows"])
    assert oa.adapter_run(gold, p) == 0
    synth = capi.synth_batch(capi.synth_cfg(77, 400, 24, n_haplotypes=2, snp_rate=0.02, frac_partial=0.25), 0, 12)
    assert oa.adapter_run(synth, p) == 0
    fx, wb = fixtures.load_plumbing()
    batch, _ = wb.build()
    wb.close()
    assert oa.adapter_run(batch, p) == 0


@pytest.mark.parametrize("mode,key", [(0, "hap"), (1, "linear")])
def test_polish_loop_compiled_against_the_reference(built, mode, key):
    """HipPolisher::polish() as INTEGRATION.md describes it, compiled against the reference's own window.hpp
    (oracle/ref_adapter.cpp: the loop of CUDAPolisher::polish, src/cuda/cudapolisher.cpp:217-414): windows of two targets in
    batches of three through the device, stitched with LN/RC/XC tags -- the same text as the loop run on the reference's CPU
    path, and as the fixture's expected sequences; with graph capacities too small for some windows those come back as
    overflowed, take the reference's CPU path like CUDAPolisher's failed windows (:355-379), and the text does not change."""
    if not oa.have_adapter():
        pytest.skip("oracle/_ref/libvcadapter.so not built (no reference tree at build time)")
    fx, wb = fixtures.load_plumbing()
    batch, ids = wb.build()
    wb.close()
    names = [s[0] for s in fx["sequences"][:fx["n_targets"]]]
    cov = [int(n.split("RC:i:")[1].split()[0]) for n, _ in fx["expected"][key]["stitched"]]     # targets_coverages_: kept overlaps per target
    p = capi.default_params(mode=mode)
    want = "".join(f">{n}\n{d}\n" for n, d in fx["expected"][key]["stitched"])
    cpu, ncpu = oa.adapter_polish(batch, ids, names, cov, p, cpu_only=True)
    assert ncpu == batch.n_windows and cpu == want
    gpu, ncpu = oa.adapter_polish(batch, ids, names, cov, p, batch_windows=3)
    assert ncpu == 0 and gpu == want
    small, ncpu = oa.adapter_polish(batch, ids, names, cov, p, batch_windows=4, max_nodes=1280, max_edges=2560)     # three of the seven graphs need more nodes
    assert ncpu == 3 and small == want


def test_thread_per_alignment_backtrack_agrees(built, monkeypatch):
    """The simple one-thread-per-alignment backtrack (kept as a cross-check of the cooperative k_tracew)."""
    monkeypatch.setenv("VC_TRACE_THREAD", "1")
    c = HipContext(device=0)
    for seed, L, D, n, kw in [(1002, 500, 24, 4, {}), (13, 300, 20, 6, dict(n_haplotypes=2, snp_rate=0.02, frac_partial=0.3))]:
        _check(c, capi.synth_batch(capi.synth_cfg(seed, L, D, **kw), 0, n), f"thread trace seed{seed}")
    c.close()


def test_tie_resolution_by_exact_dfs(built, monkeypatch):
    """End-cell ties are normally settled by the closure shortcut; force the exact-DFS fallback (which works
    out of an HBM workspace) on every tie and require the same bytes."""
    monkeypatch.setenv("VC_RESOLVE_FORCE_DFS", "1")
    c = HipContext(device=0)
    for seed, L, D, n, kw in [(1001, 500, 32, 8, {}), (13, 500, 40, 6, dict(n_haplotypes=2, snp_rate=0.02, frac_partial=0.2))]:
        _check(c, capi.synth_batch(capi.synth_cfg(seed, L, D, **kw), 0, n), f"dfs seed{seed}")
    c.close()


def test_raw_rows_with_a_stored_band(built, monkeypatch):
    """VC_BAND_RAW=1 in a VC_EXPERIMENTS=1 build (round 6; measured slower on 3 kb windows, so not in the default library): rows that stay raw int16 -- scores outside the byte form -- store the band
    of the rank diagonal like byte-packed rows do, and the backtrack reads it.  Same bytes as the oracle; alignments that leave the band are redone."""
    if not capi.load_hip().vc_has_experiments():
        pytest.skip("library built without VC_EXPERIMENTS")
    monkeypatch.setenv("VC_BAND_RAW", "1")
    c = HipContext(device=0, match=12, mismatch=-12, gap=-20)            # (cpl - 1) * (m - 2 g) > 255 from 6 columns per lane up: raw rows, still int16
    for seed, L, D, n, kw in [(31, 500, 24, 8, {}), (32, 300, 16, 8, dict(frac_partial=0.3, n_haplotypes=2, snp_rate=0.02))]:
        _check(c, capi.synth_batch(capi.synth_cfg(seed, L, D, **kw), 0, n), f"raw band seed{seed}")
    c.close()
    c = HipContext(device=0)                                             # default scores at 48 columns per lane (the shape of the bench's config W): raw rows as well
    st = _check(c, capi.synth_batch(capi.synth_cfg(33, 2100, 6, profile=capi.ONT), 0, 4), "raw band, 2.1 kb windows")
    c.close()


def test_lds_row_backtrack_agrees_with_the_speculative_one(built, monkeypatch):
    """k_traceb (vc_traceb.h): the backtrack walked out of LDS-resident blocks of stored rows, an experiment of round 6 (bit-identical, slower;
    profiles/r6_ab_traceb.txt) that only VC_EXPERIMENTS=1 builds carry.  Same pairs -> same bytes, statuses and work counters as k_tracew, on
    build-phase bands, re-alignment bands, whole rows of the redo pass, partial-span (local) layers and both overloads."""
    if not capi.load_hip().vc_has_experiments():
        pytest.skip("library built without VC_EXPERIMENTS")
    cases = [(capi.synth_cfg(1002, 500, 24), 16, {}),
             (capi.synth_cfg(13, 400, 20, n_haplotypes=2, snp_rate=0.02, frac_partial=0.3), 24, dict(chunk_windows=16, n_streams=2)),
             (capi.synth_cfg(77, 250, 12, frac_partial=0.5, fastq=0, backbone_fastq=0), 16, dict(mode=1)),
             (capi.synth_cfg(1005, 1000, 40, profile=capi.ONT), 6, {})]
    for cfg, n, kw in cases:
        batch = capi.synth_batch(cfg, 0, n)
        out = []
        for tb in ("0", "1"):
            monkeypatch.setenv("VC_TRACEB", tb)               # (read by vc_create)
            c = HipContext(device=0, **kw)
            cons, status = c.consensus(batch)
            st = c.stats()
            out.append((cons, [int(x) for x in status], st["cells"], st["dp_rows"], st["band_redo"], st["trace_steps"], st["trace_spec"]))
            c.close()
        assert out[0][6] > 0 and out[1][6] == 0               # the walks really were different ones: only k_tracew speculates
        assert out[0][:6] == out[1][:6], (kw, [o[1:] for o in out])
    _check(HipContext(device=0), capi.synth_batch(capi.synth_cfg(4242, 200, 12, frac_partial=0.25), 0, 8), "k_traceb against the oracle")


def test_persistent_build_pipeline_agrees_with_the_lock_step_plan(built):
    """The two execution plans of the build loop (vc_set_pipeline): lock-step launches per layer, and the persistent pipeline of
    resident forward / backtrack waves handing windows over through device-side queues.  Same bytes, same statuses, same work
    counters (cells, rows, alignments that left the band); partial-span layers (Subgraph rows inside the forward wave), two
    haplotypes, ragged depths, a window that overflows its capacity, several chunks, both overloads, and a pipeline squeezed
    into very few resident waves (every hand-over then waits for a free wave).  The pipeline is an experiment (measured slower): it is
    compiled only with VC_EXPERIMENTS=1, the default library refuses vc_set_pipeline(on)."""
    if not capi.load_hip().vc_has_experiments():
        c = HipContext(device=0)
        assert c.lib.vc_set_pipeline(c.h, 1, 0, 0) == capi.VC_ERR_ARG
        c.close()
        pytest.skip("library built without VC_EXPERIMENTS")
    cases = [(capi.synth_cfg(1002, 500, 24), 24, {}),
             (capi.synth_cfg(13, 400, 20, n_haplotypes=2, snp_rate=0.02, frac_partial=0.3), 40, dict(chunk_windows=16, n_streams=2)),
             (capi.synth_cfg(77, 250, 12, frac_partial=0.5, fastq=0, backbone_fastq=0), 30, dict(mode=1)),
             (capi.synth_cfg(5, 300, 30), 12, dict(max_nodes=512, max_edges=1408))]            # most windows outgrow 512 nodes: reported, not hidden
    for cfg, n, kw in cases:
        batch = capi.synth_batch(cfg, 0, n)
        out = []
        for pipeline in (False, True, (3, 1)):
            c = HipContext(device=0, pipeline=pipeline, **kw)
            c.submit(batch); c.run(); c.sync()
            cons, status = c.collect()
            st = c.stats()
            out.append((cons, [int(x) for x in status], st["cells"], st["dp_rows"], st["band_redo"], st["kernels"]["k_pipe"]["launches"]))
            c.close()
        assert out[0][5] == 0 and out[1][5] > 0 and out[2][5] > 0          # the plans really were different ones
        assert out[0][:4] == out[1][:4] == out[2][:4], (kw, [o[1:] for o in out])
        # (how many backtracks leave the stored band depends on the width class a short partial-span layer is run in: the
        # pipeline takes everything below the batch's two widest classes in the lower of them)
        assert out[1][4] == out[2][4] and (cfg.frac_partial > 0 or out[0][4] == out[1][4])
        if "max_nodes" not in kw:
            ref, pol, _ = oa.oracle_run(batch, capi.default_params(**{k: v for k, v in kw.items() if k not in ('chunk_windows', 'n_streams')}))
            assert out[1][0] == ref


def test_threaded_and_single_thread_host_schedulers_agree(built, monkeypatch):
    """the stage-digest tests run the single-thread lock-step host scheduler (vc_debug_stop_after), production runs one
    host thread per chunk stream and relies on k_addaln clearing the layer's tie / redo counters; same batch through both."""
    batch = capi.synth_batch(capi.synth_cfg(29, 400, 24, frac_partial=0.2, n_haplotypes=2, snp_rate=0.02), 0, 96)
    out = []
    for threads in ("1", "0"):
        monkeypatch.setenv("VC_HOST_THREADS", threads)
        c = HipContext(device=0, chunk_windows=16, n_streams=3)
        c.submit(batch); c.run(); c.sync()
        cons, status = c.collect()
        st = c.stats()
        out.append((cons, [int(x) for x in status], st["cells"], st["band_redo"], st["trace_steps"]))
        c.close()
    assert out[0] == out[1]
    assert st["band_redo"] > 0          # the redo list (and its counter reset) really was in use


def test_reserved_workspace_serves_batches_of_any_shape(built):
    """vc_reserve: the workspaces' memory in one piece, batches of different shapes (and the window type set after creation, as
    a caller does that starts the device while its reads are still being parsed) laid out inside it -- same bytes as a context
    that allocates per batch, and the reservation is what the context holds."""
    shapes = [capi.synth_cfg(61, 150, 8), capi.synth_cfg(62, 500, 24, frac_partial=0.2), capi.synth_cfg(63, 260, 40, n_haplotypes=2, snp_rate=0.02),
              capi.synth_cfg(61, 150, 8)]
    batches = [capi.synth_batch(cfg, 0, n) for cfg, n in zip(shapes, (40, 24, 12, 40))]
    plain = HipContext(device=0, window_type=1)
    want = [plain.consensus(b) for b in batches]
    plain.close()
    res = HipContext(device=0, reserve=3 << 30)
    res.set_window_type(1)
    for b, (cons, status) in zip(batches, want):
        c2, s2 = res.consensus(b)
        assert c2 == cons and list(s2) == list(status)
        assert (3 << 30) <= res.stats()["device_bytes"] < (3 << 30) + (256 << 20)       # the arena + the batch's own arrays, no second set of workspaces
    res.close()
    # a reservation too small for the batch: the rest is allocated the usual way, the results do not change
    tiny = HipContext(device=0, window_type=1, reserve=8 << 20)
    c3, s3 = tiny.consensus(batches[1])
    assert c3 == want[1][0] and list(s3) == list(want[1][1])
    tiny.close()


def test_launch_plans_of_round_four_agree(built, monkeypatch):
    """Three choices the host makes for speed must not show in the bytes: layers whose sequences fall into more than two width
    classes run as ONE forward launch built for the two widest (VC_NO_FOLD=1: one launch per class); the backtrack walks eight
    alignments of eight lanes per wave (VC_TRACE_TL=16: four of sixteen); the chunk streams are picked per batch."""
    batch = capi.synth_batch(capi.synth_cfg(71, 500, 20, frac_partial=0.5, n_haplotypes=2, snp_rate=0.02), 0, 48)
    lens = np.diff(batch.seq_off)
    classes = {next(c for c in (4, 6, 8, 10, 12, 16, 20, 24, 32) if 64 * c >= int(x)) for x in lens if x > 0}      # vc_cpl_for
    assert len(classes) > 2                                                   # the batch really spans more than two width classes
    out = []
    for env in ({}, {"VC_NO_FOLD": "1"}, {"VC_TRACE_TL": "16"}, {"VC_NO_FOLD": "1", "VC_TRACE_TL": "16"}):
        for k in ("VC_NO_FOLD", "VC_TRACE_TL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = HipContext(device=0, chunk_windows=16)
        cons, status = c.consensus(batch)
        st = c.stats()
        out.append((cons, [int(x) for x in status], st["cells"], st["trace_steps"]))
        assert st["n_streams"] == 4                                           # a small batch: four chunk streams
        c.close()
    assert all(o == out[0] for o in out[1:])
    ref, _, _ = oa.oracle_run(batch, capi.default_params(), 0, 8)
    assert out[0][0][:8] == list(ref)
    # from 12 288 windows up a context with n_streams = 0 takes eight streams; an explicit count is kept
    many = capi.synth_batch(capi.synth_cfg(72, 60, 3), 0, 12288)
    c = HipContext(device=0)
    c.submit(many); c.run(); c.sync()
    assert c.stats()["n_streams"] == 8
    assert len(c.collect()[0]) == 12288                                        # (results are handed out in the order of the runs)
    c.submit(batch); c.run(); c.sync()
    assert c.stats()["n_streams"] == 4 and c.collect()[0] == out[0][0]
    c.close()
    c = HipContext(device=0, n_streams=3)
    c.submit(many); c.run(); c.sync()
    assert c.stats()["n_streams"] == 3
    c.close()


def test_a_staged_batch_cannot_run_on_released_workspaces(built):
    """vc_release / vc_reserve give the chunk workspaces back; a batch staged before that must not be runnable (its plan would
    launch on freed pointers) -- vc_run says VC_ERR_STATE until the batch is submitted again.  Results of a finished run stay collectable."""
    batch = capi.synth_batch(capi.synth_cfg(21, 200, 10), 0, 6)
    ref, pol, _ = oa.oracle_run(batch, capi.default_params())
    c = HipContext(device=0)
    c.submit(batch)
    c.release()
    assert c.lib.vc_run(c.h) == capi.VC_ERR_STATE
    c.submit(batch); c.run(); c.sync()
    c.release()                                       # ran, not yet collected: the results live in the batch's own buffers
    cons, status = c.collect()
    assert cons == list(ref)
    assert c.lib.vc_run(c.h) == capi.VC_ERR_STATE      # "the same batch again" needs its workspaces
    c.reserve(2 << 30)
    assert c.lib.vc_run(c.h) == capi.VC_ERR_STATE
    cons, status = c.consensus(batch, retry_overflow=False)
    assert cons == list(ref)
    c.close()


def test_both_forward_kernels_give_the_same_bytes(built, monkeypatch):
    """Global alignments on byte-packed rows run on k_fwd_dt (doubly tilted unsigned rows, buffer stores: vc_fwd_dt.h); VC_DT=0 keeps them
    on k_fwd (singly tilted, signed).  Same bytes, statuses, cell counts and number of alignments that left the band -- on config C's
    shape, with partial-span layers (local alignments stay on k_fwd either way), two haplotypes, and a batch of several width classes;
    three runs each: a hazard between a store and the next row's pack would show on different windows from run to run."""
    cases = [(capi.synth_cfg(1002, 500, 64), 24), (capi.synth_cfg(13, 400, 20, n_haplotypes=2, snp_rate=0.02, frac_partial=0.3), 40),
             (capi.synth_cfg(8, 120, 9), 16), (capi.synth_cfg(1005, 1000, 40, profile=capi.ONT), 6)]
    for cfg, n in cases:
        batch = capi.synth_batch(cfg, 0, n)
        ref, pol, ost = oa.oracle_run(batch, capi.default_params())
        seen = []
        for dt in ("1", "0"):
            monkeypatch.setenv("VC_DT", dt)
            c = HipContext(device=0)
            for _ in range(3):
                cons, status = c.consensus(batch, retry_overflow=False)
                assert [int(x) for x in status] == [capi.VC_WIN_OK if p else capi.VC_WIN_UNPOLISHED for p in pol]
                assert cons == list(ref)
                st = c.stats()
                assert st["cells"] == ost.cells
                seen.append((dt, st["band_redo"], st["dp_rows"]))
            c.close()
        assert len({x[1:] for x in seen}) == 1, seen
    monkeypatch.delenv("VC_DT")


def _digest(cons):
    return hashlib.sha256(b"".join(hashlib.sha256(x).digest() for x in cons)).hexdigest()


def test_eight_stream_plan_on_config_c_shape(built):
    """What only the bench used to cover: >= 12 288 windows of BASELINE config C's real shape (500 bp x 64 reads) through the
    eight-stream plan.  Every window polished, the DP work of a sample equal to the oracle's cell count, the sample's bytes equal,
    and the whole result independent of the number of chunk streams (8 picked by the batch, 4 asked for)."""
    n = 12288
    batch = capi.synth_batch(capi.synth_cfg(1002, 500, 64), 0, n)
    c8 = HipContext(device=0)
    cons8, st8 = c8.consensus(batch)
    s8 = c8.stats()
    assert s8["n_streams"] == 8 and (st8 == capi.VC_WIN_OK).all()
    sample = batch.slice(4096, 4096 + 64)                                       # windows of the middle chunks
    ref, pol, ost = oa.oracle_run(sample, c8.params)
    assert all(pol) and cons8[4096:4096 + 64] == list(ref)
    cs, _ = c8.consensus(sample)                                               # the same windows as a batch of their own: same bytes, and the cells can be compared
    assert cs == list(ref) and c8.stats()["cells"] == ost.cells
    c8.close()
    c4 = HipContext(device=0, n_streams=4)
    cons4, st4 = c4.consensus(batch)
    assert c4.stats()["n_streams"] == 4 and c4.stats()["cells"] == s8["cells"]
    c4.close()
    assert _digest(cons4) == _digest(cons8) and (st4 == st8).all()


def test_two_contexts_driven_concurrently(built):
    """Two contexts on two host threads at the same time, no lock between them, batches of different shapes: the process-wide
    chunk streams interleave their chunks; every result must equal the one the context produces alone."""
    import threading
    ba = capi.synth_batch(capi.synth_cfg(81, 500, 24), 0, 3000)
    bb = capi.synth_batch(capi.synth_cfg(82, 300, 12, frac_partial=0.3, n_haplotypes=2, snp_rate=0.02), 0, 5000)
    ca, cb = HipContext(device=0), HipContext(device=0, n_streams=3, chunk_windows=700)
    alone = [_digest(ca.consensus(ba)[0]), _digest(cb.consensus(bb)[0])]
    got, err = [[], []], []

    def drive(k, c, b):
        try:
            for _ in range(3):
                cons, status = c.consensus(b)
                assert (status <= capi.VC_WIN_UNPOLISHED).all()
                got[k].append(_digest(cons))
        except Exception as e:                                                   # (an assertion on a thread must fail the test)
            err.append(e)

    th = [threading.Thread(target=drive, args=(0, ca, ba)), threading.Thread(target=drive, args=(1, cb, bb))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ca.close(); cb.close()
    assert not err, err
    assert got[0] == [alone[0]] * 3 and got[1] == [alone[1]] * 3
    ref, _, _ = oa.oracle_run(bb, capi.default_params(), 0, 6)
    c = HipContext(device=0)
    assert c.consensus(bb.slice(0, 6))[0] == list(ref)
    c.close()


def test_batches_queued_behind_each_other_in_one_context(built):
    """The loop of include/vechat_hip.h ("Pipelining inside one context"): submit(b[i+1]) while b[i] runs, collect() hands out
    b[i-1].  Batches of different sizes and shapes (the workspaces must grow under a running batch once), results in the order
    of the runs, every one equal to the batch run alone; a run may be repeated and a batch may stay uncollected until both slots
    are needed."""
    cfgs = [capi.synth_cfg(91, 400, 16), capi.synth_cfg(92, 500, 20, frac_partial=0.25), capi.synth_cfg(93, 200, 8, fastq=0, backbone_fastq=0)]
    parts = [capi.synth_batch(cfgs[i % 3], 100 * i, n) for i, n in enumerate((900, 1500, 40, 2000, 1, 700))]
    solo = HipContext(device=0)
    alone = [solo.consensus(p) for p in parts]
    solo.close()
    c = HipContext(device=0)
    out = []
    c.submit(parts[0]); c.run()
    for i in range(1, len(parts)):
        c.submit(parts[i]); c.run()
        out.append(c.collect())
    out.append(c.collect())
    for i, ((cons, status), (rc, rs)) in enumerate(zip(out, alone)):
        assert cons == rc and (status == rs).all(), i
    # the same batch run twice without a submit in between, then a new batch while its results wait: nothing is lost
    c.run(); c.sync()
    c.submit(parts[1]); c.run()
    again, _ = c.collect()
    assert again == alone[-1][0]
    assert c.collect()[0] == alone[1][0]
    ref, _, _ = oa.oracle_run(parts[3], c.params, 0, 4)
    assert alone[3][0][:4] == list(ref)
    c.close()


def test_config_d_rank_shard_properties(built):
    """One rank's share of BASELINE config D (1 M windows of config C's shape over 8 GPUs = 125 000 windows; rank 3's windows
    of the stream) at full size: every window polished, plausible lengths, deterministic, outliers and a random sample confirmed
    by the oracle, and the shard's result independent of how it is cut into batches (one batch of 125 000 against batches of
    20 000 queued behind each other)."""
    n, first = 125000, 3 * 125000
    batch = capi.synth_batch(capi.synth_cfg(1002, 500, 64), first, n)
    c = HipContext(device=0)
    cons, status = c.consensus(batch)
    assert (status == capi.VC_WIN_OK).all()
    lens = np.array([len(x) for x in cons])
    assert 470 < np.median(lens) < 510 and lens.max() < 640
    h1 = _digest(cons)
    rng = np.random.default_rng(11)
    for w in list(np.argsort(lens)[:3]) + list(np.argsort(lens)[-2:]) + list(rng.choice(n, size=8, replace=False)):
        ref, pol, _ = oa.oracle_run(batch, capi.default_params(), int(w), int(w) + 1)
        assert cons[int(w)] == ref[0] and pol[0], int(w)
    del cons
    parts = [batch.slice(lo, min(lo + 20000, n)) for lo in range(0, n, 20000)]
    out = []
    c.submit(parts[0]); c.run()
    for i in range(1, len(parts)):
        c.submit(parts[i]); c.run()
        out.extend(c.collect()[0])
    out.extend(c.collect()[0])
    c.close()
    assert _digest(out) == h1


def test_prune_parameters_and_rounds(built):
    batch = capi.synth_batch(capi.synth_cfg(41, 180, 14, n_haplotypes=2, snp_rate=0.03), 0, 6)
    for kw in (dict(num_prune=1), dict(num_prune=2), dict(num_prune=4, min_confidence=0.22, min_support=0.19),
               dict(min_confidence=0.0, min_support=0.0), dict(match=5, mismatch=-4, gap=-8)):
        c = HipContext(device=0, **kw)
        _check(c, batch, str(kw))
        c.close()


def test_ragged_batch_and_edge_windows(ctx):
    """Windows of different depth/length in one batch, incl. <3 sequences and backbone-only."""
    parts = [capi.synth_batch(capi.synth_cfg(50 + i, L, D, frac_partial=fp), 0, 2)
             for i, (L, D, fp) in enumerate([(80, 1, 0), (300, 30, 0.2), (64, 2, 0), (150, 3, 0.5), (500, 9, 0)])]
    wins, fl = [], []
    for p in parts:
        for w in range(p.n_windows):
            wins.append(p.window(w)); fl.append(int(p.win_fasta[w]))
    batch = capi.Batch.from_windows(wins, fl, presorted=True)
    _check(ctx, batch, "ragged")


def test_result_is_independent_of_chunking(built):
    batch = capi.synth_batch(capi.synth_cfg(61, 150, 10, frac_partial=0.2), 0, 150)
    a = HipContext(device=0, chunk_windows=64, n_streams=1)
    b = HipContext(device=0, chunk_windows=150, n_streams=3)
    ca, sa = a.consensus(batch)
    cb, sb = b.consensus(batch)
    assert ca == cb and (sa == sb).all()
    assert a.stats()["chunk_windows"] == 64
    a.close(); b.close()


def test_a_small_scratch_budget_only_changes_the_chunking(built):
    """Chunks take as many windows as the scratch budget holds (whole waves of 64, no halving): the result does not change,
    the context holds about what it was given, and several chunks per stream run."""
    batch = capi.synth_batch(capi.synth_cfg(62, 300, 24, frac_partial=0.1), 0, 700)
    ref = HipContext(device=0)
    cr, sr = ref.consensus(batch)
    small = HipContext(device=0, scratch_bytes=192 << 20, n_streams=2)
    cs, ss = small.consensus(batch)
    st = small.stats()
    assert cs == cr and (ss == sr).all()
    assert st["chunk_windows"] < 350, st["chunk_windows"]                     # without the budget: 700 windows / 2 streams
    assert st["chunk_windows"] % 64 == 0 or st["chunk_windows"] < 64, st["chunk_windows"]
    assert st["device_bytes"] < (1 << 30), st["device_bytes"]
    ref.close(); small.close()


def test_graph_images_beyond_the_lds_use_the_hbm_workspace(built):
    """Capacities whose topo / prune / consensus images exceed 160 KB are not refused: those kernels then work
    from an HBM workspace.  Same bytes as the oracle."""
    batch = capi.synth_batch(capi.synth_cfg(91, 300, 10, n_haplotypes=2, snp_rate=0.02, frac_partial=0.2), 0, 4)
    c = HipContext(device=0, max_nodes=8192, max_edges=20032)             # topo image 190 KB, prune image 200 KB
    _check(c, batch, "big images, hap")
    c.close()
    c = HipContext(device=0, mode=1, max_nodes=8192, max_edges=20032)     # + consensus image 400 KB
    _check(c, batch, "big images, linear")
    c.close()
    big = capi.synth_batch(capi.synth_cfg(1005, 1000, 48, profile=capi.ONT), 0, 2)
    c = HipContext(device=0, mode=1)                                      # the case that used to be refused at submit
    _check(c, big, "1 kb x 48 linear")
    c.close()


def test_sequences_beyond_2048_columns_and_int32_scores(built):
    """What the packed-int16 kernel declines goes to k_fwd_wide (int32 lanes, column tiles) instead of being refused:
    `-w 3000`-style windows (layers of ~3 000 columns, full and partial spans, both overloads), a window whose one layer
    is longer than 2 048 columns next to ordinary ones, and score sets whose worst case leaves int16 half way through a
    window (the reference's switch to 32-bit lanes, simd_alignment_engine_implementation.hpp:699-706)."""
    big = capi.synth_batch(capi.synth_cfg(3001, 3000, 12, frac_partial=0.25), 0, 3)
    for mode in (0, 1):
        c = HipContext(device=0, mode=mode)
        st = _check(c, big, f"3 kb windows mode{mode}")
        assert c.stats()["cells"] == st.cells
        c.close()
    # (round 5: the classes of 32+ columns per lane build a row's profile on the fly, which needs A / C / G / T only and mismatch - gap
    # == -1, VcFwdArgs::lean; the same windows with an N in one read, and with other scores, take k_fwd_wide -- same bytes, same cells)
    wins3 = [big.window(w) for w in range(2)]
    seqs3, quals3, b3, e3 = wins3[1]
    seqs3 = list(seqs3); seqs3[2] = seqs3[2][:700] + b"N" + seqs3[2][701:]
    with_n = capi.Batch.from_windows([wins3[0], (seqs3, quals3, b3, e3)], [int(big.win_fasta[0]), int(big.win_fasta[1])], presorted=True)
    c = HipContext(device=0)
    st = _check(c, with_n, "3 kb windows with an N")
    assert c.stats()["cells"] == st.cells
    c.close()
    c = HipContext(device=0, match=3, mismatch=-6, gap=-4)
    _check(c, big.slice(0, 2), "3 kb windows, mismatch - gap != -1")
    c.close()
    good = capi.synth_batch(capi.synth_cfg(81, 120, 6), 0, 3)
    wins = [good.window(w) for w in range(3)]
    seqs, quals, b, e = good.window(1)
    long_layer = (seqs[1] * 20)[:2100]
    mixed = capi.Batch.from_windows([wins[0], (seqs[:1] + [long_layer] + seqs[2:], quals[:1] + [None] + quals[2:], b, e), wins[2]],
                                    [0, 0, 0], presorted=True)
    c = HipContext(device=0)
    _check(c, mixed, "one overlong layer")
    c.close()
    deep = capi.synth_batch(capi.synth_cfg(3002, 1000, 48, profile=capi.ONT, frac_partial=0.2), 0, 2)
    c = HipContext(device=0, match=5, mismatch=-4, gap=-8)          # -8 * (len + 8 + rows) passes -31744 around 3 000 rows
    _check(c, deep, "int32 score range")
    c.close()
    # (round 4: the 3 kb windows above now fit the packed kernel -- width classes of 48 and 64 columns per lane, int16 judged on the
    # tilted matrix, vc_int16_ok -- so k_fwd_wide is met here: layers beyond 4 096 columns, a gap score that leaves int16 on rows alone)
    wider = capi.synth_batch(capi.synth_cfg(3004, 4400, 5, frac_partial=0.2), 0, 2)
    c = HipContext(device=0)
    _check(c, wider, "4.4 kb windows")
    c.close()
    edge = capi.synth_batch(capi.synth_cfg(3006, 3400, 14), 0, 2)        # the graph passes 7 928 rows (-31 744 / gap) on its way up:
    c = HipContext(device=0)                                              # early layers on the packed kernel, late ones on k_fwd_wide
    _check(c, edge, "rows across the int16 bound")
    assert c.stats()["max_nodes"] > 7928
    c.close()
    c = HipContext(device=0, match=3, mismatch=-5, gap=-40)
    _check(c, capi.synth_batch(capi.synth_cfg(3005, 600, 10), 0, 3), "gap -40")
    c.close()
    c = HipContext(device=0, match=2, mismatch=-3, gap=-12, sw_match=4, sw_mismatch=1, sw_gap=-2)     # unusual signs and magnitudes
    _check(c, capi.synth_batch(capi.synth_cfg(3003, 300, 16, frac_partial=0.3), 0, 4), "odd scores")
    c.close()


def test_a_layer_too_long_for_the_device_takes_only_its_window_out(built):
    """a single ~38 k-base layer made vc_run fail for the whole batch (k_addaln's per-pair notes did not fit the LDS).
    Such a window is now reported VC_WIN_OVERFLOW at submit and the rest of the batch is computed."""
    good = capi.synth_batch(capi.synth_cfg(83, 300, 8), 0, 4)
    wins = [good.window(w) for w in range(4)]
    seqs, quals, b, e = wins[2]
    huge = (seqs[1] * 200)[:40000]
    wins[2] = (seqs[:2] + [huge] + seqs[2:], quals[:2] + [b"5" * len(huge)] + quals[2:], b[:2] + [0] + b[2:], e[:2] + [len(seqs[0]) - 1] + e[2:])
    batch = capi.Batch.from_windows(wins, [int(good.win_fasta[w]) for w in range(4)])
    c = HipContext(device=0)
    cons, status = c.consensus(batch, retry_overflow=False)
    ref, pol, _ = oa.oracle_run(good, c.params)
    assert int(status[2]) == capi.VC_WIN_OVERFLOW and cons[2] == b""
    for w in (0, 1, 3):
        assert int(status[w]) == (capi.VC_WIN_OK if pol[w] else capi.VC_WIN_UNPOLISHED) and cons[w] == ref[w]
    c.close()


def test_overflow_is_reported_not_hidden(built):
    batch = capi.synth_batch(capi.synth_cfg(71, 200, 30), 0, 4)
    c = HipContext(device=0, max_nodes=256, max_edges=640)
    cons, status = c.consensus(batch, retry_overflow=False)
    assert all(int(s) == capi.VC_WIN_OVERFLOW for s in status) and all(len(x) == 0 for x in cons)
    # the adapter's retry (still on the device) grows the capacities until the windows fit
    _check(c, batch, "overflow-retry")
    c.close()


def test_batch_processor_mirror(ctx):
    """addWindow / generateConsensus with the reference's Window shape (cudabatch.hpp:39-59)."""
    gold = fixtures.load_windows()["windows"][:6]
    proc = HipBatchProcessor(ctx)
    for i, win in enumerate(gold):
        w = create_window(i, 0, 1, win["backbone"].encode(), win["backbone_quality"].encode())
        for l in win["layers"]:
            w.add_layer(l["seq"].encode(), None if l["qual"] is None else l["qual"].encode(), l["begin"], l["end"])
        assert proc.addWindow(w)
    assert proc.hasWindows()
    flags = proc.generateConsensus()
    for w, win, f in zip(proc.windows, gold, flags):
        assert w.consensus.decode() == win["expected"]["hap"]["consensus"] and f == win["expected"]["hap"]["polished"]
    proc.reset()
    assert not proc.hasWindows()


def test_full_size_config_b_properties(built):
    """BASELINE config B at full size (10k windows, 500 bp x 32): size-independent properties --
    determinism, all windows polished, plausible lengths, a checksum of checksums that does not
    depend on how the batch is chunked, and a random sample re-checked against the oracle."""
    n = 10000
    batch = capi.synth_batch(capi.synth_cfg(1001, 500, 32), 0, n)
    c1 = HipContext(device=0)
    cons, status = c1.consensus(batch)
    assert (status == capi.VC_WIN_OK).all()
    lens = np.array([len(x) for x in cons])
    assert 470 < np.median(lens) < 510 and lens.max() < 600
    # outliers (pruning can fragment a window's graph) are legitimate only if the oracle agrees
    for w in list(np.argsort(lens)[:4]) + list(np.argsort(lens)[-2:]):
        ref, pol, _ = oa.oracle_run(batch, capi.default_params(), int(w), int(w) + 1)
        assert cons[int(w)] == ref[0] and pol[0], int(w)
    h1 = hashlib.sha256(b"".join(hashlib.sha256(x).digest() for x in cons)).hexdigest()
    c1.run(); c1.sync()
    cons2, _ = c1.collect()
    assert hashlib.sha256(b"".join(hashlib.sha256(x).digest() for x in cons2)).hexdigest() == h1
    c1.close()
    c2 = HipContext(device=0, chunk_windows=3000, n_streams=2)
    cons3, _ = c2.consensus(batch)
    assert hashlib.sha256(b"".join(hashlib.sha256(x).digest() for x in cons3)).hexdigest() == h1
    c2.close()
    rng = np.random.default_rng(7)
    for w in rng.choice(n, size=12, replace=False):
        ref, pol, _ = oa.oracle_run(batch, capi.default_params(), int(w), int(w) + 1)
        assert cons[int(w)] == ref[0] and pol[0]


def test_config_e_full_depth_parity(built):
    """BASELINE config E at its real shape (ONT-profile 1 kb windows x 128 reads, d=0.2, 3 prune rounds): every
    window byte-identical to the oracle in both overloads, same DP work counted on both sides."""
    batch = capi.synth_batch(capi.synth_cfg(1005, 1000, 128, profile=capi.ONT), 0, 3)
    for mode in (0, 1):
        c = HipContext(device=0, mode=mode, min_confidence=0.2, num_prune=3)
        st = _check(c, batch, f"config E mode{mode}")
        assert c.stats()["cells"] == st.cells
        c.close()


def test_config_e_properties_at_2048_windows(built):
    """Config E's shape over 2 048 windows: all windows polished, a checksum of checksums that does not depend on
    chunking or stream count, 8 sampled windows re-checked against the oracle."""
    n = 2048
    batch = capi.synth_batch(capi.synth_cfg(1005, 1000, 128, profile=capi.ONT), 0, n)
    c1 = HipContext(device=0)
    cons, status = c1.consensus(batch)
    c1.close()
    assert (status == capi.VC_WIN_OK).all(), [int(s) for s in status if s != capi.VC_WIN_OK][:8]
    lens = np.array([len(x) for x in cons])
    assert 950 < np.median(lens) < 1020
    h1 = hashlib.sha256(b"".join(hashlib.sha256(x).digest() for x in cons)).hexdigest()
    c2 = HipContext(device=0, chunk_windows=600, n_streams=3)
    cons2, status2 = c2.consensus(batch)
    c2.close()
    assert (status2 == capi.VC_WIN_OK).all()
    assert hashlib.sha256(b"".join(hashlib.sha256(x).digest() for x in cons2)).hexdigest() == h1
    rng = np.random.default_rng(11)
    for w in rng.choice(n, size=8, replace=False):
        ref, pol, _ = oa.oracle_run(batch, capi.default_params(), int(w), int(w) + 1)
        assert cons[int(w)] == ref[0] and pol[0], int(w)


def test_randomised_sweep_fixed_seed(built):
    """48 random shapes / parameter sets (fixed seed), small batches, both overloads, packed and raw rows, partial spans,
    several haplotypes: consensus bytes, polished flags and DP cell counts equal to the oracle's."""
    import random
    rng = random.Random(20260929)
    ctxs = {}
    bad = []
    for case in range(48):
        L = rng.choice([60, 120, 250, 400, 500, 500, 640, 800, 1000])
        D = rng.choice([3, 5, 8, 12, 20, 32, 48])
        n = rng.choice([4, 8])
        kw = dict(frac_partial=rng.choice([0, 0, 0.2, 0.5]), n_haplotypes=rng.choice([1, 1, 2, 3]), snp_rate=rng.choice([0.005, 0.02]),
                  fastq=rng.choice([0, 1, 1]), backbone_fastq=rng.choice([0, 1, 1]), profile=rng.choice([capi.PACBIO, capi.ONT]))
        pk = dict(mode=rng.choice([0, 0, 0, 1]), num_prune=rng.choice([1, 2, 3, 3, 4]), min_confidence=rng.choice([0.2, 0.2, 0.1, 0.3]),
                  min_support=rng.choice([0.2, 0.2, 0.15]), trim=rng.choice([0, 1]))
        if rng.random() < 0.15:
            pk.update(match=5, mismatch=-4, gap=-8)            # raw (unpacked) rows
        key = tuple(sorted(pk.items()))
        if key not in ctxs:
            ctxs[key] = HipContext(device=0, **pk)
        c = ctxs[key]
        seed = rng.randrange(1, 1 << 30)
        batch = capi.synth_batch(capi.synth_cfg(seed, L, D, **kw), 0, n)
        cons, status = c.consensus(batch)
        ref, pol, st = oa.oracle_run(batch, c.params)
        mism = [w for w in range(n) if cons[w] != ref[w] or int(status[w]) != (capi.VC_WIN_OK if pol[w] else capi.VC_WIN_UNPOLISHED)]
        retried = any(e != (0, 0) for e in c.errinfo())         # an overflow retry reruns part of the batch: counters differ
        if mism or (not retried and c.stats()["cells"] != st.cells):
            bad.append((case, seed, L, D, n, kw, pk, mism, [int(x) for x in status]))
    for c in ctxs.values():
        c.close()
    assert not bad, bad[:3]


def test_bench_single_rank_through_rccl(built):
    """bench.py's distributed path (RCCL process group, barrier, gather to rank 0) with one rank on one GPU."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VC_FORCE_DIST="1", MASTER_PORT="29573")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--windows", "512",
                          "--layers", "16", "--no-cpu", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["windows_not_ok"] == 0


def _recode(batch, alphabet, rate, seed):
    """Windows of `batch` with a fraction of the layer bases replaced by letters of `alphabet` (columns then hold many
    distinct bytes: aligned groups larger than A/C/G/T/N)."""
    rng = np.random.default_rng(seed)
    wins = []
    for w in range(batch.n_windows):
        seqs, quals, b, e = batch.window(w)
        out = []
        for k, sq in enumerate(seqs):
            a = np.frombuffer(sq, dtype=np.uint8).copy()
            if k > 0:
                hit = rng.random(a.size) < rate
                a[hit] = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=int(hit.sum()))
            out.append(a.tobytes())
        wins.append((out, quals, b, e))
    return capi.Batch.from_windows(wins, [int(x) for x in batch.win_fasta], presorted=True)


def test_wide_alphabet_aligned_groups_beyond_five(built):
    """IUPAC-style reads: a column can hold far more than five distinct bytes, i.e. aligned groups (Node::aligned_nodes,
    graph.cpp:258-277) beyond A/C/G/T/N.  The aligned lists are sized from the batch's alphabet; same bytes as the oracle."""
    base = capi.synth_batch(capi.synth_cfg(301, 160, 40, frac_partial=0.2), 0, 6)
    batch = _recode(base, b"ACGTURYSWKMBDHVN", 0.35, 5)
    for mode in (0, 1):
        c = HipContext(device=0, mode=mode)
        _check(c, batch, f"iupac mode{mode}")
        c.close()
    lower = _recode(base, bytes(range(97, 123)), 0.5, 6)                     # 26 more symbols
    c = HipContext(device=0)
    _check(c, lower, "26-letter alphabet")
    c.close()


def test_stage_digests_on_the_device(built):
    """SURVEY 8(c) golden intermediates on the HIP path: the run is stopped after every build layer, every prune, every
    AddWeights round and at the end, and the window's graph (node bytes, aligned lists, edges with weights) and the
    alignment walked last are digested and compared with what the REAL reference had at that stage
    (tests/golden/stages.json).  A kernel regression shows up at the first stage it touches."""
    import json
    import os
    st = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stages.json")))
    gold = fixtures.load_windows()
    by_name = {w["name"]: w for w in gold["windows"]}
    c = HipContext(device=0)
    for name, exp in st["windows"].items():
        c.submit(fixtures.fixture_batch([by_name[name]]))
        for k, e in enumerate(exp):
            kind, index = e[0], e[1]
            got = c.stage_digest(kind, index)
            want = [e[2], e[3], int(e[4], 16), int(e[5], 16), e[6], int(e[7], 16)]
            if kind in (2, 3):
                got, want = got[:4], want[:4]
            assert got == want, (name, "first differing stage", k, e[:2], got, want)
    c.close()


def test_backtracks_that_leave_the_stored_band(built):
    """k_fwd stores 16 lanes per row around the rank diagonal; an alignment whose path leaves them (here: reads with a 140-base
    insertion or deletion against their window) is put on the redo list, stored whole and walked again.  Same bytes as the
    oracle, and the redo path did run."""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    wins, fasta = [], []
    for w in range(6):
        L = 520
        bb = rng.choice(alpha, L)
        seqs, quals, b, e = [bb.tobytes()], [b"5" * L], [0], [0]
        for k in range(14):
            r = bb.copy()
            mut = rng.random(L) < 0.06
            r[mut] = rng.choice(alpha, int(mut.sum()))
            r = r.tobytes()
            if k % 5 == 1:                                   # long insertion in the middle of the read
                cut = 200 + 10 * k
                r = r[:cut] + rng.choice(alpha, 110).tobytes() + r[cut:]
            elif k % 5 == 3:                                 # long deletion
                cut = 150 + 10 * k
                r = r[:cut] + r[cut + 140:]
            seqs.append(r); quals.append(bytes(rng.integers(40, 70, len(r), dtype=np.uint8))); b.append(0); e.append(L - 1)
        wins.append((seqs, quals, b, e)); fasta.append(0)
    batch = capi.Batch.from_windows(wins, fasta)
    c = HipContext(device=0)
    _check(c, batch, "band redo")
    st = c.stats()
    c.close()
    assert st["band_redo"] > 0, st


def test_command_line_refuses_windows_the_device_cannot_hold(built, tmp_path, capfd):
    """a window whose graph does not fit the device (here: capacity pinned to 704 nodes, no retry -- the same path a
    graph beyond the 16-bit id space takes after the retries) must not come out as silently different bytes: the command names
    the windows and exits 3; with --keep-going it says so and emits them unpolished."""
    from vechat_amd import polish
    from test_seqio import write_inputs
    fx, wb = fixtures.load_plumbing()
    wb.close()
    rp, op, tp = write_inputs(fx, tmp_path, sam=True)
    argv = [str(rp), str(op), str(tp), "-p", "-d", "0.2", "-s", "0.2", "--max-nodes", "704", "--no-capacity-retry"]
    assert polish.main(argv) == 3
    out, err = capfd.readouterr()
    assert out == "" and "could not be computed on the device" in err and "window" in err
    assert polish.main(argv + ["--keep-going", "-u"]) == 0
    out, err = capfd.readouterr()
    assert out.count(">") == fx["n_targets"] and "kept as unpolished backbone" in err


def test_two_ranks_on_two_devices(built, tmp_path):
    """bench.py --gpus 2 on a box with at least two devices: two ranks over RCCL, n_gpus == 2, every window gathered."""
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--windows", "512", "--no-cpu",
                          "--no-extras"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().split("\n")[-1])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["windows_not_ok"] == 0
    assert line["config"]["windows_per_step"] == 1024
