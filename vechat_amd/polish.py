"""Command line over the whole path (fragment correction, the way the VeChat driver calls vechat_racon:
`-f -p -d 0.2 -s 0.2` for round 1, `-f` for round 2; scripts/vechat:70-72,91-93):

  python -m vechat_amd.polish reads.fastq overlaps.sam targets.fastq > corrected.fasta
  python -m torch.distributed.run --nproc-per-node 8 -m vechat_amd.polish ...        (one process per GPU)

Overlaps may be SAM, PAF with cg:Z:, or plain PAF / MHAP (then they are aligned on the device first).  There is no CPU
path: without the HIP library and a GPU this exits with an error.

One process per GPU: windows are independent and stitched per target in target order (src/polisher.cpp:497-547), so the
TARGETS are split into contiguous ranges of (nearly) equal estimated work before anything large is loaded; a rank parses
the overlap records, keeps those of its own targets, loads only those targets and the reads they mention, builds and
polishes its windows, stitches its targets, and sends the finished FASTA text to rank 0 -- the one exchange of the path,
exact sizes, point to point (vechat_amd/shard.py).  Rank 0 writes the ranks' texts in rank order = target order."""
import argparse
import os
import sys

import numpy as np

from . import capi
from .engine import HipContext
from .seqio import (NativeOverlaps, NativeSequences, align_missing, align_missing_native, load_polisher_input, load_polisher_input_native,
                    native_parsers, read_inputs_native, read_overlaps, read_sequences, sequence_index)
from .windows import WindowBuilder


class DeviceWindowError(RuntimeError):
    """A window whose graph the device cannot hold even at the largest capacities."""


def target_cost(index, overlaps):
    """Estimated work per target, SURVEY 8(e): windows x depth x length ~ the bases of the overlaps laid on it (+ its own)."""
    pos = {n: k for k, (n, _) in enumerate(index)}
    cost = np.array([float(l) for _, l in index])
    for o in overlaps:
        k = pos.get(o.t_name)
        if k is not None:
            cost[k] += float(o.t_end - o.t_begin)
    return cost


def main(argv=None, shared=None):
    ap = argparse.ArgumentParser(prog="python -m vechat_amd.polish", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("sequences"); ap.add_argument("overlaps"); ap.add_argument("targets")
    ap.add_argument("-p", "--haplotype", action="store_true", help="haplotype-aware (variation graph) correction")
    ap.add_argument("-f", "--fragment-correction", action="store_true", help="accepted for compatibility: this command always runs fragment correction")
    ap.add_argument("-d", "--min-confidence", type=float, default=0.22)          # src/main.cpp:56-57
    ap.add_argument("-s", "--min-support", type=float, default=0.19)
    ap.add_argument("-k", "--num-prune", type=int, default=3)
    ap.add_argument("-w", "--window-length", type=int, default=500)
    ap.add_argument("-q", "--quality-threshold", type=float, default=10.0)
    ap.add_argument("-e", "--error-threshold", type=float, default=0.3)
    ap.add_argument("-m", "--match", type=int, default=3)
    ap.add_argument("-x", "--mismatch", type=int, default=-5)
    ap.add_argument("-g", "--gap", type=int, default=-4)
    ap.add_argument("-t", "--threads", type=int, default=1, help="accepted for compatibility (the work runs on the GPU)")
    ap.add_argument("-u", "--include-unpolished", action="store_true")
    ap.add_argument("--no-trimming", action="store_true")
    # the reference's accelerator switches (src/main.cpp:31-35,125-137; scripts/vechat:59-66 passes them with -b): this
    # polisher is always the accelerated one, so they select nothing -- accepted so that an unchanged wrapper keeps working
    ap.add_argument("-c", "--cudapoa-batches", nargs="?", const=1, default=0, type=int, help="accepted for compatibility")
    ap.add_argument("-b", "--cuda-banded-alignment", action="store_true", help="accepted for compatibility")
    ap.add_argument("--cudaaligner-batches", type=int, default=0, help="accepted for compatibility")
    ap.add_argument("--cudaaligner-band-width", type=int, default=0, help="accepted for compatibility")
    ap.add_argument("--keep-going", action="store_true", help="a window the device cannot hold (graph beyond the 16-bit id space) "
                    "keeps its backbone and counts as unpolished; without this flag the command names such windows and exits 3")
    ap.add_argument("--max-nodes", type=int, default=0, help="pin the per-window graph capacity (0: estimated from the batch)")
    ap.add_argument("--no-capacity-retry", action="store_true", help="do not re-run overflowed windows with doubled capacities")
    ap.add_argument("--streams", type=int, default=0, help="chunk streams on the device (0: default)")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    distributed = world > 1 or os.environ.get("VC_FORCE_DIST") == "1"
    device = int(os.environ.get("LOCAL_RANK", a.device)) if distributed else a.device
    # The C++ readers behind the C ABI (vc_io_read_sequences / vc_io_read_overlaps / vc_io_load) hold the records and feed the window
    # builder directly -- in one process, and in a rank of a multi-GPU run, which first plans on names, lengths and overlap records
    # (vc_io_target_cost / vc_io_rank_names) and then loads only its own targets and the reads their overlaps mention.  The Python
    # readers (VC_PY_PARSERS=1; MHAP in a multi-GPU run: its records name sequences by file position) give the same records
    # (tests/test_seqio.py) and remain as the second restatement.
    native = native_parsers() and not (distributed and str(a.overlaps).endswith((".mhap", ".mhap.gz")))      # (the readers' own test of the format, vc_io.cpp / seqio.py)
    native_targets = None
    ctx_kw = dict(device=device, mode=0 if a.haplotype else 1, min_confidence=a.min_confidence, min_support=a.min_support,
                  num_prune=a.num_prune, match=a.match, mismatch=a.mismatch, gap=a.gap, trim=0 if a.no_trimming else 1,
                  max_nodes=a.max_nodes, n_streams=a.streams)
    # shared: the driver's --in-process mode keeps ONE context for every invocation of a run (both rounds, every --split chunk); its
    # polishing parameters are switched in place (vc_set_polish_params) where they differ, creation-time settings must agree
    fixed_kw = dict(device=device, max_nodes=a.max_nodes, n_streams=a.streams)
    soft_kw = {k: v for k, v in ctx_kw.items() if k not in fixed_kw}
    reuse = None
    if shared is not None and not distributed and shared.get("ctx") is not None:
        if shared.get("fixed") == fixed_kw:
            reuse = shared["ctx"]
        else:
            shared["ctx"].close(); shared["ctx"] = None
    ctx_future = None
    all_reads = None
    r_idx = t_idx = None
    if native and not distributed:
        # the device starts up while the files are parsed (the workspaces are reserved below, once the aligner is done with the memory)
        if reuse is None:
            ctx_future = HipContext.in_background(**ctx_kw)
        native_reads, overlaps, native_targets = read_inputs_native(a.sequences, a.overlaps, a.targets)
        all_reads = native_reads
        lengths = all_reads.lengths
    elif native:
        # a rank: names and lengths of everything, every overlap record -- no bases yet
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(3) as ex:
            fo = ex.submit(NativeOverlaps, a.overlaps)
            ft = ex.submit(NativeSequences, a.targets, None, True)
            fr = ex.submit(NativeSequences, a.sequences, None, True)
            overlaps, t_idx, r_idx = fo.result(), ft.result(), fr.result()
        lengths = r_idx.lengths
    else:
        overlaps = read_overlaps(a.overlaps)
        # window type comes from the mean length of ALL reads (polisher.cpp:300-306).  One process reads the file once and keeps
        # the records; a rank of a multi-GPU run only needs names and lengths here and loads its own share of the reads below
        if distributed:
            r_index = sequence_index(a.sequences)
            lengths = [l for _, l in r_index]
        else:
            all_reads = read_sequences(a.sequences)
            lengths = [len(d) for _, d, _ in all_reads]
    if not len(lengths):
        raise ValueError("empty sequences set")
    window_type = 0 if float(sum(int(x) for x in lengths)) / float(len(lengths)) <= 1000 else 1
    keep_t = keep_r = None
    if distributed:
        import ctypes as C
        import torch
        import torch.distributed as dist
        from .seqio import _resolve_indices
        from .shard import gather_consensus, shard_range_balanced
        backend = os.environ.get("VC_DIST_BACKEND", "nccl")       # "gloo": the CPU tests of this orchestration
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
        # my contiguous range of targets, by estimated work; nothing but names and lengths has been read so far
        if native:
            lib = overlaps.lib
            cost = np.zeros(max(len(t_idx), 1), np.float64)
            if lib.vc_io_target_cost(overlaps.h, t_idx.h, cost.ctypes.data_as(C.POINTER(C.c_double))) != 0:
                raise ValueError("vc_io_target_cost failed")
            lo, hi = shard_range_balanced(cost[:len(t_idx)], rank, world)
            nn = C.c_uint64(0)
            blob = lib.vc_io_rank_names(overlaps.h, t_idx.h, r_idx.h, lo, hi, C.byref(nn))
            if not blob:
                raise ValueError("vc_io_rank_names failed")
            try:
                keep_blob = C.string_at(blob)
            finally:
                lib.vc_io_free(blob)
            native_targets = NativeSequences(a.targets, keep=set(t_idx.names()[lo:hi]))
            all_reads = NativeSequences(a.sequences, keep_blob=keep_blob)
            t_idx.close(); r_idx.close()
        else:
            t_index = sequence_index(a.targets)
            _resolve_indices(t_index, r_index, overlaps)              # MHAP names sequences by file position
            lo, hi = shard_range_balanced(target_cost(t_index, overlaps), rank, world)
            keep_t = {n for n, _ in t_index[lo:hi]}
            overlaps = [o for o in overlaps if o.t_name in keep_t]
            keep_r = {o.q_name for o in overlaps} | keep_t

    text = b""
    n_targets = n_windows = n_polished = kept = n_aligned = 0
    failure = None                                                # (exit code, message): reported by every rank through the collective below
    try:
        targets = native_targets if native else read_sequences(a.targets, keep_t)
        reads = all_reads if all_reads is not None else read_sequences(a.sequences, keep_r)
        n_targets = len(targets)
        if not distributed and not (len(targets) and len(reads) and len(overlaps)):
            raise ValueError("empty overlap set")
        if len(targets):
            wb = WindowBuilder(a.window_length, a.quality_threshold)
            if native:
                n_aligned = align_missing_native(targets, reads, overlaps, a.error_threshold, device)
                kept, _ = load_polisher_input_native(wb, targets, reads, overlaps, a.error_threshold, allow_empty=distributed)
                target_name = None
            else:
                if reads and overlaps:
                    n_aligned = align_missing(targets, reads, overlaps, a.error_threshold, device)   # PAF / MHAP without a CIGAR (overlap.cpp:205-220)
                # A rank of a multi-GPU run may own targets that keep no overlap at all (skewed input, the round-2 filters).  The
                # reference builds windows for EVERY target (polisher.cpp:389-411), so such targets still come out -- unpolished,
                # i.e. only with -u -- exactly as the single-rank run emits them.
                kept, _ = load_polisher_input(wb, targets, reads, overlaps, a.error_threshold, allow_empty=distributed)
                target_name = lambda t: targets[t][0]
            if kept or a.include_unpolished:
                # (laid out now, written slice by slice while the slices in front run on the device; the builder lives until the text is stitched)
                batch, ids, fill_windows = wb.build_streaming()
                if reuse is not None:
                    ctx = reuse
                    ctx.set_polish_params(window_type=window_type, **soft_kw)
                elif ctx_future is not None:
                    ctx, ctx_future = ctx_future.result(), None
                    ctx.set_window_type(window_type)
                else:
                    ctx = HipContext(window_type=window_type, **ctx_kw)
                if shared is not None and not distributed:
                    shared["ctx"], shared["fixed"] = ctx, fixed_kw
                if n_aligned:
                    from .align import release as release_aligner
                    release_aligner()                       # the overlap aligner's matrix buffer goes back before the workspaces are laid out
                # Large inputs: the workspaces in one piece (vc_reserve: batches of any shape are then laid out without further
                # allocations -- the counterpart of the reference sizing its batches' device memory up front, cudapolisher.cpp:229-243).
                # A small input allocates the little it needs itself, and a reservation that fails is not an error.
                if batch.n_windows >= 4096:
                    try:
                        ctx.reserve(0)
                    except Exception:                       # noqa: BLE001
                        pass
                # (a large input goes through the context in slices queued behind each other: copies in and out run beside the kernels)
                cons, status = ctx.consensus_batched(batch, retry_overflow=not a.no_capacity_retry, fill=fill_windows)
                if shared is None or distributed:
                    ctx.close()
                # Every valid window is computed on the device.  What can remain is a graph beyond the 16-bit id space after the
                # capacity retries (VC_WIN_OVERFLOW) or input the reference would throw on (VC_WIN_INVALID).  The reference's
                # accelerated polisher re-runs such windows on the CPU (cudapolisher.cpp:355-379); this command has no CPU path,
                # so it refuses to print bytes the reference would not print: it names the windows and exits non-zero, unless
                # --keep-going asks for their backbones to be kept as unpolished stretches.
                bad = [w for w in range(batch.n_windows) if int(status[w]) > capi.VC_WIN_UNPOLISHED]
                if bad:
                    if target_name is None:
                        tn = targets.names()
                        target_name = lambda t: tn[t]
                    names = ", ".join(f"target {target_name(ids[w][0])} window {ids[w][1]} (status {int(status[w])})" for w in bad[:8])
                    msg = f"{len(bad)} window(s) could not be computed on the device: {names}{' ...' if len(bad) > 8 else ''}"
                    if not a.keep_going:
                        raise DeviceWindowError(msg + "; rerun with --keep-going to emit them unpolished")
                    print(f"[vechat_amd] warning: {msg}; kept as unpolished backbone (--keep-going)", file=sys.stderr)
                    status = status.copy()
                    for w in bad:
                        cons[w] = batch.window(w)[0][0]
                        status[w] = capi.VC_WIN_UNPOLISHED
                text = b"".join(b">" + name.encode() + b"\n" + data + b"\n"
                                for name, data in wb.stitch(cons, status, drop_unpolished=not a.include_unpolished, fragment_correction=True))
                n_windows, n_polished = batch.n_windows, sum(int(s) == capi.VC_WIN_OK for s in status)
            wb.close()
    except DeviceWindowError as e:
        failure = (3, str(e))
    except Exception as e:                                        # noqa: BLE001 -- a failing rank must still take part in the collective
        if not distributed:
            raise
        failure = (1, f"{type(e).__name__}: {e}")
    if ctx_future is not None:                                   # nothing reached the device: the context made in the background goes unused
        try:
            ctx_future.result().close()
        except Exception:                                         # noqa: BLE001
            pass
    print(f"[vechat_amd] rank {rank}/{world}: {n_targets} targets, {kept} overlaps ({n_aligned} aligned on the device), {n_windows} windows, "
          f"{n_polished} polished", file=sys.stderr)
    if distributed:
        # error flags first, so that no rank is left waiting in the gather for one that failed
        codes = torch.tensor([failure[0] if failure else 0], dtype=torch.int64, device=dev)
        all_codes = [torch.zeros_like(codes) for _ in range(world)]
        dist.all_gather(all_codes, codes)
        worst = max(int(c.item()) for c in all_codes)
        if worst:
            if failure:
                print(f"[vechat_amd] rank {rank}: error: {failure[1]}", file=sys.stderr)
            dist.barrier()
            dist.destroy_process_group()
            return worst
        payload = torch.from_numpy(np.frombuffer(text + b"\0", dtype=np.uint8).copy()[:-1]).to(dev)
        call, lall = gather_consensus(payload, torch.tensor([len(text)], dtype=torch.int64, device=dev), dst=0, force=True)
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return 0
        text = call.cpu().numpy().tobytes()
    elif failure:
        print(f"[vechat_amd] error: {failure[1]}", file=sys.stderr)
        return failure[0]
    sys.stdout.write(text.decode())
    sys.stdout.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
