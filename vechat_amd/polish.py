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
from .seqio import align_missing, load_polisher_input, read_overlaps, read_sequences, sequence_index
from .windows import WindowBuilder


def target_cost(index, overlaps):
    """Estimated work per target, SURVEY 8(e): windows x depth x length ~ the bases of the overlaps laid on it (+ its own)."""
    pos = {n: k for k, (n, _) in enumerate(index)}
    cost = np.array([float(l) for _, l in index])
    for o in overlaps:
        k = pos.get(o.t_name)
        if k is not None:
            cost[k] += float(o.t_end - o.t_begin)
    return cost


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m vechat_amd.polish", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("sequences"); ap.add_argument("overlaps"); ap.add_argument("targets")
    ap.add_argument("-p", "--haplotype", action="store_true", help="haplotype-aware (variation graph) correction")
    ap.add_argument("-f", "--fragment-correction", action="store_true", help="accepted for compatibility: this command always runs fragment correction")
    ap.add_argument("-d", "--min-confidence", type=float, default=0.2)
    ap.add_argument("-s", "--min-support", type=float, default=0.2)
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
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    distributed = world > 1 or os.environ.get("VC_FORCE_DIST") == "1"
    device = int(os.environ.get("LOCAL_RANK", a.device)) if distributed else a.device
    overlaps = read_overlaps(a.overlaps)
    r_index = sequence_index(a.sequences)                         # window type comes from the mean length of ALL reads (polisher.cpp:300-306)
    if not r_index:
        raise ValueError("empty sequences set")
    window_type = 0 if sum(l for _, l in r_index) / float(len(r_index)) <= 1000 else 1
    keep_t = keep_r = None
    if distributed:
        import torch
        import torch.distributed as dist
        from .seqio import _resolve_indices
        from .shard import gather_consensus, shard_range_balanced
        backend = os.environ.get("VC_DIST_BACKEND", "nccl")       # "gloo": the CPU tests of this orchestration
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
        # my contiguous range of targets, by estimated work; nothing but names and lengths has been read so far
        t_index = sequence_index(a.targets)
        _resolve_indices(t_index, r_index, overlaps)              # MHAP names sequences by file position
        lo, hi = shard_range_balanced(target_cost(t_index, overlaps), rank, world)
        keep_t = {n for n, _ in t_index[lo:hi]}
        overlaps = [o for o in overlaps if o.t_name in keep_t]
        keep_r = {o.q_name for o in overlaps} | keep_t

    targets, reads = read_sequences(a.targets, keep_t), read_sequences(a.sequences, keep_r)
    text = b""
    n_windows = n_polished = kept = n_aligned = 0
    if targets and reads and overlaps:
        wb = WindowBuilder(a.window_length, a.quality_threshold)
        n_aligned = align_missing(targets, reads, overlaps, a.error_threshold, device)      # PAF / MHAP without a CIGAR (overlap.cpp:205-220)
        try:
            kept, _ = load_polisher_input(wb, targets, reads, overlaps, a.error_threshold)
        except ValueError as e:
            if not distributed or "empty overlap set" not in str(e):
                raise
            kept = 0
        if kept:
            batch, ids = wb.build()
            ctx = HipContext(device=device, mode=0 if a.haplotype else 1, min_confidence=a.min_confidence, min_support=a.min_support,
                             num_prune=a.num_prune, match=a.match, mismatch=a.mismatch, gap=a.gap, trim=0 if a.no_trimming else 1,
                             window_type=window_type)
            cons, status = ctx.consensus(batch)
            ctx.close()
            # every valid window is computed on the device; what can remain is a graph beyond the 16-bit id space after the capacity
            # retries (VC_WIN_OVERFLOW) or input the reference would throw on (VC_WIN_INVALID): such a window keeps its backbone
            # and counts as unpolished, like a window the reference leaves untouched (polisher.cpp:520-547)
            bad = [w for w in range(batch.n_windows) if int(status[w]) > capi.VC_WIN_UNPOLISHED]
            if bad:
                print(f"[vechat_amd] warning: {len(bad)} window(s) left unpolished (first: window {bad[0]}, status {int(status[bad[0]])})", file=sys.stderr)
                status = status.copy()
                for w in bad:
                    cons[w] = batch.window(w)[0][0]
                    status[w] = capi.VC_WIN_UNPOLISHED
            text = b"".join(b">" + name.encode() + b"\n" + data + b"\n"
                            for name, data in wb.stitch(cons, status, drop_unpolished=not a.include_unpolished, fragment_correction=True))
            n_windows, n_polished = batch.n_windows, sum(int(s) == capi.VC_WIN_OK for s in status)
            wb.close()
    elif not distributed:
        raise ValueError("empty overlap set")
    print(f"[vechat_amd] rank {rank}/{world}: {len(targets)} targets, {kept} overlaps ({n_aligned} aligned on the device), {n_windows} windows, "
          f"{n_polished} polished", file=sys.stderr)
    if distributed:
        payload = torch.from_numpy(np.frombuffer(text + b"\0", dtype=np.uint8).copy()[:-1]).to(dev)
        call, lall = gather_consensus(payload, torch.tensor([len(text)], dtype=torch.int64, device=dev), dst=0, force=True)
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return 0
        text = call.cpu().numpy().tobytes()
    sys.stdout.write(text.decode())
    sys.stdout.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
