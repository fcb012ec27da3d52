"""Command line over the whole path on one MI355X (fragment correction, the way the VeChat driver calls
vechat_racon: `-f -p -d 0.2 -s 0.2` for round 1, `-f` for round 2; scripts/vechat:70-72,91-93):

  python -m vechat_amd.polish reads.fastq overlaps.sam targets.fastq > corrected.fasta

Overlaps may be SAM, PAF with cg:Z:, or plain PAF (then they are aligned on the device first).  There is no CPU path: without the HIP library and a GPU
this exits with an error."""
import argparse
import os
import sys

from . import capi
from .engine import HipContext
from .seqio import align_missing, load_polisher_input, read_overlaps, read_sequences
from .windows import WindowBuilder


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m vechat_amd.polish", description=__doc__)
    ap.add_argument("sequences"); ap.add_argument("overlaps"); ap.add_argument("targets")
    ap.add_argument("-p", "--haplotype", action="store_true", help="haplotype-aware (variation graph) correction")
    ap.add_argument("-d", "--min-confidence", type=float, default=0.2)
    ap.add_argument("-s", "--min-support", type=float, default=0.2)
    ap.add_argument("-k", "--num-prune", type=int, default=3)
    ap.add_argument("-w", "--window-length", type=int, default=500)
    ap.add_argument("-q", "--quality-threshold", type=float, default=10.0)
    ap.add_argument("-e", "--error-threshold", type=float, default=0.3)
    ap.add_argument("-m", "--match", type=int, default=3)
    ap.add_argument("-x", "--mismatch", type=int, default=-5)
    ap.add_argument("-g", "--gap", type=int, default=-4)
    ap.add_argument("-u", "--include-unpolished", action="store_true")
    ap.add_argument("--no-trimming", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)

    wb = WindowBuilder(a.window_length, a.quality_threshold)
    # one process per GPU when launched through torch.distributed.run: every rank reads the inputs, aligns its share of
    # the CIGAR-less overlaps, takes a contiguous cost-balanced range of windows; rank 0 gathers the results
    # over RCCL and writes the output
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    distributed = world > 1 or os.environ.get("VC_FORCE_DIST") == "1"
    device = int(os.environ.get("LOCAL_RANK", a.device)) if distributed else a.device
    shard = None
    if distributed:
        import numpy as np
        import torch
        import torch.distributed as dist
        from .shard import estimated_cells, gather_consensus, shard_range_balanced
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dev = torch.device("cuda", device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

        def exchange(items):              # list of bytes per rank -> everybody's, in rank order
            payload = torch.from_numpy(np.frombuffer(b"".join(items) + b"\0", dtype=np.uint8).copy()[:-1]).to(dev)
            lens = torch.tensor([len(x) for x in items], dtype=torch.int64, device=dev)
            call, lall = gather_consensus(payload, lens, dst=None, force=True)
            blob, lall = call.cpu().numpy().tobytes(), lall.cpu().numpy()
            off = np.concatenate([[0], np.cumsum(lall)])
            return [blob[int(off[k]):int(off[k + 1])] for k in range(len(lall))]
        shard = (rank, world, exchange)

    targets, reads, overlaps = read_sequences(a.targets), read_sequences(a.sequences), read_overlaps(a.overlaps)
    n_aligned = align_missing(targets, reads, overlaps, a.error_threshold, device, shard)     # PAF / MHAP without a CIGAR (overlap.cpp:205-220)
    kept, window_type = load_polisher_input(wb, targets, reads, overlaps, a.error_threshold)
    batch, ids = wb.build()
    ctx = HipContext(device=device, mode=0 if a.haplotype else 1, min_confidence=a.min_confidence, min_support=a.min_support,
                     num_prune=a.num_prune, match=a.match, mismatch=a.mismatch, gap=a.gap, trim=0 if a.no_trimming else 1,
                     window_type=window_type)
    if not distributed:
        cons, status = ctx.consensus(batch)
    else:
        lo, hi = shard_range_balanced(estimated_cells(batch), rank, world)
        lc, ls = ctx.consensus(batch.slice(lo, hi)) if hi > lo else ([], np.zeros(0, np.uint8))
        payload = torch.from_numpy(np.frombuffer(b"".join(lc) + b"\0", dtype=np.uint8).copy()[:-1]).to(dev)
        # length and status of a window travel together: status in the bits above 40
        lens = torch.tensor([len(x) | (int(s) << 40) for x, s in zip(lc, ls)], dtype=torch.int64, device=dev)
        call, lall = gather_consensus(payload, lens, dst=0, force=True)
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return 0
        lall = lall.cpu().numpy()
        blob = call.cpu().numpy().tobytes()
        off = np.concatenate([[0], np.cumsum(lall & ((1 << 40) - 1))])
        cons = [blob[int(off[w]):int(off[w + 1])] for w in range(batch.n_windows)]
        status = (lall >> 40).astype(np.uint8)
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
    for name, data in wb.stitch(cons, status, drop_unpolished=not a.include_unpolished, fragment_correction=True):
        sys.stdout.write(f">{name}\n{data.decode()}\n")
    print(f"[vechat_amd] {kept} overlaps ({n_aligned} aligned on the device), {batch.n_windows} windows, {sum(int(s) == capi.VC_WIN_OK for s in status)} polished",
          file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
