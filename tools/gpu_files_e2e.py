"""Files -> corrected FASTA, timed phase by phase: does the host side keep up with the device?

Writes a synthetic read set (targets of T kb, D noisy copies each, PacBio-like 15 % error) as FASTQ + SAM with the CIGARs of the
simulated edits, then runs the steps of `python -m vechat_amd.polish -f -p` one by one with a clock around each:
parse (FASTQ, SAM) / window assembly (vc_wb_*: breaking points, layers) / device (submit + run + collect) / stitch.

usage: gpu_files_e2e.py [targets=200] [target_len=10000] [depth=64] [outdir=/tmp/vc_files]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vechat_amd import capi                                                     # noqa: E402
from vechat_amd.engine import HipContext                                        # noqa: E402
from vechat_amd.seqio import load_polisher_input, read_overlaps, read_sequences  # noqa: E402
from vechat_amd.windows import WindowBuilder                                    # noqa: E402

NT = np.frombuffer(b"ACGT", dtype=np.uint8)


def noisy_copy(rng, t, sub=0.05, ins=0.05, dele=0.05):
    """One read of target t (uint8 bases) and the CIGAR of its global alignment to t."""
    L = len(t)
    u = rng.random(L)
    is_del = u < dele
    is_ins = (u >= dele) & (u < dele + ins)
    is_sub = (u >= dele + ins) & (u < dele + ins + sub)
    base = t.copy()
    base[is_sub] = NT[(np.searchsorted(NT, t[is_sub]) + rng.integers(1, 4, int(is_sub.sum()))) % 4]
    # per target position: 0 bases (deletion), 2 bases (an inserted base, then the position's own), else 1
    cnt = np.where(is_del, 0, np.where(is_ins, 2, 1))
    out = np.repeat(base, cnt)
    starts = np.cumsum(cnt) - cnt
    out[starts[is_ins]] = NT[rng.integers(0, 4, int(is_ins.sum()))]
    # op string per target position: D | IM | M, run-length encoded
    ops = np.repeat(np.where(is_del, ord("D"), ord("M")).astype(np.uint8), np.where(is_ins, 2, 1))
    pos = np.cumsum(np.where(is_ins, 2, 1)) - np.where(is_ins, 2, 1)
    ops[pos[is_ins]] = ord("I")
    edge = np.flatnonzero(np.diff(ops)) + 1
    b = np.concatenate(([0], edge)); e = np.concatenate((edge, [len(ops)]))
    cigar = "".join(f"{int(n)}{chr(int(o))}" for n, o in zip(e - b, ops[b]))
    return out, cigar


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    tl = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    out = sys.argv[4] if len(sys.argv) > 4 else "/tmp/vc_files"
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(7)
    t0 = time.time()
    fq, sam, tg = os.path.join(out, "reads.fastq"), os.path.join(out, "ovl.sam"), os.path.join(out, "targets.fastq")
    with open(fq, "wb") as f_r, open(sam, "w") as f_s, open(tg, "wb") as f_t:
        for t in range(nt):
            tseq = NT[rng.integers(0, 4, tl)]
            tname = f"t{t}"
            f_t.write(b"@" + tname.encode() + b"\n" + tseq.tobytes() + b"\n+\n" + b"5" * tl + b"\n")
            for d in range(depth):
                r, cg = noisy_copy(rng, tseq)
                q = (rng.integers(8, 30, len(r)) + 33).astype(np.uint8).tobytes()
                rn = f"r{t}_{d}"
                f_r.write(b"@" + rn.encode() + b"\n" + r.tobytes() + b"\n+\n" + q + b"\n")
                f_s.write(f"{rn}\t0\t{tname}\t1\t60\t{cg}\t*\t0\t0\t*\t*\n")
    mb = (os.path.getsize(fq) + os.path.getsize(sam) + os.path.getsize(tg)) / 1e6
    print(f"generated {nt} targets x {tl} bp x {depth} reads: {mb:.0f} MB of files in {time.time() - t0:.1f} s", flush=True)

    T = {}
    t0 = time.time(); overlaps = read_overlaps(sam); T["parse overlaps (SAM)"] = time.time() - t0
    t0 = time.time(); targets, reads = read_sequences(tg), read_sequences(fq); T["parse sequences (FASTQ)"] = time.time() - t0
    t0 = time.time()
    wb = WindowBuilder(500, 10.0)
    kept, _ = load_polisher_input(wb, targets, reads, overlaps, 0.3)
    batch, ids = wb.build()
    T["window assembly"] = time.time() - t0
    ctx = HipContext(device=0, mode=0, min_confidence=0.2, min_support=0.2, num_prune=3)
    ctx.consensus(capi.synth_batch(capi.synth_cfg(1, 200, 8), 0, 64))           # context and workspaces exist before the clock starts
    t0 = time.time(); cons, status = ctx.consensus(batch); T["device (submit + run + collect)"] = time.time() - t0
    t0 = time.time()
    text = b"".join(b">" + n.encode() + b"\n" + d + b"\n" for n, d in wb.stitch(cons, status, drop_unpolished=True, fragment_correction=True))
    T["stitch"] = time.time() - t0
    nw = batch.n_windows
    tot = sum(T.values())
    print(f"{nw} windows, {kept} overlaps, {len(text) / 1e6:.1f} MB of corrected FASTA; polished {sum(int(s) == capi.VC_WIN_OK for s in status)}")
    for k, v in T.items():
        print(f"  {k:34s} {v:8.2f} s  {nw / v:10.0f} windows/s")
    print(f"  {'files -> FASTA':34s} {tot:8.2f} s  {nw / tot:10.0f} windows/s   ({mb / tot:.0f} MB/s of input)")


if __name__ == "__main__":
    main()
