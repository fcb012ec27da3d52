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
from vechat_amd.seqio import (load_polisher_input, load_polisher_input_native, read_inputs_native,  # noqa: E402
                                read_overlaps, read_sequences)
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


def main(nt=None, tl=None, depth=None, out=None, python_too=True, quiet=False, copies=1):
    """-> dict(windows_per_s, ...) of the C++-reader path (and the Python-reader path beside it).
    copies > 1: the files hold that many renamed copies of the nt simulated targets and their reads (a large input without simulating
    every read: the work per window is what it is for distinct reads, the simulation is what would take minutes)."""
    nt = nt or (int(sys.argv[1]) if len(sys.argv) > 1 else 200)
    tl = tl or (int(sys.argv[2]) if len(sys.argv) > 2 else 10000)
    depth = depth or (int(sys.argv[3]) if len(sys.argv) > 3 else 64)
    out = out or (sys.argv[4] if len(sys.argv) > 4 else "/tmp/vc_files")
    say = (lambda *a, **k: None) if quiet else print
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(7)
    t0 = time.time()
    fq, sam, tg = os.path.join(out, "reads.fastq"), os.path.join(out, "ovl.sam"), os.path.join(out, "targets.fastq")
    sim = []
    for t in range(nt):
        tseq = NT[rng.integers(0, 4, tl)]
        rs = []
        for d in range(depth):
            r, cg = noisy_copy(rng, tseq)
            rs.append((r.tobytes(), (rng.integers(8, 30, len(r)) + 33).astype(np.uint8).tobytes(), cg))
        sim.append((tseq.tobytes(), rs))
    with open(fq, "wb") as f_r, open(sam, "w") as f_s, open(tg, "wb") as f_t:
        for c in range(copies):
            sfx = f"x{c}" if copies > 1 else ""
            for t, (tseq, rs) in enumerate(sim):
                tname = f"t{t}{sfx}"
                f_t.write(b"@" + tname.encode() + b"\n" + tseq + b"\n+\n" + b"5" * tl + b"\n")
                for d, (r, q, cg) in enumerate(rs):
                    rn = f"r{t}_{d}{sfx}"
                    f_r.write(b"@" + rn.encode() + b"\n" + r + b"\n+\n" + q + b"\n")
                    f_s.write(f"{rn}\t0\t{tname}\t1\t60\t{cg}\t*\t0\t0\t*\t*\n")
    del sim
    mb = (os.path.getsize(fq) + os.path.getsize(sam) + os.path.getsize(tg)) / 1e6
    say(f"generated {nt} targets x {tl} bp x {depth} reads x {copies} copies: {mb:.0f} MB of files in {time.time() - t0:.1f} s", flush=True)

    keep = []

    def run(native, dev):
        T = {}
        if native:
            t0 = time.time(); reads, overlaps, targets = read_inputs_native(fq, sam, tg); T["parse the three files (side by side)"] = time.time() - t0
        else:
            t0 = time.time(); overlaps = read_overlaps(sam); T["parse overlaps (SAM)"] = time.time() - t0
            t0 = time.time(); targets, reads = read_sequences(tg), read_sequences(fq); T["parse sequences (FASTQ)"] = time.time() - t0
        t0 = time.time()
        wb = WindowBuilder(500, 10.0)
        kept, _ = (load_polisher_input_native if native else load_polisher_input)(wb, targets, reads, overlaps, 0.3)
        T["load (records -> window builder)"] = time.time() - t0
        t0 = time.time()
        fill = None
        if native and dev and os.environ.get("VC_FILES_STREAM", "1") != "0":
            batch, ids, fill = wb.build_streaming()           # laid out now; written slice by slice while the device runs (the command line's way)
        else:
            batch, ids = wb.build(copy=not native)
        T["window assembly (vc_wb_build)" if fill is None else "window layout (vc_wb_build_begin; the bytes are written beside the device, vc_wb_build_fill)"] = time.time() - t0
        nw = batch.n_windows
        text = b""
        if dev:
            t0 = time.time(); cons, status = ctx.consensus_batched(batch, fill=fill); T["device (submit + run + collect, slices queued behind each other)"] = time.time() - t0
            t0 = time.time()
            text = b"".join(b">" + n.encode() + b"\n" + d + b"\n" for n, d in wb.stitch(cons, status, drop_unpolished=True, fragment_correction=True))
            T["stitch"] = time.time() - t0
            say(f"{nw} windows, {kept} overlaps, {len(text) / 1e6:.1f} MB of corrected FASTA; polished {sum(int(s) == capi.VC_WIN_OK for s in status)}")
        keep.append((wb, targets, reads, overlaps))               # (the native batch is a view of the builder's buffers)
        tot = sum(T.values())
        say(("C++ readers (vc_io)" if native else "Python readers (seqio)") + (":" if dev else ", host side only:"))
        for k, v in T.items():
            say(f"  {k:34s} {v:8.2f} s  {nw / v:10.0f} windows/s")
        say(f"  {'files -> FASTA' if dev else 'files -> batch':34s} {tot:8.2f} s  {nw / tot:10.0f} windows/s   ({mb / tot:.0f} MB/s of input)", flush=True)
        return text, batch, nw / tot, T

    dev = os.environ.get("VC_FILES_HOST_ONLY") != "1"
    if dev:
        # context and workspaces exist before the clock starts, as they do in `python -m vechat_amd.polish`, which starts the device
        # and reserves its workspaces (vc_reserve) on a thread while the files are parsed
        ctx = HipContext(device=0, mode=0, min_confidence=0.2, min_support=0.2, num_prune=3, reserve=0)
        ctx.consensus(capi.synth_batch(capi.synth_cfg(1, 200, 8), 0, 64))
    t_nat, b_nat, rate_nat, T_nat = run(True, dev)
    res = {"windows_per_s": rate_nat, "windows": int(b_nat.n_windows), "input_mb": mb, "seconds_by_phase": {k: round(v, 4) for k, v in T_nat.items()},
           "workload": f"{nt * copies} targets x {tl} bp x {depth} reads{' (' + str(copies) + ' renamed copies of ' + str(nt) + ' simulated ones)' if copies > 1 else ''} as FASTQ + SAM files -> corrected FASTA text (C++ readers vc_io_*, window builder, device, stitching; "
                       "one process, phases in series)", "batch": b_nat, "text": t_nat}
    if python_too:
        t_py, b_py, rate_py, _ = run(False, dev)
        same = all(np.array_equal(getattr(b_nat, k), getattr(b_py, k)) for k in ("win_seq_off", "seq_off", "seq_begin", "seq_end", "seq_has_qual", "bases", "quals", "win_fasta"))
        say(f"batches identical: {same}; corrected FASTA identical: {t_nat == t_py}; files_to_fasta {rate_nat:.0f} windows/s (C++) vs {rate_py:.0f} (Python)")
        res.update({"windows_per_s_python_readers": rate_py, "identical_to_python_path": bool(same and t_nat == t_py)})
    if os.environ.get("VC_FILES_JSON"):
        import json
        json.dump({k: v for k, v in res.items() if k not in ("batch", "text")}, open(os.environ["VC_FILES_JSON"], "w"))
    if dev:
        ctx.close()
    res["_keep"] = keep
    return res


if __name__ == "__main__":
    main()
