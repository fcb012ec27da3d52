#!/usr/bin/env python3
"""Headline benchmark: POA windows/s of the per-window SPOA + prune + consensus hot path.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched through torch.distributed.run)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): synthetic windows,
500 bp backbone x 64 reads, PacBio profile (15 % error, ins:del:sub 0.40:0.30:0.30), FASTQ weights,
haplotype mode -d 0.2 -s 0.2 -k 3, scores 3/-5/-4.  One step = one pass of the hot path over one
batch of `--windows` windows per GPU, inputs already resident in HBM, followed by the gather of the
corrected sequences to rank 0 (RCCL over xGMI when N>1).  Weak scaling: every rank gets its own
`--windows` windows of the stream.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from vechat_amd import capi
from vechat_amd.engine import HipContext
from vechat_amd.shard import gather_consensus

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_CELL = 4.0           # one int16 score store + one load by a successor row


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(batch, params, budget_s):
    """CHECKER/BASELINE leg (rank 0, N=1): the reference itself when oracle/_ref travelled with the
    repo ("reference"), else our C restatement ("port"), on a bounded sample of the same workload,
    one window per task on all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as oa
    cores = usable_cores()
    kind = "reference" if oa.have_ref("sse41") else "port"
    if kind == "reference":
        oa.load_ref("sse41")
        fn = lambda w: oa.ref_window(batch, w, params)[0]
    else:
        oa.load_oracle()
        fn = lambda w: oa.oracle_run(batch, params, w, w + 1)[0][0]
    done, out = 0, {}
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        while time.time() - t0 < budget_s and done < batch.n_windows:
            ws = list(range(done, min(done + cores, batch.n_windows)))
            for w, c in zip(ws, ex.map(fn, ws)):
                out[w] = c
            done += len(ws)
    dt = time.time() - t0
    return dict(value=done / dt, unit="windows/s", cores=cores, kind=kind,
                sample=f"first {done} windows of the bench batch, one window per task on {cores} threads, {dt:.1f} s"), out


def stub_main(a, world):
    """Launch-plumbing check for boxes without a GPU (tests/test_shard.py): the same rank fan-out, barrier, MAX-over-ranks
    timing and gather, over gloo, with the kernels replaced by "consensus = backbone".  Never a measurement: the line says
    metric "stub"."""
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = capi.synth_cfg(1002, a.length, min(a.layers, 4), profile=capi.PACBIO)
    batch = capi.synth_batch(cfg, rank * a.windows, a.windows)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        backs = [batch.window(w)[0][0] for w in range(batch.n_windows)]
        cons = torch.from_numpy(np.frombuffer(b"".join(backs), dtype=np.uint8).copy())
        lens = torch.tensor([len(x) for x in backs], dtype=torch.int64)
        cons_all, lens_all = gather_consensus(cons, lens, dst=0)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": a.windows * world * a.steps / float(t.item()), "unit": "windows/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "data": "stub", "windows_gathered": int(lens_all.numel())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=32768, help="windows per GPU per step")
    ap.add_argument("--layers", type=int, default=64)
    ap.add_argument("--length", type=int, default=500)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # one process per GPU: start N ranks of this same command line (rank r on device r), as the reference fans one
        # invocation out over its devices (src/cuda/cudapolisher.cpp:229-241)
        import subprocess
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if os.environ.get("VC_BENCH_STUB") == "1":
        return stub_main(a, world)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("VC_FORCE_DIST") == "1"      # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = capi.synth_cfg(1002, a.length, a.layers, profile=capi.PACBIO)
    batch = capi.synth_batch(cfg, rank * a.windows, a.windows)
    ctx = HipContext(device=local, profile=1, chunk_windows=a.chunk, n_streams=a.streams)
    ctx.submit(batch)                                   # H2D: inputs resident before the timed region

    n = batch.n_windows
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    d_status = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_cons = torch.zeros(n * (a.length + 256), dtype=torch.uint8, device=dev)

    def step():
        ctx.run()
        ctx.sync()
        rc = ctx.lib.vc_collect_device(ctx.h, d_cons.data_ptr(), d_cons.numel(), d_off.data_ptr(), d_status.data_ptr())
        if rc != 0:
            raise RuntimeError(ctx.lib.vc_last_error(ctx.h).decode())
        lens = d_off[1:] - d_off[:-1]
        total = int(d_off[-1].item())
        return gather_consensus(d_cons[:total], lens, dst=0, force=force_dist)

    def fence():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kms, cells = {}, 0
    for _ in range(a.steps):
        cons_all, lens_all = step()
        s = ctx.stats()
        cells += s["cells"]
        for k, v in s["kernels"].items():
            kms[k] = kms.get(k, 0.0) + v["ms"]
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    out_line = None
    if rank == 0:
        total_windows = a.windows * world * a.steps
        bases = int(lens_all.sum().item()) * a.steps
        s = ctx.stats()
        status = d_status.cpu().numpy()
        fwd_ms = kms.get("k_fwd", 0.0)
        fwd_launches = s["kernels"]["k_fwd"]["launches"] * a.steps
        achieved = BYTES_PER_CELL * cells / (fwd_ms * 1e-3) / 1e9 if fwd_ms > 0 else 0.0
        # measured HBM bytes per cell of k_fwd (rocprofv3 PMC passes, see profiles/r1_hbm_traffic.json)
        traffic = None
        valu = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")))
            traffic = tj["bytes_per_cell"] * cells / max(fwd_launches, 1)
            # the kernel moves less than the 4 B/cell model, so the VALU issue bound is stated beside it:
            # VALU instructions per DP row (PMC) x 4 cycles per wave64 instruction, against SIMD-cycles of the k_fwd launches
            prop = torch.cuda.get_device_properties(local)
            simds = prop.multi_processor_count * 4
            mhz = getattr(prop, "clock_rate", 0) / 1e3 or 2400.0          # MI355X peak engine clock (MI355X_MICROARCH.md)
            ipr = tj["instructions_per_dp_row"]["VALU"]
            rows = s["dp_rows"] * a.steps
            valu = {"valu_insts_per_dp_row": ipr, "dp_rows_per_step": rows / a.steps, "simds": simds, "clock_mhz": mhz,
                    "frac_of_valu_issue_peak": (rows * ipr * 4.0) / (fwd_ms * 1e-3 * simds * mhz * 1e6) if fwd_ms > 0 else None}
        except Exception as e:
            valu = valu or {"error": repr(e)}
        line = {
            "metric": "POA windows/sec (500 bp x 64-read)", "value": total_windows / dt, "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16",
            "data": "synthetic",
            "config": {"workload": f"synthetic windows {a.length} bp x {a.layers} reads, PacBio 15% error, "
                                   f"FASTQ weights, haplotype mode d=0.2 s=0.2 k=3; {a.windows} windows per GPU per step "
                                   f"(stream of BASELINE config C)",
                       "windows_per_gpu_per_step": a.windows, "backbone_len": a.length, "reads_per_window": a.layers,
                       "chunk_windows": s["chunk_windows"], "streams": s["n_streams"], "max_nodes": s["max_nodes"], "max_edges": s["max_edges"]},
            "corrected_bases_per_s": bases * world / dt if world == 1 else bases / dt,
            "gcups": cells * world / dt / 1e9,
            "windows_not_ok": int((status > 1).sum()),
            "roofline": {"bound": "hbm", "kernel": "k_fwd", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": BYTES_PER_CELL * cells / max(fwd_launches, 1),
                         "algorithmic_bytes_per_cell": BYTES_PER_CELL, "cells_per_step": cells / a.steps,
                         "avg_launch_ms": fwd_ms / max(fwd_launches, 1), "launches_per_step": fwd_launches / a.steps,
                         "valu_issue": valu},
            "kernel_ms_per_step": {k: v / a.steps for k, v in kms.items()},
        }
        if world == 1 and not a.no_cpu:
            cb, ref_out = cpu_baseline(batch, ctx.params, a.cpu_seconds)
            cons_np = cons_all.cpu().numpy()
            off = np.concatenate([[0], np.cumsum(lens_all.cpu().numpy())])
            bad = sum(1 for w, c in ref_out.items() if cons_np[off[w]:off[w + 1]].tobytes() != c)
            cb["parity_windows_checked"] = len(ref_out)
            cb["parity_mismatches"] = bad
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_baseline"] = line["value"] / cb["value"] if cb["value"] else None
        out_line = json.dumps(line)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out_line is not None:
        sys.stderr.flush()
        print(out_line, flush=True)          # the one JSON line, last thing on stdout


if __name__ == "__main__":
    main()
