#!/usr/bin/env python3
"""Headline benchmark: POA windows/s of the per-window SPOA + prune + consensus hot path.

  python bench.py --gpus N --steps K --warmup W      (N > 1: this command starts N ranks itself, one per GPU;
                                                       under torch.distributed.run it is one of the ranks)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): synthetic windows, 500 bp backbone x
64 reads, PacBio profile (15 % error, ins:del:sub 0.40:0.30:0.30), FASTQ weights, haplotype mode -d 0.2 -s 0.2 -k 3,
scores 3/-5/-4.  One step = one pass of the hot path over `--windows` DISTINCT windows per GPU (default 100 000 =
config C), inputs already resident in HBM, followed by the gather of the corrected sequences to rank 0 (RCCL over
xGMI when N > 1).  Weak scaling: every rank gets its own `--windows` windows of the stream (N = 8: 800 000 windows
per step, config D's shape).

The one JSON line also carries
  value_e2e   the same windows from host memory to host memory (vc_submit -> vc_run -> vc_collect, H2D and D2H
              included), batches of 32 768 windows queued behind each other in one context -- SURVEY 8(d)'s definition
              of the metric; `value` is the resident-input rate the driver's contract asks for
  roofline    the bound that holds: VALU issue.  VALU wave-instructions of a step (rocprofv3 PMC counts per window, profiles/
              r4_hbm_traffic.json, refused when taken for other kernel sources than the ones built here) / step wall time / SIMDs,
              against the issue rate of packed-int16 max / add measured on this device in this run (lib/valu_peak.bin);
              roofline.k_fwd: the forward DP alone; roofline.hbm: SURVEY 8(d)'s 4 B/cell model and the measured HBM bytes
  cpu_baseline  the reference itself (oracle/_ref, built in place from /root/reference) on the host cores, bounded
  configs     windows/s on BASELINE configs B and E, short runs
"""
import argparse
import hashlib
import json
import os
import sys
import threading
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
E2E_BATCH = 32768            # windows per host batch of the host-to-host pass (16 384: 34.5 k, 25 088: 34.6 k, 32 768: 35.4 k windows/s on one box, profiles/r5_e2e_timeline.txt)
E2E_FIRST = 8192             # the first batch of the host-to-host pass is smaller: the device starts after 0.5 GB of H2D instead of 1.1 GB


usable_cores = capi.usable_cores      # affinity mask, cgroup quota, shared out over the ranks of this node


def kernel_hash():
    """Identity of the kernels this bench runs: measured-traffic files are only valid for the sources they were taken on.
    Comments and whitespace do not count (a reworded comment does not invalidate a measurement)."""
    import re
    h = hashlib.sha256()
    for f in ("vc_kernels.h", "vc_fwd_dt.h", "vc_api.hip", "vc_device.h", "vc_pipe.h"):
        src = open(os.path.join(ROOT, "vechat_amd", "csrc", f), "r").read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        src = re.sub(r"//[^\n]*", " ", src)
        h.update(" ".join(src.split()).encode())
    return h.hexdigest()[:16]


def valu_peak_now(device):
    """The VALU issue rates the roofline is priced against, measured on this device in this run: vechat_amd/lib/valu_peak.bin
    (built by __graft_entry__.build() from tools/valu_peak.hip) in its quick mode, ~1 s -- independent v_pk_max_i16 / v_pk_add_i16
    chains at 4 and 8 waves per SIMD (the rate rounds 2-4 priced against), then every opcode class of k_fwd's row loop on its own at
    5 and 8 waves per SIMD, each with the shader clock it ran at (s_memtime over s_memrealtime) and the SIMD cycles per instruction.
    -> dict(pk16=rate or None, classes={name: best test record}, source=...)"""
    import subprocess
    exe = os.path.join(ROOT, "vechat_amd", "lib", "valu_peak.bin")
    res = {"pk16": None, "classes": {}, "source": None}
    try:
        env = dict(os.environ)
        vis = [x for x in env.get("HIP_VISIBLE_DEVICES", "").split(",") if x]
        env["HIP_VISIBLE_DEVICES"] = vis[device] if device < len(vis) else str(device)      # the calibration binary uses device 0 of what it sees
        out = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=180, env=env).stdout
        tests = [json.loads(l) for l in out.splitlines() if l.startswith('{"test"')]
        pk = [t for t in tests if t["test"] == "pk_i16_independent"]
        if pk:
            best = max(pk, key=lambda t: t["inst_per_us_per_simd"])
            res["pk16"] = best["inst_per_us_per_simd"]
            res["pk16_sclk_mhz"] = best.get("sclk_mhz"); res["pk16_simd_cycles_per_inst"] = best.get("simd_cycles_per_inst")
        mixed = [t for t in tests if t["test"] == "pk16_and_salu_interleaved"]
        if mixed:
            bm = max(mixed, key=lambda t: t["inst_per_us_per_simd"])
            res["mixed_issue"] = {x: bm[x] for x in ("inst_per_us_per_simd", "waves_per_simd", "sclk_mhz", "simd_cycles_per_inst") if x in bm}
        salu = [t for t in tests if t["test"] == "salu_independent"]
        if salu:
            res["salu_only"] = max(t["inst_per_us_per_simd"] for t in salu)
        for t in tests:
            if t["test"].startswith("class_"):
                k = t["test"][6:]
                if k not in res["classes"] or t["inst_per_us_per_simd"] > res["classes"][k]["inst_per_us_per_simd"]:
                    res["classes"][k] = {x: t[x] for x in ("inst_per_us_per_simd", "waves_per_simd", "sclk_mhz", "simd_cycles_per_inst") if x in t}
        res["source"] = ("this run (vechat_amd/lib/valu_peak.bin quick, after the timed region on the same device: independent chains of each opcode class, "
                         "best of 5 / 8 waves per SIMD; pk16: v_pk_max_i16 / v_pk_add_i16 interleaved, best of three passes at 4 / 8 waves)")
        if not tests:
            res["source"] = "calibration printed nothing"
    except Exception as e:
        res["source"] = f"calibration failed: {e!r}"
    return res


def mix_peak(classes, khash):
    """The issue rate of k_fwd's own instruction mix: sum(n_i) / sum(n_i / rate_i) over the opcode classes of its row loop
    (profiles/r6_valu_mix.json: histogram of the commonest row from the device ISA, tools/valu_mix.py), every class's rate measured
    in this run.  -> (rate or None, detail)"""
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "r6_valu_mix.json")))
    except Exception as e:
        return None, {"note": repr(e)}
    if mix.get("kernel_hash") != khash:
        return None, {"note": f"profiles/r6_valu_mix.json was made for kernels {mix.get('kernel_hash')}, these are {khash}: not used"}
    n = mix["representative_row"]
    missing = [k for k in n if k not in classes]
    if missing:
        return None, {"note": f"no calibration for classes {missing}"}
    tot = sum(n.values())
    t = sum(v / classes[k]["inst_per_us_per_simd"] for k, v in n.items())
    return tot / t, {"instructions_per_row_by_class": n, "rate_by_class": {k: classes[k]["inst_per_us_per_simd"] for k in n},
                     "simd_cycles_per_inst_by_class": {k: classes[k].get("simd_cycles_per_inst") for k in n},
                     "sclk_mhz_by_class": {k: classes[k].get("sclk_mhz") for k in n},
                     "histogram_source": mix["source"] + "; the cheapest way round the row loop = the row whose only predecessor is the row above"}


def cpu_baseline(batch, params, budget_s):
    """CHECKER/BASELINE leg (rank 0, N=1): the reference itself when oracle/_ref travelled with the repo ("reference"),
    else our C restatement ("port"), on a bounded sample of the same workload, one window per task on all host cores."""
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


def e2e_rate(batch, device, reps=1):
    """Host memory in, host memory out -- SURVEY 8(d)'s definition of the metric -- through ONE context on ONE host thread, the
    loop include/vechat_hip.h describes: submit(b[i+1]) copies the next batch in while b[i] runs, collect() hands out b[i-1]; the
    library's stream workers go from the last chunk of one batch straight to the first of the next.
    Returns (windows/s, consensus bytes by window)."""
    n = batch.n_windows
    # batch sizes are the caller's choice: a small first batch (the device starts after 0.5 GB of H2D, not 1.1 GB), the odd remainder
    # next -- it runs beside full batches -- and full batches to the end, so that the last windows drain on all chunk streams
    sizes = [min(E2E_FIRST, n)]
    rem = (n - sizes[0]) % E2E_BATCH
    if rem:
        sizes.append(rem)
    sizes += [E2E_BATCH] * ((n - sum(sizes)) // E2E_BATCH)
    cuts = [0]
    for k in sizes:
        cuts.append(cuts[-1] + k)
    assert cuts[-1] == n
    parts = [batch.slice(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]
    ctx = HipContext(device=device)
    out = [None] * len(parts)

    def once():
        t0 = time.perf_counter()
        ctx.submit(parts[0]); ctx.run()
        for i in range(1, len(parts)):
            ctx.submit(parts[i]); ctx.run()
            out[i - 1] = ctx.collect()
        out[-1] = ctx.collect()
        return time.perf_counter() - t0

    # first use allocates the workspaces and both batch slots (seconds): not part of the rate
    ctx.submit(parts[-1]); ctx.run(); ctx.submit(parts[0]); ctx.run(); ctx.collect(); ctx.collect()
    dt = min(once() for _ in range(reps))
    ctx.close()
    cons = [x for p in out for x in p[0]]
    return n / dt, cons


def short_config(device, seed, L, D, n, profile, frac_partial=0.0, n_haplotypes=1, snp_rate=0.01, first=0, check=0, pipeline=None, chunk=0, streams=0):
    """One short resident-input run of another workload shape; `check` > 0: that many of its windows are also run through the
    reference itself (oracle/_ref) on the host cores and compared byte for byte (CHECKER leg, after the timed region)."""
    cfg = capi.synth_cfg(seed, L, D, profile=profile, frac_partial=frac_partial, n_haplotypes=n_haplotypes, snp_rate=snp_rate)
    b = capi.synth_batch(cfg, first, n, n_threads=usable_cores())
    c = HipContext(device=device, pipeline=pipeline, chunk_windows=chunk, n_streams=streams)
    c.submit(b)
    c.run(); c.sync()
    t0 = time.perf_counter()
    c.run(); c.sync()
    dt = time.perf_counter() - t0
    cons, status = c.collect()
    st = c.stats()
    params = c.params
    c.close()
    out = {"windows_per_s": n / dt, "windows": n, "backbone_len": L, "reads_per_window": D, "gcups": st["cells"] / dt / 1e9,
           "windows_not_ok": int((status > 1).sum()), "band_redo": st.get("band_redo", 0)}
    if pipeline:
        out["plan"] = ("persistent build pipeline (vc_set_pipeline 1): the build loop of a chunk as two resident kernels and device-side queues, "
                       "the re-alignment rounds lock-step on the same stream; the headline `value` is the lock-step plan")
        out["launches"] = int(sum(k["launches"] for k in st["kernels"].values()))
        out["chunk_windows"] = st["chunk_windows"]; out["streams"] = st["n_streams"]
    if frac_partial:
        out["frac_partial_layers"] = frac_partial
    if n_haplotypes > 1:
        out["n_haplotypes"] = n_haplotypes; out["snp_rate"] = snp_rate
    if check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api as oa
        if oa.have_ref("sse41"):
            oa.load_ref("sse41")
            ws = list(range(0, n, max(1, n // check)))[:check]
            with ThreadPoolExecutor(usable_cores()) as ex:
                ref = list(ex.map(lambda w: oa.ref_window(b, w, params)[0], ws))
            out["parity_windows_checked"] = len(ws)
            out["parity_mismatches"] = sum(1 for w, r in zip(ws, ref) if cons[w] != r)
            out["parity_against"] = "oracle/_ref (the reference built in place), byte comparison of the consensus"
        else:
            out["parity_windows_checked"] = 0
    return out


def files_to_fasta_large():
    """The same path on a LARGE input (51 200 windows: 20 renamed copies of 128 simulated targets of 10 kb x 64 reads, 4 GB of files): the
    device phase runs the slices of the batch queued behind each other in one context (HipContext.consensus_batched), and the host
    phases in front of it -- parse, load, window assembly -- are what a command line adds to the device's own rate."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("gpu_files_e2e", os.path.join(ROOT, "tools", "gpu_files_e2e.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory(prefix="vc_files_large_", dir="/tmp") as d:
        res = mod.main(nt=128, tl=10000, depth=64, out=d, python_too=True, quiet=True, copies=20)
    for k in ("batch", "text", "_keep"):
        res.pop(k, None)
    return res


def files_to_fasta(params):
    """The path a user runs, from files: FASTQ + SAM on disk -> corrected FASTA text through the C++ readers (vc_io_*), the window
    builder, the device and the stitcher (tools/gpu_files_e2e.py), with the Python readers beside it (must give the same text) and
    the reference's CPU rate on a sample of the very windows those files make (CHECKER leg: byte comparison)."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("gpu_files_e2e", os.path.join(ROOT, "tools", "gpu_files_e2e.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory(prefix="vc_files_", dir="/tmp") as d:
        res = mod.main(nt=96, tl=10000, depth=64, out=d, python_too=True, quiet=True)
    batch, text = res.pop("batch"), res.pop("text")
    keep = res.pop("_keep", None)                           # the batch is a view of the window builder's buffers: the builder stays until the end
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api as oa
        if oa.have_ref("sse41"):
            oa.load_ref("sse41")
            cores = usable_cores()
            p = capi.default_params(min_confidence=0.2, min_support=0.2, num_prune=3)
            ws = list(range(0, batch.n_windows, max(1, batch.n_windows // 96)))[:96]
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                ref = list(ex.map(lambda w: oa.ref_window(batch, w, p)[0], ws))
            dt = time.perf_counter() - t0
            res["reference_cpu"] = {"windows_per_s": len(ws) / dt, "cores": cores, "sample": f"{len(ws)} of the {batch.n_windows} windows these files make (window.cpp through oracle/_ref; parsing not included)"}
    except Exception as e:                                  # the checker leg must not take the bench line down
        res["reference_cpu"] = {"error": repr(e)}
    del batch, keep
    return res


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
    ap.add_argument("--config", default=None, choices=["C", "D", "E", "W"],
                    help="BASELINE.json configuration: C = 100 000 windows of 500 bp x 64 reads on one GPU (default at --gpus 1..7); D = 1 M such windows "
                         "over 8 GPUs (125 000 per rank; default at --gpus 8); E = 50 000 ONT windows of 1 kb x 128 reads over the ranks; "
                         "W = 3 kb x 12 reads (-w 3000 windows: the widest classes of the packed kernel)")
    ap.add_argument("--windows", type=int, default=0, help="distinct windows per GPU per step (0: what --config says)")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--length", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip value_e2e, the per-kernel profile pass and the config B / E lines")
    ap.add_argument("--ab", action="store_true", help="development: only the per-kernel profile pass of the extras")
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
    # the BASELINE.json configuration this run measures (configs[2] / [3] / [4]); explicit --windows / --layers / --length override it
    cfg_name = a.config or ("D" if world == 8 else "C")
    profile = capi.PACBIO
    if cfg_name == "E":
        dflt = (max(1, 50000 // world), 128, 1000); profile = capi.ONT
    elif cfg_name == "W":
        dflt = (2048, 12, 3000)
    elif cfg_name == "D":
        dflt = (1000000 // world, 64, 500)
    else:
        dflt = (100000, 64, 500)
    a.windows, a.layers, a.length = a.windows or dflt[0], a.layers or dflt[1], a.length or dflt[2]
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

    cfg = capi.synth_cfg(1002, a.length, a.layers, profile=profile)
    batch = capi.synth_batch(cfg, rank * a.windows, a.windows)
    ctx = HipContext(device=local, profile=2, chunk_windows=a.chunk, n_streams=a.streams)   # profile 2: HIP events around k_fwd only
    ctx.submit(batch)                                   # H2D: inputs resident before the timed region

    n = batch.n_windows
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    d_status = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_cons = torch.zeros(n * (a.length + 256), dtype=torch.uint8, device=dev)

    times = {"compute_s": 0.0, "gather_s": 0.0}

    def step():
        t_a = time.perf_counter()
        ctx.run()
        ctx.sync()
        rc = ctx.lib.vc_collect_device(ctx.h, d_cons.data_ptr(), d_cons.numel(), d_off.data_ptr(), d_status.data_ptr())
        if rc != 0:
            raise RuntimeError(ctx.lib.vc_last_error(ctx.h).decode())
        lens = d_off[1:] - d_off[:-1]
        total = int(d_off[-1].item())
        t_b = time.perf_counter()
        out = gather_consensus(d_cons[:total], lens, dst=0, force=force_dist)
        if world > 1 or force_dist:
            torch.cuda.synchronize()
        t_c = time.perf_counter()
        times["compute_s"] += t_b - t_a; times["gather_s"] += t_c - t_b
        return out

    def fence():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    times["compute_s"] = times["gather_s"] = 0.0
    fence()
    t0 = time.perf_counter()
    fwd_ms, fwd_busy_ms, fwd_launches, cells, rows = 0.0, 0.0, 0, 0, 0
    for _ in range(a.steps):
        cons_all, lens_all = step()
        s = ctx.stats()
        cells += s["cells"]; rows += s["dp_rows"]
        fwd_ms += s["kernels"]["k_fwd"]["ms"]; fwd_launches += s["kernels"]["k_fwd"]["launches"]
        fwd_busy_ms += s["kernels"]["k_fwd"]["busy_ms"]
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = None
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # every rank's own rate (its windows over its own compute time) and its share of the gather, for the scaling record
        mine = torch.tensor([a.windows * a.steps / max(times["compute_s"], 1e-9), times["gather_s"] / a.steps * 1e3], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).cpu().numpy()
    dt = float(t.item())

    out_line = None
    if rank == 0:
        total_windows = a.windows * world * a.steps
        bases = int(lens_all.sum().item()) * a.steps
        s = ctx.stats()
        status = d_status.cpu().numpy()
        khash = kernel_hash()
        simds = torch.cuda.get_device_properties(local).multi_processor_count * 4
        wall_s = dt / a.steps
        cal = valu_peak_now(local)
        peak_pk16, peak_src = cal["pk16"], cal["source"]
        peak_mix, mix_detail = mix_peak(cal["classes"], khash)
        peak_now = peak_pk16                 # ONE denominator, always -- the packed-int16 issue rate measured in this run (rounds 2-4's); the mix-weighted view is frac_vs_mix
        # The forward DP keeps its predecessor rows in registers / LDS and stores a byte-packed band: it moves ~0.5 B per cell, an
        # eighth of SURVEY 8(d)'s 4 B/cell model, so the bound that holds is the instruction stream ("then the VALU
        # issue bound is the honest limiter and must be stated").  achieved = VALU wave-instructions of ALL kernels of a step
        # (rocprofv3 PMC SQ_INSTS_VALU per window of this workload, measured on these kernel sources) / step wall time / SIMDs;
        # peak = the issue rate of k_fwd's OWN instruction mix (peak_mix: packed int16 max / add, v_perm / v_alignbit, DPP forms, lane
        # moves, 32-bit odds and ends weighted by the histogram of its row loop), every class measured on THIS device in THIS run;
        # peak_pk16 = the packed-int16 class alone (what rounds 2-4 priced against), frac_vs_pk16 beside it.
        roof = {"bound": "valu_issue", "kernel": "all kernels of a step (k_fwd alone: roofline.k_fwd)", "achieved": None, "peak": peak_now,
                "peak_mix": peak_mix, "peak_pk16": peak_pk16, "peak_mix_detail": mix_detail,
                "unit": "VALU wave-instructions / us / SIMD", "frac": None, "traffic": None, "peak_source": peak_src, "simds": simds,
                "kernel_hash": khash,
                # the shader clock: s_memtime ticks (one per shader cycle) over s_memrealtime ticks (100 MHz), wave by wave -- under the job
                # (summed over k_fwd's row loops inside the timed region) and under the calibration loops.  The chip clocks to its power
                # budget (MI355X_MICROARCH.md, DVFS): a dense vector stream runs well below the 2.4 GHz the device reports.
                "sclk_mhz": {"timed_region_k_fwd": s.get("fwd_sclk_mhz"), "calibration_pk16": cal.get("pk16_sclk_mhz"), "device_reports": torch.cuda.get_device_properties(local).clock_rate / 1e3 if hasattr(torch.cuda.get_device_properties(local), "clock_rate") else None},
                "pk16_simd_cycles_per_inst": cal.get("pk16_simd_cycles_per_inst")}
        hbm = {"peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_cell": BYTES_PER_CELL, "cells_per_step": cells / a.steps,
               "algorithmic_gbs_over_wall": BYTES_PER_CELL * cells / a.steps / wall_s / 1e9}
        hbm["algorithmic_model_exceeds_peak"] = hbm["algorithmic_gbs_over_wall"] > HBM_PEAK_GBS
        hbm["algorithmic_frac"] = None if hbm["algorithmic_model_exceeds_peak"] else hbm["algorithmic_gbs_over_wall"] / HBM_PEAK_GBS
        kf = {"avg_launch_ms": fwd_ms / max(fwd_launches, 1), "launches_per_step": fwd_launches / a.steps, "busy_ms_per_step": fwd_busy_ms / a.steps,
              "dp_rows_per_step": rows / a.steps,
              "timing": "HIP events around every k_fwd launch on its own stream, inside the timed region (vc_params.profile = 2); the chunk streams "
                        "overlap, so launches run beside each other: busy_ms = time during which at least one k_fwd launch was running"}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r6_hbm_traffic.json")))
            if tj.get("kernel_hash") != khash:
                roof["counts_note"] = f"profiles/r6_hbm_traffic.json was measured for kernels {tj.get('kernel_hash')}, these are {khash}: not used"
            else:
                roof["counts_source"] = ("rocprofv3 --pmc passes over the same kernel sources (profiles/r6_hbm_traffic.json from profiles/r6_pmc_counters.txt: "
                                         "SQ_INSTS_VALU, FETCH_SIZE, WRITE_SIZE, one counter group per pass) scaled by this run's windows / DP rows / cells")
                wps = a.windows * world                                             # windows per step
                job_valu = tj["valu_insts_per_window_all_kernels"] * wps
                roof["valu_wave_insts_per_step"] = job_valu
                if peak_now:
                    roof["achieved"] = job_valu / (wall_s * 1e6 * simds)
                    roof["frac"] = roof["achieved"] / peak_now
                    roof["frac_vs_pk16"] = roof["frac"]
                    if peak_mix:
                        roof["frac_vs_mix"] = roof["achieved"] / peak_mix
                roof["_valu_per_window_by_config"] = tj.get("valu_insts_per_window_by_config", {})
                # Second view, the one that explains why nothing moves this design by much any more: EVERY instruction the SQ counts (vector,
                # scalar, branch, LDS, memory) of a step over the same time, against what a SIMD issues when vector and scalar instructions
                # come 1 : 1 (independent v_pk_max / v_pk_add_i16 interleaved with independent s_add / s_and / s_max / s_lshl, same run) --
                # roughly k_fwd's own ratio.  A SIMD issues a packed-int16 instruction every ~4.2 cycles, a scalar one every ~4.7, and
                # both side by side at ~2.8 cycles per instruction: the scalar half of a DP row is not free.
                ipw = tj.get("insts_per_window_all_kernels")
                if ipw and cal.get("mixed_issue"):
                    tot = sum(ipw.values())
                    ach = tot * wps / (wall_s * 1e6 * simds)
                    roof["instruction_issue"] = {"insts_per_window_by_class": ipw, "achieved": ach, "unit": "wave-instructions of any class / us / SIMD",
                                                 "peak_1to1_vector_scalar": cal["mixed_issue"]["inst_per_us_per_simd"], "frac": ach / cal["mixed_issue"]["inst_per_us_per_simd"],
                                                 "peak_detail": cal["mixed_issue"], "salu_only_rate": cal.get("salu_only"),
                                                 "note": "not a hard ceiling (the attainable total depends on the mix); reported beside roofline.frac, which stays the vector-issue fraction"}
                ipr = tj["instructions_per_dp_row"]["VALU"]
                kf.update({"valu_insts_per_dp_row": ipr, "valu_wave_insts_per_step": ipr * rows / a.steps})
                if peak_now:
                    kf["frac_of_valu_issue_peak_over_wall"] = ipr * rows / a.steps / (wall_s * 1e6 * simds * peak_now)
                    if fwd_busy_ms > 0:
                        kf["frac_of_valu_issue_peak_over_busy_time"] = ipr * rows / (fwd_busy_ms * 1e3 * simds * peak_now)
                roof["traffic"] = tj["bytes_per_cell"] * cells / max(fwd_launches, 1)       # measured HBM bytes per k_fwd launch
                hbm.update({"measured_bytes_per_cell": tj["bytes_per_cell"], "measured_gbs_over_wall": tj["bytes_per_cell"] * cells / a.steps / wall_s / 1e9,
                            "measured_frac": tj["bytes_per_cell"] * cells / a.steps / wall_s / 1e9 / HBM_PEAK_GBS,
                            "k_tracew_bytes_fetched_per_move": tj.get("k_tracew_bytes_fetched_per_move")})
        except Exception as e:
            roof["counts_note"] = repr(e)
        roof["k_fwd"] = kf
        roof["hbm"] = hbm
        line = {
            "metric": "POA windows/sec (500 bp x 64-read)", "value": total_windows / dt, "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16",
            "data": "synthetic",
            "config": {"workload": f"synthetic windows {a.length} bp x {a.layers} reads, {'ONT' if profile == capi.ONT else 'PacBio 15% error'}, FASTQ weights, "
                                   f"haplotype mode d=0.2 s=0.2 k=3; {a.windows} distinct windows per GPU per step x {world} GPU(s) = "
                                   f"{a.windows * world} windows per step (BASELINE config {cfg_name})",
                       "baseline_config": cfg_name, "windows_per_step": a.windows * world,
                       "windows_per_gpu_per_step": a.windows, "backbone_len": a.length, "reads_per_window": a.layers,
                       "chunk_windows": s["chunk_windows"], "streams": s["n_streams"], "max_nodes": s["max_nodes"], "max_edges": s["max_edges"]},
            "corrected_bases_per_s": bases * world / dt if world == 1 else bases / dt,
            "gcups": cells * world / dt / 1e9,
            "windows_not_ok": int((status > 1).sum()),
            "roofline": roof,
        }
        if world > 1 or force_dist:
            line["world_size"] = dist.get_world_size()           # as RCCL reports it
            line["gather_ms"] = times["gather_s"] / a.steps * 1e3
            if per_rank is not None:
                line["per_rank_windows_per_s"] = {"min": float(per_rank[:, 0].min()), "max": float(per_rank[:, 0].max())}
                line["per_rank_gather_ms"] = {"min": float(per_rank[:, 1].min()), "max": float(per_rank[:, 1].max())}
        if world == 1 and not a.no_extras and cfg_name == "C":
            # separate pass with every kernel class bracketed by events: the breakdown, not part of `value`
            ctx.lib.vc_set_profile(ctx.h, 1)
            ctx.run(); ctx.sync()
            sp = ctx.stats()
            line["kernel_ms_per_step"] = {k: v["ms"] for k, v in sp["kernels"].items()}
            line["kernel_ms_note"] = "separate profiled pass (vc_params.profile = 1); sums exceed ms_per_step because chunk streams overlap"
            ctx.lib.vc_set_profile(ctx.h, 2)
    ctx_params = ctx.params
    if rank == 0 and world == 1 and not a.no_extras and not a.ab and cfg_name == "C":
        cons_np = cons_all.cpu().numpy()
        off = np.concatenate([[0], np.cumsum(lens_all.cpu().numpy())])
        ctx.close()                                     # its workspaces go back before the two e2e contexts plan theirs
        torch.cuda.empty_cache()
        rate, cons_e2e = e2e_rate(batch, local, reps=2)     # (best of two passes: the first can run into the driver still clearing the memory the headline context gave back)
        line["value_e2e"] = rate
        line["value_resident"] = line["value"]          # (`value` IS the resident-input rate the bench contract defines; named once more beside the host-to-host rate)
        line["e2e"] = {"definition": "host arrays -> vc_submit -> vc_run -> vc_collect -> host bytes, H2D and D2H included (SURVEY 8(d)'s metric): one context, "
                                     f"one host thread, batches of {E2E_BATCH} windows (the first: {E2E_FIRST}) queued behind each other (submit of batch i+1 and collect of batch i-1 "
                                     "while batch i runs)",
                       "of_value": rate / line["value"],
                       "identical_to_resident_run": all(cons_np[off[w]:off[w + 1]].tobytes() == cons_e2e[w] for w in range(0, n, 97))}
        line["configs"] = {"B": short_config(local, 1001, 500, 32, 10000, capi.PACBIO, check=256),
                           "E": short_config(local, 1005, 1000, 128, 4096, capi.ONT),
                           "W": short_config(local, 1007, 3000, 12, 1024, capi.PACBIO, check=64),      # 3 kb windows: width classes 48 / 64 of the packed kernel (k_fwd_wide until round 4)
                           # the hard cases of SURVEY 8(d): partial-span layers (Subgraph + local re-alignment), two haplotypes
                           # (graphs that stay branched after pruning), and the per-rank shards of configs D and E on this one GPU
                           # the other execution plan of the build loop, on the same workload (32 768 windows of config C)
                           **({"C_pipeline": short_config(local, 1002, 500, 64, 32768, capi.PACBIO, check=256, pipeline=True, chunk=16384, streams=1)}
                              if capi.load_hip().vc_has_experiments() else {}),      # (an experiment: only in a library built with VC_EXPERIMENTS=1)
                           "C_mixed": short_config(local, 1011, 500, 64, 16384, capi.PACBIO, frac_partial=0.2, check=256),
                           "C_hap2": short_config(local, 1012, 500, 64, 16384, capi.PACBIO, n_haplotypes=2, snp_rate=0.01, check=256),
                           "D_shard": short_config(local, 1002, 500, 64, 125000, capi.PACBIO, first=3 * 125000, check=256),
                           "E_shard": short_config(local, 1005, 1000, 128, 6250, capi.ONT, first=5 * 6250, check=256)}
        # the same roofline for the other shapes: VALU wave-instructions per window of configs E and W by their own PMC passes
        # (profiles/r6_hbm_traffic.json), priced against the same peak (the mix is config C's: the wider classes spend a larger share
        # of a row in the packed-int16 class, whose rate is the highest -- the fraction is, if anything, flattered by a few percent)
        pc = {}
        for name, v in (line["roofline"].get("_valu_per_window_by_config") or {}).items():
            cfg_line = line["configs"].get(name)
            if cfg_line and line["roofline"].get("peak"):
                ach = v * cfg_line["windows_per_s"] / 1e6 / line["roofline"]["simds"]
                pc[name] = {"valu_wave_insts_per_window": v, "windows_per_s": cfg_line["windows_per_s"], "achieved": ach, "frac": ach / line["roofline"]["peak"]}
        line["roofline"]["per_config"] = pc
    if rank == 0:
        line["roofline"].pop("_valu_per_window_by_config", None)
    if rank == 0 and world == 1 and not a.no_extras and not a.ab and cfg_name == "C":
        line["files_to_fasta"] = files_to_fasta(ctx_params)
        if os.environ.get("VC_BENCH_LARGE_FILES", "1") != "0":
            line["files_to_fasta_large"] = files_to_fasta_large()
    if rank == 0 and world == 1 and not a.no_cpu:
        cb, ref_out = cpu_baseline(batch, ctx_params, a.cpu_seconds)
        cons_np = cons_all.cpu().numpy()
        off = np.concatenate([[0], np.cumsum(lens_all.cpu().numpy())])
        bad = sum(1 for w, c in ref_out.items() if cons_np[off[w]:off[w + 1]].tobytes() != c)
        cb["parity_windows_checked"] = len(ref_out)
        cb["parity_mismatches"] = bad
        line["cpu_baseline"] = cb
        line["speedup_vs_cpu_baseline"] = line["value"] / cb["value"] if cb["value"] else None
    if rank == 0:
        out_line = json.dumps(line)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out_line is not None:
        sys.stderr.flush()
        print(out_line, flush=True)          # the one JSON line, last thing on stdout


if __name__ == "__main__":
    main()
