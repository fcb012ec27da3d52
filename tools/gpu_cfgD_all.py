"""Config D (1 M windows of config C's distribution over 8 ranks) on the one GPU there is: the eight ranks' shares one after the other through
ONE warm context, host arrays -> host bytes (submit / run / collect in slices, HipContext.consensus_batched), with a sample of every share run
through the reference itself (oracle/_ref) and compared byte for byte.  Shows that the rate holds over a million windows (no growth of the
workspaces, no drift) and what an 8-GPU node's ranks each do.      usage: python tools/gpu_cfgD_all.py [ranks=8] [windows_per_rank=125000] [check=128] [D|E]
(E: BASELINE config E -- 50 000 windows of 1 kb x 128 reads, ONT profile, seed 1005 -- 6 250 per rank)"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from concurrent.futures import ThreadPoolExecutor
from vechat_amd import capi
from vechat_amd.engine import HipContext
import oracle_api as oa

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 125000
check = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cores = len(os.sched_getaffinity(0))
have_ref = oa.have_ref("sse41")
if have_ref:
    oa.load_ref("sse41")
ctx = HipContext(device=0)
which = sys.argv[4] if len(sys.argv) > 4 else "D"
cfg = capi.synth_cfg(1005, 1000, 128, profile=capi.ONT) if which == "E" else capi.synth_cfg(1002, 500, 64, profile=capi.PACBIO)
tot_w = tot_t = 0.0
rows = []
for r in range(ranks):
    t0 = time.perf_counter()
    b = capi.synth_batch(cfg, r * per, per, n_threads=cores)
    t_syn = time.perf_counter() - t0
    t0 = time.perf_counter()
    cons, status = ctx.consensus_batched(b)
    dt = time.perf_counter() - t0
    st = ctx.stats()
    bad = -1
    ws = []
    if have_ref and check:
        ws = list(range(0, per, max(1, per // check)))[:check]
        with ThreadPoolExecutor(cores) as ex:
            ref = list(ex.map(lambda w: oa.ref_window(b, w, ctx.params)[0], ws))
        bad = sum(1 for w, x in zip(ws, ref) if cons[w] != x)
    dig = hashlib.sha256(b"|".join(cons) + bytes(status)).hexdigest()[:16]
    row = {"rank": r, "windows": per, "windows_per_s_host_to_host": per / dt, "seconds": dt, "synth_seconds": t_syn, "not_ok": int((status > 1).sum()),
           "bases_out": int(sum(len(c) for c in cons)), "device_bytes_gib": st.get("device_bytes", 0) / 2**30, "checked_against_reference": len(ws), "mismatches": bad, "sha256_16": dig}
    rows.append(row); tot_w += per; tot_t += dt
    print(json.dumps(row), flush=True)
    del b, cons, status
ctx.close()
print(json.dumps({"config": which + " on one GPU, rank after rank", "windows": int(tot_w), "device_seconds": tot_t, "windows_per_s": tot_w / tot_t,
                  "slowest_rank": min(x["windows_per_s_host_to_host"] for x in rows), "fastest_rank": max(x["windows_per_s_host_to_host"] for x in rows),
                  "mismatches": sum(max(0, x["mismatches"]) for x in rows), "checked": sum(x["checked_against_reference"] for x in rows)}))
