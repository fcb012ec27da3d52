"""EVERY window of the bench batch against the reference itself: BASELINE config C's 100 000 windows (seed 1002, the batch bench.py times) through the
device, host arrays -> host bytes, and through oracle/_ref (the reference's window.cpp + spoa built in place, SSE4.1 dispatch) on all host cores, one
window per task; byte comparison of every consensus and of every 'polished' flag.  ~9 minutes of host time for the reference.
usage: python tools/gpu_full_parity.py [n_windows=100000] [first=0] [C|E|Cmix|Chap2]      (E: 1 kb x 128, ONT, seed 1005; Cmix: 20 % partial-span layers,
seed 1011; Chap2: two haplotypes at 1 % SNPs, seed 1012 -- the shapes of bench.py's `configs`)"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from concurrent.futures import ThreadPoolExecutor
from vechat_amd import capi
from vechat_amd.engine import HipContext
import oracle_api as oa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cores = len(os.sched_getaffinity(0))
if not oa.have_ref("sse41"):
    raise SystemExit("oracle/_ref is not here (it is built from /root/reference by __graft_entry__.build() in the build container and travels with the repo)")
oa.load_ref("sse41")
which = sys.argv[3] if len(sys.argv) > 3 else "C"
cfg = {"C": lambda: capi.synth_cfg(1002, 500, 64, profile=capi.PACBIO), "E": lambda: capi.synth_cfg(1005, 1000, 128, profile=capi.ONT),
       "Cmix": lambda: capi.synth_cfg(1011, 500, 64, profile=capi.PACBIO, frac_partial=0.2),
       "Chap2": lambda: capi.synth_cfg(1012, 500, 64, profile=capi.PACBIO, n_haplotypes=2, snp_rate=0.01)}[which]()
b = capi.synth_batch(cfg, first, n, n_threads=cores)
ctx = HipContext(device=0)
t0 = time.perf_counter()
cons, status = ctx.consensus_batched(b)
t_dev = time.perf_counter() - t0
params = ctx.params
bad_bytes = bad_flag = done = 0
first_bad = []
t0 = time.time()
with ThreadPoolExecutor(cores) as ex:
    for lo in range(0, n, 4096):
        ws = list(range(lo, min(lo + 4096, n)))
        for w, (rc, pol) in zip(ws, ex.map(lambda w: oa.ref_window(b, w, params)[:2], ws)):
            if cons[w] != rc:
                bad_bytes += 1
                if len(first_bad) < 8: first_bad.append(w)
            if (int(status[w]) == capi.VC_WIN_OK) != bool(pol):
                bad_flag += 1
        done += len(ws)
        print(f"{done} windows compared, {bad_bytes} differ, {bad_flag} flags differ, {time.time() - t0:.0f} s", flush=True)
t_ref = time.time() - t0
ctx.close()
print(json.dumps({"config": which, "windows": n, "first_window": first, "consensus_differs": bad_bytes, "polished_flag_differs": bad_flag, "first_differing_windows": first_bad,
                  "device_windows_per_s_host_to_host": n / t_dev, "reference_windows_per_s": n / t_ref, "reference_threads": cores,
                  "bases_out": int(sum(len(c) for c in cons)), "sha256_16_of_device_output": hashlib.sha256(b"|".join(cons) + bytes(status)).hexdigest()[:16]}))
