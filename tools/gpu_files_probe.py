"""Where does the device phase of files -> FASTA go on a small batch?  Builds the batch of tools/gpu_files_e2e.py (host side only), then
times submit / run+sync / collect separately, cold (workspaces of another shape) and warm, for several stream counts and the pipeline.

usage: gpu_files_probe.py [targets=200]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from vechat_amd import capi                                                     # noqa: E402
from vechat_amd.engine import HipContext                                        # noqa: E402
import gpu_files_e2e as fe                                                      # noqa: E402

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 200
os.environ["VC_FILES_HOST_ONLY"] = "1"
res = fe.main(nt, 10000, 64, "/tmp/vc_files", python_too=False, quiet=True)
batch = res["batch"]
nw = batch.n_windows
print(f"{nw} windows; host side {res['windows_per_s']:.0f} windows/s: {res['seconds_by_phase']}", flush=True)


def timed(ctx, label):
    t0 = time.time(); ctx.submit(batch); t1 = time.time(); ctx.run(); ctx.sync(); t2 = time.time(); cons, status = ctx.collect(); t3 = time.time()
    st = ctx.stats()
    print(f"  {label:28s} submit {1e3 * (t1 - t0):7.1f}  run {1e3 * (t2 - t1):7.1f}  collect {1e3 * (t3 - t2):6.1f} ms  -> {nw / (t3 - t0):8.0f} windows/s"
          f"   chunk {st['chunk_windows']} x {st['n_streams']} streams, NC {st['max_nodes']}", flush=True)
    return cons


ref = None
for streams, pipe in ((4, 0), (4, 0), (2, 0), (1, 0), (8, 0), (1, 1), (2, 1), (4, 1)):
    ctx = HipContext(device=0, mode=0, min_confidence=0.2, min_support=0.2, num_prune=3, n_streams=streams, pipeline=bool(pipe))
    ctx.consensus(capi.synth_batch(capi.synth_cfg(1, 200, 8), 0, 64))
    print(f"streams {streams} pipeline {pipe}")
    c0 = timed(ctx, "cold (other-shape workspaces)")
    c1 = timed(ctx, "warm")
    c1 = timed(ctx, "warm")
    if ref is None:
        ref = c0
    print("   identical to the first run:", c0 == ref and c1 == ref)
    ctx.close()
