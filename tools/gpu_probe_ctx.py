"""Does a second context behave like the first?  (bench.py runs its hard-case configs while the headline context is alive.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vechat_amd import capi
from vechat_amd.engine import HipContext

def run(label, n, streams):
    b = capi.synth_batch(capi.synth_cfg(1003, 500, 64, profile=capi.PACBIO, n_haplotypes=2, snp_rate=0.01), 0, n, n_threads=16)
    c = HipContext(device=0, n_streams=streams)
    c.submit(b); c.run(); c.sync()
    t0 = time.perf_counter(); c.run(); c.sync(); dt = time.perf_counter() - t0
    st = c.stats()
    print(f"{label}: {n / dt:8.0f} windows/s  chunk {st['chunk_windows']} x {st['n_streams']}  device {st['device_bytes'] / 2**30:.1f} GiB", flush=True)
    c.close()

run("alone, auto", 16384, 0)
run("alone, 4", 16384, 4)
big = capi.synth_batch(capi.synth_cfg(1, 500, 64, profile=capi.PACBIO), 0, 100000, n_threads=16)
main = HipContext(device=0, profile=2)
main.submit(big); main.run(); main.sync()
print("main holds", main.stats()["device_bytes"] / 2**30, "GiB")
run("beside the headline context, auto", 16384, 0)
run("beside the headline context, 4", 16384, 4)
run("beside the headline context, 8", 16384, 8)
main.close()
