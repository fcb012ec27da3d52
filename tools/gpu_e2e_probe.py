"""bench.py's host-to-host leg alone, several times: how steady is it?  usage: gpu_e2e_probe.py [windows=65536] [reps=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from vechat_amd import capi
from vechat_amd.engine import HipContext
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b = capi.synth_batch(capi.synth_cfg(1, 500, 64, profile=capi.PACBIO), 0, n, n_threads=16)
main = HipContext(device=0, profile=2)                 # the headline context is alive while bench.py measures this leg
main.submit(b.slice(0, min(n, 100000))); main.run(); main.sync()
for r in range(reps):
    rate, _ = bench.e2e_rate(b, 0)
    print(f"e2e {rate:8.0f} windows/s", flush=True)
main.close()
