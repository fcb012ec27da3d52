"""The arena of a large batch (vc_ctx::arena, four segments made by the first submit of >= 4 096 windows) goes back to the device with the context."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from vechat_amd import capi
from vechat_amd.engine import HipContext
def free_gb():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**30
base = free_gb()
b = capi.synth_batch(capi.synth_cfg(1002, 500, 64), 0, 8192)
for rep in range(3):
    c = HipContext(device=0)
    cons, st = c.consensus(b)
    used = base - free_gb()
    c.close()
    print(f"rep {rep}: in use while alive {used:.1f} GiB, after close {base - free_gb():.2f} GiB", flush=True)
