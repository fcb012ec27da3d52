"""Development: k_fwd alone on the state of one build layer, parts of its row loop switched off (timing only).
usage: gpu_fwd_lab.py LIB [layer=48] [windows=8192] [flags,flags,...]   (LIB built with -DVC_LAB)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VECHAT_HIP_LIB"] = sys.argv[1]
from vechat_amd import capi
from vechat_amd.engine import HipContext
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 48
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
flags = [int(x, 0) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
batch = capi.synth_batch(capi.synth_cfg(1002, 500, 64), 0, n)
ctx = HipContext(device=0, n_streams=1, chunk_windows=n)
ctx.submit(batch)
f = ctx.lib.vc_debug_fwd_lab
f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
names = {1: "no-store", 2: "no-ringwrite", 4: "no-ringread", 16: "no-endcell", 256: "no-slowrows"}
for fl in flags:
    ms = C.c_float(0); cr = (C.c_uint64 * 2)()
    rc = f(ctx.h, layer, 5, fl, C.byref(ms), cr)
    assert rc == 0, ctx.lib.vc_last_error(ctx.h)
    lab = "+".join(v for k, v in names.items() if fl & k) or "full"
    print(f"flags {fl:3d} {lab:40s} {ms.value:8.3f} ms  {cr[0] / ms.value / 1e9:7.2f} TCUPS  {ms.value * 1e6 / max(cr[1], 1) * 256 * 4:7.1f} ns/row/SIMD  rows {cr[1]}", flush=True)
