#!/bin/bash
# Parity job for the LLVM developer option the product library is built with: the GPU suite and the randomised sweep
# through the library built WITH -structurizecfg-skip-uniform-regions (the product) and WITHOUT it (variant "plain", made by
# `VC_PLAIN_CFG=1 tools/build_variant.sh plain`), both against the oracle / the reference fixtures; plus a direct byte comparison of
# the two libraries' consensus on 2 048 windows of config C, in both execution plans.   usage: tools/gpu_flag_parity.sh TAG
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
P=$R/vechat_amd/lib/variants/libvechat_hip_plain.so
{
echo "ROCm: $(cat /opt/rocm/.info/version 2>/dev/null)  hipcc: $(/opt/rocm/bin/hipcc --version | grep -i 'clang version' | head -1)"
for lib in product plain; do
  [ $lib = plain ] && export VECHAT_HIP_LIB=$P || unset VECHAT_HIP_LIB
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_$lib.log 2>&1; echo "$lib: pytest exit $? $(grep -E 'passed|failed' $O/pytest_$lib.log)"
  timeout 1500 python tools/gpu_stress.py 300 29 > $O/stress_$lib.log 2>&1; echo "$lib: sweep exit $? $(tail -1 $O/stress_$lib.log)"
done
unset VECHAT_HIP_LIB
python - <<PY
import os, subprocess, sys, hashlib
sys.path.insert(0, "$R")
code = '''
import sys, hashlib
sys.path.insert(0, "$R")
from vechat_amd import capi
from vechat_amd.engine import HipContext
b = capi.synth_batch(capi.synth_cfg(1002, 500, 64), 0, 2048)
for pl in ((False, True) if capi.load_hip().vc_has_experiments() else (False,)):
    c = HipContext(device=0, pipeline=pl); cons, st = c.consensus(b); c.close()
    print(pl, hashlib.sha256(b"|".join(cons) + bytes(st)).hexdigest())
'''
outs = []
for lib in (None, "$P"):
    env = dict(os.environ)
    if lib: env["VECHAT_HIP_LIB"] = lib
    outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout)
print("product:", outs[0].strip().replace("\\n", " ; ")); print("plain:  ", outs[1].strip().replace("\\n", " ; "))
print("consensus of 2048 config-C windows, both plans:", "IDENTICAL with and without the option" if outs[0] == outs[1] and outs[0] else "DIFFERENT")
PY
} 2>&1 | tee $O/flag_parity.txt
