#!/bin/bash
# development: parity probe + kernel-alone and default timings of library variants on the GPU box
#   tools/gpu_variants.sh TAG name1 name2 ...   (variants from tools/build_variant.sh; "main" = the product build)
TAG=$1; shift; O=gpurun_out/$TAG; mkdir -p $O
for v in "$@"; do
  lib=$PWD/vechat_amd/lib/variants/libvechat_hip_$v.so; [ "$v" = main ] && lib=$PWD/vechat_amd/lib/libvechat_hip.so
  export VECHAT_HIP_LIB=$lib
  timeout 300 python tools/gpu_check.py > $O/check_$v.log 2>&1; echo "$v: $(tail -1 $O/check_$v.log)"
  timeout 300 python tools/gpu_scale.py 32768 64 500 8192 1 > $O/alone_$v.log 2>&1; echo "$v alone: $(grep 'rep 1' $O/alone_$v.log | sed 's/cells=.*redo=/redo=/; s/trace steps.*ms=/ms=/')"
  timeout 300 python tools/gpu_scale.py 32768 64 500 0 0 > $O/default_$v.log 2>&1; echo "$v 4str: $(grep 'rep 1' $O/default_$v.log | sed 's/cells=.*redo=/redo=/; s/trace steps.*ms=/ms=/')"
done
