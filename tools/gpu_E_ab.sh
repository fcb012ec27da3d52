#!/bin/bash
# development: config E's shape (1 kb x 128, ONT) through library variants, two runs each, alternating.   usage: tools/gpu_E_ab.sh TAG variant...
TAG=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do
  for lib in main "$@"; do
    path=vechat_amd/lib/libvechat_hip.so; [ "$lib" != "main" ] && path=vechat_amd/lib/variants/libvechat_hip_$lib.so
    [ -f $path ] || continue
    echo "$lib rep $rep: $(VC_PROFILE=ont VC_SEED=1005 VECHAT_HIP_LIB=$path timeout 300 python tools/gpu_scale.py 6250 128 1000 2>&1 | grep '^rep 1' | sed 's/cells=[^ ]* //; s/rows=.*redo=/redo=/; s/trace steps.*dev=/dev=/' | cut -c1-330)"
  done
done 2>&1 | tee $O/E_ab.txt
