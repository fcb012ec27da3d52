#!/bin/bash
# development: A/B of the two execution plans of the build loop on 32 768 windows of config C (lock-step default; pipeline at grid mixes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
[ -n "$2" ] && export VECHAT_HIP_LIB=$R/vechat_amd/lib/variants/libvechat_hip_$2.so
export VC_PIPE_PATIENCE=5
run() { name=$1; shift; echo "== $name"; env "$@" timeout 300 python tools/gpu_scale.py 32768 64 500 $CH $ST > $O/$name.log 2>&1; echo "exit $?"; grep "rep 1\|rror\|^pipe\|^tie\|^pickup" $O/$name.log | cut -c1-250; grep "rep 1" $O/$name.log | sed 's/.*ms=//' | cut -c1-400; }
CH=0 ST=0 run lock VC_PIPE=0
CH=16384 ST=1 run pipe_f15t5 VC_PIPE=1
CH=16384 ST=1 run pipe_f14t6 VC_PIPE=1 VC_PIPE_F=3584 VC_PIPE_T=1536
