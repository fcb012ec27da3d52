#!/bin/bash
# development: first runs of the persistent build pipeline (variant library) -- parity probe, then A/B timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
export VECHAT_HIP_LIB=$R/vechat_amd/lib/variants/libvechat_hip_$2.so
echo "== check pipe" ; VC_PIPE=1 timeout 300 python tools/gpu_check.py 500x > $O/check_pipe.log 2>&1; echo "exit $?"; tail -5 $O/check_pipe.log
echo "== scale lockstep"; VC_PIPE=0 timeout 300 python tools/gpu_scale.py 32768 64 500 > $O/scale_lock.log 2>&1; echo "exit $?"; grep "rep\|status" $O/scale_lock.log | cut -c1-400
for S in 4 2 1; do
echo "== scale pipe streams=$S"; VC_PIPE=1 timeout 300 python tools/gpu_scale.py 32768 64 500 0 $S > $O/scale_pipe_s$S.log 2>&1; echo "exit $?"; grep "rep\|status\|rror" $O/scale_pipe_s$S.log | cut -c1-400
done
