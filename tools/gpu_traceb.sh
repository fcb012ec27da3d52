#!/bin/bash
# development (round 6): first runs of k_traceb (vc_traceb.h) on a box -- the global_load_lds probe, one small batch against the oracle
# under a short timeout, the GPU suite, then alternating bench runs with the walk out of LDS on and off.   usage: tools/gpu_traceb.sh TAG [variant ...]
TAG=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
vechat_amd/lib/glds_probe.bin > $O/glds_probe.txt 2>&1; tail -2 $O/glds_probe.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.txt
grep -q "smoke ok" $O/smoke.txt || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log
specs=("tracew|VC_TRACEB=0|main" "traceb|VC_TRACEB=1|main")
for v in "$@"; do specs+=("$v|VC_TRACEB=1|$v"); done
tools/gpu_ab2.sh $O "${specs[@]}" 2>&1 | tee $O/ab.txt
timeout 1200 python tools/gpu_stress.py 300 31 > $O/stress.log 2>&1; echo "sweep exit $?"; tail -1 $O/stress.log
