#!/bin/bash
# development: single-stream kernel times of the backtrack variants (tools/gpu_scale.py 4096 windows, one chunk) + instruction counts of k_traceb
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for spec in "tracew|VC_TRACEB=0|main" "traceb|VC_TRACEB=1|main" "tb2|VC_TRACEB=1|tb2"; do
  IFS='|' read -r label envs lib <<< "$spec"
  path=vechat_amd/lib/libvechat_hip.so; [ "$lib" != "main" ] && path=vechat_amd/lib/variants/libvechat_hip_$lib.so
  [ -f $path ] || continue
  echo "== $label"; env $envs VECHAT_HIP_LIB=$path timeout 300 python tools/gpu_scale.py 4096 64 500 4096 1 2>&1 | grep "^rep" | sed 's/cells=.*redo=/redo=/' | cut -c1-420
done 2>&1 | tee $O/scale.txt
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE"; do
  rm -rf /tmp/pmcx
  VC_TRACEB=1 timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcx -- python $R/tools/gpu_scale.py 4096 64 500 4096 1 > /dev/null 2>&1
  fc=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
  echo "== $grp" >> $O/pmc.txt; python $R/tools/pmc_summary.py $fc | grep -E "k_trace|k_fwd_dt|k_addaln" | cut -c1-600 >> $O/pmc.txt
done
cat $O/pmc.txt
