#!/bin/bash
# One GPU-box call per measurement round: gpu tests, PMC passes (own runs, kernel-trace only) + VALU calibration -> traffic / count
# files, THEN the bench line (its roofline reads those counts), then rocprofv3 kernel stats of the same command.
# Everything lands in gpurun_out/$TAG/.   usage: tools/gpu_round.sh TAG [skip-tests]
TAG=${1:-r6}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ "$2" != "skip-tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
fi
cd /tmp && export TMPDIR=/tmp
i=0
rm -f $O/pmc_counters.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/pmc/p$i
  VC_STATS_JSON=$O/pmc_stats.json timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc/p$i -- python $R/tools/gpu_scale.py 4096 64 500 4096 1 > $O/pmc$i.log 2>&1
  fc=$(find /tmp/pmc/p$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $grp" >> $O/pmc_counters.txt; python $R/tools/pmc_summary.py $fc | cut -c1-700 >> $O/pmc_counters.txt
done
# the other workload shapes, VALU instruction counts only (roofline.per_config): config E (1 kb x 128, ONT) and W (3 kb x 12)
rm -rf /tmp/pmcE /tmp/pmcW
VC_PROFILE=ont VC_SEED=1005 VC_STATS_JSON=$O/pmc_stats_E.json timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pmcE -- python $R/tools/gpu_scale.py 1024 128 1000 > $O/pmcE.log 2>&1
VC_SEED=1007 VC_STATS_JSON=$O/pmc_stats_W.json timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pmcW -- python $R/tools/gpu_scale.py 512 12 3000 > $O/pmcW.log 2>&1
$R/vechat_amd/lib/valu_peak.bin > $O/valu_peak.txt
python $R/tools/make_traffic_json.py /tmp/pmc $O/pmc_stats.json $O/valu_peak.txt $O E=/tmp/pmcE:$O/pmc_stats_E.json W=/tmp/pmcW:$O/pmc_stats_W.json
cp $O/r6_hbm_traffic.json $R/profiles/r6_hbm_traffic.json      # (on this box: the bench below prices its step against these counts)
cd $R
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -c 1500 $O/bench.json
cd /tmp
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --no-cpu --no-extras > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
f=$(grep -l k_fwd $(find /tmp/prof_stats -name "*kernel_stats.csv") /dev/null < /dev/null | head -1); cp "$f" $O/kernel_stats.csv; head -12 $O/kernel_stats.csv      # (the calibration binary the bench starts writes a stats file of its own)
# the same bench with the process pinned to two cores (an 8-rank node leaves each rank about two): stream workers and the host side must not need more
cd $R
{ echo "# python bench.py --no-cpu --no-extras --steps 3: all host cores, then taskset -c 0-1"
  for pin in "" "taskset -c 0-1"; do
    timeout 600 $pin python bench.py --no-cpu --no-extras --steps 3 2> /dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-16s value %.0f windows/s  value_e2e %s  ms_per_step %.1f' % ('${pin:-all cores}', j['value'], j.get('value_e2e'), j['ms_per_step']))"
  done; } > $O/two_core_bench.txt 2>&1 < /dev/null
cat $O/two_core_bench.txt
