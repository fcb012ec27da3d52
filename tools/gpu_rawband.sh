#!/bin/bash
# development (round 6): the band for raw int16 rows (VC_BAND_RAW) -- GPU suite, sweep, config W's shape with and without.   usage: tools/gpu_rawband.sh TAG
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
grep -q "smoke ok" $O/smoke.txt || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest_gpu.log
for rep in 1 2; do for br in 0 1; do
  echo "W  VC_BAND_RAW=$br rep $rep: $(VC_BAND_RAW=$br VC_SEED=1007 timeout 600 python tools/gpu_scale.py 1024 12 3000 2>&1 | grep '^rep 1' | sed 's/cells=[^ ]* //; s/rows=.*redo=/redo=/; s/trace steps.*dev=/dev=/' | cut -c1-300)"
done; done 2>&1 | tee $O/W_ab.txt
for br in 0 1; do
  echo "raw scores (5,-4,-8) C shape VC_BAND_RAW=$br: $(VC_BAND_RAW=$br VC_SCORES=5,-4,-8 timeout 600 python tools/gpu_scale.py 8192 64 500 2>&1 | grep '^rep 1' | sed 's/cells=[^ ]* //; s/rows=.*redo=/redo=/; s/trace steps.*dev=/dev=/' | cut -c1-200)"
done 2>&1 | tee -a $O/W_ab.txt
timeout 1200 python tools/gpu_stress.py 400 53 > $O/stress.log 2>&1; echo "sweep exit $?"; tail -1 $O/stress.log
