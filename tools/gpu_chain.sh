#!/bin/bash
# development (round 6): the narrow band of the re-alignment rounds (VC_BAND_CHAIN_COLS) -- GPU suite, redo counts, alternating bench runs.  usage: tools/gpu_chain.sh TAG variant...
TAG=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
grep -q "smoke ok" $O/smoke.txt || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest_gpu.log
for lib in main "$@"; do
  path=vechat_amd/lib/libvechat_hip.so; [ "$lib" != "main" ] && path=vechat_amd/lib/variants/libvechat_hip_$lib.so
  [ -f $path ] || continue
  echo "== $lib"
  VECHAT_HIP_LIB=$path timeout 300 python tools/gpu_scale.py 32768 64 500 2>&1 | grep "^rep 1" | sed 's/cells=.*far=[0-9]* //' | cut -c1-60
  VC_PROFILE=ont VC_SEED=1005 VECHAT_HIP_LIB=$path timeout 300 python tools/gpu_scale.py 2048 128 1000 2>&1 | grep "^rep 1" | sed 's/cells=.*far=[0-9]* //' | cut -c1-60
done 2>&1 | tee $O/redo.txt
specs=("main||main"); for v in "$@"; do specs+=("$v||$v"); done
tools/gpu_ab2.sh $O "${specs[@]}" 2>&1 | tee $O/ab.txt
