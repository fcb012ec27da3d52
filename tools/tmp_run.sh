timeout 300 python tools/gpu_check.py > gpurun_out/check_main.log 2>&1; echo "main: $(tail -1 gpurun_out/check_main.log)"
for gb in 0 64 32 16; do
echo "scratch ${gb} GB: $(VC_SCRATCH_GB=$gb timeout 300 python tools/gpu_scale.py 32768 64 500 0 0 2>&1 | grep 'rep 1' | sed 's/cells=.*redo=/redo=/; s/trace steps.*NC/NC/; s/ms=.*//')"
done
timeout 900 python tools/gpu_files_e2e.py 200 10000 64 /tmp/vc_files 2>&1 | tail -8
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_main.log 2>&1; tail -2 gpurun_out/pytest_main.log
