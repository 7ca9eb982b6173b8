cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "hyena_mfma" > gpurun_out/r2e/mfma.log 2>&1; echo "mfma tests rc=$?"; tail -30 gpurun_out/r2e/mfma.log
timeout 300 python tools/bench_ops.py --only hyena --reps 10 2>&1 | grep "^\[" | tee gpurun_out/r2e/bench_ops.log
