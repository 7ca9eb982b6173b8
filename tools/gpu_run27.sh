cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
EVO_GEMM_WAVES=4 timeout 600 python -m pytest tests/test_gpu_gemm.py -q > gpurun_out/r2s/gemm4_tests.log 2>&1; echo "gemm4 tests rc=$?"; tail -5 gpurun_out/r2s/gemm4_tests.log
EVO_GEMM_WAVES=4 timeout 600 python tools/bench_gemm.py 2>&1 | grep "TF/s" | tee gpurun_out/r2s/gemm4.log
timeout 600 python tools/bench_gemm.py --quick 2>&1 | grep "TF/s" | tee gpurun_out/r2s/gemm8.log
