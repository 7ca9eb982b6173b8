cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2v
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x > gpurun_out/r2v/gemmp_tests.log 2>&1; echo "gemmp tests rc=$?"; tail -15 gpurun_out/r2v/gemmp_tests.log
timeout 600 python tools/bench_gemm.py 2>&1 | grep "TF/s" | tee gpurun_out/r2v/gemmp.log
EVO_GEMM_FORM=0 timeout 600 python tools/bench_gemm.py --quick 2>&1 | grep "TF/s" | tee gpurun_out/r2v/gemm8.log
