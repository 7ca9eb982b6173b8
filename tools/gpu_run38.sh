cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2ag
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x > gpurun_out/r2ag/gemm_tests.log 2>&1; echo "gemm tests rc=$?"; tail -5 gpurun_out/r2ag/gemm_tests.log
timeout 600 python tools/bench_gemm.py 2>&1 | grep "TF/s" | tee gpurun_out/r2ag/gemm.log
EVO_AMD_LIBNAME=libevo_grprof.so EVO_AMD_NO_REBUILD=1 python tools/gemm_stage_profile.py 2>&1 | grep "^M=" | tee gpurun_out/r2ag/stages.log
