cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2as
timeout 900 python -m pytest tests/test_gpu_gemv.py -q -x > gpurun_out/r2as/gemv_tests.log 2>&1; echo "gemv tests rc=$?"; tail -3 gpurun_out/r2as/gemv_tests.log
python tools/bench_generate.py --new 256 2>&1 | tail -3 | tee gpurun_out/r2as/gen.log
