cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2x
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x > gpurun_out/r2x/gemmr_tests.log 2>&1; echo "gemmr tests rc=$?"; tail -15 gpurun_out/r2x/gemmr_tests.log
timeout 600 python tools/bench_gemm.py 2>&1 | grep "TF/s" | tee gpurun_out/r2x/gemmr.log
for v in rabl1 rabl16 rabl17; do
  echo "== $v" | tee -a gpurun_out/r2x/abl.log
  EVO_AMD_LIBNAME=libevo_$v.so EVO_AMD_NO_REBUILD=1 timeout 300 python tools/bench_gemm.py --quick 2>&1 | grep "TF/s" | head -1 | tee -a gpurun_out/r2x/abl.log
done
