cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
export EVO_GEMM_WAVES=4
for v in mi355x abl1 abl2 abl4 abl8 abl3 abl9 abl15; do
  n=libevo_$v.so
  echo "== $v" | tee -a gpurun_out/r2u/abl.log
  EVO_AMD_LIBNAME=$n EVO_AMD_NO_REBUILD=1 timeout 300 python tools/bench_gemm.py --quick 2>&1 | grep "TF/s" | head -1 | tee -a gpurun_out/r2u/abl.log
done
