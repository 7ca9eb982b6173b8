cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2w
for v in mi355x pabl1 pabl8 pabl9 pabl2; do
  n=libevo_$v.so
  echo "== $v" | tee -a gpurun_out/r2w/abl.log
  EVO_AMD_LIBNAME=$n EVO_AMD_NO_REBUILD=1 timeout 300 python tools/bench_gemm.py --quick 2>&1 | grep "TF/s" | head -1 | tee -a gpurun_out/r2w/abl.log
done
