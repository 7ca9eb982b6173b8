cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2aa
for k in 1024 2048 4096 8192; do
  timeout 600 python tools/bench_gemm.py --k $k 2>&1 | grep "TF/s" | head -1 | tee -a gpurun_out/r2aa/ksweep.log
done
