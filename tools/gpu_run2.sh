cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
for v in A B C D; do
  EVO_AMD_LIBNAME=libhy_$v.so EVO_AMD_NO_REBUILD=1 python tools/bench_ops.py --only hyena --reps 20 2>&1 | grep "^\[" | tee -a gpurun_out/r2b/hyena_variants.log
done
timeout 1200 python -m pytest tests/test_gpu_fulldepth.py -q -s > gpurun_out/r2b/fulldepth.log 2>&1; echo "fulldepth rc=$?"
grep -n "^\[\|passed\|failed\|Error\|assert" gpurun_out/r2b/fulldepth.log | head -60
