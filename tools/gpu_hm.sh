cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2az
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "hyena_mfma" > gpurun_out/r2az/mfma.log 2>&1; echo "mfma tests rc=$?"; tail -12 gpurun_out/r2az/mfma.log
timeout 300 python tools/hm_determinism.py 2>&1 | grep "^B=\|RESULT\|Error" | tee gpurun_out/r2az/det.log
timeout 300 python tools/bench_ops.py --only hyena --reps 10 2>&1 | grep "^\[" | tee gpurun_out/r2az/bench_ops.log
EVO_AMD_LIBNAME=libevo_hmprof.so EVO_AMD_NO_REBUILD=1 python tools/hm_stage_profile.py 2>&1 | grep "^B=" | tee gpurun_out/r2az/stages.log
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $pass | cut -c1-14)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/r2az/pmc_$tag -o h -- python $R/tools/bench_ops.py --only hyena --reps 2 > $R/gpurun_out/r2az/pmc_$tag.log 2>&1
  python $R/tools/summarize_prof.py pmc $R/gpurun_out/r2az/pmc_$tag | grep -i "hyena_mfma\|counter" | tee -a $R/gpurun_out/r2az/pmc_hyena_mfma_traffic.txt
  rm -rf $R/gpurun_out/r2az/pmc_$tag
done
