R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2at
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2at/prof_bench -o b -- python $R/bench.py --skip-131k --skip-cpu --skip-gen --steps 3 --warmup 1 > $R/gpurun_out/r2at/prof_bench.log 2>&1
EVO_AMD_GEMM=mfma timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2at/prof_bench_hw -o b -- python $R/bench.py --skip-131k --skip-cpu --skip-gen --steps 3 --warmup 1 > $R/gpurun_out/r2at/prof_bench_hw.log 2>&1
cd $R
python tools/summarize_prof.py stats gpurun_out/r2at/prof_bench > gpurun_out/r2at/bench_stats.txt
python tools/summarize_prof.py stats gpurun_out/r2at/prof_bench_hw > gpurun_out/r2at/bench_stats_hw.txt
head -24 gpurun_out/r2at/bench_stats.txt; head -12 gpurun_out/r2at/bench_stats_hw.txt
grep "^{" gpurun_out/r2at/prof_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
rm -rf gpurun_out/r2at/prof_bench gpurun_out/r2at/prof_bench_hw
# PMC traffic of hyena kernels (separate passes, counters only)
cd /tmp
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $pass | cut -c1-14)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/r2at/pmc_$tag -o h -- python $R/tools/profile_ops.py --only hyena --reps 2 > $R/gpurun_out/r2at/pmc_$tag.log 2>&1
  python $R/tools/summarize_prof.py pmc $R/gpurun_out/r2at/pmc_$tag | grep -i "hyena\|counter" | tee -a $R/gpurun_out/r2at/pmc_hyena_traffic.txt
  rm -rf $R/gpurun_out/r2at/pmc_$tag
done
