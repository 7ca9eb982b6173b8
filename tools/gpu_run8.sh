cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/r2h/pmc_$tag -o hy -- python tools/bench_ops.py --only hyena --reps 2 > gpurun_out/r2h/pmc_$tag.log 2>&1
  python tools/summarize_prof.py pmc gpurun_out/r2h/pmc_$tag | grep -i "hyena_mfma\|hyena_apply\|counter" | tee -a gpurun_out/r2h/pmc_summary.txt
  rm -rf gpurun_out/r2h/pmc_$tag
done
