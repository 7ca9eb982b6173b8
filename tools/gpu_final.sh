#!/bin/bash
# the round's last check (reduced form of tools/gpu_check.sh for a short GPU budget): all GPU tests, smoke, bench, kernel-trace summaries of the 8k and 131k
# steps, then the PMC traffic passes of the group-major Hyena launch
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
R=$PWD
O=gpurun_out/final; mkdir -p $O
# (no EVO_AMD_NO_REBUILD here: ops.py rebuilds a library that is older than its sources, so the checks run the HEAD kernels)
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -E "^E  |FAILED|passed|failed" $O/gpu_tests.log | tail -6
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o b -- python $R/bench.py --skip-131k --skip-cpu --skip-gen --steps 3 --warmup 1 > $R/$O/prof_bench.log 2>&1
cd $R && python tools/summarize_prof.py stats $O/prof > $O/bench_8k_kernel_stats.txt && rm -rf $O/prof
head -10 $O/bench_8k_kernel_stats.txt
cd /tmp
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/$O/pmc_rd -o r -- python $R/tools/profile_hyena_zg.py > $R/$O/pmc_rd.log 2>&1
timeout 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $R/$O/pmc_wr -o w -- python $R/tools/profile_hyena_zg.py > $R/$O/pmc_wr.log 2>&1
cd $R && (python tools/summarize_prof.py pmc $O/pmc_rd; python tools/summarize_prof.py pmc $O/pmc_wr) | grep -E "^kernel|hyena_mfma" > $O/hyena_zg_pmc_traffic.txt; rm -rf $O/pmc_rd $O/pmc_wr
cat $O/hyena_zg_pmc_traffic.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof131 -o p -- python $R/tools/profile_131k.py > $R/$O/prof_131k.log 2>&1
cd $R && python tools/summarize_prof.py stats $O/prof131 > $O/bench_131k_kernel_stats.txt && rm -rf $O/prof131
head -8 $O/bench_131k_kernel_stats.txt
