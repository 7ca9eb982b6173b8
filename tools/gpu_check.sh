#!/bin/bash
# Everything a round is judged on, on one MI355X box:  gpurun --timeout 3000 -- 'bash tools/gpu_check.sh [notests|tests]'
#   -m gpu tests (with -rs and the parity numbers the tests print), smoke(), bench.py (default flags), the rocprofv3 kernel-trace
#   summaries of the 8k and the 131k scoring step, the PMC traffic passes of the Hyena operator and its SQ counters, the decode
#   kernel-trace -- summaries land in gpurun_out/check/ and are copied to profiles/rNN_*.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
R=$PWD
O=gpurun_out/check; mkdir -p $O
# (no EVO_AMD_NO_REBUILD here: ops.py rebuilds a library that is older than its sources, so the checks run the HEAD kernels)
if [ "$1" != "notests" ]; then
timeout 2400 python -m pytest tests -m gpu -q -s -rs --durations=25 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -2
grep -A28 "slowest 25 durations" $O/gpu_tests.log > $O/gpu_tests_durations.txt
grep -E "^\.*\[" $O/gpu_tests.log | sed 's/^\.*//' | cut -c1-1600 > $O/gpu_tests_parity_lines.txt; grep -E "SKIPPED|passed|failed" $O/gpu_tests.log | tail -20 >> $O/gpu_tests_parity_lines.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
fi
if [ "$1" == "tests" ]; then exit 0; fi
# HBM-side traffic of the Hyena operator FIRST (separate counter passes, --kernel-trace only), so that the bench line below carries THIS run's
# number: tools/pmc_live.py turns the two passes into gpurun_out/check/pmc_traffic_live.json, which bench.py prefers over the committed record
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/$O/pmc_zrd -o r -- python $R/tools/profile_hyena_ct.py > $R/$O/pmc_zrd.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $R/$O/pmc_zwr -o w -- python $R/tools/profile_hyena_ct.py > $R/$O/pmc_zwr.log 2>&1
cd $R && (python tools/summarize_prof.py pmc $O/pmc_zrd; python tools/summarize_prof.py pmc $O/pmc_zwr) | grep -E "^kernel|hyena_ct" > $O/hyena_ct_pmc_traffic.txt; rm -rf $O/pmc_zrd $O/pmc_zwr
cat $O/hyena_ct_pmc_traffic.txt
python tools/pmc_live.py $O/hyena_ct_pmc_traffic.txt $O/pmc_traffic_live.json "${EVO_COMMIT:-unknown}"
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o b -- python $R/bench.py --skip-131k --skip-cpu --skip-gen --skip-ab --steps 3 --warmup 1 > $R/$O/prof_bench.log 2>&1
cd $R && python tools/summarize_prof.py stats $O/prof > $O/bench_8k_kernel_stats.txt && rm -rf $O/prof
head -14 $O/bench_8k_kernel_stats.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof131 -o p -- python $R/tools/profile_131k.py > $R/$O/prof_131k.log 2>&1
cd $R && python tools/summarize_prof.py stats $O/prof131 > $O/bench_131k_kernel_stats.txt && rm -rf $O/prof131
head -14 $O/bench_131k_kernel_stats.txt
# SQ counters of the same launches (instruction mix, LDS activity / bank conflicts, wait states)
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/sq1 -o s -- python $R/tools/profile_hyena_ct.py > $R/$O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/sq2 -o s -- python $R/tools/profile_hyena_ct.py > $R/$O/sq2.log 2>&1
cd $R && (python tools/summarize_prof.py pmc $O/sq1; python tools/summarize_prof.py pmc $O/sq2) | grep -E "^kernel|hyena_ct" > $O/hyena_ct_sq_counters.txt; rm -rf $O/sq1 $O/sq2
cat $O/hyena_ct_sq_counters.txt | cut -c1-200
# the same for a decode run (BASELINE configs[4] shape: 8,192-nt prompt, greedy): per-kernel times of the generation leg
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/profg -o g -- python $R/tools/bench_generate.py --new 128 > $R/$O/prof_gen.log 2>&1
cd $R && python tools/summarize_prof.py stats $O/profg > $O/decode_kernel_stats.txt; python tools/decode_gaps.py $O/profg > $O/decode_launch_anatomy.txt 2>&1; rm -rf $O/profg; cat $O/decode_launch_anatomy.txt | tail -6
tail -1 $O/prof_gen.log; head -8 $O/decode_kernel_stats.txt
# SQ counters of the two prefill attention kernels (separate --pmc passes, --kernel-trace only) and the N = 2 code path of bench.py as a
# shared-GPU self-test (two ranks on this one GPU through the host-staged communicator: launch / barrier / max-over-ranks / sharded 131k)
if [ "$1" != "tests" ]; then
bash tools/attn_counters.sh 2 > /dev/null 2>&1; bash tools/attn_counters.sh 1 > /dev/null 2>&1
cp gpurun_out/attn_sq/form2.txt $O/attn_sq_form2.txt; cp gpurun_out/attn_sq/form1.txt $O/attn_sq_form1.txt
EVO_AMD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --skip-cpu --skip-gen > $O/bench_n2_selftest.json 2> $O/bench_n2_selftest.err; echo "n2 self-test rc=$?"
fi

