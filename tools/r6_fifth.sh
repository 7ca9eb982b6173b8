#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r6e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attention or rope" > $O/attn_tests.log 2>&1; echo "attn kernel tests rc=$?"; tail -4 $O/attn_tests.log
timeout 1200 python -m pytest tests/test_gpu_fulldepth.py -m gpu -q -s -k "attention_h32 or prefix_of_bench or 131k_forward" > $O/attn_full.log 2>&1; echo "attn full rc=$?"; grep -E "^\.?\[attention|passed|failed" $O/attn_full.log | cut -c1-300 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity_r6.py tests/test_gpu_model.py tests/test_gpu_parity_r4.py tests/test_gpu_sp_two_procs.py tests/test_gpu_pool.py -m gpu -q -x > $O/tests_b.log 2>&1; echo "tests_b rc=$?"; tail -4 $O/tests_b.log
timeout 600 python tools/attn_bench.py > $O/attn_bench.txt 2>&1; cat $O/attn_bench.txt | tail -12
timeout 1500 python bench.py --skip-cpu --skip-gen > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d.get('value_per_calibrated_box'), d['ms_per_step']); print(d['kernels']['attn_fwd']); print({k:v for k,v in d['attention_round4_kernel'].items() if k!='note'}); c=d['ctx131k']; print(c['value'], c['ms_per_step'], c['kernels'].get('attn_fwd')); print({k:v for k,v in c.get('attention_round4_kernel',{}).items() if k!='note'})"
