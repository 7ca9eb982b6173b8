#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r6d; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_r6.py -m gpu -q -s -k "one_weight" --durations=5 > $O/one_set.log 2>&1; echo "one_set rc=$?"; grep -E "^\.?\[one|passed|failed|Error" $O/one_set.log | cut -c1-600 | tail -12
timeout 900 python -m pytest tests/test_gpu_sp_two_procs.py tests/test_gpu_sp_rccl.py tests/test_gpu_parity_r4.py tests/test_gpu_model.py tests/test_gpu_gemv.py tests/test_gpu_pool.py -m gpu -q -x > $O/tests_a.log 2>&1; echo "tests_a rc=$?"; tail -4 $O/tests_a.log
timeout 1500 python bench.py --skip-cpu > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); b=d.get('box'); print({k:v for k,v in b.items() if not isinstance(v,dict)}); print(d['value'], d.get('value_per_calibrated_box'), d.get('weights_resident_GB')); g=d['generation']; print({k:g[k] for k in g if not isinstance(g[k],(dict,list))}); print(g.get('pool')); c=d['ctx131k']; print(c['value'], c['ms_per_step']); p=d.get('scaling_131k_predicted',{}); print(p.get('rank_step_ms_compute_only'), p.get('single_gpu_step_ms_batch1'), p.get('predicted_speedup_compute_only')); print(p.get('kernels'))"
