#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r6c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_r6.py -m gpu -q -s -k "regimes or routes or sequence_parallel" --durations=10 > $O/parity_r6.log 2>&1; echo "parity_r6 rc=$?"; grep -E "passed|failed|OUTSIDE" $O/parity_r6.log | cut -c1-600 | tail -8
timeout 900 python bench.py --skip-131k --skip-gen --skip-cpu --skip-ab > $O/bench_box.json 2> $O/bench_box.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_box.json')); b=d.get('box'); print({k:v for k,v in b.items() if not isinstance(v,dict)}); print(d['value'], d.get('value_per_calibrated_box'))"
timeout 900 python -m pytest tests/test_gpu_sp_two_procs.py tests/test_gpu_sp_rccl.py tests/test_gpu_parity_r4.py -m gpu -q -x -k "sp or configs3" > $O/sp_tests.log 2>&1; echo "sp tests rc=$?"; tail -3 $O/sp_tests.log
