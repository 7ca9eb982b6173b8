#!/bin/bash
# round 6, second GPU call: regime test with diagnostics + guard + sp rank counts; GEMM code-placement A/B; bench line with the box block
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r6b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_r6.py -m gpu -q -s -k "regimes or routes or sequence_parallel" --durations=10 > $O/parity_r6.log 2>&1; echo "parity_r6 rc=$?"; grep -E "passed|failed" $O/parity_r6.log | tail -2
timeout 900 python tools/gemm_ab.py libevo_mi355x.so libevo_p1.so libevo_p2.so libevo_a6.so libevo_a8.so > $O/gemm_placement_ab.txt 2>&1; echo "gemm_ab rc=$?"; cat $O/gemm_placement_ab.txt
timeout 900 python bench.py --skip-131k --skip-gen --skip-cpu > $O/bench_box.json 2> $O/bench_box.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_box.json')); print(json.dumps(d.get('box'), indent=1)); print(d['value'], d.get('value_per_calibrated_box'), d.get('headline_after_legs'))"
