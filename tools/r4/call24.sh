#!/bin/bash
# tail form of z^T (T = 512 k + r: unpadded rows, the last r tokens of a row in a tail block): kernel tests + A/B timing
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4x; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rs -k "hyena_ct or rmsnorm_rows" > $O/tests.log 2>&1; RC=$?; echo "tests rc=$RC"
grep -E "passed|failed|^E  |^FAILED" $O/tests.log | cut -c1-300 | tail -20
HM_ROUNDS=2 timeout 300 python tools/hc_bench.py ct:libevo_mi355x.so libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "hc_bench rc=$?"; grep -E "projection|median|vs modal" $O/hc_bench.log | cut -c1-230
