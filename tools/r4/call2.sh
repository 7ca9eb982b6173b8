#!/bin/bash
# round 4, GPU call 2: first run of the channel-stationary Hyena kernel (csrc/hyena_cs.hip): kernel tests, A/B timing against the
# round-3 kernel (8 and 4 waves per workgroup), then the two parity tests that failed on their own pins in call 1.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4b; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "hyena_cs" -m gpu -q -x -s > $O/cs_tests.log 2>&1; rc=$?; echo "cs tests rc=$rc"
grep -E "passed|failed|Error|assert" $O/cs_tests.log | cut -c1-300 | tail -15
timeout 600 python tools/hc_bench.py libevo_mi355x.so libevo_hc_nw4.so old:libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "bench rc=$?"; cat $O/hc_bench.log | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_parity_r4.py -m gpu -q -s -rs > $O/parity.log 2>&1; echo "parity rc=$?"
grep -E "^\[|passed|failed|Error|^E " $O/parity.log | cut -c1-700 | tail -30
