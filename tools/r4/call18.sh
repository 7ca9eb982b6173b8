#!/bin/bash
# hyena_ct on BLOCKED z^T ([Mp / 256][3 D][256]: the projection's output tiles contiguous), batch rows padded to 64 positions
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4r; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rs -k "hyena_ct or rmsnorm_rows" > $O/tests.log 2>&1; RC=$?; echo "tests rc=$RC"
grep -E "passed|failed|^E  |^FAILED" $O/tests.log | cut -c1-300 | tail -20
HM_ROUNDS=3 timeout 400 python tools/hc_bench.py ct:libevo_mi355x.so libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "hc_bench rc=$?"; grep -E "projection|median|vs modal" $O/hc_bench.log | cut -c1-230
