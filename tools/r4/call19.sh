#!/bin/bash
# model-level checks of the channel-major default: cached prefill (+ chunks with halo / carry-in), configs[4], the routings of the prefix test
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4s; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_pool.py "tests/test_gpu_parity_r4.py" "tests/test_gpu_fulldepth.py::test_prefix_of_bench_batch_vs_fp32_oracle" -m gpu -q -s -rs > $O/tests.log 2>&1; echo "tests rc=$?"
grep -E "^\.*\[|passed|failed|^E  |^FAILED" $O/tests.log | cut -c1-330 | tail -40
