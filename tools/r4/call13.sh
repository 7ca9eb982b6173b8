#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4m; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sp_two_procs.py tests/test_gpu_sp_rccl.py "tests/test_gpu_parity_r4.py::test_configs3_eight_virtual_ranks_16385_token_shards_d4096" tests/test_gpu_model.py -m gpu -q -s -rs > $O/tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|^E  " $O/tests.log | cut -c1-300 | tail -12
bash tools/gpu_check.sh notests
