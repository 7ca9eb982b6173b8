#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4a1; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 100 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "both_forms_of_zt" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^\.*\[|passed|failed|^E  " $O/tests.log | cut -c1-300 | tail -12
