#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4y; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "rmsnorm_rows" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
