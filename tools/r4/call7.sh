#!/bin/bash
# round 4, GPU call 7: hyena_cs with the next tile's window prefetched into registers (HC_PREF=1): tests, A/B, profile
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4g; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "hyena_cs" -m gpu -q -x -s > $O/cs_tests.log 2>&1; echo "cs tests rc=$?"
grep -E "passed|failed|Error|assert" $O/cs_tests.log | cut -c1-300 | tail -8
timeout 900 python tools/hc_bench.py libevo_mi355x.so libevo_hc_p0.so libevo_hc_pyb.so libevo_hc_ps2yb.so old:libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "bench rc=$?"; grep "after-GEMM\|rc=" $O/hc_bench.log | cut -c1-260
EVO_AMD_LIBNAME=libevo_hcprof_pyb.so timeout 300 python tools/hc_stage_profile.py > $O/prof_pyb.log 2>&1; echo "prof rc=$?"; grep -v amdgpu.ids $O/prof_pyb.log | grep -A3 "after-GEMM" | cut -c1-300
