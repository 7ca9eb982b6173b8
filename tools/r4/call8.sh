#!/bin/bash
# round 4, GPU call 8: hyena_cs with the scan powers / staged outputs requested ahead of their use
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4h; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "hyena_cs" -m gpu -q -x -s > $O/cs_tests.log 2>&1; echo "cs tests rc=$?"
grep -E "passed|failed|Error|assert" $O/cs_tests.log | cut -c1-300 | tail -8
timeout 900 python tools/hc_bench.py libevo_mi355x.so libevo_hc_yb.so libevo_hc_s2yb.so libevo_hc_s0yb.so old:libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "bench rc=$?"; grep "after-GEMM\|rc=" $O/hc_bench.log | cut -c1-260
EVO_AMD_LIBNAME=libevo_hcprof_yb.so timeout 300 python tools/hc_stage_profile.py > $O/prof_yb.log 2>&1; echo "prof rc=$?"; grep -v amdgpu.ids $O/prof_yb.log | grep -A3 "after-GEMM" | cut -c1-300
