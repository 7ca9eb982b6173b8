#!/bin/bash
# round 4, GPU call 3: hyena_cs with two window tiles in flight (HC_AHEAD=2) vs one; per-phase clock profiles; parity tests again
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4c; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "hyena_cs" -m gpu -q -x -s > $O/cs_tests.log 2>&1; echo "cs tests rc=$?"
grep -E "passed|failed|Error|assert" $O/cs_tests.log | cut -c1-300 | tail -8
timeout 600 python tools/hc_bench.py libevo_mi355x.so libevo_hc_a1.so old:libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "bench rc=$?"; grep "after-GEMM\|rc=" $O/hc_bench.log | cut -c1-260
EVO_AMD_LIBNAME=libevo_hcprof.so timeout 300 python tools/hc_stage_profile.py > $O/prof_a2.log 2>&1; echo "prof a2 rc=$?"; cat $O/prof_a2.log | grep -v amdgpu.ids | cut -c1-300
EVO_AMD_LIBNAME=libevo_hcprof1.so timeout 300 python tools/hc_stage_profile.py > $O/prof_a1.log 2>&1; echo "prof a1 rc=$?"; cat $O/prof_a1.log | grep -v amdgpu.ids | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity_r4.py -m gpu -q -s -rs > $O/parity.log 2>&1; echo "parity rc=$?"
grep -E "^\[|passed|failed|Error|^E " $O/parity.log | cut -c1-900 | tail -30
