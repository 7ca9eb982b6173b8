#!/bin/bash
# round 4, GPU call 6: blocked-y probes: pieces inside the FIR loops (HC_SPREAD=2) vs spread over the whole tile; phase profiles
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4f; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 900 python tools/hc_bench.py libevo_hc_yb.so libevo_hc_s2yb.so old:libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "bench rc=$?"; grep "after-GEMM\|rc=" $O/hc_bench.log | cut -c1-260
EVO_AMD_LIBNAME=libevo_hcprof_yb.so timeout 300 python tools/hc_stage_profile.py > $O/prof_yb.log 2>&1; echo "prof yb rc=$?"; grep -v amdgpu.ids $O/prof_yb.log | grep -A3 "after-GEMM" | cut -c1-300
EVO_AMD_LIBNAME=libevo_hcprof_s2yb.so timeout 300 python tools/hc_stage_profile.py > $O/prof_s2yb.log 2>&1; echo "prof s2yb rc=$?"; grep -v amdgpu.ids $O/prof_s2yb.log | grep -A3 "after-GEMM" | cut -c1-300
