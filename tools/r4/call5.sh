#!/bin/bash
# round 4, GPU call 5: timing probe -- does the y store pattern (32 partial lines per store instruction) bound the kernel?
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4e; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 900 python tools/hc_bench.py libevo_mi355x.so libevo_hc_burst.so libevo_hc_yb.so libevo_hc_burstyb.so old:libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "bench rc=$?"; grep "after-GEMM\|rc=" $O/hc_bench.log | cut -c1-260
