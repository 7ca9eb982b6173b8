#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4i; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 900 python tools/hc_bench.py libevo_hc_s0yb.so libevo_hc_s0yb_prio.so libevo_hc_s0yb_nw4.so libevo_hc_s2yb_nw4.so libevo_hc_s0yb_nofence.so old:libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "bench rc=$?"; grep "after-GEMM\|rc=\|reproducible False" $O/hc_bench.log | cut -c1-260
