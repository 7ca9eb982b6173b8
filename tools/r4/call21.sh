#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4u; mkdir -p $O
for G in "" 4 16; do
EVO_GEMM_GROUP_M=$G timeout 300 python tools/gemm_t_ab.py > $O/gemm_t_ab_$G.log 2>&1; echo "rc=$?"; grep -E "GROUP|round 1" $O/gemm_t_ab_$G.log
done
