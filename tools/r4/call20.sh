#!/bin/bash
# hyena_ct variants: FIR steps 2..7 first (v1), + history exchange in front of a bare barrier (v2), the latter alone (v3)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4t; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
HM_ROUNDS=4 timeout 600 python tools/hc_bench.py ct:libevo_mi355x.so ct:libevo_ht_v1.so ct:libevo_ht_v2.so ct:libevo_ht_v3.so > $O/hc_bench.log 2>&1; echo "hc_bench rc=$?"; grep -E "median|vs modal" $O/hc_bench.log | cut -c1-230
