#!/bin/bash
# hyena_ct: batch-row pitch of z^T 8 vs 64 positions, the mirrored raster of the swapped-operand projection, phase profile
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4q; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
for AL in 8 64; do
HC_ZT_ALIGN=$AL HM_ROUNDS=3 timeout 400 python tools/hc_bench.py ct:libevo_mi355x.so libevo_mi355x.so > $O/hc_bench_$AL.log 2>&1; echo "hc_bench align $AL rc=$?"; grep -E "projection|median" $O/hc_bench_$AL.log | cut -c1-230
done
for AL in 8 64; do
HC_ZT_ALIGN=$AL EVO_AMD_LIBNAME=libevo_htprof.so timeout 300 python tools/ht_stage_profile.py > $O/ht_prof_$AL.log 2>&1; echo "profile align $AL rc=$?"; grep -E "^----|wave" $O/ht_prof_$AL.log | cut -c1-260
done
