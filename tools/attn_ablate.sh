#!/bin/bash
# Ablation timing of attn_fwd_w64_kernel (csrc/attn_w64.hip): measurement builds that drop one ingredient of a trip each (results are wrong,
# only the time is read).  Build here:  bash tools/attn_ablate.sh build     Run on the GPU box:  bash tools/attn_ablate.sh run > gpurun_out/abl.txt
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
VARS="base:-DW_BASE noexp:-DW_ABL_NOEXP noside:-DW_ABL_NOSIDE nolds:-DW_ABL_NOLDS nodma:-DW_ABL_NODMA nobar:-DW_ABL_NOBAR valu0:-DW_ABL_NOEXP,-DW_ABL_NOSIDE mfma:-DW_ABL_NOEXP,-DW_ABL_NOSIDE,-DW_ABL_NOLDS,-DW_ABL_NODMA,-DW_ABL_NOBAR $EXTRA_VARS"
if [ "$1" == "build" ]; then
  for v in $VARS; do n=${v%%:*}; f=${v#*:}; EVO_AMD_LIBNAME=libevo_abl_$n.so EVO_AMD_HIPCC_FLAGS="${f//,/ }" python -m evo_amd._build > /dev/null 2>&1 & done; wait; ls evo_amd/_lib/
else
  for v in $VARS; do n=${v%%:*}; echo "== $n"; EVO_AMD_LIBNAME=libevo_abl_$n.so EVO_AMD_NO_REBUILD=1 timeout 120 python tools/attn_bench.py 2>&1 | grep "B=" | sed 's/ | rows.*//'; done
fi
