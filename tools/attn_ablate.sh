#!/bin/bash
# Ablation timing of attn_fwd_w64_kernel (csrc/attn_w64.hip): measurement builds that drop one ingredient of a trip each (results are wrong,
# only the time is read).  Build here:  bash tools/attn_ablate.sh build     Run on the GPU box:  bash tools/attn_ablate.sh run > gpurun_out/abl.txt
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
VARS="$EXTRA_VARS"
if [ "$1" == "build" ]; then
  O=/tmp/abl_objs; rm -rf $O; mkdir -p $O
  CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-inline-asm"
  for f in evo_amd/csrc/*.hip; do b=$(basename $f .hip); [ $b == attn_w64 ] || $CC -c $f -o $O/$b.o & done; wait
  n=0
  for v in $VARS; do nm=${v%%:*}; fl=${v#*:}
    ( $CC -fno-slp-vectorize -Wno-unused-value ${fl//,/ } -c evo_amd/csrc/attn_w64.hip -o $O/w64_$nm.o && $CC -shared -o evo_amd/_lib/libevo_abl_$nm.so $O/w64_$nm.o $(ls $O/*.o | grep -v w64_) ) &
    n=$((n+1)); [ $((n % 4)) == 0 ] && wait
  done; wait; ls evo_amd/_lib/
else
  for v in $VARS; do n=${v%%:*}; echo "== $n"; EVO_AMD_LIBNAME=libevo_abl_$n.so EVO_AMD_NO_REBUILD=1 timeout 120 python tools/attn_bench.py 2>&1 | grep "B=" | sed 's/ | rows.*//'; done
fi
