#!/bin/bash
# Cycle counters of the ablation builds of attn_fwd_w64_kernel at 1 x 131,073 (one rocprofv3 --pmc pass each): separates clock effects
# (GRBM_GUI_ACTIVE per XCD vs wall time) from cycle effects.   bash tools/attn_cycles.sh base noexp ...  -> gpurun_out/attn_sq/cycles.txt
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
R=$PWD; O=gpurun_out/attn_sq; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp
: > $R/$O/cycles.txt
for n in "$@"; do
  export EVO_AMD_LIBNAME=libevo_abl_$n.so
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/$O/c -o a -- python $R/tools/profile_attn.py 131073 1 > $R/$O/c.log 2>&1
  echo "== $n" >> $R/$O/cycles.txt
  (cd $R; python tools/summarize_prof.py pmc $O/c | grep -E "attn_fwd" | awk '{printf "%-28s %18.0f\n", $2, $4}' >> $O/cycles.txt; python - <<PY >> $O/cycles.txt
import csv,glob
fs=glob.glob("$O/c/**/*kernel_trace.csv",recursive=True)
d=[ (float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e6 for f in fs for r in csv.DictReader(open(f)) if "attn_fwd" in r["Kernel_Name"]]
print("kernel ms (profiled):", " ".join(f"{x:.2f}" for x in d))
PY
  rm -rf $O/c)
done
cat $R/$O/cycles.txt
