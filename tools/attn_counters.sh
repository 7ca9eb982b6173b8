#!/bin/bash
# SQ counters of the prefill attention kernel at 1 x 131,073 and 8 x 8,193 (H = 32): separate rocprofv3 --pmc passes, --kernel-trace only.
#   gpurun -- 'bash tools/attn_counters.sh [form]'   form = 2 (attn_fwd_w64_kernel, default) | 1 (attn_fwd_pipe_kernel)  -> gpurun_out/attn_sq/form<form>.txt
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
R=$PWD; F=${1:-2}; O=gpurun_out/attn_sq; mkdir -p $O
export EVO_AMD_NO_REBUILD=1 EVO_AMD_ATTN_FORM=$F
cd /tmp && export TMPDIR=/tmp
: > $R/$O/form$F.txt
for shape in "131073 1" "8193 8"; do
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/$O/p1 -o a -- python $R/tools/profile_attn.py $shape > $R/$O/p1.log 2>&1; echo "pass1 rc=$?"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/$O/p2 -o b -- python $R/tools/profile_attn.py $shape > $R/$O/p2.log 2>&1; echo "pass2 rc=$?"
echo "# shape T B = $shape, EVO_AMD_ATTN_FORM=$F" >> $R/$O/form$F.txt
(cd $R; for p in p1 p2; do python tools/summarize_prof.py pmc $O/$p | grep -E "^kernel|attn_fwd" >> $O/form$F.txt; rm -rf $O/$p; done)
done
cut -c1-170 $R/$O/form$F.txt
