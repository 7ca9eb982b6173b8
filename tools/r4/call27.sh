#!/bin/bash
# PMC traffic of hyena_ct_kernel with the TAIL form of z^T, 8 x 8,193 only (the bench shape)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
R=$PWD; O=gpurun_out/r4z; mkdir -p $O
export EVO_AMD_NO_REBUILD=1 HC_SHAPES="((8,8193),)"
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/$O/pmc_zrd -o r -- python $R/tools/profile_hyena_zg.py > $R/$O/pmc_zrd.log 2>&1
timeout 100 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $R/$O/pmc_zwr -o w -- python $R/tools/profile_hyena_zg.py > $R/$O/pmc_zwr.log 2>&1
cd $R && (python tools/summarize_prof.py pmc $O/pmc_zrd; python tools/summarize_prof.py pmc $O/pmc_zwr) | grep -E "^kernel|hyena_c[st]" > $O/hyena_ct_pmc_traffic_tail_8k.txt; rm -rf $O/pmc_zrd $O/pmc_zwr
cat $O/hyena_ct_pmc_traffic_tail_8k.txt
