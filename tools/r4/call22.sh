#!/bin/bash
# PMC traffic of hyena_ct_kernel ahead of the final check (so that bench.py's roofline.traffic names a recorded measurement)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
R=$PWD; O=gpurun_out/r4v; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/$O/pmc_zrd -o r -- python $R/tools/profile_hyena_zg.py > $R/$O/pmc_zrd.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $R/$O/pmc_zwr -o w -- python $R/tools/profile_hyena_zg.py > $R/$O/pmc_zwr.log 2>&1
cd $R && (python tools/summarize_prof.py pmc $O/pmc_zrd; python tools/summarize_prof.py pmc $O/pmc_zwr) | grep -E "^kernel|hyena_c[st]" > $O/hyena_ct_pmc_traffic.txt; rm -rf $O/pmc_zrd $O/pmc_zwr
cat $O/hyena_ct_pmc_traffic.txt
