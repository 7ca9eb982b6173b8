#!/bin/bash
# decode: per-launch anatomy from a kernel trace (tools/decode_gaps.py); the pool's 32-slot step by kernel
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
R=$PWD; O=gpurun_out/r6f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/dec -o d -- python $R/tools/bench_generate.py --new 96 > $R/$O/dec.log 2>&1
cd $R && python tools/decode_gaps.py $O/dec > $O/decode_launch_anatomy.txt 2>&1; cat $O/decode_launch_anatomy.txt; rm -rf $O/dec
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/pool -o p -- python $R/tools/bench_generate.py --prompt 1024 --pool 32 --jobs 64 --new 128 > $R/$O/pool.log 2>&1
cd $R && python tools/summarize_prof.py stats $O/pool > $O/pool32_kernel_stats.txt; rm -rf $O/pool; tail -3 $O/pool.log; head -30 $O/pool32_kernel_stats.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/pool8 -o p -- python $R/tools/bench_generate.py --prompt 1024 --pool 8 --jobs 16 --new 128 > $R/$O/pool8.log 2>&1
cd $R && python tools/summarize_prof.py stats $O/pool8 > $O/pool8_kernel_stats.txt; rm -rf $O/pool8; tail -2 $O/pool8.log; head -16 $O/pool8_kernel_stats.txt
