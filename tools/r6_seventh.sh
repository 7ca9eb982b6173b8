#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
R=$PWD; O=gpurun_out/r6g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_gemv.py -m gpu -q -x -k "attention or rope or small_m or gate" > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?"; tail -4 $O/kernel_tests.log
timeout 600 python tools/attn_bench.py > $O/attn_bench.txt 2>&1; tail -8 $O/attn_bench.txt
ATTN_BENCH_SCALE=2.5 timeout 600 python tools/attn_bench.py > $O/attn_bench_x25.txt 2>&1; tail -8 $O/attn_bench_x25.txt
timeout 900 python -m pytest tests/test_gpu_fulldepth.py -m gpu -q -s -k "attention_h32" > $O/attn_full.log 2>&1; echo "attn full rc=$?"; grep -E "^\.?\[attention|passed|failed" $O/attn_full.log | cut -c1-300 | tail -6
timeout 600 python tools/bench_generate.py --prompt 1024 --pool 32 --jobs 64 --new 128 > $O/pool32.log 2>&1; tail -2 $O/pool32.log
timeout 600 python tools/bench_generate.py --prompt 1024 --pool 8 --jobs 16 --new 128 > $O/pool8.log 2>&1; tail -2 $O/pool8.log
timeout 900 python -m pytest tests/test_gpu_pool.py tests/test_gpu_model.py tests/test_gpu_parity_r4.py -m gpu -q -x > $O/tests_c.log 2>&1; echo "tests_c rc=$?"; tail -3 $O/tests_c.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/dec -o d -- python $R/tools/bench_generate.py --new 96 > $R/$O/dec.log 2>&1
cd $R && python tools/decode_gaps.py $O/dec > $O/decode_launch_anatomy.txt 2>&1; cat $O/decode_launch_anatomy.txt; rm -rf $O/dec
