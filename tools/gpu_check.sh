#!/bin/bash
# Everything a round is judged on, on one MI355X box:  gpurun --timeout 4200 -- 'bash tools/gpu_check.sh'
#   -m gpu tests, smoke(), bench.py (default flags), and the rocprofv3 kernel-trace summary of the same bench under profiles/.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
R=$PWD
mkdir -p gpurun_out/check
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/check/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/check/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/check/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/check/smoke.log
timeout 900 python bench.py > gpurun_out/check/bench.json 2> gpurun_out/check/bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/check/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/check/prof -o b -- python $R/bench.py --skip-131k --skip-cpu --skip-gen --steps 3 --warmup 1 > $R/gpurun_out/check/prof_bench.log 2>&1
cd $R && python tools/summarize_prof.py stats gpurun_out/check/prof > gpurun_out/check/bench_kernel_stats.txt && rm -rf gpurun_out/check/prof
head -14 gpurun_out/check/bench_kernel_stats.txt
# the same for a decode run (BASELINE configs[4] shape: 8,192-nt prompt, greedy): per-kernel times of the generation leg
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/check/profg -o g -- python $R/tools/bench_generate.py --new 128 > $R/gpurun_out/check/prof_gen.log 2>&1
cd $R && python tools/summarize_prof.py stats gpurun_out/check/profg > gpurun_out/check/decode_kernel_stats.txt && rm -rf gpurun_out/check/profg
tail -1 gpurun_out/check/prof_gen.log; head -12 gpurun_out/check/decode_kernel_stats.txt
