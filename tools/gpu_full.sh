cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2bh
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2bh/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r2bh/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2bh/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/r2bh/smoke.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2bh/prof_bench -o b -- python $R/bench.py --skip-131k --skip-cpu --skip-gen --steps 3 --warmup 1 > $R/gpurun_out/r2bh/prof_bench.log 2>&1
cd $R
python tools/summarize_prof.py stats gpurun_out/r2bh/prof_bench > gpurun_out/r2bh/bench_stats.txt
head -16 gpurun_out/r2bh/bench_stats.txt
grep "^{" gpurun_out/r2bh/prof_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
rm -rf gpurun_out/r2bh/prof_bench
