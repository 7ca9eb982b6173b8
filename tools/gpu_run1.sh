cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
nproc; free -g | head -2
timeout 900 python -m pytest tests/test_gpu_fulldepth.py -x -q -s > gpurun_out/r2a/fulldepth.log 2>&1; echo "fulldepth rc=$?"
tail -40 gpurun_out/r2a/fulldepth.log
python bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/r2a/bench_gpus2.log 2>&1; echo "gpus2 rc=$?"; cat gpurun_out/r2a/bench_gpus2.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2a/bench.log 2> gpurun_out/r2a/bench.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/r2a/bench.log; tail -5 gpurun_out/r2a/bench.err
