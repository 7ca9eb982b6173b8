cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
build/valu_rate_probe > gpurun_out/r2c/valu_probe.log 2>&1; cat gpurun_out/r2c/valu_probe.log
python tools/bench_ops.py --only hyena --reps 20 2>&1 | grep "^\[" | tee gpurun_out/r2c/hyena_default.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_unembed or hyena" > gpurun_out/r2c/kernels.log 2>&1; echo "kernels rc=$?"; tail -15 gpurun_out/r2c/kernels.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "fused_tail or out_of_range or scores_vs_oracle or graph_decode" > gpurun_out/r2c/model.log 2>&1; echo "model rc=$?"; tail -15 gpurun_out/r2c/model.log
timeout 1200 python -m pytest tests/test_gpu_fulldepth.py -q -s -k "teacher or prefix or configs0" > gpurun_out/r2c/fulldepth.log 2>&1; echo "fulldepth rc=$?"
grep -n "^\[\|passed\|failed\|Error\|assert" gpurun_out/r2c/fulldepth.log | head -60
timeout 600 python bench.py --steps 5 --warmup 2 --skip-131k --skip-gen --skip-cpu > gpurun_out/r2c/bench.log 2> gpurun_out/r2c/bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r2c/bench.log
