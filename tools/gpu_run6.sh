cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "hyena_mfma" > gpurun_out/r2g/mfma.log 2>&1; echo "mfma tests rc=$?"; tail -12 gpurun_out/r2g/mfma.log
for dbg in 0 1 2 4 6 7; do EVO_HM_DBG=$dbg timeout 300 python tools/bench_ops.py --only hyena --reps 10 2>&1 | grep "hyena_mfma" | sed "s/^/dbg=$dbg /" | tee -a gpurun_out/r2g/ablate.log; done
