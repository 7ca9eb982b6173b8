cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x > gpurun_out/r2z/gemmr_tests.log 2>&1; echo "gemmr tests rc=$?"; tail -5 gpurun_out/r2z/gemmr_tests.log
timeout 600 python tools/bench_gemm.py 2>&1 | grep "TF/s" | tee gpurun_out/r2z/gemmr.log
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r2z/p -o g -- python tools/profile_gemm.py 65544 > gpurun_out/r2z/p.log 2>&1
python tools/summarize_prof.py pmc gpurun_out/r2z/p | grep -i "gemmr" | tee -a gpurun_out/r2z/pmc.txt
rm -rf gpurun_out/r2z/p
