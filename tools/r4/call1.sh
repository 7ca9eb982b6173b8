#!/bin/bash
# round 4, GPU call 1: the new model-level parity tests (configs[3], configs[4], pool vs oracle, the 64-sequence score distribution)
# on the default library, then the -DHM_XLO=0 build of the Hyena kernel through the checks its decision rule names.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_pool.py "tests/test_gpu_fulldepth.py::test_score_rel_distribution_64_sequences" \
    -m gpu -q -s -rs > $O/parity_default.log 2>&1; echo "default rc=$?"
grep -E "^\[|passed|failed|Error|error" $O/parity_default.log | cut -c1-400 | tail -40
export EVO_AMD_NO_REBUILD=1 EVO_AMD_LIBNAME=libevo_xlo0.so
timeout 1200 python -m pytest "tests/test_gpu_fulldepth.py::test_score_rel_distribution_64_sequences" \
    "tests/test_gpu_fulldepth.py::test_full_depth_every_block_teacher_forced_vs_fp32_oracle" \
    "tests/test_gpu_fulldepth.py::test_hyena_operator_8x8193_full_width_vs_fft" "tests/test_gpu_fulldepth.py::test_hyena_operator_1x131073_full_width_vs_fft" \
    "tests/test_gpu_fulldepth.py::test_131k_forward_blocks_teacher_forced_vs_fp64" \
    -m gpu -q -s -rs > $O/xlo0.log 2>&1; echo "xlo0 rc=$?"
grep -E "^\[|passed|failed|Error|error" $O/xlo0.log | cut -c1-400 | tail -40
