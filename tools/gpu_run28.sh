cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
export TMPDIR=/tmp
i=0
for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INSTS_WAVE32_LDS"; do
  i=$((i+1))
  EVO_GEMM_WAVES=4 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/r2t/own_$i -o g -- python tools/profile_gemm.py 65544 > gpurun_out/r2t/own_$i.log 2>&1
  python tools/summarize_prof.py pmc gpurun_out/r2t/own_$i | grep -i "gemm4\|counter" | tee -a gpurun_out/r2t/pmc_own.txt
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/r2t/lib_$i -o g -- python tools/profile_gemm_lib.py 65544 > gpurun_out/r2t/lib_$i.log 2>&1
  python tools/summarize_prof.py pmc gpurun_out/r2t/lib_$i | grep -i "Cijk\|counter" | tee -a gpurun_out/r2t/pmc_lib.txt
  rm -rf gpurun_out/r2t/own_$i gpurun_out/r2t/lib_$i
done
