cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
export TMPDIR=/tmp
for v in mi355x rabl1 rabl17; do
  for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
    EVO_AMD_LIBNAME=libevo_$v.so EVO_AMD_NO_REBUILD=1 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/r2y/p_$v -o g -- python tools/profile_gemm.py 65544 > gpurun_out/r2y/p_$v.log 2>&1
    echo "== $v" | tee -a gpurun_out/r2y/pmc.txt
    python tools/summarize_prof.py pmc gpurun_out/r2y/p_$v | grep -i "gemmr" | tee -a gpurun_out/r2y/pmc.txt
    python - <<PY | tee -a gpurun_out/r2y/pmc.txt
import csv, glob
for f in glob.glob("gpurun_out/r2y/p_$v/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "gemmr" in r["Kernel_Name"]]
    print("durations us:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows])
PY
    rm -rf gpurun_out/r2y/p_$v
  done
done
