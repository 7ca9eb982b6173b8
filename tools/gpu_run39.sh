cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2af
export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/r2af/p -o g -- python tools/profile_gemm.py 65544 > gpurun_out/r2af/p.log 2>&1
python tools/summarize_prof.py pmc gpurun_out/r2af/p | grep -i "gemmr" | tee -a gpurun_out/r2af/pmc.txt
python - <<'PY' | tee -a gpurun_out/r2af/pmc.txt
import csv, glob
for f in glob.glob("gpurun_out/r2af/p/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "gemmr" in r["Kernel_Name"]]
    print("durations us:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows])
PY
rm -rf gpurun_out/r2af/p
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r2af/p -o g -- python tools/profile_gemm_lib.py 65544 > gpurun_out/r2af/pl.log 2>&1
python tools/summarize_prof.py pmc gpurun_out/r2af/p | grep -i "Cijk" | tee -a gpurun_out/r2af/pmc.txt
python - <<'PY' | tee -a gpurun_out/r2af/pmc.txt
import csv, glob
for f in glob.glob("gpurun_out/r2af/p/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "Cijk" in r["Kernel_Name"]]
    print("lib durations us:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows])
PY
rm -rf gpurun_out/r2af/p
