cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/r2d/pmc_sq -o hy -- python tools/profile_ops.py --only hyena --reps 3 > gpurun_out/r2d/pmc_sq.log 2>&1; echo "pmc rc=$?"
python tools/summarize_prof.py pmc gpurun_out/r2d/pmc_sq | grep -i "hyena" | tee gpurun_out/r2d/pmc_sq_summary.txt
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r2d/pmc_sq/**/*kernel_trace.csv',recursive=True):
    rows=list(csv.DictReader(open(f)))
    from collections import defaultdict
    d=defaultdict(list)
    for r in rows:
        d[r['Kernel_Name'][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,v in d.items():
        if 'hyena' in k: print('DUR',k,len(v),' '.join('%.1f'%x for x in v))
PY
find gpurun_out/r2d/pmc_sq -name "*.csv" -size +2000k -delete
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "padding_mask or hyena_prefill_matches" > gpurun_out/r2d/kernels.log 2>&1; echo "kernels rc=$?"; tail -5 gpurun_out/r2d/kernels.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "padding_mask or out_of_range or fused_tail" > gpurun_out/r2d/model.log 2>&1; echo "model rc=$?"; tail -8 gpurun_out/r2d/model.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "reproducible" > gpurun_out/r2d/fullsize.log 2>&1; echo "fullsize rc=$?"; tail -8 gpurun_out/r2d/fullsize.log
python tools/bench_ops.py --only attn --reps 9 2>&1 | grep "^\["
