cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2bb
timeout 600 python -m pytest tests/test_gpu_sp_rccl.py -m gpu -x -q > gpurun_out/r2bb/sp_tests.log 2>&1; echo "sp tests rc=$?"; tail -4 gpurun_out/r2bb/sp_tests.log
timeout 900 python bench.py > gpurun_out/r2bb/bench.log 2> gpurun_out/r2bb/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r2bb/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2bb/bench.log") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "all_hand", d.get("all_hand_written_gemm"))
print("roofline", json.dumps(d["roofline"]))
print("131k", d["ctx131k"].get("value"), json.dumps(d["ctx131k"].get("roofline")), d["ctx131k"].get("error"))
print("gen", d["generation"].get("decode_ms_per_token"), d["generation"].get("prefill_ms"), d["generation"].get("error"))
print("cpu", d.get("cpu_baseline"))
PY
