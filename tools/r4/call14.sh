#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_model.py "tests/test_gpu_fulldepth.py::test_full_depth_every_block_teacher_forced_vs_fp32_oracle" "tests/test_gpu_fulldepth.py::test_prefix_of_bench_batch_vs_fp32_oracle" "tests/test_gpu_fulldepth.py::test_score_rel_distribution_64_sequences" -m gpu -q -s -rs > $O/tests.log 2>&1; echo "tests rc=$?"
grep -E "^\.*\[|passed|failed|^E  " $O/tests.log | cut -c1-420 | tail -24
timeout 900 python bench.py --skip-cpu --skip-gen --skip-131k --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench.py rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4n/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
print("kernels", {k:(v["launches_per_step"], round(v["avg_ms"],4)) for k,v in d["kernels"].items()})
for k in ("library_gemm_l3","mlp_gate_unfused","hyena_round3_kernel"): print(k, {a:b for a,b in d.get(k,{}).items() if a!="note"})
PY
