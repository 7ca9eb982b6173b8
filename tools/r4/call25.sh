#!/bin/bash
# full GPU suite + smoke + a short bench line on the tail-form state
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/check2; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 620 python -m pytest tests -m gpu -q -s -rs > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -2
grep -E "^\.*\[" $O/gpu_tests.log | sed 's/^\.*//' | cut -c1-1600 > $O/gpu_tests_parity_lines.txt; grep -E "SKIPPED|passed|failed" $O/gpu_tests.log | tail -20 >> $O/gpu_tests_parity_lines.txt
grep -E "^E  |^FAILED" $O/gpu_tests.log | cut -c1-300 | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 200 python bench.py --skip-cpu --skip-gen --skip-131k --skip-sp-predict > $O/bench_short.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/check2/bench_short.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
print("kernels", {k:(v["launches_per_step"], round(v["avg_ms"],4)) for k,v in d["kernels"].items()})
print("ab_reference", {a:b for a,b in d.get("ab_reference",{}).items() if a!="note"})
for k in ("library_gemm_l3","mlp_gate_unfused","hyena_group_major_kernel","hyena_round3_kernel"): print(k, {a:b for a,b in d.get(k,{}).items() if a!="note"})
PY
