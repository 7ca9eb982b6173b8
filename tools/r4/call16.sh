#!/bin/bash
# first hardware run of csrc/hyena_ct.hip (channel-major z^T) + the swapped-operand projection + rmsnorm_rows
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -rs -k "hyena_ct or rmsnorm_rows" > $O/tests.log 2>&1; RC=$?; echo "tests rc=$RC"
grep -E "^\.*\[|passed|failed|^E  |^FAILED" $O/tests.log | cut -c1-300 | tail -60
HM_ROUNDS=4 timeout 600 python tools/hc_bench.py ct:libevo_mi355x.so libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "hc_bench rc=$?"; tail -14 $O/hc_bench.log | cut -c1-260
if [ $RC -eq 0 ]; then
timeout 900 python bench.py --skip-cpu --skip-gen --skip-131k --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench.py rc=$?"; tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4p/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
print("kernels", {k:(v["launches_per_step"], round(v["avg_ms"],4)) for k,v in d["kernels"].items()})
for k in ("library_gemm_l3","mlp_gate_unfused","hyena_group_major_kernel","hyena_round3_kernel"): print(k, {a:b for a,b in d.get(k,{}).items() if a!="note"})
PY
fi
