#!/bin/bash
# bench.py with the ab_reference leg (default routing timed like the A/B legs, in front of and behind them)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4w; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 900 python bench.py --skip-cpu --skip-gen --skip-131k --skip-sp-predict > $O/bench.json 2> $O/bench.err; echo "bench.py rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4w/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
print("ab_reference", {a:b for a,b in d.get("ab_reference",{}).items() if a!="note"})
for k in ("library_gemm_l3","mlp_gate_unfused","hyena_group_major_kernel","hyena_round3_kernel"): print(k, {a:b for a,b in d.get(k,{}).items() if a!="note"})
PY
