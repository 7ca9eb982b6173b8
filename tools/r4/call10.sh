#!/bin/bash
# round 4, GPU call 10: blocked y end to end (hyena_cs -> evo_linear_xblk_mfma_bf16): kernel / GEMM tests, A/B, parity, bench.py
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4j; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm.py -k "hyena_cs or blocked_hyena" -m gpu -q -x -s > $O/k_tests.log 2>&1; echo "kernel tests rc=$?"
grep -E "passed|failed|Error|assert" $O/k_tests.log | cut -c1-300 | tail -8
timeout 900 python tools/hc_bench.py libevo_mi355x.so libevo_hc_s2.so rm:libevo_mi355x.so old:libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "bench rc=$?"; grep "after-GEMM\|rc=\|reproducible False" $O/hc_bench.log | cut -c1-260
timeout 1200 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_model.py -m gpu -q -s -rs > $O/parity.log 2>&1; echo "parity+model rc=$?"
grep -E "^\.?\[|passed|failed|Error|^E " $O/parity.log | cut -c1-600 | tail -20
timeout 900 python bench.py --skip-cpu --skip-gen --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench.py rc=$?"; tail -c 300 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4j/bench.json"))
print("headline", d["value"], d["ms_per_step"], {k:d["roofline"][k] for k in ("frac","avg_launch_ms")})
print("kernels", {k:(v["launches_per_step"], round(v["avg_ms"],4)) for k,v in d["kernels"].items()})
c=d.get("ctx131k",{})
print("131k", c.get("value"), c.get("ms_per_step"), {k:c["roofline"][k] for k in ("frac","avg_launch_ms")} if c.get("roofline") else None)
print("131k kernels", {k:(v["launches"], round(v["avg_ms"],4)) for k,v in c.get("kernels",{}).items()})
p=d.get("scaling_131k_predicted")
if p: print("predicted", {k:v for k,v in p.items() if k not in ("kernels","what")})
PY
