#!/bin/bash
# round 4, GPU call 4: hyena_cs with the window pieces / stores interleaved with the arithmetic; phase profile; bench.py in the model
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4d; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "hyena_cs" -m gpu -q -x -s > $O/cs_tests.log 2>&1; echo "cs tests rc=$?"
grep -E "passed|failed|Error|assert" $O/cs_tests.log | cut -c1-300 | tail -8
timeout 600 python tools/hc_bench.py libevo_mi355x.so libevo_hc_nw4.so old:libevo_mi355x.so > $O/hc_bench.log 2>&1; echo "bench rc=$?"; grep "after-GEMM\|rc=\|vs modal" $O/hc_bench.log | cut -c1-260
EVO_AMD_LIBNAME=libevo_hcprof.so timeout 300 python tools/hc_stage_profile.py > $O/prof.log 2>&1; echo "prof rc=$?"; cat $O/prof.log | grep -v amdgpu.ids | cut -c1-300
timeout 900 python bench.py --skip-cpu --skip-gen --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench.py rc=$?"; tail -c 300 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4d/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"])
print("kernels", {k:(v["launches_per_step"], round(v["avg_ms"],4)) for k,v in d["kernels"].items()})
c=d.get("ctx131k",{})
print("131k", c.get("value"), c.get("ms_per_step"), c.get("roofline"))
print("131k kernels", {k:(v["launches"], round(v["avg_ms"],4)) for k,v in c.get("kernels",{}).items()})
p=d.get("scaling_131k_predicted")
if p: print("predicted", {k:v for k,v in p.items() if k!="kernels"})
PY
