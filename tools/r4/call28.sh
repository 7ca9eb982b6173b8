#!/bin/bash
# bench.py (all legs but the CPU baseline) on the final state
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out/r4zz; mkdir -p $O
export EVO_AMD_NO_REBUILD=1
timeout 165 python bench.py --skip-cpu > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json
