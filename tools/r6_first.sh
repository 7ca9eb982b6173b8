#!/bin/bash
# round 6, first GPU call: pinned contractive gains, the new parity tests, what the box exposes for clock / power, baseline bench
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/r6a; mkdir -p $O
timeout 600 python tools/dump_contractive_gains.py $O/contractive_gains.json > $O/gains.log 2>&1; echo "gains rc=$?"
cp $O/contractive_gains.json evo_amd/configs/contractive_gains.json
timeout 1500 python -m pytest tests/test_gpu_parity_r6.py -m gpu -q -s -x --durations=10 > $O/parity_r6.log 2>&1; echo "parity_r6 rc=$?"; grep -E "passed|failed" $O/parity_r6.log | tail -2
timeout 900 python -m pytest tests/test_gpu_fulldepth.py -m gpu -q -s -k "paired or contractive" --durations=10 > $O/paired.log 2>&1; echo "paired rc=$?"; grep -E "passed|failed" $O/paired.log | tail -2
(ls -la /sys/class/drm/; for d in /sys/class/drm/card*/device; do echo "== $d"; ls $d | tr '\n' ' '; echo; for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input freq1_label freq2_label temp1_input; do [ -e $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"; done; done; for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk gpu_busy_percent; do [ -e $d/$f ] && (echo "$f:"; cat $d/$f 2>&1 | head -12); done; done) > $O/sysfs.txt 2>&1
(which rocm-smi amd-smi; python -c "import amdsmi; print('amdsmi ok', amdsmi.__file__)"; timeout 30 rocm-smi --showclocks --showpower --showmaxpower --json; timeout 30 amd-smi metric --json 2>&1 | head -150) > $O/smi.txt 2>&1
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json
