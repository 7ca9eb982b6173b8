cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2bj
export TMPDIR=/tmp
python tools/bench_generate.py --new 128 2>&1 | tail -3 | tee gpurun_out/r2bj/gen.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2bj/p -o g -- python tools/bench_generate.py --new 128 --prompt 1024 > gpurun_out/r2bj/p.log 2>&1
python tools/summarize_prof.py stats gpurun_out/r2bj/p | head -40 | tee gpurun_out/r2bj/dec_stats.txt
python - <<'PY' | tee -a gpurun_out/r2bj/dec_stats.txt
import csv, glob
f = glob.glob("gpurun_out/r2bj/p/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode steps: take the final 2000 kernels, find period by kernel name pattern; report busy vs span
tail = rows[-1500:]
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
gaps = [int(tail[i + 1]["Start_Timestamp"]) - int(tail[i]["End_Timestamp"]) for i in range(len(tail) - 1)]
print(f"last 1500 kernels: span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms ({busy/span:.3f}), mean gap {sum(gaps)/len(gaps)/1e3:.2f} us, max gap {max(gaps)/1e3:.1f} us")
from collections import Counter
c = Counter(r["Kernel_Name"][:60] for r in tail)
for k, v in c.most_common(12):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail if r["Kernel_Name"][:60] == k]
    print(f"{v:5d} x {k:60s} avg {sum(d)/len(d)/1e3:8.2f} us  total {sum(d)/1e6:7.3f} ms")
PY
rm -rf gpurun_out/r2bj/p
