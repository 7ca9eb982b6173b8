#!/usr/bin/env python
"""Where a decode step's time goes, MEASURED from a rocprofv3 kernel trace ("per-launch ramp / tail measured, not
estimated").  Input: the directory of `rocprofv3 --kernel-trace --output-format csv -- python tools/bench_generate.py --new N`.

A decode step = the dispatches from one single-token `embed_kernel` to the next.  For every step: sum of kernel durations, sum of the gaps
between consecutive dispatches (end -> next start), span.  Per weight-streaming launch class (by kernel name and position in the block) the
mean duration against the bytes of weights it streams; a least-squares line  duration = t0 + bytes / BW  over the classes gives the fixed
cost per launch (ramp + tail: the part of a launch during which HBM is not saturated) and the steady streaming rate.

    python tools/decode_gaps.py <dir> > profiles/r06_decode_launch_anatomy.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

D, I3, V = 4096, 11008, 512
BYTES = {  # weights a launch streams at batch 1 (bf16)
    "gemv_norm_hyena_kernel": 3 * D * D * 2,      # pre-norm + projections + FIR / modal step
    "gemv_norm_kernel": 3 * D * D * 2,            # pre-norm + Wqkv
    "gemv_gate_kernel": 2 * I3 * D * 2,           # post-norm + l1 | l2 + gate
}


def main(d):
    files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        print("no *kernel_trace.csv under", d)
        return
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    short = lambda n: n.split("(")[0].replace("void ", "").split("<")[0].strip()   # noqa: E731
    # decode steps: runs that start at an embed_kernel and contain no prefill-only kernel (hyena_ct / gemmr / attn_fwd)
    starts = [i for i, r in enumerate(rows) if short(r[2]) == "embed_kernel"]
    steps = []
    for a, b in zip(starts, starts[1:]):
        seg = rows[a:b]
        names = {short(r[2]) for r in seg}
        if names & {"hyena_ct_kernel", "gemmr_bf16_kernel", "attn_fwd_w64_kernel", "hyena_apply_kernel"}:
            continue
        if 100 <= len(seg) <= 400:
            steps.append(seg)
    if len(steps) < 8:
        print(f"only {len(steps)} decode steps found")
        return
    steps = steps[len(steps) // 4:]                               # drop the first quarter (graph warm-up, short KV)
    n = len(steps)
    kd = [sum(e - s for s, e, _ in st) for st in steps]
    gaps = [sum(max(0, st[i + 1][0] - st[i][1]) for i in range(len(st) - 1)) for st in steps]
    span = [st[-1][1] - st[0][0] for st in steps]
    nl = [len(st) for st in steps]
    m = lambda v: sum(v) / len(v)   # noqa: E731
    print(f"# {n} decode steps (batch 1, hipGraph replays) from {os.path.basename(os.path.normpath(d))}")
    print(f"launches per step {m(nl):.1f}; span {m(span) / 1e3:.1f} us; sum of kernel durations {m(kd) / 1e3:.1f} us; sum of gaps between dispatches "
          f"{m(gaps) / 1e3:.1f} us ({m(gaps) / m(nl):.0f} ns per boundary)")
    # per class: name + size class of plain gemv launches (out projection 4096 x 4096 vs l3 4096 x 11008, told apart by duration)
    cls = defaultdict(list)
    for st in steps:
        for s, e, nm in st:
            k = short(nm)
            if k == "gemv_kernel":
                k = "gemv_kernel[l3 4096x11008]" if e - s > 10000 else "gemv_kernel[out 4096x4096]"
            cls[k].append(e - s)
    print(f"{'kernel':44s} {'per step':>9s} {'mean us':>9s} {'total us/step':>14s} {'weight MB':>10s} {'GB/s':>8s}")
    pts = []
    for k, v in sorted(cls.items(), key=lambda kv: -sum(kv[1])):
        per = len(v) / n
        by = BYTES.get(k, {"gemv_kernel[l3 4096x11008]": D * I3 * 2, "gemv_kernel[out 4096x4096]": D * D * 2}.get(k))
        rate = f"{by / m(v):8.0f}" if by else " " * 8
        print(f"{k[:44]:44s} {per:9.1f} {m(v) / 1e3:9.2f} {sum(v) / n / 1e3:14.1f} {(by or 0) / 1e6:10.1f} {rate}")
        if by and per >= 3:
            pts.append((by, m(v), per))
    if len(pts) >= 3:                                              # weighted least squares: dur = t0 + bytes / BW
        sw = sum(p[2] for p in pts)
        mx = sum(p[0] * p[2] for p in pts) / sw
        my = sum(p[1] * p[2] for p in pts) / sw
        sl = sum(p[2] * (p[0] - mx) * (p[1] - my) for p in pts) / sum(p[2] * (p[0] - mx) ** 2 for p in pts)
        t0 = my - sl * mx
        nw = sum(p[2] for p in pts)
        print(f"\nfit over {len(pts)} weight-streaming classes ({nw:.0f} launches per step): duration = {t0 / 1e3:.2f} us + bytes / {1 / sl:.0f} GB/s")
        print(f"  -> fixed cost per launch (ramp + tail) {t0 / 1e3:.2f} us x {nw:.0f} = {t0 * nw / 1e3:.0f} us per step; streaming {12.906e9 * sl / 1e3:.0f} us for 12.9 GB "
              f"at the steady rate; dispatch gaps {m(gaps) / 1e3:.0f} us; everything else (attention, rotary, small kernels) "
              f"{(m(kd) - sum(p[1] * p[2] for p in pts)) / 1e3:.0f} us")


if __name__ == "__main__":
    main(sys.argv[1])
