#!/usr/bin/env python
"""Which attention kernel wins INSIDE a step?  The two kernels (csrc/attn_w64.hip: 4 waves x 64 rows, one per SIMD; csrc/attn.hip
attn_fwd_pipe_kernel: 8 waves x 32 rows) timed on one MI355X (a) back to back on their own and (b) right behind a burst of the
projection's dense layer (the state of the chip a scoring step leaves the attention kernel in: the governor's clock follows the dense
layers' power draw, and a 5 ms kernel runs at whatever clock it inherits), at the shapes the routing decides between.
Prints one line per shape and kernel; every GPU call goes through the C ABI (evo_amd.ops)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops  # noqa: E402

dev = "cuda:0"
ops = default_ops()
g = torch.Generator(device=dev).manual_seed(0)
xw = torch.randn(65536, 4096, generator=g, device=dev).bfloat16()
ww = (torch.randn(12288, 4096, generator=g, device=dev) * 0.02).bfloat16()


def burst(n):
    for _ in range(n):
        ops.linear(xw, ww, None, mfma=True)


def timed(fn, pre, reps=6):
    ts = []
    for _ in range(reps):
        pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def model_qkv(B, T):
    """The q / k / v the scoring step itself hands to the first attention block (synthetic 7B weights, synthetic ACGT rows)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    m = bench.build_model("evo-1-8k-base" if T <= 8193 else "evo-1-131k-base", dev)
    ids = bench.acgt_ids(B, T - 1, 1234, dev)
    got = {}
    real = m.ops.attention

    class Done(Exception):
        pass

    def spy(q, k, v, off):
        got["qkv"] = (q.clone(), k.clone(), v.clone())
        raise Done()
    m.ops.attention = spy
    try:
        with torch.no_grad():
            m(ids)
    except Done:
        pass
    m.ops.attention = real
    del m
    torch.cuda.empty_cache()
    return got["qkv"]


args = [a for a in sys.argv[1:] if not a.startswith("--")]
from_model = "--model" in sys.argv
shapes = [(8, 8193), (1, 8193), (4, 16385), (2, 32769), (1, 65537), (1, 131073)]
if args:
    shapes = [tuple(int(x) for x in s.split("x")) for s in args]
for (B, T) in shapes:
    if from_model:
        q, k, v = model_qkv(B, T)
        sc = (q[0, -256:, 0].float() @ k[0, :, 0].float().t()) / 128 ** 0.5 * 1.4427
        print(f"[B={B} T={T}] q / k / v of the model's block 8: |q| rms {float(q.float().pow(2).mean().sqrt()):.3f} |k| rms {float(k.float().pow(2).mean().sqrt()):.3f}; "
              f"scores (log2 units) of head 0, last 256 rows: std {float(sc.std()):.2f} max {float(sc.max()):.1f}", flush=True)
    else:
        qkv = torch.randn(B, T, 3, 32, 128, generator=g, device=dev).bfloat16()
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    fl = B * 4 * 4096 * T * T / 2
    out = {}
    for name, flag in (("w64", True), ("pipe", False)):
        ops.attn_w64 = flag
        f = lambda: ops.attention(q, k, v, 0)
        f(); f()
        alone = timed(f, lambda: None)
        hot = timed(f, lambda: burst(10))
        out[name] = (alone, hot)
        print(f"[B={B} T={T}] {name:5s} alone median {alone[0]:8.3f} ms (min {alone[1]:8.3f}) = {fl / alone[0] / 1e9:6.0f} TFLOP/s | behind 10 dense-layer launches "
              f"median {hot[0]:8.3f} ms (min {hot[1]:8.3f}) = {fl / hot[0] / 1e9:6.0f} TFLOP/s", flush=True)
    print(f"[B={B} T={T}] w64 / pipe: alone {out['w64'][0][0] / out['pipe'][0][0]:.3f}, in-step {out['w64'][1][0] / out['pipe'][1][0]:.3f}", flush=True)
    q = k = v = qkv = None
ops.attn_w64 = True
