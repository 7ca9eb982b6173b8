#!/usr/bin/env python
"""What the norm folding costs the dense layers themselves: each launch of the 8 x 8,193 scoring step timed alone on one MI355X with and
without its NF epilogue (row factor in / sums of squares out), same operands, alternating, median of 7 (C ABI through evo_amd.ops)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops, _check, _stream, _ptr  # noqa: E402

ops = default_ops()
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B, T, D, I = 8, 8193, 4096, 11008
M = B * T
Mm = M - M % 256
x = torch.randn(M, D, generator=g, device=dev).bfloat16()
rstd = torch.rand((M + 255) // 256 * 256, generator=g, device=dev) + 0.5


def med(fn, n=7):
    fn(); fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2]


def report(name, plain, nf):
    tp, tn = [], []
    for _ in range(3):
        tp.append(med(plain)); tn.append(med(nf))
    tp, tn = sorted(tp)[1], sorted(tn)[1]
    print(f"[{name}] plain {tp:.4f} ms, norm-folded {tn:.4f} ms ({(tn / tp - 1) * 100:+.2f} %)", flush=True)


lib = ops.lib
# l1 | l2 gated (MODE 1): row factor
w12g = (torch.randn(2 * I, D, generator=g, device=dev) * 0.02).bfloat16()
a = torch.empty(Mm, I, dtype=torch.bfloat16, device=dev)
report("gated l1|l2 65536 x 22016 x 4096 (row factor)",
       lambda: _check(lib.evo_mlp_gate_mfma_bf16(x.data_ptr(), w12g.data_ptr(), a.data_ptr(), Mm, I, D, _stream()), "g"),
       lambda: _check(lib.evo_mlp_gate_mfma_nf_bf16(x.data_ptr(), rstd.data_ptr(), w12g.data_ptr(), a.data_ptr(), Mm, I, D, _stream()), "g"))
del w12g
# l3 (MODE 0 + residual): sums of squares out
w3 = (torch.randn(D, I, generator=g, device=dev) * 0.02).bfloat16()
res = torch.randn(Mm, D, generator=g, device=dev).bfloat16()
ss = torch.empty(D // 128, Mm, dtype=torch.float32, device=dev)
report("l3 65536 x 4096 x 11008 + residual (sums of squares)",
       lambda: _check(lib.evo_linear_mfma_bf16(a.data_ptr(), w3.data_ptr(), None, res.data_ptr(), res.data_ptr(), Mm, D, I, _stream()), "l"),
       lambda: _check(lib.evo_linear_mfma_nf_bf16(a.data_ptr(), w3.data_ptr(), None, res.data_ptr(), res.data_ptr(), None, ss.data_ptr(), Mm, Mm, D, I, _stream()), "l"))
del w3, a
# Hyena output projection on blocked y (+ bias + residual): sums of squares out
wo = (torch.randn(D, D, generator=g, device=dev) * 0.02).bfloat16()
bo = torch.randn(D, generator=g, device=dev).bfloat16()
yb = ops.yblk_empty(Mm, D, dev).normal_()
report("out-proj on blocked y 65536 x 4096 x 4096 + bias + residual (sums of squares)",
       lambda: _check(lib.evo_linear_xblk_mfma_bf16(yb.data_ptr(), wo.data_ptr(), bo.data_ptr(), res.data_ptr(), res.data_ptr(), Mm, D, D, _stream()), "o"),
       lambda: _check(lib.evo_linear_xblk_mfma_nf_bf16(yb.data_ptr(), wo.data_ptr(), bo.data_ptr(), res.data_ptr(), res.data_ptr(), ss.data_ptr(), Mm, Mm, D, D, _stream()), "o"))
# Hyena projection, swapped (MODE 3): plain on a gathered copy vs row factor + stream rows
wp = (torch.randn(3 * D, D, generator=g, device=dev) * 0.02).bfloat16()
bp = torch.randn(3 * D, generator=g, device=dev).bfloat16()
xm = x.view(B, T, D)[:, :8192].reshape(Mm, D).contiguous()
zt = torch.empty(Mm // 256, 3 * D, 256, dtype=torch.bfloat16, device=dev)
report("projection z^T 12288 x 65536 x 4096 + bias (row factor, rows from the stream)",
       lambda: _check(lib.evo_linear_t_mfma_bf16(xm.data_ptr(), wp.data_ptr(), bp.data_ptr(), zt.data_ptr(), Mm, 3 * D, D, _stream()), "t"),
       lambda: _check(lib.evo_linear_t_mfma_nf_bf16(x.data_ptr(), rstd.data_ptr(), wp.data_ptr(), bp.data_ptr(), zt.data_ptr(), Mm, 3 * D, D, M, 8192, 1, _stream()), "t"))
# Wqkv (MODE 0 + bias): row factor
y = torch.empty(Mm, 3 * D, dtype=torch.bfloat16, device=dev)
report("Wqkv 65536 x 12288 x 4096 + bias (row factor)",
       lambda: _check(lib.evo_linear_mfma_bf16(x.data_ptr(), wp.data_ptr(), bp.data_ptr(), None, y.data_ptr(), Mm, 3 * D, D, _stream()), "q"),
       lambda: _check(lib.evo_linear_mfma_nf_bf16(x.data_ptr(), wp.data_ptr(), bp.data_ptr(), None, y.data_ptr(), rstd.data_ptr(), None, 0, Mm, 3 * D, D, _stream()), "q"))
