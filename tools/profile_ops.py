#!/usr/bin/env python
"""Runs the hand-written kernels alone at the bench shapes, for rocprofv3 (kernel-trace/stats or one --pmc pass).

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ops -o ops -- python tools/profile_ops.py
    rocprofv3 --pmc FETCH_SIZE  -d gpurun_out/pmc_fetch -o ops -- python tools/profile_ops.py --reps 2
    rocprofv3 --pmc WRITE_SIZE  -d gpurun_out/pmc_write -o ops -- python tools/profile_ops.py --reps 2
Random (never zero) inputs; shapes: Hyena / RMSNorm / GELU-gate at B=1, T=131,073, D=4096 (BASELINE configs[2])
and at B=8, T=8,193 (configs[1]); attention at B=1, H=32, T=16,385 (one 131k/8 shard length) unless --attn-T; the
MFMA dense layer at M = 65,536 (Wqkv, out_proj) and the weight-streaming dense layer at M = 1, 8, 16.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--attn-T", type=int, default=16385)
    ap.add_argument("--seg-len", type=int, default=0)
    ap.add_argument("--only", default="", help="hyena: just the Hyena operator (both shapes)")
    args = ap.parse_args()
    from evo_amd.ops import default_ops
    ops = default_ops()
    dev = "cuda:0"
    D, H, I = 4096, 32, 10928
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, std=1.0: (torch.randn(*s, generator=g, device=dev) * std)   # noqa: E731
    fir_w = rn(3 * D, 3, std=0.3).bfloat16()
    fir_b = rn(3 * D, std=0.1).bfloat16()
    u = torch.rand(D, 8, generator=g, device=dev)
    mag = 1.0 - 10.0 ** (-5.0 + 4.0 * u)
    ang = (torch.rand(D, 8, generator=g, device=dev) * 2 - 1) * math.pi
    poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous()
    res = rn(D, 8, 2, std=0.25).float().contiguous()
    dskip = rn(D, std=0.5).bfloat16()
    scale = (1 + rn(D, std=0.02)).bfloat16()
    bias = rn(D, std=0.02).bfloat16()
    for (B, T) in ((1, 131073), (8, 8193)):
        z = rn(B, T, 3 * D).bfloat16()
        x = rn(B * T, D).bfloat16()
        gg = rn(B * T, 2 * I).bfloat16()
        for _ in range(args.reps):
            ops.hyena_prefill(z, fir_w, fir_b, poles, res, dskip, H, want_state=True, seg_len=args.seg_len or None)
            if args.only == "hyena":
                continue
            ops.rmsnorm(x, None, scale, 1e-6)
            ops.rmsnorm(x, bias, scale, 1e-6)
            ops.gelu_gate(gg)
        del z, x, gg
        torch.cuda.synchronize()
    if args.only == "hyena":
        print("profile_ops done")
        return
    T = args.attn_T
    qkv = rn(1, T, 3, H, 128).bfloat16()
    cos = torch.rand(T, 64, generator=g, device=dev)
    sin = torch.rand(T, 64, generator=g, device=dev)
    for _ in range(max(1, args.reps // 2)):
        ops.rope_(qkv, cos, sin)
        ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0)
    # dense layers written here: the MFMA tile kernel at the Wqkv / out_proj shapes, the weight-streaming forms at M = 1, 8, 16
    M = 8 * 8192
    xa = rn(M, D).bfloat16()
    wq = rn(3 * D, D, std=1 / 64).bfloat16()
    wo = rn(D, D, std=1 / 64).bfloat16()
    bq = rn(3 * D, std=0.02).bfloat16()
    r = rn(M, D).bfloat16()
    for _ in range(max(1, args.reps // 2)):
        ops.linear_mfma(xa, wq, bq)
        ops.linear_mfma(xa, wo, None, r)
        for m in (1, 8, 16):
            ops._linear_small_m(xa[:m], wq, bq, None)
    torch.cuda.synchronize()
    print("profile_ops done")


if __name__ == "__main__":
    main()
