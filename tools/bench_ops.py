#!/usr/bin/env python
"""Per-kernel HIP-event timings of the hand-written kernels at the bench shapes (A/B helper).
    [EVO_AMD_LIBNAME=libevo_variant.so EVO_AMD_NO_REBUILD=1] python tools/bench_ops.py [--seg-len C] [--only hyena|attn]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--seg-len", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--attn-T", type=int, default=16385)
    ap.add_argument("--attn-sweep", action="store_true")
    ap.add_argument("--fft", action="store_true", help="with --only hyena: also time a rocFFT long convolution of the same size")
    args = ap.parse_args()
    from evo_amd.ops import KernelTimer, default_ops
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
    tag = os.environ.get("EVO_AMD_LIBNAME", "default")
    if args.only in ("", "hyena"):
        for (B, T) in ((8, 8193), (1, 131073)):
            z = rn(B, T, 3 * D).bfloat16()
            for _ in range(2):
                ops.hyena_prefill(z, fir_w, fir_b, poles, res, dskip, H, seg_len=args.seg_len or None)
            ops.timer = KernelTimer()
            for _ in range(args.reps):
                ops.hyena_prefill(z, fir_w, fir_b, poles, res, dskip, H, seg_len=args.seg_len or None)
            torch.cuda.synchronize()
            s = ops.timer.summary()
            ops.timer = None
            alg = B * T * D * 8
            tot = sum(v[1] for v in s.values())
            print(f"[{tag}] hyena B={B} T={T} seg={args.seg_len or 'auto'}: " +
                  " ".join(f"{k.replace('hyena_', '')}={v[1]:.3f}ms" for k, v in s.items()) +
                  f" | apply {alg / s['hyena_apply'][1] / 1e6:.0f} GB/s, operator {alg / tot / 1e6:.0f} GB/s")
            try:                                                   # single-pass matrix-core form of the same operator
                from evo_amd.hyena_tables import mfma_operand_table
                tab = mfma_operand_table(poles, res, dskip)
                zt = ops.zt_from_rows(z, B, T)                         # (the layout the projection writes: csrc/hyena_ct.hip)
                yb = ops.yblk_empty(B * T, D, dev)
                for _ in range(2):
                    ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, y_blk=yb)
                ops.timer = KernelTimer()
                for _ in range(args.reps):
                    ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, y_blk=yb)
                torch.cuda.synchronize()
                ms = ops.timer.summary()["hyena_mfma"][1]
                ops.timer = None
                print(f"[{tag}] hyena_ct B={B} T={T}: {ms:.3f}ms | {alg / ms / 1e6:.0f} GB/s = {alg / ms / 1e6 / 8000:.3f} of 8 TB/s")
            except Exception as e:  # noqa: BLE001
                ops.timer = None
                print(f"[{tag}] hyena_ct B={B} T={T}: FAILED {type(e).__name__}: {e}")
            if args.only == "hyena" and args.fft:
                # the north-star's FFT form as a DATA POINT: the long convolution alone (no FIR, no gates, no layout change) by
                # rocFFT through torch.fft in fp32 -- rfft of x1*v [B, D, n], times the filter's spectrum, irfft.  n = 2T rounded
                # up to a power of two, as the reference's fftconv pads.
                n = 1 << (2 * T - 1).bit_length()
                Bc = B if T < 20000 else 1
                x = torch.randn(Bc, D, T, device=dev)
                Hf = torch.fft.rfft(torch.randn(D, T, device=dev), n=n)
                def fftconv():
                    return torch.fft.irfft(torch.fft.rfft(x, n=n) * Hf, n=n)[..., :T]
                for _ in range(2):
                    fftconv()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fftconv()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                print(f"[{tag}] rocFFT fp32 long convolution alone B={Bc} T={T} n={n}: {ms:.3f}ms "
                      f"(whole operator: three launches {tot:.3f}ms)")
                del x, Hf
            del z
    if args.only in ("", "attn"):
        shapes = [(8, 8193), (1, args.attn_T)] + ([(1, 8193), (2, 8193), (4, 8193), (8, 4097), (8, 16385), (2, 32769)] if args.attn_sweep else [])
        for (Bq, T) in shapes:
            qkv = rn(Bq, T, 3, H, 128).bfloat16()
            for _ in range(1):
                ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0)
            ops.timer = KernelTimer()
            for _ in range(max(2, args.reps // 3)):
                ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0)
            torch.cuda.synchronize()
            ms = ops.timer.summary()["attn_fwd"][1]
            ops.timer = None
            fl = Bq * 4 * D * T * T / 2
            print(f"[{tag}] attn B={Bq} T={T}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s")
            del qkv

    if args.only in ("", "elem"):
        M = 8 * 8193
        gg = rn(M, 2 * 11008).bfloat16()
        x = rn(M, D).bfloat16()
        sc = rn(D).bfloat16()
        bi = rn(D).bfloat16()
        for _ in range(2):
            ops.gelu_gate(gg); ops.rmsnorm(x, None, sc, 1e-6); ops.rmsnorm(x, bi, sc, 1e-6)
        ops.timer = KernelTimer()
        for _ in range(args.reps):
            ops.gelu_gate(gg); ops.rmsnorm(x, None, sc, 1e-6); ops.rmsnorm(x, bi, sc, 1e-6)
        torch.cuda.synchronize()
        s = ops.timer.summary()
        ops.timer = None
        byt = {"gelu_gate": M * 11008 * 6, "rmsnorm": M * D * 4, "rmsnorm_bias": M * D * 6}
        print(f"[{tag}] elementwise M={M}: " + " ".join(f"{k}={v[1]:.3f}ms ({byt[k] / v[1] / 1e6:.0f} GB/s)" for k, v in s.items()))


if __name__ == "__main__":
    main()
