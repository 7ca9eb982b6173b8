#!/usr/bin/env python
"""Times the five dense-layer GEMM shapes of one block at the bench token counts through the same torch calls
the engine uses (hipBLASLt).  Run with PYTORCH_TUNABLEOP_ENABLED=1 to A/B TunableOp's pick."""
import os
import sys
import time

import torch

M_LIST = [int(x) for x in (sys.argv[1:] or ["65544", "131073"])]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
tag = "tunable" if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1" else "default"
for M in M_LIST:
    tot_ms, tot_fl = 0.0, 0.0
    for name, N, K, mode in (("proj/qkv", 12288, 4096, "bias"), ("out", 4096, 4096, "res"),
                             ("l1l2", 21856, 4096, "plain"), ("l3", 4096, 10928, "res"), ("unembed", 512, 4096, "plain")):
        x = torch.randn(M, K, generator=g, device=dev).bfloat16()
        w = (torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16()
        b = torch.randn(N, generator=g, device=dev).bfloat16()
        r = torch.randn(M, N, generator=g, device=dev).bfloat16()
        fn = {"bias": lambda: torch.addmm(b, x, w.t()), "res": lambda: r.addmm_(x, w.t()), "plain": lambda: torch.mm(x, w.t())}[mode]
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 2.0 * M * N * K
        per_block = {"proj/qkv": 1, "out": 1, "l1l2": 1, "l3": 1, "unembed": 1 / 32}[name]
        tot_ms += ms * per_block
        tot_fl += fl * per_block
        print(f"[{tag}] M={M} {name:9s} N={N:5d} K={K:5d} {mode:5s}: {ms:8.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s")
        del x, w, b, r
    print(f"[{tag}] M={M} per-block GEMM time {tot_ms:.3f} ms -> x32 = {tot_ms * 32:.1f} ms, {tot_fl / tot_ms / 1e9:.0f} TFLOP/s")
