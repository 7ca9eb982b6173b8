#!/usr/bin/env python
"""Does padding the MLP inner size (10928) help hipBLASLt?  Times l3 (K = inner) and l1l2 (N = 2*inner)."""
import torch
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M = 65544


def t(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for I in (10928, 10944, 11008, 11264):
    x = torch.randn(M, I, generator=g, device=dev).bfloat16()
    w = (torch.randn(4096, I, generator=g, device=dev) * 0.02).bfloat16()
    r = torch.randn(M, 4096, generator=g, device=dev).bfloat16()
    ms3 = t(lambda: r.addmm_(x, w.t()))
    del x, w, r
    x = torch.randn(M, 4096, generator=g, device=dev).bfloat16()
    w = (torch.randn(2 * I, 4096, generator=g, device=dev) * 0.02).bfloat16()
    ms12 = t(lambda: torch.mm(x, w.t()))
    fl3, fl12 = 2.0 * M * 4096 * 10928, 2.0 * M * 4096 * 2 * 10928          # USEFUL flops
    print(f"[pad] inner={I}: l3 {ms3:.3f} ms ({fl3 / ms3 / 1e9:.0f} useful TF/s)  l1l2 {ms12:.3f} ms ({fl12 / ms12 / 1e9:.0f} useful TF/s)  sum {ms3 + ms12:.3f}")
    del x, w
