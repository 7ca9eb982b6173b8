#!/usr/bin/env python
"""GB/s of the skinny dense layer (evo_linear_small_m_bf16) vs torch/hipBLASLt for the decode shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops  # noqa: E402

ops = default_ops()
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
tag = os.environ.get("EVO_AMD_LIBNAME", "default")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M in [int(x) for x in os.environ.get("GEMV_MS", "1,2,4,5,8,12,16,32,48,64").split(",")]:
    tot_mine = tot_torch = 0.0
    line = []
    for name, N, K in (("proj", 12288, 4096), ("out", 4096, 4096), ("l1l2", 22016, 4096), ("l3", 4096, 11008)):
        # rotate over several weight copies so the 256 MB Infinity Cache cannot hold the operand
        ws = [(torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16() for _ in range(6)]
        x = torch.randn(M, K, generator=g, device=dev).bfloat16()
        it = [0]

        def mine():
            it[0] = (it[0] + 1) % len(ws)
            return ops._linear_small_m(x, ws[it[0]], None, None)

        def ref():
            it[0] = (it[0] + 1) % len(ws)
            return torch.mm(x, ws[it[0]].t())

        a, b = timeit(mine), timeit(ref)
        tot_mine += a
        tot_torch += b
        line.append(f"{name} {N * K * 2 / a / 1e6:5.0f}/{N * K * 2 / b / 1e6:5.0f}")
        del ws
    print(f"[{tag}] M={M}: GB/s mine/torch: " + "  ".join(line) + f" | per block {tot_mine * 1e3:.0f} vs {tot_torch * 1e3:.0f} us")
