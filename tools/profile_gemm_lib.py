#!/usr/bin/env python
"""hipBLASLt on the Wqkv shape (for a rocprofv3 --pmc pass next to tools/profile_gemm.py)."""
import sys
import torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65544
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
g = torch.Generator(device="cuda:0").manual_seed(0)
x = torch.randn(M, K, generator=g, device="cuda:0").bfloat16()
w = (torch.randn(12288, K, generator=g, device="cuda:0") / 64).bfloat16()
for _ in range(3):
    torch.mm(x, w.t())
torch.cuda.synchronize()
print("done")
