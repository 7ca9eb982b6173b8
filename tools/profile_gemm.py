#!/usr/bin/env python
"""The MFMA dense-layer kernel alone (for rocprofv3 --pmc passes): Wqkv shape, M = 16,392."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16392
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ops = default_ops()
g = torch.Generator(device="cuda:0").manual_seed(0)
x = torch.randn(M, K, generator=g, device="cuda:0").bfloat16()
w = (torch.randn(12288, K, generator=g, device="cuda:0") / 64).bfloat16()
for _ in range(3):
    ops.linear_mfma(x, w)
torch.cuda.synchronize()
print("done")
