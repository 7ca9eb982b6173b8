#!/usr/bin/env python
"""Attention kernel alone (for rocprofv3 --pmc passes): python tools/profile_attn.py [T [B]], H=32, causal (default 1 x 16,385)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16385
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ops = default_ops()
g = torch.Generator(device="cuda:0").manual_seed(0)
qkv = torch.randn(B, T, 3, 32, 128, generator=g, device="cuda:0").bfloat16()
for _ in range(3):
    ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0)
torch.cuda.synchronize()
print("done")
