#!/usr/bin/env python
"""Per-segment shader clocks of a trip of attn_fwd_w64_kernel from a -DW_PROFILE=1 build (every wave accumulates readcyclecounter deltas:
rescale path | phase 1 (QK^T + exp stream) | between (row sums, masks) | phase 2 (P.V + side stream) | waits (lgkmcnt / vmcnt) | barrier):
    EVO_AMD_LIBNAME=libevo_wprof.so EVO_AMD_HIPCC_FLAGS="-DW_PROFILE=1" python -m evo_amd._build
    EVO_AMD_LIBNAME=libevo_wprof.so EVO_AMD_NO_REBUILD=1 python tools/attn_phase_profile.py
The build overwrites the first bytes of the output with the counters (a timing build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops
ops = default_ops(); dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
names = ["rescale", "phase 1", "between", "phase 2", "waits", "barrier"]
shapes = ((1, 131073),) if os.environ.get("ATTN_PROFILE_SHAPES") == "131k" else ((1, 131073), (8, 8193))
for (B, T) in shapes:
    H = 32
    qkv = torch.randn(B, T, 3, H, 128, generator=g, device=dev).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    c = ops.attn_q_scale(128)
    for pre in (False, True):
        qx = (q.float() * c).bfloat16() if pre else q
        kw = {"prescaled": True} if pre else {}
        for _ in range(3):
            o = ops.attention(qx, k, v, 0, **kw)
        torch.cuda.synchronize()
        n_wg = ((T + 255) // 256) * B * H
        rec = o.reshape(-1)[: n_wg * 4 * 16].view(torch.float32).view(n_wg, 4, 8).cpu().double()
        trips = rec[:, :, 6]
        long_ = trips[:, 0] >= trips[:, 0].max() * 0.5                       # workgroups that walk at least half of the longest key range
        per = rec[long_][:, :, :6] / trips[long_][:, :, None]
        med = per.median(dim=0).values                                      # [wave, segment]
        print(f"---- B={B} T={T} {'PRE' if pre else 'plain'}: {int(long_.sum())} long workgroups, trips {trips[long_].min().item():.0f}..{trips[long_].max().item():.0f}; "
              f"cycles per trip (median over workgroups), waves 0..3:")
        for i, nm in enumerate(names):
            print(f"      {nm:8s} " + "  ".join(f"{med[w, i].item():8.0f}" for w in range(4)))
        print(f"      {'total':8s} " + "  ".join(f"{med[w].sum().item():8.0f}" for w in range(4)), flush=True)
