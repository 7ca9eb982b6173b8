#!/usr/bin/env python
"""Per-stage cycle breakdown of hyena_mfma_kernel from a -DHM_PROFILE=1 build (wave 0 of workgroup 0 accumulates
shader-clock deltas per stage and writes them over the first words of y):
    EVO_AMD_LIBNAME=libevo_hmprof.so EVO_AMD_HIPCC_FLAGS=-DHM_PROFILE=1 python -c "from evo_amd import _build; _build.build(force=True)"
    EVO_AMD_LIBNAME=libevo_hmprof.so EVO_AMD_NO_REBUILD=1 python tools/hm_stage_profile.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops
from evo_amd.hyena_tables import mfma_operand_table
ops = default_ops(); dev = "cuda:0"; D, H = 4096, 32
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, std=1.0: torch.randn(*s, generator=g, device=dev) * std
fir_w = rn(3 * D, 3, std=0.3).bfloat16(); fir_b = rn(3 * D, std=0.1).bfloat16()
mag = 1.0 - 10.0 ** (-5.0 + 4.0 * torch.rand(D, 8, generator=g, device=dev)); ang = (torch.rand(D, 8, generator=g, device=dev) * 2 - 1) * math.pi
poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous(); res = rn(D, 8, 2, std=0.25).float().contiguous()
dskip = rn(D, std=0.5).bfloat16(); tab = mfma_operand_table(poles, res, dskip)
for (B, T) in ((8, 8193), (1, 131073)):
    z = rn(B, T, 3 * D).bfloat16()
    for _ in range(3):
        y = ops.hyena_mfma_prefill(z, fir_w, fir_b, dskip, tab, H)
    torch.cuda.synchronize()
    t = y.view(-1)[:12].view(torch.float32)[:6].tolist()
    n = t[5]
    names = ["wait DMA", "barriers", "DMA issue + stage 1", "stage 2", "stage 3"]
    print(f"B={B} T={T}: steps={n:.0f}; cycles per step (100 MHz s_memtime ticks x ?): " + ", ".join(f"{nm}={v / n:.0f}" for nm, v in zip(names, t[:5])) + f" | total {sum(t[:5]) / n:.0f}")
