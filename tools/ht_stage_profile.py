#!/usr/bin/env python
"""Per-phase clock breakdown of hyena_ct_kernel from a -DHT_PROFILE=1 build (every wave accumulates shader-clock deltas per phase and
writes 16 floats at y + 64 B * (8 * workgroup + wave); a timing build: it overwrites y):
    EVO_AMD_LIBNAME=libevo_htprof.so EVO_AMD_HIPCC_FLAGS="-DHT_PROFILE=1" python -m evo_amd._build
    EVO_AMD_LIBNAME=libevo_htprof.so EVO_AMD_NO_REBUILD=1 python tools/ht_stage_profile.py
Conditions: the last of 12 launches that each follow the projection's dense layer (what the kernel meets inside a scoring step) and
the last of 12 back-to-back launches."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops
from evo_amd.hyena_tables import mfma_operand_table
ops = default_ops(); dev = "cuda:0"; D, H = 4096, 32
type(ops).ZT_ALIGN = int(os.environ.get("HC_ZT_ALIGN", type(ops).ZT_ALIGN))
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, std=1.0: torch.randn(*s, generator=g, device=dev) * std
fir_w = rn(3 * D, 3, std=0.3).bfloat16(); fir_b = rn(3 * D, std=0.1).bfloat16()
mag = 1.0 - 10.0 ** (-5.0 + 4.0 * torch.rand(D, 8, generator=g, device=dev)); ang = (torch.rand(D, 8, generator=g, device=dev) * 2 - 1) * math.pi
poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous(); res = rn(D, 8, 2, std=0.25).float().contiguous()
dskip = rn(D, std=0.5).bfloat16(); tab = mfma_operand_table(poles, res, dskip)
NW = 8
names = ["wait loads", "barrier", "staged out + history", "FIR", "MFMA + scan + stores", "gate + stage"]
cols = [0, 1, 8, 2, 3, 4]
st = torch.cuda.current_stream().cuda_stream
for (B, T) in ((8, 8193), (1, 131073)):
    Tm, Tp, Mp, r_tail = ops.zt_layout(B, T)
    zt = rn(Mp // 256 + (1 if r_tail else 0), 3 * D, 256).bfloat16()
    xp = rn(Mp, D).bfloat16(); wgt = rn(3 * D, D, std=0.02).bfloat16()
    for mode in ("after-GEMM x12", "back-to-back x12"):
        for _ in range(12):
            if mode.startswith("after"):
                ops.lib.evo_linear_t_mfma_bf16(xp.data_ptr(), wgt.data_ptr(), None, zt.data_ptr(), Mp, 3 * D, D, st)
            y = ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, y_blk=ops.yblk_empty(B * T, D, dev))
        torch.cuda.synchronize()
        rec = y.view(-1)[:256 * NW * 32].view(torch.float32).view(256, NW, 16).cpu()
        n = rec[0, 0, 5].item()
        tot_us = rec[:, 0, 6] / 100.0
        ghz = rec[:, 0, 7] / (rec[:, 0, 6] * 10.0)
        med = rec.median(dim=0).values
        print(f"---- B={B} T={T} Tp={Tp} {mode}: {n:.0f} tiles per workgroup; workgroup duration us min/median/max {tot_us.min():.1f} / {tot_us.median():.1f} / "
              f"{tot_us.max():.1f}; clock GHz median {ghz.median():.2f}; per tile {tot_us.median() / n:.2f} us")
        for w in (0, NW // 2, NW - 1):
            print(f"     wave {w}, clocks per tile, median over workgroups: " + ", ".join(f"{nm}={med[w, c].item() / n:.0f}" for c, nm in zip(cols, names))
                  + f" | sum {sum(med[w, c].item() for c in cols) / n:.0f}")
    del xp, wgt, zt
