#!/usr/bin/env python
"""The single-pass Hyena operator as the scoring path launches it (csrc/hyena_ct.hip: channel-major z^T in, blocked y out) alone, for
rocprofv3 --pmc passes: 3 launches at 8 x 8,193 x 4096, then 3 at 1 x 131,073 x 4096."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops
from evo_amd.hyena_tables import mfma_operand_table
ops = default_ops(); dev = "cuda:0"; D, H = 4096, 32
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, std=1.0: torch.randn(*s, generator=g, device=dev) * std
fir_w = rn(3 * D, 3, std=0.3).bfloat16(); fir_b = rn(3 * D, std=0.1).bfloat16()
om = 10.0 ** (-5.0 + 4.0 * torch.rand(D, 8, generator=g, device=dev))
mag = 1.0 - om; ang = (torch.rand(D, 8, generator=g, device=dev) * 2 - 1) * math.pi
poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous()
res = (rn(D, 8, 2, std=0.25) * torch.sqrt(om).unsqueeze(-1) * 4).float().contiguous()
dskip = rn(D, std=0.5).bfloat16(); tab = mfma_operand_table(poles, res, dskip)
for (B, T) in (((8, 8193), (1, 131073)) if os.environ.get("HC_SHAPES") is None else eval(os.environ["HC_SHAPES"])):
    Tm, Tp, Mp, r_tail = ops.zt_layout(B, T)
    zt = rn(Mp // 256 + (1 if r_tail else 0), 3 * D, 256).bfloat16()
    split = r_tail == 1 and getattr(ops, "hyena_tail_split", False) and os.environ.get("HC_TAIL_IN_OPERATOR") is None
    for _ in range(3):
        if split:       # the scoring path's launch since round 6: the 512 k main tokens of every row, end state out (the last token: the fused single-token launch)
            ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, want_state=True, poles=poles, y_blk=ops.yblk_empty(B * T, D, dev), main_only=True)
        else:
            ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, y_blk=ops.yblk_empty(B * T, D, dev))
    torch.cuda.synchronize()
print("done")
