#!/usr/bin/env python
"""A/B of csrc/hyena_ct.hip builds IN ONE PROCESS: every library named on the command line (files in evo_amd/_lib/) is loaded through its own
HipOps and timed alternately, each launch behind the projection's dense layer that writes its z^T (what the kernel meets inside a scoring step).
    python tools/hc_ab.py libevo_mi355x.so libevo_htold.so"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd import _build, ops as ops_mod
from evo_amd.hyena_tables import mfma_operand_table
dev = "cuda:0"; D, H = 4096, 32
libs = []
for name in sys.argv[1:]:
    os.environ["EVO_AMD_NO_REBUILD"] = "1"
    _build.LIBNAME = name                                         # load_library() loads _build.lib_path() ...
    ops_mod._LIB = None                                           # ... once per process unless its cache is cleared
    libs.append((name, ops_mod.HipOps()))
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, std=1.0: torch.randn(*s, generator=g, device=dev) * std
fir_w = rn(3 * D, 3, std=0.3).bfloat16(); fir_b = rn(3 * D, std=0.1).bfloat16()
om = 10.0 ** (-5.0 + 4.0 * torch.rand(D, 8, generator=g, device=dev))
mag = 1.0 - om; ang = (torch.rand(D, 8, generator=g, device=dev) * 2 - 1) * math.pi
poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous()
res = (rn(D, 8, 2, std=0.25) * torch.sqrt(om).unsqueeze(-1) * 4).float().contiguous()
dskip = rn(D, std=0.5).bfloat16(); tab = mfma_operand_table(poles, res, dskip)
for (B, T) in ((8, 8193), (1, 131073)):
    ops0 = libs[0][1]
    Tm, Tp, Mp, r_tail = ops0.zt_layout(B, T)
    x = rn(B * T, D).bfloat16(); w = rn(3 * D, D, std=0.02).bfloat16()
    xp = ops0.rmsnorm_rows(x, torch.ones(D, device=dev).bfloat16(), 1e-6, B, T)
    times = {n: [] for n, _ in libs}
    outs = {}
    for rnd in range(7):
        for name, ops in libs:
            ts = []
            for _ in range(4):
                zt = ops.linear_t(xp, w, None, B, T)
                yb = ops.yblk_empty(B * T, D, dev)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, y_blk=yb); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            if rnd:
                times[name] += ts
            outs[name] = yb
    alg = B * T * (3 * D * 2 + D * 2)
    for name, _ in libs:
        t = sorted(times[name]); med = t[len(t) // 2]
        same = bool(torch.equal(outs[name], outs[libs[0][0]]))
        print(f"B={B} T={T} {name:24s} median {med:.4f} ms (min {t[0]:.4f}) = {alg / med / 1e6 / 8000:.3f} of 8 TB/s | bit-identical to the first: {same}", flush=True)
