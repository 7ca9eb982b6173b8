#!/usr/bin/env python
"""Times evo_attn_fwd_causal_bf16 at the two bench shapes (HIP events, 5 launches each after 2 warm-ups) and checks query rows against
fp32 eager attention -- both forms interleaved: plain queries (softmax_scale passed to the kernel) and PRE (round 6: queries pre-scaled by
softmax_scale * log2(e), scores taken as exponents).  A/B of builds: EVO_AMD_LIBNAME=<lib> EVO_AMD_NO_REBUILD=1 python tools/attn_bench.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops
from bench import TelemetrySampler                            # sclk / power of THIS GPU from its hwmon files while the launches run
ops = default_ops(); dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
scale = float(os.environ.get("ATTN_BENCH_SCALE", "1.0"))       # 2.5: scores with the spread of the model's block 8
for (B, T) in ((8, 8193), (1, 131073)):
    H = 32
    qkv = torch.randn(B, T, 3, H, 128, generator=g, device=dev).bfloat16()
    q, k, v = (qkv[:, :, 0].float() * scale).bfloat16(), (qkv[:, :, 1].float() * scale).bfloat16(), qkv[:, :, 2]
    c = ops.attn_q_scale(128)
    for rep in range(2):
        for pre in (False, True):
            qx = (q.float() * c).bfloat16() if pre else q
            kw = {"prescaled": True} if pre else {}
            for _ in range(2):
                o = ops.attention(qx, k, v, 0, **kw)
            ts = []
            with TelemetrySampler(0, period=0.01) as tel:
                for _ in range(5 if T > 100000 else 40):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); o = ops.attention(qx, k, v, 0, **kw); b.record(); torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b))
            ts.sort()
            tl = tel.summary()
            ts = [ts[0], ts[len(ts) // 4], ts[len(ts) // 2]]
            fl = B * 4 * 4096 * T * T / 2
            rows = torch.tensor([0, 1, 255, 256, 257, T // 2, T - 2, T - 1], device=dev)
            worst = 0.0
            for h in (0, 17, 31):
                qq = qx[0, rows, h].float() / c if pre else q[0, rows, h].float()
                sc = (qq @ k[0, :, h].float().t()) / math.sqrt(128.0)
                sc = sc.masked_fill(torch.arange(T, device=dev)[None, :] > rows[:, None], float("-inf"))
                ref = torch.softmax(sc, -1) @ v[0, :, h].float()
                worst = max(worst, float((o[0, rows, h].float() - ref).norm() / ref.norm()))
            o2 = ops.attention(qx, k, v, 0, **kw)
            print(f"{os.environ.get('EVO_AMD_LIBNAME', 'default')}{' PRE  ' if pre else ' plain'} rep{rep}: B={B} T={T} x{scale:g}: median {ts[2]:.3f} ms (min {ts[0]:.3f}) = "
                  f"{fl / ts[2] / 1e9:.0f} TFLOP/s = {fl / ts[2] / 1e9 / 2500:.3f} of 2.5 PFLOP/s | rows vs fp32 eager rel-L2 {worst:.2e}, "
                  f"bit-reproducible {bool(torch.equal(o, o2))} | sclk {tl.get('sclk_MHz_mean', 0):.0f} MHz, {tl.get('power_W_mean', 0):.0f} W of {tl.get('power_cap_W', 0):.0f} "
                  f"({tl.get('samples', 0)} samples)", flush=True)
