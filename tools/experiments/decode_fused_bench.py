"""Per-launch time (hipGraph replay, rotating weight sets) and checksums of the fused decode launches at M = 1..4:
pre-norm + projections + Hyena step | post-norm + l1|l2 + gate | pre-norm + Wqkv.   EVO_AMD_LIBNAME picks the library build."""
import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from evo_amd.ops import default_ops
ops = default_ops(); dev = "cuda:0"; g = torch.Generator(device=dev).manual_seed(0)
D, H, I = 4096, 32, 11008
rn = lambda *s, std=1.0: torch.randn(*s, generator=g, device=dev) * std
tag = os.environ.get("EVO_AMD_LIBNAME", "default")

def graph_time(fns):
    for f in fns: f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr):
            for f in fns: f()
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (20 * len(fns)) * 1e3

for M in [int(a) for a in os.environ.get("MS", "1,2,3,4,5,6,8").split(",")]:
    x = rn(M, D).bfloat16()
    sets = []
    for _ in range(6):
        u = torch.rand(D, 8, generator=g, device=dev); mag = 1 - 10 ** (-5 + 4 * u); ang = (torch.rand(D, 8, generator=g, device=dev) * 2 - 1) * math.pi
        sets.append(dict(pre=rn(D, std=.2).add_(1).bfloat16(), wp=rn(3 * D, D, std=.02).bfloat16(), bp=rn(3 * D, std=.1).bfloat16(),
                         fw=rn(3 * D, 3, std=.3).bfloat16(), fb=rn(3 * D, std=.1).bfloat16(),
                         poles=torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous(),
                         res=rn(D, 8, 2, std=.25).float().contiguous(), dk=rn(D, std=.5).bfloat16(), fs=rn(M, 3 * D, 2).bfloat16(),
                         iir=torch.view_as_complex(rn(M, D, 8, 2, std=.5).float().contiguous()),
                         w12=rn(2 * I, D, std=.02).bfloat16(), wo=rn(D, D, std=.02).bfloat16(), bo=rn(D, std=.1).bfloat16(),
                         w3=rn(D, I, std=.02).bfloat16()))
    a_in = rn(M, I).bfloat16(); y_in = rn(M, D).bfloat16(); xr = rn(M, D).bfloat16(); xr0 = xr.clone()
    hy = lambda b: ops.hyena_decode_fused(x, b["pre"], 1e-6, b["wp"], b["bp"], b["fs"], b["iir"], b["fw"], b["fb"], b["poles"], b["res"], b["dk"], H)
    gt = lambda b: ops.mlp_gate(x, b["w12"], b["pre"], 1e-6)
    nl = lambda b: ops.norm_linear(x, b["pre"], 1e-6, b["wp"], b["bp"])
    ou = lambda b: ops.linear_residual_(xr, y_in, b["wo"], bias=b["bo"])
    l3 = lambda b: ops.linear_residual_(xr, a_in, b["w3"])
    t_ou = graph_time([lambda b=b: ou(b) for b in sets])
    t_l3 = graph_time([lambda b=b: l3(b) for b in sets])
    xr.copy_(xr0); ou(sets[0]); c_ou = float(xr.float().sum()); xr.copy_(xr0); l3(sets[0]); c_l3 = float(xr.float().sum())
    t_hy = graph_time([lambda b=b: hy(b) for b in sets])
    t_gt = graph_time([lambda b=b: gt(b) for b in sets])
    t_nl = graph_time([lambda b=b: nl(b) for b in sets])
    b = sets[0]
    c = (float(hy(b).float().sum()), float(b["iir"].abs().sum()), float(gt(b).float().sum()), float(nl(b).float().sum()))
    torch.cuda.synchronize()
    print(f"[{tag}] M={M}: hyena {t_hy:.1f} us  norm+l1l2+gate {t_gt:.1f} us  norm+Wqkv {t_nl:.1f} us  out {t_ou:.1f} us  l3 {t_l3:.1f} us | checksums {c_ou:.3f} {c_l3:.3f} {c[0]:.4f} {c[1]:.2f} {c[2]:.4f} {c[3]:.3f}")
