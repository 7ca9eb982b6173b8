import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from evo_amd.ops import default_ops
ops = default_ops(); dev="cuda:0"; g=torch.Generator(device=dev).manual_seed(0)
D,H=4096,32
rn=lambda *s, std=1.0: torch.randn(*s, generator=g, device=dev)*std
tag=os.environ.get("EVO_AMD_LIBNAME","default")
for M in (1,2,4):
    sets=[]
    for _ in range(8):
        u=torch.rand(D,8,generator=g,device=dev); mag=1-10**(-5+4*u); ang=(torch.rand(D,8,generator=g,device=dev)*2-1)*math.pi
        sets.append(dict(pre=rn(D,std=.2).add_(1).bfloat16(), wp=rn(3*D,D,std=.02).bfloat16(), bp=rn(3*D,std=.1).bfloat16(),
            fw=rn(3*D,3,std=.3).bfloat16(), fb=rn(3*D,std=.1).bfloat16(), poles=torch.stack([mag*torch.cos(ang),mag*torch.sin(ang)],-1).float().contiguous(),
            res=rn(D,8,2,std=.25).float().contiguous(), dk=rn(D,std=.5).bfloat16(), fs=rn(M,3*D,2).bfloat16(), iir=torch.view_as_complex(rn(M,D,8,2,std=.5).float().contiguous())))
    x=rn(M,D).bfloat16()
    def run(b): return ops.hyena_decode_fused(x,b["pre"],1e-6,b["wp"],b["bp"],b["fs"],b["iir"],b["fw"],b["fb"],b["poles"],b["res"],b["dk"],H)
    for b in sets: run(b)
    torch.cuda.synchronize()
    gr=torch.cuda.CUDAGraph(); st=torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr):
            for b in sets: run(b)
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gr.replay()
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/(20*len(sets))*1e3
    y=run(sets[0]); torch.cuda.synchronize()
    print(f"[{tag}] M={M}: fused hyena decode {t:.1f} us ({3*D*D*2/1e6/t:.2f} TB/s) checksum {float(y.float().sum()):.4f} {float(sets[0]['iir'].abs().sum()):.3f}")
