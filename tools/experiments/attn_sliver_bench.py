import os, sys, torch
sys.path.insert(0, os.getcwd())
from evo_amd.ops import default_ops, KernelTimer
ops = default_ops(); dev="cuda:0"; g=torch.Generator(device=dev).manual_seed(0)
H=32; D=4096
for (B,T) in ((8,8192),(8,8193),(8,8224),(8,8448),(1,131072),(1,131073)):
    qkv=(torch.randn(B,T,3,H,128,generator=g,device=dev)).bfloat16()
    for _ in range(2): ops.attention(qkv[:,:,0],qkv[:,:,1],qkv[:,:,2],0)
    ops.timer=KernelTimer()
    for _ in range(4 if T<20000 else 2): ops.attention(qkv[:,:,0],qkv[:,:,1],qkv[:,:,2],0)
    torch.cuda.synchronize(); ms=ops.timer.summary()["attn_fwd"][1]; ops.timer=None
    fl=B*4*D*T*T/2
    print(f"attn B={B} T={T}: {ms:.3f} ms {fl/ms/1e9:.0f} TFLOP/s")
    del qkv
