"""Random-shape check of the streaming decode attention (evo_attn_decode_bf16) against an fp32 softmax over keys [0, pos[b]]:
batch 1-9, cache capacity 1-3000, one position per row, split counts from 4 to 128 (incl. more splits than key blocks).
Usage: python tools/decode_fuzz.py [n_cases]"""
import math
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops  # noqa: E402

ops = default_ops()
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rnd = random.Random(11)
g = torch.Generator(device=dev).manual_seed(3)
bad = 0
H, hd = 32, 128
for case in range(n):
    B = rnd.choice([1, 1, 2, 3, 5, 9])
    cap = rnd.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 257, 1000, rnd.randint(1, 3000)])
    pos = torch.tensor([rnd.randint(0, cap - 1) for _ in range(B)], dtype=torch.int64, device=dev)
    if rnd.random() < 0.3:
        pos[:] = cap - 1
    ns = rnd.choice([None, 4, 8, 64, 128])
    kv = torch.randn(B + 1, cap, 2, H, hd, generator=g, device=dev).bfloat16()
    q = (torch.randn(B, 1, H, hd, generator=g, device=dev) * 1.5).bfloat16()
    got = ops.attention_decode(q, kv[:B, :, 0], kv[:B, :, 1], pos=pos, n_splits=ns).float()
    got2 = ops.attention_decode(q, kv[:B, :, 0], kv[:B, :, 1], pos=pos, n_splits=ns).float()
    ref = torch.empty_like(got)
    for b in range(B):
        nk = int(pos[b]) + 1
        k = kv[b, :nk, 0].float()                     # [nk, H, hd]
        v = kv[b, :nk, 1].float()
        s = torch.einsum("hd,khd->hk", q[b, 0].float(), k) / math.sqrt(hd)
        p = torch.softmax(s, dim=-1)
        ref[b, 0] = torch.einsum("hk,khd->hd", p, v)
    torch.cuda.synchronize()
    err = (got - ref).abs()
    tol = ref.abs() * 2 ** -7 + float(ref.abs().max()) * 4e-3
    nb = int((err > tol).sum())
    same = bool(torch.equal(got, got2))
    if nb or not same or not bool(torch.isfinite(got).all()):
        bad += 1
        print(f"case {case}: B={B} cap={cap} pos={pos.tolist()} n_splits={ns}: bad={nb} max_err={float(err.max()):.3g} reproducible={same}")
print(f"RESULT {n - bad}/{n} cases agree")
