#!/usr/bin/env python
"""Development check of the 64-rows-per-wave attention kernel (csrc/attn_w64.hip) on one MI355X: correctness against eager fp64
attention at shapes that exercise every path (short first block, ragged last tile, query offsets, strided views, score spikes that
force the deferred-max rescale), bit-identity across query offsets, agreement of the W_THR = 8 build with a W_THR = 0 build
(EVO_AMD_LIBNAME=libevo_thr0.so built with -DW_THR=0.0f, when present), then timings at the two bench shapes.
Every GPU call goes through the C ABI (evo_attn_fwd_causal_bf16)."""
import ctypes, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd import ops as O

dev = "cuda:0"
lib = O.load_library()
lib0 = None
p0 = os.path.join(os.path.dirname(O._build.lib_path()), "libevo_thr0.so")
if os.path.exists(p0):
    lib0 = ctypes.CDLL(p0)
    lib0.evo_attn_fwd_causal_bf16.argtypes = lib.evo_attn_fwd_causal_bf16.argtypes
    lib0.evo_attn_fwd_causal_bf16.restype = ctypes.c_int


def attn(l, q, k, v, off):
    B, Tq, H, hd = q.shape
    o = torch.empty(B, Tq, H, hd, dtype=torch.bfloat16, device=q.device)
    vt = torch.full((B, H, hd, (k.shape[1] + 63) // 64 * 64), float("nan"), dtype=torch.bfloat16, device=q.device)      # (poisoned workspace)
    rc = l.evo_attn_fwd_causal_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, Tq, k.shape[1], int(off),
                                    q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                                    v.stride(0), v.stride(1), v.stride(2), 1.0 / math.sqrt(hd), vt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return o


def ref64(q, k, v, off, rows=None):
    B, Tq, H, hd = q.shape
    Tk = k.shape[1]
    out = torch.empty(B, Tq if rows is None else len(rows), H, hd, dtype=torch.float64, device=q.device)
    r = torch.arange(Tq, device=q.device) if rows is None else rows
    for b in range(B):
        for h in range(H):
            sc = (q[b, r, h].double() @ k[b, :, h].double().t()) / math.sqrt(hd)
            sc = sc.masked_fill(torch.arange(Tk, device=q.device)[None, :] > (r + off)[:, None], float("-inf"))
            out[b, :, h] = torch.softmax(sc, -1) @ v[b, :, h].double()
    return out


def rl2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for (B, H, Tq, Tk, off) in [(1, 2, 129, 129, 0), (1, 2, 255, 255, 0), (1, 2, 256, 256, 0), (1, 2, 257, 257, 0), (1, 2, 513, 513, 0),
                            (2, 3, 1000, 1000, 0), (2, 2, 300, 700, 400), (1, 1, 130, 700, 570), (1, 2, 320, 321, 1), (1, 2, 4099, 4099, 0),
                            (1, 2, 2048, 2048, 0), (1, 4, 8193, 8193, 0)]:
    q = torch.randn(B, Tq, H, 128, generator=g, device=dev).bfloat16()
    kv = torch.randn(B, Tk, 2, H, 128, generator=g, device=dev).bfloat16()     # strided K / V views (the KV-cache layout)
    k, v = kv[:, :, 0], kv[:, :, 1]
    o = attn(lib, q, k, v, off)
    torch.cuda.synchronize()
    r = ref64(q, k, v, off)
    e = rl2(o, r)
    worst = float(((o.double() - r).abs() - r.abs() * 2 ** -7).max())
    fin = bool(torch.isfinite(o.float()).all())
    same0 = None if lib0 is None else rl2(attn(lib0, q, k, v, off), o)
    ok = fin and e < 4e-3 and worst < 2e-2
    bad += not ok
    print(f"[shape B={B} H={H} Tq={Tq} Tk={Tk} off={off}] rel-L2 {e:.3e} worst |err|-2^-7|ref| {worst:.2e} finite {fin}"
          f"{'' if same0 is None else f' | vs THR=0 build rel-L2 {same0:.2e}'} {'ok' if ok else 'FAIL'}", flush=True)

# score spikes: a late key dominates a row (reference point jumps by far more than W_THR), early spike too
q = torch.randn(1, 640, 2, 128, generator=g, device=dev).bfloat16()
k = torch.randn(1, 640, 2, 128, generator=g, device=dev).bfloat16()
v = torch.randn(1, 640, 2, 128, generator=g, device=dev).bfloat16()
k[0, 200, 0] = q[0, 300, 0] * 3
k[0, 70, 1] = q[0, 90, 1] * 2
k[0, 500, 0] = q[0, 639, 0] * 4
k[0, 0, 1] = q[0, 400, 1] * 5
o = attn(lib, q, k, v, 0)
r = ref64(q, k, v, 0)
e = rl2(o, r)
print(f"[spikes] rel-L2 {e:.3e} max|err| {float((o.double() - r).abs().max()):.3e}"
      f"{'' if lib0 is None else f' | vs THR=0 build rel-L2 {rl2(attn(lib0, q, k, v, 0), o):.2e}'} {'ok' if e < 4e-3 else 'FAIL'}", flush=True)
bad += not (e < 4e-3)

# bit identity across launches and query offsets (H = 8, T = 16,385)
T, H = 16385, 8
qkv = torch.randn(1, T, 3, H, 128, generator=g, device=dev).bfloat16()
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
full = attn(lib, q, k, v, 0)
same = all(torch.equal(attn(lib, q, k, v, 0), full) for _ in range(4))
offs_ok = {off: bool(torch.equal(attn(lib, q[:, off:], k, v, off), full[:, off:])) for off in (256, 1000, 4097, 12289, T - 257, T - 129)}
rows = torch.tensor([0, 63, 64, 255, 256, 4096, 8191, T - 1], device=dev)
e = rl2(full[:, rows], ref64(q, k, v, 0, rows))
print(f"[repro T={T}] launches identical {same}; offsets identical {offs_ok}; rows vs fp64 rel-L2 {e:.2e}", flush=True)
bad += (not same) or (not all(offs_ok.values())) or not (e < 4e-3)

if "--no-time" not in sys.argv:
    for (B, T) in ((8, 8193), (1, 131073)):
        H = 32
        qkv = torch.randn(B, T, 3, H, 128, generator=g, device=dev).bfloat16()
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        for _ in range(2):
            o = attn(lib, q, k, v, 0)
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); o = attn(lib, q, k, v, 0); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        fl = B * 4 * 4096 * T * T / 2
        rows = torch.tensor([0, 1, 255, 256, 257, T // 2, T - 2, T - 1], device=dev)
        e = rl2(o[:1, rows], ref64(q[:1], k[:1], v[:1], 0, rows))
        print(f"[time form={os.environ.get('EVO_AMD_ATTN_FORM', '2')}] B={B} T={T}: median {ts[2]:.3f} ms (min {ts[0]:.3f}) = {fl / ts[2] / 1e9:.0f} TFLOP/s = "
              f"{fl / ts[2] / 1e9 / 2500:.3f} of 2.5 PFLOP/s | rows vs fp64 rel-L2 {e:.2e}", flush=True)
        del qkv, q, k, v, o
print("CHECK", "FAILED" if bad else "PASSED", flush=True)
sys.exit(1 if bad else 0)
