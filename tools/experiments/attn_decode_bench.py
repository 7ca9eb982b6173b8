"""Decode attention (one query against the KV cache): time and KV bytes/s against context length and split count."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from evo_amd.ops import default_ops
ops = default_ops(); dev = "cuda:0"; g = torch.Generator(device=dev).manual_seed(0)
H, hd = 32, 128
def timeit(fn, n=20, calls=6):
    """GPU time per call: `calls` calls captured in a hipGraph (no host launch cost), replayed n times."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr):
            for _ in range(calls): fn()
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * calls) * 1e3
for B, T in ((1, 8192), (3, 8192), (1, 32768), (1, 131072)):
    # several caches in rotation so the 256 MB Infinity Cache cannot hold them
    n_c = max(2, int(600e6 // (B * T * 2 * H * hd * 2)) + 1)
    kvs = [torch.randn(B, T, 2, H, hd, generator=g, device=dev).bfloat16() for _ in range(min(n_c, 6))]
    q = torch.randn(B, 1, H, hd, generator=g, device=dev).bfloat16()
    pos = torch.full((B,), T - 1, dtype=torch.int64, device=dev)
    mb = B * T * 2 * H * hd * 2 / 1e6
    line = []
    for ns in (None, 16, 32, 64, 128, 256):
        it = [0]
        def f():
            it[0] = (it[0] + 1) % len(kvs)
            kv = kvs[it[0]]
            return ops.attention_decode(q, kv[:, :, 0], kv[:, :, 1], pos=pos, n_splits=ns)
        t = timeit(f)
        line.append(f"{'auto' if ns is None else ns}: {t:.1f} us ({mb / t:.2f} TB/s)")
    print(f"B={B} T={T} KV {mb:.0f} MB: " + "  ".join(line))
    del kvs
