#!/usr/bin/env python
"""Does the 256 MB Infinity Cache help a weight-streaming launch?  (a) the same 180 MB matrix again and again (resident) against
rotating copies (HBM); (b) a reader on a second stream touching matrix i+1 while the GEMV consumes matrix i."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from evo_amd.ops import default_ops
ops = default_ops(); dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)

def ev():
    return torch.cuda.Event(enable_timing=True)

for name, N, K, ncopy in (("l1l2", 22016, 4096, 8), ("proj", 12288, 4096, 12), ("out", 4096, 4096, 24), ("l3", 4096, 11008, 12)):
    ws = [(torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16() for _ in range(ncopy)]
    x = torch.randn(1, K, generator=g, device=dev).bfloat16()
    mb = N * K * 2 / 1e6
    for w in ws[:3]:
        ops._linear_small_m(x, w, None, None)
    torch.cuda.synchronize()
    # (a1) rotating
    e0, e1 = ev(), ev(); e0.record()
    for r in range(3):
        for w in ws:
            ops._linear_small_m(x, w, None, None)
    e1.record(); torch.cuda.synchronize(); t_rot = e0.elapsed_time(e1) / (3 * ncopy)
    # (a2) resident
    e0, e1 = ev(), ev(); e0.record()
    for r in range(3 * ncopy):
        ops._linear_small_m(x, ws[0], None, None)
    e1.record(); torch.cuda.synchronize(); t_res = e0.elapsed_time(e1) / (3 * ncopy)
    # (b) reader one matrix ahead on a side stream (torch sum over an int32 view: a plain streaming read)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    iv = [w.view(torch.int32) for w in ws]
    torch.cuda.synchronize()
    e0, e1 = ev(), ev(); e0.record()
    done = [None] * (3 * ncopy + 1)
    with torch.cuda.stream(side):
        side.wait_stream(main)
        iv[0].sum(); d = ev(); d.record(); done[0] = d
    for i in range(3 * ncopy):
        with torch.cuda.stream(side):
            iv[(i + 1) % ncopy].sum(); d = ev(); d.record(); done[i + 1] = d
        main.wait_event(done[i])
        ops._linear_small_m(x, ws[i % ncopy], None, None)
    e1.record(); main.wait_stream(side); torch.cuda.synchronize(); t_pf = e0.elapsed_time(e1) / (3 * ncopy)
    # reader alone
    e0, e1 = ev(), ev(); e0.record()
    for r in range(3):
        for v in iv:
            v.sum()
    e1.record(); torch.cuda.synchronize(); t_rd = e0.elapsed_time(e1) / (3 * ncopy)
    print(f"{name} {mb:.0f} MB: rotating {t_rot*1e3:.1f} us ({mb/t_rot/1e3:.2f} TB/s) | resident {t_res*1e3:.1f} us ({mb/t_res/1e3:.2f} TB/s) | "
          f"with a reader one ahead {t_pf*1e3:.1f} us per matrix ({mb/t_pf/1e3:.2f} TB/s) | reader alone {t_rd*1e3:.1f} us ({mb/t_rd/1e3:.2f} TB/s)")
    del ws, iv
