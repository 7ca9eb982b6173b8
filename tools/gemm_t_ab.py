#!/usr/bin/env python
"""The Hyena projection's three launch forms on the same operands (HIP events, 8 launches each after 2 warm-ups), at the two bench shapes:
mode 2 (group-major result, evo_linear_zg_mfma_bf16), mode 3 (swapped operands, blocked z^T, evo_linear_t_mfma_bf16) and the plain dense
layer on the padded rows (evo_linear_mfma_bf16, row-major result).  EVO_GEMM_GROUP_M=<n> changes the raster group of all three."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.ops import default_ops
ops = default_ops(); dev = "cuda:0"; D = 4096
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
print("EVO_GEMM_GROUP_M =", os.environ.get("EVO_GEMM_GROUP_M", "(default)"))
for (B, T) in ((8, 8193), (1, 131073)):
    Tp, Mp = ops.zt_geometry(B, T)
    M = B * T
    x = (torch.randn(Mp, D, generator=g, device=dev)).bfloat16()
    w = (torch.randn(3 * D, D, generator=g, device=dev) * 0.02).bfloat16()
    zg = torch.empty(3 * D // 48, M, 48, dtype=torch.bfloat16, device=dev)
    zt = torch.empty(Mp // 256, 3 * D, 256, dtype=torch.bfloat16, device=dev)
    y = torch.empty(Mp, 3 * D, dtype=torch.bfloat16, device=dev)
    calls = [("mode 2 group-major, M = %d" % (M // 256 * 256), lambda: ops.lib.evo_linear_zg_mfma_bf16(x.data_ptr(), w.data_ptr(), None, zg.data_ptr(), M // 256 * 256, M, 3 * D, D, st)),
             ("mode 3 swapped, Mp = %d" % Mp, lambda: ops.lib.evo_linear_t_mfma_bf16(x.data_ptr(), w.data_ptr(), None, zt.data_ptr(), Mp, 3 * D, D, st)),
             ("mode 0 plain, M = %d" % Mp, lambda: ops.lib.evo_linear_mfma_bf16(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), Mp, 3 * D, D, st)),
             ("mode 0 plain, M = %d" % (M // 256 * 256), lambda: ops.lib.evo_linear_mfma_bf16(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), M // 256 * 256, 3 * D, D, st))]
    for rnd in range(2):
        for nm, call in calls:
            for _ in range(2):
                assert call() == 0
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(8):
                call()
            b_.record(); torch.cuda.synchronize()
            print(f"{B}x{T} round {rnd} {nm}: {a_.elapsed_time(b_) / 8:.4f} ms", flush=True)
