"""Repeat-launch determinism + agreement of evo_hyena_mfma with the three-launch modal path (itself oracle-checked).
Usage: python tools/hm_determinism.py  (EVO_AMD_LIBNAME picks the library build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from evo_amd.ops import default_ops
from evo_amd.hyena_tables import mfma_operand_table, group_permutation
from test_gpu_kernels import hyena_params, gen, bf
ops = default_ops(); DEV = "cuda:0"
d = lambda t: t.to(DEV)
worst = 0
for (B, T, D, H) in [(1, 1024, 128, 1), (2, 2500, 256, 2), (1, 8193, 512, 4), (8, 8193, 4096, 32)]:
    prm = hyena_params(D, 60); fir_w, fir_b, poles, res, dskip = [d(t) for t in prm]
    z = d(bf(torch.randn(B, T, 3 * D, generator=gen(61))))
    tab = mfma_operand_table(poles, res, dskip)
    zg = z[..., group_permutation(D, H, DEV)].contiguous()
    ref = ops.hyena_prefill(z, fir_w, fir_b, poles, res, dskip, H)
    ref = ref[0] if isinstance(ref, tuple) else ref
    ys = [ops.hyena_mfma_prefill(zg, fir_w, fir_b, dskip, tab, H).clone() for _ in range(8)]
    torch.cuda.synchronize()
    nd = [int((ys[k] != ys[0]).sum()) for k in range(1, 8)]
    e = (ys[0].double() - ref.double())
    rl2 = float(e.norm() / ref.double().norm())
    bad = int((e.abs() > ref.double().abs() * 2 ** -7 + float(ref.abs().max()) * 4e-3).sum())
    firstbad = (e.abs() > ref.double().abs() * 2 ** -7 + float(ref.abs().max()) * 4e-3).nonzero()[:3].tolist()
    print(f"B={B} T={T} D={D}: run-to-run differing elements {nd}; vs 3-launch rel-L2 {rl2:.2e} bad {bad} {firstbad}")
    worst = max(worst, max(nd), bad)
print("RESULT", "OK" if worst == 0 else "FAIL")
