"""A/B of the hand-written MFMA dense layer (evo_linear_mfma_bf16) against hipBLASLt (torch.mm) on the model's
layer shapes.  Usage: python tools/bench_gemm.py [--m 65544]"""
import argparse
import sys
import pathlib
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from evo_amd.ops import default_ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=65544)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--quick", action="store_true", help="Wqkv and out_proj shapes, own kernel only")
    a = ap.parse_args()
    ops = default_ops()
    M = a.m
    for name, N, K, bias, res in [("Wqkv", 12288, a.k, True, False), ("out_proj", 4096, a.k, False, True),
                                  ("l1l2(pad)", 22016, 4096, False, False), ("l3(pad)", 4096, 11008, False, True)][:2 if a.quick else 4]:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda").to(torch.bfloat16) if bias else None
        r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
        fl = 2.0 * M * N * K
        if a.quick:
            t_lib = float("nan")
        elif res:
            t_lib = timeit(lambda: r.addmm_(x, w.t()))
        elif bias:
            t_lib = timeit(lambda: torch.addmm(b, x, w.t()))
        else:
            t_lib = timeit(lambda: torch.mm(x, w.t()))
        t_own = timeit(lambda: ops.linear_mfma(x, w, b, r))
        print(f"{name:10s} M={M} N={N} K={K}: hipBLASLt {t_lib:7.3f} ms {fl / t_lib / 1e9:7.1f} TF/s | "
              f"mfma {t_own:7.3f} ms {fl / t_own / 1e9:7.1f} TF/s", flush=True)
        del x, w, b, r


if __name__ == "__main__":
    main()
