#!/usr/bin/env python
"""Interleaved A/B of evo_linear_mfma_bf16 builds against hipBLASLt (torch.mm / addmm_) on the model's four dense-layer shapes.
    python tools/gemm_ab.py libevo_mi355x.so libevo_u2.so ...        (files in evo_amd/_lib/)
Every build is checked against the library product (max |diff| in units of the bf16 spacing) before it is timed."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd import _build

P = ctypes.c_void_p; I = ctypes.c_int64
libs = []
for arg in sys.argv[1:]:
    lib = ctypes.CDLL(str(_build.LIBDIR / arg))
    fn = lib.evo_linear_mfma_bf16
    fn.argtypes = [P] * 5 + [I] * 3 + [P]; fn.restype = ctypes.c_int
    libs.append((arg, fn))
dev = "cuda:0"; M = int(os.environ.get("GEMM_M", "65536"))
st = torch.cuda.current_stream().cuda_stream
rounds = int(os.environ.get("GEMM_ROUNDS", "5")); reps = int(os.environ.get("GEMM_REPS", "8"))
g = torch.Generator(device=dev).manual_seed(0)
for (lname, N, K, res) in (("proj", 12288, 4096, False), ("out", 4096, 4096, True), ("l1l2", 22016, 4096, False), ("l3", 4096, 11008, True)):
    x = torch.randn(M, K, generator=g, device=dev).bfloat16()
    w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).bfloat16()
    r0 = torch.randn(M, N, generator=g, device=dev).bfloat16() if res else None
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ref = torch.mm(x, w.t()) if not res else r0.clone().addmm_(x, w.t())
    def run(fn):
        if res:
            y.copy_(r0)
        rc = fn(x.data_ptr(), w.data_ptr(), None, y.data_ptr() if res else None, y.data_ptr(), M, N, K, st)
        assert rc == 0, rc
    def run_timed(fn):                                         # (the residual is read from y itself: no copy inside the timed region)
        rc = fn(x.data_ptr(), w.data_ptr(), None, y.data_ptr() if res else None, y.data_ptr(), M, N, K, st)
        assert rc == 0, rc
    def lib_mm():
        if res:
            y.addmm_(x, w.t())
        else:
            torch.mm(x, w.t(), out=y)
    errs = {}
    for name, fn in libs:
        run(fn); torch.cuda.synchronize()
        errs[name] = float(((y.float() - ref.float()).abs() / (ref.float().abs() * 2.0 ** -8 + 1e-3)).max())
    times = {name: [] for name, _ in libs}; times["hipBLASLt"] = []
    for r in range(rounds + 1):
        for name, fn in [("hipBLASLt", None)] + libs:
            f = lib_mm if fn is None else (lambda fn=fn: run_timed(fn))
            f(); 
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(reps):
                f()
            b_.record(); torch.cuda.synchronize()
            if r:
                times[name].append(a_.elapsed_time(b_) / reps)
    base = sorted(times["hipBLASLt"])[len(times["hipBLASLt"]) // 2]
    fl = 2.0 * M * N * K
    print(f"{lname:5s} M={M} N={N} K={K}: hipBLASLt {base:.3f} ms {fl / base / 1e9:.0f} TF/s", flush=True)
    for name, _ in libs:
        ts = sorted(times[name]); med = ts[len(ts) // 2]
        print(f"      {name:22s} {med:.3f} ms (min {ts[0]:.3f}) {fl / med / 1e9:.0f} TF/s = {base / med * 100:.1f} % of the library | max err {errs[name]:.2f} bf16 spacings", flush=True)
