#!/usr/bin/env python
"""A/B timing of single-pass Hyena kernels on GROUP-MAJOR z (HIP events on the launch stream), D = 4096, H = 32, at the two bench
shapes (8 x 8,193 and 1 x 131,073 tokens).
    python tools/hc_bench.py libevo_mi355x.so old:libevo_mi355x.so libevo_hc_nw4.so
Each argument is a file in evo_amd/_lib/: prefix `ct:` = its evo_hyena_ct (csrc/hyena_ct.hip: channel-major z^T, blocked y -- what the model
runs), plain = its evo_hyena_cs_zg (csrc/hyena_cs.hip, group-major z) with the BLOCKED y output, prefix `rm:` = the same with row-major y,
prefix `old:` = its evo_hyena_mfma_zg (csrc/hyena_mfma.hip, round 3).  Every build is checked against the three-launch modal operator of the default
library on the same data (and for bit-reproducibility) before it is timed.  Timing as in a scoring step: every launch follows the
projection's dense layer that writes its z (evo_linear_zg_mfma_bf16 of the default library); events bracket the Hyena launch only;
the builds are interleaved round by round in ONE process (boxes differ by 25 % in clock)."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd import _build
from evo_amd.ops import default_ops
from evo_amd.hyena_tables import mfma_operand_table, group_permutation

ops = default_ops(); dev = "cuda:0"; D, H = 4096, 32
type(ops).ZT_ALIGN = int(os.environ.get("HC_ZT_ALIGN", type(ops).ZT_ALIGN))          # batch-row pitch of z^T (A/B: 8 | 64)
print("z^T batch rows padded to", type(ops).ZT_ALIGN, "positions", flush=True)
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, std=1.0: torch.randn(*s, generator=g, device=dev) * std
fir_w = rn(3 * D, 3, std=0.3).bfloat16(); fir_b = rn(3 * D, std=0.1).bfloat16()
om = 10.0 ** (-5.0 + 4.0 * torch.rand(D, 8, generator=g, device=dev))
mag = 1.0 - om; ang = (torch.rand(D, 8, generator=g, device=dev) * 2 - 1) * math.pi
poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous()
res = (rn(D, 8, 2, std=0.25) * torch.sqrt(om).unsqueeze(-1) * 4).float().contiguous()
dskip = rn(D, std=0.5).bfloat16(); tab = mfma_operand_table(poles, res, dskip)
perm = group_permutation(D, H, dev)
P = ctypes.c_void_p; I = ctypes.c_int64
libs = []
for arg in sys.argv[1:]:
    old = "old" if arg.startswith("old:") else ("rm" if arg.startswith("rm:") else ("ct" if arg.startswith("ct:") else ""))
    name = arg.split(":", 1)[1] if old else arg
    lib = ctypes.CDLL(str(_build.LIBDIR / name))
    if old == "old":
        fn = lib.evo_hyena_mfma_zg
        fn.argtypes = [P] * 10 + [I] * 4 + [P]
    elif old == "ct":
        fn = lib.evo_hyena_ct
        fn.argtypes = [P] * 9 + [I] * 12 + [P]
    else:
        fn = lib.evo_hyena_cs_zg
        fn.argtypes = [P] * 9 + [I] * 8 + [P]
    fn.restype = ctypes.c_int
    libs.append((arg, fn, old))
st = torch.cuda.current_stream().cuda_stream
rounds = int(os.environ.get("HM_ROUNDS", "6")); batch = int(os.environ.get("HM_BATCH", "12"))
shapes = [(8, 8193), (1, 131073)] if os.environ.get("HC_SHAPES") is None else eval(os.environ["HC_SHAPES"])
for (B, T) in shapes:
    z = rn(B, T, 3 * D).bfloat16()
    ref, sref = ops.hyena_prefill(z, fir_w, fir_b, poles, res, dskip, H, want_state=True)
    zg = z[..., perm].view(B * T, D // 16, 48).transpose(0, 1).contiguous()          # [groups, B T, 48]
    Tm, Tp, Mp, r_tail = ops.zt_layout(B, T)
    zt = ops.zt_from_rows(z, B, T)                                                   # [Mp / 256, 3 D, 256]: batch rows at a pitch of Tp
    nbytes = B * T * D * 8
    y = torch.empty(B, T, D, dtype=torch.bfloat16, device=dev)
    yb = ops.yblk_empty(B * T, D, dev)
    sout = torch.zeros(B, D, 8, 2, dtype=torch.float32, device=dev)

    def launch(fn, old, want_state=False):
        so = sout.data_ptr() if want_state else None
        if old == "old":
            rc = fn(zg.data_ptr(), None, fir_w.data_ptr(), fir_b.data_ptr(), dskip.data_ptr(), tab.data_ptr(), y.data_ptr(), None, so,
                    poles.data_ptr(), B, T, D, H, st)
        elif old == "ct":
            rc = fn(zt.data_ptr(), None, fir_w.data_ptr(), fir_b.data_ptr(), tab.data_ptr(), yb.data_ptr(), None, so, poles.data_ptr(),
                    B, T, D, H, zt.shape[0] * 256, Tp, 0, Tm if r_tail else 0, Mp, 0, yb.shape[0] * 128, 0, st)
        elif old == "rm":
            rc = fn(zg.data_ptr(), None, fir_w.data_ptr(), fir_b.data_ptr(), tab.data_ptr(), y.data_ptr(), None, so, poles.data_ptr(),
                    B, T, D, H, B * T, 0, 0, 0, st)
        else:
            rc = fn(zg.data_ptr(), None, fir_w.data_ptr(), fir_b.data_ptr(), tab.data_ptr(), yb.data_ptr(), None, so, poles.data_ptr(),
                    B, T, D, H, B * T, 0, yb.shape[0] * 128, 0, st)
        assert rc == 0, rc
    info = {}
    for (name, fn, old) in libs:
        y.zero_()
        launch(fn, old, True)
        torch.cuda.synchronize()
        if old in ("", "ct"):
            y.copy_(ops.yblk_to_rows(yb, B * T).view(B, T, D))
        rl2 = float((y.double() - ref.double()).norm() / ref.double().norm())
        worst = float(((y.double() - ref.double()).abs() - ref.double().abs() * 2.0 ** -7).max() / ref.abs().max())
        srel = float((torch.view_as_complex(sout) - sref).abs().max() / sref.abs().max())
        y1 = y.clone()
        same = True
        for _ in range(3):
            launch(fn, old)
            torch.cuda.synchronize()
            if old in ("", "ct"):
                y.copy_(ops.yblk_to_rows(yb, B * T).view(B, T, D))
            same = same and bool(torch.equal(y, y1))
        info[name] = (rl2, worst, srel, same)
        print(f"{B}x{T} {name:28s} vs modal: rel-L2 {rl2:.2e}, worst (|err| - 2^-7|ref|)/max {worst:.1e}, end-state rel {srel:.1e}, "
              f"bit-reproducible {same}", flush=True)
    xin = rn(B * T, D, std=1.0).bfloat16()
    wg = rn(3 * D, D, std=0.02).bfloat16()
    xpad = torch.zeros(Mp + 16, D, dtype=torch.bfloat16, device=dev)
    xpad[:B * Tp].view(B, Tp, D)[:, :Tm] = xin.view(B, T, D)[:, :Tm]
    for nm, call in (("group-major projection (mode 2)", lambda: ops.lib.evo_linear_zg_mfma_bf16(xin.data_ptr(), wg.data_ptr(), None, zg.data_ptr(), (B * T) // 256 * 256, B * T, 3 * D, D, st)),
                     ("transposed projection (swapped operands)", lambda: ops.lib.evo_linear_t_mfma_bf16(xpad.data_ptr(), wg.data_ptr(), None, zt.data_ptr(), Mp, 3 * D, D, st))):
        for _ in range(2):
            call()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        for _ in range(6):
            assert call() == 0
        b_.record(); torch.cuda.synchronize()
        print(f"{B}x{T} {nm}: {a_.elapsed_time(b_) / 6:.4f} ms", flush=True)
    times = {name: [] for name, _, _ in libs}
    Mfull = (B * T) // 256 * 256
    for r in range(rounds + 1):
        for (name, fn, old) in libs:
            evs = []
            for _ in range(batch):
                if old == "ct":          # (its own projection launch in front: the dense layer with swapped operands)
                    rc = ops.lib.evo_linear_t_mfma_bf16(xpad.data_ptr(), wg.data_ptr(), None, zt.data_ptr(), Mp, 3 * D, D, st)
                else:
                    rc = ops.lib.evo_linear_zg_mfma_bf16(xin.data_ptr(), wg.data_ptr(), None, zg.data_ptr(), Mfull, B * T, 3 * D, D, st)
                assert rc == 0
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record(); launch(fn, old); b_.record()
                evs.append((a_, b_))
            torch.cuda.synchronize()
            if r:                                            # round 0 = warm-up
                times[name].append(sum(x.elapsed_time(y_) for x, y_ in evs) / batch)
    solo = {name: [] for name, _, _ in libs}
    for r in range(3):
        for (name, fn, old) in libs:
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(batch):
                launch(fn, old)
            b_.record()
            torch.cuda.synchronize()
            if r:
                solo[name].append(a_.elapsed_time(b_) / batch)
    for (name, fn, old) in libs:
        ts = sorted(times[name]); med = ts[len(ts) // 2]
        print(f"{B}x{T} {name:28s} after-GEMM median {med:.4f} ms (min {ts[0]:.4f}, max {ts[-1]:.4f}) = {nbytes / med / 1e6 / 8000:.3f} of 8 TB/s; "
              f"back-to-back {min(solo[name]):.4f} ms = {nbytes / min(solo[name]) / 1e6 / 8000:.3f}", flush=True)
