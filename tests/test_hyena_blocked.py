"""CPU: the blocked (matrix-core) form of the Hyena long convolution -- block Toeplitz + block aggregates + Kogge-Stone
block scan + carry product, with the kernel's operand precisions (bf16 hi/lo data, bf16-split T0 / W, fp32 G / P, fp32
accumulation; the carry product G . S with both operands bf16-split) -- emulated in torch and compared with the fp64 oracle.  This pins the MATH and the PRECISION of
csrc/hyena_ct.hip (constants from evo_amd/hyena_tables.py) independently of any GPU layout question."""
import math

import pytest
import torch

from evo_amd import hyena_tables as HT
from oracle import stripedhyena_ref as R


def emulate_blocked(z, fir_w, fir_b, poles, residues, dskip, H):
    """z [B,T,3D] bf16 -> y [B,T,D] (fp32 before the output rounding), end state [B,D,8] complex64-equivalent."""
    B, T, D3 = z.shape
    D = D3 // 3
    hd = D // H
    L, NB = HT.L, HT.NB
    C = HT.blocked_constants(poles, residues, dskip)                         # filter.D rides on T0's diagonal
    f32 = torch.float32
    # FIR + bias in fp32 (as the kernel: three fp32 FMAs on bf16 inputs)
    zt = torch.nn.functional.pad(z.to(f32).transpose(1, 2), (2, 0))            # [B,3D,T+2]
    w = fir_w.to(f32)
    zc = w[None, :, 0, None] * zt[..., 0:T] + w[None, :, 1, None] * zt[..., 1:T + 1] + w[None, :, 2, None] * zt[..., 2:T + 2] \
        + fir_b.to(f32)[None, :, None]
    z4 = zc.reshape(B, H, 3 * hd, T)
    x2 = z4[:, :, :hd].reshape(B, D, T)
    x1 = z4[:, :, hd:2 * hd].reshape(B, D, T)
    v = z4[:, :, 2 * hd:].reshape(B, D, T)
    x = x1 * v                                                                  # fp32
    TT = L * NB
    nt = (T + TT - 1) // TT
    xp = torch.nn.functional.pad(x, (0, nt * TT - T)).reshape(B, D, nt, NB, L)
    x_hi = xp.to(torch.bfloat16)
    x_lo = (xp - x_hi.to(f32)).to(torch.bfloat16)
    xh, xl = x_hi.to(f32), x_lo.to(f32)
    T0h, T0l = (t.to(f32) for t in C["T0"])                                     # [D,L,L]
    Wh, Wm, Wl = (t.to(f32) for t in C["W"])                                    # [D,16,L]
    P = C["P"]                                                                  # [D,4,16]
    Gh, Gl = (t.to(f32) for t in C["Gs"])                                       # [D,L,16] bf16 hi / lo of G
    # block Toeplitz and aggregates: exact bf16 products, fp32 accumulation (einsum in fp32)
    y0 = torch.einsum("dij,bdtaj->bdtai", T0h, xh) + torch.einsum("dij,bdtaj->bdtai", T0h, xl) \
        + torch.einsum("dij,bdtaj->bdtai", T0l, xh)
    # (round 3: W in two terms -- the third, 2^-25, and the 2^-26 cross term W_mid . X_lo are below X's own 2^-17)
    E = torch.einsum("dmj,bdtaj->bdtam", Wm, xh) + torch.einsum("dmj,bdtaj->bdtam", Wh, xl) \
        + torch.einsum("dmj,bdtaj->bdtam", Wh, xh)
    del Wl

    def cmul_add(acc, coef, src):
        """acc += coef * src over interleaved (re, im) pairs, fp32."""
        ar, ai = acc[..., 0::2], acc[..., 1::2]
        cr, ci = coef[..., 0::2], coef[..., 1::2]
        sr, si = src[..., 0::2], src[..., 1::2]
        out = torch.empty_like(acc)
        out[..., 0::2] = ar + (cr * sr - ci * si)
        out[..., 1::2] = ai + (cr * si + ci * sr)
        return out

    y = torch.empty(B, D, nt, NB, L, dtype=f32)
    carry = torch.zeros(B, D, 16, dtype=f32)
    for t in range(nt):                                                         # tiles are sequential
        S = E[:, :, t].clone()                                                  # [B,D,NB,16]
        S[:, :, 0] = cmul_add(S[:, :, 0], P[None, :, 0], carry)                 # entering state -> block 0's aggregate
        for k in range(4):                                                      # Kogge-Stone inclusive scan over blocks
            d = 1 << k
            sh = torch.zeros_like(S)
            sh[:, :, d:] = S[:, :, :-d]
            S = cmul_add(S, P[None, :, k, None], sh)
        S_start = torch.cat([carry[:, :, None], S[:, :, :-1]], 2)               # state entering each block
        Sh = S_start.to(torch.bfloat16).to(f32)                                 # the kernel splits the states hi + lo in bf16
        Sl = (S_start - Sh).to(torch.bfloat16).to(f32)
        yc = torch.einsum("dim,bdam->bdai", Gh, Sh) + torch.einsum("dim,bdam->bdai", Gh, Sl) \
            + torch.einsum("dim,bdam->bdai", Gl, Sh)
        y[:, :, t] = y0[:, :, t] + yc
        carry = S[:, :, -1]
    yconv = y.reshape(B, D, nt * TT)[..., :T]
    out = yconv * x2                                                            # yconv already holds y + x1v * D
    return out.transpose(1, 2).contiguous(), yconv


def params(D, seed):
    g = torch.Generator().manual_seed(seed)
    fir_w = (torch.randn(3 * D, 3, generator=g) * 0.3).bfloat16()
    fir_b = (torch.randn(3 * D, generator=g) * 0.1).bfloat16()
    om = 10.0 ** (-5.0 + 4.0 * torch.rand(D, 8, generator=g))
    mag, ang = 1.0 - om, (torch.rand(D, 8, generator=g) * 2 - 1) * math.pi
    poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous()
    res = (torch.randn(D, 8, 2, generator=g) * 0.25 * torch.sqrt(om).unsqueeze(-1) * 4).float().contiguous()
    dskip = (torch.randn(D, generator=g) * 0.5).bfloat16()
    return fir_w, fir_b, poles, res, dskip


@pytest.mark.parametrize("B,T,D,H", [(2, 37, 128, 1), (1, 513, 256, 2), (1, 1537, 128, 1)])
def test_blocked_form_matches_oracle(B, T, D, H):
    prm = params(D, 50)
    z = torch.randn(B, T, 3 * D, generator=torch.Generator().manual_seed(51)).bfloat16()
    y, _ = emulate_blocked(z, *prm, H)
    ry, _ = R.op_hyena(z, *prm, H)
    err = (y.double() - ry).abs()
    # BEFORE the bf16 output rounding the blocked form must sit at fp32-accumulation level, far below one bf16 ulp
    assert (err.norm() / ry.norm()).item() < 2e-5
    assert (err <= 1e-4 * ry.abs() + 1e-4 * float(ry.abs().max())).all()


def test_blocked_form_long_memory_131k():
    """T = 131,073 with |p| up to 0.99999: 257 sequential tiles of fp32 carries and fp32 p^32 ... p^256 powers."""
    D, H, T = 128, 1, 131073
    prm = params(D, 52)
    z = torch.randn(1, T, 3 * D, generator=torch.Generator().manual_seed(53)).bfloat16()
    y, _ = emulate_blocked(z, *prm, H)
    ry, _ = R.op_hyena(z, *prm, H)
    rl2 = ((y.double() - ry).norm() / ry.norm()).item()
    assert rl2 < 1e-4, rl2


def test_tables_are_consistent():
    fir_w, fir_b, poles, res, dskip = params(16, 54)
    C = HT.blocked_constants(poles, res)
    p = torch.view_as_complex(poles.double())
    r = torch.view_as_complex(res.double())
    h = C["h"]
    t = torch.arange(HT.L, dtype=torch.float64)
    want = (r[..., None] * torch.exp(torch.log(p)[..., None] * t)).real.sum(1)
    assert (h - want).abs().max() < 1e-12
    T0 = sum(x.double() for x in C["T0"])
    assert (T0[:, 5, 2] - h[:, 3]).abs().max() < 2e-5 * h.abs().max() and (T0[:, 2, 5] == 0).all()
    W = sum(x.double() for x in C["W"])
    assert (W[:, 0, HT.L - 1] - 1).abs().max() < 1e-7 and W[:, 1, HT.L - 1].abs().max() < 1e-7      # p^0 = 1
    assert (C["P"][:, 0, 0::2].double() - (p ** HT.L).real).abs().max() < 1e-6


def test_table_precision_guard_admits_plain_filters_and_refuses_cancelling_ones():
    """evo_amd/hyena_tables.py table_precision: the table-build guard of the matrix-core Hyena operator.  Silent on the synthetic law of the
    bench (long-memory poles), on upstream's init-like |p| = 0.01, on |p| = 1 exactly and on residues of any scale; fires on modes that
    cancel to 1 % at |p| = 0.5 / 0.9 -- and there the emulated kernel arithmetic indeed leaves the fp32 level (error before the output
    rounding ~1e-3 of the channel's largest output, a bf16 rounding), while an admitted filter stays at 1e-5."""
    D = 64
    g = torch.Generator().manual_seed(1)
    fir_w, fir_b, poles, res, dskip = params(D, 60)
    assert HT.table_ok(poles, res, dskip) and HT.table_precision(poles, res, dskip).max() < 1e-4

    def plain(pm, scale):
        ang = (torch.rand(D, 8, generator=g, dtype=torch.float64) * 2 - 1) * math.pi
        r = torch.randn(D, 8, 2, generator=g, dtype=torch.float64) * scale
        return torch.stack([pm * torch.cos(ang), pm * torch.sin(ang)], -1).float().contiguous(), r.float().contiguous()

    def cancelling(pm, frac):
        ang = (torch.rand(D, 8, generator=g, dtype=torch.float64) * 2 - 1) * math.pi
        ang[:, 1::2] = ang[:, 0::2] + 1e-3
        r = torch.randn(D, 8, 2, generator=g, dtype=torch.float64) * 30
        r[:, 1::2] = -frac * r[:, 0::2]
        return torch.stack([pm * torch.cos(ang), pm * torch.sin(ang)], -1).float().contiguous(), r.float().contiguous()
    for pm in (1e-2, 0.5, 0.9, 1.0 - 1e-6, 1.0):
        for scale in (1e-2, 1.0, 30.0):
            po, re = plain(pm, scale)
            assert HT.table_ok(po, re, dskip), (pm, scale, HT.table_precision(po, re, dskip).max())
    z = torch.randn(1, 1500, 3 * D, generator=torch.Generator().manual_seed(61)).bfloat16()
    seen = {}
    for pm in (0.5, 0.9):
        po, re = cancelling(pm, 0.99)
        prec = HT.table_precision(po, re, dskip)
        assert not HT.table_ok(po, re, dskip) and (prec > HT.TABLE_TOL).double().mean() > 0.5, (pm, prec.median())
        y, _ = emulate_blocked(z, fir_w, fir_b, po, re, dskip, 1)
        ry, _ = R.op_hyena(z, fir_w, fir_b, po, re, dskip, 1)
        seen[pm] = ((y.double() - ry).abs().amax((0, 1)) / ry.abs().amax((0, 1))).max().item()
        assert seen[pm] > 3e-4, seen                                     # the split does lose precision there: the guard is not crying wolf
    y, _ = emulate_blocked(z, fir_w, fir_b, poles, res, dskip, 1)
    ry, _ = R.op_hyena(z, fir_w, fir_b, poles, res, dskip, 1)
    assert ((y.double() - ry).abs().amax((0, 1)) / ry.abs().amax((0, 1))).max().item() < 5e-5
