"""GPU (-m gpu): every C-ABI kernel against the fp64 op-level oracle on the same seeded bf16 inputs.

Tolerances (floating point, stated per SURVEY.md 8c): outputs are bf16, so one output rounding costs
<= 2^-9 relative per element (rel-L2 ~ 1.1e-3).  Kernels that accumulate in fp32 from exact bf16 inputs
must stay within rel-L2 2e-3 and |err| <= 2^-8*|ref| + atol; attention additionally rounds P to bf16
before P.V (FlashAttention-2 numerics) and gets rel-L2 4e-3.
"""
import math

import pytest
import torch

from oracle import stripedhyena_ref as R

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from evo_amd.ops import HipOps
    return HipOps()


def gen(seed):
    return torch.Generator().manual_seed(seed)


def bf(x):
    return x.to(torch.bfloat16)


def rel_l2(a, ref):
    a, ref = a.double().cpu(), ref.double().cpu()
    return ((a - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def assert_close_bf16(got, ref, rl2=2e-3, atol=None):
    got, ref = got.double().cpu(), ref.double().cpu()
    assert torch.isfinite(got).all()
    atol = float(ref.abs().max()) * 2e-3 if atol is None else atol
    err = (got - ref).abs()
    bound = ref.abs() * 2 ** -8 + atol
    assert (err <= bound).all(), f"max excess {(err - bound).max().item():.3e}"
    assert rel_l2(got, ref) < rl2, rel_l2(got, ref)


# ------------------------------------------------------------------------------------------------ elementwise
def test_embed(ops):
    w = bf(torch.randn(512, 256, generator=gen(0)))
    ids = torch.randint(0, 512, (3, 17), generator=gen(1))
    out = ops.embed(ids.to(DEV), w.to(DEV))
    assert torch.equal(out.cpu(), w[ids.reshape(-1)])          # bit exact gather


@pytest.mark.parametrize("M,D", [(5, 256), (37, 4096), (3, 1024), (2, 8192)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_rmsnorm(ops, M, D, with_bias):
    x = bf(torch.randn(M, D, generator=gen(2)) * 3)
    scale = bf(1 + 0.1 * torch.randn(D, generator=gen(3)))
    bias = bf(torch.randn(D, generator=gen(4))) if with_bias else None
    xd = x.to(DEV).clone()
    out = ops.rmsnorm(xd, bias.to(DEV) if with_bias else None, scale.to(DEV), 1e-6)
    if with_bias:
        x_new = bf(x.double() + bias.double())
        assert torch.equal(xd.cpu(), x_new)                     # in-place residual write, RNE
        _, ref = R.op_rmsnorm(x_new, scale, 1e-6)
    else:
        assert torch.equal(xd.cpu(), x)
        _, ref = R.op_rmsnorm(x, scale, 1e-6)
    assert_close_bf16(out, ref)


def test_rmsnorm_zero_row_uses_eps(ops):
    x = torch.zeros(2, 256, dtype=torch.bfloat16)
    out = ops.rmsnorm(x.to(DEV), None, torch.ones(256, dtype=torch.bfloat16, device=DEV), 1e-6)
    assert torch.equal(out.cpu(), x)


@pytest.mark.parametrize("B,T,H,hd,scaling", [(2, 19, 2, 128, 1.0), (1, 300, 4, 128, 16.0), (1, 5, 1, 64, 1.0)])
def test_rope(ops, B, T, H, hd, scaling):
    qkv = bf(torch.randn(B, T, 3, H, hd, generator=gen(5)))
    t = torch.arange(7, 7 + T, dtype=torch.float32) / scaling
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.outer(t, inv)
    cos, sin = torch.cos(fr).bfloat16().float(), torch.sin(fr).bfloat16().float()
    ref = R.op_rope(qkv, cos, sin)
    got = ops.rope_(qkv.to(DEV).clone(), cos.to(DEV), sin.to(DEV))
    assert torch.equal(got[:, :, 2].cpu(), qkv[:, :, 2])        # v untouched
    assert_close_bf16(got, ref)
    # q_scale (round 6): the rotated QUERIES times the factor, one rounding; k and v as before, bit for bit; 1.0 is the plain launch
    c = ops.attn_q_scale(hd)
    got_s = ops.rope_(qkv.to(DEV).clone(), cos.to(DEV), sin.to(DEV), q_scale=c)
    assert torch.equal(got_s[:, :, 1:], got[:, :, 1:])
    ref_s = ref.clone()
    ref_s[:, :, 0] *= c
    assert_close_bf16(got_s, ref_s)
    assert torch.equal(ops.rope_(qkv.to(DEV).clone(), cos.to(DEV), sin.to(DEV), q_scale=1.0), got)


@pytest.mark.parametrize("M,I", [(7, 64), (33, 10928)])
def test_gelu_gate(ops, M, I):
    g = bf(torch.randn(M, 2 * I, generator=gen(6)) * 2)
    assert_close_bf16(ops.gelu_gate(g.to(DEV)), R.op_gelu_gate(g))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_logprob_entropy(ops, dtype):
    logits = (torch.randn(41, 512, generator=gen(7)) * 4).to(dtype)
    tgt = torch.randint(0, 512, (41,), generator=gen(8))
    tgt[3] = -1
    lp, en = ops.logprob_entropy(logits.to(DEV), tgt.to(DEV), want_logprob=True, want_entropy=True)
    rlp, ren = R.op_logprob_entropy(logits, tgt)
    assert (lp.double().cpu() - rlp).abs().max() < 2e-5 * 20
    assert (en.double().cpu() - ren).abs().max() < 1e-4
    assert lp[3].item() == 0.0


@pytest.mark.parametrize("M,K", [(1, 256), (63, 4096), (64, 4096), (513, 4096), (8193, 4096)])
def test_fused_unembed_logprob_matches_two_kernel_path_and_oracle(ops, M, K):
    """evo_unembed_logprob_bf16 (unembed + log-softmax + gather + entropy in one kernel, logits never in HBM) vs
    (a) the fp64 oracle on bf16-rounded logits and (b) the two-kernel path it replaces [REF evo/scoring.py:47-57]."""
    hid = bf(torch.randn(M, K, generator=gen(30)))
    emb = bf(torch.randn(512, K, generator=gen(31)) * (4.0 / math.sqrt(K)))
    tgt = torch.randint(0, 512, (M,), generator=gen(32))
    tgt[M // 2] = -1                                                   # masked position -> log-prob 0
    lp, en = ops.unembed_logprob(hid.to(DEV), emb.to(DEV), tgt.to(DEV), want_logprob=True, want_entropy=True)
    logits = bf(hid.double() @ emb.double().t())                       # one rounding of every logit to bf16
    rlp, ren = R.op_logprob_entropy(logits, tgt)
    # a logit that lands on a rounding boundary may differ by one bf16 ulp between two fp32 summation orders
    ulp = float(logits.abs().max()) * 2 ** -7
    assert lp[M // 2].item() == 0.0
    assert (lp.double().cpu() - rlp).abs().max() <= 2.5 * ulp
    assert (en.double().cpu() - ren).abs().max() <= 2.5 * ulp
    assert (lp.double().cpu() - rlp).abs().mean() <= 0.25 * ulp        # ... and almost none of them do
    lg2 = ops.linear(hid.to(DEV), emb.to(DEV), None)
    lp2, en2 = ops.logprob_entropy(lg2, tgt.to(DEV), want_logprob=True, want_entropy=True)
    assert (lp - lp2).abs().max().item() <= 2.5 * ulp and (en - en2).abs().max().item() <= 2.5 * ulp


# ------------------------------------------------------------------------------------------------ Hyena operator
def hyena_params(D, seed):
    g = gen(seed)
    fir_w = bf(torch.randn(3 * D, 3, generator=g) * 0.3)
    fir_b = bf(torch.randn(3 * D, generator=g) * 0.1)
    u = torch.rand(D, 8, generator=g)
    one_minus = 10.0 ** (-5.0 + 4.0 * u)
    mag, ang = 1.0 - one_minus, (torch.rand(D, 8, generator=g) * 2 - 1) * math.pi
    poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous()
    res = (torch.randn(D, 8, 2, generator=g) * 0.25 * torch.sqrt(one_minus).unsqueeze(-1) * 4).float().contiguous()
    dskip = bf(torch.randn(D, generator=g) * 0.5)
    return fir_w, fir_b, poles, res, dskip


def run_hyena(ops, z, prm, H, **kw):
    fir_w, fir_b, poles, res, dskip = prm
    d = lambda t: None if t is None else t.to(DEV)
    kw = {k: (d(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
    return ops.hyena_prefill(z.to(DEV), d(fir_w), d(fir_b), d(poles), d(res), d(dskip), H, **kw)


def run_ct(ops, z, prm, H, z_halo=None, s0=None, want_state=False):
    """The single-pass operator as the product launches it (csrc/hyena_ct.hip through evo_hyena_ct): token-major z [B, T, 3 D] is laid
    out as the channel-major z^T the projection writes (HipOps.zt_from_rows; pad / tail-block positions = NaN: nothing there may reach an
    output).  -> (y [B, T, D], end state or None)."""
    from evo_amd.hyena_tables import mfma_operand_table
    fir_w, fir_b, poles, res, dskip = [t.to(DEV) for t in prm]
    B, T, _ = z.shape
    tab = mfma_operand_table(poles, res, dskip)
    zt = ops.zt_from_rows(z.to(DEV), B, T, float("nan"))
    out = ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, z_halo=None if z_halo is None else z_halo.to(DEV),
                       s0=None if s0 is None else s0.to(DEV), want_state=want_state, poles=poles)
    return out if want_state else (out, None)


@pytest.mark.parametrize("B,T,D,H,seg", [
    (2, 37, 256, 2, 8),          # ragged last segment, scalar tail
    (1, 1, 128, 1, 64),          # single token
    (2, 3, 128, 1, 4),           # shorter than one unrolled group
    (1, 513, 4096, 32, 64),      # BASELINE configs[0] length at the real width
    (2, 8193, 256, 2, 512),      # BASELINE configs[1] length (T = nt + 1 is odd)
    (1, 8193, 256, 2, None),     # default segment heuristic
])
def test_hyena_prefill_matches_oracle(ops, B, T, D, H, seg):
    prm = hyena_params(D, 10)
    z = bf(torch.randn(B, T, 3 * D, generator=gen(11)))
    y, st = run_hyena(ops, z, prm, H, want_state=True, seg_len=seg)
    ry, rst = R.op_hyena(z, *prm, H)
    assert_close_bf16(y, ry)
    assert (st.cpu().to(torch.complex128) - rst).abs().max() <= 2e-5 * rst.abs().max()


@pytest.mark.parametrize("B,T,D,H", [
    (2, 37, 128, 1),             # one ragged tile
    (1, 1, 128, 1),              # single token
    (2, 513, 256, 2),            # a full tile + 1 row: the tile-to-tile carry
    (1, 513, 4096, 32),          # BASELINE configs[0] length at the real width
    (2, 8193, 256, 2),           # BASELINE configs[1] length: 17 tiles
    (1, 3000, 128, 1),
    (40, 300, 128, 1),           # more batch rows than row streams (8 groups x 32): workgroups walk 1 or 2 rows each
    (3, 1100, 256, 2),           # three row streams of three tiles; the pipeline crosses a row boundary mid-stream
])
def test_hyena_single_pass_matches_oracle(ops, B, T, D, H):
    """evo_hyena_ct (block Toeplitz + aggregates on bf16 MFMA with hi/lo-split operands, fp32 block scan, carry on bf16
    MFMA with hi/lo-split states) vs the fp64 oracle -- outputs AND the end state (== prefill_via_modal_fft) -- and vs the
    three-launch modal kernels on the same data."""
    prm = hyena_params(D, 60)
    z = bf(torch.randn(B, T, 3 * D, generator=gen(61)))
    y, st = run_ct(ops, z, prm, H, want_state=True)
    ry, rst = R.op_hyena(z, *prm, H)
    assert_close_bf16(y, ry, rl2=2e-3 if y.numel() > 4096 else 3.5e-3)      # (few outputs: the rel-L2 estimate is noisy)
    assert (st.cpu().to(torch.complex128) - rst).abs().max() <= 2e-5 * rst.abs().max()
    y_modal, _ = run_hyena(ops, z, prm, H)
    assert rel_l2(y, y_modal) < 2.5e-3
    if T > 4:                                                          # with a halo (sequence-parallel / resumed shard)
        cut = max(2, T // 3)
        yb, _ = run_ct(ops, z[:, cut:].contiguous(), prm, H, z_halo=z[:, cut - 2:cut].contiguous())
        ya = R.op_hyena(z[:, cut:], *prm, H, z_halo=z[:, cut - 2:cut])[0]
        assert_close_bf16(yb, ya)


@pytest.mark.parametrize("B,T,D,H,cut", [
    (2, 1300, 256, 2, 517),      # the cut inside a block of the second tile; the tail ends mid-block
    (1, 8193, 128, 1, 4096),     # the cut on a tile boundary; T - 1 = 16 tiles exactly
    (3, 700, 128, 1, 33),        # a first piece shorter than two blocks
    (1, 2050, 128, 1, 2049),     # a one-token tail
])
def test_hyena_single_pass_carry_in_and_end_state(ops, B, T, D, H, cut):
    """The single-pass kernel takes the modal state entering t = 0 and returns the state after t = T-1 (cached
    prefill [REF evo/generation.py:117,152], sequence-parallel shards).  Sharp check of the mechanics: a sequence
    evaluated in two pieces -- piece B seeded with piece A's end state and FIR history -- must reproduce the one-piece
    outputs and end state to fp32 rounding (only the tile / block alignment of the sums differs), and all of it must match
    the fp64 oracle at the operator's tolerance."""
    prm = hyena_params(D, 70)
    z = bf(torch.randn(B, T, 3 * D, generator=gen(71)))
    y1, s1 = run_ct(ops, z, prm, H, want_state=True)
    ya, sa = run_ct(ops, z[:, :cut].contiguous(), prm, H, want_state=True)
    yb, sb = run_ct(ops, z[:, cut:].contiguous(), prm, H, want_state=True, z_halo=z[:, cut - 2:cut].contiguous(), s0=sa)
    y2 = torch.cat([ya, yb], 1)
    assert (sb - s1).abs().max().item() <= 2e-5 * s1.abs().max().item()
    d = (y2.double() - y1.double()).abs()
    assert (d <= y1.double().abs() * 2.0 ** -7 + 1e-4 * float(y1.abs().max())).all()   # at most one bf16 ulp (a rounding boundary crossed)
    assert float((d > 0).double().mean()) < 0.02
    ry, rst = R.op_hyena(z, *prm, H)
    assert_close_bf16(y2, ry)
    assert (sb.cpu().to(torch.complex128) - rst).abs().max() <= 2e-5 * rst.abs().max()
    # the modal three-launch kernels continue from the single-pass kernel's state and vice versa (decode after a cached prefill)
    ym, sm = run_hyena(ops, z[:, cut:].contiguous(), prm, H, want_state=True, z_halo=z[:, cut - 2:cut].contiguous(), s0=sa)
    assert rel_l2(ym, yb) < 2.5e-3 and (sm - sb).abs().max().item() <= 2e-5 * sb.abs().max().item()


def test_hyena_single_pass_131k_long_memory(ops):
    """T = 131,073 with |p| up to 0.99999: 257 sequential tiles of carried fp32 state (one head keeps the fp64 oracle
    affordable)."""
    B, T, D, H = 1, 131073, 128, 1
    prm = hyena_params(D, 62)
    z = bf(torch.randn(B, T, 3 * D, generator=gen(63)))
    y, st = run_ct(ops, z, prm, H, want_state=True)
    ry, rst = R.op_hyena(z, *prm, H)
    assert_close_bf16(y, ry)
    assert (st.cpu().to(torch.complex128) - rst).abs().max() <= 1e-4 * rst.abs().max()


def test_hyena_single_pass_is_bit_reproducible_at_bench_size(ops):
    """8 x 8,193 x 4096 (BASELINE configs[1]): eight launches on the same data are bit-identical (the hazard stress: two waves of a
    SIMD share the matrix pipe; an early schedule read MFMA results before they were written, on a timing-dependent ~10 % of the
    elements) and agree with the three-launch modal path (itself oracle-checked above) to one bf16 rounding."""
    from evo_amd.hyena_tables import mfma_operand_table
    B, T, D, H = 8, 8193, 4096, 32
    prm = [t.to(DEV) for t in hyena_params(D, 64)]
    fir_w, fir_b, poles, res, dskip = prm
    z = bf(torch.randn(B, T, 3 * D, generator=gen(65))).to(DEV)
    tab = mfma_operand_table(poles, res, dskip)
    zt = ops.zt_from_rows(z, B, T, float("nan"))
    ys = [ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H).clone() for _ in range(8)]
    for k in range(1, 8):
        assert torch.equal(ys[k], ys[0]), k
    ref = ops.hyena_prefill(z, fir_w, fir_b, poles, res, dskip, H)
    ref = ref[0] if isinstance(ref, tuple) else ref
    e = (ys[0].double() - ref.double()).abs()
    assert float(e.norm() / ref.double().norm()) < 3e-4                     # two bf16 roundings of (almost) the same fp32
    assert (e <= ref.double().abs() * 2.0 ** -7 + float(ref.abs().max()) * 1e-3).all()


def test_hyena_prefill_with_padding_mask_matches_oracle(ops):
    """mask [B,T]: a padded position contributes nothing to the modes and outputs zero (upstream multiplies the FIR
    output by padding_mask).  Pads in the middle, at the start, across a segment boundary; all-ones == no mask."""
    B, T, D, H = 2, 301, 256, 2
    prm = hyena_params(D, 40)
    z = bf(torch.randn(B, T, 3 * D, generator=gen(41)))
    mask = torch.ones(B, T, dtype=torch.bool)
    mask[0, 0] = False
    mask[0, 30:35] = False
    mask[1, 250:] = False
    y, st = run_hyena(ops, z, prm, H, want_state=True, seg_len=32, mask=mask)
    ry, rst = R.op_hyena(z, *prm, H, mask=mask)
    assert_close_bf16(y, ry)
    assert (y[0, 30:35] == 0).all() and (y[1, 250:] == 0).all()
    assert (st.cpu().to(torch.complex128) - rst).abs().max() <= 2e-5 * rst.abs().max()
    y1, _ = run_hyena(ops, z, prm, H, seg_len=32, mask=torch.ones(B, T, dtype=torch.bool))
    y0, _ = run_hyena(ops, z, prm, H, seg_len=32)
    assert torch.equal(y1, y0)


def test_hyena_prefill_131k_long_memory(ops):
    """BASELINE configs[2] length: T = 131,073 with |p| up to 0.99999.  One head (128 channels) keeps the fp64
    oracle affordable; the segment/carry machinery is independent of the head count."""
    B, T, D, H = 1, 131073, 128, 1
    prm = hyena_params(D, 12)
    z = bf(torch.randn(B, T, 3 * D, generator=gen(13)))
    y, st = run_hyena(ops, z, prm, H, want_state=True, seg_len=512)
    ry, rst = R.op_hyena(z, *prm, H)
    assert_close_bf16(y, ry)
    assert (st.cpu().to(torch.complex128) - rst).abs().max() <= 1e-4 * rst.abs().max()
    # a different segmentation must give the same function (size-independent property)
    y2, _ = run_hyena(ops, z, prm, H, seg_len=1024)
    assert rel_l2(y2, y) < 2e-3


def test_hyena_prefill_linearity_in_v_at_full_size(ops):
    """Full width, T = 8,193: scaling the v third by 2 scales x1v (minus bias terms) -- checked through the
    exact identity  op(z; b=0) is degree-1 in v: y(2v) = 2 y(v)  when FIR biases are zero."""
    B, T, D, H = 1, 8193, 4096, 32
    fir_w, fir_b, poles, res, dskip = hyena_params(D, 14)
    prm = (fir_w, torch.zeros_like(fir_b), poles, res, dskip)
    z = bf(torch.randn(B, T, 3 * D, generator=gen(15)))
    z2 = z.clone().view(B, T, H, 3, 128)
    z2[:, :, :, 2] = z2[:, :, :, 2] * 2                      # exact in bf16
    y1, _ = run_hyena(ops, z, prm, H)
    y2, _ = run_hyena(ops, z2.view(B, T, 3 * D), prm, H)
    assert rel_l2(y2, 2 * y1.double()) < 2.5e-3


def test_hyena_prefill_split_with_halo_and_state(ops):
    B, T, D, H = 2, 300, 256, 2
    prm = hyena_params(D, 16)
    z = bf(torch.randn(B, T, 3 * D, generator=gen(17)))
    y, st = run_hyena(ops, z, prm, H, want_state=True, seg_len=32)
    ya, sa = run_hyena(ops, z[:, :131].contiguous(), prm, H, want_state=True, seg_len=32)
    yb, sb = run_hyena(ops, z[:, 131:].contiguous(), prm, H, want_state=True, seg_len=16,
                       z_halo=z[:, 129:131].contiguous(), s0=sa)
    ry, rst = R.op_hyena(z, *prm, H)
    assert_close_bf16(torch.cat([ya, yb], 1), ry)
    assert (sb.cpu().to(torch.complex128) - rst).abs().max() <= 2e-5 * rst.abs().max()


def test_hyena_step_matches_oracle_and_prefill(ops):
    B, T, D, H = 3, 40, 256, 2
    prm = hyena_params(D, 18)
    fir_w, fir_b, poles, res, dskip = prm
    z = bf(torch.randn(B, T + 5, 3 * D, generator=gen(19)))
    _, st = run_hyena(ops, z[:, :T].contiguous(), prm, H, want_state=True, seg_len=8)
    fir_state = z[:, T - 2:T].transpose(1, 2).contiguous().to(DEV)
    r_fs, r_st = fir_state.cpu(), st.cpu().to(torch.complex128)
    ry_full, _ = R.op_hyena(z, *prm, H)
    for k in range(5):
        y = ops.hyena_step(z[:, T + k].contiguous().to(DEV), fir_state, st, fir_w.to(DEV), fir_b.to(DEV),
                           poles.to(DEV), res.to(DEV), dskip.to(DEV), H)
        ry, r_fs, r_st = R.op_hyena_step(z[:, T + k], r_fs, r_st, *prm, H)
        assert_close_bf16(y, ry)
        assert_close_bf16(y, ry_full[:, T + k], rl2=3e-3)
        assert torch.equal(fir_state.cpu(), r_fs.to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------ attention
def _attn(ops, q, k, v, off, pre):
    """(engine output, fp64 oracle output).  `pre` (round 6): the queries carry softmax_scale * log2(e) (what evo_rope_qk_bf16's q_scale
    folds into its one rounding: here bf16(q c)), the kernels take scores as exponents (csrc/attn_w64.hip PRE; softmax_scale = 0 in the C
    ABI) -- the oracle is given the SAME rounded queries, un-scaled in fp64."""
    if not pre:
        return ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), off), R.op_attention(q, k, v, off)
    c = ops.attn_q_scale(128)
    qp = (q.float() * c).bfloat16()
    return ops.attention(qp.to(DEV), k.to(DEV), v.to(DEV), off, prescaled=True), R.op_attention(qp.double() / c, k, v, off)


@pytest.mark.parametrize("pre", [False, True])
@pytest.mark.parametrize("B,H,Tq,Tk,off", [
    (2, 2, 37, 37, 0),            # one ragged tile
    (1, 2, 513, 513, 0),          # BASELINE configs[0] length
    (1, 1, 128, 128, 0),          # exactly one query block / two key tiles
    (1, 2, 1000, 1000, 0),
    (2, 2, 1, 300, 299),          # decode against a KV cache
    (1, 2, 64, 200, 136),         # chunk continuation
    (1, 1, 130, 700, 570),        # sequence-parallel shard: local queries, gathered keys
])
def test_attention_matches_oracle(ops, B, H, Tq, Tk, off, pre):
    q = bf(torch.randn(B, Tq, H, 128, generator=gen(20)))
    k = bf(torch.randn(B, Tk, H, 128, generator=gen(21)))
    v = bf(torch.randn(B, Tk, H, 128, generator=gen(22)))
    o, ref = _attn(ops, q, k, v, off, pre)
    assert_close_bf16(o, ref, rl2=4e-3, atol=2e-2)


def test_attention_packed_qkv_views_and_kv_cache_layout(ops):
    B, T, H = 2, 257, 2
    qkv = bf(torch.randn(B, T, 3, H, 128, generator=gen(23))).to(DEV)
    o = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0)
    ref = R.op_attention(qkv[:, :, 0].cpu(), qkv[:, :, 1].cpu(), qkv[:, :, 2].cpu(), 0)
    assert_close_bf16(o, ref, rl2=4e-3, atol=2e-2)
    kv = torch.full((B, 400, 2, H, 128), float("nan"), dtype=torch.bfloat16, device=DEV)   # poison past the end
    kv[:, :T] = qkv[:, :, 1:3]
    o2 = ops.attention(qkv[:, :, 0], kv[:, :T, 0], kv[:, :T, 1], 0)
    assert torch.equal(o2, o)


@pytest.mark.parametrize("pre", [False, True])
def test_attention_outlier_scores(ops, pre):
    """Large score outliers late in the key range force big running-max jumps (the rescale path)."""
    B, H, T = 1, 1, 384
    q = bf(torch.randn(B, T, H, 128, generator=gen(24)))
    k = bf(torch.randn(B, T, H, 128, generator=gen(25)))
    v = bf(torch.randn(B, T, H, 128, generator=gen(26)))
    k[0, 200, 0] = q[0, 300, 0] * 3       # key 200 dominates query 300 (and nearby rows see a huge score)
    k[0, 70, 0] = q[0, 90, 0] * 2
    o, ref = _attn(ops, q, k, v, 0, pre)
    assert_close_bf16(o, ref, rl2=4e-3, atol=2e-2)


@pytest.mark.parametrize("pre", [False, True])
def test_attention_reference_point_moves_several_times_in_one_row(ops, pre):
    """The 64-rows-per-wave kernel keeps a row's reference point until a tile's largest exponent exceeds it by W_THR = 32 log2 units
    (csrc/attn_w64.hip); a staircase of spikes 1.5 x, 3 x, 5 x, 8 x |q|^2 (steps of 24-49 units, one per key tile and later) walks the
    rescale path four times in the same row, a -5 x spike on the row's FIRST key starts it 81 units below, and the neighbours in the
    wave stay put (alpha = 1 exactly).  `pre` (queries pre-scaled, reference points start at 0 and move beyond +-64 units only): the 5 x and
    8 x spikes (80 / 130 units) move the staircase row twice, from then on the wave adds its rows' offsets in the burst path."""
    B, H, T = 1, 2, 900
    q = bf(torch.randn(B, T, H, 128, generator=gen(124)))
    k = bf(torch.randn(B, T, H, 128, generator=gen(125)))
    v = bf(torch.randn(B, T, H, 128, generator=gen(126)))
    for key, mul in ((100, 1.5), (300, 3.0), (500, 5.0), (700, 8.0)):
        k[0, key, 0] = q[0, 850, 0] * mul
    k[0, 0, 1] = q[0, 640, 1] * -5.0
    k[0, 600, 1] = q[0, 640, 1] * 4.0
    o, ref = _attn(ops, q, k, v, 0, pre)
    assert_close_bf16(o, ref, rl2=4e-3, atol=2e-2)
    assert_close_bf16(o[:, 850], ref[:, 850], rl2=4e-3, atol=2e-2)                  # the staircase row itself
    assert_close_bf16(o[:, 640], ref[:, 640], rl2=4e-3, atol=2e-2)


@pytest.mark.parametrize("pre", [False, True])
def test_attention_wide_scores_like_the_models_block_8(ops, pre):
    """Scores with a standard deviation of ~9 log2 units (q, k of rms 2.5: what the synthetic 7B model hands block 8,
    tools/attn_instep_ab.py --model): softmax rows are dominated by a handful of keys, the running maximum climbs ~20 units along a
    row.  Every output row against the fp64 oracle."""
    B, H, T = 1, 2, 2049
    q = bf(torch.randn(B, T, H, 128, generator=gen(127)) * 2.5)
    k = bf(torch.randn(B, T, H, 128, generator=gen(128)) * 2.5)
    v = bf(torch.randn(B, T, H, 128, generator=gen(129)))
    o, ref = _attn(ops, q, k, v, 0, pre)
    assert_close_bf16(o, ref, rl2=4e-3, atol=2e-2)


@pytest.mark.parametrize("shift", [-5.0, -1.2, 3.5])
def test_attention_prescaled_reference_point_leaves_zero_on_the_first_tile(ops, shift):
    """PRE form only: every score of a head sits `shift` x |u|^2 (x softmax_scale log2 e: -81 / -19.6 / +57 log2 units) away from 0 -- all
    queries ~ u, all keys ~ shift u.  At -81 a row's FIRST visible tile pulls its reference point down (nothing accumulated yet: no
    rescale); at -19.6 and +57 the reference stays at 0 and P = 2^s runs at 2^-20 / 2^57 through the bf16 P and the fp32 sums."""
    B, H, T = 1, 2, 1500
    u = torch.randn(128, generator=gen(130))
    u = u / u.norm() * math.sqrt(128.0)
    q = bf(u[None, None, None, :] + 0.05 * torch.randn(B, T, H, 128, generator=gen(131)))
    k = bf(shift * u[None, None, None, :] + 0.3 * torch.randn(B, T, H, 128, generator=gen(132)))
    v = bf(torch.randn(B, T, H, 128, generator=gen(133)))
    o, ref = _attn(ops, q, k, v, 0, True)
    assert torch.isfinite(o.float()).all()
    assert_close_bf16(o, ref, rl2=4e-3, atol=2e-2)


@pytest.mark.parametrize("pre", [False, True])
def test_attention_4k_causal(ops, pre):
    B, H, T = 1, 2, 4099
    q = bf(torch.randn(B, T, H, 128, generator=gen(27)))
    k = bf(torch.randn(B, T, H, 128, generator=gen(28)))
    v = bf(torch.randn(B, T, H, 128, generator=gen(29)))
    o, ref = _attn(ops, q, k, v, 0, pre)
    assert_close_bf16(o, ref, rl2=4e-3, atol=2e-2)


# ------------------------------------------------------------------------------------------------ decode attention
@pytest.mark.parametrize("B,H,Tk,splits", [(1, 2, 1, None), (2, 2, 300, None), (1, 4, 8193, 8), (3, 2, 700, 1), (1, 2, 130, 5),
                                            # round 6 (32-key halves, double-buffered): a last block with one key / exactly one half / one key into
                                            # its second half, more splits than blocks, long runs of halves per wave
                                            (1, 2, 65, 3), (1, 2, 96, 2), (2, 3, 2048 + 33, None), (1, 2, 32, 4), (1, 1, 20000, 4)])
def test_attention_decode_matches_oracle(ops, B, H, Tk, splits):
    q = bf(torch.randn(B, 1, H, 128, generator=gen(30)))
    kv = bf(torch.randn(B, Tk + 37, 2, H, 128, generator=gen(31)))           # cache with slack past Tk
    ref = R.op_attention(q, kv[:, :Tk, 0], kv[:, :Tk, 1], Tk - 1)
    kvd = kv.to(DEV)
    o = ops.attention_decode(q.to(DEV), kvd[:, :Tk, 0], kvd[:, :Tk, 1], n_splits=splits)
    assert_close_bf16(o, ref, rl2=4e-3, atol=2e-2)
    # position read from device memory (the hipGraph form): keys [0, pos] of the full-capacity view
    pos = torch.tensor([Tk - 1], dtype=torch.int64, device=DEV)
    o2 = ops.attention_decode(q.to(DEV), kvd[:, :, 0], kvd[:, :, 1], pos=pos, n_splits=splits)
    assert_close_bf16(o2, ref, rl2=4e-3, atol=2e-2)
    # queries pre-scaled by softmax_scale * log2(e) (round 6: what the model's rotary launch hands over), softmax_scale = 0 in the C ABI
    c = ops.attn_q_scale(128)
    qp = (q.float() * c).bfloat16()
    ref_p = R.op_attention(qp.double() / c, kv[:, :Tk, 0], kv[:, :Tk, 1], Tk - 1)
    o3 = ops.attention_decode(qp.to(DEV), kvd[:, :, 0], kvd[:, :, 1], pos=pos, n_splits=splits, prescaled=True)
    assert_close_bf16(o3, ref_p, rl2=4e-3, atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("positions,scaling", [([5], 1.0), ([100, 60000, 131071], 16.0), ([100, 8191, 70000], 1.0),
                                               ([(37 * b + 5) % 300 for b in range(9)], 16.0)])
def test_rope_append_decode_is_bitwise_table_rope_and_indexed_copy(positions, scaling):
    """evo_rope_append_decode_bf16 (rotary at per-row positions + KV append, one launch) against the path it replaces:
    cos/sin table for those positions -> evo_rope_qk_bf16 -> indexed copy into the cache.  Bit for bit, including angles of
    ~1e5 rad (the range reduction of sinf / cosf)."""
    from evo_amd.ops import default_ops
    ops = default_ops()
    H, hd, B = 32, 128, len(positions)
    g = gen(B)
    qkv = bf(torch.randn(B, 1, 3, H, hd, generator=g)).to(DEV)
    pos = torch.tensor(positions, dtype=torch.int64, device=DEV)
    cap = max(positions) + 1
    kv_a = torch.zeros(B + 1, cap, 2, H, hd, dtype=torch.bfloat16, device=DEV)
    kv_b = torch.zeros_like(kv_a)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32, device=DEV) / hd))
    t = pos.to(torch.float32)
    if scaling != 1.0:
        t = t / scaling
    fr = torch.outer(t, inv)
    cos, sin = torch.cos(fr).to(torch.bfloat16).float().contiguous(), torch.sin(fr).to(torch.bfloat16).float().contiguous()
    want = qkv.clone()
    ops.rope_(want.view(1, B, 3, H, hd), cos, sin)
    kv_b[torch.arange(B, device=DEV), pos] = want[:, 0, 1:3]
    got = qkv.clone()
    ops.rope_append_decode(got, kv_a[:B], pos, inv, scaling)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(kv_a, kv_b)
    # q_scale: the same factor in both launches -> the same bits (queries scaled, k / v / cache untouched by it)
    c = ops.attn_q_scale(hd)
    want_s = qkv.clone()
    ops.rope_(want_s.view(1, B, 3, H, hd), cos, sin, q_scale=c)
    got_s = qkv.clone()
    kv_c = torch.zeros_like(kv_a)
    ops.rope_append_decode(got_s, kv_c[:B], pos, inv, scaling, q_scale=c)
    assert torch.equal(got_s, want_s) and torch.equal(kv_c, kv_b) and torch.equal(got_s[:, :, 1:], got[:, :, 1:])
    assert not torch.equal(got_s[:, :, 0], got[:, :, 0])


# ---- the single-pass operator on CHANNEL-MAJOR z (csrc/hyena_ct.hip): forms of its launch --------------------------------------------
def _zt_of(ops, z, B, T, pad_value=float("nan")):
    """z [B, T, 3 D] -> z^T [Mp / 256, 3 D, 256] as rmsnorm_rows + linear_t lay it out (batch rows at a pitch of Tp); the pad positions hold
    `pad_value` (NaN by default: whatever sits there must never reach an output or the state)."""
    return ops.zt_from_rows(z, B, T, pad_value)


@pytest.mark.parametrize("B,T,D,H", [
    (2, 37, 128, 1),             # one ragged tile
    (1, 1, 128, 1),              # single token
    (2, 513, 256, 2),            # a full tile + 1 step: the tile-to-tile carry, lane 0's history from lane 63 of the previous tile
    (1, 513, 4096, 32),          # BASELINE configs[0] length at the real width
    (2, 8193, 256, 2),           # BASELINE configs[1] length: 17 tiles, batch rows padded 8,193 -> 8,200
    (1, 3000, 128, 1),
    (40, 300, 128, 1),           # more batch rows than row streams: workgroups walk several rows (history re-seeded per row)
    (3, 1100, 256, 2),           # the pipeline crosses a row boundary mid-stream
    (8, 2049, 1024, 8),
    (9, 1024, 256, 2),           # T a multiple of the tile: no ragged tile, no padding
])
def test_hyena_ct_matches_oracle_in_every_launch_form(ops, B, T, D, H):
    """evo_hyena_ct (z^T in, a lane's eight steps loaded as 16 contiguous bytes, no window in LDS) vs the fp64 oracle -- outputs and
    end state, with and without FIR history and a carry-in state, the state-only walk, row-major and blocked y (every form gives
    the same bits)."""
    from evo_amd.hyena_tables import mfma_operand_table
    prm = hyena_params(D, 100)
    fir_w, fir_b, poles, res, dskip = [t.to(DEV) for t in prm]
    z = bf(torch.randn(B, T, 3 * D, generator=gen(101))).to(DEV)
    halo = bf(torch.randn(B, 2, 3 * D, generator=gen(102))).to(DEV)
    s0 = torch.view_as_complex(torch.randn(B, D, 8, 2, generator=gen(103)).contiguous()).to(DEV)
    tab = mfma_operand_table(poles, res, dskip)
    zt = _zt_of(ops, z, B, T)
    for kw in (dict(), dict(z_halo=halo), dict(z_halo=halo, s0=s0)):
        y_new, s_new = ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, want_state=True, poles=poles, **kw)
        assert not bool(torch.isnan(y_new.float()).any()) and not bool(torch.isnan(torch.view_as_real(s_new)).any())
        y_pln = ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, **kw)                     # the scoring instantiation (no end state)
        assert torch.equal(y_pln, y_new), list(kw)
        yb = ops.yblk_empty(B * T + 77, D, DEV).fill_(7.0)
        ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, y_blk=yb, y_row0=77, **kw)
        rows = ops.yblk_to_rows(yb, B * T + 77)
        assert torch.equal(rows[77:].view(B, T, D), y_new), list(kw)
        assert bool((rows[:77] == 7.0).all())                                          # nothing written in front of y_row0
        s_only = ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H, poles=poles, state_only=True, **kw)
        assert torch.equal(torch.view_as_real(s_only), torch.view_as_real(s_new)), list(kw)
        ry, rst = R.op_hyena(z.cpu(), *prm, H, **{k: (halo if k == "z_halo" else s0).cpu() for k in kw})
        assert_close_bf16(y_new, ry, rl2=2e-3 if y_new.numel() > 4096 else 3.5e-3)
        assert (s_new.cpu().to(torch.complex128) - rst).abs().max() <= 2e-5 * rst.abs().max(), list(kw)


def test_hyena_ct_pad_positions_do_not_matter(ops):
    """What sits between T and the row pitch (and behind the last row) never reaches an output: NaN, zeros and large finite values there
    give the same bits."""
    from evo_amd.hyena_tables import mfma_operand_table
    B, T, D, H = 5, 1003, 256, 2
    prm = hyena_params(D, 120)
    fir_w, fir_b, poles, res, dskip = [t.to(DEV) for t in prm]
    z = bf(torch.randn(B, T, 3 * D, generator=gen(121))).to(DEV)
    tab = mfma_operand_table(poles, res, dskip)
    outs = [ops.hyena_ct(_zt_of(ops, z, B, T, v), B, T, fir_w, fir_b, tab, H, want_state=True, poles=poles) for v in (float("nan"), 0.0, 3e38)]
    for y, s in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(torch.view_as_real(s), torch.view_as_real(outs[0][1]))


@pytest.mark.parametrize("T", [700, 1026])                   # (plain form | tail form with two tail tokens per row)
def test_hyena_ct_row_subrange(ops, T):
    """`b_first`: the row groups of a sequence-parallel shard are launched on sub-ranges of the batch rows of ONE z^T tensor."""
    from evo_amd.hyena_tables import mfma_operand_table
    B, D, H = 5, 256, 2
    prm = hyena_params(D, 110)
    fir_w, fir_b, poles, res, dskip = [t.to(DEV) for t in prm]
    z = bf(torch.randn(B, T, 3 * D, generator=gen(111))).to(DEV)
    tab = mfma_operand_table(poles, res, dskip)
    zt = _zt_of(ops, z, B, T)
    y_all = ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H)
    assert torch.equal(ops.hyena_ct(zt, 2, T, fir_w, fir_b, tab, H, b_first=3, b_total=B), y_all[3:5])
    assert torch.equal(ops.hyena_ct(_zt_of(ops, z[1:4].contiguous(), 3, T), 3, T, fir_w, fir_b, tab, H), y_all[1:4])
    assert torch.equal(ops.zt_rows(zt, B, T, T - 2, 2), z[:, T - 2:])


@pytest.mark.parametrize("B,T,D", [(3, 1003, 256), (1, 2500, 512), (8, 8193, 256), (2, 640, 4096), (4, 1028, 256), (1, 16385, 512)])
def test_rmsnorm_rows_and_transposed_projection_are_bitwise_the_plain_ones(ops, B, T, D):
    """rmsnorm_rows + linear_t (the dense layer with swapped operands, bias along the rows; the tail tokens of the tail form through the
    weight-streaming kernel) = rmsnorm + the plain hand-written dense layer, transposed and in z^T's position order: bit for bit."""
    x = bf(torch.randn(B * T, D, generator=gen(130))).to(DEV)
    scale = bf(1.0 + 0.1 * torch.randn(D, generator=gen(131))).to(DEV)
    w = bf(torch.randn(3 * D, D, generator=gen(132)) * D ** -0.5).to(DEV)
    b = bf(torch.randn(3 * D, generator=gen(133)) * 0.1).to(DEV)
    Tm, Tp, Mp, r = ops.zt_layout(B, T)
    assert Tp % 8 == 0 and Tp >= Tm and Tp - Tm < ops.ZT_ALIGN and Mp % 256 == 0 and Mp >= B * Tp and ops.zt_shape_ok(B, T, 3 * D, D)
    assert (r == 0 and Tm == T) or (1 <= r <= 8 and Tm == T - r and Tm % 512 == 0 and Tp == Tm)
    n_ref = ops.rmsnorm(x, None, scale, 1e-6).view(B, T, D)
    xp = ops.rmsnorm_rows(x, scale, 1e-6, B, T)
    assert tuple(xp.shape) == (Mp + 16, D)
    assert torch.equal(xp[:B * Tp].view(B, Tp, D)[:, :Tm], n_ref[:, :Tm])
    assert bool((xp[:B * Tp].view(B, Tp, D)[:, Tm:] == 0).all()) and bool((xp[B * Tp:Mp] == 0).all())
    if r:                                                                               # the tail tokens, compactly behind the main rows
        assert torch.equal(xp[Mp:Mp + B * r].view(B, r, D), n_ref[:, Tm:]) and bool((xp[Mp + B * r:] == 0).all())
    for bias in (b, None):
        zt = ops.linear_t(xp, w, bias, B, T)
        assert tuple(zt.shape) == (Mp // 256 + (1 if r else 0), 3 * D, 256)
        # main area: the same kernel in the plain orientation on the same rows -- bit for bit on the whole 256-row tiles
        z_main = ops.linear_mfma(xp[:Mp].contiguous(), w, bias)                           # [Mp, 3 D]
        got_main = zt[:Mp // 256].permute(0, 2, 1).reshape(Mp, 3 * D)
        assert torch.equal(got_main, z_main), int((got_main != z_main).sum())
        if r:                                                                           # tail block: the weight-streaming kernel's rows
            got_tail = ops.zt_rows(zt, B, T, Tm, r).reshape(B * r, 3 * D)
            assert torch.equal(got_tail, ops._linear_small_m(xp[Mp:Mp + B * r], w, bias, None))
            ref_tail = torch.nn.functional.linear(n_ref[:, Tm:].reshape(B * r, D).float(), w.float(), None if bias is None else bias.float())
            assert (got_tail.float() - ref_tail).abs().max().item() <= 2.0 ** -7 * ref_tail.abs().max().item()
        z_all = torch.nn.functional.linear(n_ref.reshape(B * T, D).float(), w.float(), None if bias is None else bias.float())
        err = (ops.zt_rows(zt, B, T, 0, T).reshape(B * T, 3 * D).float() - z_all).abs().max().item()
        assert err <= 2.0 ** -7 * z_all.abs().max().item(), err


def test_hyena_ct_is_bit_reproducible_at_bench_size(ops):
    """8 x 8,193 x 4096 (BASELINE configs[1]): eight launches on the same data are bit-identical and agree with the three-launch
    modal path to one bf16 rounding."""
    from evo_amd.hyena_tables import mfma_operand_table
    B, T, D, H = 8, 8193, 4096, 32
    prm = [t.to(DEV) for t in hyena_params(D, 64)]
    fir_w, fir_b, poles, res, dskip = prm
    z = bf(torch.randn(B, T, 3 * D, generator=gen(65))).to(DEV)
    tab = mfma_operand_table(poles, res, dskip)
    zt = _zt_of(ops, z, B, T)
    ys = [ops.hyena_ct(zt, B, T, fir_w, fir_b, tab, H).clone() for _ in range(8)]
    for k in range(1, 8):
        assert torch.equal(ys[k], ys[0]), k
    ref = ops.hyena_prefill(z, fir_w, fir_b, poles, res, dskip, H)
    ref = ref[0] if isinstance(ref, tuple) else ref
    e = (ys[0].double() - ref.double()).abs()
    assert float(e.norm() / ref.double().norm()) < 3e-4
    assert (e <= ref.double().abs() * 2.0 ** -7 + float(ref.abs().max()) * 1e-3).all()
