"""GPU parity of the hand-written MFMA dense layer (csrc/gemm.hip, evo_linear_mfma_bf16) through the C ABI.

Reference: the same product accumulated in fp32 (fp64 for the tolerance floor) from the bf16 operands, plus bias and
residual in fp32, rounded once to bf16 -- what nn.Linear (+ residual add) computes in the reference's attention block
[REF stripedhyena/model.py:52-59], up to summation order.  Tolerance: 1 bf16 ulp of the result magnitude + fp32
accumulation-order noise.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from evo_amd.ops import default_ops
    return default_ops()


def _ref(x, w, b, r):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    if r is not None:
        y = y + r.double()
    return y


@pytest.mark.parametrize("M,N,K", [(9, 256, 64), (256, 256, 128), (300, 512, 192), (1000, 768, 4096), (513, 12288, 512),
                                   (2049, 4096, 4096)])
@pytest.mark.parametrize("bias,res", [(False, False), (True, False), (False, True), (True, True)])
def test_linear_mfma_matches_fp64(M, N, K, bias, res):
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16) if bias else None
    r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if res else None
    want = _ref(x, w, b, r)
    got = ops.linear_mfma(x, w, b, r.clone() if res else None)
    assert got.shape == (M, N) and got.dtype == torch.bfloat16
    err = (got.double() - want).abs()
    tol = want.abs() * 2.0 ** -8 + 1e-3          # one bf16 rounding (half-ulp = 2^-9 relative) with margin
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e} at {err.argmax().item()}"
    # and no worse than the library path (GEMM rounded to bf16, then bias / residual added and rounded again)
    lib = torch.mm(x, w.t()).float()
    if b is not None:
        lib = lib + b.float()
    if r is not None:
        lib = lib + r.float()
    lib = lib.to(torch.bfloat16).double()
    assert err.mean().item() <= (lib - want).abs().mean().item() * 1.05 + 1e-6


def test_linear_mfma_residual_in_place_and_untouched_rows():
    ops = _ops()
    M, N, K = 700, 512, 256
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / 16).to(torch.bfloat16)
    buf = torch.randn(M + 64, N, device="cuda").to(torch.bfloat16)     # rows past M must stay untouched
    keep = buf.clone()
    out = ops.linear_mfma(x, w, None, buf[:M])
    assert out.data_ptr() == buf.data_ptr()
    assert torch.equal(buf[M:], keep[M:])
    want = _ref(x, w, None, keep[:M])
    assert (buf[:M].double() - want).abs().max().item() < 0.05


def test_linear_mfma_rejects_unsupported_shapes():
    ops = _ops()
    x = torch.zeros(16, 96, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(256, 96, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.linear_mfma(x, w)                                            # K % 64 != 0
    x = torch.zeros(16, 64, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(100, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.linear_mfma(x, w)                                            # N % 256 != 0

@pytest.mark.parametrize("M,N,K,bias,res", [(8192 + 77, 4096, 128, False, False), (20000, 2048, 192, True, False),
                                            (16384, 4096, 4096, True, True), (65544, 512, 256, False, True),
                                            (33000, 1024, 1024, False, False)])
def test_linear_mfma_persistent_stream_many_tiles_per_workgroup(M, N, K, bias, res):
    """More output tiles than workgroups (514 ... 1,056 tiles on 256 CUs): every workgroup of the persistent kernel walks
    several tiles, so the (tile, k-step) stage stream crosses tile boundaries -- the fetch cursor switches tiles two
    stages ahead of the MFMAs (every 2 or 3 stages at K = 128 / 192), the epilogue runs between two tiles of one stream,
    the ragged last M tile sits in the middle of some workgroup's list."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16) if bias else None
    r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if res else None
    want = _ref(x, w, b, r)
    got = ops.linear_mfma(x, w, b, r.clone() if res else None)
    err = (got.double() - want).abs()
    tol = want.abs() * 2.0 ** -8 + 1e-3
    bad = err > tol
    assert not bool(bad.any()), f"{int(bad.sum())} elements off, first at {bad.nonzero()[0].tolist()}, max err {err.max().item():.3e}"
    again = ops.linear_mfma(x, w, b, r.clone() if res else None)
    assert torch.equal(got, again)                                       # no order-dependent accumulation anywhere


@pytest.mark.parametrize("M", [4096 + 8, 8192 + 1, 4096 + 16, 4096 + 17, 4096])
@pytest.mark.parametrize("mode", ["plain", "bias", "residual", "mfma_bias", "mfma_residual"])
def test_linear_peels_the_row_sliver(M, mode):
    """HipOps.linear / linear_residual_ send the M % 256 (<= 16) trailing rows through the weight-streaming kernel and
    the rest through the tile GEMM; the result must not show the seam."""
    ops = _ops()
    N, K = 512, 256
    g = torch.Generator(device="cuda").manual_seed(M)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    mfma = mode.startswith("mfma")
    if mode.endswith("residual"):
        got = ops.linear_residual_(r.clone(), x, w, mfma=mfma)
        want = _ref(x, w, None, r)
    elif mode.endswith("bias"):
        got = ops.linear(x, w, b, mfma=mfma)
        want = _ref(x, w, b, None)
    else:
        got = ops.linear(x, w, None)
        want = _ref(x, w, None, None)
    err = (got.double() - want).abs()
    assert bool((err <= want.abs() * 2.0 ** -8 + 2e-3).all()), f"max err {err.max().item():.3e} at row {err.argmax().item() // N}"
    assert HipOpsTail(ops, x, w) == (M % 256 if 1 <= M % 256 <= 16 else 0)


def HipOpsTail(ops, x, w):
    return ops._tail_rows(x, w)


@pytest.mark.parametrize("M,N,K,res", [(65544, 22016, 4096, False),      # l1|l2 of the gated MLP (evo-1-8k-base yml: inner 10928 -> 11008)
                                       (65544, 4096, 11008, True),       # l3 + residual: 172 k-steps through the persistent stream
                                       (131073, 12288, 4096, False)])    # a projection of the 1 x 131,073 forward
def test_linear_mfma_mlp_shapes_of_the_bench_step_vs_fp64(M, N, K, res):
    """EVO_AMD_GEMM=mfma puts every dense layer of the bench step on csrc/gemm.hip; these are its largest shapes
    [REF evo/configs/evo-1-8k-base_inference.yml: hidden 4096, inner_mlp_size 10928], at the bench's own M = 8 x 8,193
    (the BOS sliver M % 256 goes through the weight-streaming kernel inside HipOps.linear: both seams are checked)."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if res else None
    was = ops.all_gemm_mfma
    ops.all_gemm_mfma = True
    try:
        got = ops.linear_residual_(r.clone(), x, w) if res else ops.linear(x, w, None)
        again = ops.linear_residual_(r.clone(), x, w) if res else ops.linear(x, w, None)
    finally:
        ops.all_gemm_mfma = was
    assert torch.equal(got, again)
    wd = w.double().t().contiguous()
    worst = 0.0
    for i in range(0, M, 8192):                                          # fp64 reference in row chunks (<= 1.4 GB each)
        want = x[i:i + 8192].double() @ wd
        if res:
            want += r[i:i + 8192].double()
        err = (got[i:i + 8192].double() - want).abs()
        tol = want.abs() * 2.0 ** -8 + 1e-3
        bad = err > tol
        assert not bool(bad.any()), f"rows {i}+: {int(bad.sum())} elements off, max err {err.max().item():.3e}"
        worst = max(worst, (err / tol).max().item())
    print(f"[linear_mfma {M}x{N}x{K}] worst err / (2^-8 |ref| + 1e-3) = {worst:.3f}")


@pytest.mark.parametrize("M,I,K", [(256, 128, 128), (700, 256, 192), (4096 + 8, 1024, 512), (16384, 2048, 4096), (65544, 11008, 4096)])
def test_mlp_gate_fused_is_the_unfused_arithmetic(M, I, K):
    """evo_mlp_gate_mfma_bf16 (GELU * gate in the dense layer's epilogue) against the two-launch form on the SAME dense-layer kernel
    (evo_linear_mfma_bf16 on [W1; W2], then evo_gelu_gate_bf16): the accumulation order per output is identical and so are the
    roundings, so the results must agree bit for bit; and against the fp64 restatement of the reference's arithmetic
    [REF stripedhyena/layers.py ParallelGatedMLP: gelu(l1 x) * l2 x with bf16 dense-layer outputs]."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + I + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w12 = (torch.randn(2 * I, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    w12g = ops.pack_gate_weights(w12)
    assert torch.equal(w12g.view(I // 32, 2, 32, K)[:, 0].reshape(I, K), w12[:I])
    assert torch.equal(w12g.view(I // 32, 2, 32, K)[:, 1].reshape(I, K), w12[I:])
    was = ops.mlp_gate_fused
    ops.mlp_gate_fused = True
    try:
        assert ops.mlp_gate_fused_ok(x, w12g)
        got = ops.mlp_gate(x, w12, w12g=w12g)
        again = ops.mlp_gate(x, w12, w12g=w12g)
    finally:
        ops.mlp_gate_fused = was
    assert got.shape == (M, I) and torch.equal(got, again)
    r = ops._tail_rows(x, w12)
    two = ops.gelu_gate(ops.linear_mfma(x[: M - r], w12))
    assert torch.equal(got[: M - r], two), f"{int((got[: M - r] != two).sum())} elements differ from the two-launch form"
    worst = 0.0
    for i in range(0, M, 8192):                                          # fp64 reference in row chunks
        z = (x[i:i + 8192].double() @ w12.double().t()).to(torch.bfloat16).double()
        want = torch.nn.functional.gelu(z[:, :I]) * z[:, I:]
        err = (got[i:i + 8192].double() - want).abs()
        # one bf16 rounding of the product, plus what a one-ulp difference of the bf16-rounded z1 / z2 (summation order vs fp64) moves it
        tol = want.abs() * 2.0 ** -6 + 2.0 ** -7 * z[:, I:].abs() + 2e-3
        assert bool((err <= tol).all()), f"rows {i}+: max err {err.max().item():.3e}"
        worst = max(worst, (err / tol).max().item())
    print(f"[mlp_gate fused {M}x{I}x{K}] worst err / tol = {worst:.3f}")


@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (1024, 4096, 4096), (65536 + 8, 4096, 4096), (768 + 200, 512, 256)])
def test_output_projection_on_the_blocked_hyena_output_is_the_row_major_dense_layer(M, N, K):
    """evo_linear_xblk_mfma_bf16 (round 4): the Hyena block's output projection reads y in the BLOCKED layout the channel-stationary
    operator writes ([ceil(M / 128)][K / 16][128][16]).  Only the source addresses of the X tiles differ from evo_linear_mfma_bf16 --
    the result, residual included, must be that kernel's bit for bit on the 256-row tiles; the last M % 256 rows go through the
    weight-streaming kernel (<= 16 rows) or the ordinary dense layer and are checked against fp64."""
    ops = _ops()
    DEV = "cuda:0"
    g = torch.Generator(device=DEV).manual_seed(41)
    y = torch.randn(M, K, generator=g, device=DEV).bfloat16()
    w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).bfloat16()
    res = torch.randn(M, N, generator=g, device=DEV).bfloat16()
    nrb = (M + 127) // 128
    pad = torch.zeros(nrb * 128, K, dtype=torch.bfloat16, device=DEV)
    pad[:M] = y
    y_blk = pad.view(nrb, 128, K // 16, 16).permute(0, 2, 1, 3).contiguous()
    assert torch.equal(ops.yblk_to_rows(y_blk, M), y)
    bias = torch.randn(N, generator=g, device=DEV).bfloat16() if N <= 512 else None          # (with and without the epilogue's bias)
    got = ops.linear_residual_yblk_(res.clone(), y_blk, w, bias=bias)
    Mf = M // 256 * 256
    want = ops.linear_mfma(y[:Mf].contiguous(), w, bias, res[:Mf].clone())
    assert torch.equal(got[:Mf], want)
    ref = (y.double() @ w.double().t() + res.double()) + (0 if bias is None else bias.double())
    err = (got.double() - ref).abs()
    assert (err <= ref.abs() * 2.0 ** -8 + 2e-3 * float(ref.abs().max())).all()


# ---- RMSNorm folded into the dense layers around it (gemmr_bf16_kernel NF; include/evo_mi355x.h) ----------------------------------
def _pow2_rows(M, g):
    """A per-row factor that is a power of two (2^-3 .. 2^3): scaling by it commutes with every rounding, so a launch with this row
    factor must equal the plain launch's result times the factor BIT FOR BIT (no bias) -- an exact check of which row gets which factor."""
    e = torch.randint(-3, 4, (M,), device="cuda", generator=g)
    return torch.ldexp(torch.ones(M, device="cuda"), e).float()


@pytest.mark.parametrize("M,N,K,bias", [(512, 256, 128, False), (1026, 768, 256, True), (2049, 4096, 4096, True), (4104, 512, 1024, False)])
def test_stream_writing_dense_layer_emits_the_rows_rms_factor(M, N, K, bias):
    """evo_linear_mfma_nf_bf16 (sumsq) + evo_rms_finalize_f32 through ops.linear_residual_stats_: the residual update equals the plain
    launch bit for bit, and rstd[m] = 1 / (rms(updated row m) + eps) for EVERY row -- main rows from the epilogue's partial sums,
    the sliver rows (M % 256 <= 16: the weight-streaming launch) from the rows themselves."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16) if bias else None
    res = (torch.randn(M, N, device="cuda", generator=g) * 3).to(torch.bfloat16)
    want = ops.linear_residual_(res.clone(), x, w, mfma=True, bias=b)
    got = res.clone()
    eps = 1e-6
    rstd = ops.linear_residual_stats_(got, x, w, b, eps)
    assert torch.equal(got, want)
    ref = 1.0 / (got.double().pow(2).sum(-1).sqrt() * N ** -0.5 + eps)
    rel = ((rstd[:M].double() - ref).abs() / ref).max().item()
    assert rstd.dtype == torch.float32 and rstd.numel() >= M and rel < 2e-6, rel
    again = res.clone()
    assert torch.equal(ops.linear_residual_stats_(again, x, w, b, eps)[:M], rstd[:M])      # fixed summation order: reproducible


@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (1032, 512, 256), (4104, 4096, 4096)])
def test_blocked_input_dense_layer_emits_the_rows_rms_factor(M, N, K):
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 1)
    y = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    yb = ops.yblk_empty(M, K, "cuda")
    yb.zero_()
    nrb = yb.shape[0]
    ypad = torch.zeros(nrb * 128, K, dtype=torch.bfloat16, device="cuda")
    ypad[:M] = y
    yb.copy_(ypad.view(nrb, 128, K // 16, 16).permute(0, 2, 1, 3))
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    want = ops.linear_residual_yblk_(res.clone(), yb, w, bias=b)
    got = res.clone()
    rstd = ops.linear_residual_yblk_stats_(got, yb, w, b, 1e-6)
    assert torch.equal(got, want)
    ref = 1.0 / (got.double().pow(2).sum(-1).sqrt() * N ** -0.5 + 1e-6)
    assert ((rstd[:M].double() - ref).abs() / ref).max().item() < 2e-6


@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (1026, 768, 256), (2049, 12288, 4096)])
def test_row_factor_in_the_dense_layers_epilogue(M, N, K):
    """evo_linear_mfma_nf_bf16 (row_scale) through ops.linear_rs: (a) with power-of-two factors and no bias the result is the plain
    launch's, scaled -- bit for bit, every row; (b) with the real factors, the folded weight and a bias: within one bf16 rounding of
    r_m (W diag(g)) x_m + b in fp64, and as close to the fp64 norm -> dense layer as the two-pass form (rmsnorm, then the plain launch)."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M * 3 + N + K)
    x = (torch.randn(M, K, device="cuda", generator=g) * 2).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    scale = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).to(torch.bfloat16)
    eps = 1e-6
    Mm = ops._nf_main_rows(M)
    p2 = torch.ones((M + 255) // 256 * 256, device="cuda")
    p2[:M] = _pow2_rows(M, g)
    ones = torch.ones(K, dtype=torch.bfloat16, device="cuda")
    plain = ops.linear_mfma(x[:Mm].contiguous(), w, None)
    got = ops.linear_rs(x, p2, w, None, w, ones, eps)
    assert torch.equal(got[:Mm].float(), plain.float() * p2[:Mm, None])
    rstd = ops.rms_finalize(None, x, 0, eps)
    wf = ops.fold_norm_scale(w, scale)
    got = ops.linear_rs(x, rstd, wf, b, w, scale, eps)
    exact = rstd[:M, None].double() * (x.double() @ wf.double().t()) + b.double()
    err = (got.double() - exact).abs()
    assert bool((err[:Mm] <= exact[:Mm].abs() * 2.0 ** -8 + 1e-3).all()), err[:Mm].max().item()
    xd = x.double()
    truth = ((xd / (xd.pow(2).sum(-1, keepdim=True).sqrt() * K ** -0.5 + eps)) * scale.double()) @ w.double().t() + b.double()
    two_pass = ops.linear(ops.rmsnorm(x, None, scale, eps), w, b, mfma=True)
    e_f, e_t = (got.double() - truth).norm() / truth.norm(), (two_pass.double() - truth).norm() / truth.norm()
    print(f"[norm folded {M}x{N}x{K}] rel-L2 vs fp64 norm -> dense: folded {e_f:.3e}, two-pass {e_t:.3e}")
    assert e_f <= 1.15 * e_t + 1e-5
    err_t = (got.double() - truth).abs()
    assert bool((err_t <= truth.abs() * 2.0 ** -7 + 2e-2).all())                      # (sliver rows included: the norm-folding weight-streaming launch)


@pytest.mark.parametrize("M,I,K", [(512, 128, 128), (1026, 256, 256), (2049, 11008, 4096)])
def test_row_factor_in_the_gated_launch(M, I, K):
    """evo_mlp_gate_mfma_nf_bf16 through ops.mlp_gate_rs: unit factors == the plain gated launch bit for bit; real factors + folded
    weight vs the two-pass form (rmsnorm, gated launch) against fp64."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + I + K)
    x = (torch.randn(M, K, device="cuda", generator=g) * 2).to(torch.bfloat16)
    w12 = (torch.randn(2 * I, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    scale = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).to(torch.bfloat16)
    eps = 1e-6
    Mm = ops._nf_main_rows(M)
    w12g = ops.pack_gate_weights(w12)
    one = torch.ones((M + 255) // 256 * 256, device="cuda")
    ones = torch.ones(K, dtype=torch.bfloat16, device="cuda")
    plain = ops.mlp_gate(x[:Mm].contiguous(), w12, w12g=w12g)
    assert torch.equal(ops.mlp_gate_rs(x, one, w12g, w12, ones, eps)[:Mm], plain)
    rstd = ops.rms_finalize(None, x, 0, eps)
    got = ops.mlp_gate_rs(x, rstd, ops.pack_gate_weights(ops.fold_norm_scale(w12, scale)), w12, scale, eps)
    xd = x.double()
    n = (xd / (xd.pow(2).sum(-1, keepdim=True).sqrt() * K ** -0.5 + eps)) * scale.double()
    z = n @ w12.double().t()
    truth = torch.nn.functional.gelu(z[:, :I]) * z[:, I:]
    two_pass = ops.mlp_gate(ops.rmsnorm(x, None, scale, eps), w12, w12g=w12g)
    e_f, e_t = (got.double() - truth).norm() / truth.norm(), (two_pass.double() - truth).norm() / truth.norm()
    print(f"[norm folded, gated {M}x{I}x{K}] rel-L2 vs fp64: folded {e_f:.3e}, two-pass {e_t:.3e}")
    assert e_f <= 1.15 * e_t + 1e-5


@pytest.mark.parametrize("B,T,N,K", [(2, 513, 768, 256), (3, 1025, 768, 128), (8, 8193, 12288, 4096)])
def test_row_factor_and_stream_rows_in_the_transposed_projection(B, T, N, K):
    """evo_linear_t_mfma_nf_bf16 through ops.linear_t_rs (tail form of z^T): the launch reads its token rows from the stream in
    (batch row, token) order and scales by the token's factor.  (a) power-of-two factors, no bias: every position of z^T equals the
    plain swapped launch on a hand-gathered copy of the rows, scaled, bit for bit (which row, which factor); the tail block too;
    (b) real factors: against linear_t(rmsnorm_rows(x)) and fp64."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(B * T + N + K)
    M = B * T
    x = (torch.randn(M, K, device="cuda", generator=g) * 2).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    scale = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).to(torch.bfloat16)
    eps = 1e-6
    Tm, Tp, Mp, r = ops.zt_layout(B, T)
    assert r > 0 and Tp == Tm
    p2 = torch.ones((M + 255) // 256 * 256, device="cuda")
    p2[:M] = _pow2_rows(M, g)
    ones = torch.ones(K, dtype=torch.bfloat16, device="cuda")
    xm = x.view(B, T, K)[:, :Tm].reshape(Mp, K).contiguous()                          # the main rows, gathered by hand
    plain = torch.empty(Mp // 256, N, 256, dtype=torch.bfloat16, device="cuda")
    from evo_amd.ops import _check, _stream
    _check(ops.lib.evo_linear_t_mfma_bf16(xm.data_ptr(), w.data_ptr(), None, plain.data_ptr(), Mp, N, K, _stream()), "evo_linear_t_mfma_bf16")
    zt = ops.linear_t_rs(x, p2, w, None, w, ones, eps, B, T)
    fac = p2[:M].view(B, T)[:, :Tm].reshape(Mp // 256, 1, 256)
    assert torch.equal(zt[:-1].float(), plain.float() * fac)
    rstd = ops.rms_finalize(None, x, 0, eps)
    zt = ops.linear_t_rs(x, rstd, ops.fold_norm_scale(w, scale), b, w, scale, eps, B, T)
    two_pass = ops.linear_t(ops.rmsnorm_rows(x, scale, eps, B, T), w, b, B, T)
    xd = x.double()
    truth = ((xd / (xd.pow(2).sum(-1, keepdim=True).sqrt() * K ** -0.5 + eps)) * scale.double()) @ w.double().t() + b.double()   # [M, N]
    bb = torch.arange(B, device="cuda")[:, None].expand(B, T).reshape(-1)
    tt = torch.arange(T, device="cuda")[None, :].expand(B, T).reshape(-1)
    pos = ops.zt_positions(B, T, bb, tt)

    def rows(z):                                                                       # z^T -> [M, N] in (batch row, token) order
        return z.permute(0, 2, 1).reshape(-1, N)[pos].double()
    e_f, e_t = (rows(zt) - truth).norm() / truth.norm(), (rows(two_pass) - truth).norm() / truth.norm()
    print(f"[norm folded, z^T {B}x{T}x{N}x{K}] rel-L2 vs fp64: folded {e_f:.3e}, two-pass {e_t:.3e}")
    assert e_f <= 1.15 * e_t + 1e-5
    ta, tb = rows(zt)[tt >= Tm], rows(two_pass)[tt >= Tm]                              # tail tokens: the weight-streaming launches on the same rows
    assert bool(((ta - tb).abs() <= tb.abs() * 2.0 ** -7 + 1e-2).all())
