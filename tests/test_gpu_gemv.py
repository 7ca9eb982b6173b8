"""GPU (-m gpu): the skinny (M <= 64) weight-streaming dense layers (dot2 and MFMA forms; round 6: 17-64 rows with 2-4 m tiles per weight
pass) against fp64 on the same bf16 inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M", [1, 2, 3, 5, 8, 11, 16, 17, 32, 33, 48, 49, 64])
@pytest.mark.parametrize("N,K", [(12288, 4096), (4096, 10928), (4096, 11008), (512, 4096), (37, 264), (37, 288),
                                 # round 6, the n-split MFMA form (N >= 8192, K % 256 == 0; x rows of a 256-k chunk staged in LDS once per workgroup): four
                                 # waves per workgroup (N >= 16,384), a ragged last n tile with idle waves, one / three chunks of K
                                 (22016, 4096), (8200, 768), (8200, 256)])
@pytest.mark.parametrize("mode", ["plain", "bias", "residual"])
def test_linear_small_m(M, N, K, mode):
    from evo_amd.ops import default_ops
    ops = default_ops()
    if M > 8 and K % 32 != 0:
        pytest.skip("M > 8 needs K % 32 == 0 (library path otherwise)")
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    r = torch.randn(M, N, generator=g).bfloat16()
    ref = x.double() @ w.double().t()
    if mode == "bias":
        got = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV))
        ref = ref + b.double()
    elif mode == "residual":
        rd = r.to(DEV).clone()
        got = ops.linear_residual_(rd, x.to(DEV), w.to(DEV))
        assert got.data_ptr() == rd.data_ptr()
        ref = ref + r.double()
    else:
        got = ops.linear(x.to(DEV), w.to(DEV), None)
    got = got.double().cpu()
    err = (got - ref).abs()
    assert (err <= ref.abs() * 2 ** -8 + 2e-3 * ref.abs().max()).all()
    assert ((got - ref).norm() / ref.norm()).item() < 2e-3


@pytest.mark.parametrize("M,norm", [(1, False), (3, False), (4, True), (8, True), (6, False), (20, False), (20, True)])
def test_mlp_gate_grouped_weight_layout_is_bitwise_the_plain_layout(M, norm):
    """One weight set (round 6, ABI 10 `grouped`): the decode gate launches reading l1 | l2 in the gated MFMA launch's row order (blocks of
    32 rows of W1 | the same 32 of W2: HipOps.pack_gate_weights) give the bits of the plain [W1; W2] order -- weight-streaming launches
    (M <= 4; <= 8 with the norm) and the unfused fallback (dense layer + un-grouping + gate kernel) alike."""
    from evo_amd.ops import default_ops
    ops = default_ops()
    I, K = 11008, 4096
    g = torch.Generator().manual_seed(M * 31 + 5)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w12 = (torch.randn(2 * I, K, generator=g) * (1.5 / K ** 0.5)).bfloat16().to(DEV)
    sc = (1.0 + 0.1 * torch.randn(K, generator=g)).bfloat16().to(DEV)
    w12g = ops.pack_gate_weights(w12)
    kw = dict(norm_scale=sc, eps=1e-6) if norm else {}
    want = ops.mlp_gate(x, w12, **kw)
    got = ops.mlp_gate(x, None, w12g=w12g, **kw)
    assert torch.equal(got, want)


@pytest.mark.parametrize("M", [1, 2, 3, 4, 6])
@pytest.mark.parametrize("I,K", [(11008, 4096), (40, 264), (8, 8)])
def test_mlp_gate_fused_matches_unfused_and_fp64(M, I, K):
    """gelu(x W1^T) * (x W2^T) in one weight-streaming launch (M <= 4) == dense layer + gate kernel, and both within
    bf16 rounding of fp64 (M = 6 exercises the unfused fallback of the same call)."""
    from evo_amd.ops import default_ops
    ops = default_ops()
    g = torch.Generator().manual_seed(M * 77 + I + K)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w12 = (torch.randn(2 * I, K, generator=g) * (1.5 / K ** 0.5)).bfloat16().to(DEV)
    got = ops.mlp_gate(x, w12)
    unfused = ops.gelu_gate(ops.linear(x, w12, None))
    assert got.shape == (M, I) and got.dtype == torch.bfloat16
    # the two paths sum K in different orders: a product may round to the neighbouring bf16, then the gate output too
    d = (got.double() - unfused.double()).abs()
    assert bool((d <= unfused.double().abs() * 2.0 ** -6 + 1e-3 * unfused.double().abs().max()).all())
    p = x.double() @ w12.double().t()
    u, v = p[:, :I].bfloat16().double(), p[:, I:].bfloat16().double()
    want = 0.5 * u * (1 + torch.erf(u / 2 ** 0.5)) * v
    err = (got.double() - want).abs()
    assert bool((err <= want.abs() * 2.0 ** -6 + 2e-3 * want.abs().max()).all())
    assert ((got.double() - want).norm() / want.norm()).item() < 4e-3


@pytest.mark.parametrize("M", [1, 2, 4, 5, 7, 8])
@pytest.mark.parametrize("N,K", [(12288, 4096), (4104, 264)])
def test_norm_linear_fused_is_bitwise_the_two_kernels(M, N, K):
    """RMSNorm folded into the weight-streaming dense layer (M <= 4, or M <= 8 at K = 4096; N > 4096): the normalised row is
    rebuilt with the rmsnorm kernel's own reduction order, so the result equals rmsnorm -> linear bit for bit up to M = 4
    (beyond: the two-kernel path sums on the MFMA; M = 7 at K = 264: the fallback)."""
    from evo_amd.ops import default_ops
    ops = default_ops()
    g = torch.Generator().manual_seed(M * 13 + N + K)
    x = (torch.randn(M, K, generator=g) * 3).bfloat16().to(DEV)
    scale = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(DEV)
    b = torch.randn(N, generator=g).bfloat16().to(DEV)
    got = ops.norm_linear(x, scale, 1e-6, w, b)
    two = ops.linear(ops.rmsnorm(x.clone(), None, scale, 1e-6), w, b)
    if M <= 4 or K != 4096:
        assert torch.equal(got, two)
    else:     # 5-8 rows at K = 4096: the fused launch sums on the VALU, the two-kernel path on the MFMA (other order)
        assert (got.double() - two.double()).abs().max().item() <= 2.0 ** -7 * two.double().abs().max().item()
    xd = x.double()
    n = (scale.double() * xd / (xd.pow(2).mean(-1, keepdim=True).sqrt() + 1e-6)).bfloat16().double()
    want = n @ w.double().t() + b.double()
    assert ((got.double() - want).norm() / want.norm()).item() < 4e-3


@pytest.mark.parametrize("M", [1, 3, 4, 6, 8])
def test_norm_mlp_gate_fused_is_bitwise_norm_then_gate(M):
    from evo_amd.ops import default_ops
    ops = default_ops()
    I, K = 11008, 4096
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, K, generator=g) * 2).bfloat16().to(DEV)
    scale = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().to(DEV)
    w12 = (torch.randn(2 * I, K, generator=g) * (1.5 / K ** 0.5)).bfloat16().to(DEV)
    got = ops.mlp_gate(x, w12, scale, 1e-6)
    two = ops.mlp_gate(ops.rmsnorm(x.clone(), None, scale, 1e-6), w12)
    if M <= 4:
        assert torch.equal(got, two)
    else:     # 5-8 rows: fused = VALU sums, unfused = MFMA sums; both round l1 / l2 to bf16 before the gate
        xd = ops.rmsnorm(x.clone(), None, scale, 1e-6).double()
        u, v = (xd @ w12[:I].double().t()).bfloat16().double(), (xd @ w12[I:].double().t()).bfloat16().double()
        want = torch.nn.functional.gelu(u) * v
        err = (got.double() - want).abs()
        assert bool((err <= want.abs() * 2.0 ** -6 + 2e-3 * want.abs().max()).all())
        assert ((got.double() - want).norm() / want.norm()).item() < 4e-3


def test_linear_residual_with_bias_small_m():
    from evo_amd.ops import default_ops
    ops = default_ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4096, generator=g).bfloat16().to(DEV)
    w = (torch.randn(4096, 4096, generator=g) / 64).bfloat16().to(DEV)
    b = torch.randn(4096, generator=g).bfloat16().to(DEV)
    r = torch.randn(2, 4096, generator=g).bfloat16().to(DEV)
    want = r.double() + x.double() @ w.double().t() + b.double()
    got = ops.linear_residual_(r.clone(), x, w, bias=b)
    assert ((got.double() - want).abs() <= want.abs() * 2.0 ** -8 + 1e-2).all()


@pytest.mark.parametrize("M", [1, 2, 4, 5])
def test_hyena_decode_fused_is_bitwise_the_separate_kernels(M):
    """pre-norm + projections + FIR/modal step + gate in one launch == evo_norm_linear + evo_hyena_step, outputs AND the
    carried states, bit for bit, over several consecutive tokens (M = 5: the fallback of the same call)."""
    import math
    from evo_amd.ops import default_ops
    ops = default_ops()
    D, H = 512, 4
    g = torch.Generator().manual_seed(M)
    scale = (1 + 0.1 * torch.randn(D, generator=g)).bfloat16().to(DEV)
    w = (torch.randn(3 * D, D, generator=g) / D ** 0.5).bfloat16().to(DEV)
    b = (torch.randn(3 * D, generator=g) * 0.1).bfloat16().to(DEV)
    fir_w = (torch.randn(3 * D, 3, generator=g) * 0.3).bfloat16().to(DEV)
    fir_b = (torch.randn(3 * D, generator=g) * 0.1).bfloat16().to(DEV)
    mag = 1.0 - 10.0 ** (-5.0 + 4.0 * torch.rand(D, 8, generator=g))
    ang = (torch.rand(D, 8, generator=g) * 2 - 1) * math.pi
    poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous().to(DEV)
    res = (torch.randn(D, 8, 2, generator=g) * 0.25).float().contiguous().to(DEV)
    dskip = (torch.randn(D, generator=g) * 0.5).bfloat16().to(DEV)
    fs_a = (torch.randn(M, 3 * D, 2, generator=g)).bfloat16().to(DEV)
    st_a = torch.view_as_complex(torch.randn(M, D, 8, 2, generator=g).contiguous()).to(DEV)
    fs_b, st_b = fs_a.clone(), st_a.clone()
    for step in range(3):
        x = (torch.randn(M, D, generator=g) * 2).bfloat16().to(DEV)
        ya = ops.hyena_decode_fused(x, scale, 1e-6, w, b, fs_a, st_a, fir_w, fir_b, poles, res, dskip, H)
        z = ops.linear(ops.rmsnorm(x.clone(), None, scale, 1e-6), w, b)
        yb = ops.hyena_step(z, fs_b, st_b, fir_w, fir_b, poles, res, dskip, H)
        assert torch.equal(ya, yb), f"step {step}"
        assert torch.equal(fs_a, fs_b) and torch.equal(torch.view_as_real(st_a), torch.view_as_real(st_b))


@pytest.mark.parametrize("M", [1, 3, 4, 5, 8])
def test_hyena_decode_fused_full_width(M):
    """D = 4096 (the LDS-staged launches, incl. 5-8 batch rows): outputs and carried states against the separate kernels --
    bit for bit up to M = 4 (same summation order), to bf16 rounding beyond (the separate projection runs on the MFMA)."""
    import math
    from evo_amd.ops import default_ops
    ops = default_ops()
    D, H = 4096, 32
    g = torch.Generator().manual_seed(100 + M)
    scale = (1 + 0.1 * torch.randn(D, generator=g)).bfloat16().to(DEV)
    w = (torch.randn(3 * D, D, generator=g) / D ** 0.5).bfloat16().to(DEV)
    b = (torch.randn(3 * D, generator=g) * 0.1).bfloat16().to(DEV)
    fir_w = (torch.randn(3 * D, 3, generator=g) * 0.3).bfloat16().to(DEV)
    fir_b = (torch.randn(3 * D, generator=g) * 0.1).bfloat16().to(DEV)
    mag = 1.0 - 10.0 ** (-5.0 + 4.0 * torch.rand(D, 8, generator=g))
    ang = (torch.rand(D, 8, generator=g) * 2 - 1) * math.pi
    poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous().to(DEV)
    res = (torch.randn(D, 8, 2, generator=g) * 0.25).float().contiguous().to(DEV)
    dskip = (torch.randn(D, generator=g) * 0.5).bfloat16().to(DEV)
    fs_a = (torch.randn(M, 3 * D, 2, generator=g)).bfloat16().to(DEV)
    st_a = torch.view_as_complex(torch.randn(M, D, 8, 2, generator=g).contiguous()).to(DEV)
    fs_b, st_b = fs_a.clone(), st_a.clone()
    for step in range(2):
        x = (torch.randn(M, D, generator=g) * 2).bfloat16().to(DEV)
        ya = ops.hyena_decode_fused(x, scale, 1e-6, w, b, fs_a, st_a, fir_w, fir_b, poles, res, dskip, H)
        z = ops.linear(ops.rmsnorm(x.clone(), None, scale, 1e-6), w, b)
        yb = ops.hyena_step(z, fs_b, st_b, fir_w, fir_b, poles, res, dskip, H)
        if M <= 4:
            assert torch.equal(ya, yb), f"step {step}"
            assert torch.equal(fs_a, fs_b) and torch.equal(torch.view_as_real(st_a), torch.view_as_real(st_b))
        else:
            tol = 2.0 ** -6
            assert (ya.double() - yb.double()).abs().max().item() <= tol * yb.double().abs().max().item() + 1e-2
            ra, rb = torch.view_as_real(st_a).double(), torch.view_as_real(st_b).double()
            assert (ra - rb).abs().max().item() <= tol * rb.abs().max().item()
            assert (fs_a.double() - fs_b.double()).abs().max().item() <= tol * fs_b.double().abs().max().item()
            fs_b.copy_(fs_a); st_b.copy_(st_a)      # (keep the two histories from drifting apart on rounding differences)


@pytest.mark.parametrize("M", [5, 8, 13, 16, 17, 32, 40, 64])
@pytest.mark.parametrize("layout", ["plain", "grouped"])
@pytest.mark.parametrize("norm", [False, True])
def test_mlp_gate_mfma_small_m_is_bitwise_dense_layer_then_gate(M, layout, norm):
    """Round 6: gelu(x W1^T) * (x W2^T) at 5-64 rows as ONE MFMA weight-streaming launch (csrc/gemv.hip skinny_nw_kernel GATE: a workgroup's four waves hold the
    32 + 32 matching rows of W1 / W2, z2 crosses through LDS) -- bit for bit the dense layer on [W1; W2] followed by the gate kernel,
    in both weight layouts, with and without the norm in front; and within bf16 rounding of fp64."""
    from evo_amd.ops import default_ops
    ops = default_ops()
    I, K = 11008, 4096
    g = torch.Generator().manual_seed(M * 91 + 3)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w12 = (torch.randn(2 * I, K, generator=g) * (1.5 / K ** 0.5)).bfloat16().to(DEV)
    sc = (1.0 + 0.1 * torch.randn(K, generator=g)).bfloat16().to(DEV)
    kw = dict(norm_scale=sc, eps=1e-6) if norm else {}
    args = (x, w12) if layout == "plain" else (x, None)
    if layout == "grouped":
        kw["w12g"] = ops.pack_gate_weights(w12)
    assert ops.gate_small_m_mfma
    got = ops.mlp_gate(*args, **kw)
    xn = ops.rmsnorm(x, None, sc, 1e-6) if norm else x
    want = ops.gelu_gate(ops.linear(xn, w12, None))             # the composition it replaces: dense layer on [W1; W2] (same MFMA form, same k order), gate kernel
    if M > 8 or not norm:                                        # (5-8 rows WITH a norm stay on the dot2 launch that norms for itself: another summation order)
        assert torch.equal(got, want)
        ops.gate_small_m_mfma = False
        try:
            assert torch.equal(ops.mlp_gate(*args, **kw), want)
        finally:
            ops.gate_small_m_mfma = True
    z = xn.double() @ w12.double().t()
    ref = torch.nn.functional.gelu(z[:, :I].bfloat16().double()) * z[:, I:].bfloat16().double()
    err = (got.double() - ref).abs()
    assert (err <= ref.abs() * 2 ** -7 + 2e-3 * ref.abs().max()).all()
