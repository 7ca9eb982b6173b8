"""GPU (-m gpu): the skinny (M <= 16) weight-streaming dense layers (dot2 and MFMA forms) against fp64 on the same
bf16 inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M", [1, 2, 3, 5, 8, 11, 16])
@pytest.mark.parametrize("N,K", [(12288, 4096), (4096, 10928), (4096, 11008), (512, 4096), (37, 264), (37, 288)])
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
