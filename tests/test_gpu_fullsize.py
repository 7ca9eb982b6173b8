"""GPU (-m gpu): BASELINE-size checks through size-independent properties (the fp64 oracle cannot run at these
sizes in seconds): BASELINE configs[1] = 8 x 8,193 tokens, configs[2] = 1 x 131,073 tokens, D = 4096, H = 32."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from evo_amd.ops import HipOps
    return HipOps()


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm()).item()


def test_attention_full_width_split_property(ops):
    """H = 32, T = 8,193: attention of the LAST 3,000 queries computed alone (q_pos0 offset, as a sequence-parallel
    shard or a cache continuation would) equals the same rows of the full launch; and the first row equals V[0]."""
    g = torch.Generator(device=DEV).manual_seed(0)
    T, H = 8193, 32
    qkv = torch.randn(1, T, 3, H, 128, generator=g, device=DEV).bfloat16()
    full = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0)
    tail = ops.attention(qkv[:, T - 3000:, 0], qkv[:, :, 1], qkv[:, :, 2], T - 3000)
    assert torch.equal(tail, full[:, T - 3000:])                      # same tiles, same order -> bit identical
    assert torch.equal(full[0, 0], qkv[0, 0, 2])                      # one visible key: softmax = 1
    # decode form of the very last row (split-K, different reduction order)
    last = ops.attention_decode(qkv[:, T - 1:, 0], qkv[:, :, 1], qkv[:, :, 2])
    assert rel_l2(last, full[:, T - 1:]) < 4e-3


def test_hyena_131k_full_width_segmentation_invariance(ops):
    """B = 1, T = 131,073, D = 4096: the operator must not depend on how time is cut into segments, and the
    carried state of a two-piece evaluation must reproduce the one-piece result (halo + s0)."""
    g = torch.Generator(device=DEV).manual_seed(1)
    D, H, T = 4096, 32, 131073
    z = torch.randn(1, T, 3 * D, generator=g, device=DEV).bfloat16()
    fir_w = (torch.randn(3 * D, 3, generator=g, device=DEV) * 0.3).bfloat16()
    fir_b = (torch.randn(3 * D, generator=g, device=DEV) * 0.1).bfloat16()
    u = torch.rand(D, 8, generator=g, device=DEV)
    mag = 1.0 - 10.0 ** (-5.0 + 4.0 * u)
    ang = (torch.rand(D, 8, generator=g, device=DEV) * 2 - 1) * math.pi
    poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous()
    res = (torch.randn(D, 8, 2, generator=g, device=DEV) * 0.25).float().contiguous()
    dskip = (torch.randn(D, generator=g, device=DEV) * 0.5).bfloat16()
    prm = (fir_w, fir_b, poles, res, dskip, H)
    y1, s1 = ops.hyena_prefill(z, *prm, want_state=True, seg_len=1024)
    y2, s2 = ops.hyena_prefill(z, *prm, want_state=True, seg_len=256)
    assert torch.isfinite(y1.float()).all()
    assert rel_l2(y2, y1) < 2e-3
    assert (s2 - s1).abs().max() <= 2e-4 * s1.abs().max()
    cut = 70001
    ya, sa = ops.hyena_prefill(z[:, :cut].contiguous(), *prm, want_state=True)
    yb, sb = ops.hyena_prefill(z[:, cut:].contiguous(), *prm, want_state=True, z_halo=z[:, cut - 2:cut].contiguous(), s0=sa)
    assert rel_l2(torch.cat([ya, yb], 1), y1) < 2e-3
    assert (sb - s1).abs().max() <= 2e-4 * s1.abs().max()


def test_scoring_step_8x8193_is_deterministic_and_finite():
    """The bench workload itself (synthetic 7B weights would take too long to build twice; 4 full-width blocks):
    two runs are bit-identical (no atomics / no order-dependent reductions anywhere on the path)."""
    from evo_amd.sh.model import StripedHyena
    from evo_amd.synthetic import synthetic_state_dict
    from evo_amd.scoring import logits_to_logprobs
    m = StripedHyena(dict(vocab_size=512, hidden_size=4096, num_layers=4, attn_layer_idxs=[2], num_attention_heads=32))
    m.load_state_dict(synthetic_state_dict(m, seed=0, device=DEV), strict=True)
    m.to_bfloat16_except_poles_residues()
    m = m.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(2)
    ids = torch.randint(65, 85, (8, 8193), generator=g, device=DEV)
    ids[:, 0] = 0
    a = logits_to_logprobs(m(ids)[0], ids)
    b = logits_to_logprobs(m(ids)[0], ids)
    assert torch.equal(a, b) and torch.isfinite(a).all() and a.shape == (8, 8192)
    assert (a <= 0).all()


@pytest.mark.parametrize("pre", [False, True])
def test_attention_pipelined_kernel_is_reproducible_across_launches_and_query_offsets(ops, pre):
    """Hazard stress for the pipelined kernel: hand-written v_max3 on MFMA accumulators used to
    depend on timing.  Eight launches on the same data must be bit-identical, and query ranges that START on a
    non-diagonal first tile (q_pos0 > 0: the prologue's QK^T feeds the row max directly) must reproduce the same rows
    of the full launch bit for bit, at several offsets.  H = 32, T = 16,385."""
    g = torch.Generator(device=DEV).manual_seed(5)
    T, H = 16385, 32
    qkv = torch.randn(1, T, 3, H, 128, generator=g, device=DEV).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    kw = {}
    if pre:                                                  # round 6: queries pre-scaled by softmax_scale * log2(e), scores taken as exponents
        c = ops.attn_q_scale(128)
        qs = (q.float() * c).bfloat16()
        kw = {"prescaled": True}
    else:
        qs = q
    full = ops.attention(qs, k, v, 0, **kw)
    for _ in range(7):
        assert torch.equal(ops.attention(qs, k, v, 0, **kw), full)
    for off in (256, 1000, 4097, 12289, T - 257):
        part = ops.attention(qs[:, off:], k, v, off, **kw)
        assert torch.equal(part, full[:, off:]), off
    if pre:
        q = qs.float() / c                                  # (the eager check below un-scales the rounded queries)
    # and against eager fp32 attention on a few query rows spread over the sequence (full key range, all heads)
    rows = torch.tensor([0, 63, 64, 255, 256, 4096, 8191, T - 1], device=DEV)
    for h in (0, 13, 31):
        sc = (q[0, rows, h].float() @ k[0, :, h].float().t()) / math.sqrt(128.0)
        sc = sc.masked_fill(torch.arange(T, device=DEV)[None, :] > rows[:, None], float("-inf"))
        ref = torch.softmax(sc, -1) @ v[0, :, h].float()
        assert rel_l2(full[0, rows, h].float(), ref) < 4e-3
