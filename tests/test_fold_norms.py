"""CPU: StripedHyena.fold_norms_ (one weight set, round 6) on the oracle backend -- the folded model is the same function up to one bf16
rounding of the weights, through the uncached forward, cached prefill and decode steps; its state_dict is the unit-scale equivalent."""
import numpy as np
import pytest
import torch

from evo_amd.sh.model import StripedHyena
from oracle import stripedhyena_ref as R
from oracle_ops import OracleOps

CFG = dict(vocab_size=512, hidden_size=256, num_layers=4, attn_layer_idxs=[2], num_attention_heads=2)


def _model():
    cfg = R.RefConfig.from_dict(CFG)
    sd = R.make_synthetic_state_dict(cfg, 3)
    for k in list(sd):                                          # norm scales far from 1, so that a missing fold would show
        if k.endswith("norm.scale") and k != "norm.scale":
            sd[k] = (sd[k].float() * torch.linspace(0.5, 1.5, sd[k].numel())).to(sd[k].dtype)
    m = StripedHyena(dict(CFG), ops=OracleOps(torch.float64))
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m.to_bfloat16_except_poles_residues()
    return m


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def test_folded_model_is_the_same_function():
    ids = torch.from_numpy(np.random.default_rng(0).integers(0, 4, size=(2, 40)) + 65)
    m = _model()
    want = m(ids)[0].double()
    c0 = m.initialize_inference_params()
    pre0 = m(ids[:, :33], c0)[0].double()
    c0["mha"].seqlen_offset = c0["hyena"].seqlen_offset = 33
    step0 = m(ids[:, 33:34], c0)[0].double()
    before = m.resident_bytes()
    m.fold_norms_()
    assert m.fold_norms_() is m                                  # idempotent
    for blk in m.blocks:
        assert (blk.pre_norm.scale == 1).all() and (blk.post_norm.scale == 1).all()
    assert m.resident_bytes() <= before
    got = m(ids)[0].double()
    assert _rel(got, want) < 1.5e-2, _rel(got, want)             # one bf16 rounding of every folded weight, through 4 layers
    c1 = m.initialize_inference_params()
    pre1 = m(ids[:, :33], c1)[0].double()
    c1["mha"].seqlen_offset = c1["hyena"].seqlen_offset = 33
    step1 = m(ids[:, 33:34], c1)[0].double()
    assert _rel(pre1, pre0) < 1.5e-2 and _rel(step1, step0) < 1.5e-2
    # the folded state dict IS a model: loaded into a fresh StripedHyena it reproduces the folded forward exactly
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert sd["blocks.0.mlp.l1.weight"].shape == (m.inner_size, 256)
    m2 = StripedHyena(dict(CFG), ops=OracleOps(torch.float64))
    m2.load_state_dict(sd, strict=True)
    m2.to_bfloat16_except_poles_residues()
    assert torch.equal(m2(ids)[0].double(), got)
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd)


def test_row_groups_of_a_stateless_pass():
    """StripedHyena._row_groups (round 6): one pass up to `max_rows_per_pass` rows (the persistent launches' 4 GiB operands) unless a pass of
    >= 64 k rows would leave more than 8 rows beyond a multiple of 256 (they take the fused single-token launches); balanced groups, as few
    as fit, every group >= 32 k rows when a split is a choice."""
    from evo_amd.sh.model import StripedHyena

    class Probe:
        max_rows_per_pass = StripedHyena.max_rows_per_pass
        DECODE_ROWS = StripedHyena.DECODE_ROWS
        _row_groups = StripedHyena._row_groups
    g = Probe()._row_groups
    assert g(8, 8193) == [8] and g(1, 131073) == [1] and g(16, 513) == [16] and g(64, 1000) == [64]      # the bench shapes and small batches: one pass
    assert g(2, 131073) == [1, 1] and g(8, 131073) == [1] * 8                                            # beyond 4 GiB of z^T: a row at a time
    assert g(32, 8193) == [8, 8, 8, 8] and g(22, 8193) == [8, 7, 7]                                      # scripts/score.py's default batch at 8 k nt
    assert g(16, 8193) == [8, 8] and g(9, 8193) == [5, 4]                                                # one pass would fit but leave 16 / 9 sliver rows
    assert g(40, 5000) == [20, 20]                                                                       # no split with <= 8 sliver rows: as few groups as fit
    for B, T in ((32, 8193), (22, 8193), (100, 5000), (7, 70000), (3, 300000)):
        gs = g(B, T)
        assert sum(gs) == B and max(gs) - min(gs) <= 1 and (max(gs) * T <= Probe.max_rows_per_pass or max(gs) == 1)
