"""CPU: checkpoint ingestion [REF evo/models.py:91-150] and the FASTA/CLI helpers."""
import json
import os

import pytest
import torch

from evo_amd import models as M
from evo_amd.fasta import length_buckets, read_fasta
from evo_amd.sh.model import StripedHyena
from evo_amd.sh.utils import dotdict
from oracle import stripedhyena_ref as R

TINY = dict(vocab_size=512, hidden_size=128, num_layers=3, attn_layer_idxs=[1], num_attention_heads=1)


def test_model_names_and_errors():
    assert M.MODEL_NAMES[1] == "evo-1-8k-base" and set(M.HF_MODEL_NAME_MAP) == set(M.MODEL_NAMES)
    with pytest.raises(ValueError):
        M.Evo("evo-9000")
    with pytest.raises(FileNotFoundError):
        M.read_safetensors_dir("/nonexistent-dir")


def test_configs_match_reference_values():
    c8, c131 = M.load_config("configs/evo-1-8k-base_inference.yml"), M.load_config("configs/evo-1-131k-base_inference.yml")
    assert c8.hidden_size == 4096 and c8.num_layers == 32 and c8.attn_layer_idxs == [8, 16, 24] and c8.state_size == 8
    assert c8.max_seqlen is None and c8.use_interpolated_rotary_pos_emb is None        # absent keys read as None
    assert c131.use_interpolated_rotary_pos_emb is True and c131.rotary_emb_scaling_factor == 16
    m = StripedHyena(c131)
    assert m.inner_size == 10928 and m.rotary_scaling == 16.0 and len(m.hyena_layer_idxs) == 29
    assert sum(p.numel() for p in m.parameters()) + 3 * 64 == 6_452_781_248


def test_dotdict_like_upstream():
    d = dotdict({"a": 1}, Loader="x")            # the reference passes a stray kwarg [REF evo/models.py:142]
    assert d.a == 1 and d.missing is None and d.Loader == "x"
    d.b = 2
    assert d["b"] == 2


def _write_ckpt(tmp_path, sharded):
    from safetensors.torch import save_file
    sd = R.make_synthetic_state_dict(R.RefConfig.from_dict(TINY), seed=5)
    sd = {("backbone." + k): v.contiguous() for k, v in sd.items() if k != "unembed.weight"}   # HF layout: prefix, tied
    if sharded:
        keys = sorted(sd)
        half = len(keys) // 2
        parts = {"model-00001-of-00002.safetensors": keys[:half], "model-00002-of-00002.safetensors": keys[half:]}
        for fn, ks in parts.items():
            save_file({k: sd[k] for k in ks}, os.path.join(tmp_path, fn))
        with open(os.path.join(tmp_path, "model.safetensors.index.json"), "w") as f:
            json.dump({"weight_map": {k: fn for fn, ks in parts.items() for k in ks}}, f)
    else:
        save_file(sd, os.path.join(tmp_path, "model.safetensors"))
    return sd


@pytest.mark.parametrize("sharded", [False, True])
def test_local_checkpoint_roundtrip(tmp_path, sharded):
    raw = _write_ckpt(str(tmp_path), sharded)
    sd = M.read_safetensors_dir(str(tmp_path))
    assert not any(k.startswith("backbone.") for k in sd)
    assert sd["unembed.weight"] is sd["embedding_layer.weight"]                        # tied [REF evo/models.py:132-137]
    m = StripedHyena(dict(TINY))
    m.load_state_dict(sd, strict=True)
    m.to_bfloat16_except_poles_residues()
    assert m.unembed.weight is m.embedding_layer.weight
    assert m.blocks[0].filter.poles.dtype == torch.float32 and m.blocks[0].projections.weight.dtype == torch.bfloat16
    assert torch.equal(m.blocks[1].inner_mha_cls.Wqkv.weight, raw["backbone.blocks.1.inner_mha_cls.Wqkv.weight"])
    bad = dict(sd)
    bad.pop("blocks.0.filter.D")
    with pytest.raises(RuntimeError):
        StripedHyena(dict(TINY)).load_state_dict(bad, strict=True)
    bad = dict(sd, extra=torch.zeros(1))
    with pytest.raises(RuntimeError):
        StripedHyena(dict(TINY)).load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["norm.scale"] = torch.zeros(3)
    with pytest.raises(RuntimeError):
        StripedHyena(dict(TINY)).load_state_dict(bad, strict=True)


def test_fasta_and_buckets(tmp_path):
    p = tmp_path / "x.fa"
    p.write_text(">a desc\nACGT\nAC\n\n>b\nGG\n>empty\n>c\nTTTTTTTT\n")
    recs = list(read_fasta(str(p)))
    assert recs == [("a", "ACGTAC"), ("b", "GG"), ("empty", ""), ("c", "TTTTTTTT")]
    b = length_buckets([s for _, s in recs], 2)
    assert b == [[3, 0], [1, 2]]


def test_contractive_profile_is_pinned_for_the_7b_dims():
    """The trained-like parity weights are a pure function of (seed, dims, evo_amd/configs/contractive_gains.json): no engine run builds
    them.  The table covers blocks 1..31 of the 7B dims at seed 0, other (seed, dims) fall back to the calibration."""
    import types

    import torch
    from evo_amd.synthetic import apply_contractive_gains, pinned_contractive_gains
    m7 = types.SimpleNamespace(hidden_size=4096, num_layers=32, num_heads=32, inner_size=10928, attn_layer_idxs=[8, 16, 24])
    g = pinned_contractive_gains(m7, 0)
    assert g is not None and sorted(g) == list(range(1, 32)) and all(0.02 < v < 0.2 for v in g.values())
    assert pinned_contractive_gains(m7, 1) is None
    assert pinned_contractive_gains(types.SimpleNamespace(hidden_size=512, num_layers=4, num_heads=4, inner_size=1376, attn_layer_idxs=[2]), 0) is None
    sd = {"blocks.1.mlp.l3.weight": torch.ones(4, 4, dtype=torch.bfloat16), "blocks.1.mlp.l1.weight": torch.ones(4, 4, dtype=torch.bfloat16)}
    apply_contractive_gains(sd, {1: 0.5})
    assert (sd["blocks.1.mlp.l3.weight"] == 0.5).all() and (sd["blocks.1.mlp.l1.weight"] == 1).all()
