"""`Evo` loader with the reference's interface [REF evo/models.py:13-150]: name -> StripedHyena config ->
weights -> model + CharLevelTokenizer(512).

Weight sources, in order: (1) `weights="synthetic"` (or EVO_AMD_WEIGHTS=synthetic): random weights of
the real shapes, built on the device -- the only source available offline; (2) `weights=<directory>`
(or EVO_AMD_CHECKPOINT_DIR/<model_name>): a local snapshot holding `model.safetensors[.index.json]`;
(3) the HuggingFace hub / HF cache, same repo ids and revision as the reference
[REF evo/models.py:65-71,91-99].  The `backbone.` prefix strip and the tied `unembed.weight` follow
[REF evo/models.py:122-137]."""
import json
import os
import pkgutil
import re

import torch
import yaml

from .sh.model import StripedHyena
from .sh.utils import dotdict
from .tokenizer import CharLevelTokenizer

MODEL_NAMES = [
    "evo-1.5-8k-base",
    "evo-1-8k-base",
    "evo-1-131k-base",
    "evo-1-8k-crispr",
    "evo-1-8k-transposon",
]

HF_MODEL_NAME_MAP = {
    "evo-1.5-8k-base": "evo-design/evo-1.5-8k-base",
    "evo-1-8k-base": "togethercomputer/evo-1-8k-base",
    "evo-1-131k-base": "togethercomputer/evo-1-131k-base",
    "evo-1-8k-crispr": "LongSafari/evo-1-8k-crispr",
    "evo-1-8k-transposon": "LongSafari/evo-1-8k-transposon",
}

_CONFIG_FOR = {name: "configs/evo-1-8k-base_inference.yml" for name in MODEL_NAMES}
_CONFIG_FOR["evo-1-131k-base"] = "configs/evo-1-131k-base_inference.yml"


def _check_name(model_name: str) -> None:
    if model_name not in MODEL_NAMES:
        raise ValueError(f"Invalid model name {model_name}. Should be one of: {', '.join(MODEL_NAMES)}.")


def load_config(config_path: str) -> dotdict:
    """YAML -> dotdict.  The reference passes `Loader=yaml.FullLoader` to dotdict, which lands as a stray
    'Loader' key [REF evo/models.py:142]; nothing reads it, so it is not reproduced."""
    raw = pkgutil.get_data(__name__.rsplit(".", 1)[0], config_path)
    return dotdict(yaml.safe_load(raw))


def read_safetensors_dir(model_dir: str) -> dict:
    """All tensors of a (possibly sharded) safetensors checkpoint directory, `backbone.` prefix removed."""
    from safetensors.torch import load_file
    index_path = os.path.join(model_dir, "model.safetensors.index.json")
    single_path = os.path.join(model_dir, "model.safetensors")
    raw = {}
    if os.path.exists(index_path):
        with open(index_path) as f:
            shards = sorted(set(json.load(f)["weight_map"].values()))
        for shard in shards:
            raw.update(load_file(os.path.join(model_dir, shard)))
    elif os.path.exists(single_path):
        raw = load_file(single_path)
    else:
        raise FileNotFoundError(f"No safetensors files found in {model_dir}. "
                                f"Expected model.safetensors.index.json or model.safetensors.")
    sd = {(k[len("backbone."):] if k.startswith("backbone.") else k): v for k, v in raw.items()}
    if "unembed.weight" not in sd and "embedding_layer.weight" in sd:
        sd["unembed.weight"] = sd["embedding_layer.weight"]
    return sd


def _locate_checkpoint(model_name: str, weights) -> str:
    if weights not in (None, "hub"):
        return str(weights)
    root = os.environ.get("EVO_AMD_CHECKPOINT_DIR")
    if root and os.path.isdir(os.path.join(root, model_name)):
        return os.path.join(root, model_name)
    from huggingface_hub import snapshot_download
    revision = "1.1_fix" if re.match(r"evo-1-.*-base", model_name) else "main"
    return snapshot_download(HF_MODEL_NAME_MAP[model_name], revision=revision)


def load_checkpoint(model_name: str = MODEL_NAMES[1], config_path: str = "configs/evo-1-131k-base_inference.yml",
                    device: str = None, weights=None, seed: int = 0, *args, **kwargs) -> StripedHyena:
    config = load_config(config_path)
    model = StripedHyena(config)
    weights = weights if weights is not None else os.environ.get("EVO_AMD_WEIGHTS")
    if weights == "synthetic":
        from .synthetic import synthetic_state_dict
        state_dict = synthetic_state_dict(model, seed=seed, device=device or "cpu")
    else:
        state_dict = read_safetensors_dir(_locate_checkpoint(model_name, weights))
    model.load_state_dict(state_dict, strict=True)
    model.to_bfloat16_except_poles_residues()
    if device is not None:
        model = model.to(device)
    return model


class Evo:
    def __init__(self, model_name: str = MODEL_NAMES[1], device: str = None, weights=None, seed: int = 0):
        """Loads an Evo model by name.  `weights`: None (HF hub / local cache), a checkpoint directory, or
        "synthetic"."""
        self.device = device
        _check_name(model_name)
        self.model = load_checkpoint(model_name=model_name, config_path=_CONFIG_FOR[model_name], device=self.device,
                                     weights=weights, seed=seed)
        self.tokenizer = CharLevelTokenizer(512)
