"""Synthetic StripedHyena weights of the real shapes, built directly on the target
device.  No checkpoint is reachable offline, so bench.py / smoke() / the demo loader run on these.  Poles
sit inside the unit circle with long memory (1-|p| log-uniform in [1e-5, 1e-1]) so the long convolution
is exercised at 131k tokens."""
import math
from typing import Dict

import torch


def synthetic_state_dict(model, seed: int = 0, device=None) -> Dict[str, torch.Tensor]:
    """State dict for `model` (an evo_amd.sh.model.StripedHyena): bf16 except fp32 poles/residues/inv_freq."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    g = torch.Generator(device=device).manual_seed(seed)
    D, L, S = model.hidden_size, model.num_layers, model.state_size
    out_scale = 4.0 / math.sqrt(2.0 * L)

    def rn(shape, std):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std

    sd: Dict[str, torch.Tensor] = {}
    for name, p in model.named_parameters(remove_duplicate=False):
        shape = tuple(p.shape)
        if name == "unembed.weight":
            continue
        if name.endswith("poles"):
            u = torch.rand(D, S, generator=g, device=device)
            mag = 1.0 - 10.0 ** (-5.0 + 4.0 * u)
            ang = (torch.rand(D, S, generator=g, device=device) * 2.0 - 1.0) * math.pi
            t = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], dim=-1).reshape(shape)
        elif name.endswith("residues"):
            pol = sd[name.replace("residues", "poles")].reshape(D, S, 2)
            one_minus = (1.0 - torch.linalg.vector_norm(pol, dim=-1)).clamp_min(1e-6)
            t = (rn((D, S, 2), math.sqrt(1.0 / (2 * S))) * torch.sqrt(one_minus).unsqueeze(-1) * 4.0).reshape(shape)
        elif name.endswith("scale"):
            t = 1.0 + rn(shape, 0.02)
        elif name == "embedding_layer.weight":
            t = rn(shape, 2.0 / math.sqrt(D))
        elif name.endswith("short_filter_weight"):
            t = rn(shape, 0.3)
        elif name.endswith("filter.D"):
            t = rn(shape, 0.5)
        elif name.endswith("bias"):
            t = rn(shape, 0.02)
        elif "projections.weight" in name or "Wqkv.weight" in name:
            t = rn(shape, 0.04)
        elif "out_filter_dense.weight" in name or "out_proj.weight" in name or "l3.weight" in name:
            t = rn(shape, 0.02 * out_scale)
        else:
            t = rn(shape, 0.02)
        sd[name] = t if (name.endswith("poles") or name.endswith("residues")) else t.to(torch.bfloat16)
    sd["unembed.weight"] = sd["embedding_layer.weight"]
    hd = model.head_dim
    for name, _ in model.named_buffers():
        sd[name] = 1.0 / (model.rotary_base ** (torch.arange(0, hd, 2, dtype=torch.float32, device=device) / hd))
    return sd
