"""`stripedhyena.cache` mirror: the two cache records evo's generation loop reads and mutates
[REF evo/generation.py:105-120,138-148]."""
from dataclasses import dataclass, field
from typing import Optional

from torch import Tensor


@dataclass
class InferenceParams:
    """Attention layers: key_value_memory_dict[layer] = [B_max, max_seqlen, 2, H, hd] bf16."""
    max_seqlen: int
    max_batch_size: int
    seqlen_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)
    lengths_per_sample: Optional[Tensor] = None

    def reset(self, max_seqlen, max_batch_size):
        self.max_seqlen = max_seqlen
        self.max_batch_size = max_batch_size
        self.seqlen_offset = 0
        if self.lengths_per_sample is not None:
            self.lengths_per_sample.zero_()


@dataclass
class RecurrentInferenceParams:
    """Hyena layers: fir_state_dict[layer] = [B, 3D, 2] (last two pre-FIR inputs, oldest first),
    state_dict[layer] = [B, D, 8] complex64 (modal state after the last token)."""
    fir_filter_length: int = 3
    state_dim: int = 16
    seqlen_offset: int = 0
    fir_state_dict: dict = field(default_factory=dict)
    state_dict: dict = field(default_factory=dict)
    max_batch_size: int = 1

    def reset(self):
        self.fir_filter_length = 3
        self.state_dim = 16
        self.seqlen_offset = 0
