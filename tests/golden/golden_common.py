"""Shared by make_golden.py (reference host code) and tests/test_host_golden.py (this repo's host code):
the tiny fp64 model both drive, on the CPU oracle backend."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SEQS = ["ACGTACGTAC", "GATTACA", "TTTTTTTTTTTTT", "C"]
PROMPTS = ["ACGTAC", "GGGTTT"]
TINY = dict(vocab_size=512, hidden_size=32, num_layers=3, attn_layer_idxs=[1], num_attention_heads=2,
            max_seqlen=64)


def tiny_model():
    from oracle.stripedhyena_ref import RefConfig, make_synthetic_state_dict
    from oracle_ops import OracleOps
    from evo_amd.sh.model import StripedHyena
    sd = make_synthetic_state_dict(RefConfig.from_dict(TINY), seed=3)
    m = StripedHyena(dict(TINY), ops=OracleOps(torch.float64))
    m.load_state_dict({k: (v.double() if v.dtype == torch.bfloat16 else v) for k, v in sd.items()})
    return m
