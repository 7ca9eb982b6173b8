"""CPU: the C-ABI shared library builds, loads, and exports every symbol include/evo_mi355x.h declares;
the ctypes table in evo_amd/ops.py covers exactly that set; the product path refuses to run without a GPU
instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from evo_amd import _build
from evo_amd import ops as evo_ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "evo_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(evo_\w+)\s*\(", text)))


def test_header_table_and_exports_agree():
    declared = _declared()
    assert declared, "no declarations parsed"
    assert sorted(_build.EXPORTS) == declared
    assert sorted(evo_ops._SIGNATURES) == declared


def test_library_builds_and_exports_every_symbol():
    path = _build.build()
    lib = ctypes.CDLL(str(path))
    for name in _declared():
        assert hasattr(lib, name), name
    lib.evo_abi_version.restype = ctypes.c_int
    assert lib.evo_abi_version() == evo_ops.ABI_VERSION == int(re.search(r"#define EVO_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "evo_mi355x.h")).read()).group(1))
    assert evo_ops.load_library() is not None


def test_argument_validation_needs_no_gpu():
    lib = evo_ops.load_library()
    # bad shapes are rejected on the host before any launch
    assert lib.evo_rmsnorm_bf16(None, None, None, None, 4, 7, 1e-6, None) == -1
    assert lib.evo_hyena_apply(None, None, None, None, None, None, None, None, None, None, 1, 16, 256, 3, 64, None) == -1
    assert lib.evo_hyena_apply(None, None, None, None, None, None, None, None, None, None, 1, 16, 256, 2, 6, None) == -1
    assert lib.evo_hyena_seg_state(None, None, None, None, None, None, None, 1, 16, 256, 2, 6, None) == -1
    assert lib.evo_embed_bf16(None, None, None, 4, 12, 512, None, None) == -1          # D not a multiple of 8
    assert lib.evo_embed_bf16(None, None, None, 4, 16, 0, None, None) == -1            # empty vocabulary
    assert lib.evo_unembed_logprob_bf16(None, None, None, None, None, 8, 256, 4096, None) == -1    # V must be 512
    assert lib.evo_unembed_logprob_bf16(None, None, None, None, None, 8, 512, 100, None) == -1     # K % 32
    assert lib.evo_linear_mfma_bf16(None, None, None, None, None, 16, 256, 96, None) == -1           # K % 64
    assert lib.evo_linear_mfma_bf16(None, None, None, None, None, 16, 100, 64, None) == -1           # N % 256
    assert lib.evo_gelu_gate_bf16(None, None, 4, 12, None) == -1
    assert lib.evo_mlp_gate_mfma_bf16(None, None, None, 256, 100, 128, None) == -1                   # (2 I) % 256
    assert lib.evo_mlp_gate_mfma_bf16(None, None, None, 256, 128, 96, None) == -1                    # K % 64
    one = ctypes.c_void_p(16)                                                                       # (a non-null, 16-byte aligned pointer value: never dereferenced)
    # evo_hyena_ct(..., B, T, D, H, zt_pitch, row_pitch, zt_row0, tail_T, tail_pos0, state_only, y_blocked_rows, y_row0, y_row_pitch, stream)
    assert lib.evo_hyena_ct(one, None, None, None, None, one, None, None, None, 2, 100, 256, 2, 256, 100, 0, 0, 0, 0, 0, 0, 0, None) == -1   # row pitch % 8
    assert lib.evo_hyena_ct(one, None, None, None, None, one, None, None, None, 2, 100, 256, 2, 200, 104, 0, 0, 0, 0, 0, 0, 0, None) == -1   # z^T not whole 256-position blocks
    assert lib.evo_hyena_ct(one, None, None, None, None, one, None, None, None, 3, 100, 256, 2, 256, 104, 0, 0, 0, 0, 0, 0, 0, None) == -1   # rows beyond z^T
    assert lib.evo_hyena_ct(one, None, None, None, None, one, None, None, None, 2, 100, 256, 2, 256, 104, 4, 0, 0, 0, 0, 0, 0, None) == -1   # first position % 8
    assert lib.evo_hyena_ct(one, None, None, None, None, None, None, None, None, 2, 100, 256, 2, 256, 104, 0, 0, 0, 1, 0, 0, 0, None) == -1  # state-only without s_out
    assert lib.evo_hyena_ct(one, None, None, None, None, one, None, None, None, 2, 513, 256, 2, 1280, 512, 0, 500, 1024, 0, 0, 0, 0, None) == -1   # tail_T % 512
    assert lib.evo_hyena_ct(one, None, None, None, None, one, None, None, None, 2, 530, 256, 2, 1280, 512, 0, 512, 1024, 0, 0, 0, 0, None) == -1   # more than 8 tail tokens
    assert lib.evo_hyena_ct(one, None, None, None, None, one, None, None, None, 2, 513, 256, 2, 1280, 512, 0, 512, 1000, 0, 0, 0, 0, None) == -1   # tail block inside the main area
    assert lib.evo_hyena_ct(one, None, None, None, None, one, None, None, None, 2, 104, 256, 2, 256, 104, 0, 0, 0, 0, 0, 0, 100, None) == -1   # y_row_pitch below T
    assert lib.evo_linear_t_mfma_bf16(None, None, None, None, 300, 768, 256, None) == -1              # Mp % 256
    assert lib.evo_linear_t_mfma_bf16(None, None, None, None, 512, 700, 256, None) == -1              # N % 256
    assert lib.evo_rmsnorm_rows_bf16(None, None, None, None, 200, 64, 1e-6, 100, 96, 100, 0, None) == -1      # pitch < Tm
    assert lib.evo_rmsnorm_rows_bf16(None, None, None, None, 250, 64, 1e-6, 100, 104, 100, 0, None) == -1     # M % T
    assert lib.evo_rmsnorm_rows_bf16(None, None, None, None, 200, 64, 1e-6, 100, 96, 96, 100, None) == -1     # tail rows inside the main rows
    lib.evo_rope_append_decode_bf16.restype = ctypes.c_int
    assert lib.evo_rope_append_decode_bf16(None, None, None, None, ctypes.c_float(1.0), 1, 32, 128, 8, 8, 8, 8, ctypes.c_float(1.0), None) == -1   # null tensors
    assert lib.evo_attn_fwd_causal_bf16(None, None, None, None, 1, 1, 4, 4, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0, None, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_cpu_fallback():
    from evo_amd.sh.model import StripedHyena
    with pytest.raises(evo_ops.EvoLibraryError):
        evo_ops.HipOps()
    m = StripedHyena(dict(hidden_size=128, num_layers=1, attn_layer_idxs=[], num_attention_heads=1))
    with pytest.raises(evo_ops.EvoLibraryError):
        m(torch.zeros(1, 4, dtype=torch.long))


def test_pack_gate_weights_layout():
    """HipOps.pack_gate_weights: the row order evo_mlp_gate_mfma_bf16 documents in include/evo_mi355x.h -- blocks of 64 rows =
    rows 32 q .. 32 q + 31 of W1 followed by the same rows of W2 (pure host logic, no GPU)."""
    import torch
    I, K = 96, 8
    w12 = torch.arange(2 * I * K, dtype=torch.float32).view(2 * I, K)
    g = evo_ops.HipOps.pack_gate_weights(w12)
    assert g.shape == w12.shape and g.is_contiguous()
    for q in range(I // 32):
        assert torch.equal(g[64 * q: 64 * q + 32], w12[32 * q: 32 * q + 32])                 # W1 rows
        assert torch.equal(g[64 * q + 32: 64 * q + 64], w12[I + 32 * q: I + 32 * q + 32])     # the same rows of W2
    import pytest
    with pytest.raises(ValueError):
        evo_ops.HipOps.pack_gate_weights(torch.zeros(2 * 40, K))                             # inner size not a multiple of 32
