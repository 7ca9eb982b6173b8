import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


# ---- the 32-layer, D = 4096 engine on synthetic 7B weights, shared by every full-depth GPU test module (built once per session) ----
FULL = dict(vocab_size=512, hidden_size=4096, num_layers=32, attn_layer_idxs=[8, 16, 24], num_attention_heads=32)
FULL_131K = dict(FULL, use_interpolated_rotary_pos_emb=True, rotary_emb_scaling_factor=16)


def _host_mem_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            return int(line.split()[1]) / 1e6
    return 0.0


@pytest.fixture(scope="session")
def full():
    """One synthetic 7B state dict (built on the GPU, 12.9 GB), the HIP models of both yml configs sharing it, and
    a host copy for the fp32 / bf16-faithful CPU oracles."""
    import time

    import torch
    from evo_amd.sh.model import StripedHyena
    from evo_amd.synthetic import synthetic_state_dict
    DEV = "cuda:0"
    t0 = time.time()
    m8 = StripedHyena(dict(FULL))
    sd = synthetic_state_dict(m8, seed=0, device=DEV)
    m8.load_state_dict(sd, strict=True)
    m8.to_bfloat16_except_poles_residues()
    m8 = m8.to(DEV)
    m131 = StripedHyena(dict(FULL_131K))
    m131.load_state_dict(m8.state_dict(), strict=True)       # adopts the same (already packed) tensors
    m131.to_bfloat16_except_poles_residues()
    m131 = m131.to(DEV)
    sd_cpu = {k: v.cpu() for k, v in m8.state_dict().items()}
    print(f"[full-depth fixture] weights in {time.time() - t0:.1f} s; host MemAvailable {_host_mem_gb():.0f} GB, "
          f"{os.cpu_count()} cpus, torch threads {torch.get_num_threads()}")
    return dict(m8=m8, m131=m131, sd_cpu=sd_cpu, oracles={})
