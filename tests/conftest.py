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


@pytest.fixture(scope="session")
def contractive():
    """The trained-like ("contractive") synthetic 7B weights (evo_amd/synthetic.py: every block past the first changes the stream by
    ~7 %) on the HIP engine under both yml configs (one set of tensors), shared by the tests that judge the north-star's 1e-3 on them.
    `oracles` is filled lazily by the tests (GPU-executed fp32 / eager-bf16 oracle objects on the same tensors)."""
    import time

    import torch
    from evo_amd.sh.model import StripedHyena
    from evo_amd.synthetic import synthetic_state_dict
    DEV = "cuda:0"
    t0 = time.time()
    m8 = StripedHyena(dict(FULL))
    sd = synthetic_state_dict(m8, seed=0, device=DEV, profile="contractive")
    m8.load_state_dict(sd, strict=True)
    m8.to_bfloat16_except_poles_residues()
    m8 = m8.to(DEV)
    m131 = StripedHyena(dict(FULL_131K))
    m131.load_state_dict(m8.state_dict(), strict=True)
    m131.to_bfloat16_except_poles_residues()
    m131 = m131.to(DEV)
    torch.cuda.synchronize()
    print(f"[contractive fixture] weights in {time.time() - t0:.1f} s")
    return dict(m8=m8, m131=m131, oracles={})


def contractive_oracle(contractive, cfgd, mode):
    """The oracle class (oracle/stripedhyena_ref.py) on the contractive tensors, executed on the GPU by torch's eager kernels -- one
    object per numeric mode for the session; long inputs: score tiles of 2^27 elements, the filter / FFT in chunks of 256 channels
    (the same statements with bounded temporaries)."""
    from oracle import stripedhyena_ref as R
    if mode not in contractive["oracles"]:
        sd = {k: v for k, v in contractive["m131"].state_dict().items()}
        o = R.RefStripedHyena(R.RefConfig.from_dict(cfgd), sd, mode, device="cuda:0")
        o.attn_chunk_elems = 1 << 27
        o.chan_chunk = 256
        contractive["oracles"][mode] = o
    o = contractive["oracles"][mode]
    o.cfg = R.RefConfig.from_dict(cfgd)
    return o
