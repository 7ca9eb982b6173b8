"""GPU (-m gpu): continuous batching of decode streams on the HIP kernels (per-row positions in the rotary table,
the KV append and the split-K decode attention; the step captured in a hipGraph).

bf16 sampling is not bit-reproducible across batch compositions (the split-K partition depends on B), so the tokens are
taken as the pool sampled them and the LOGITS are checked: the logits the pool recorded for every generated token must be
the logits (prompt + generated tokens) gets at that position from
  * the ORACLE's forward in fp64 (oracle/stripedhyena_ref.py; the yardstick -- tolerance: 1.5 x the rel-L2 of the oracle's own
    eager-bf16 restatement, the same rule as tests/test_gpu_model.py), and
  * the engine's ordinary parallel forward (self-consistency of the cached / batched path with the scoring path).
Usage profile: /root/reference/semantic_design/semantic_design.py:271-360 (many prompts, several samples each)."""
import pytest
import torch

from oracle import stripedhyena_ref as R
from test_gpu_model import DEV, SMALL4, build, rel_l2

pytestmark = pytest.mark.gpu

PROMPTS = ["ACGTACGTAGCTAGCTAGCATCGATCGATGCATGCATGCATGACTAGCTAGCTAGCATGCATCAGTCAGTCAGCATGCA", "GGATTACA",
           "TTTACGATTACAGATTACAGATTACATTT" * 5, "C", "GATTACAGATTCCCGGGAAATTT" * 3, "ACGT" * 40]


# the 7B model's width (D = 4096, 32 heads, inner 10,928) at 4 layers: the pool's regime of the weight-streaming kernels -- M = 5 ... 8 rows
# take the LDS-staged fused launches at D = 4096 only, M = 9 ... 16 the MFMA form -- is not reachable with the toy dimensions
WIDE4 = dict(vocab_size=512, hidden_size=4096, num_layers=4, attn_layer_idxs=[1], num_attention_heads=32)


@pytest.mark.parametrize("dims,n_slots,use_graph", [("toy", 4, True), ("toy", 3, False), ("toy", 8, True), ("d4096", 8, True), ("d4096", 12, True)])
def test_pool_logits_match_parallel_forward(dims, n_slots, use_graph):
    from evo_amd.pool import DecodePool
    from evo_amd.scoring import prepare_batch
    from evo_amd.tokenizer import CharLevelTokenizer
    tok = CharLevelTokenizer(512)
    cfgd = dict(SMALL4 if dims == "toy" else WIDE4, use_interpolated_rotary_pos_emb=True, rotary_emb_scaling_factor=16)
    cfg, sd, m = build(cfgd)
    odev = None if dims == "toy" else DEV                       # (the wide oracles run on torch's eager GPU kernels: checker only)
    osd = sd if odev is None else {k: v.to(DEV) for k, v in sd.items()}
    oracle, oracle_bf16 = R.RefStripedHyena(cfg, osd, "fp64", device=odev), R.RefStripedHyena(cfg, osd, "bf16", device=odev)
    pool = DecodePool(m, tok, n_slots=n_slots, top_k=4, top_p=1.0, temperature=0.7, device=DEV, use_graph=use_graph)
    torch.manual_seed(0)
    n_tok = 24
    seqs, scores, owner = pool.generate(PROMPTS, n_tokens=n_tok, n_sample_per_prompt=2)
    assert len(seqs) == 2 * len(PROMPTS) and owner == [i for i in range(len(PROMPTS)) for _ in range(2)]
    assert pool.stats["prefills"] == len(PROMPTS)
    worst = worst_oracle = worst_floor = 0.0
    for j, pi in enumerate(owner):
        ids = prepare_batch([PROMPTS[pi]], tok, prepend_bos=False, device=DEV)[0]
        P = ids.shape[1]
        full_ids = torch.cat([ids, pool.last_ids[j: j + 1].to(DEV)], dim=1)
        with torch.inference_mode():
            full = m(full_ids)[0][0].float().cpu()                # [P + n, V]
        want = full[P - 1: P - 1 + n_tok]
        got = pool.last_logits[j]
        worst = max(worst, ((got - want).norm() / want.norm()).item())
        oid = full_ids.cpu() if odev is None else full_ids
        ref = oracle(oid)[0][0][P - 1: P - 1 + n_tok].cpu()                   # fp64 oracle on the very same tokens
        flo = oracle_bf16(oid)[0][0][P - 1: P - 1 + n_tok].cpu()
        worst_oracle = max(worst_oracle, rel_l2(got, ref))
        worst_floor = max(worst_floor, rel_l2(flo, ref))
    print(f"[pool {dims} {n_slots} slots, graph={use_graph}] recorded logits vs the fp64 oracle: worst rel-L2 {worst_oracle:.3e} "
          f"(eager-bf16 oracle {worst_floor:.3e}); vs the engine's parallel forward {worst:.3e}")
    assert worst_oracle < max(1.5 * worst_floor, 4e-3), (worst_oracle, worst_floor)
    assert worst < 2e-2, worst
    assert all(s == s and s <= 0 for s in scores)
