"""GPU (-m gpu): continuous batching of decode streams on the HIP kernels (per-row positions in the rotary table,
the KV append and the split-K decode attention; the step captured in a hipGraph).

bf16 sampling is not bit-reproducible across batch compositions (the split-K partition depends on B), so the check
is self-consistency: the logits the pool recorded for every generated token must be the logits the ordinary
parallel forward assigns at that position of (prompt + generated tokens)."""
import pytest
import torch

from test_gpu_model import DEV, SMALL4, build

pytestmark = pytest.mark.gpu

PROMPTS = ["ACGTACGTAGCTAGCTAGCATCGATCGATGCATGCATGCATGACTAGCTAGCTAGCATGCATCAGTCAGTCAGCATGCA", "GGATTACA",
           "TTTACGATTACAGATTACAGATTACATTT" * 5, "C", "GATTACAGATTCCCGGGAAATTT" * 3, "ACGT" * 40]


@pytest.mark.parametrize("n_slots,use_graph", [(4, True), (3, False), (8, True)])
def test_pool_logits_match_parallel_forward(n_slots, use_graph):
    from evo_amd.pool import DecodePool
    from evo_amd.scoring import prepare_batch
    from evo_amd.tokenizer import CharLevelTokenizer
    tok = CharLevelTokenizer(512)
    cfgd = dict(SMALL4, use_interpolated_rotary_pos_emb=True, rotary_emb_scaling_factor=16)
    _, _, m = build(cfgd)
    pool = DecodePool(m, tok, n_slots=n_slots, top_k=4, top_p=1.0, temperature=0.7, device=DEV, use_graph=use_graph)
    torch.manual_seed(0)
    n_tok = 24
    seqs, scores, owner = pool.generate(PROMPTS, n_tokens=n_tok, n_sample_per_prompt=2)
    assert len(seqs) == 2 * len(PROMPTS) and owner == [i for i in range(len(PROMPTS)) for _ in range(2)]
    assert pool.stats["prefills"] == len(PROMPTS)
    worst = 0.0
    for j, pi in enumerate(owner):
        ids = prepare_batch([PROMPTS[pi]], tok, prepend_bos=False, device=DEV)[0]
        P = ids.shape[1]
        full_ids = torch.cat([ids, pool.last_ids[j: j + 1].to(DEV)], dim=1)
        with torch.inference_mode():
            full = m(full_ids)[0][0].float().cpu()                # [P + n, V]
        want = full[P - 1: P - 1 + n_tok]
        got = pool.last_logits[j]
        worst = max(worst, ((got - want).norm() / want.norm()).item())
    assert worst < 2e-2, worst
    assert all(s == s and s <= 0 for s in scores)
