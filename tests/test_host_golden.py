"""CPU: this repo's host-side mirror of the evo API (evo_amd.tokenizer / scoring / generation) against
fixtures produced by the UNMODIFIED reference host code (tests/golden/make_golden.py), driving the same
tiny fp64 model on the CPU oracle backend."""
import json
import os

import numpy as np
import pytest
import torch

from golden_common import PROMPTS, SEQS, tiny_model

from evo_amd.generation import Generator, generate
from evo_amd.scoring import logits_to_logprobs, positional_entropies, prepare_batch, score_sequences
from evo_amd.tokenizer import CharLevelTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "host_golden.json")) as f:
    GOLD = json.load(f)
TOK = CharLevelTokenizer(512)


def test_tokenizer_matches_reference():
    for text, ids in GOLD["tokenize"].items():
        got = TOK.tokenize(text)
        assert [int(x) for x in got] == ids
        assert all(isinstance(x, np.uint8) for x in got)
    for ids_json, text in GOLD["detokenize"].items():
        assert TOK.detokenize(json.loads(ids_json)) == text
    assert TOK.detokenize_batch(torch.tensor([[65, 66], [0, 300]])) == GOLD["detokenize_batch"]
    p = GOLD["props"]
    assert (TOK.vocab_size, TOK.eod, TOK.eos, TOK.pad_id, TOK.eod_id) == (
        p["vocab_size"], p["eod"], p["eos"], p["pad_id"], p["eod_id"])
    assert TOK.tokenize_batch(["AC", "G"]) == [TOK.tokenize("AC"), TOK.tokenize("G")]


def test_prepare_batch_matches_reference():
    for case in GOLD["prepare_batch"].values():
        ids, lens = prepare_batch(case["seqs"], TOK, prepend_bos=case["bos"], device="cpu")
        assert ids.tolist() == case["ids"] and lens == case["lens"] and str(ids.dtype) == case["dtype"]


def test_prepare_batch_non_ascii_rows_misalign_like_reference():
    with pytest.raises(RuntimeError):
        prepare_batch(["é", "AB"], TOK, device="cpu")


def test_logits_to_logprobs_matches_reference():
    z = np.load(os.path.join(HERE, "golden", "l2l_inputs.npz"))
    logits, ids = torch.from_numpy(z["logits"]), torch.from_numpy(z["ids"])
    np.testing.assert_allclose(logits_to_logprobs(logits, ids, trim_bos=True).numpy(), np.array(GOLD["l2l_trim"]),
                               rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(logits_to_logprobs(logits, ids, trim_bos=False).numpy(),
                               np.array(GOLD["l2l_notrim"]), rtol=1e-5, atol=1e-5)


def test_score_sequences_and_entropies_match_reference():
    m = tiny_model()
    np.testing.assert_allclose(score_sequences(SEQS, m, TOK, "mean", device="cpu"), GOLD["score_mean"], rtol=1e-5)
    np.testing.assert_allclose(score_sequences(SEQS, m, TOK, "sum", device="cpu"), GOLD["score_sum"], rtol=1e-5)
    for got, want in zip(positional_entropies(SEQS, m, TOK, device="cpu"), GOLD["entropies"]):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)
    with pytest.raises(ValueError):
        score_sequences(SEQS, m, TOK, reduce_method="median", device="cpu")


@pytest.mark.parametrize("name,kw", [
    ("cached", dict(cached_generation=True, force_prompt_threshold=128)),
    ("cached_forced", dict(cached_generation=True, force_prompt_threshold=4, legacy_offsets=True)),
    ("cached_bos", dict(cached_generation=True, prepend_bos=True, force_prompt_threshold=128)),
])
def test_generate_matches_reference(name, kw):
    seqs, scores = generate(PROMPTS, tiny_model(), TOK, n_tokens=6, top_k=1, verbose=0, device="cpu", **kw)
    assert seqs == GOLD["generate"][name]["seqs"]
    np.testing.assert_allclose(scores, GOLD["generate"][name]["scores"], rtol=1e-5)


def test_generate_unbatched_matches_reference():
    seqs, scores = generate(["ACG", "ACGTT"], tiny_model(), TOK, n_tokens=4, top_k=1, verbose=0, device="cpu",
                            cached_generation=True)
    assert seqs == GOLD["generate"]["unbatched"]["seqs"]
    np.testing.assert_allclose(scores, GOLD["generate"]["unbatched"]["scores"], rtol=1e-5)


def test_generator_resume_matches_reference():
    want = GOLD["generate"]["generator"]
    G = Generator(tiny_model(), TOK, top_k=1, top_p=1.0, temperature=1.0)
    x = torch.tensor([[65, 67, 71, 84, 65]])
    ids1, sc1, cache = G.generate(device="cpu", input_ids=x, num_tokens=5, cached_generation=True,
                                  print_generation=False, stop_at_eos=False)
    ids2, sc2, cache = G.generate(device="cpu", input_ids=ids1[:, -1:], num_tokens=3, print_generation=False,
                                  stop_at_eos=False, inference_params_dict=cache)
    assert ids1.tolist() == want["ids1"] and ids2.tolist() == want["ids2"]
    assert abs(float(sc1.double().sum()) - want["sc1_sum"]) < 1e-3
    assert abs(float(sc2.double().sum()) - want["sc2_sum"]) < 1e-3
    assert int(cache["mha"].seqlen_offset) == want["offset"]


def test_fixes_over_reference():
    """(1) the reference's uncached path raises UnboundLocalError; here it works and equals the cached path.
    (2) full-prompt parallel prefill and correct offsets give the same tokens as token-by-token forcing."""
    assert GOLD["generate"]["uncached_reference_error"] == "UnboundLocalError"
    a, sa = generate(PROMPTS, tiny_model(), TOK, n_tokens=5, top_k=1, verbose=0, device="cpu", cached_generation=False)
    b, sb = generate(PROMPTS, tiny_model(), TOK, n_tokens=5, top_k=1, verbose=0, device="cpu", cached_generation=True)
    c, sc = generate(PROMPTS, tiny_model(), TOK, n_tokens=5, top_k=1, verbose=0, device="cpu", cached_generation=True,
                     force_prompt_threshold=2)
    assert a == b == c
    np.testing.assert_allclose(sa, sb, rtol=1e-6)
    np.testing.assert_allclose(sb, sc, rtol=1e-6)
