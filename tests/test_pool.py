"""CPU: continuous batching of decode streams (evo_amd/pool.py) on the fp64 oracle backend.

Property: with greedy sampling a pool run over prompts of DIFFERENT lengths -- which the reference (and
evo_amd.generate) can only generate one at a time [REF evo/generation.py:237-262] -- reproduces the per-prompt
`generate` output token for token and score for score, whatever the number of slots (1 = sequential, 2 = slots are
re-filled while others are mid-stream, 8 = more slots than jobs)."""
import numpy as np
import pytest
import torch

from golden_common import tiny_model

from evo_amd.generation import generate
from evo_amd.pool import DecodePool, sample_many
from evo_amd.tokenizer import CharLevelTokenizer

TOK = CharLevelTokenizer(512)
PROMPTS = ["ACGTAC", "GG", "TTTACGATTACA", "C", "GATTACAGATT"]


@pytest.fixture(scope="module")
def model_and_reference():
    m = tiny_model()
    want = [generate([p], m, TOK, n_tokens=7, temperature=0.0, top_k=1, cached_generation=True, verbose=0,
                     device="cpu") for p in PROMPTS]
    return m, [w[0][0] for w in want], [float(w[1][0]) for w in want]


@pytest.mark.parametrize("n_slots", [1, 2, 8])
def test_pool_reproduces_per_prompt_generate(model_and_reference, n_slots):
    m, seqs, scores = model_and_reference
    pool = DecodePool(m, TOK, n_slots=n_slots, top_k=1, top_p=1.0, temperature=0.0, device="cpu")
    got_seqs, got_scores, owner = pool.generate(PROMPTS, n_tokens=7)
    assert owner == list(range(len(PROMPTS)))
    assert got_seqs == seqs
    np.testing.assert_allclose(got_scores, scores, rtol=1e-9, atol=1e-12)
    assert pool.stats["prefills"] == len(PROMPTS)
    assert pool.stats["tokens"] == len(PROMPTS) * 6            # the first token of a stream comes from its prefill


def test_samples_of_one_prompt_share_a_prefill(model_and_reference):
    m, seqs, scores = model_and_reference
    pool = DecodePool(m, TOK, n_slots=3, top_k=1, top_p=1.0, temperature=0.0, device="cpu")
    got_seqs, got_scores, owner = pool.generate(PROMPTS[:2], n_tokens=7, n_sample_per_prompt=4)
    assert owner == [0, 0, 0, 0, 1, 1, 1, 1]
    assert got_seqs == [seqs[0]] * 4 + [seqs[1]] * 4           # greedy: the copies agree
    np.testing.assert_allclose(got_scores, [scores[0]] * 4 + [scores[1]] * 4, rtol=1e-9, atol=1e-12)
    assert pool.stats["prefills"] == 2


def test_pool_reuse_and_growth(model_and_reference):
    m, seqs, scores = model_and_reference
    pool = DecodePool(m, TOK, n_slots=2, top_k=1, top_p=1.0, temperature=0.0, device="cpu")
    a, _, _ = pool.generate(PROMPTS[:3], n_tokens=3)
    cap0 = pool.capacity
    b, sb, _ = pool.generate(PROMPTS, n_tokens=7)               # longer job on the same pool: caches grow
    assert pool.capacity >= cap0
    assert [s[:3] for s in seqs[:3]] == a
    assert b == seqs
    np.testing.assert_allclose(sb, scores, rtol=1e-9, atol=1e-12)


def test_sample_many_stochastic_is_well_formed(model_and_reference):
    m, _, _ = model_and_reference
    torch.manual_seed(0)
    prompts, seqs, scores = sample_many(PROMPTS[:3], m, TOK, n_tokens=5, temp=0.7, top_k=4, n_sample_per_prompt=2,
                                        n_slots=4, device="cpu")
    assert prompts == [PROMPTS[0]] * 2 + [PROMPTS[1]] * 2 + [PROMPTS[2]] * 2
    assert len(seqs) == len(scores) == 6 and all(len(s) == 5 for s in seqs)
    assert all(np.isfinite(scores)) and all(s <= 0 for s in scores)


def test_sample_many_cli_writes_reference_csv(tmp_path, monkeypatch, model_and_reference):
    """scripts/sample_many.py end to end (FASTA in, CSV with the reference's columns out) on the oracle backend."""
    import csv
    import types

    import evo_amd
    from scripts import sample_many as cli
    m, seqs, scores = model_and_reference
    monkeypatch.setattr(evo_amd, "Evo", lambda name, device=None, weights=None: types.SimpleNamespace(model=m, tokenizer=TOK))
    fa = tmp_path / "prompts.fa"
    fa.write_text(">p0 first\n" + PROMPTS[0] + "\n>p1\n" + PROMPTS[1] + "\n>blank\n\n")
    out = tmp_path / "out.csv"
    rows = cli.main(["--prompts", str(fa), "--output-csv", str(out), "--n-tokens", "7", "--n-sample-per-prompt", "2",
                     "--top-k", "1", "--temperature", "0.0", "--n-slots", "3", "--device", "cpu"])
    got = list(csv.reader(open(out)))
    assert got[0] == ["UUID", "Prompt", "Generated Sequence", "Score"]
    assert len(got) == 1 + 4 and len(rows) == 4
    assert [r[1] for r in got[1:]] == [PROMPTS[0]] * 2 + [PROMPTS[1]] * 2
    assert [r[2] for r in got[1:]] == [seqs[0]] * 2 + [seqs[1]] * 2
    assert all(len(r[0]) == 32 for r in got[1:]) and len({r[0] for r in got[1:]}) == 4
    np.testing.assert_allclose([float(r[3]) for r in got[1:]], [scores[0]] * 2 + [scores[1]] * 2, rtol=1e-6)


def test_pool_argument_validation_and_degenerate_jobs(model_and_reference):
    m, seqs, scores = model_and_reference
    pool = DecodePool(m, TOK, n_slots=2, top_k=1, top_p=1.0, temperature=0.0, device="cpu")
    assert pool.generate([], n_tokens=5) == ([], [], [])
    with pytest.raises(ValueError):
        pool.generate(["ACGT"], n_tokens=0)
    with pytest.raises(ValueError):
        pool.generate(["ACGT", ""], n_tokens=3)
    one, _, _ = pool.generate(PROMPTS[:3], n_tokens=1)               # every stream ends at its prefill
    assert one == [s[:1] for s in seqs[:3]]
    bos, sc, _ = pool.generate(["", "AC"], n_tokens=3, prepend_bos=True)   # an empty prompt is just the BOS token
    assert len(bos) == 2 and all(len(s) == 3 for s in bos) and all(np.isfinite(sc))
