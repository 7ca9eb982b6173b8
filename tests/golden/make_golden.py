"""Regenerates tests/golden/*.json|npz by running the UNMODIFIED reference host code from /root/reference.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py

What can be pinned against the real reference: everything in `evo/` that does not need
the absent `stripedhyena` / flash-attn arithmetic --
  * evo.tokenizer.CharLevelTokenizer            [REF evo/tokenizer.py]
  * evo.scoring.prepare_batch / logits_to_logprobs / score_sequences / positional_entropies
                                                [REF evo/scoring.py]
  * evo.generation.Generator.generate / generate (the token loop, cache-offset handling, score pairing)
                                                [REF evo/generation.py]
The reference modules are imported with the in-repo `stripedhyena` shim on the path; the model object
they drive is this repo's host model running on the CPU oracle backend (tests/oracle_ops.py) in fp64, so
the fixtures pin the HOST logic of the reference, not its (unavailable) kernels.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "evo_amd", "shim"))
sys.path.insert(0, "/root/reference")

import evo as ref_evo                                   # noqa: E402  the reference package
from evo.scoring import prepare_batch, logits_to_logprobs, score_sequences, positional_entropies  # noqa: E402
from evo.generation import generate as ref_generate, Generator as RefGenerator   # noqa: E402
from evo.tokenizer import CharLevelTokenizer            # noqa: E402

assert ref_evo.__file__.startswith("/root/reference"), ref_evo.__file__

from golden_common import tiny_model, SEQS, PROMPTS     # noqa: E402


def main():
    out = {}
    tok = CharLevelTokenizer(512)
    texts = ["ACGT", "", "acgtN*", "GATTACA" * 3, "~\t\n", "éA"]
    out["tokenize"] = {t: [int(x) for x in tok.tokenize(t)] for t in texts}
    out["detokenize"] = {json.dumps(ids): tok.detokenize(ids) for ids in ([65, 67, 71, 84], [0, 1, 31, 32, 255, 511, 600], [])}
    out["detokenize_batch"] = tok.detokenize_batch(torch.tensor([[65, 66], [0, 300]]))
    out["props"] = dict(vocab_size=tok.vocab_size, eod=tok.eod, eos=tok.eos, pad_id=tok.pad_id, eod_id=tok.eod_id)

    pb = {}
    for name, seqs, bos in (("ragged_bos", SEQS, True), ("ragged_nobos", SEQS, False), ("single", ["ACGTAC"], True)):
        ids, lens = prepare_batch(seqs, tok, prepend_bos=bos, device="cpu")
        pb[name] = dict(seqs=seqs, bos=bos, ids=ids.tolist(), lens=lens, dtype=str(ids.dtype))
    out["prepare_batch"] = pb

    g = torch.Generator().manual_seed(7)
    logits = torch.randn(3, 9, 512, generator=g) * 3
    ids = torch.randint(0, 512, (3, 9), generator=g)
    out["l2l_trim"] = logits_to_logprobs(logits, ids, trim_bos=True).tolist()
    out["l2l_notrim"] = logits_to_logprobs(logits, ids, trim_bos=False).tolist()
    np.savez_compressed(os.path.join(HERE, "l2l_inputs.npz"), logits=logits.numpy(), ids=ids.numpy())

    model = tiny_model()
    out["score_mean"] = [float(x) for x in score_sequences(SEQS, model, tok, reduce_method="mean", device="cpu")]
    out["score_sum"] = [float(x) for x in score_sequences(SEQS, model, tok, reduce_method="sum", device="cpu")]
    out["entropies"] = [e.tolist() for e in positional_entropies(SEQS, model, tok, device="cpu")]

    gen = {}
    # (a) uncached greedy; (b) cached greedy, prompt shorter than the forcing threshold; (c) cached greedy with
    # prompt forcing (threshold 4): exercises the reference's offset jump [REF evo/generation.py:142-145]
    # NOTE: the reference's UNCACHED path (its own default, cached_generation=False) raises
    # UnboundLocalError('prefilled') [REF evo/generation.py:105-132] -- recorded below, not reproduced.
    try:
        ref_generate(PROMPTS, tiny_model(), tok, n_tokens=2, top_k=1, verbose=0, device="cpu", cached_generation=False)
        gen_uncached_error = None
    except Exception as e:  # noqa: BLE001
        gen_uncached_error = type(e).__name__
    gen["uncached_reference_error"] = gen_uncached_error
    for name, kw in (("cached", dict(cached_generation=True)),
                     ("cached_forced", dict(cached_generation=True, force_prompt_threshold=4)),
                     ("cached_bos", dict(cached_generation=True, prepend_bos=True))):
        model = tiny_model()
        seqs, scores = ref_generate(PROMPTS, model, tok, n_tokens=6, top_k=1, verbose=0, device="cpu", **kw)
        gen[name] = dict(seqs=seqs, scores=[float(s) for s in scores])
    model = tiny_model()
    seqs, scores = ref_generate(["ACG", "ACGTT"], model, tok, n_tokens=4, top_k=1, verbose=0, device="cpu",
                                cached_generation=True)
    gen["unbatched"] = dict(seqs=seqs, scores=[float(s) for s in scores])
    # Generator.generate directly: ids, logits, and resuming from the returned cache
    model = tiny_model()
    G = RefGenerator(model, tok, top_k=1, top_p=1.0, temperature=1.0)
    x = torch.tensor([[65, 67, 71, 84, 65]])
    ids1, sc1, cache = G.generate(device="cpu", input_ids=x, num_tokens=5, cached_generation=True,
                                  print_generation=False, verbose=False, stop_at_eos=False)
    ids2, sc2, cache = G.generate(device="cpu", input_ids=ids1[:, -1:], num_tokens=3, print_generation=False,
                                  verbose=False, stop_at_eos=False, inference_params_dict=cache)
    gen["generator"] = dict(ids1=ids1.tolist(), ids2=ids2.tolist(), sc1_sum=float(sc1.double().sum()),
                            sc2_sum=float(sc2.double().sum()),
                            offset=int(cache["mha"].seqlen_offset))
    out["generate"] = gen

    with open(os.path.join(HERE, "host_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "host_golden.json"))


if __name__ == "__main__":
    main()
