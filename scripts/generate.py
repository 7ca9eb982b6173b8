#!/usr/bin/env python
"""Prompted sampling (counterpart of the reference's `scripts/generate.py` [REF scripts/generate.py:17-63]).

    python -m scripts.generate --prompt ACGT --n-samples 2 --n-tokens 100 --model-name evo-1-8k-base
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _flag(v: str) -> bool:
    """Real booleans (the reference's `type=bool` turns any non-empty string into True -- SURVEY.md C-7)."""
    return str(v).lower() in ("1", "true", "yes", "y", "on")


def main():
    ap = argparse.ArgumentParser(description="Generate sequences with an Evo model on MI355X")
    ap.add_argument("--model-name", default="evo-1-8k-base")
    ap.add_argument("--prompt", default="ACGT")
    ap.add_argument("--n-samples", type=int, default=3)
    ap.add_argument("--n-tokens", type=int, default=100)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--top-k", type=int, default=4)
    ap.add_argument("--top-p", type=float, default=1.0)
    ap.add_argument("--cached-generation", type=_flag, default=True)
    ap.add_argument("--batched", type=_flag, default=True)
    ap.add_argument("--prepend-bos", type=_flag, default=False)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--verbose", type=int, default=1)
    args = ap.parse_args()

    import evo_amd
    m = evo_amd.Evo(args.model_name, device=args.device, weights=args.weights)
    m.model.eval()
    seqs, scores = evo_amd.generate([args.prompt] * args.n_samples, m.model, m.tokenizer, n_tokens=args.n_tokens,
                                    temperature=args.temperature, top_k=args.top_k, top_p=args.top_p,
                                    cached_generation=args.cached_generation, batched=args.batched,
                                    prepend_bos=args.prepend_bos, device=args.device, verbose=args.verbose)
    print("Generated sequences:")
    for s in seqs:
        print(s)


if __name__ == "__main__":
    main()
