#!/usr/bin/env python
"""Many prompts x many samples with continuous batching -- the `semantic_design.sample_model` job
[REF semantic_design/semantic_design.py:271-400]: prompts (any lengths) from a FASTA / text file, `--n-sample-per-prompt`
generations of `--n-tokens` each, CSV out with the reference's columns (UUID, Prompt, Generated Sequence, Score).

    python -m scripts.sample_many --prompts prompts.fasta --output-csv out.csv --n-sample-per-prompt 8 --n-slots 16
"""
import argparse
import csv
import math
import os
import sys
import uuid

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_prompts(path: str):
    """FASTA (records) or plain text (one prompt per non-empty line)."""
    from evo_amd.fasta import read_fasta
    with open(path) as f:
        head = f.read(1)
    if head == ">":
        return [seq for _, seq in read_fasta(path) if seq.strip()]
    with open(path) as f:
        return [ln.strip() for ln in f if ln.strip()]


def main(argv=None):
    ap = argparse.ArgumentParser(description="Sample many sequences from an Evo model on MI355X (continuous batching)")
    ap.add_argument("--prompts", required=True, help="FASTA or one-prompt-per-line text file")
    ap.add_argument("--output-csv", required=True)
    ap.add_argument("--model-name", default="evo-1-8k-base")
    ap.add_argument("--n-tokens", type=int, default=1000)
    ap.add_argument("--n-sample-per-prompt", type=int, default=1)
    ap.add_argument("--temperature", type=float, default=0.7)
    ap.add_argument("--top-k", type=int, default=4)
    ap.add_argument("--top-p", type=float, default=1.0)
    ap.add_argument("--n-slots", type=int, default=16, help="decode streams advanced together per step")
    ap.add_argument("--prepend-bos", action="store_true")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--weights", default=None)
    args = ap.parse_args(argv)

    import evo_amd
    from evo_amd.pool import DecodePool
    prompts = read_prompts(args.prompts)
    if not prompts:
        raise SystemExit(f"no prompts in {args.prompts}")
    m = evo_amd.Evo(args.model_name, device=args.device, weights=args.weights)
    pool = DecodePool(m.model, m.tokenizer, n_slots=args.n_slots, top_k=args.top_k, top_p=args.top_p,
                      temperature=args.temperature, device=args.device)
    seqs, scores, owner = pool.generate(prompts, n_tokens=args.n_tokens, n_sample_per_prompt=args.n_sample_per_prompt,
                                        prepend_bos=args.prepend_bos)
    rows = [[uuid.uuid4().hex, prompts[o], s, str(sc)] for s, sc, o in zip(seqs, scores, owner)
            if s.strip() and not math.isnan(sc)]              # the reference drops empty / NaN-scored generations
    with open(args.output_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["UUID", "Prompt", "Generated Sequence", "Score"])
        w.writerows(rows)
    print(f"{len(rows)} generations of {args.n_tokens} tokens from {len(prompts)} prompts -> {args.output_csv} "
          f"({pool.stats['steps']} pooled steps, {pool.stats['prefills']} prefills)")
    return rows


if __name__ == "__main__":
    main()
