#!/usr/bin/env python
"""FASTA -> TSV of per-sequence log-likelihood scores (counterpart of the reference's `scripts/score.py`
[REF scripts/score.py:17-62]: same flags, same output columns; sequences are bucketed by length before batching).

    python -m scripts.score --input-fasta in.fa --output-tsv out.tsv --model-name evo-1-8k-base --device cuda:0
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description="Score sequences with an Evo model on MI355X")
    ap.add_argument("--input-fasta", required=True)
    ap.add_argument("--output-tsv", required=True)
    ap.add_argument("--model-name", default="evo-1-8k-base")
    ap.add_argument("--batch-size", type=int, default=32)          # [REF scripts/score.py:28]
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--weights", default=None, help='checkpoint directory, or "synthetic"')
    ap.add_argument("--reduce-method", default="mean", choices=["mean", "sum"])
    args = ap.parse_args()

    import evo_amd
    from evo_amd.fasta import length_buckets, read_fasta
    m = evo_amd.Evo(args.model_name, device=args.device, weights=args.weights)
    m.model.eval()
    records = list(read_fasta(args.input_fasta))
    seqs = [s for _, s in records]
    scores = [None] * len(seqs)
    for idxs in length_buckets(seqs, args.batch_size):
        got = evo_amd.score_sequences([seqs[i] for i in idxs], m.model, m.tokenizer,
                                      reduce_method=args.reduce_method, device=args.device)
        for i, s in zip(idxs, got):
            scores[i] = float(s)
    with open(args.output_tsv, "w") as f:
        f.write("seqs\tscores\n")
        for s, sc in zip(seqs, scores):
            f.write(f"{s}\t{sc}\n")


if __name__ == "__main__":
    main()
