#!/usr/bin/env python
"""BASELINE configs[4]: evo-1-131k-base generation, 8,192-nt prompt -> N new tokens (greedy), recurrent Hyena
state + KV cache, 1 x MI355X.  Reports prefill time and decode tokens/s.
    python tools/bench_generate.py [--new 256] [--batch 1] [--graph]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=8192)
    ap.add_argument("--new", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--model", default="evo-1-131k-base")
    args = ap.parse_args()
    from bench import build_model
    from evo_amd.generation import Generator
    from evo_amd.tokenizer import CharLevelTokenizer
    dev = "cuda:0"
    model = build_model(args.model, dev)
    tok = CharLevelTokenizer(512)
    rng = np.random.default_rng(7)
    ids = torch.from_numpy(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(args.batch, args.prompt)).astype(np.int64)).to(dev)
    g = Generator(model, tok, top_k=1, top_p=1.0, temperature=1.0)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, scores, cache = g.generate(device=dev, input_ids=ids, num_tokens=1, cached_generation=True,
                                        print_generation=False, stop_at_eos=False)
        torch.cuda.synchronize()
        t_prefill = time.perf_counter() - t0
        t0 = time.perf_counter()
        out2, scores2, cache = g.generate(device=dev, input_ids=out[:, -1:], num_tokens=args.new, print_generation=False,
                                          stop_at_eos=False, inference_params_dict=cache)
        torch.cuda.synchronize()
        t_dec = time.perf_counter() - t0
        print(f"[gen rep{rep}] B={args.batch} prompt={args.prompt}: prefill {t_prefill * 1e3:.1f} ms "
              f"({args.batch * args.prompt / t_prefill:.0f} nt/s); decode {args.new} tok in {t_dec * 1e3:.1f} ms = "
              f"{t_dec / args.new * 1e3:.2f} ms/tok, {args.batch * args.new / t_dec:.1f} tok/s; "
              f"offset={cache['mha'].seqlen_offset}")


if __name__ == "__main__":
    main()
