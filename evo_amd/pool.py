"""Continuous batching of decode streams: the `semantic_design.sample_model` usage profile --
many prompts x `n_sample_per_prompt` generations of `n_tokens` each [REF semantic_design/semantic_design.py:271-360,
121-179].

The reference's `generate` batches only prompts of EQUAL length and otherwise generates one prompt at a time
[REF evo/generation.py:237-262]: a decode step streams all 12.9 GB of weights for a single token.  Here a fixed set
of `n_slots` decode streams shares every step.  Each slot owns one row of every cache tensor (KV rows, FIR history,
modal state) and its own position, held in DEVICE memory: the rotary table, the KV append and the split-K decode
attention all read per-row positions (`evo_attn_decode_bf16` dyn_pos[B]), so streams of different prompt lengths
and different ages advance together, a finished stream's slot is re-filled while the others keep going, and the
step has one shape for the whole job -- it is captured once in a hipGraph and replayed.

Prompts are prefilled one at a time with the ordinary parallel forward (the long-convolution kernel ends with the
exact modal state); the `n_sample_per_prompt` copies of a prompt share ONE prefill, whose caches are replicated
into their slots.  Sampling and scoring follow the reference wrapper verbatim (same `sample`, same shifted
logits/token pairing [REF evo/generation.py:162-167,287]), so with greedy sampling a pool run reproduces per-prompt
`generate` token for token (tests/test_pool.py).
"""
from __future__ import annotations

from collections import deque
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .scoring import logits_to_logprobs, prepare_batch
from .sh.sample import sample


class DecodePool:
    def __init__(self, model, tokenizer, n_slots: int = 8, top_k: int = 4, top_p: float = 1.0,
                 temperature: float = 0.7, device: Optional[str] = None, use_graph: Optional[bool] = None):
        self.model = model
        self.tok = tokenizer
        self.n_slots = int(n_slots)
        self.top_k, self.top_p, self.temperature = top_k, top_p, temperature
        self.device = torch.device(device) if device is not None else model.device
        self.use_graph = (self.device.type == "cuda" and getattr(model, "decode_graph", False)) \
            if use_graph is None else bool(use_graph)
        self.ipd = None
        self.capacity = 0
        self._graph = None
        self.stats = {"steps": 0, "prefills": 0, "tokens": 0}

    # ------------------------------------------------------------------ cache rows
    def _allocate(self, capacity: int) -> None:
        """Pool-wide caches with `capacity` KV rows per slot (grown geometrically, outside any captured graph)."""
        m, S, dev = self.model, self.n_slots, self.device
        if self.ipd is not None and capacity <= self.capacity:
            return
        capacity = max(capacity, 2 * self.capacity)
        old = self.ipd
        ipd = m.initialize_inference_params()
        ipd["mha"].max_batch_size = ipd["hyena"].max_batch_size = S
        ipd["mha"].max_seqlen = capacity
        D, H, hd = m.hidden_size, m.num_heads, m.head_dim
        dt = m.embedding_layer.weight.dtype
        for i in m.attn_layer_idxs:
            kv = torch.zeros(S, capacity, 2, H, hd, dtype=dt, device=dev)
            if old is not None:
                prev = old["mha"].key_value_memory_dict[i]
                kv[:, : prev.shape[1]] = prev
            ipd["mha"].key_value_memory_dict[i] = kv
        for i in m.hyena_layer_idxs:
            if old is not None:
                ipd["hyena"].fir_state_dict[i] = old["hyena"].fir_state_dict[i]
                ipd["hyena"].state_dict[i] = old["hyena"].state_dict[i]
            else:
                ipd["hyena"].fir_state_dict[i] = torch.zeros(S, 3 * D, m.short_filter_length - 1, dtype=dt, device=dev)
                ipd["hyena"].state_dict[i] = torch.zeros(S, D, m.state_size, dtype=torch.complex64, device=dev)
        self.ipd, self.capacity = ipd, capacity
        self._graph = None
        self.pos = torch.zeros(S, dtype=torch.int64, device=dev)
        self.ids = torch.zeros(S, 1, dtype=torch.int64, device=dev)

    def _prefill(self, ids: torch.Tensor):
        """ids [1, P] -> (last-position logits [V] f32, the B = 1 caches of the prompt)."""
        m = self.model
        tmp = m.initialize_inference_params()
        tmp["mha"].max_seqlen = ids.shape[1]
        with torch.inference_mode():
            logits, tmp = m(ids, inference_params_dict=tmp)
        self.stats["prefills"] += 1
        return logits[0, -1].float(), tmp

    def _install(self, slot: int, tmp: dict, P: int) -> None:
        m = self.model
        for i in m.attn_layer_idxs:
            self.ipd["mha"].key_value_memory_dict[i][slot, :P] = tmp["mha"].key_value_memory_dict[i][0, :P]
        for i in m.hyena_layer_idxs:
            self.ipd["hyena"].fir_state_dict[i][slot] = tmp["hyena"].fir_state_dict[i][0]
            self.ipd["hyena"].state_dict[i][slot] = tmp["hyena"].state_dict[i][0].reshape(
                self.ipd["hyena"].state_dict[i][slot].shape)

    # ------------------------------------------------------------------ one token for every slot
    def _step_eager(self) -> torch.Tensor:
        m, mha = self.model, self.ipd["mha"]
        mha.pos_tensor = self.pos
        try:
            h = m.hidden_states(self.ids, self.ipd)
            return m.ops.linear(h, m.unembed.weight, None).view(self.n_slots, m.vocab_size)
        finally:
            mha.pos_tensor = None

    def _step(self) -> torch.Tensor:
        """self.ids [S,1] at positions self.pos [S] -> logits [S, V] f32 (every slot, active or not)."""
        self.stats["steps"] += 1
        with torch.inference_mode():
            if not self.use_graph:
                return self._step_eager().float()
            if self._graph is None:
                # The first step runs eagerly (it also warms the library handles and the M = S GEMM choices) and IS the
                # step: the caches advance in place, so nothing may run twice.  Capture records without executing.
                first = self._step_eager().float()
                torch.cuda.synchronize(self.device)
                self.model._row_index(self.n_slots, self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self._step_eager()
                self._graph = (g, out)
                return first
            g, out = self._graph
            g.replay()
            return out.float()

    # ------------------------------------------------------------------ the job
    def generate(self, prompts: Sequence[str], n_tokens: int = 1000, n_sample_per_prompt: int = 1,
                 prepend_bos: bool = False) -> Tuple[List[str], List[float], List[int]]:
        """Returns (generated strings, mean log-likelihood scores, index of the prompt each came from), in job
        order: prompt 0's samples first."""
        m, tok, S, dev = self.model, self.tok, self.n_slots, self.device
        if hasattr(m, "eval"):
            m.eval()
        n_tokens = int(n_tokens)
        if n_tokens < 1 or int(n_sample_per_prompt) < 1:
            raise ValueError("n_tokens and n_sample_per_prompt must be >= 1")
        prompts = list(prompts)
        if not prompts:
            return [], [], []
        if any((not isinstance(p, str)) or (len(p) == 0 and not prepend_bos) for p in prompts):
            raise ValueError("every prompt must be a non-empty string (or use prepend_bos=True)")
        encoded = [prepare_batch([p], tok, prepend_bos=prepend_bos, device=str(dev))[0] for p in prompts]
        self._allocate(max(e.shape[1] for e in encoded) + n_tokens)
        # job = (output index, prompt index); the copies of one prompt are adjacent so that they share a prefill
        jobs = deque((pi * n_sample_per_prompt + c, pi) for pi in range(len(prompts)) for c in range(n_sample_per_prompt))
        n_jobs = len(jobs)
        out_ids = torch.zeros(n_jobs, n_tokens, dtype=torch.long)
        out_logits = torch.zeros(n_jobs, n_tokens, m.vocab_size, dtype=torch.float32)
        slot_job: List[Optional[int]] = [None] * S
        slot_n = [0] * S
        cached_prefill: Dict[int, tuple] = {}

        def fill(slot: int) -> None:
            j, pi = jobs.popleft()
            if pi not in cached_prefill:
                cached_prefill.clear()                               # keep one prompt's caches alive at a time
                cached_prefill[pi] = self._prefill(encoded[pi])
            last_logits, tmp = cached_prefill[pi]
            P = encoded[pi].shape[1]
            self._install(slot, tmp, P)
            first = sample(last_logits[None], top_k=self.top_k, top_p=self.top_p, temperature=self.temperature)
            out_ids[j, 0] = int(first[0])
            out_logits[j, 0] = last_logits.cpu()
            self.ids[slot, 0] = first[0]
            self.pos[slot] = P                                        # the sampled token sits at position P
            slot_job[slot], slot_n[slot] = j, 1

        done = 0
        while done < n_jobs:
            for s in range(S):
                if slot_job[s] is None and jobs:
                    fill(s)
                    if n_tokens == 1:
                        slot_job[s] = None
                        done += 1
            active = [s for s in range(S) if slot_job[s] is not None]
            if not active:
                continue
            logits = self._step()                                     # [S, V]
            nxt = sample(logits, top_k=self.top_k, top_p=self.top_p, temperature=self.temperature)
            lg_cpu, nxt_cpu = logits.cpu(), nxt.cpu()
            self.ids[:, 0] = nxt
            self.pos += 1                                             # (idle slots drift harmlessly; fill() resets them)
            self.pos.clamp_(max=self.capacity - 1)
            for s in active:
                j, k = slot_job[s], slot_n[s]
                out_ids[j, k] = nxt_cpu[s]
                out_logits[j, k] = lg_cpu[s]
                slot_n[s] = k + 1
                self.stats["tokens"] += 1
                if k + 1 == n_tokens:
                    slot_job[s] = None
                    done += 1

        self.last_ids, self.last_logits = out_ids, out_logits        # (kept for inspection / tests)
        seqs = list(tok.detokenize_batch(out_ids))
        lp = logits_to_logprobs(out_logits, out_ids).float().numpy()   # the reference's shifted pairing
        scores = [float(np.mean(lp[j])) for j in range(n_jobs)]
        owner = [pi for pi in range(len(prompts)) for _ in range(n_sample_per_prompt)]
        return seqs, scores, owner


def sample_many(prompts: Sequence[str], model, tokenizer, n_tokens: int = 1000, temp: float = 0.7, top_k: int = 4,
                top_p: float = 1.0, n_sample_per_prompt: int = 1, n_slots: int = 8, prepend_bos: bool = False,
                device: Optional[str] = None):
    """`semantic_design.run_model` / `sample_model` without the equal-length restriction: (prompts repeated per
    sample, generated sequences, scores)."""
    pool = DecodePool(model, tokenizer, n_slots=n_slots, top_k=top_k, top_p=top_p, temperature=temp, device=device)
    seqs, scores, owner = pool.generate(prompts, n_tokens=n_tokens, n_sample_per_prompt=n_sample_per_prompt,
                                        prepend_bos=prepend_bos)
    return [prompts[i] for i in owner], seqs, scores
