"""Scoring entry points with the reference's signatures [REF evo/scoring.py:9-131].

Differences from the reference, all host-side and documented in DESIGN.md:
  * `prepare_batch` assembles the padded id matrix on the host and does ONE host->device copy (the
    reference copies every sequence separately [REF evo/scoring.py:23-31]); the ids are identical.
  * log-softmax / entropy run in fp32 inside one gfx950 kernel (`evo_logprob_entropy`) instead of a
    bf16 `torch.log_softmax` [REF evo/scoring.py:47,119] -- the reference rounds log-probs to 3
    significant digits; pass `bf16_logprobs=True` to reproduce that rounding.
"""
from typing import List, Tuple

import numpy as np
import torch

from .tokenizer import CharLevelTokenizer


def prepare_batch(seqs: List[str], tokenizer: CharLevelTokenizer, prepend_bos: bool = True,
                  device: str = "cuda:0") -> Tuple[torch.Tensor, List[int]]:
    """Tokenise, (optionally) prepend BOS = eod_id, right-pad with pad_id to the longest sequence.
    Returns (input_ids [B, T] int64 on `device`, list of sequence lengths)."""
    seq_lengths = [len(s) for s in seqs]
    longest = max(seq_lengths)
    bos = int(prepend_bos)
    rows = [np.frombuffer(seq.encode(), dtype=np.uint8) for seq in seqs]
    # every row is  BOS? + bytes + (longest - len(seq)) pads; with non-ASCII text the byte count differs
    # from the character count and rows stop lining up -- the reference's torch.cat raises there too
    widths = {bos + r.size + (longest - n) for r, n in zip(rows, seq_lengths)}
    if len(widths) != 1:
        raise RuntimeError("prepare_batch: rows have different token lengths (non-ASCII input?)")
    width = widths.pop()
    ids = np.full((len(seqs), width), tokenizer.pad_id, dtype=np.int64)
    if prepend_bos:
        ids[:, 0] = tokenizer.eod_id
    for i, r in enumerate(rows):
        ids[i, bos:bos + r.size] = r
    return torch.from_numpy(ids).to(device), seq_lengths


def _ops_for(t: torch.Tensor):
    from .ops import default_ops
    return default_ops()


def logits_to_logprobs(logits: torch.Tensor, input_ids: torch.Tensor, trim_bos: bool = True,
                       bf16_logprobs: bool = False) -> torch.Tensor:
    """(batch, length, vocab) logits -> (batch, length) log-likelihood of each provided token.
    With trim_bos the last prediction and the first (BOS) id are dropped so position t scores token t+1."""
    if trim_bos:
        logits = logits[:, :-1]
        input_ids = input_ids[:, 1:]
    assert logits.shape[1] == input_ids.shape[1]
    B, L, V = logits.shape
    if logits.is_cuda:
        lg = logits.reshape(B * L, V)
        if lg.dtype not in (torch.bfloat16, torch.float32):
            lg = lg.float()
        lp, _ = _ops_for(lg).logprob_entropy(lg.contiguous(), input_ids.reshape(-1).to(lg.device))
        out = lp.view(B, L)
    else:   # host tensors (utility use only; the scoring hot path always hands over device logits)
        out = torch.log_softmax(logits.float(), dim=-1).gather(2, input_ids.unsqueeze(-1).long()).squeeze(-1)
    return out.to(torch.bfloat16) if bf16_logprobs else out


def _fused_tail_ok(model, input_ids) -> bool:
    """The fused unembed + log-softmax + gather kernel serves the engine's own model class on the GPU."""
    ops = getattr(model, "ops", None) if hasattr(model, "hidden_states") else None
    return (ops is not None and getattr(ops, "name", "") == "hip-gfx950" and hasattr(ops, "unembed_logprob")
            and ops.unembed_logprob_ok(model.unembed.weight.new_empty(1, model.hidden_size), model.unembed.weight))


def score_logprobs_device(model, input_ids: torch.Tensor, want_entropy: bool = False):
    """What `score_sequences` / `positional_entropies` compute on the device for a BOS-prefixed id matrix [B, T]:
    (log-prob of token t+1 at position t  [B, T-1] f32 | None, entropy of the next-token distribution [B, T-1] f32 |
    None).  On the MI355X engine the unembedding, the log-softmax and the gather run as ONE kernel
    (evo_unembed_logprob_bf16) and the [B, T, 512] logits are never materialised; any other model object takes
    model(ids) -> logits_to_logprobs like the reference [REF evo/scoring.py:80-84,116-121]."""
    B, T = input_ids.shape
    if _fused_tail_ok(model, input_ids):
        with torch.no_grad():
            hid = model.hidden_states(input_ids)                       # [B*T, D] final-norm output
            tgt = torch.full((B, T), -1, dtype=torch.int64, device=hid.device)
            tgt[:, :-1] = input_ids[:, 1:].to(hid.device)
            lp, en = model.ops.unembed_logprob(hid, model.unembed.weight, tgt.reshape(-1),
                                               want_logprob=not want_entropy, want_entropy=want_entropy)
        return (None if lp is None else lp.view(B, T)[:, :-1]), (None if en is None else en.view(B, T)[:, :-1])
    logits, _ = model(input_ids)
    if not want_entropy:
        return logits_to_logprobs(logits, input_ids, trim_bos=True), None
    logits = logits[:, :-1]                                            # BOS was prepended: drop the last prediction
    L, V = logits.shape[1], logits.shape[2]
    if logits.is_cuda:
        lg = logits.reshape(B * L, V).contiguous()
        _, ent = _ops_for(lg).logprob_entropy(lg, None, want_logprob=False, want_entropy=True)
        return None, ent.view(B, L)
    lsm = torch.log_softmax(logits.float(), dim=-1)
    return None, -(lsm.exp() * lsm).sum(-1)


def _reduce(logprobs: np.ndarray, seq_lengths: List[int], reduce_method: str) -> List[float]:
    if reduce_method == "mean":
        fn = np.mean
    elif reduce_method == "sum":
        fn = np.sum
    else:
        raise ValueError(f"Invalid reduce_method {reduce_method}")
    return [fn(logprobs[i][: seq_lengths[i]]) for i in range(len(seq_lengths))]


def score_sequences(seqs: List[str], model, tokenizer: CharLevelTokenizer, reduce_method: str = "mean",
                    device: str = "cuda:0") -> List[float]:
    """Mean (or sum) per-token log-likelihood of each sequence under the model."""
    if reduce_method not in ("mean", "sum"):
        raise ValueError(f"Invalid reduce_method {reduce_method}")
    input_ids, seq_lengths = prepare_batch(seqs, tokenizer, device=device, prepend_bos=True)
    assert len(seq_lengths) == input_ids.shape[0]
    with torch.inference_mode():
        logprobs, _ = score_logprobs_device(model, input_ids)          # (batch, length - 1)
    return _reduce(logprobs.float().cpu().numpy(), seq_lengths, reduce_method)


def positional_entropies(seqs: List[str], model, tokenizer: CharLevelTokenizer,
                         device: str = "cuda:0") -> List[np.ndarray]:
    """Per-position entropy of the next-token distribution, one array (len(seq)) per sequence."""
    input_ids, seq_lengths = prepare_batch(seqs, tokenizer, device=device, prepend_bos=True)
    assert len(seq_lengths) == input_ids.shape[0]
    with torch.inference_mode():
        _, ent = score_logprobs_device(model, input_ids, want_entropy=True)
    ent = ent.float().cpu().numpy()
    out = [ent[i][: seq_lengths[i]] for i in range(len(seq_lengths))]
    assert all(len(s) == len(e) for s, e in zip(seqs, out))
    return out
