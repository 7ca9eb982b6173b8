"""`stripedhyena.sample` mirror [REF evo/generation.py:7,162-167]."""
import torch


def modify_logits_for_top_k_filtering(logits: torch.Tensor, top_k: int) -> None:
    """Keep the top_k largest logits of each row, set the rest to -inf (in place)."""
    kth = torch.topk(logits, top_k, dim=-1)[0][..., -1, None]
    logits.masked_fill_(logits < kth, float("-inf"))


def modify_logits_for_top_p_filtering(logits: torch.Tensor, top_p: float) -> None:
    """Drop the low-probability tail whose cumulative mass is <= 1 - top_p (in place)."""
    if top_p <= 0.0 or top_p >= 1.0:
        return
    sorted_logits, sorted_idx = torch.sort(logits, descending=False)
    cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    drop_sorted = cum <= (1.0 - top_p)
    drop = drop_sorted.scatter(1, sorted_idx, drop_sorted)
    logits.masked_fill_(drop, float("-inf"))


def sample(logits: torch.Tensor, top_k: int = 1, top_p: float = 0.0, temperature: float = 1.0) -> torch.Tensor:
    """[B, V] logits -> [B] int64 token ids.  top_k == 1 is greedy; otherwise top-k filter, divide by
    temperature, top-p filter, then one multinomial draw."""
    logits = logits.float()
    if top_k == 1:
        return logits.argmax(dim=-1)
    logits = logits.clone()
    if top_p > 0.0:
        assert top_p <= 1.0, "top-p should be in (0, 1]."
    if top_k > 0:
        modify_logits_for_top_k_filtering(logits, min(top_k, logits.size(-1)))
    if temperature != 1.0 and temperature > 0.0:
        logits /= temperature
    modify_logits_for_top_p_filtering(logits, top_p)
    return torch.multinomial(torch.softmax(logits, dim=-1), num_samples=1).squeeze(dim=-1)
