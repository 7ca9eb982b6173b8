"""GPU: the tail-split routing of the Hyena block (ops.hyena_tail_split) against the ragged-tile routing over a handful of (B, T = 512 k + 1) shapes at\nD = 4096, 4 layers: the main rows and the scoring log-probs must be the same bits, the last row close (python tools/fuzz_tail_split.py)."""
import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd.sh.model import StripedHyena
from evo_amd.synthetic import synthetic_state_dict
from evo_amd.scoring import score_logprobs_device
DEV = "cuda:0"
cfgd = dict(vocab_size=512, hidden_size=4096, num_layers=4, attn_layer_idxs=[1], num_attention_heads=32)
m = StripedHyena(dict(cfgd)); m.load_state_dict(synthetic_state_dict(m, seed=5, device=DEV), strict=True)
m.to_bfloat16_except_poles_residues(); m = m.to(DEV)
def ids(B, L, seed=9):
    rows = [np.random.default_rng(seed + b).choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L) for b in range(B)]
    x = torch.from_numpy(np.stack(rows).astype(np.int64))
    return torch.cat([torch.zeros(B, 1, dtype=torch.long), x], 1).to(DEV)
with torch.inference_mode():
    for B, L in ((8, 512), (3, 2048), (8, 1536), (7, 1024), (2, 4096), (5, 512), (1, 1024)):
        x = ids(B, L)
        out = {}
        for split in (True, False):
            m.ops.hyena_tail_split = split
            out[split] = (m(x)[0].float(), score_logprobs_device(m, x)[0].float())
        m.ops.hyena_tail_split = True
        same_main = torch.equal(out[True][0][:, :L], out[False][0][:, :L])
        d_last = ((out[True][0][:, L:] - out[False][0][:, L:]).norm() / out[False][0][:, L:].norm()).item()
        same_lp = torch.equal(out[True][1], out[False][1])   # log-probs exclude the last position's prediction
        fin = bool(torch.isfinite(out[True][0]).all())
        print(f"B={B} T={L+1}: main rows bitwise {same_main}, last row rel {d_last:.2e}, scoring log-probs bitwise {same_lp}, finite {fin}")
        assert same_main and same_lp and fin and d_last < 3e-2
print("fuzz OK")
