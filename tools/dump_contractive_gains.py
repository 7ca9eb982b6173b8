"""Runs calibrate_contractive once on the GPU (7B dims, seed 0) and writes the per-block factors it found -- the table
evo_amd/configs/contractive_gains.json pins, so that the parity weights no longer depend on the engine that judges them.
    python tools/dump_contractive_gains.py gpurun_out/contractive_gains.json"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from evo_amd.sh.model import StripedHyena  # noqa: E402
from evo_amd.synthetic import calibrate_contractive, synthetic_state_dict  # noqa: E402

FULL = dict(vocab_size=512, hidden_size=4096, num_layers=32, attn_layer_idxs=[8, 16, 24], num_attention_heads=32)
m = StripedHyena(dict(FULL))
sd = synthetic_state_dict(m, seed=0, device="cuda:0", profile="default")
ratios = calibrate_contractive(m, sd)
gains = calibrate_contractive.last_gains
key = f"seed0_D{m.hidden_size}_L{m.num_layers}_H{m.num_heads}_I{m.inner_size}_attn{'-'.join(map(str, m.attn_layer_idxs))}"
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:  # noqa: BLE001
    commit = ""
out = {key: {"gains": {str(i): g for i, g in gains.items()}, "ratios_after": {str(i): r for i, r in ratios.items()},
             "target": 0.07, "passes": 3, "commit": commit or os.environ.get("EVO_COMMIT", "")}}
path = sys.argv[1] if len(sys.argv) > 1 else "contractive_gains.json"
with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out))
