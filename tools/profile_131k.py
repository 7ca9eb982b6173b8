#!/usr/bin/env python
"""One warm-up + two scoring steps of BASELINE configs[2] (evo-1-131k-base, 1 x 131,072 nt) for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof131 -o p -- python tools/profile_131k.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2)
a = ap.parse_args()
from evo_amd.ops import default_ops  # noqa: E402
args = argparse.Namespace(steps_131k=a.steps, sp_timeout=600.0, skip_ab=True, skip_sp_predict=True)
out = bench.bench_131k(args, torch.device("cuda:0"), 0, 1, False, default_ops())
print({k: out[k] for k in ("value", "ms_per_step") if k in out})
