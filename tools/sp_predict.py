#!/usr/bin/env python
"""One rank's COMPUTE of BASELINE configs[3] (evo-1-131k-base scoring, batch 8 x 131,072 nt, the sequence split over 8 ranks) on
ONE GPU: `SequenceParallelScorer` with `evo_amd.sp.StubComm` (exchanges return resident buffers at once) runs exactly the kernels
rank `--rank` of an 8-rank job runs -- 8 rows x 16,385-token shard, two Hyena launches per row group and layer (state-only walk +
the pass seeded with the carried state), Ulysses attention over all 131,073 keys for 4 of the 32 heads and all 8 rows.
Prints per-kernel means, the rank's step time, the bytes a real communicator would have moved, and the predicted 8-GPU rate:

    predicted nt/s (compute only)      = 8 * 131,072 / t_rank
    predicted nt/s (collectives serial) = 8 * 131,072 / (t_rank + bytes_all_to_all / (7 links * LINK_GBS) + n_small * LAT_US)

against the measured single-GPU rate of the same process (batch 1 x 131,072 nt).  LINK_GBS (default 48: one direction of an xGMI
link as RCCL all-to-all sustains it, MI355X_MICROARCH.md quotes ~153 GB/s per link bidirectional peak) and LAT_US (default 30 per
small collective) are stated assumptions, not measurements: no N > 1 run exists on this pool.
    python tools/sp_predict.py [--rank 6] [--steps 2]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sp_predict(model, device, ops, single_ms=None, rank=6, world=8, steps=2, link_gbs=48.0, lat_us=30.0, acgt_ids=None,
               scoring_step=None):
    if acgt_ids is None:
        from bench import acgt_ids, scoring_step
    from evo_amd.ops import KernelTimer
    from evo_amd.sp import SequenceParallelScorer, StubComm
    nt = 131072
    T = nt + 1
    with torch.inference_mode():
        if single_ms is None:
            ids1 = acgt_ids(1, nt, 4321, device)
            scoring_step(model, ids1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                scoring_step(model, ids1)
            torch.cuda.synchronize()
            single_ms = (time.perf_counter() - t0) / steps * 1e3
        ids = acgt_ids(world, nt, 4321, device)
        comm = StubComm(world)
        sc = SequenceParallelScorer(model, rank, world, comm=comm)
        sc.score_logprobs(ids)                                # warm-up (buffers, packs)
        torch.cuda.synchronize()
        for k in comm.bytes:
            comm.bytes[k] = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            sc.score_logprobs(ids)
        torch.cuda.synchronize()
        rank_ms = (time.perf_counter() - t0) / steps * 1e3
        moved = {k: v // steps for k, v in comm.bytes.items()}
        ops.timer = KernelTimer()
        sc.score_logprobs(ids)
        torch.cuda.synchronize()
        ks = ops.timer.summary()
        ops.timer = None
    Tl = (T + world - 1) // world
    D = model.hidden_size
    n_small = 29 * (1 + sc.row_groups)                        # halo send/recv + end-state all-gathers per Hyena layer
    a2a_ms = moved["all_to_all"] / ((world - 1) * link_gbs * 1e9) * 1e3
    small_ms = n_small * lat_us * 1e-3
    comp = world * nt / (rank_ms * 1e-3)
    ser = world * nt / ((rank_ms + a2a_ms + small_ms) * 1e-3)
    one = nt / (single_ms * 1e-3)
    hy = {}
    if "hyena_mfma" in ks:
        n2, ms2 = ks["hyena_mfma"]
        n1, ms1 = ks.get("hyena_mfma_state", (0, 0.0))
        per_layer = max(1, n2 // 29)
        rows = world // per_layer
        b2 = rows * Tl * (3 * D * 2 + D * 2)                 # stage 2: z read + y written
        b1 = rows * Tl * (3 * D * 2)                         # stage 1: z read (8 B/token/channel... 24,576 B/token/layer)
        hy = {"launches_per_layer": per_layer, "stage1_state_only_ms": ms1, "stage2_ms": ms2,
              "algorithmic_bytes_per_token_layer": 57344, "stage1_frac_of_8TBs": b1 / (ms1 * 1e-3) / 8e12 if ms1 else None,
              "stage2_frac_of_8TBs": b2 / (ms2 * 1e-3) / 8e12, "operator_frac_of_8TBs": (b1 + b2) / ((ms1 + ms2) * 1e-3) / 8e12}
    return {"what": f"rank {rank} of {world}: the kernels of one sequence-parallel rank of BASELINE configs[3] executed on one GPU behind a stub "
                    f"communicator (evo_amd.sp.StubComm) -- a PREDICTION, no N > 1 run exists",
            "rank_step_ms_compute_only": rank_ms, "single_gpu_step_ms_batch1": single_ms,
            "predicted_nt_per_s_compute_only": comp, "predicted_speedup_compute_only": comp / one,
            "bytes_received_per_rank_and_step": moved, "assumed_link_GBps_one_direction": link_gbs, "assumed_small_collective_us": lat_us,
            "all_to_all_ms_if_serialised": a2a_ms, "small_collectives_ms_if_serialised": small_ms,
            "predicted_nt_per_s_collectives_serialised": ser, "predicted_speedup_collectives_serialised": ser / one,
            "hyena_shard": hy, "kernels": {k: {"launches": v[0], "avg_ms": v[1]} for k, v in ks.items()}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, default=6)
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    from bench import build_model
    from evo_amd.ops import default_ops
    dev = "cuda:0"
    ops = default_ops()
    m = build_model("evo-1-131k-base", dev)
    print(json.dumps(sp_predict(m, dev, ops, rank=a.rank, steps=a.steps)))
