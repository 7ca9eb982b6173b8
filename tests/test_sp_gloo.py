"""CPU, world_size 2 and 3 over gloo: the sequence-parallel engine (evo_amd/sp.py: halo exchange, end-state
all-gather + modal carry, K/V all-gather with position offsets, ragged last shard, rank-local scoring tail)
reproduces the unsharded forward.  Compute backend = the fp64 CPU oracle ops (test infrastructure); on the GPU
box the same host code runs on HipOps over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_common import TINY


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, T, B, out_q, attn_mode="auto"):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from golden_common import tiny_model
    from evo_amd.sp import SequenceParallelScorer
    from evo_amd.scoring import logits_to_logprobs
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = tiny_model()
        ids = torch.randint(0, 512, (B, T), generator=torch.Generator().manual_seed(11))
        ids[:, 0] = 0
        sp = SequenceParallelScorer(m, rank, world)
        sp.attn_mode = attn_mode
        local = sp.forward_local(ids)
        _, t0, t1 = sp.shard(T)
        full = m(ids)[0]
        err_logits = float((local - full[:, t0:t1]).abs().max())
        lp_local = sp.score_logprobs(ids)
        lp_all = sp.gather_logprobs(lp_local, T)
        want = logits_to_logprobs(full, ids, trim_bos=True)
        err_lp = float((lp_all.double() - want.double()).abs().max())
        out_q.put((rank, t0, t1, err_logits, err_lp))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T,B,attn_mode", [
    (2, 37, 2, "auto"),          # 2 heads / 2 ranks: Ulysses head<->sequence all-to-all
    (2, 37, 2, "allgather"),     # same split through the K/V all-gather fallback
    (3, 41, 1, "auto"),          # 2 heads / 3 ranks: falls back to all-gather; ragged last shard
    (2, 5, 1, "auto"),           # shards shorter than the FIR halo
])
def test_sequence_parallel_matches_unsharded(world, T, B, attn_mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, B, q, attn_mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = sorted((t0, t1) for _, t0, t1, _, _ in res)
    assert covered[0][0] == 0 and covered[-1][1] == T
    for _, _, _, e_logits, e_lp in res:
        assert e_logits < 1e-9 and e_lp < 1e-5


def test_shard_geometry_verdict_is_rank_independent():
    """(world-1)*ceil(T/world) >= T leaves an empty shard (e.g. T = 9, R = 4) -- every rank must raise
    BEFORE the first collective, not only the empty one (the others would hang in all_gather)."""
    from evo_amd.sp import SequenceParallelScorer

    class _NoComm:
        world = 4

    for T, world, ok in [(9, 4, False), (10, 4, True), (3, 4, False), (4, 4, False), (8, 4, True), (131073, 8, True)]:
        verdicts = []
        for r in range(world):
            sp = SequenceParallelScorer(None, r, world, comm=_NoComm())
            try:
                sp.check_geometry(T)
                verdicts.append(True)
            except ValueError:
                verdicts.append(False)
        assert verdicts == [ok] * world, (T, world, verdicts)


def test_pole_powers_integer_exponentiation():
    """p^(Tl*k) in fp64 by repeated squaring: exact 1 at k = 0 even for a zero pole (log(0)*0 was NaN)."""
    from evo_amd.sp import SequenceParallelScorer

    class _NoComm:
        world = 3

    sp = SequenceParallelScorer(None, 1, 3, comm=_NoComm())
    poles = torch.tensor([[[0.0, 0.0], [0.5, 0.5]], [[0.99999, 0.0], [-0.3, 0.9]]], dtype=torch.float32)
    pw = sp._pole_powers(poles, 1000)
    assert pw.shape == (3, 2, 2) and torch.isfinite(torch.view_as_real(pw)).all()
    p = torch.view_as_complex(poles.double())
    assert torch.equal(pw[0], torch.ones_like(p))
    want1 = torch.where(p == 0, torch.zeros_like(p), torch.exp(torch.log(torch.where(p == 0, torch.ones_like(p), p)) * 1000))
    assert (pw[1] - want1).abs().max() < 1e-12
    assert (pw[2] - want1 * want1).abs().max() < 1e-12
