"""GPU (-m gpu): the sequence-parallel scorer with TWO REAL RANKS on ONE GPU -- two processes, both on cuda:0, the HIP
kernels of libevo_mi355x.so, and a host-staged communicator over a gloo group (evo_amd.sp.HostStagedComm).  The box has one
GPU, so RCCL cannot connect two ranks; what this exercises and the world-1 RCCL test / the CPU gloo tests (oracle ops) /
the in-process virtual-rank test cannot: asynchronous work handles posted by one process while its ctypes-launched kernels
are in flight, the halo send/recv under the projection GEMM, the end-state all-gather pipelined over row groups, the grouped
Ulysses all-to-all, and the rank-local fused scoring tail -- against the unsharded forward of the parent process.
(VERDICT r2 next #6.)  [The reference has no multi-GPU path: SURVEY 2.4.]"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cfg_name, B, L, attn_mode, out_q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import test_gpu_model as G
    from evo_amd.sp import HostStagedComm, SequenceParallelScorer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfgd = dict(getattr(G, cfg_name), use_interpolated_rotary_pos_emb=True, rotary_emb_scaling_factor=16)
        _, _, m = G.build(cfgd)
        ids = G.acgt(B, L).to(G.DEV)
        sp = SequenceParallelScorer(m, rank, world, comm=HostStagedComm())
        sp.attn_mode = attn_mode
        with torch.inference_mode():
            lg = sp.forward_local(ids)
            lp = sp.score_logprobs(ids)
            bad = ids.clone()
            bad[0, 3 if rank == 0 else -2] = 700             # an id outside the vocabulary in ONE rank's shard ...
            try:
                sp.forward_local(bad)
                raised = False
            except IndexError:
                raised = True                                # ... must raise on EVERY rank, before any collective
        torch.cuda.synchronize()
        # (numpy: pickled by value -- a CPU tensor would travel as a shared-memory handle that dies with this process)
        out_q.put((rank, sp.shard(ids.shape[1]), lg.float().cpu().numpy(), lp.float().cpu().numpy(), raised,
                   sorted(m.ops.last_hyena_io)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg_name,B,L,attn_mode", [
    ("SMALL", 2, 700, "auto"),
    ("SMALL4", 3, 1001, "allgather"),
    ("SMALL", 5, 62, "auto"),        # T = 63: shards of 32 and 31 tokens -- the second is below the single-pass kernel's floor, so
                                      # BOTH ranks must take the modal kernels (the halo travels in the projection's column order)
])
def test_two_ranks_one_gpu_host_staged(cfg_name, B, L, attn_mode):
    import test_gpu_model as G
    from evo_amd.scoring import logits_to_logprobs
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg_name, B, L, attn_mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfgd = dict(getattr(G, cfg_name), use_interpolated_rotary_pos_emb=True, rotary_emb_scaling_factor=16)
    _, _, m = G.build(cfgd)
    ids = G.acgt(B, L)
    # like with like: the sequence-parallel ranks (evo_amd/sp.py) run every RMSNorm as its own pass; the unsharded forward would fold
    # the norms of a >= 1,024-row batch into its dense layers (another, equally valid set of roundings: tests/PARITY.md rows 11a-c)
    # (round 6: the shards fold their RMSNorm passes like the unsharded forward -- both sides run the default routing; until round 5 the
    #  shards ran 65 separate passes and this yardstick was taken with fuse_norm = False)
    with torch.inference_mode():
        full = m(ids.to(G.DEV))[0].float().cpu()
    sharded = torch.cat([torch.from_numpy(r[2]) for r in res], 1)
    assert sharded.shape == full.shape
    # same kernels on the same data: the shards add a carried state (fp64 pole powers), another tiling of the sums and (attention)
    # another split of the keys -- bf16 rounding noise through 4 blocks; the bound is the one of the in-process virtual-rank
    # test (tests/test_gpu_model.py: 2 x max(1.5 x bf16 floor, 4e-3))
    assert G.rel_l2(sharded, full) < 8e-3
    lp = torch.cat([torch.from_numpy(r[3]) for r in res], 1)
    want = logits_to_logprobs(full, ids, trim_bos=True)
    assert lp.shape == want.shape
    assert (lp.double() - want.double()).abs().mean() < 5e-2
    assert all(r[4] for r in res)                            # the bad id raised on both ranks
    kinds = {tuple(r[5]) for r in res}                       # same Hyena operator form on every rank
    assert len(kinds) == 1
    if L >= 200:
        assert kinds == {("mfma",)}                          # the single-pass kernel (state-only walk + seeded pass)
