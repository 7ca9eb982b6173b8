"""GPU (-m gpu): the whole engine (HIP kernels + hipBLASLt GEMMs behind the StripedHyena host class)
against the fp64 oracle on the same bf16-rounded synthetic weights and the same byte-tokenised input.

Metrics: rel-L2 of the logits vs the fp64 oracle, reported next to the bf16-faithful
oracle's own rel-L2 (the noise floor any bf16 pipeline carries); the HIP path must not be worse than
that floor by more than a small factor.  Score-level: |score - score_fp64| / |score_fp64| <= 3e-3.
"""
import numpy as np
import pytest
import torch

from oracle import stripedhyena_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cfgd, seed=0):
    from evo_amd.sh.model import StripedHyena
    cfg = R.RefConfig.from_dict(cfgd)
    sd = R.make_synthetic_state_dict(cfg, seed)
    m = StripedHyena(dict(cfgd))
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m.to_bfloat16_except_poles_residues()
    m = m.to(DEV)
    return cfg, sd, m


def rel_l2(a, ref):
    a, ref = a.double().cpu(), ref.double().cpu()
    return ((a - ref).norm() / ref.norm()).item()


def acgt(B, L, seed=1234):
    rows = []
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        rows.append(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L))
    ids = torch.from_numpy(np.stack(rows).astype(np.int64))
    return torch.cat([torch.zeros(B, 1, dtype=torch.long), ids], dim=1)      # BOS


SMALL = dict(vocab_size=512, hidden_size=256, num_layers=4, attn_layer_idxs=[2], num_attention_heads=2)
SMALL4 = dict(vocab_size=512, hidden_size=512, num_layers=4, attn_layer_idxs=[1], num_attention_heads=4)


def test_native_library_is_what_runs():
    import evo_amd.ops as eo
    ops = eo.default_ops()
    assert ops.name == "hip-gfx950" and ops.lib.evo_abi_version() == eo.ABI_VERSION
    maps = open("/proc/self/maps").read()
    assert "libevo_mi355x.so" in maps


@pytest.mark.parametrize("B,L,scaling", [(2, 300, None), (1, 1500, 16)])
def test_small_model_logits_vs_oracle(B, L, scaling):
    cfgd = dict(SMALL)
    if scaling:
        cfgd.update(use_interpolated_rotary_pos_emb=True, rotary_emb_scaling_factor=scaling)
    cfg, sd, m = build(cfgd)
    ids = acgt(B, L)
    ref = R.RefStripedHyena(cfg, sd, "fp64")(ids)[0]
    floor = rel_l2(R.RefStripedHyena(cfg, sd, "bf16")(ids)[0], ref)
    logits, cache = m(ids.to(DEV))
    assert cache is None and logits.dtype == torch.bfloat16 and logits.shape == ref.shape
    err = rel_l2(logits, ref)
    print(f"logits rel-L2 hip={err:.3e} bf16-oracle floor={floor:.3e}")
    assert err < max(1.5 * floor, 4e-3)
    assert ref.std() > 0.1


def test_full_width_two_blocks_vs_oracle():
    """D=4096, H=32, inner 10928 (the real block shapes), one Hyena + one attention block, T = 513."""
    cfgd = dict(vocab_size=512, hidden_size=4096, num_layers=2, attn_layer_idxs=[1], num_attention_heads=32)
    cfg, sd, m = build(cfgd)
    ids = acgt(1, 512)
    ref = R.RefStripedHyena(cfg, sd, "fp32")(ids)[0]
    floor = rel_l2(R.RefStripedHyena(cfg, sd, "bf16")(ids)[0], ref)
    err = rel_l2(m(ids.to(DEV))[0], ref)
    print(f"full-width logits rel-L2 hip={err:.3e} floor={floor:.3e}")
    assert err < max(1.5 * floor, 4e-3)


def test_cached_decode_consistent_with_parallel_and_oracle():
    cfg, sd, m = build(SMALL)
    ids = acgt(2, 90)
    full = m(ids.to(DEV))[0].float().cpu()
    ref = R.RefStripedHyena(cfg, sd, "fp64")(ids)[0]
    c = m.initialize_inference_params()
    c["mha"].max_batch_size = c["hyena"].max_batch_size = 2
    floor = rel_l2(R.RefStripedHyena(cfg, sd, "bf16")(ids)[0], ref)
    tol = max(1.5 * floor, 4e-3)
    l0, c = m(ids[:, :64].to(DEV), c)
    assert rel_l2(l0, ref[:, :64]) < tol
    steps = []
    for t in range(64, 91):
        c["mha"].seqlen_offset = c["hyena"].seqlen_offset = t
        lt, c = m(ids[:, t:t + 1].to(DEV), c)
        steps.append(lt[:, 0].float().cpu())
    steps = torch.stack(steps, 1)
    assert rel_l2(steps, ref[:, 64:]) < tol
    assert rel_l2(steps, full[:, 64:]) < 2 * tol
    assert c["hyena"].state_dict[0].dtype == torch.complex64 and c["hyena"].fir_state_dict[0].shape == (2, 768, 2)
    assert c["mha"].key_value_memory_dict[2].shape[2:] == (2, 2, 128)


@pytest.mark.parametrize("P", [513, 1026, 600])             # prompt lengths: tail form of z^T with r = 1, r = 2; plain (padded) form
def test_cached_prefill_through_both_forms_of_zt_then_decode(P):
    """The cached prompt pass on the channel-major path (csrc/hyena_ct.hip): for P = 512 k + r the last r tokens of every row live in
    the tail block of z^T, and the FIR state the cache keeps (the last two z rows) straddles the main area and the tail block.  The
    decode steps that follow must continue the parallel forward of the whole sequence."""
    from evo_amd.ops import HipOps
    cfg, sd, m = build(SMALL)
    B, n_new = 2, 6
    ids = acgt(B, P - 1 + n_new)                              # BOS + P - 1 + n_new tokens
    Tm, _, _, r = HipOps.zt_layout(B, P)
    assert (r > 0) == (P % 512 in (1, 2)) and m._hyena_ct_ok(torch.empty(1, device=DEV), m.blocks[0], B, P)
    full = m(ids.to(DEV))[0].float().cpu()
    ref = R.RefStripedHyena(cfg, sd, "fp64")(ids)[0]
    floor = rel_l2(R.RefStripedHyena(cfg, sd, "bf16")(ids)[0], ref)
    tol = max(1.5 * floor, 4e-3)
    c = m.initialize_inference_params()
    c["mha"].max_batch_size = c["hyena"].max_batch_size = B
    l0, c = m(ids[:, :P].to(DEV), c)
    assert rel_l2(l0, ref[:, :P]) < tol and rel_l2(l0, full[:, :P]) < 2 * tol
    steps = []
    for t in range(P, P + n_new):
        c["mha"].seqlen_offset = c["hyena"].seqlen_offset = t
        lt, c = m(ids[:, t:t + 1].to(DEV), c)
        steps.append(lt[:, 0].float().cpu())
    steps = torch.stack(steps, 1)
    print(f"[cached prefill P={P} (z^T tail tokens per row: {r})] prompt logits vs fp64 {rel_l2(l0, ref[:, :P]):.3e}, "
          f"{n_new} decode steps vs fp64 {rel_l2(steps, ref[:, P:]):.3e} (tol {tol:.3e}), vs the parallel forward {rel_l2(steps, full[:, P:]):.3e}")
    assert rel_l2(steps, ref[:, P:]) < tol
    assert rel_l2(steps, full[:, P:]) < 2 * tol


@pytest.mark.parametrize("B,P", [(2, 1025), (1, 2050), (4, 1280), (3, 700)])   # tail form r = 1 + sliver rows; r = 2; plain form of z^T without pad positions (folded too); plain form with padded rows (Hyena pre-norms unfolded)
def test_norm_folded_forward_and_cached_prefill_vs_oracle(B, P):
    """RMSNorm folded into the dense layers (csrc/gemm.hip NF; evo_amd/sh/model.py _nf_ok) on a model small enough for the fp64 oracle:
    D = 512 (the gated launch and every dense layer shape take the fold), prompts of >= 1,024 rows.  (a) the fold really runs (launch
    counts), (b) folded and unfolded forwards against the fp64 oracle: the folded one is no further away, (c) the cached prompt pass on
    the folded path, then decode steps, continue the parallel forward."""
    from evo_amd.ops import HipOps, KernelTimer
    cfg, sd, m = build(SMALL4)
    ops = m.ops
    n_new = 4
    ids = acgt(B, P - 1 + n_new)
    ref = R.RefStripedHyena(cfg, sd, "fp64")(ids)[0]
    floor = rel_l2(R.RefStripedHyena(cfg, sd, "bf16")(ids)[0], ref)
    tol = max(1.5 * floor, 4e-3)
    tail = HipOps.zt_layout(B, P)[3] > 0
    stream_rows = m.ops.zt_stream_rows_ok(B, P)
    if not m._packed:
        m._pack()
    assert m._nf_ok(B * P, None)
    was = ops.fuse_norm
    try:
        errs = {}
        for fold in (True, False):
            ops.fuse_norm = fold
            ops.timer = KernelTimer()
            got = m(ids[:, :P].to(DEV))[0]
            torch.cuda.synchronize()
            n = {k: v[0] for k, v in ops.timer.summary().items()}
            ops.timer = None
            errs[fold] = rel_l2(got, ref[:, :P])
            if fold:      # 3 Hyena blocks + 1 attention block: every post-norm and every pre-norm but block 0's is folded (the Hyena pre-norms only in
                # the tail form).  What is left in the tail-form cases beside block 0's pre-norm and the final norm: the <= 16 sliver / tail rows
                # of the three folded pre-norms, normed by the small kernel in front of the weight-streaming launch (at D = 4096 that launch
                # norms for itself: gemv_norm)
                # (round 6, ops.hyena_tail_split: with ONE tail token per row the two folded Hyena pre-norms' tail rows are normed inside the
                #  fused single-token launch: what is left of the three is the attention block's sliver rows)
                sliver = 1 <= (B * P) % 256 <= 16
                split = ops.hyena_tail_split and HipOps.zt_layout(B, P)[3] == 1 and B <= 4
                want = 2 + (0 if stream_rows else 2) + (((1 if split else 3) if stream_rows else 1) if (tail or sliver) else 0)
                assert n.get("rms_finalize", 0) == 8 and n.get("rmsnorm", 0) == want, (n, want)
            else:
                assert n.get("rms_finalize", 0) == 0 and n.get("rmsnorm", 0) == 9, n
        print(f"[norm folded B={B} P={P}] logits rel-L2 vs fp64: folded {errs[True]:.3e}, separate passes {errs[False]:.3e} (eager-bf16 oracle {floor:.3e})")
        assert errs[True] < tol and errs[True] <= 1.1 * errs[False] + 1e-4
        ops.fuse_norm = True
        c = m.initialize_inference_params()
        c["mha"].max_batch_size = c["hyena"].max_batch_size = B
        l0, c = m(ids[:, :P].to(DEV), c)
        assert rel_l2(l0, ref[:, :P]) < tol
        steps = []
        for t in range(P, P + n_new):
            c["mha"].seqlen_offset = c["hyena"].seqlen_offset = t
            lt, c = m(ids[:, t:t + 1].to(DEV), c)
            steps.append(lt[:, 0].float().cpu())
        assert rel_l2(torch.stack(steps, 1), ref[:, P:]) < tol
    finally:
        ops.fuse_norm = was
        ops.timer = None


def test_evo_api_scores_vs_oracle():
    import evo_amd
    from evo_amd.tokenizer import CharLevelTokenizer
    cfg, sd, m = build(SMALL)
    tok = CharLevelTokenizer(512)
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGT"), size=n)) for n in (700, 333, 64)]
    got = evo_amd.score_sequences(seqs, m, tok, device=DEV)
    ents = evo_amd.positional_entropies(seqs, m, tok, device=DEV)
    oracle = R.RefStripedHyena(cfg, sd, "fp64")
    for s, g, e in zip(seqs, got, ents):
        ids = torch.tensor([[0] + list(s.encode())])
        lsm = torch.log_softmax(oracle(ids)[0][0, :-1], -1)
        want = lsm.gather(-1, ids[0, 1:, None]).mean().item()
        assert abs(g - want) / abs(want) < 3e-3, (g, want)
        want_e = -(lsm.exp() * lsm).sum(-1).numpy()
        assert len(e) == len(s) and np.abs(e - want_e).max() < 5e-2


def test_scoring_fused_tail_equals_logits_path(monkeypatch):
    """score_sequences / positional_entropies through the fused unembed+log-softmax kernel == the same calls through
    model(ids) -> logits -> evo_logprob_entropy (EVO_AMD_FUSED_TAIL=0), up to one bf16 logit ulp."""
    import evo_amd
    from evo_amd.tokenizer import CharLevelTokenizer
    cfg, sd, m = build(SMALL4)
    tok = CharLevelTokenizer(512)
    rng = np.random.default_rng(9)
    seqs = ["".join(rng.choice(list("ACGT"), size=n)) for n in (257, 100, 31)]
    a = evo_amd.score_sequences(seqs, m, tok, device=DEV)
    ea = evo_amd.positional_entropies(seqs, m, tok, device=DEV)
    monkeypatch.setenv("EVO_AMD_FUSED_TAIL", "0")
    b = evo_amd.score_sequences(seqs, m, tok, device=DEV)
    eb = evo_amd.positional_entropies(seqs, m, tok, device=DEV)
    np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-3)
    for x, y in zip(ea, eb):
        assert np.abs(x - y).max() < 5e-2


def test_padding_mask_matches_oracle_on_gpu():
    """model(ids, padding_mask=...) on the HIP engine vs the fp64 oracle's masked forward (VERDICT r1 missing #6)."""
    cfg, sd, m = build(SMALL)
    ids = acgt(2, 200)
    mask = torch.ones(2, 201, dtype=torch.bool)
    mask[0, 150:] = False
    mask[1, 77:80] = False
    ref = R.RefStripedHyena(cfg, sd, "fp64")(ids, padding_mask=mask)[0]
    floor = rel_l2(R.RefStripedHyena(cfg, sd, "bf16")(ids, padding_mask=mask)[0], ref)
    got = m(ids.to(DEV), padding_mask=mask.to(DEV))[0]
    assert rel_l2(got, ref) < max(1.5 * floor, 4e-3)
    keep, m.ops.hyena_mfma = m.ops.hyena_mfma, False                   # (masked calls take the modal Hyena kernels: compare like with like)
    try:
        plain = m(ids.to(DEV))[0]
    finally:
        m.ops.hyena_mfma = keep
    assert rel_l2(got[0, :150], plain[0, :150]) < 1e-6                 # tail pads never reach back
    assert rel_l2(got[1, 80:], plain[1, 80:]) > 1e-4                   # interior pads change what follows


def test_out_of_range_ids_raise_like_the_reference():
    """ids outside [0, vocab) must not index the embedding table out of bounds; the reference's F.embedding
    device-asserts, this engine raises IndexError."""
    cfg, sd, m = build(SMALL)
    ids = acgt(1, 40)
    ids[0, 7] = 512
    with pytest.raises(IndexError):
        m(ids.to(DEV))
    ids[0, 7] = -3
    with pytest.raises(IndexError):
        m(ids.to(DEV))
    ids[0, 7] = 65
    assert torch.isfinite(m(ids.to(DEV))[0].float()).all()           # and the flag does not stick


def test_generate_greedy_on_gpu():
    import evo_amd
    from evo_amd.tokenizer import CharLevelTokenizer
    cfg, sd, m = build(SMALL)
    tok = CharLevelTokenizer(512)
    prompts = ["ACGTTGCAACGT" * 8, "GGGTTTAAACCC" * 8]
    a, sa = evo_amd.generate(prompts, m, tok, n_tokens=12, top_k=1, cached_generation=True, verbose=0, device=DEV)
    b, sb = evo_amd.generate(prompts, m, tok, n_tokens=12, top_k=1, cached_generation=True, verbose=0, device=DEV,
                             force_prompt_threshold=16)
    c, sc = evo_amd.generate(prompts, m, tok, n_tokens=12, top_k=1, cached_generation=False, verbose=0, device=DEV)
    assert len(a) == 2 and all(len(s) == 12 for s in a)
    np.testing.assert_allclose(sa, sb, rtol=2e-2, atol=2e-2)      # full prefill vs forced recurrence
    np.testing.assert_allclose(sa, sc, rtol=2e-2, atol=2e-2)      # cached vs uncached


class _ThreadComm:
    """R 'virtual ranks' as threads of one process sharing the one GPU of the box: all_gather = slot write +
    barrier.  Exercises the sequence-parallel host code on the HIP kernels without RCCL (the box has 1 GPU)."""

    def __init__(self, world):
        import threading
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()

    class _W:
        def wait(self):
            return True

    def bind(self, rank):
        self.local.rank = rank

    def all_gather(self, t, async_op=False):
        torch.cuda.synchronize()
        self.slots[self.local.rank] = t.contiguous().clone()
        self.bar.wait()
        out = torch.stack(list(self.slots), 0)
        torch.cuda.synchronize()
        self.bar.wait()
        return out, _ThreadComm._W()

    def all_to_all(self, t, async_op=False):
        g, w = self.all_gather(t)                      # [R(src), R(dst), ...] -> take what every src sent to me
        return g[:, self.local.rank].contiguous(), w


@pytest.mark.parametrize("world,L", [(2, 700), (4, 1001)])
def test_sequence_parallel_virtual_ranks_on_hip(world, L):
    import threading
    from evo_amd.sp import SequenceParallelScorer
    # world 2 on a 2-head model and world 4 on a 4-head model take the Ulysses path (heads % world == 0)
    cfgd = dict(SMALL if world == 2 else SMALL4, use_interpolated_rotary_pos_emb=True, rotary_emb_scaling_factor=16)
    cfg, sd, m = build(cfgd)
    m._pack()
    ids = acgt(2, L).to(DEV)
    T = ids.shape[1]
    full = m(ids)[0].float().cpu()
    ref = R.RefStripedHyena(cfg, sd, "fp64")(ids.cpu())[0]
    comm = _ThreadComm(world)
    outs, errs = [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            comm.bind(r)
            sp = SequenceParallelScorer(m, r, world, comm=comm)
            with torch.inference_mode():
                lg = sp.forward_local(ids)
                lp = sp.score_logprobs(ids)
            outs[r] = (sp.shard(T), lg.float().cpu(), lp.float().cpu())
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            comm.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not errs, errs
    sharded = torch.cat([o[1] for o in outs], 1)
    assert sharded.shape == full.shape
    floor = rel_l2(R.RefStripedHyena(cfg, sd, "bf16")(ids.cpu())[0], ref)
    tol = max(1.5 * floor, 4e-3)
    assert rel_l2(sharded, ref) < tol
    assert rel_l2(sharded, full) < 2 * tol
    lp = torch.cat([o[2] for o in outs], 1)
    from evo_amd.scoring import logits_to_logprobs
    want = logits_to_logprobs(ref, ids.cpu(), trim_bos=True)
    unsharded = logits_to_logprobs(full, ids.cpu(), trim_bos=True)
    assert lp.shape == want.shape
    # log-probs of a sharply peaked toy model reach -19: compare means (the reported score) and per-token noise
    assert (lp.double() - want.double()).abs().mean() < 5e-2
    assert (lp.double() - unsharded.double()).abs().mean() < 5e-2
    assert abs(lp.double().mean() - want.double().mean()) / abs(want.double().mean()) < 3e-3


def test_graph_decode_equals_eager_decode():
    """The hipGraph-captured decode step (device-side position) reproduces the eager step token for token."""
    cfg, sd, m = build(dict(SMALL, use_interpolated_rotary_pos_emb=True, rotary_emb_scaling_factor=16))
    ids = acgt(2, 140)

    def run(graph):
        m.decode_graph = graph
        m._dgraph = None
        m._dgraph_warm = None
        c = m.initialize_inference_params()
        c["mha"].max_batch_size = c["hyena"].max_batch_size = 2
        m(ids[:, :100].to(DEV), c)
        outs = []
        for t in range(100, 141):
            c["mha"].seqlen_offset = c["hyena"].seqlen_offset = t
            outs.append(m(ids[:, t:t + 1].to(DEV), c)[0][:, 0].float().cpu())
        return torch.stack(outs, 1)

    eager = run(False)
    graphed = run(True)
    assert m._dgraph is not None and m.decode_graph, "graph capture did not engage"
    assert torch.equal(graphed, eager)
    ref = R.RefStripedHyena(cfg, sd, "fp64")(ids)[0][:, 100:]
    floor = rel_l2(R.RefStripedHyena(cfg, sd, "bf16")(ids)[0][:, 100:], ref)
    assert rel_l2(graphed, ref) < max(1.5 * floor, 4e-3)
