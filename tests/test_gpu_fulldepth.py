"""GPU (-m gpu): model-level parity at the REAL depth and width (32 layers, D = 4096, H = 32, inner 10928) on
BASELINE configs, and operator-level cross-checks at the BASELINE sizes the CPU oracle cannot reach.

What is compared with what (reference path: /root/reference/evo/scoring.py:80-84 -> model(input_ids)):

 (a) BASELINE configs[0] (1 x 512 nt, T = 513), evo-1-8k AND evo-1-131k (rotary / 16) yml: the HIP engine vs
     the CPU oracle in fp32 mode on the SAME synthetic 7B weights (bf16-rounded) and the same ids.
     Metrics, pinned to ABSOLUTE numbers (tests/PARITY.md records the measured values they come from):
       * score_rel   = |score_hip - score_oracle| / |score_oracle|   (north-star "1e-3 relative" -- met here)
       * logits rel-L2 and max |delta| in units of a bf16 half-ulp at the row's scale
     The bf16-faithful oracle (a rounding after every eager op = what upstream computes) is run beside it: the
     engine must not be further from fp32 than that restatement is.
 (b) prefix check at bench length: rows of the 8 x 8,193 HIP run (BASELINE configs[1]) vs the fp32 oracle on the
     first 2,049 tokens of the same row -- causality makes them comparable.
 (c) full-size operator checks against GPU restatements in fp64 (TEST INFRASTRUCTURE, rocFFT / eager matmul):
     the SHIPPED Hyena operator kernels (hyena_ct on z^T in its tail and padded forms and on a sequence-parallel shard with
     halo + carried state in row groups, and the modal three-launch path) vs an FFT long convolution at 8 x 8,193 x 4096,
     1 x 131,073 x 4096 and 8 x 16,385 x 4096, and causal attention vs eager softmax attention at H = 32, T = 8,193 / 131,073.
"""
import math
import os
import time

import numpy as np
import pytest
import torch

from oracle import stripedhyena_ref as R
from gpu_ref64 import attn_block64, causal_attention64, gpu_fft_hyena, hyena_block64

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

FULL = dict(vocab_size=512, hidden_size=4096, num_layers=32, attn_layer_idxs=[8, 16, 24], num_attention_heads=32)
FULL_131K = dict(FULL, use_interpolated_rotary_pos_emb=True, rotary_emb_scaling_factor=16)


def rel_l2(a, ref):
    a, ref = a.double().cpu(), ref.double().cpu()
    return ((a - ref).norm() / ref.norm()).item()


def acgt_ids(B, L, seed=1234):
    """SURVEY 8(d) inputs: default_rng(1234 + b).choice("ACGT"), BOS prepended."""
    rows = []
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        rows.append(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L))
    ids = torch.from_numpy(np.stack(rows).astype(np.int64))
    return torch.cat([torch.zeros(B, 1, dtype=torch.long), ids], dim=1)


def score_of(logits, ids):
    lsm = torch.log_softmax(logits.double()[:, :-1], -1)
    return lsm.gather(-1, ids[:, 1:, None].long()).squeeze(-1).mean(-1)


def half_ulps(got, ref):
    """max |got - ref| in units of half a bf16 ulp at the scale of the row (max(|ref|, rms of the row))."""
    got, ref = got.double().cpu(), ref.double().cpu()
    scale = torch.maximum(ref.abs(), ref.pow(2).mean(-1, keepdim=True).sqrt())
    return ((got - ref).abs() / (scale * 2.0 ** -9)).max().item()


def _host_mem_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            return int(line.split()[1]) / 1e6
    return 0.0


def oracle_for(full, cfgd, mode):
    """One oracle object per numeric mode (fp32: 26 GB of up-cast weights, bf16: aliases the host copy); the two yml
    configs differ only in the rotary table, which the oracle derives from `cfg` on every call."""
    if mode not in full["oracles"]:
        need = 34.0 if mode == "fp32" else 4.0
        if _host_mem_gb() < need:
            pytest.skip(f"host has {_host_mem_gb():.0f} GB available; the {mode} full-depth oracle needs {need:.0f} GB")
        full["oracles"][mode] = R.RefStripedHyena(R.RefConfig.from_dict(cfgd), full["sd_cpu"], mode)
    o = full["oracles"][mode]
    o.cfg = R.RefConfig.from_dict(cfgd)
    return o


def gpu_oracle(full, cfgd, mode):
    """The SAME oracle class with its weights on the GPU: its statements then run on torch's eager GPU kernels (rocBLAS /
    rocFFT; test infrastructure, nothing of libevo_mi355x.so) -- pinned to the CPU execution by
    tests/test_gpu_parity_r4.py::test_oracle_on_the_gpu_is_the_cpu_oracle.  fp32: 26 GB of up-cast weights; bf16 aliases the
    engine's tensors."""
    key = "gpu_" + mode
    if key not in full["oracles"]:
        sd = {k: v for k, v in full["m131"].state_dict().items()}
        full["oracles"][key] = R.RefStripedHyena(R.RefConfig.from_dict(cfgd), sd, mode, device=DEV)
    o = full["oracles"][key]
    o.cfg = R.RefConfig.from_dict(cfgd)
    return o


# ---- (a) BASELINE configs[0] at full depth ------------------------------------------------------------------------
# Measured on MI355X in round 2 (tests/PARITY.md).  A 32-block bf16 stack of RANDOM weights amplifies rounding noise
# chaotically: the eager-bf16 restatement of the reference itself sits at logits rel-L2 1.8e-1 / score_rel 9e-4 from
# the fp32 oracle, so the end-to-end numbers below are floors of the number format, not of the kernels.  The sharp
# check is (a1): every one of the 32 full-width blocks, fed the engine's OWN input, against the fp32 oracle block.
# (a1) pins: measured worst block 6.1e-3 (block 0, where the whole stream IS the block's update) / update 1.7e-2
# (attention block 16: the stream's own bf16 rounding, |stream| ~ 4x |update|, is half of that).
PIN_BLOCK = dict(hyena=9.0e-3, attn=9.0e-3, hulp=28.0, upd=3.0e-2)   # per-block output rel-L2, half-ulps, update rel-L2
PIN_E2E = dict(score=3.0e-3, rl2=1.6e-1, hulp=400.0)       # end to end; also asserted: not worse than eager bf16


@pytest.mark.parametrize("name", ["8k", "131k"])
def test_full_depth_every_block_teacher_forced_vs_fp32_oracle(full, name):
    """(a1) The engine's residual stream entering block i (tapped on the device) is fed to the fp32 oracle's block i;
    the engine's block output must match it to bf16 rounding.  All 32 blocks at D = 4096 on BASELINE configs[0]."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfgd = FULL if name == "8k" else FULL_131K
    m = full["m8"] if name == "8k" else full["m131"]
    ids = acgt_ids(1, 512)
    m.block_taps = []
    try:
        m(ids.to(DEV))
        taps = [t.float().cpu().view(1, 513, 4096) for t in m.block_taps]
    finally:
        m.block_taps = None
    assert len(taps) == 33
    o = oracle_for(full, cfgd, "fp32")
    ob = oracle_for(full, cfgd, "bf16")                               # eager-bf16 restatement of the same block = floor
    todo = range(32) if name == "8k" else o.cfg.attn_layer_idxs      # the 131k yml differs in the rotary table only
    worst = {"hyena": (0.0, 0.0, 0.0, 0.0, -1), "attn": (0.0, 0.0, 0.0, 0.0, -1)}
    bad = []
    for i in todo:
        kind = "attn" if i in o.cfg.attn_layer_idxs else "hyena"
        ref = (o.attn_block if kind == "attn" else o.hyena_block)(taps[i], i, None)
        flo = (ob.attn_block if kind == "attn" else ob.hyena_block)(taps[i].bfloat16(), i, None).float()
        # compare the block's UPDATE of the residual stream too: the stream itself is dominated by its input
        err, hu = rel_l2(taps[i + 1], ref), half_ulps(taps[i + 1], ref)
        upd, upd_floor = rel_l2(taps[i + 1] - taps[i], ref - taps[i]), rel_l2(flo - taps[i], ref - taps[i])
        if upd > worst[kind][2]:
            worst[kind] = (err, hu, upd, upd_floor, i)
        if not (err <= PIN_BLOCK[kind] and hu <= PIN_BLOCK["hulp"] and upd <= PIN_BLOCK["upd"]
                and upd <= 1.3 * upd_floor + 1e-3):
            bad.append((i, kind, err, hu, upd, upd_floor))
    for kind in ("hyena", "attn"):
        e, h_, u, f, i = worst[kind]
        print(f"[per-block {name}] worst {kind} block = {i}: output rel-L2 {e:.3e}, half-ulps {h_:.1f}, update rel-L2 {u:.3e} "
              f"(eager-bf16 oracle block: {f:.3e})")
    assert not bad, bad


@pytest.mark.parametrize("name", ["8k", "131k"])
def test_full_depth_7b_configs0_vs_fp32_oracle(full, name):
    """(a2) end to end: logits and score of the 32-layer engine vs the fp32 oracle, beside the eager-bf16 restatement."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfgd = FULL if name == "8k" else FULL_131K
    m = full["m8"] if name == "8k" else full["m131"]
    ids = acgt_ids(1, 512)
    t0 = time.time()
    ref = oracle_for(full, cfgd, "fp32")(ids)[0]
    t_cpu = time.time() - t0
    floor_logits = oracle_for(full, cfgd, "bf16")(ids)[0]
    logits = m(ids.to(DEV))[0]
    assert logits.shape == ref.shape == (1, 513, 512) and logits.dtype == torch.bfloat16
    assert ref.std() > 0.1                                    # non-degenerate logits
    err, floor = rel_l2(logits, ref), rel_l2(floor_logits, ref)
    hu, hu_floor = half_ulps(logits, ref), half_ulps(floor_logits, ref)
    s_hip, s_ref, s_floor = (score_of(x.cpu(), ids).item() for x in (logits, ref, floor_logits))
    srel, srel_floor = abs(s_hip - s_ref) / abs(s_ref), abs(s_floor - s_ref) / abs(s_ref)
    print(f"[full-depth {name}] oracle fp32 pass {t_cpu:.1f} s ({512 / t_cpu:.0f} nt/s on {torch.get_num_threads()} threads); "
          f"logits std {ref.std().item():.2f}")
    print(f"[full-depth {name}] logits rel-L2 hip={err:.3e} (bf16-faithful oracle {floor:.3e}); "
          f"half-ulps hip={hu:.1f} (oracle-bf16 {hu_floor:.1f}); score hip={s_hip:.6f} fp32={s_ref:.6f} "
          f"rel={srel:.2e} (oracle-bf16 {srel_floor:.2e})")
    assert srel <= PIN_E2E["score"]
    assert err <= PIN_E2E["rl2"] and hu <= PIN_E2E["hulp"]
    assert err <= floor                                       # never further from fp32 than eager bf16 is


@pytest.mark.parametrize("gemm", ["default", "library_l3", "unfused", "attention_round4", "modal_hyena", "norm_unfused", "tail_in_operator"])
def test_prefix_of_bench_batch_vs_fp32_oracle(full, gemm):
    """(b) BASELINE configs[1]: the 8 x 8,193 scoring batch on the HIP engine; row 3's first 2,049 positions vs the fp32
    oracle run on that prefix alone (the model is causal) -- end to end, and block by block with the engine's own
    block inputs (teacher-forced), which is the check that is not blurred by 32 layers of bf16 noise.
    `gemm`: the default routing (every dense layer on the hand-written kernel of csrc/gemm.hip -- the Hyena projections with a
    transposed result for hyena_ct, the Hyena output projections gathering the operator's blocked y, the gated MLP's first half with
    GELU * gate in the epilogue; attention on attn_fwd_w64_kernel -- launch counts asserted: zero library GEMMs), and the in-process
    A/B routings bench.py times beside the headline: `library_l3` (ops.all_gemm_mfma = False: l3 / unembedding on hipBLASLt),
    `unfused` (dense layer + gate kernel), `attention_round4` (ops.attn_w64 = False: the 8-wave attention kernel of rounds 2-4) and
    `modal_hyena` (ops.hyena_mfma = False: the three-launch modal Hyena kernels on token-major z), `norm_unfused` (ops.fuse_norm = False:
    the 65 separate RMSNorm passes of rounds 1-4 instead of the norm folded into the dense layers' epilogues), `tail_in_operator`
    (ops.hyena_tail_split = False: the last token of every row as a ragged tile of hyena_ct instead of one fused single-token launch)."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    P, row = 2049, 3
    ids = acgt_ids(8, 8192)
    m = full["m8"]
    ops = m.ops
    was = ops.all_gemm_mfma, ops.mlp_gate_fused, ops.attn_w64, ops.hyena_mfma, ops.fuse_norm
    was_split, ops.hyena_tail_split = ops.hyena_tail_split, gemm != "tail_in_operator"
    ops.all_gemm_mfma = gemm != "library_l3"
    ops.mlp_gate_fused = gemm != "unfused"
    ops.attn_w64 = gemm != "attention_round4"
    ops.hyena_mfma = gemm != "modal_hyena"
    ops.fuse_norm = gemm != "norm_unfused"
    if ops.timer is None:
        from evo_amd.ops import KernelTimer
        ops.timer = KernelTimer()
    ops.timer.pairs.clear()
    m.block_taps = []
    try:
        logits = m(ids.to(DEV))[0]
        taps = [t.view(8, 8193, 4096)[row:row + 1, :P].float().cpu() for t in m.block_taps]
        torch.cuda.synchronize()
        launches = {k_: n for k_, (n, _) in ops.timer.summary().items()}
    finally:
        m.block_taps = None
        ops.all_gemm_mfma, ops.mlp_gate_fused, ops.attn_w64, ops.hyena_mfma, ops.fuse_norm = was
        ops.hyena_tail_split = was_split
        ops.timer = None
    # the routing under test really ran (here `model(ids)` materialises logits through ops.linear: one more dense layer than a
    # scoring step, whose unembedding is fused into the tail kernel)
    print(f"[prefix {gemm}] launches: {launches}")
    if gemm == "modal_hyena":
        assert launches.get("gemm_zt", 0) == 0 and launches.get("hyena_mfma", 0) == 0 and launches.get("hyena_apply", 0) == 29
    else:
        assert launches.get("gemm_zt", 0) == 29 and launches.get("hyena_mfma", 0) == 29 and launches.get("hyena_apply", 0) == 0
    assert launches.get("attn_fwd", 0) == 3
    # the token behind the 8,192 main tokens of every row: one fused single-token launch per Hyena layer (default), or inside the operator
    assert launches.get("gemv_hyena", 0) == (0 if gemm in ("tail_in_operator", "modal_hyena") else 29), launches
    # the RMSNorm passes: folded into the dense layers on the default routing (what is left: block 0's pre-norm and the final norm);
    # the folding needs the hand-written dense layer everywhere and the gated launch, the modal Hyena path norms for itself
    n_norm = launches.get("rmsnorm", 0)
    if gemm in ("default", "attention_round4", "tail_in_operator"):
        assert n_norm == 2 and launches.get("rms_finalize", 0) == 64, launches
    elif gemm == "modal_hyena":                            # (the modal Hyena path norms for itself; only the attention blocks' MLPs fold)
        assert n_norm == 62 and launches.get("rms_finalize", 0) == 6, launches
    else:
        assert n_norm == 65 and launches.get("rms_finalize", 0) == 0, launches
    if gemm == "library_l3":
        assert launches.get("gemm", 0) >= 32 and launches.get("gemm_gate", 0) == 32
    elif gemm == "unfused":
        assert launches.get("gemm", 0) == 0 and launches.get("gemm_gate", 0) == 0 and launches.get("gelu_gate", 0) >= 32
    else:                                                  # 29 + 32 + 6 + 1 plain launches (modal_hyena: + 29 token-major projections), 32 gated, no library
        assert launches.get("gemm", 0) == 0 and launches.get("gemm_mfma", 0) >= 68 and launches.get("gemm_gate", 0) == 32
    assert logits.shape == (8, 8193, 512) and len(taps) == 33
    o = oracle_for(full, FULL, "fp32")
    t0 = time.time()
    worst = 0.0
    for i in range(32):
        kind = "attn" if i in o.cfg.attn_layer_idxs else "hyena"
        ref = (o.attn_block if kind == "attn" else o.hyena_block)(taps[i], i, None)
        err, hu = rel_l2(taps[i + 1], ref), half_ulps(taps[i + 1], ref)
        worst = max(worst, err)
        upd = rel_l2(taps[i + 1] - taps[i], ref - taps[i])
        assert err <= PIN_BLOCK[kind] and hu <= PIN_BLOCK["hulp"] and upd <= PIN_BLOCK["upd"], (i, kind, err, hu, upd)
    print(f"[prefix {gemm}] 32 teacher-forced blocks on {P} tokens of row {row}: worst rel-L2 {worst:.3e} "
          f"({time.time() - t0:.1f} s of oracle)")
    if "prefix_ref" not in full:
        t0 = time.time()
        full["prefix_ref"] = o(ids[row:row + 1, :P])[0]
        print(f"[prefix] oracle fp32 end to end on {P} tokens: {time.time() - t0:.1f} s")
    ref = full["prefix_ref"]
    got = logits[row:row + 1, :P]
    err, hu = rel_l2(got, ref), half_ulps(got, ref)
    s_hip, s_ref = score_of(got.cpu(), ids[row:row + 1, :P]).item(), score_of(ref, ids[row:row + 1, :P]).item()
    srel = abs(s_hip - s_ref) / abs(s_ref)
    print(f"[prefix {gemm}] end to end: logits rel-L2 {err:.3e}, half-ulps {hu:.1f}, score hip={s_hip:.6f} fp32={s_ref:.6f} rel {srel:.2e}")
    assert srel <= PIN_E2E["score"] and err <= 2.0e-1 and hu <= 600.0


# ---- (c) full-size operator cross-checks -------------------------------------------------------------------------
def _hyena_inputs(B, T, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    D, H = 4096, 32
    z = torch.randn(B, T, 3 * D, generator=g, device=DEV).bfloat16()
    fir_w = (torch.randn(3 * D, 3, generator=g, device=DEV) * 0.3).bfloat16()
    fir_b = (torch.randn(3 * D, generator=g, device=DEV) * 0.1).bfloat16()
    one_minus = 10.0 ** (-5.0 + 4.0 * torch.rand(D, 8, generator=g, device=DEV))
    mag = 1.0 - one_minus
    ang = (torch.rand(D, 8, generator=g, device=DEV) * 2 - 1) * math.pi
    poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], -1).float().contiguous()
    res = (torch.randn(D, 8, 2, generator=g, device=DEV) * torch.sqrt(one_minus).unsqueeze(-1)).float().contiguous()
    dskip = (torch.randn(D, generator=g, device=DEV) * 0.5).bfloat16()
    return z, (fir_w, fir_b, poles, res, dskip, H)


def _check_hyena_fullsize(B, T, seed, state_tol, form):
    """The operator kernels the PRODUCT launches at this shape, each against the SAME fp64 restatement (rocFFT long convolution,
    tests/gpu_ref64.py): `hyena_ct_kernel` on channel-major z^T (csrc/hyena_ct.hip via the C-ABI entry evo_hyena_ct: the default of every
    scoring forward and of cached prefill, the kernel bench.py's `roofline` reports) in the form of z^T named by `form` ("tail": T = 512 k + r,
    the bench shapes; "padded": every other T), and the three-launch modal path (evo_hyena_seg_state / carry_scan / apply: padding masks,
    inputs shorter than 32 tokens).  Pad and tail-block positions of z^T are filled with NaN: they must not reach any output.
    Beside the fp64 truth the restatement is evaluated once more with a bf16 rounding wherever the reference's eager bf16 pipeline rounds
    (`ref_rounding`): the engine must be no further from fp64 than that.  [REF evo-1-131k-base_inference.yml:33,37: use_flashfft False,
    prefill_style fft -- what the reference computes here is the FFT form]"""
    from evo_amd.hyena_tables import mfma_operand_table
    from evo_amd.ops import HipOps
    ops = HipOps()
    z, prm = _hyena_inputs(B, T, seed)
    fir_w, fir_b, poles, res, dskip, H = prm
    D = 4096
    t0 = time.time()
    ry, rst = gpu_fft_hyena(z, *prm)
    rfloor, sfloor = gpu_fft_hyena(z, *prm, ref_rounding=True)
    torch.cuda.synchronize()
    floor_rl2 = ((rfloor - ry).norm() / ry.norm()).item()
    floor_srel = ((sfloor - rst).abs().max() / rst.abs().max()).item()
    del rfloor, sfloor
    print(f"[fft cross-check {B}x{T}] fp64 rocFFT restatements: {time.time() - t0:.1f} s; the reference's eager-bf16 "
          f"arithmetic sits at y rel-L2 {floor_rl2:.3e}, end-state rel {floor_srel:.2e} from fp64")
    bound = ry.abs() * 2 ** -8 + float(ry.abs().max()) * 2e-3
    table = mfma_operand_table(poles, res, dskip)
    Tm, Tp, Mp, r = ops.zt_layout(B, T)
    assert (r > 0) == (form == "tail"), (form, Tm, Tp, Mp, r)               # the shape takes the form of z^T it is meant to test
    assert ops.zt_shape_ok(B, T, 3 * D, D)                                    # ... and the model routes it to hyena_ct (sh/model.py:_hyena_ct_ok)
    # round 6: T = 512 k + 1 -- what the scoring path launches since then (ops.hyena_tail_split): the operator on the 512 k main tokens of
    # every row (whole tiles, y rows at a pitch of T), the token behind them as ONE step of the decode kernel from the operator's end state
    for path in ("modal", "hyena_ct") + (("hyena_ct main + step",) if r == 1 else ()):
        if path == "modal":
            y, st = ops.hyena_prefill(z, *prm, want_state=True)
            assert "apply" in ops.last_hyena_io
        elif path == "hyena_ct main + step":
            zt = ops.zt_from_rows(z, B, T, pad_value=float("nan"))
            yb = ops.yblk_empty(B * T, D, z.device)
            yb.fill_(float("nan"))
            yb, st = ops.hyena_ct(zt, B, T, fir_w, fir_b, table, H, want_state=True, poles=poles, y_blk=yb, main_only=True)
            y = ops.yblk_to_rows(yb, B * T).view(B, T, D).clone()
            assert torch.isnan(y[:, Tm:]).all() and torch.isfinite(y[:, :Tm]).all()      # the rows behind the main tokens are the caller's
            fir = z[:, Tm - 2:Tm].transpose(1, 2).contiguous()                            # [B, 3 D, 2], oldest first
            y[:, Tm] = ops.hyena_step(z[:, Tm].contiguous(), fir, st, fir_w, fir_b, poles, res, dskip, H)   # (st updated in place: the state after T - 1)
            del zt, yb
        else:
            zt = ops.zt_from_rows(z, B, T, pad_value=float("nan"))
            yb, st = ops.hyena_ct(zt, B, T, fir_w, fir_b, table, H, want_state=True, poles=poles, y_blk=ops.yblk_empty(B * T, D, z.device))
            y = ops.yblk_to_rows(yb, B * T).view(B, T, D)
            del zt, yb
        yd = y.double()
        err = (yd - ry).abs()
        rl2 = ((yd - ry).norm() / ry.norm()).item()
        excess = (err - bound).max().item()
        srel = ((st.to(torch.complex128) - rst).abs().max() / rst.abs().max()).item()
        print(f"[fft cross-check {B}x{T} {form}] {path}: y rel-L2 {rl2:.3e}, worst excess over the bf16 bound {excess:.3e}, "
              f"end-state rel {srel:.2e}")
        assert torch.isfinite(yd).all(), path
        assert (err <= bound).all(), path
        assert rl2 < 2e-3 and rl2 <= floor_rl2, (path, rl2, floor_rl2)     # one bf16 output rounding alone = 1.1e-3
        assert srel <= state_tol and srel <= floor_srel, (path, srel, floor_srel)
        del yd, err, y


def test_hyena_operator_8x8193_full_width_vs_fft():
    """BASELINE configs[1] shape of one Hyena layer: every one of the 8 x 8,193 x 4096 outputs of the shipped operator
    (hyena_ct_kernel, tail form of z^T) and of the modal path vs the FFT form."""
    _check_hyena_fullsize(8, 8193, 21, 2e-5, "tail")


def test_hyena_operator_1x131073_full_width_vs_fft():
    """BASELINE configs[2] shape: T = 131,073 at all 4096 channels, |p| up to 0.99999 (hyena_ct_kernel, tail form)."""
    _check_hyena_fullsize(1, 131073, 22, 1e-4, "tail")


def test_hyena_operator_3x5003_padded_form_full_width_vs_fft():
    """A shape outside the tail form (T = 5,003 = 512 x 9 + 395): batch rows of z^T padded to 64 positions, NaN in the pads."""
    _check_hyena_fullsize(3, 5003, 24, 2e-5, "padded")


def test_hyena_ct_shard_8x16385_full_width_with_halo_and_carry_vs_fft():
    """BASELINE configs[3]: what ONE sequence-parallel rank runs per Hyena layer (evo_amd/sp.py) -- `hyena_ct_kernel` on the z^T of a shard
    of 8 x 16,385 tokens at D = 4096 (tail form: 16,385 = 512 x 32 + 1) with the two halo rows of the left neighbour and a carried-in
    modal state, launched per ROW GROUP (b_first / b_total: two groups of four rows into one blocked y) as the scorer does; stage 1 =
    the state-only walk from a zero state is checked beside it -- every output and the end state vs the fp64 FFT restatement continued
    from the same halo / state."""
    from evo_amd.hyena_tables import mfma_operand_table
    from evo_amd.ops import HipOps
    ops = HipOps()
    B, T, D = 8, 16385, 4096
    z, prm = _hyena_inputs(B, T + 2, 25)
    fir_w, fir_b, poles, res, dskip, H = prm
    halo, z = z[:, :2].contiguous(), z[:, 2:].contiguous()
    g = torch.Generator(device=DEV).manual_seed(26)
    s0 = torch.view_as_complex((torch.randn(B, D, 8, 2, generator=g, device=DEV) * 3.0).contiguous())
    ry, rst = gpu_fft_hyena(z, *prm, z_halo=halo, s0=s0)
    rst0 = gpu_fft_hyena(z, *prm, z_halo=halo)[1]
    rfloor, sfloor = gpu_fft_hyena(z, *prm, z_halo=halo, s0=s0, ref_rounding=True)
    floor_rl2 = ((rfloor - ry).norm() / ry.norm()).item()
    floor_srel = ((sfloor - rst).abs().max() / rst.abs().max()).item()
    del rfloor, sfloor
    bound = ry.abs() * 2 ** -8 + float(ry.abs().max()) * 2e-3
    table = mfma_operand_table(poles, res, dskip)
    assert ops.zt_layout(B, T)[3] == 1                                          # tail form
    zt = ops.zt_from_rows(z, B, T, pad_value=float("nan"))
    del z
    yb = ops.yblk_empty(B * T, D, zt.device)
    s1, st = [], []
    for b0 in (0, 4):
        s1.append(ops.hyena_ct(zt, 4, T, fir_w, fir_b, table, H, z_halo=halo[b0:b0 + 4], poles=poles, state_only=True, b_first=b0, b_total=B))
        st.append(ops.hyena_ct(zt, 4, T, fir_w, fir_b, table, H, z_halo=halo[b0:b0 + 4], s0=s0[b0:b0 + 4], want_state=True, poles=poles,
                               b_first=b0, b_total=B, y_blk=yb, y_row0=b0 * T)[1])
    s1, st = torch.cat(s1), torch.cat(st)
    yd = ops.yblk_to_rows(yb, B * T).view(B, T, D).double()
    err = (yd - ry).abs()
    rl2 = ((yd - ry).norm() / ry.norm()).item()
    srel = ((st.to(torch.complex128) - rst).abs().max() / rst.abs().max()).item()
    srel1 = ((s1.to(torch.complex128) - rst0).abs().max() / rst0.abs().max()).item()
    print(f"[fft cross-check shard 8x16385, halo + carried state, two row groups] hyena_ct: y rel-L2 {rl2:.3e} (eager-bf16 arithmetic {floor_rl2:.3e}), "
          f"worst excess over the bf16 bound {(err - bound).max().item():.3e}, end-state rel {srel:.2e} (floor {floor_srel:.2e}), state-only walk {srel1:.2e}")
    assert torch.isfinite(yd).all() and (err <= bound).all()
    assert rl2 < 2e-3 and rl2 <= floor_rl2
    assert srel <= 2e-5 and srel <= floor_srel and srel1 <= 2e-5


@pytest.mark.parametrize("pre", [False, True])
def test_attention_h32_t8193_vs_eager_fp64(pre):
    """BASELINE configs[1] attention shape (one batch row, all 32 heads, T = 8,193) vs eager softmax attention in
    fp64 on the GPU (TEST INFRASTRUCTURE; FlashAttention-2 numerics tolerance: P is rounded to bf16)."""
    from evo_amd.ops import HipOps
    ops = HipOps()
    g = torch.Generator(device=DEV).manual_seed(23)
    T, H = 8193, 32
    qkv = torch.randn(1, T, 3, H, 128, generator=g, device=DEV).bfloat16()
    c = ops.attn_q_scale(128)
    if pre:            # round 6, the model's default: queries pre-scaled by softmax_scale * log2(e) (one rounding), scores taken as exponents
        qp = (qkv[:, :, 0].float() * c).bfloat16()
        o = ops.attention(qp, qkv[:, :, 1], qkv[:, :, 2], 0, prescaled=True)
    else:
        o = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0)
    mask = torch.ones(T, T, dtype=torch.bool, device=DEV).triu(1)
    worst_rl2, worst_abs = 0.0, 0.0
    for h in range(H):
        q, k, v = (qkv[0, :, i, h].double() for i in range(3))
        if pre:
            q = qp[0, :, h].double() / c
        sc = (q @ k.t()) / math.sqrt(128.0)
        sc.masked_fill_(mask, float("-inf"))
        ref = torch.softmax(sc, -1) @ v
        got = o[0, :, h].double()
        worst_rl2 = max(worst_rl2, ((got - ref).norm() / ref.norm()).item())
        worst_abs = max(worst_abs, ((got - ref).abs() - ref.abs() * 2 ** -7).max().item())
        del sc, ref
    print(f"[attention H=32 T=8193{' pre-scaled q' if pre else ''}] worst head rel-L2 {worst_rl2:.3e}, worst |err| - 2^-7|ref| = {worst_abs:.3e}")
    assert worst_rl2 < 4e-3
    assert worst_abs < 2e-2


@pytest.mark.parametrize("pre", [False, True])
def test_attention_h32_t131073_vs_eager_fp64(pre):
    """BASELINE configs[2] attention shape: one row, all 32 heads, T = 131,073 -- the pipelined kernel walks up to 2,049
    key tiles per query block.  768 query rows (the first block, a block in the middle, the last 256 rows) of the full
    launch, and a launch that STARTS at q_pos0 = 98,305 (a sequence-parallel shard / cache continuation), vs chunked eager
    softmax attention in fp64 (TEST INFRASTRUCTURE).  [REF evo/scoring.py:81 with evo-1-131k-base_inference.yml:39-40]"""
    from evo_amd.ops import HipOps
    ops = HipOps()
    g = torch.Generator(device=DEV).manual_seed(29)
    T, H = 131073, 32
    qkv = torch.randn(1, T, 3, H, 128, generator=g, device=DEV).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    kw = {}
    if pre:
        c = ops.attn_q_scale(128)
        q = (q.float() * c).bfloat16()
        kw = {"prescaled": True}
    o = ops.attention(q, k, v, 0, **kw)
    off = 98305
    o_tail = ops.attention(q[:, off:], k, v, off, **kw)
    assert torch.equal(o_tail, o[:, off:])                               # same tiles in the same order: bit identical
    rows = torch.cat([torch.arange(0, 256), torch.arange(65408, 65664), torch.arange(T - 256, T)]).to(DEV)
    t0 = time.time()
    ref = causal_attention64(q[0, rows].double() / c if pre else q[0, rows], k[0], v[0], rows)               # [768, 32, 128] fp64
    torch.cuda.synchronize()
    got = o[0, rows].double()
    worst_rl2 = max(((got[:, h] - ref[:, h]).norm() / ref[:, h].norm()).item() for h in range(H))
    worst_abs = ((got - ref).abs() - ref.abs() * 2 ** -7).max().item()
    # late rows average ~1e5 values: |o| ~ 1/sqrt(n) -- judge the error against the row's own scale too
    row_rl2 = ((got - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)).max().item()
    print(f"[attention H=32 T=131073{' pre-scaled q' if pre else ''}] 768 rows vs fp64 ({time.time() - t0:.1f} s): worst head rel-L2 {worst_rl2:.3e}, "
          f"worst row rel-L2 {row_rl2:.3e}, worst |err| - 2^-7|ref| = {worst_abs:.3e}")
    assert worst_rl2 < 4e-3 and row_rl2 < 8e-3
    assert worst_abs < 2e-2


# ---- (d) model level on BASELINE configs[2]: 1 x 131,073 tokens through the 32-layer engine -------------------------------
def test_131k_forward_blocks_teacher_forced_vs_fp64(full):
    """[REF evo/scoring.py:81 on evo-1-131k-base_inference.yml] The 1 x 131,073 forward of the 131k yml (rotary / 16) runs
    the DEFAULT kernels (single-pass Hyena, pipelined attention, fused scoring tail untouched here); the residual stream
    entering blocks 0, 8 (attention), 16 (attention) and 31 is tapped and fed to the fp64 GPU restatement of the oracle's
    block (tests/gpu_ref64.py -- projection, long convolution / attention over the WHOLE sequence; the row-local tail on
    2,048 rows: the first 512, 512 around the middle, the last 1,024).  Same pins as at configs[0] (PIN_BLOCK)."""
    m = full["m131"]
    T = 131073
    ids = acgt_ids(1, T - 1)
    check = (0, 8, 16, 31)
    m.block_taps = []
    m.block_tap_idxs = {i for c in check for i in (c, c + 1)}
    try:
        logits = m(ids.to(DEV))[0]
        taps = dict(zip(sorted(m.block_tap_idxs), m.block_taps))
    finally:
        m.block_taps = None
        m.block_tap_idxs = None
    assert logits.shape == (1, T, 512) and torch.isfinite(logits.float()).all()
    assert sorted(taps) == [0, 1, 8, 9, 16, 17, 31, 32]
    sd = {k_: v_ for k_, v_ in m.state_dict().items()}
    rows = torch.cat([torch.arange(0, 512), torch.arange(65280, 65792), torch.arange(T - 1024, T)]).to(DEV)
    for i in check:
        kind = "attn" if i in FULL_131K["attn_layer_idxs"] else "hyena"
        t0 = time.time()
        u = taps[i].view(T, 4096)
        ref = (attn_block64 if kind == "attn" else hyena_block64)(u, sd, i, FULL_131K, rows)
        got = taps[i + 1].view(T, 4096)[rows]
        uin = u[rows].double()
        err, hu = rel_l2(got, ref), half_ulps(got, ref)
        upd = rel_l2(got.double().cpu() - uin.cpu(), ref.cpu() - uin.cpu())
        torch.cuda.synchronize()
        print(f"[131k block {i} ({kind})] output rel-L2 {err:.3e}, half-ulps {hu:.1f}, update rel-L2 {upd:.3e} "
              f"({time.time() - t0:.1f} s of fp64 restatement)")
        assert err <= PIN_BLOCK[kind] and hu <= PIN_BLOCK["hulp"] and upd <= PIN_BLOCK["upd"], (i, kind, err, hu, upd)
        del ref, u


def test_gpu_ref64_blocks_agree_with_the_cpu_oracle(full):
    """Pins the fp64 GPU restatements used at the big sizes to the CPU oracle where the latter can run: blocks 3 (Hyena) and
    8 (attention) on a BASELINE configs[0] input (T = 513), oracle in fp32 mode -- agreement to fp32 rounding."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    m = full["m131"]
    ids = acgt_ids(1, 512)
    m.block_taps = []
    try:
        m(ids.to(DEV))
        taps = [t.view(513, 4096) for t in m.block_taps]
    finally:
        m.block_taps = None
    o = oracle_for(full, FULL_131K, "fp32")
    sd = {k_: v_ for k_, v_ in m.state_dict().items()}
    rows = torch.arange(513, device=DEV)
    for i in (3, 8):
        kind = "attn" if i in FULL_131K["attn_layer_idxs"] else "hyena"
        cpu = (o.attn_block if kind == "attn" else o.hyena_block)(taps[i].float().cpu()[None], i, None)[0]
        gpu = (attn_block64 if kind == "attn" else hyena_block64)(taps[i], sd, i, FULL_131K, rows)
        d = rel_l2(cpu, gpu)
        upd = rel_l2(cpu.double() - taps[i].double().cpu(), gpu.cpu() - taps[i].double().cpu())
        print(f"[gpu_ref64 vs cpu oracle, block {i} ({kind})] rel-L2 {d:.2e}, update rel-L2 {upd:.2e}")
        assert d < 2e-5 and upd < 2e-4


# ---- (e) the score tolerance as a distribution -----------------------------------------------------------------------------
def test_score_rel_distribution_256_sequences_paired(full):
    """North-star: "logits within 1e-3 relative of the reference".  Element-wise no bf16 pipeline can meet that (a bf16 ulp is
    3.9e-3); the quantity evo reports is the per-sequence score [REF evo/scoring.py:84-96].  256 BASELINE configs[0]
    sequences (1 x 512 nt each, SURVEY 8(d) seeds 1234..1489) through the 32-layer engine on the RANDOM (non-contractive) weights, the
    fp32 oracle and the eager-bf16 oracle (= the reference's own arithmetic), oracles executed on the GPU.
    Round 6: the two norm routings -- folded into the dense layers (the default) and 65 separate passes (`fuse_norm = False`) -- are
    judged on PAIRED data: d_i = score_rel_folded(i) - score_rel_separate(i) on the same sequence against the same oracle value; the
    mean of d with its standard error is printed and pinned (round 5 reported 1.06e-3 vs 8.9e-4 on 64 sequences with no paired
    statistic).  Pins (tests/PARITY.md row 3): both routings mean <= 1.25e-3 and <= the eager-bf16 mean, max <= the eager-bf16 max;
    mean(d) <= 3 standard errors (the fold is not worse beyond noise -- otherwise the default must go back)."""
    m = full["m8"]
    N = 256
    ids = acgt_ids(N, 512)
    t0 = time.time()
    o32, o16 = gpu_oracle(full, FULL, "fp32"), gpu_oracle(full, FULL, "bf16")
    s_ref = torch.cat([score_of(o32(ids[i:i + 16])[0].float().cpu(), ids[i:i + 16]) for i in range(0, N, 16)])
    s_flo = torch.cat([score_of(o16(ids[i:i + 16])[0].float().cpu(), ids[i:i + 16]) for i in range(0, N, 16)])
    torch.cuda.synchronize()
    t_or = time.time() - t0

    def engine_scores(fold):
        was = m.ops.fuse_norm
        m.ops.fuse_norm = fold
        try:
            return torch.cat([score_of(m(ids[i:i + 16].to(DEV))[0].cpu(), ids[i:i + 16]) for i in range(0, N, 16)])
        finally:
            m.ops.fuse_norm = was
    s_f, s_u = engine_scores(True), engine_scores(False)
    rel_f, rel_u, rel_flo = ((s - s_ref).abs() / s_ref.abs() for s in (s_f, s_u, s_flo))
    sgn_f, sgn_u, sgn_flo = (((s - s_ref) / s_ref.abs()).mean().item() for s in (s_f, s_u, s_flo))
    d = rel_f - rel_u
    se = d.std(unbiased=True).item() / math.sqrt(N)
    print(f"[score distribution] {N} x 513 tokens, oracles (on the GPU) {t_or:.0f} s")
    for name, r_, sg in (("engine, norms folded (default)", rel_f, sgn_f), ("engine, fuse_norm = False", rel_u, sgn_u),
                         ("eager-bf16 oracle", rel_flo, sgn_flo)):
        print(f"[score distribution] {name}: score_rel mean {r_.mean():.3e} +- {r_.std(unbiased=True).item() / math.sqrt(N):.1e} "
              f"median {r_.median():.2e} max {r_.max():.2e}; signed mean {sg:+.2e}; first 64 mean {r_[:64].mean():.3e}")
    print(f"[score distribution] PAIRED folded - separate: mean {d.mean().item():+.3e} +- {se:.1e} (SE, n = {N}) = {d.mean().item() / se:+.2f} SE; "
          f"folded worse on {(d > 0).sum().item()} of {N} sequences")
    for r_ in (rel_f, rel_u):
        assert r_.mean().item() <= 1.25e-3 and r_.mean().item() <= rel_flo.mean().item()
        assert r_.max().item() <= rel_flo.max().item()
    assert d.mean().item() <= 3.0 * se, "the norm fold is worse than the separate passes beyond noise: the default must go back"


# ---- (f) the north-star tolerance on trained-like weights ------------------------------------------------------------------------
def test_contractive_profile_logits_and_scores_vs_fp32_and_eager_bf16(contractive):
    """North-star: "logits within 1e-3 relative of the reference".  On the default synthetic weights that sentence can be neither met nor
    refuted (32 blocks that each re-write the stream amplify rounding noise until ANY bf16 evaluation, the reference's included, sits
    0.1-0.2 rel-L2 from fp32 -- test (e) above judges the per-sequence score instead).  `profile="contractive"`
    (evo_amd/synthetic.py: blocks 1.. change the stream by ~7 % of its norm, as trained residual stacks do; real checkpoints are not
    reachable offline [REF evo/models.py:91-99]) removes the amplification.  16 BASELINE configs[0] sequences (1 x 512 nt) through the
    engine, the oracle in fp32 and the oracle in its eager-bf16 mode (= the arithmetic the reference runs: /root/reference/evo/scoring.py:81-84
    on a bf16 StripedHyena), oracles executed on the GPU by torch's eager kernels.  Reported and pinned: logits rel-L2 per sequence
    and the score's relative error, engine and eager-bf16, both against fp32."""
    m = contractive["m8"]
    ids = acgt_ids(16, 512)
    m.block_taps = []
    try:
        got = m(ids.to(DEV))[0].float().cpu()
        taps = [t.float() for t in m.block_taps]
    finally:
        m.block_taps = None
    ratios = [float((taps[i + 1] - taps[i]).norm() / taps[i].norm()) for i in range(32)]
    print(f"[contractive] block update / stream norm: block 0 {ratios[0]:.2f}, blocks 1..31 min {min(ratios[1:]):.3f} max {max(ratios[1:]):.3f}")
    assert 0.03 <= min(ratios[1:]) and max(ratios[1:]) <= 0.15
    from conftest import contractive_oracle
    o32, o16 = contractive_oracle(contractive, FULL, "fp32"), contractive_oracle(contractive, FULL, "bf16")
    ref = o32(ids)[0].float().cpu()
    flo = o16(ids)[0].float().cpu()
    assert ref.std() > 0.1                                                      # non-degenerate logits
    rl = lambda a: ((a.double() - ref.double()).flatten(1).norm(dim=1) / ref.double().flatten(1).norm(dim=1))   # noqa: E731
    e_eng, e_flo = rl(got), rl(flo)
    s_ref, s_got, s_flo = score_of(ref, ids), score_of(got, ids), score_of(flo, ids)
    r_eng, r_flo = (s_got - s_ref).abs() / s_ref.abs(), (s_flo - s_ref).abs() / s_ref.abs()
    print(f"[contractive] logits std {ref.std().item():.2f}; logits rel-L2 vs fp32: engine mean {e_eng.mean():.3e} max {e_eng.max():.3e} | "
          f"eager-bf16 reference arithmetic mean {e_flo.mean():.3e} max {e_flo.max():.3e}")
    print(f"[contractive] score rel vs fp32: engine mean {r_eng.mean():.3e} max {r_eng.max():.3e} | eager-bf16 mean {r_flo.mean():.3e} max {r_flo.max():.3e} "
          f"(north-star: 1e-3 relative -- {'met' if r_eng.max() <= 1e-3 else 'NOT met'} on the score by every sequence, "
          f"{'met' if e_eng.max() <= 1e-3 else 'not met'} on the logits' rel-L2; a bf16 ulp is 3.9e-3)")
    # the engine is at least as close to fp32 as the reference's own arithmetic, on every aggregate; and the amplification is gone
    # measured (round 5, MI355X): logits rel-L2 engine 1.62e-2 / eager-bf16 1.94e-2 (64 bf16 roundings of the stream accumulate as
    # sqrt(64) x 2e-3 -- no amplification left, and no bf16 pipeline gets below that); score rel engine mean 1.6e-4 max 5.5e-4, eager-bf16
    # mean 1.5e-4 max 4.5e-4 (16 samples: the two are the same distribution)
    assert e_eng.mean() <= e_flo.mean() and e_eng.max() <= 1.1 * e_flo.max() and e_eng.max() <= 2.5e-2
    assert r_eng.mean() <= max(1.5 * r_flo.mean(), 3e-4) and r_eng.max() <= 1.0e-3        # the north-star's 1e-3, on every sequence
