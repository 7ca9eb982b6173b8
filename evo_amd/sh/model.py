"""`stripedhyena.model.StripedHyena` mirror: the host side of the MI355X-native forward engine.

The class keeps the contract evo consumes [REF evo/models.py:146-150; evo/scoring.py:81;
evo/generation.py:105-155]:
    model = StripedHyena(config); model.load_state_dict(sd, strict=True)
    model.to_bfloat16_except_poles_residues(); model.to(device)
    logits, cache = model(input_ids, inference_params_dict=None, padding_mask=None)
    cache = model.initialize_inference_params()
with the state-dict key set of SURVEY.md section B.  All arithmetic is dispatched to an `ops` object:
`evo_amd.ops.HipOps` (hand-written gfx950 kernels over the C ABI, the dense layers included) in the product.
There is no CPU fallback here: with no `ops` injected and no usable HIP library the first forward raises.
(`tests/` inject an oracle-backed ops object to exercise this host logic on CPU.)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .cache import InferenceParams, RecurrentInferenceParams
from .utils import dotdict


def _cfg(config, key, default):
    v = config.get(key, None) if isinstance(config, dict) else getattr(config, key, None)
    return default if v is None else v


class _Params(nn.Module):
    """A bag of named parameters (keeps the upstream module/parameter names for state-dict parity)."""

    def __init__(self, **shapes):
        super().__init__()
        for name, (shape, dtype) in shapes.items():
            self.register_parameter(name, nn.Parameter(torch.empty(*shape, dtype=dtype), requires_grad=False))


class _MLP(nn.Module):
    def __init__(self, D, inner):
        super().__init__()
        bf = torch.bfloat16
        self.l1 = _Params(weight=((inner, D), bf))
        self.l2 = _Params(weight=((inner, D), bf))
        self.l3 = _Params(weight=((D, inner), bf))


class _HyenaBlock(nn.Module):
    """ParallelGatedConvBlock parameters."""

    def __init__(self, D, inner, state_size, filt_len):
        super().__init__()
        bf = torch.bfloat16
        self.pre_norm = _Params(scale=((D,), bf))
        self.post_norm = _Params(scale=((D,), bf))
        self.projections = _Params(weight=((3 * D, D), bf), bias=((3 * D,), bf))
        self.filter = _Params(short_filter_weight=((3 * D, 1, filt_len), bf), short_filter_bias=((3 * D,), bf),
                              D=((D,), bf), poles=((D, state_size, 1, 2), torch.float32),
                              residues=((D, state_size, 1, 2), torch.float32))
        self.out_filter_dense = _Params(weight=((D, D), bf), bias=((D,), bf))
        self.mlp = _MLP(D, inner)


class _MHA(nn.Module):
    def __init__(self, D, hd):
        super().__init__()
        bf = torch.bfloat16
        self.Wqkv = _Params(weight=((3 * D, D), bf), bias=((3 * D,), bf))
        self.out_proj = _Params(weight=((D, D), bf), bias=((D,), bf))
        self.rotary_emb = nn.Module()
        self.rotary_emb.register_buffer("inv_freq", torch.empty(hd // 2, dtype=torch.float32), persistent=True)


class _AttentionBlock(nn.Module):
    def __init__(self, D, inner, hd):
        super().__init__()
        bf = torch.bfloat16
        self.pre_norm = _Params(scale=((D,), bf))
        self.post_norm = _Params(scale=((D,), bf))
        self.inner_mha_cls = _MHA(D, hd)
        self.mlp = _MLP(D, inner)


class StripedHyena(nn.Module):
    def __init__(self, config, ops=None):
        super().__init__()
        if not isinstance(config, dict):
            raise TypeError("config must be a dict / dotdict")
        self.config = config if isinstance(config, dotdict) else dotdict(config)
        c = self.config
        self.vocab_size = int(_cfg(c, "vocab_size", 512))
        self.hidden_size = D = int(_cfg(c, "hidden_size", 4096))
        self.num_layers = int(_cfg(c, "num_layers", 32))
        self.attn_layer_idxs = list(_cfg(c, "attn_layer_idxs", [8, 16, 24]))
        self.hyena_layer_idxs = list(_cfg(c, "hyena_layer_idxs",
                                          [i for i in range(self.num_layers) if i not in self.attn_layer_idxs]))
        self.num_heads = int(_cfg(c, "num_attention_heads", 32))
        self.head_dim = D // self.num_heads
        self.state_size = int(_cfg(c, "state_size", 8))
        self.short_filter_length = int(_cfg(c, "short_filter_length", 3))
        self.eps = float(_cfg(c, "eps", 1e-6))
        mult = int(_cfg(c, "inner_size_multiple_of", 16))
        inner = _cfg(c, "inner_mlp_size", None)
        if inner is None:
            inner = mult * ((int(2 * D * 4 / 3) + mult - 1) // mult)
        self.inner_size = int(inner)
        self.rotary_base = float(_cfg(c, "rotary_emb_base", 10000.0))
        self.rotary_scaling = float(_cfg(c, "rotary_emb_scaling_factor", 1.0)) \
            if bool(_cfg(c, "use_interpolated_rotary_pos_emb", False)) else 1.0
        self.max_seqlen = int(_cfg(c, "max_seqlen", 8192))
        if self.state_size != 8 or self.short_filter_length != 3:
            raise ValueError("the gfx950 Hyena kernels are built for state_size 8 and short_filter_length 3")

        bf = torch.bfloat16
        self.embedding_layer = _Params(weight=((self.vocab_size, D), bf))
        self.norm = _Params(scale=((D,), bf)) if bool(_cfg(c, "final_norm", True)) else None
        self.unembed = nn.Module()
        self.unembed.weight = self.embedding_layer.weight          # tied [REF evo/models.py:132-137]
        blocks = []
        for i in range(self.num_layers):
            if i in self.attn_layer_idxs:
                blocks.append(_AttentionBlock(D, self.inner_size, self.head_dim))
            else:
                blocks.append(_HyenaBlock(D, self.inner_size, self.state_size, self.short_filter_length))
        self.blocks = nn.ModuleList(blocks)
        self._ops = ops
        self._packed = False
        self._rot_cache: Dict[Tuple[int, int, str], Tuple[torch.Tensor, torch.Tensor]] = {}

    # ------------------------------------------------------------------ state-dict / dtype / device
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = True):
        """Strict key/shape check, then ADOPT the given tensors (no 13 GB copy)."""
        own = dict(self.named_parameters(remove_duplicate=False))
        own.update(dict(self.named_buffers(remove_duplicate=False)))
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own]
        if "unembed.weight" in missing and "embedding_layer.weight" in state_dict:
            missing.remove("unembed.weight")
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for StripedHyena: missing keys {missing}, "
                               f"unexpected keys {unexpected}")
        if getattr(self, "_norms_folded", False):
            raise RuntimeError("load_state_dict on a model that holds one folded weight set (fold_norms_): build a fresh StripedHyena")
        for k, t in state_dict.items():
            if k not in own or k == "unembed.weight":
                continue
            if tuple(own[k].shape) != tuple(t.shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)} vs model {tuple(own[k].shape)}")
            own[k].data = t.detach()
        self._packed = False
        self._rot_cache.clear()
        from torch.nn.modules.module import _IncompatibleKeys
        return _IncompatibleKeys(missing, unexpected)

    def to_bfloat16_except_poles_residues(self):
        """bf16 everywhere except the fp32 poles / residues (and inv_freq) [REF evo/models.py:148]."""
        for k, p in self.named_parameters():
            if k.endswith("poles") or k.endswith("residues"):
                p.data = p.data.to(torch.float32)
            else:
                p.data = p.data.to(torch.bfloat16)
        for _, b in self.named_buffers():
            b.data = b.data.to(torch.float32)
        self._packed = False

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._packed = False
        self._rot_cache.clear()
        return out

    @property
    def device(self):
        return self.embedding_layer.weight.device

    @property
    def ops(self):
        if self._ops is None:
            from ..ops import default_ops
            self._ops = default_ops()          # raises if the HIP library / GPU is unavailable
        return self._ops

    def _pack(self):
        if getattr(self, "_norms_folded", False):
            raise RuntimeError("this StripedHyena holds ONE folded weight set (fold_norms_): its derived tensors cannot be rebuilt from the "
                               "parameters -- moving / re-typing / re-loading needs a fresh model")
        self._pack_impl()

    def _pack_impl(self):
        """Derived device-side layouts: fused [l1;l2] weight (one GEMM for the gated MLP) with the inner size
        zero-padded to a multiple of 256 (10928 -> 11008: hipBLASLt runs the l3 GEMM, K = inner, 14 % faster
        without a K tail and the fused l1|l2 GEMM 3 % faster; the padded rows/columns are exact zeros, so
        gelu(0)*0 = 0 flows through and the result is bit-identical), contiguous [3D,3] FIR taps and [D,8,2]
        poles/residues.  l1/l2/l3 keep their state-dict entries as views of the padded buffers."""
        for blk in self.blocks:
            l1, l2, l3 = blk.mlp.l1.weight, blk.mlp.l2.weight, blk.mlp.l3.weight
            inner, D_ = l1.shape
            ipad = ((inner + 255) // 256) * 256 if inner >= 1024 else inner
            w12 = l1.data.new_zeros(2 * ipad, D_)
            w12[:inner] = l1.data
            w12[ipad:ipad + inner] = l2.data
            w3 = l3.data.new_zeros(D_, ipad)
            w3[:, :inner] = l3.data
            l1.data = w12[:inner]
            l2.data = w12[ipad:ipad + inner]
            l3.data = w3[:, :inner]
            blk.mlp._w12 = w12
            blk.mlp._w3 = w3
            # the same rows regrouped for the one-launch gated form of the prefill path (GELU * gate in the dense layer's epilogue:
            # csrc/gemm.hip) are a copy (180 MB per layer at D = 4096: 5.8 GB for 32 layers; the decode kernels stream `w12` as it is):
            # built on the first prefill-sized call (_gate_pack), so that decode-only use never pays for it
            blk.mlp._w12g = None
            blk.mlp._w12g_f = None                            # ... and its norm-folded form (_folded), like the projections' below
            if isinstance(blk, _HyenaBlock):
                blk._wp_f = None
            else:
                blk.inner_mha_cls._wqkv_f = None
            blk.mlp._w12g_ok = (2 * ipad) % 256 == 0 and ipad % 32 == 0 and D_ % 64 == 0 and D_ >= 128
            if isinstance(blk, _HyenaBlock):
                f = blk.filter
                D = self.hidden_size
                f._fir_w = f.short_filter_weight.data.reshape(3 * D, self.short_filter_length).contiguous()
                f._poles = f.poles.data.reshape(D, self.state_size, 2).float().contiguous()
                f._residues = f.residues.data.reshape(D, self.state_size, 2).float().contiguous()
                # the [D,52,64] operand table of the matrix-core operator (54 MB per layer, 1.6 GB for the 29 Hyena layers of the 7B
                # model): built by _mfma_table on the first parallel Hyena call of the layer -- or up front by prepare() --, never for
                # decode-only use.  The operator reads the projection weight as it is (no regrouped copy).
                f._mfma_tab = None
                f._tab_prec = None                              # hyena_tables.table_precision's worst channel (decided with the table: _table_ok)
        self._packed = True

    def prepare(self, prefill: bool = True) -> "StripedHyena":
        """Builds the derived device-side tensors NOW instead of inside the first forward: the fused / padded MLP weights (always), and
        with `prefill` the regrouped l1 | l2 copies of the one-launch gated MLP (5.8 GB at 7B) and the Hyena layers' MFMA operand tables
        (1.6 GB) that prefill-sized batches use.  A server calls it at load time: an out-of-memory condition then surfaces before any
        KV cache or activation exists, the first request does not pay the packing, and the tensors are ordinary (not inference-mode)
        tensors whatever mode the first forward runs under.  Decode-only deployments pass prefill=False.
        Footprint at 7B: 12.9 GB of weights + 5.8 GB (l1 | l2 folded and regrouped) + 3.2 GB (folded projections / Wqkv) + 1.6 GB (operand
        tables).  A batch below 1,024 rows, a padding mask or `fuse_norm = False` later adds the UNFOLDED regrouped l1 | l2 (5.8 GB) on
        first use -- not covered by this call's out-of-memory check: `release_unused_weight_copies()` drops whichever set the current
        routing does not read, `fold_norms_()` keeps ONE weight set for good (14.5 GB)."""
        with torch.inference_mode(False), torch.no_grad():
            if not self._packed and not getattr(self, "_norms_folded", False):
                self._pack()
            if prefill:
                nf = getattr(self.ops, "fuse_norm", False) and hasattr(self.ops, "fold_norm_scale") and self.blocks[0].mlp._w12g_ok
                for blk in self.blocks:
                    if getattr(self, "_norms_folded", False):
                        pass                                     # (one weight set: nothing to derive but the operand tables)
                    elif nf:                                   # the norm-folded copies the prefill launches read (see _nf_ok)
                        self._folded(blk.mlp, "_w12g_f", blk.mlp._w12, blk.post_norm.scale, gate=True)
                        if isinstance(blk, _HyenaBlock):
                            self._folded(blk, "_wp_f", blk.projections.weight, blk.pre_norm.scale)
                        else:
                            self._folded(blk.inner_mha_cls, "_wqkv_f", blk.inner_mha_cls.Wqkv.weight, blk.pre_norm.scale)
                    else:
                        self._gate_pack(blk, 1 << 20)
                    if isinstance(blk, _HyenaBlock) and hasattr(self.ops, "hyena_ct") and self._table_ok(blk):
                        self._mfma_table(blk)
        return self

    # ------------------------------------------------------------------ one weight set (round 6)
    def fold_norms_(self) -> "StripedHyena":
        """ONE resident copy of every weight: the block norms' scale vectors are folded INTO the dense layers that consume them, in place --
        projections / Wqkv <- bf16(W diag(g_pre)), l1 | l2 <- bf16([W1; W2] diag(g_post)) kept only in the gated launch's row order, the
        scale vectors set to 1 -- and every derived copy is dropped (the unfolded regrouped l1 | l2, the folded projection copies).  The model
        is then the SAME function with unit norm scales (what the prefill path has computed since round 5: one rounding of the weight where
        the reference rounds the normalised activation); the decode launches take the same tensors (their RMSNorm multiplies by 1; the
        weight-streaming gate launches read l1 | l2 in the grouped order, include/evo_mi355x.h ABI 10).  Resident at 7B: 12.9 GB of weights +
        1.6 GB of Hyena operand tables instead of ~23.5 GB [REF evo/models.py:146-150: the reference holds one bf16 copy of the model].
        One-way: `state_dict()` afterwards returns the folded, unit-scale equivalent (l1 / l2 un-grouped on demand); moving the model to
        another device / dtype or loading another state dict needs a fresh model.  A serving deployment calls
        `model.to(device).prepare().fold_norms_()` once after loading."""
        if getattr(self, "_norms_folded", False):
            return self
        from ..ops import HipOps                                 # (fold_norm_scale / pack_gate_weights are pure tensor functions)
        ops = self.ops
        with torch.inference_mode(False), torch.no_grad():
            if not self._packed:
                self._pack()
            for blk in self.blocks:
                mlp = blk.mlp
                inner = mlp.l1.weight.shape[0]
                w12f = HipOps.fold_norm_scale(mlp._w12, blk.post_norm.scale.data)
                if mlp._w12g_ok and hasattr(ops, "pack_gate_weights"):
                    mlp._w12g = ops.pack_gate_weights(w12f)          # the ONLY copy of l1 | l2 from here on
                    mlp._w12 = None
                    mlp._l12_shape = (inner, w12f.shape[1])
                    mlp.l1.weight.data = w12f.new_empty(0)
                    mlp.l2.weight.data = w12f.new_empty(0)
                else:                                               # (toy dims outside the gated launch's contract: the plain order stays)
                    ipad = w12f.shape[0] // 2
                    mlp._w12 = w12f
                    mlp._w12g = None
                    mlp.l1.weight.data = w12f[:inner]
                    mlp.l2.weight.data = w12f[ipad:ipad + inner]
                mlp._w12g_f = None
                del w12f
                if isinstance(blk, _HyenaBlock):
                    blk.projections.weight.data = HipOps.fold_norm_scale(blk.projections.weight.data, blk.pre_norm.scale.data)
                    blk._wp_f = None
                else:
                    mha = blk.inner_mha_cls
                    mha.Wqkv.weight.data = HipOps.fold_norm_scale(mha.Wqkv.weight.data, blk.pre_norm.scale.data)
                    mha._wqkv_f = None
                blk.pre_norm.scale.data = torch.ones_like(blk.pre_norm.scale.data)
                blk.post_norm.scale.data = torch.ones_like(blk.post_norm.scale.data)
            self._norms_folded = True
            self._dgraph = None
        return self

    def release_unused_weight_copies(self) -> int:
        """Drops the derived weight copies the CURRENT routing does not read (a long-lived server that once ran a sub-1,024-row
        batch, a padding mask or `fuse_norm = False` holds the unfolded regrouped l1 | l2 (5.8 GB at 7B) beside the folded one that
        prepare() built; prepare()'s out-of-memory check covers only one of the two).  With the norms folded (the default) the unfolded
        regrouped copy goes, otherwise the folded ones; whatever is dropped is rebuilt on the next call that needs it.  Returns the bytes
        released.  (`fold_norms_()` is the permanent form: one weight set.)"""
        before = self.resident_bytes()
        fold = getattr(self.ops, "fuse_norm", False)
        for blk in self.blocks:
            if getattr(self, "_norms_folded", False):
                break
            if fold and getattr(blk.mlp, "_w12g_f", None) is not None:
                blk.mlp._w12g = None
            if not fold:
                blk.mlp._w12g_f = None
                if isinstance(blk, _HyenaBlock):
                    blk._wp_f = None
                else:
                    blk.inner_mha_cls._wqkv_f = None
        return before - self.resident_bytes()

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        if getattr(self, "_norms_folded", False):
            prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
            for i, blk in enumerate(self.blocks):
                mlp = blk.mlp
                if mlp._w12 is None:                              # l1 / l2 live in the grouped buffer only: un-group on demand (a copy)
                    inner, K = mlp._l12_shape
                    g = mlp._w12g.view(-1, 2, 32, K)
                    sd[f"{prefix}blocks.{i}.mlp.l1.weight"] = g[:, 0].reshape(-1, K)[:inner]
                    sd[f"{prefix}blocks.{i}.mlp.l2.weight"] = g[:, 1].reshape(-1, K)[:inner]
        return sd

    def resident_bytes(self) -> int:
        """Device bytes this model holds: parameters, buffers and every derived tensor hanging off its modules (fused / regrouped / folded
        weight copies, Hyena operand tables), each storage counted once."""
        seen, total = set(), 0

        def add(t):
            nonlocal total
            if isinstance(t, torch.Tensor) and t.numel():
                st = t.untyped_storage()
                key = (st.data_ptr(), st.nbytes())
                if key not in seen:
                    seen.add(key)
                    total += st.nbytes()
        for mod in self.modules():
            for t in list(mod._parameters.values()) + list(mod._buffers.values()):
                add(t)
            for k, v in vars(mod).items():
                if k.startswith("_") and isinstance(v, torch.Tensor):
                    add(v)
        return total

    # ------------------------------------------------------------------ caches
    def initialize_inference_params(self):
        """[REF evo/generation.py:116-120]"""
        return {
            "mha": InferenceParams(max_seqlen=self.max_seqlen, max_batch_size=1, seqlen_offset=0),
            "hyena": RecurrentInferenceParams(fir_filter_length=self.short_filter_length,
                                              state_dim=self.state_size, seqlen_offset=0),
        }

    def precompute_filters(self, L, device):
        """Upstream materialises h[D,L] here; this engine never builds the filter (it is evaluated through
        its modes on chip), so there is nothing to precompute.  Kept for API compatibility."""
        return None

    def _rotary(self, pos0: int, T: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos/sin [T, hd/2] f32 for absolute positions pos0..pos0+T-1: fp32 angles (positions divided by
        the interpolation factor for the 131k model [REF evo-1-131k-base_inference.yml:39-40]) rounded to
        bf16 values, as flash-attn caches them in the activation dtype."""
        key = (pos0, T, str(device))
        hit = self._rot_cache.get(key)
        if hit is not None:
            return hit
        hd = self.head_dim
        inv_freq = 1.0 / (self.rotary_base ** (torch.arange(0, hd, 2, dtype=torch.float32, device=device) / hd))
        t = torch.arange(pos0, pos0 + T, dtype=torch.float32, device=device)
        if self.rotary_scaling != 1.0:
            t = t / self.rotary_scaling
        freqs = torch.outer(t, inv_freq)
        cos = torch.cos(freqs).to(torch.bfloat16).float().contiguous()
        sin = torch.sin(freqs).to(torch.bfloat16).float().contiguous()
        if len(self._rot_cache) > 8:               # a decode step reuses its table for all attention layers
            self._rot_cache.clear()
        self._rot_cache[key] = (cos, sin)
        return cos, sin

    def _row_index(self, B: int, device) -> torch.Tensor:
        hit = getattr(self, "_rows", None)
        if hit is None or hit.numel() < B or hit.device != torch.device(device):
            hit = self._rows = torch.arange(max(B, 8), dtype=torch.int64, device=device)
        return hit[:B]

    def _inv_freq(self, dev) -> torch.Tensor:
        inv = getattr(self, "_inv_freq_dev", None)
        if inv is None or inv.device != torch.device(dev):
            hd = self.head_dim
            inv = 1.0 / (self.rotary_base ** (torch.arange(0, hd, 2, dtype=torch.float32, device=dev) / hd))
            self._inv_freq_dev = inv
        return inv

    def _rotary_dyn(self, pos: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Same table for positions held in device memory (int64 [n]: one per decode stream) -- no host read,
        graph-capturable.  Returns cos, sin [n, hd/2]."""
        hd = self.head_dim
        inv = self._inv_freq(pos.device)
        t = pos.to(torch.float32)
        if self.rotary_scaling != 1.0:
            t = t / self.rotary_scaling
        freqs = torch.outer(t, inv)
        return (torch.cos(freqs).to(torch.bfloat16).float().contiguous(),
                torch.sin(freqs).to(torch.bfloat16).float().contiguous())

    # ------------------------------------------------------------------ blocks
    DECODE_ROWS = 8          # batches this small take the fused single-token launches (csrc/gemv.hip; 5-8 rows at D = 4096 only)

    def _mixer_out_(self, blk, x2d, y, w, bias, mfma=False):
        """x += y @ w^T (the mixer's output projection); returns the bias still to be added (folded into the next
        RMSNorm pass for prefill-sized batches, into this launch for decode-sized ones)."""
        # round 4: the bias rides in the dense layer's epilogue at every batch size (x <- bf16(x + y W^T + b): one rounding where the
        # reference rounds twice); the post-mixer RMSNorm then reads x only (rmsnorm_kernel<false>: 4 D instead of 6 D bytes per token)
        if y.dim() == 4:                                     # the Hyena operator's blocked output (ops.hyena_ct)
            self.ops.linear_residual_yblk_(x2d, y, w, bias=bias)
            return None
        self.ops.linear_residual_(x2d, y, w, mfma=mfma, bias=bias)
        return None

    # ---- RMSNorm folded into the dense layers around it (round 5; csrc/gemm.hip NF, include/evo_mi355x.h) ---------------------------
    # The dense layer that writes the residual stream (mixer output projection, l3) also emits every row's 1 / (rms + eps) -- `rs` below,
    # handed from block to block --, and the layer that consumes the norm (Hyena projections, Wqkv, l1 | l2) reads the stream itself and
    # applies rs in its epilogue, with the norm's scale vector folded into a copy of its weight.  What remains of the 65 RMSNorm
    # passes of a forward: block 0's pre-norm (the embedding has no dense layer in front of it) and the final norm.
    def _nf_ok(self, M: int, mask) -> bool:
        ops = self.ops
        D = self.hidden_size
        if mask is not None or not getattr(ops, "fuse_norm", False) or not hasattr(ops, "nf_shape_ok"):
            return False
        m0 = self.blocks[0].mlp
        ipad = (m0._w12 if m0._w12 is not None else m0._w12g).shape[0] // 2
        return (ops.nf_shape_ok(M, D, D) and ops.nf_shape_ok(M, 3 * D, D) and ops.nf_shape_ok(M, 2 * ipad, D) and ops.nf_shape_ok(M, D, ipad)
                and getattr(self.blocks[0].mlp, "_w12g_ok", False) and getattr(ops, "mlp_gate_fused", False))

    def _folded(self, owner, name: str, w: torch.Tensor, g: torch.Tensor, gate: bool = False) -> torch.Tensor:
        """bf16(W diag(g)) of a norm-consuming layer (for l1 | l2: in the gated launch's row order), built on first use / by prepare()."""
        if getattr(self, "_norms_folded", False):            # one weight set: the parameters ARE the folded weights (fold_norms_)
            return owner._w12g if gate else w.data
        hit = getattr(owner, name, None)
        if hit is None or hit.device != w.device:
            with torch.inference_mode(False), torch.no_grad():
                hit = self.ops.fold_norm_scale(w.data, g.data)
                if gate:
                    hit = self.ops.pack_gate_weights(hit)
            setattr(owner, name, hit)
        return hit

    def _mixer_out_rs_(self, blk, x2d, y, w, bias):
        """_mixer_out_ + the post-mixer norm's row factors."""
        if y.dim() == 4:
            return self.ops.linear_residual_yblk_stats_(x2d, y, w, bias, self.eps)
        return self.ops.linear_residual_stats_(x2d, y, w, bias, self.eps)

    def _mlp_residual_rs_(self, blk, x2d, rs_post):
        """The block's MLP on the raw stream (post-norm as a row factor in the gated launch), residual added in place; returns the NEXT
        pre-norm's row factors (l3's epilogue statistic)."""
        ops = self.ops
        mlp = blk.mlp
        w12g_f = self._folded(mlp, "_w12g_f", mlp._w12, blk.post_norm.scale, gate=True)
        a = ops.mlp_gate_rs(x2d, rs_post, w12g_f, mlp._w12, blk.post_norm.scale, self.eps)
        return ops.linear_residual_stats_(x2d, a, mlp._w3, None, self.eps)

    def _gate_pack(self, blk, M: int):
        """The regrouped l1 | l2 weight of the one-launch gated MLP, built the first time a prefill-sized batch needs it."""
        mlp = blk.mlp
        if mlp._w12g is None and mlp._w12 is not None and M >= 256 and getattr(mlp, "_w12g_ok", False) and getattr(self.ops, "mlp_gate_fused", False) \
                and hasattr(self.ops, "pack_gate_weights"):
            mlp._w12g = self.ops.pack_gate_weights(mlp._w12)
        return mlp._w12g

    def _mlp_residual_(self, blk, x2d, bias, mask=None):
        ops = self.ops
        self._gate_pack(blk, x2d.shape[0])
        if mask is not None:                                             # upstream: (mixer out + u) * padding_mask
            if bias is not None:
                x2d.add_(bias)
                bias = None
            x2d.mul_(mask)
            n2 = ops.rmsnorm(x2d, None, blk.post_norm.scale, self.eps)
            a = ops.mlp_gate(n2, blk.mlp._w12, w12g=blk.mlp._w12g)
            ops.linear_residual_(x2d, a, blk.mlp._w3)
            return
        if bias is None:                                                 # decode: norm + l1/l2 + gate in one launch
            a = ops.mlp_gate(x2d, blk.mlp._w12, blk.post_norm.scale, self.eps, w12g=blk.mlp._w12g)
        else:
            n2 = ops.rmsnorm(x2d, bias, blk.post_norm.scale, self.eps)   # x += bias (in place); n2 = norm(x)
            a = ops.mlp_gate(n2, blk.mlp._w12, w12g=blk.mlp._w12g)
        ops.linear_residual_(x2d, a, blk.mlp._w3)

    def _mfma_hyena_ok(self, B: int, T: int) -> bool:
        """The single-pass matrix-core operator (csrc/hyena_ct.hip) serves every parallel (T > 1) Hyena call without a padding mask on
        the HIP backend -- scoring, cached prefill with carry-in / end state, sequence-parallel shards -- when the shape fits its launch
        contract; masks and very short inputs take the modal kernels.  (Shape part only: _hyena_ct_ok adds the tensors' checks.)"""
        ops = self.ops
        D, H = self.hidden_size, self.num_heads
        if not getattr(ops, "hyena_mfma", False) or not hasattr(ops, "hyena_ct") or D != H * 128:
            return False
        return T >= 32 and B * T >= 256

    def _mfma_table(self, blk):
        """The MFMA operand table of a Hyena block's filter (evo_amd/hyena_tables.py), built on first use."""
        f = blk.filter
        if getattr(f, "_mfma_tab", None) is None or f._mfma_tab.device != blk.projections.weight.device:
            from ..hyena_tables import mfma_operand_table
            f._mfma_tab = mfma_operand_table(f._poles, f._residues, f.D.data)
        return f._mfma_tab

    def _table_ok(self, blk) -> bool:
        """Can this layer's filter go through the bf16 hi / lo operand tables of csrc/hyena_ct.hip?  (evo_amd/hyena_tables.py: filters whose
        modes cancel to ~1 % amplify the splits' 2^-16 to a bf16 rounding; such a layer runs the modal kernels -- fp32 states, every regime.)
        Decided once per layer from the poles / residues alone; `ops.hyena_table_guard = False` skips the check (tests of the guard itself)."""
        f = blk.filter
        if not getattr(self.ops, "hyena_table_guard", True):
            return True
        if getattr(f, "_tab_prec", None) is None:
            from ..hyena_tables import table_precision
            f._tab_prec = float(table_precision(f._poles, f._residues, f.D.data).max())
        from ..hyena_tables import TABLE_TOL
        return f._tab_prec <= TABLE_TOL

    def _hyena_ct_ok(self, x2d, blk, B, T) -> bool:
        ops = self.ops
        w = blk.projections.weight
        return (getattr(ops, "hyena_ct_flag", False) and hasattr(ops, "hyena_ct") and x2d.is_cuda and w.dtype == torch.bfloat16
                and w.is_contiguous() and ops.zt_shape_ok(B, T, w.shape[0], w.shape[1]))

    def _hyena_block(self, i, blk, x2d, B, T, cache: Optional[RecurrentInferenceParams], mask=None, rs=None):
        """`mask` = (flat [B*T,1] bf16, [B,T] uint8) of upstream's padding_mask, or None: the projections output, the FIR
        output and the mixer output + residual are multiplied by it, as upstream's ParallelGatedConvBlock /
        engine.parallel_fir do [UPSTREAM-RECALLED; evo never passes one, SURVEY 8b]."""
        ops = self.ops
        D, H = self.hidden_size, self.num_heads
        f = blk.filter
        have_state = cache is not None and i in cache.fir_state_dict
        K1 = self.short_filter_length - 1
        if have_state and T == 1:                 # decode: norm + projections + FIR/modal step + gate in one launch
            y = ops.hyena_decode_fused(x2d, blk.pre_norm.scale, self.eps, blk.projections.weight, blk.projections.bias,
                                       cache.fir_state_dict[i], cache.state_dict[i], f._fir_w, f.short_filter_bias,
                                       f._poles, f._residues, f.D, H)
        elif mask is None and self._mfma_hyena_ok(B, T) and self._hyena_ct_ok(x2d, blk, B, T) and self._table_ok(blk):
            # the whole operator in ONE pass on the matrix cores (csrc/hyena_ct.hip) -- scoring and cached prefill (carry-in state + FIR
            # history in, end state out) alike.
            # z CHANNEL-MAJOR.  The pre-norm writes its rows in z^T's position order (HipOps.zt_layout: batch
            # rows padded to a multiple of 64 positions, or -- T = 512 k + r, the bench shapes -- unpadded rows of 512 k positions with
            # the last r tokens of every row in a tail block), the projection's dense layer runs with swapped operands (result = z^T, the weight as it is -- no
            # regrouped copy) and the operator loads a lane's eight steps of a channel as 16 contiguous bytes straight into
            # registers (no window in LDS); y BLOCKED: the operator stores whole cache lines and the output projection's dense layer
            # gathers them -- no row-major y exists on this path.
            table = self._mfma_table(blk)
            nf = self._nf_ok(B * T, mask)
            pb = None if blk.projections.bias is None else blk.projections.bias.data
            # T = 512 k + 1 (a BOS token in front of 2^k nucleotides), scoring: the one token behind the whole tiles of every row does not
            # go through the operator as a ragged tile (one valid step at a full tile's issue time: 8 of 136 tile steps at 8 x 8,193) but
            # through the fused single-token launch of the decode path, from the operator's end state (ops.hyena_tail_split)
            Tm, _, _, r_tail = ops.zt_layout(B, T)
            split = (cache is None and r_tail == 1 and getattr(ops, "hyena_tail_split", False) and B <= self.DECODE_ROWS
                     and (B <= 4 or D == 4096) and pb is not None)
            if nf and rs is not None and ops.zt_stream_rows_ok(B, T):
                # pre-norm folded: the projection reads the stream itself (z^T layouts without pad positions inside the main area: no padded copy either)
                wp_f = self._folded(blk, "_wp_f", blk.projections.weight, blk.pre_norm.scale)
                zt = ops.linear_t_rs(x2d, rs, wp_f, pb, blk.projections.weight.data, blk.pre_norm.scale, self.eps, B, T, tail=not split)
            else:
                xp = ops.rmsnorm_rows(x2d, blk.pre_norm.scale, self.eps, B, T)
                zt = ops.linear_t(xp, blk.projections.weight.data, pb, B, T, tail=not split)
            yb = ops.yblk_empty(B * T, D, zt.device)
            if split:
                y, state = ops.hyena_ct(zt, B, T, f._fir_w, f.short_filter_bias, table, H, want_state=True, poles=f._poles, y_blk=yb,
                                        main_only=True)
                ix = self._tail_split_index(B, T, zt.device)
                fir = zt[ix["zblk"], :, 256 - K1:].contiguous()                         # [B, 3 D, 2]: steps Tm - 2, Tm - 1 of every row, oldest first
                x_tail = x2d.view(B, T, D)[:, Tm].contiguous()                          # the raw stream's rows (the launch norms them itself)
                y_tail = ops.hyena_decode_fused(x_tail, blk.pre_norm.scale, self.eps, blk.projections.weight, blk.projections.bias, fir, state,
                                                f._fir_w, f.short_filter_bias, f._poles, f._residues, f.D, H)
                yb[ix["yblk"], :, ix["yrow"], :] = y_tail.view(B, D // 16, 16)          # rows b T + Tm of the blocked y
            elif cache is None:
                y = ops.hyena_ct(zt, B, T, f._fir_w, f.short_filter_bias, table, H, y_blk=yb)
            else:
                halo = s0 = None
                if have_state:              # continue a cached prefix with more than one token
                    halo = cache.fir_state_dict[i].transpose(1, 2).contiguous()
                    s0 = cache.state_dict[i]
                y, state = ops.hyena_ct(zt, B, T, f._fir_w, f.short_filter_bias, table, H, z_halo=halo, s0=s0,
                                        want_state=True, poles=f._poles, y_blk=yb)
                cache.fir_state_dict[i] = ops.zt_rows(zt, B, T, T - K1, K1).transpose(1, 2).contiguous()   # [B, 3D, 2]
                cache.state_dict[i] = state
            if nf:
                return self._mlp_residual_rs_(blk, x2d, self._mixer_out_rs_(blk, x2d, y, blk.out_filter_dense.weight, blk.out_filter_dense.bias))
            self._mlp_residual_(blk, x2d, self._mixer_out_(blk, x2d, y, blk.out_filter_dense.weight, blk.out_filter_dense.bias), None)
            return None
        else:
            z = ops.norm_linear(x2d, blk.pre_norm.scale, self.eps, blk.projections.weight, blk.projections.bias)   # [B*T, 3D]
            if mask is not None:
                z.mul_(mask[0])
            z3 = z.view(B, T, 3 * D)
            halo = s0 = None
            if have_state:                      # continue a cached prefix with more than one token
                halo = cache.fir_state_dict[i].transpose(1, 2).contiguous()
                s0 = cache.state_dict[i]
            kw = {} if mask is None else {"mask": mask[1]}
            y3, state = ops.hyena_prefill(z3, f._fir_w, f.short_filter_bias, f._poles, f._residues, f.D, H,
                                          z_halo=halo, s0=s0, want_state=cache is not None, **kw)
            y = y3.view(B * T, D)
            if cache is not None:
                tail = z3[:, -K1:, :]
                if halo is not None and T < K1:
                    tail = torch.cat([halo[:, T:], tail], dim=1)
                elif T < K1:
                    tail = torch.cat([z3.new_zeros(B, K1 - T, 3 * D), tail], dim=1)
                cache.fir_state_dict[i] = tail.transpose(1, 2).contiguous()      # [B, 3D, 2]
                cache.state_dict[i] = state
        self._mlp_residual_(blk, x2d, self._mixer_out_(blk, x2d, y, blk.out_filter_dense.weight, blk.out_filter_dense.bias),
                            None if mask is None else mask[0])
        return None

    def _tail_split_index(self, B: int, T: int, device):
        """Index tensors of the tail-split routing (_hyena_block), cached per (B, T): the z^T block holding the last two main steps of every
        batch row, and block / row of the blocked y for the rows b T + Tm."""
        key = (B, T, str(device))
        cur = getattr(self, "_tail_ix", None)
        if cur is None or cur[0] != key:
            Tm = self.ops.zt_layout(B, T)[0]
            b = torch.arange(B, device=device)
            R = b * T + Tm
            cur = (key, {"zblk": (b * Tm + Tm - 1) // 256, "yblk": R // self.ops.YBLK, "yrow": R % self.ops.YBLK})
            self._tail_ix = cur
        return cur[1]

    def _kv_buffer(self, cache: InferenceParams, i: int, B: int, need: int, like: torch.Tensor):
        H, hd = self.num_heads, self.head_dim
        kv = cache.key_value_memory_dict.get(i)
        cap_b = max(int(cache.max_batch_size or 1), B)
        if kv is None or kv.shape[0] < B or kv.shape[1] < need or kv.device != like.device:
            cap_t = max(int(cache.max_seqlen or 0), need)
            if kv is not None and kv.shape[1] < need:
                cap_t = max(cap_t, 2 * kv.shape[1])           # grow geometrically past max_seqlen (a20)
            new = torch.zeros(cap_b, cap_t, 2, H, hd, dtype=like.dtype, device=like.device)
            if kv is not None:
                b0, t0 = min(kv.shape[0], cap_b), min(kv.shape[1], cap_t)
                new[:b0, :t0] = kv[:b0, :t0].to(like.device)
            cache.key_value_memory_dict[i] = kv = new
        return kv

    def _attn_block(self, i, blk, x2d, B, T, cache: Optional[InferenceParams], mask=None, rs=None):
        ops = self.ops
        D, H, hd = self.hidden_size, self.num_heads, self.head_dim
        mha = blk.inner_mha_cls
        if mask is not None:                      # upstream AttentionBlock: u * padding_mask before and after the mixer
            x2d.mul_(mask[0])
        nf = T > 1 and self._nf_ok(B * T, mask)
        if nf and rs is not None:
            wqkv_f = self._folded(mha, "_wqkv_f", mha.Wqkv.weight, blk.pre_norm.scale)
            qkv = ops.linear_rs(x2d, rs, wqkv_f, mha.Wqkv.bias, mha.Wqkv.weight, blk.pre_norm.scale, self.eps).view(B, T, 3, H, hd)
        else:
            qkv = ops.norm_linear(x2d, blk.pre_norm.scale, self.eps, mha.Wqkv.weight, mha.Wqkv.bias, mfma=True).view(B, T, 3, H, hd)
        off = int(cache.seqlen_offset) if cache is not None else 0
        pos = getattr(cache, "pos_tensor", None) if cache is not None else None
        q = qkv[:, :, 0]
        # round 6: the rotary kernel folds softmax_scale * log2(e) into its one rounding of q and the attention kernels take scores as
        # exponents (ops.attn_prescale; csrc/attn_w64.hip PRE: no per-score multiply) -- q is never cached, so K / V and the KV cache are untouched
        pre = bool(getattr(ops, "attn_prescale", False)) and hasattr(ops, "attn_q_scale")
        rk = {"q_scale": ops.attn_q_scale(hd)} if pre else {}
        ak = {"prescaled": True} if pre else {}
        if pos is not None and T == 1:
            # position-independent decode step (hipGraph replay, continuous batching): one position PER ROW, in
            # device memory.  The rotary kernel indexes its table by token, so the B rows are presented as one
            # sequence of B tokens with the per-row table.
            kv = cache.key_value_memory_dict[i][:B]
            if hasattr(ops, "rope_append_decode") and pos.numel() == B:
                # rotary at each row's position + the KV append in one launch (no per-step cos / sin table)
                ops.rope_append_decode(qkv, kv, pos, self._inv_freq(x2d.device), self.rotary_scaling, **rk)
            else:
                cos, sin = getattr(cache, "_rot_dyn", None) or self._rotary_dyn(pos)
                ops.rope_(qkv.view(1, B, 3, H, hd), cos, sin, **rk)
                kv[self._row_index(B, x2d.device), pos] = qkv[:, 0, 1:3]
            a = ops.attention_decode(q, kv[:, :, 0], kv[:, :, 1], pos=pos, **ak).view(B, D)
        else:
            cos, sin = self._rotary(off, T, x2d.device)
            ops.rope_(qkv, cos, sin, **rk)
            if cache is not None:
                kv = self._kv_buffer(cache, i, B, off + T, qkv)
                kv[:B, off:off + T].copy_(qkv[:, :, 1:3])
                k = kv[:B, : off + T, 0]
                v = kv[:B, : off + T, 1]
            else:
                k, v = qkv[:, :, 1], qkv[:, :, 2]
            if T == 1 and cache is not None:
                a = ops.attention_decode(q, k, v, **ak).view(B, D)      # split-K over the KV cache
            else:
                a = ops.attention(q, k, v, off, **ak).view(B * T, D)
        if nf:
            return self._mlp_residual_rs_(blk, x2d, self._mixer_out_rs_(blk, x2d, a, mha.out_proj.weight, mha.out_proj.bias))
        self._mlp_residual_(blk, x2d, self._mixer_out_(blk, x2d, a, mha.out_proj.weight, mha.out_proj.bias, mfma=True),
                            None if mask is None else mask[0])
        return None

    # ------------------------------------------------------------------ forward
    max_rows_per_pass = 160 * 1024     # rows (B T) of one stateless pass; larger batches run in row groups (hidden_states)

    def _row_groups(self, B: int, T: int):
        """How a stateless pass over B rows of T tokens is run: [B] (one pass) or the sizes of the row groups run one after the other.
        Batch rows are independent and every launch is row-independent, so grouping changes no row's arithmetic except through which
        rows fall beyond a multiple of 256 (they take the weight-streaming launches).  Two reasons to group:
        (1) more rows than `max_rows_per_pass`: the persistent dense layers address their operands with 32-bit offsets (4 GiB: z^T at
            174 k positions, l3's input at 195 k rows); beyond that a pass would leave them for the library / three-launch routing --
            e.g. scripts/score.py's default batch of 32 sequences at 8 k nt;
        (2) a pass of >= 64 k rows whose rows beyond a multiple of 256 do not fit the fused single-token launches (> 8 rows, e.g.
            16 x 8,193): its norms and gates would run unfused.
        Groups are balanced, as few as fit, and preferably with <= 8 such rows each, as long as every group keeps >= 32 k rows (enough
        tiles for 256 CUs; the per-group launch overhead is ~1e-4 of its time)."""
        cap = max(1, self.max_rows_per_pass // T)

        def fits(p):
            return (p * T) % 256 <= self.DECODE_ROWS

        if B <= cap and (fits(B) or B * T < 65536):
            return [B]
        n0 = (B + cap - 1) // cap

        def split(n):
            q, r = divmod(B, n)
            return [q + 1] * r + [q] * (n - r)

        n = n0
        while n <= B and (B // n) * T >= 32768:
            g = split(n)
            if all(fits(p) for p in set(g)):
                return g
            n += 1
        return split(n0)

    def hidden_states(self, x: torch.Tensor, inference_params_dict=None, padding_mask=None) -> torch.Tensor:
        """ids [B,T] -> final-norm hidden states [B*T, D] (logits = hidden @ E^T)."""
        if not self._packed:
            self._pack()
        if x.dim() != 2:
            raise ValueError("input_ids must be [batch, length]")
        B, T = x.shape
        ops = self.ops
        if padding_mask is not None and tuple(padding_mask.shape) != (B, T):
            raise ValueError(f"padding_mask must be [batch, length] = {(B, T)}, got {tuple(padding_mask.shape)}")
        groups = self._row_groups(B, T) if inference_params_dict is None else [B]
        if len(groups) > 1:
            out, b0 = None, 0
            for nb in groups:
                pm = None if padding_mask is None else padding_mask[b0:b0 + nb]
                hg = self.hidden_states(x[b0:b0 + nb], None, pm)
                if out is None:
                    out = torch.empty(B * T, hg.shape[1], dtype=hg.dtype, device=hg.device)
                out[b0 * T:(b0 + nb) * T] = hg
                b0 += nb
            return out
        h = ops.embed(x.to(self.device), self.embedding_layer.weight)             # [B*T, D]
        mask = None
        if padding_mask is not None:
            pm = padding_mask.to(self.device) != 0
            mask = (pm.reshape(B * T, 1).to(h.dtype), pm.to(torch.uint8).contiguous())
        mha_c = inference_params_dict["mha"] if inference_params_dict is not None else None
        hy_c = inference_params_dict["hyena"] if inference_params_dict is not None else None
        dyn = T == 1 and mha_c is not None and getattr(mha_c, "pos_tensor", None) is not None
        if dyn and not hasattr(ops, "rope_append_decode"):   # (fallback path) one rotary table per decode step for all layers
            mha_c._rot_dyn = self._rotary_dyn(mha_c.pos_tensor)
        rs = None
        taps = getattr(self, "block_taps", None)     # debugging / parity hook: residual stream entering every block
        tap_idxs = getattr(self, "block_tap_idxs", None)   # ... or only the listed ones (index num_layers = the final stream)
        try:
            for i, blk in enumerate(self.blocks):
                if taps is not None and (tap_idxs is None or i in tap_idxs):
                    taps.append(h.clone())
                # rs: 1 / (rms + eps) of every row of h, from the epilogue of the dense layer that wrote it (None: the block norms h itself)
                if isinstance(blk, _AttentionBlock):
                    rs = self._attn_block(i, blk, h, B, T, mha_c, mask, rs)
                else:
                    rs = self._hyena_block(i, blk, h, B, T, hy_c, mask, rs)
            if taps is not None and (tap_idxs is None or self.num_layers in tap_idxs):
                taps.append(h.clone())
        finally:
            if dyn:
                mha_c._rot_dyn = None
        if self.norm is not None:
            h = ops.rmsnorm(h, None, self.norm.scale, self.eps)
        return h

    @torch.no_grad()
    def forward(self, x, inference_params_dict=None, padding_mask=None):
        """(logits [B,T,V] bf16, cache-or-None).  `padding_mask` [B,T] (1 = token, 0 = pad) is applied as upstream's
        stateless_forward applies it (block inputs / FIR output / mixer output multiplied by it); evo itself never passes
        one [REF evo/scoring.py:81; evo/generation.py:152-155] -- without it pads are ordinary tokens.  It is a scoring
        option: with a cache the call is upstream's stateful_forward, which ignores the mask -- so does this (with a warning)."""
        B, T = x.shape
        if padding_mask is not None and inference_params_dict is not None:
            # upstream routes this call to stateful_forward, which never looks at the mask: same here (a caller written against
            # upstream keeps working); the warning says that pads then enter the carried modal / FIR / KV state as ordinary tokens
            import warnings
            warnings.warn("padding_mask is ignored when inference_params_dict is given (as in upstream's stateful_forward)")
            padding_mask = None
        if padding_mask is not None:
            h = self.hidden_states(x, inference_params_dict, padding_mask)
            return self.ops.linear(h, self.unembed.weight, None).view(B, T, self.vocab_size), inference_params_dict
        if T == 1 and inference_params_dict is not None and self._graph_eligible(inference_params_dict):
            logits = self._graph_decode_step(x, inference_params_dict)
            if logits is not None:
                return logits, inference_params_dict
        h = self.hidden_states(x, inference_params_dict)
        logits = self.ops.linear(h, self.unembed.weight, None).view(B, T, self.vocab_size)
        return logits, inference_params_dict

    # ------------------------------------------------------------------ hipGraph-captured decode step
    # One decode step is ~270 short launches (the reference pays that per token too); at batch 1 the step is
    # launch-bound (5.2 ms against a 2.1 ms weight-streaming floor).  The step is therefore captured ONCE into a
    # hipGraph whose kernels read the token position from device memory, and replayed for every token.
    decode_graph = True          # set False (or EVO_AMD_DECODE_GRAPH=0) to run every decode step eagerly

    def _graph_eligible(self, ipd) -> bool:
        import os
        if not self.decode_graph or os.environ.get("EVO_AMD_DECODE_GRAPH", "1") == "0":
            return False
        if getattr(self.ops, "name", "") != "hip-gfx950" or getattr(self.ops, "timer", None) is not None:
            return False
        hy, mha = ipd["hyena"], ipd["mha"]
        # every layer must already hold state (i.e. a prefill happened), on this device
        return all(i in hy.fir_state_dict for i in self.hyena_layer_idxs) and \
            all(i in mha.key_value_memory_dict for i in self.attn_layer_idxs)

    def _graph_decode_step(self, x, ipd):
        mha, hy = ipd["mha"], ipd["hyena"]
        B = x.shape[0]
        off = int(mha.seqlen_offset)
        st = getattr(self, "_dgraph", None)
        cap = min(mha.key_value_memory_dict[i].shape[1] for i in self.attn_layer_idxs) if self.attn_layer_idxs else 1 << 62
        # The captured graph is bound to the ADDRESSES of the KV, modal-state and FIR-state tensors.  `st` keeps strong
        # references to the cache objects and to every captured tensor and compares by identity: a later cache can then
        # neither recycle a CPython id() nor a caching-allocator address of a freed buffer.
        same_bufs = (st is not None and st["B"] == B and st["mha"] is mha and st["hyena"] is hy
                     and all(mha.key_value_memory_dict.get(i) is t for i, t in st["kv"].items())
                     and all(hy.state_dict.get(i) is t for i, t in st["iir"].items())
                     and all(hy.fir_state_dict.get(i) is t for i, t in st["fir"].items()))
        if off + 1 > cap or not same_bufs:
            st = None
            self._dgraph = None                      # drops the references the stale graph held
            if off + 1 > cap or any(mha.key_value_memory_dict[i].shape[0] < B for i in self.attn_layer_idxs):
                for i in self.attn_layer_idxs:      # grow outside the graph, with headroom for the tokens to come
                    self._kv_buffer(mha, i, B, off + 1 + 4096, mha.key_value_memory_dict[i])
        try:
            if st is None:
                warm = getattr(self, "_dgraph_warm", None)
                if warm is None or warm[0] is not mha or warm[1] is not hy or warm[2] != B:
                    self._dgraph_warm = (mha, hy, B)  # first step with these caches runs eagerly (warms M=B GEMMs)
                    return None
                dev = self.device
                st = {"B": B, "mha": mha, "hyena": hy,
                      "ids": torch.zeros(B, 1, dtype=torch.int64, device=dev),
                      "pos": torch.zeros(B, dtype=torch.int64, device=dev),
                      "kv": {i: mha.key_value_memory_dict[i] for i in self.attn_layer_idxs},
                      "iir": {i: hy.state_dict[i] for i in self.hyena_layer_idxs},
                      "fir": {i: hy.fir_state_dict[i] for i in self.hyena_layer_idxs}}
                st["ids"].copy_(x)
                st["pos"].fill_(off)
                mha.pos_tensor = st["pos"]
                self._row_index(B, dev)             # (allocated outside the capture)
                self._inv_freq(dev)
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g):
                        h = self.hidden_states(st["ids"], ipd)
                        st["logits"] = self.ops.linear(h, self.unembed.weight, None).view(B, 1, self.vocab_size)
                        st["pos"].add_(1)
                finally:
                    mha.pos_tensor = None
                # the eager decode path updates every state in place; a capture that re-bound one would replay into a
                # buffer the cache no longer holds
                assert all(hy.state_dict[i] is t for i, t in st["iir"].items())
                assert all(hy.fir_state_dict[i] is t for i, t in st["fir"].items())
                st["graph"] = g
                st["next"] = off
                self._dgraph = st
            st["ids"].copy_(x)
            if st["next"] != off:
                st["pos"].fill_(off)
            st["graph"].replay()
            self.decode_graph_replays = getattr(self, "decode_graph_replays", 0) + 1
            st["next"] = off + 1
            return st["logits"].clone()
        except Exception as e:  # noqa: BLE001   capture is an optimisation: fall back to eager HIP launches
            import warnings
            warnings.warn(f"decode hipGraph disabled ({type(e).__name__}: {e}); continuing with eager launches")
            self.decode_graph = False
            self._dgraph = None
            return None

    def release_decode_graph(self):
        """Drop the captured decode step and the strong references it holds to the cache objects and to every KV / modal /
        FIR state tensor it was captured on (6.4 GB of KV cache after a 131k-context generation).  Called at the end of
        `Generator.generate` / `DecodePool.generate`; a later decode step on a live cache simply captures again."""
        self._dgraph = None
        self._dgraph_warm = None

    # upstream names, kept for callers that reach for them
    def stateless_forward(self, x, padding_mask=None):
        return self.forward(x, None, padding_mask)

    def stateful_forward(self, x, inference_params_dict=None):
        return self.forward(x, inference_params_dict)
