"""ctypes binding of libevo_mi355x.so (include/evo_mi355x.h) and the op set the StripedHyena host code calls.

`HipOps` is the ONLY compute backend the product ships: it raises if the shared library cannot be
loaded or a tensor is not on a ROCm device -- there is no CPU / eager fallback.  Every kernel of a scoring or generation
step -- the dense layers included (csrc/gemm.hip, csrc/gemv.hip) -- is a hand-written gfx950 kernel reached through the C ABI with
raw device pointers and the caller's current stream; hipBLASLt (through `torch.addmm`) serves only shapes no 7B layer has and the
in-process A/B knob `all_gemm_mfma = False`.
"""
from __future__ import annotations

import ctypes
import math
import os
import threading
from typing import Optional, Tuple

import torch

from . import _build

_c = ctypes
_PTR = _c.c_void_p
_I64 = _c.c_int64
_F32 = _c.c_float

_SIGNATURES = {
    "evo_abi_version": ([], _c.c_int),
    "evo_embed_bf16": ([_PTR, _PTR, _PTR, _I64, _I64, _I64, _PTR, _PTR], _c.c_int),
    "evo_rmsnorm_bf16": ([_PTR, _PTR, _PTR, _PTR, _I64, _I64, _F32, _PTR], _c.c_int),
    "evo_hyena_seg_state": ([_PTR, _PTR, _PTR, _PTR, _PTR, _PTR, _PTR, _I64, _I64, _I64, _I64, _I64, _PTR], _c.c_int),
    "evo_hyena_carry_scan": ([_PTR, _PTR, _PTR, _PTR, _I64, _I64, _I64, _I64, _PTR], _c.c_int),
    "evo_hyena_carry_add": ([_PTR, _PTR, _PTR, _I64, _I64, _I64, _I64, _PTR], _c.c_int),
    "evo_hyena_apply": ([_PTR] * 10 + [_I64] * 5 + [_PTR], _c.c_int),
    "evo_hyena_step": ([_PTR] * 9 + [_I64] * 3 + [_PTR], _c.c_int),
    "evo_rope_qk_bf16": ([_PTR, _PTR, _PTR, _I64, _I64, _I64, _I64, _F32, _PTR], _c.c_int),
    "evo_attn_fwd_causal_bf16": ([_PTR] * 4 + [_I64] * 14 + [_F32, _PTR, _PTR], _c.c_int),
    "evo_attn_decode_bf16": ([_PTR] * 4 + [_I64] * 11 + [_PTR] * 3 + [_I64, _F32, _PTR], _c.c_int),
    "evo_linear_small_m_bf16": ([_PTR] * 5 + [_I64] * 3 + [_PTR, _I64, _PTR], _c.c_int),
    "evo_linear_mfma_bf16": ([_PTR] * 5 + [_I64] * 3 + [_PTR], _c.c_int),
    "evo_mlp_gate_mfma_bf16": ([_PTR] * 3 + [_I64] * 3 + [_PTR], _c.c_int),
    "evo_hyena_ct": ([_PTR] * 9 + [_I64] * 13 + [_PTR], _c.c_int),
    "evo_linear_t_mfma_bf16": ([_PTR] * 4 + [_I64] * 3 + [_PTR], _c.c_int),
    "evo_rmsnorm_rows_bf16": ([_PTR, _PTR, _PTR, _PTR, _I64, _I64, _F32, _I64, _I64, _I64, _I64, _PTR], _c.c_int),
    "evo_linear_xblk_mfma_bf16": ([_PTR] * 5 + [_I64] * 3 + [_PTR], _c.c_int),
    "evo_linear_mfma_nf_bf16": ([_PTR] * 7 + [_I64] * 4 + [_PTR], _c.c_int),
    "evo_linear_xblk_mfma_nf_bf16": ([_PTR] * 6 + [_I64] * 4 + [_PTR], _c.c_int),
    "evo_mlp_gate_mfma_nf_bf16": ([_PTR] * 4 + [_I64] * 3 + [_PTR], _c.c_int),
    "evo_linear_t_mfma_nf_bf16": ([_PTR] * 5 + [_I64] * 6 + [_PTR], _c.c_int),
    "evo_rms_finalize_f32": ([_PTR, _I64, _I64, _PTR, _I64, _I64, _I64, _F32, _PTR, _PTR], _c.c_int),
    "evo_probe_copy_f4": ([_PTR, _PTR, _I64, _PTR], _c.c_int),
    "evo_probe_mfma_bf16": ([_PTR, _I64, _I64, _PTR], _c.c_int),
    "evo_mlp_gate_small_m_bf16": ([_PTR] * 3 + [_I64] * 4 + [_PTR], _c.c_int),
    "evo_norm_linear_small_m_bf16": ([_PTR] * 5 + [_I64] * 3 + [_c.c_float, _PTR], _c.c_int),
    "evo_norm_mlp_gate_small_m_bf16": ([_PTR] * 4 + [_I64] * 3 + [_c.c_float, _I64, _PTR], _c.c_int),
    "evo_hyena_decode_fused_small_m": ([_PTR] * 12 + [_I64] * 3 + [_c.c_float, _PTR], _c.c_int),
    "evo_gelu_gate_bf16": ([_PTR, _PTR, _I64, _I64, _PTR], _c.c_int),
    "evo_logprob_entropy": ([_PTR, _I64, _PTR, _PTR, _PTR, _I64, _I64, _PTR], _c.c_int),
    "evo_unembed_logprob_bf16": ([_PTR] * 5 + [_I64] * 3 + [_PTR], _c.c_int),
    "evo_rope_append_decode_bf16": ([_PTR] * 4 + [_F32] + [_I64] * 7 + [_F32, _PTR], _c.c_int),
}

_LIB = None
ABI_VERSION = 10         # must equal EVO_ABI_VERSION in include/evo_mi355x.h (bumped on every signature change)


class EvoLibraryError(RuntimeError):
    pass


def load_library(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load libevo_mi355x.so and type every entry point.  Raises EvoLibraryError when it is absent and
    cannot be built -- the product path never degrades to a non-HIP implementation."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.lib_path()
    if (not path.exists() or (_build.is_stale() and os.environ.get("EVO_AMD_NO_REBUILD") != "1")) and build_if_missing:
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001
            if not path.exists():
                raise EvoLibraryError(f"libevo_mi355x.so is missing and could not be built: {e}") from e
            import warnings                                  # a stale binary is usable only if its ABI still matches
            warnings.warn(f"{path} is older than its sources and the rebuild failed ({e}); loading the stale library "
                          f"(its ABI version is checked below)")
    if not path.exists():
        raise EvoLibraryError(f"{path} not found; run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(str(path))
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise EvoLibraryError(f"{path} does not export {name}")
        fn.argtypes = argtypes
        fn.restype = restype
    got = lib.evo_abi_version()
    if got != ABI_VERSION:
        raise EvoLibraryError(f"{path} reports ABI version {got}, this binding needs {ABI_VERSION}: the library was built "
                              f"from other sources -- rebuild it (python -m evo_amd._build)")
    _LIB = lib
    return lib


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _check(code: int, name: str):
    if code != 0:
        raise RuntimeError(f"{name} failed with hipError/arg code {code}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def pick_segment_length(B: int, T: int, H: int, target_waves: int = 4096) -> int:
    """Time-segment length C of the Hyena kernels: enough (b, head, segment) waves to fill 256 CUs,
    power of two in [64, 1024]."""
    c = 1024
    while c > 64 and B * H * ((T + c - 1) // c) < target_waves:
        c //= 2
    return c


class KernelTimer:
    """HIP-event timing of individual launches ON THE STREAM THEY ARE LAUNCHED ON (torch's current stream).
    bench.py turns it on for the timed region to get per-kernel average durations for the roofline line."""

    def __init__(self):
        self.pairs = {}

    class _Span:
        def __init__(self, timer, name):
            self.timer, self.name = timer, name

        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()

        def __exit__(self, *exc):
            self.b.record()
            self.timer.pairs.setdefault(self.name, []).append((self.a, self.b))

    def span(self, name):
        return KernelTimer._Span(self, name)

    def summary(self):
        """{name: (launches, mean_ms)} -- call after a device synchronize."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.pairs.items()}


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOSPAN = _NoSpan()


class HipOps:
    """The gfx950 op set.  All tensors must live on the same ROCm device."""

    name = "hip-gfx950"

    def __init__(self):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise EvoLibraryError("HipOps needs a ROCm GPU (torch.cuda.is_available() is False)")
        self.seg_len_override = int(os.environ.get("EVO_AMD_SEG_LEN", "0"))
        # Routing (round 4): every prefill dense layer runs on the hand-written persistent kernel of csrc/gemm.hip -- no vendor GEMM in
        # a scoring step.  The attributes below are in-process A/B knobs for bench.py's legs and the tests (no environment switches):
        self.attn_gemm_mfma = True
        # prefill attention on the 64-rows-per-wave kernel of round 5 (csrc/attn_w64.hip); False: the 8-wave kernel of rounds 2-4 (bench A/B leg)
        self.attn_w64 = True
        # round 5: the RMSNorm passes of prefill-sized batches ride in the epilogues of the dense layers around them (csrc/gemm.hip, NF;
        # False = the separate rmsnorm launches: bench.py's A/B leg, the routing test)
        self.fuse_norm = True
        self.attn_prescale = True         # the model folds softmax_scale * log2(e) into the rotary kernel's one rounding of q; attention kernels take scores as exponents (csrc/attn_w64.hip PRE)
        # round 6: T = 512 k + 1 scoring batches (a BOS token in front of 2^k nucleotides) run the 512 k main tokens of every row through hyena_ct in
        # whole tiles and the one token behind them through the fused single-token launch, from the operator's end state; False: the ragged last
        # tile (one valid step at a full tile's issue time) inside the operator, the tail tokens' projections through the weight-streaming launch
        self.hyena_tail_split = True
        # round 6: the gated MLP's first half at 5-64 rows as ONE MFMA weight-streaming launch (False: dot2 launch up to 8 rows with a norm, dense layer + gate kernel above)
        self.gate_small_m_mfma = True
        self.hyena_table_guard = True     # Hyena layers whose filter the bf16 hi / lo operand tables cannot hold run the modal kernels (hyena_tables.table_precision)
        # all_gemm_mfma = False puts the plain dense layers (l3, the unembedding of model(ids)) back on hipBLASLt through torch.addmm:
        # the library is 1-3 % faster on l3's shape (K = 11,008; profiles/r03_gemm_notes.txt) -- bench.py times that leg beside the headline
        self.all_gemm_mfma = True
        # the gated MLP's first half as ONE launch of the dense layer with GELU * gate in its epilogue; False: dense layer + gate kernel
        self.mlp_gate_fused = True
        self.timer: Optional[KernelTimer] = None
        self.validate_ids = os.environ.get("EVO_AMD_VALIDATE_IDS", "1") != "0"   # one 4-byte D2H read per forward
        # the single-pass matrix-core Hyena operator on CHANNEL-MAJOR z^T (csrc/hyena_ct.hip: the projection launched with swapped
        # operands, no input window in LDS); False: the modal three-launch kernels (the tests' yardstick, padding masks, tiny inputs)
        self.hyena_mfma = True
        self.hyena_ct_flag = True
        self._xpad = threading.local()  # per thread: the zero-initialised padded input of the swapped-operand projection (_xpad_buffer)
        self.last_hyena_io = {}

    def _t(self, name):
        return self.timer.span(name) if self.timer is not None else _NOSPAN

    # ---- plumbing -------------------------------------------------------------------------------
    @staticmethod
    def _need(t: torch.Tensor, dtype, what: str):
        if not t.is_cuda:
            raise RuntimeError(f"{what}: tensor is on {t.device}; the HIP path has no CPU fallback")
        if t.dtype != dtype:
            raise RuntimeError(f"{what}: expected {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError(f"{what}: tensor must be contiguous")

    # ---- dense layers (hipBLASLt via torch) --------------------------------------------------------
    @staticmethod
    def _tail_rows(x, w) -> int:
        """Rows to peel off a big dense layer: the BOS token makes M = B * (nt + 1) = 256 k + (1..16), and that last
        sliver costs a whole extra row of 256-row tiles (measured on hipBLASLt: +0.7 ... +4.7 % per GEMM at
        M = 65,544, +0.5 ... +3.7 % at 131,073).  Dense layers are row-independent, so the sliver goes through the
        weight-streaming kernel instead."""
        M, K = x.shape
        r = M % 256
        if M >= 4096 and 1 <= r <= 16 and K % 32 == 0 and x.is_cuda and x.dtype == torch.bfloat16 \
                and w.dtype == torch.bfloat16 and x.is_contiguous() and w.is_contiguous():
            return r
        return 0

    def linear(self, x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, mfma: bool = False) -> torch.Tensor:
        """x [M,K] @ w[N,K]^T (+ b) -> [M,N] bf16.  M <= 16 (decode) takes the weight-streaming kernels; `mfma=True`
        (the attention block's projections) takes the hand-written MFMA kernel of csrc/gemm.hip when the shape allows
        (`attn_gemm_mfma` / `all_gemm_mfma` False route to hipBLASLt: bench.py's A/B legs)."""
        if self._use_small_m(x, w):
            return self._linear_small_m(x, w, b, None)
        r = self._tail_rows(x, w)
        if r:
            M = x.shape[0]
            y = torch.empty(M, w.shape[0], dtype=torch.bfloat16, device=x.device)
            self._linear_into(y[: M - r], x[: M - r], w, b, mfma)
            self._linear_small_m(x[M - r:], w, b, None, out=y[M - r:])
            return y
        if self._take_mfma(mfma) and self.mfma_linear_ok(x, w):
            return self.linear_mfma(x, w, b)
        with self._t("gemm"):
            if b is not None:
                return torch.addmm(b, x, w.t())
            return torch.mm(x, w.t())

    def _take_mfma(self, mfma: bool) -> bool:
        return self.all_gemm_mfma or (mfma and self.attn_gemm_mfma)

    def _linear_into(self, y, x, w, b, mfma):
        if self._take_mfma(mfma) and self.mfma_linear_ok(x, w):
            with self._t("gemm_mfma"):
                _check(self.lib.evo_linear_mfma_bf16(x.data_ptr(), w.data_ptr(), _ptr(b), None, y.data_ptr(),
                                                     x.shape[0], w.shape[0], x.shape[1], _stream()), "evo_linear_mfma_bf16")
            return
        with self._t("gemm"):
            if b is not None:
                torch.addmm(b, x, w.t(), out=y)
            else:
                torch.mm(x, w.t(), out=y)

    def linear_residual_(self, res: torch.Tensor, x: torch.Tensor, w: torch.Tensor, mfma: bool = False,
                         bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """res += x @ w^T (+ bias) (fp32 accumulate, one rounding), in place.  `bias` is for the decode path, where the
        weight-streaming kernel adds it in the same pass."""
        if self._use_small_m(x, w) and res.is_contiguous():
            return self._linear_small_m(x, w, bias, res)
        r = self._tail_rows(x, w) if res.is_contiguous() else 0
        if r:
            M = x.shape[0]
            self.linear_residual_(res[: M - r], x[: M - r], w, mfma, bias=bias)
            self._linear_small_m(x[M - r:], w, bias, res[M - r:])
            return res
        if self._take_mfma(mfma) and self.mfma_linear_ok(x, w) and res.is_contiguous():
            return self.linear_mfma(x, w, bias, res)          # (bias and residual in the dense layer's epilogue: one rounding)
        with self._t("gemm"):
            res.addmm_(x, w.t())
        return res if bias is None else res.add_(bias)

    @staticmethod
    def _use_small_m(x, w):
        """The weight-streaming kernels (csrc/gemv.hip) serve every decode-sized batch: dot2 form up to M = 4 (and
        for K % 32 != 0 up to M = 8), MFMA form for 5 <= M <= 16 (tools/bench_gemv.py for the crossovers) and -- round 6 -- for
        17 <= M <= 64 with 2-4 m tiles per weight pass (the pooled decode step at 17-64 live slots ran the persistent 256 x 256 GEMM
        there: N / 256 of the 256 CUs busy, 80 % of a 32-slot step; profiles/r06_pool32_kernel_stats.txt)."""
        M, K = x.shape
        if not (1 <= M <= (64 if K % 32 == 0 else 16)):
            return False
        if K % 32 != 0 and not (M <= 4 or (M <= 8 and w.shape[0] <= 4096)):
            return False
        return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.is_contiguous()
                and w.is_contiguous() and K % 8 == 0)

    def _linear_small_m(self, x, w, b, res, out=None):
        M, K = x.shape
        N = w.shape[0]
        y = res if res is not None else (out if out is not None else torch.empty(M, N, dtype=torch.bfloat16, device=x.device))
        # 17-64 rows on a narrow layer (N < 8192): a workspace for the partial sums of the split over K (csrc/gemv.hip skinny_nw_kernel SPLITK)
        ws = torch.empty(8 * M * N, dtype=torch.float32, device=x.device) if (M > 16 and N < 8192 and K % 256 == 0 and N % 64 == 0) else None
        with self._t("gemv"):
            _check(self.lib.evo_linear_small_m_bf16(x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(res), y.data_ptr(),
                                                    M, N, K, _ptr(ws), 0 if ws is None else ws.numel() * 4, _stream()), "evo_linear_small_m_bf16")
        return y

    @staticmethod
    def mfma_linear_ok(x, w):
        return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.is_contiguous()
                and w.is_contiguous() and w.shape[0] % 256 == 0 and x.shape[1] % 64 == 0 and x.shape[0] > 8)

    def linear_mfma(self, x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None,
                    residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Hand-written MFMA dense layer (csrc/gemm.hip): x [M,K] @ w[N,K]^T (+ b) (+ residual, in place) -> [M,N]."""
        self._need(x, torch.bfloat16, "linear_mfma x")
        self._need(w, torch.bfloat16, "linear_mfma w")
        M, K = x.shape
        N = w.shape[0]
        if residual is not None:
            self._need(residual, torch.bfloat16, "linear_mfma residual")
        y = residual if residual is not None else torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
        with self._t("gemm_mfma"):
            _check(self.lib.evo_linear_mfma_bf16(x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(residual), y.data_ptr(),
                                                 M, N, K, _stream()), "evo_linear_mfma_bf16")
        return y

    # ---- kernels -------------------------------------------------------------------------------------
    def embed(self, ids: torch.Tensor, weight: torch.Tensor, validate: bool = True) -> torch.Tensor:
        """Row gather.  Ids outside [0, vocab) raise IndexError (the reference's F.embedding device-asserts): the
        kernel never indexes the table with them and raises a device flag, which is read back here -- except while a
        hipGraph is being captured (no host read is possible there; the decode loop's ids come from `sample`).  `validate=False`: the
        caller has checked the ids itself (a sequence-parallel rank checks ALL of them before its first collective)."""
        ids = ids.reshape(-1).to(torch.int64).contiguous()
        self._need(ids, torch.int64, "embed ids")
        self._need(weight, torch.bfloat16, "embed weight")
        V, D = weight.shape
        out = torch.empty(ids.numel(), D, dtype=torch.bfloat16, device=weight.device)
        # a fresh 4-byte flag per call (a cached one would be an inference tensor when first made under
        # torch.inference_mode and could not be reset outside it); none while a hipGraph is being captured
        flag = None if torch.cuda.is_current_stream_capturing() or not (self.validate_ids and validate) else \
            torch.zeros(1, dtype=torch.int32, device=weight.device)
        _check(self.lib.evo_embed_bf16(ids.data_ptr(), weight.data_ptr(), out.data_ptr(), ids.numel(), D, V,
                                       _ptr(flag), _stream()), "evo_embed_bf16")
        if flag is not None and int(flag.item()) != 0:
            raise IndexError(f"input_ids contain values outside [0, {V}) (embedding table has {V} rows)")
        return out

    def rmsnorm(self, x: torch.Tensor, bias: Optional[torch.Tensor], scale: torch.Tensor, eps: float) -> torch.Tensor:
        """out = scale * x / (rms(x) + eps); with `bias`, x is first updated in place (x += bias)."""
        self._need(x, torch.bfloat16, "rmsnorm x")
        self._need(scale, torch.bfloat16, "rmsnorm scale")
        if bias is not None:
            self._need(bias, torch.bfloat16, "rmsnorm bias")
        M, D = x.shape
        out = torch.empty_like(x)
        with self._t("rmsnorm_bias" if bias is not None else "rmsnorm"):
            _check(self.lib.evo_rmsnorm_bf16(x.data_ptr(), _ptr(bias), scale.data_ptr(), out.data_ptr(), M, D,
                                             float(eps), _stream()), "evo_rmsnorm_bf16")
        return out

    # ---- the channel-major (z^T) form of the Hyena block's input: csrc/hyena_ct.hip -------------------------------------------
    ZT_ALIGN = 64       # positions a batch row of z^T is padded to (% 8 == 0: 16-byte loads; 64 = whole cache lines -- tools/hc_bench.py A/B)

    @staticmethod
    def zt_layout(B: int, T: int):
        """(Tm, Tp, Mp, r): where batch row b / token t of a [B, T] batch sits in z^T -- a pure function of the sizes.
        Position b * Tp + t for t < Tm (the MAIN area: Mp = B Tp rounded up to the dense layer's 256-row tile positions), and for the
        r = T - Tm last tokens of a row position Mp + 8 b + (t - Tm) (the TAIL block, one more block of 256 positions behind the main area).
        Plain form: r = 0, Tm = T, Tp = T rounded up to ZT_ALIGN.  Tail form: T = 512 k + r with 1 <= r <= 8 (the bench shapes: a BOS token in
        front of 2^k nucleotides) -- rows of 512 k positions need no padding and B * 512 k is a whole number of tiles, where the padded rows
        cost the persistent dense layer one more round (+2.5 ... 4 % per projection, profiles/r04_hyena_ct_notes.txt section 3); the B r <= 16
        tail tokens go through the weight-streaming kernel, as the BOS sliver of every other dense layer does (_tail_rows)."""
        al = HipOps.ZT_ALIGN
        Tp = (T + al - 1) // al * al
        Mp = (B * Tp + 255) // 256 * 256
        r = T % 512
        Tm = T - r
        if 1 <= r <= 8 and Tm >= 512 and B * r <= 16 and Tm % al == 0 and (B * Tm + 255) // 256 * 256 < Mp:
            return Tm, Tm, (B * Tm + 255) // 256 * 256, r
        return T, Tp, Mp, 0

    @staticmethod
    def zt_geometry(B: int, T: int):
        """(Tp, Mp) of zt_layout: the row pitch and the number of positions of the main area."""
        _, Tp, Mp, _ = HipOps.zt_layout(B, T)
        return Tp, Mp

    @staticmethod
    def zt_positions(B: int, T: int, b: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """Position in z^T of (batch row b, token t), elementwise (zt_layout)."""
        Tm, Tp, Mp, r = HipOps.zt_layout(B, T)
        return torch.where(t < Tm, b * Tp + t, Mp + 8 * b + (t - Tm))

    def zt_shape_ok(self, B: int, T: int, N: int, K: int) -> bool:
        _, _, Mp, r = self.zt_layout(B, T)
        P = Mp + (256 if r else 0)
        return (self.hyena_ct_flag and B * T >= 256 and N % 256 == 0 and N % 384 == 0 and K % 64 == 0 and K >= 128
                and Mp * K * 2 < 0xffffffff and N * K * 2 < 0xffffffff and P * N * 2 < 0xfffffff0)

    def _xpad_buffer(self, B: int, T: int, D: int, device) -> torch.Tensor:
        """The cached [Mp + 16, D] bf16 workspace of rmsnorm_rows: ONE layout at a time (a scoring run keeps its shape; another (B, T)
        -- even one with the same number of rows -- replaces the buffer by a freshly zeroed one, so pad rows never hold another
        layout's activations), one instance PER THREAD (callers that drive one HipOps from several threads -- the virtual ranks of the
        sequence-parallel tests, a server's worker threads -- must not share it: launches of different threads are not ordered with each
        other) and per STREAM (a second stream would otherwise reuse it with no event ordering).  Pad rows are zero when the buffer is
        made and never written by rmsnorm_rows; what the projection computes from them lands in pad positions of z^T, which hyena_ct
        masks.  `release_workspaces()` drops it (a long-lived server between shapes)."""
        key = (B, T, D, str(device), torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)
        cur = getattr(self._xpad, "entry", None)
        if cur is None or cur[0] != key:
            _, _, Mp, _ = self.zt_layout(B, T)
            cur = (key, torch.zeros(Mp + 16, D, dtype=torch.bfloat16, device=device))
            self._xpad.entry = cur
        return cur[1]

    def release_workspaces(self) -> None:
        """Frees this thread's cached rmsnorm_rows workspace (0.5 GB at 8 x 8,193, 1 GB at 1 x 131,073 for D = 4096)."""
        self._xpad.entry = None

    def rmsnorm_rows(self, x: torch.Tensor, scale: torch.Tensor, eps: float, B: int, T: int) -> torch.Tensor:
        """RMSNorm of x [B T, D] written in the row order of zt_layout: -> [Mp + 16, D], row b T + t at its z^T position for t < Tm, the
        B r tail tokens compactly at rows Mp + b r + (t - Tm) (the weight-streaming kernel's input).  The buffer is a cached workspace
        (_xpad_buffer: pad rows are zero and never written), valid until this thread's next call on this stream."""
        self._need(x, torch.bfloat16, "rmsnorm x")
        self._need(scale, torch.bfloat16, "rmsnorm scale")
        M, D = x.shape
        assert M == B * T
        Tm, Tp, Mp, r = self.zt_layout(B, T)
        out = self._xpad_buffer(B, T, D, x.device)
        with self._t("rmsnorm"):
            _check(self.lib.evo_rmsnorm_rows_bf16(x.data_ptr(), None, scale.data_ptr(), out.data_ptr(), M, D, float(eps), T, Tp, Tm, Mp,
                                                  _stream()), "evo_rmsnorm_rows_bf16")
        return out

    def linear_t(self, xp: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], B: int, T: int, tail: bool = True) -> torch.Tensor:
        """z^T = (x @ w [N, K]^T + b)^T in blocks of 256 positions, [Mp / 256 (+ 1), N, 256] bf16, from rmsnorm_rows' buffer xp: the dense
        layer launched with swapped operands on the Mp main rows (csrc/gemm.hip, mode 3), the tail tokens (zt_layout) through the
        weight-streaming kernel into the tail block; w in the REFERENCE's row order (no regrouped copy).  `tail=False`: the main area
        only (the caller takes the tail tokens through the single-token launch: hyena_ct(main_only=True))."""
        self._need(xp, torch.bfloat16, "linear_t x")
        self._need(w, torch.bfloat16, "linear_t w")
        Tm, Tp, Mp, r = self.zt_layout(B, T)
        K = xp.shape[1]
        N = w.shape[0]
        assert xp.shape[0] >= Mp + B * r
        zt = torch.empty(Mp // 256 + (1 if r and tail else 0), N, 256, dtype=torch.bfloat16, device=xp.device)
        with self._t("gemm_zt"):
            _check(self.lib.evo_linear_t_mfma_bf16(xp.data_ptr(), w.data_ptr(), _ptr(b), zt.data_ptr(), Mp, N, K, _stream()),
                   "evo_linear_t_mfma_bf16")
        if r and tail:
            zt[-1].zero_()                                                                # (6 MB: the tail block's unused positions hold zeros, not whatever the allocator left)
            z_tail = self._linear_small_m(xp[Mp:Mp + B * r], w, b, None)                # [B r, N]
            zt[-1].view(N, 32, 8)[:, :B, :r] = z_tail.view(B, r, N).permute(2, 0, 1)      # position Mp + 8 b + j
        return zt

    @staticmethod
    def zt_rows(zt: torch.Tensor, B: int, T: int, t0: int, n: int) -> torch.Tensor:
        """Steps t0 .. t0 + n - 1 of every batch row of a channel-major z^T [blocks, 3 D, 256] as token-major [B, n, 3 D] (reference
        column order)."""
        bb = torch.arange(B, device=zt.device)[:, None].expand(B, n)
        tt = torch.arange(t0, t0 + n, device=zt.device)[None, :].expand(B, n)
        pos = HipOps.zt_positions(B, T, bb, tt).reshape(-1)
        return zt[pos // 256, :, pos % 256].view(B, n, zt.shape[1])

    @staticmethod
    def zt_from_rows(z: torch.Tensor, B: int, T: int, pad_value: float = 0.0) -> torch.Tensor:
        """The inverse for whole sequences (tests, tools): token-major z [B, T, C] -> z^T [blocks, C, 256], pad positions = pad_value."""
        Tm, Tp, Mp, r = HipOps.zt_layout(B, T)
        C = z.shape[-1]
        P = Mp + (256 if r else 0)
        flat = torch.full((C, P), pad_value, dtype=z.dtype, device=z.device)
        flat[:, :B * Tp].view(C, B, Tp)[:, :, :Tm] = z[:, :Tm].permute(2, 0, 1)
        if r:
            flat[:, Mp:].view(C, 32, 8)[:, :B, :r] = z[:, Tm:].permute(2, 0, 1)
        return flat.view(C, P // 256, 256).permute(1, 0, 2).contiguous()

    def hyena_ct(self, zt, B, T, fir_w, fir_b, table, n_heads, z_halo=None, s0=None, want_state=False, poles=None,
                 state_only=False, b_first=0, y_blk=None, y_row0=0, b_total=None, main_only=False):
        """The channel-stationary single-pass operator on CHANNEL-MAJOR z (csrc/hyena_ct.hip: evo_hyena_ct): zt [blocks, 3 D, 256] bf16 =
        linear_t's result (zt_layout of a [b_total, T] batch; b_total defaults to b_first + B); B batch rows of T tokens starting at
        batch row `b_first` of the tensor (a sub-range).  -> y [B,T,D] bf16 | (y, end state [B,D,8] complex64) with `want_state` | the end state alone with `state_only` (stage 1 of a
        sequence-parallel shard: nothing else is written).  `z_halo` [B, 2, 3 D] in the REFERENCE's column order, `s0` [B,D,8] complex:
        FIR history / modal state before the first token.  `y_blk` (from yblk_empty): the outputs go THERE, blocked, batch row b /
        token t as row y_row0 + b T + t -- the form the kernel stores fastest and linear_residual_yblk_ reads.
        `main_only` (tail form of zt_layout only): the operator walks the Tm = 512 k main tokens of every row -- whole tiles -- and leaves
        rows b T + Tm .. of y untouched; with `want_state` the state returned is the one after token Tm - 1: the caller finishes the r
        tail tokens with the single-token launch (hyena_decode_fused / hyena_step) from that state (StripedHyena._hyena_block; a ragged
        tile with one valid step costs the kernel a full tile's issue time: 8 of 136 tile steps at 8 x 8,193)."""
        self._need(zt, torch.bfloat16, "hyena z^T")
        assert zt.dim() == 3 and zt.shape[2] == 256
        D3, P = zt.shape[1], zt.shape[0] * 256
        D = D3 // 3
        Tm, Tp, Mp, r = self.zt_layout(b_first + B if b_total is None else b_total, T)
        assert D3 == 3 * D and (b_first + B) * Tp <= Mp
        if main_only:
            assert r > 0 and P in (Mp, Mp + 256) and z_halo is None and s0 is None and not state_only
        else:
            assert P == Mp + (256 if r else 0)
        T_run = Tm if main_only else T
        if table.dtype != torch.int32 or tuple(table.shape) != (D, 52, 64) or not table.is_contiguous() or not table.is_cuda:
            raise RuntimeError("hyena_ct: table must be the contiguous int32 [D, 52, 64] tensor of mfma_operand_table")
        for t, nm in ((fir_w, "fir_w"), (fir_b, "fir_b")):
            self._need(t, torch.bfloat16, "hyena " + nm)
        if z_halo is not None:
            self._need(z_halo, torch.bfloat16, "hyena z_halo")
            assert z_halo.shape == (B, 2, 3 * D)
        s0r = None
        if s0 is not None:
            s0r = torch.view_as_real(s0.to(torch.complex64).contiguous())
            assert s0r.shape == (B, D, 8, 2) and s0r.is_cuda
        s_fin = None
        if want_state or state_only:
            if poles is None:
                raise RuntimeError("hyena_ct: the end state needs the poles")
            self._need(poles, torch.float32, "hyena poles")
            assert tuple(poles.shape) == (D, 8, 2)
            s_fin = torch.empty(B, D, 8, 2, dtype=torch.float32, device=zt.device)
        yb_rows = 0
        if state_only:
            y = None
        elif y_blk is not None:
            self._need(y_blk, torch.bfloat16, "hyena y (blocked)")
            assert y_blk.dim() == 4 and tuple(y_blk.shape[1:]) == (D // 16, self.YBLK, 16)
            yb_rows = y_blk.shape[0] * self.YBLK
            assert 0 <= y_row0 and y_row0 + B * T <= yb_rows
            y = y_blk
        else:
            y = torch.empty(B, T, D, dtype=torch.bfloat16, device=zt.device)
        with self._t("hyena_mfma_state" if state_only else "hyena_mfma"):
            _check(self.lib.evo_hyena_ct(zt.data_ptr(), _ptr(z_halo), fir_w.data_ptr(), fir_b.data_ptr(), table.data_ptr(), _ptr(y),
                                         _ptr(s0r), _ptr(s_fin), _ptr(poles), B, T_run, D, n_heads, P, Tp, b_first * Tp,
                                         Tm if r and not main_only else 0, Mp + 8 * b_first, 1 if state_only else 0, yb_rows, y_row0,
                                         T if main_only else 0, _stream()),
                   "evo_hyena_ct")
        if state_only:
            return torch.view_as_complex(s_fin)
        self.last_hyena_io = {"mfma": B * T_run * 3 * D * 2 + B * T_run * D * 2}
        return (y, torch.view_as_complex(s_fin)) if want_state else y

    def hyena_prefill(self, z: torch.Tensor, fir_w: torch.Tensor, fir_b: torch.Tensor, poles: torch.Tensor,
                      residues: torch.Tensor, dskip: torch.Tensor, n_heads: int,
                      z_halo: Optional[torch.Tensor] = None, s0: Optional[torch.Tensor] = None,
                      want_state: bool = False, seg_len: Optional[int] = None,
                      mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """z [B,T,3D] bf16 -> y [B,T,D] bf16 (+ complex64 state [B,D,8] after the last token).  `mask` [B,T] (bool /
        uint8, 1 = token) is upstream's padding_mask: padded positions get a zero FIR output.  The three-launch MODAL form
        (csrc/hyena.hip: segment states, carry scan, apply) on token-major z: padding masks, inputs the single-pass kernel's
        contract excludes, and the yardstick the tests hold hyena_ct against."""
        if mask is not None:
            mask = mask.to(device=z.device, dtype=torch.uint8).contiguous()
            assert mask.shape == z.shape[:2]
        self._need(z, torch.bfloat16, "hyena z")
        B, T, D3 = z.shape
        D = D3 // 3
        for t, nm in ((fir_w, "fir_w"), (fir_b, "fir_b"), (dskip, "dskip")):
            self._need(t, torch.bfloat16, "hyena " + nm)
        self._need(poles, torch.float32, "hyena poles")
        self._need(residues, torch.float32, "hyena residues")
        if z_halo is not None:
            self._need(z_halo, torch.bfloat16, "hyena z_halo")
            assert z_halo.shape == (B, 2, D3)
        s0r = None
        if s0 is not None:
            s0r = torch.view_as_real(s0.to(torch.complex64).contiguous())
            assert s0r.shape == (B, D, 8, 2)
        C = seg_len or self.seg_len_override or pick_segment_length(B, T, n_heads)
        n_seg = (T + C - 1) // C
        agg = torch.empty(B, n_seg, D, 8, 2, dtype=torch.float32, device=z.device)
        y = torch.empty(B, T, D, dtype=torch.bfloat16, device=z.device)
        s_final = torch.empty(B, D, 8, 2, dtype=torch.float32, device=z.device) if want_state else None
        st = _stream()
        with self._t("hyena_seg_state"):
            _check(self.lib.evo_hyena_seg_state(z.data_ptr(), _ptr(z_halo), fir_w.data_ptr(), fir_b.data_ptr(),
                                                poles.data_ptr(), agg.data_ptr(), _ptr(mask), B, T, D, n_heads, C, st),
                   "evo_hyena_seg_state")
        with self._t("hyena_carry_scan"):
            _check(self.lib.evo_hyena_carry_scan(agg.data_ptr(), poles.data_ptr(), _ptr(s0r), _ptr(s_final), B, T, D,
                                                 C, st), "evo_hyena_carry_scan")
        with self._t("hyena_apply"):
            _check(self.lib.evo_hyena_apply(z.data_ptr(), _ptr(z_halo), fir_w.data_ptr(), fir_b.data_ptr(),
                                            poles.data_ptr(), residues.data_ptr(), dskip.data_ptr(), agg.data_ptr(),
                                            y.data_ptr(), _ptr(mask), B, T, D, n_heads, C, st), "evo_hyena_apply")
        state = torch.view_as_complex(s_final) if want_state else None
        # bytes of the tensors each launch touched (bench.py prints them beside the algorithmic figure)
        self.last_hyena_io = {"seg_state": z.numel() * 2 * 2 // 3 + agg.numel() * 4,
                              "apply": z.numel() * 2 + y.numel() * 2 + agg.numel() * 4}
        return y, state

    YBLK = 128          # rows per block of the blocked y layout (csrc/hyena_ct.hip)

    def yblk_empty(self, rows: int, D: int, device) -> torch.Tensor:
        """Uninitialised BLOCKED y for a [rows, D] matrix: [ceil(rows / 128), D / 16, 128, 16] bf16 (hyena_ct writes it, the output
        projection's dense layer reads it: linear_residual_yblk_)."""
        return torch.empty((rows + self.YBLK - 1) // self.YBLK, D // 16, self.YBLK, 16, dtype=torch.bfloat16, device=device)

    @staticmethod
    def yblk_to_rows(y_blk: torch.Tensor, rows: int) -> torch.Tensor:
        """Blocked y -> the row-major [rows, D] matrix it stands for (a copy; tests, and shapes the blocked dense layer does not take)."""
        nrb, G, R, c = y_blk.shape
        return y_blk.permute(0, 2, 1, 3).reshape(nrb * R, G * c)[:rows]

    def linear_residual_yblk_(self, res: torch.Tensor, y_blk: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """res [M, N] += y @ w^T (+ bias, in the same epilogue: one rounding) with y given BLOCKED (hyena_ct's y_blk: [ceil(M / 128), K / 16, 128, 16]) -- the Hyena block's output
        projection on the hand-written dense layer (csrc/gemm.hip, X operand gathered from the blocked form); the last M % 256
        rows (the BOS sliver) go row-major through the weight-streaming kernel, as everywhere."""
        M, N = res.shape
        K = w.shape[1]
        assert y_blk.shape[0] * self.YBLK >= M and y_blk.shape[1] * 16 == K and res.is_contiguous()
        Mf = M // 256 * 256
        ok = (Mf > 0 and N % 256 == 0 and K % 64 == 0 and K >= 128 and Mf * K * 2 < 0xffffffff and N * K * 2 < 0xffffffff
              and w.dtype == torch.bfloat16 and w.is_contiguous())
        if not ok:
            return self.linear_residual_(res, self.yblk_to_rows(y_blk, M).contiguous(), w, bias=bias)
        with self._t("gemm_mfma"):
            _check(self.lib.evo_linear_xblk_mfma_bf16(y_blk.data_ptr(), w.data_ptr(), _ptr(bias), res.data_ptr(), res.data_ptr(), Mf, N, K,
                                                      _stream()), "evo_linear_xblk_mfma_bf16")
        if M > Mf:                                           # (<= 255 rows: rows Mf .. M - 1 of the matrix, gathered row-major)
            r = M - Mf
            nb0 = Mf // self.YBLK
            tail = y_blk[nb0:].permute(0, 2, 1, 3).reshape(-1, K)[:r].contiguous()
            self.linear_residual_(res[Mf:], tail, w, bias=bias)
        return res

    # ---- RMSNorm folded into the dense layers around it (csrc/gemm.hip NF; include/evo_mi355x.h "RMSNorm folded ...") -----------------
    @staticmethod
    def fold_norm_scale(w: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
        """W diag(g) as a bf16 copy (fp32 product, one rounding): the weight the norm-consuming launches multiply the RAW stream by."""
        return (w.float() * g.float()[None, :]).to(torch.bfloat16).contiguous()

    def nf_shape_ok(self, M: int, N: int, K: int) -> bool:
        """Shapes whose main rows the persistent dense layer takes with the norm folded in (sliver rows <= 16 or none)."""
        r = M % 256
        Mm = M - r if 1 <= r <= 16 else M
        # (below 1,024 rows a forward is launch-bound -- four row tiles on 256 CUs -- and the fold would trade 63 small norm launches for 64 finalize launches)
        return (self.fuse_norm and self.all_gemm_mfma and M >= 1024 and N % 256 == 0 and K % 64 == 0 and K >= 128
                and Mm * K * 2 < 0xffffffff and N * K * 2 < 0xffffffff)      # (operands below 4 GiB: the kernel's 32-bit DMA offsets; outputs are addressed per tile)

    def _nf_main_rows(self, M: int) -> int:
        r = M % 256
        return M - r if 1 <= r <= 16 else M

    def rms_finalize(self, ss: Optional[torch.Tensor], x: torch.Tensor, M_main: int, eps: float) -> torch.Tensor:
        """rstd [M rounded up to 256] fp32 = 1 / (rms(x_m) + eps): rows < M_main from the dense layer's partial sums `ss`
        [strips, ld], the others from x itself (evo_rms_finalize_f32)."""
        M, D = x.shape
        rstd = torch.empty((M + 255) // 256 * 256, dtype=torch.float32, device=x.device)
        with self._t("rms_finalize"):
            _check(self.lib.evo_rms_finalize_f32(_ptr(ss), 0 if ss is None else ss.shape[0], 0 if ss is None else ss.shape[1], x.data_ptr(),
                                                 M_main, M, D, float(eps), rstd.data_ptr(), _stream()), "evo_rms_finalize_f32")
        return rstd

    def linear_residual_stats_(self, res: torch.Tensor, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], eps: float):
        """res += x @ w^T (+ bias) in place, as linear_residual_, and the RMSNorm factor of every updated row: rstd [.] fp32.  The
        main rows' sums of squares come out of the dense layer's epilogue."""
        M, N = res.shape
        K = w.shape[1]
        Mm = self._nf_main_rows(M)
        ss = torch.empty(N // 128, (Mm + 255) // 256 * 256, dtype=torch.float32, device=res.device)
        with self._t("gemm_mfma"):
            _check(self.lib.evo_linear_mfma_nf_bf16(x.data_ptr(), w.data_ptr(), _ptr(bias), res.data_ptr(), res.data_ptr(), None, ss.data_ptr(),
                                                    ss.shape[1], Mm, N, K, _stream()), "evo_linear_mfma_nf_bf16")
        if M > Mm:
            self._linear_small_m(x[Mm:], w, bias, res[Mm:])
        return self.rms_finalize(ss, res, Mm, eps)

    def linear_residual_yblk_stats_(self, res: torch.Tensor, y_blk: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], eps: float):
        """linear_residual_yblk_ + the RMSNorm factor of every updated row (see linear_residual_stats_)."""
        M, N = res.shape
        K = w.shape[1]
        Mf = M // 256 * 256
        ss = torch.empty(N // 128, Mf, dtype=torch.float32, device=res.device)
        with self._t("gemm_mfma"):
            _check(self.lib.evo_linear_xblk_mfma_nf_bf16(y_blk.data_ptr(), w.data_ptr(), _ptr(bias), res.data_ptr(), res.data_ptr(), ss.data_ptr(),
                                                         Mf, Mf, N, K, _stream()), "evo_linear_xblk_mfma_nf_bf16")
        if M > Mf:
            r = M - Mf
            tail = y_blk[Mf // self.YBLK:].permute(0, 2, 1, 3).reshape(-1, K)[:r].contiguous()
            self.linear_residual_(res[Mf:], tail, w, bias=bias)
        return self.rms_finalize(ss, res, Mf, eps)

    def linear_rs(self, x: torch.Tensor, rstd: torch.Tensor, w_folded: torch.Tensor, b: Optional[torch.Tensor], w: torch.Tensor,
                  scale: torch.Tensor, eps: float) -> torch.Tensor:
        """linear(rmsnorm(x) * scale, w, b) with the norm as a row factor in the dense layer's epilogue: x is the raw stream, rstd its
        rows' factors, w_folded = fold_norm_scale(w, scale); the sliver rows take the norm-folding weight-streaming launch on w itself."""
        M, K = x.shape
        N = w.shape[0]
        Mm = self._nf_main_rows(M)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
        with self._t("gemm_mfma"):
            _check(self.lib.evo_linear_mfma_nf_bf16(x.data_ptr(), w_folded.data_ptr(), _ptr(b), None, y.data_ptr(), rstd.data_ptr(), None, 0,
                                                    Mm, N, K, _stream()), "evo_linear_mfma_nf_bf16")
        if M > Mm:
            y[Mm:] = self.norm_linear(x[Mm:], scale, eps, w, b)
        return y

    def mlp_gate_rs(self, x: torch.Tensor, rstd: torch.Tensor, w12g_folded: torch.Tensor, w12: torch.Tensor, scale: torch.Tensor,
                    eps: float) -> torch.Tensor:
        """mlp_gate(rmsnorm(x) * scale, w12) with the norm as a row factor in the gated dense layer's epilogue (see linear_rs);
        w12g_folded = pack_gate_weights(fold_norm_scale(w12, scale))."""
        M, K = x.shape
        I = w12g_folded.shape[0] // 2
        Mm = self._nf_main_rows(M)
        a = torch.empty(M, I, dtype=torch.bfloat16, device=x.device)
        with self._t("gemm_gate"):
            _check(self.lib.evo_mlp_gate_mfma_nf_bf16(x.data_ptr(), rstd.data_ptr(), w12g_folded.data_ptr(), a.data_ptr(), Mm, I, K, _stream()),
                   "evo_mlp_gate_mfma_nf_bf16")
        if M > Mm:
            a[Mm:] = self.mlp_gate(x[Mm:], w12, scale, eps, w12g=None if w12 is not None else w12g_folded)
        return a

    def zt_stream_rows_ok(self, B: int, T: int) -> bool:
        """z^T layouts whose main area the swapped-operand projection can fill straight from the stream's rows (linear_t_rs): the tail
        form (rows of Tm = T - r positions: position b Tm + t = row b T + t), and the plain form when it has no pad position at all
        (T a multiple of 64 and B T a multiple of 256: position p = row p)."""
        Tm, Tp, Mp, r = self.zt_layout(B, T)
        if r > 0:
            return Tp == Tm and Mp == B * Tm and Tm % 256 == 0
        return Tp == T and Mp == B * T

    def linear_t_rs(self, x: torch.Tensor, rstd: torch.Tensor, w_folded: torch.Tensor, b: Optional[torch.Tensor], w: torch.Tensor,
                    scale: torch.Tensor, eps: float, B: int, T: int, tail: bool = True) -> torch.Tensor:
        """linear_t(rmsnorm_rows(x), w, b) without the normalised copy, for the layouts of zt_stream_rows_ok: the swapped-operand dense layer
        reads the raw stream x [B T, K] and scales by rstd in its epilogue; in the tail form the B r tail tokens take the norm-folding
        weight-streaming launch on w itself."""
        Tm, Tp, Mp, r = self.zt_layout(B, T)
        assert self.zt_stream_rows_ok(B, T)
        K = x.shape[1]
        N = w.shape[0]
        if r == 0:                                    # plain form without a single pad position: z^T position p IS row p of the stream
            zt = torch.empty(Mp // 256, N, 256, dtype=torch.bfloat16, device=x.device)
            with self._t("gemm_zt"):
                _check(self.lib.evo_linear_t_mfma_nf_bf16(x.data_ptr(), rstd.data_ptr(), w_folded.data_ptr(), _ptr(b), zt.data_ptr(), Mp, N, K,
                                                          Mp, Mp, 0, _stream()), "evo_linear_t_mfma_nf_bf16")
            return zt
        zt = torch.empty(Mp // 256 + (1 if tail else 0), N, 256, dtype=torch.bfloat16, device=x.device)
        with self._t("gemm_zt"):
            _check(self.lib.evo_linear_t_mfma_nf_bf16(x.data_ptr(), rstd.data_ptr(), w_folded.data_ptr(), _ptr(b), zt.data_ptr(), Mp, N, K,
                                                      B * T, Tm, r, _stream()), "evo_linear_t_mfma_nf_bf16")
        if not tail:                                  # (the caller runs the tail tokens through the single-token launch: hyena_ct(main_only=True))
            return zt
        zt[-1].zero_()
        x_tail = x.view(B, T, K)[:, Tm:].reshape(B * r, K).contiguous()                           # (a copy of B r <= 16 rows)
        z_tail = self.norm_linear(x_tail, scale, eps, w, b)                             # [B r, N]
        zt[-1].view(N, 32, 8)[:, :B, :r] = z_tail.view(B, r, N).permute(2, 0, 1)        # position Mp + 8 b + j
        return zt

    # The same operator in two stages, for sequence parallelism: stage 1 (launches 1+2) yields the shard's end
    # state from a ZERO carry-in; after the ranks exchange those, stage 2 (carry-add + launch 3) finishes.
    def hyena_stage1(self, z, fir_w, fir_b, poles, n_heads, z_halo=None, seg_len=None):
        self._need(z, torch.bfloat16, "hyena z")
        B, T, D3 = z.shape
        D = D3 // 3
        C = seg_len or self.seg_len_override or pick_segment_length(B, T, n_heads)
        n_seg = (T + C - 1) // C
        agg = torch.empty(B, n_seg, D, 8, 2, dtype=torch.float32, device=z.device)
        s_end = torch.empty(B, D, 8, 2, dtype=torch.float32, device=z.device)
        st = _stream()
        with self._t("hyena_seg_state"):
            _check(self.lib.evo_hyena_seg_state(z.data_ptr(), _ptr(z_halo), fir_w.data_ptr(), fir_b.data_ptr(),
                                                poles.data_ptr(), agg.data_ptr(), None, B, T, D, n_heads, C, st),
                   "evo_hyena_seg_state")
        with self._t("hyena_carry_scan"):
            _check(self.lib.evo_hyena_carry_scan(agg.data_ptr(), poles.data_ptr(), None, s_end.data_ptr(), B, T, D, C,
                                                 st), "evo_hyena_carry_scan")
        return (agg, C), torch.view_as_complex(s_end)

    def hyena_stage2(self, z, fir_w, fir_b, poles, residues, dskip, n_heads, stage1, z_halo=None, s0=None):
        agg, C = stage1
        B, T, D3 = z.shape
        D = D3 // 3
        st = _stream()
        if s0 is not None:
            s0r = torch.view_as_real(s0.to(torch.complex64).contiguous())
            with self._t("hyena_carry_add"):
                _check(self.lib.evo_hyena_carry_add(agg.data_ptr(), poles.data_ptr(), s0r.data_ptr(), B, T, D, C, st),
                       "evo_hyena_carry_add")
        y = torch.empty(B, T, D, dtype=torch.bfloat16, device=z.device)
        with self._t("hyena_apply"):
            _check(self.lib.evo_hyena_apply(z.data_ptr(), _ptr(z_halo), fir_w.data_ptr(), fir_b.data_ptr(),
                                            poles.data_ptr(), residues.data_ptr(), dskip.data_ptr(), agg.data_ptr(),
                                            y.data_ptr(), None, B, T, D, n_heads, C, st), "evo_hyena_apply")
        return y

    def hyena_step(self, z_t: torch.Tensor, fir_state: torch.Tensor, iir_state: torch.Tensor, fir_w, fir_b, poles,
                   residues, dskip, n_heads: int) -> torch.Tensor:
        """One decode step; fir_state [B,3D,2] bf16 and iir_state [B,D,8] complex64 are updated in place."""
        self._need(z_t, torch.bfloat16, "hyena_step z_t")
        self._need(fir_state, torch.bfloat16, "hyena_step fir_state")
        if iir_state.dtype != torch.complex64 or not iir_state.is_contiguous():
            raise RuntimeError("hyena_step: iir_state must be contiguous complex64")
        B, D3 = z_t.shape
        D = D3 // 3
        y = torch.empty(B, D, dtype=torch.bfloat16, device=z_t.device)
        sr = torch.view_as_real(iir_state)
        _check(self.lib.evo_hyena_step(z_t.data_ptr(), fir_state.data_ptr(), sr.data_ptr(), fir_w.data_ptr(),
                                       fir_b.data_ptr(), poles.data_ptr(), residues.data_ptr(), dskip.data_ptr(),
                                       y.data_ptr(), B, D, n_heads, _stream()), "evo_hyena_step")
        return y

    def hyena_decode_fused(self, x, norm_scale, eps, proj_w, proj_b, fir_state, iir_state, fir_w, fir_b, poles,
                           residues, dskip, n_heads: int) -> torch.Tensor:
        """One decode token through pre-norm + projections + FIR/modal step + gate in ONE launch (M = batch <= 4, or <= 8 at D = 4096);
        states are updated in place.  Falls back to the two kernels otherwise."""
        M, D = x.shape
        ok = ((1 <= M <= 4 or (M <= 8 and D == 4096)) and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and proj_w.is_contiguous()
              and proj_w.dtype == torch.bfloat16 and norm_scale.dtype == torch.bfloat16 and proj_b is not None
              and fir_state.dtype == torch.bfloat16 and fir_state.is_contiguous() and fir_state.shape[0] == M
              and iir_state.dtype == torch.complex64 and iir_state.is_contiguous() and iir_state.shape[0] == M
              and D == n_heads * 128)
        if not ok:
            z = self.norm_linear(x, norm_scale, eps, proj_w, proj_b)
            return self.hyena_step(z, fir_state, iir_state, fir_w, fir_b, poles, residues, dskip, n_heads)
        y = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
        sr = torch.view_as_real(iir_state)
        with self._t("gemv_hyena"):
            _check(self.lib.evo_hyena_decode_fused_small_m(
                x.data_ptr(), norm_scale.data_ptr(), proj_w.data_ptr(), proj_b.data_ptr(), fir_state.data_ptr(),
                sr.data_ptr(), fir_w.data_ptr(), fir_b.data_ptr(), poles.data_ptr(), residues.data_ptr(),
                dskip.data_ptr(), y.data_ptr(), M, D, n_heads, float(eps), _stream()), "evo_hyena_decode_fused_small_m")
        return y

    @staticmethod
    def attn_q_scale(hd: int) -> float:
        """softmax_scale * log2(e) for head dim hd: what `rope_(..., q_scale=)` folds into the queries when the attention runs `prescaled`."""
        return (1.0 / math.sqrt(hd)) * 1.4426950408889634

    def rope_(self, qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, q_scale: float = 1.0) -> torch.Tensor:
        """In-place NeoX rotary on the q and k thirds of qkv [B,T,3,H,hd].  `q_scale`: the rotated queries are multiplied by it before their
        one rounding (attn_q_scale(hd) when the attention that follows is called with prescaled=True; 1.0: plain rotary, exact)."""
        self._need(qkv, torch.bfloat16, "rope qkv")
        self._need(cos, torch.float32, "rope cos")
        self._need(sin, torch.float32, "rope sin")
        B, T, three, H, hd = qkv.shape
        assert three == 3 and cos.shape == (T, hd // 2)
        _check(self.lib.evo_rope_qk_bf16(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, T, H, hd, float(q_scale), _stream()),
               "evo_rope_qk_bf16")
        return qkv

    def rope_append_decode(self, qkv: torch.Tensor, kv: torch.Tensor, pos: torch.Tensor, inv_freq: torch.Tensor,
                           scaling: float, q_scale: float = 1.0) -> None:
        """One decode token per stream: NeoX rotary on q and k of qkv [B,1,3,H,hd] (in place) at position pos[b] (/ scaling) and
        kv[b, pos[b]] = (k, v) -- kv [>=B, cap, 2, H, hd].  One launch; bit-identical to the rotary table + rope_ + indexed copy."""
        self._need(qkv, torch.bfloat16, "rope_append_decode qkv")
        B, T, three, H, hd = qkv.shape
        if T != 1 or three != 3 or kv.dtype != torch.bfloat16 or not kv.is_cuda or kv.stride(-1) != 1 or kv.shape[0] < B:
            raise RuntimeError("rope_append_decode: expects qkv [B,1,3,H,hd] and a bf16 KV cache [>=B,cap,2,H,hd]")
        if pos.dtype != torch.int64 or not pos.is_cuda or pos.numel() != B or not pos.is_contiguous():
            raise RuntimeError("rope_append_decode pos: need a contiguous device int64 tensor with B entries")
        self._need(inv_freq, torch.float32, "rope_append_decode inv_freq")
        _check(self.lib.evo_rope_append_decode_bf16(qkv.data_ptr(), kv.data_ptr(), pos.data_ptr(), inv_freq.data_ptr(),
                                                    float(scaling), B, H, hd, kv.stride(0), kv.stride(1), kv.stride(2),
                                                    kv.stride(3), float(q_scale), _stream()), "evo_rope_append_decode_bf16")

    def attention(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_pos0: int, prescaled: bool = False) -> torch.Tensor:
        """Causal attention; q [B,Tq,H,128], k/v [B,Tk,H,128] (strided views allowed, last dim dense).  `prescaled`: q already carries
        softmax_scale * log2(e) (rope_'s q_scale = attn_q_scale(hd)): the 64-rows-per-wave kernel then runs without its per-score multiply."""
        for t, nm in ((q, "q"), (k, "k"), (v, "v")):
            if not t.is_cuda or t.dtype != torch.bfloat16 or t.stride(-1) != 1:
                raise RuntimeError(f"attention {nm}: need a ROCm bf16 tensor with a dense last dim")
        B, Tq, H, hd = q.shape
        Tk = k.shape[1]
        if hd != 128:
            raise RuntimeError("attention: head dim must be 128")
        o = torch.empty(B, Tq, H, hd, dtype=torch.bfloat16, device=q.device)
        # query ranges longer than one 128-row block: the 64-rows-per-wave kernel, which reads V^T from a workspace its pre-pass fills
        vt = torch.empty(B, H, hd, (Tk + 63) // 64 * 64, dtype=torch.bfloat16, device=q.device) if (Tq > 128 and self.attn_w64) else None
        with self._t("attn_fwd"):
            _check(self.lib.evo_attn_fwd_causal_bf16(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, Tq, Tk, int(q_pos0),
                q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                v.stride(0), v.stride(1), v.stride(2), 0.0 if prescaled else 1.0 / math.sqrt(hd), _ptr(vt), _stream()), "evo_attn_fwd_causal_bf16")
        return o

    def attention_decode(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                         pos: Optional[torch.Tensor] = None, n_splits: Optional[int] = None, prescaled: bool = False) -> torch.Tensor:
        """One query per sequence against the KV cache: q [B,1,H,128]; k/v [B,Tk,H,128] views.  With `pos`
        (device int64 [B], or [1] = same for every row) row b's query sits at pos[b] and sees keys [0,pos[b]];
        Tk is then just the capacity."""
        B, Tq, H, hd = q.shape
        if pos is not None:
            if pos.dtype != torch.int64 or not pos.is_cuda or pos.numel() not in (1, B):
                raise RuntimeError("attention_decode pos: need a device int64 tensor with 1 or B entries")
            pos = (pos.reshape(1).expand(B) if pos.numel() == 1 and B > 1 else pos.reshape(-1)).contiguous()
        if Tq != 1 or hd != 128:
            raise RuntimeError("attention_decode: expects [B,1,H,128] queries")
        for t, nm in ((q, "q"), (k, "k"), (v, "v")):
            if not t.is_cuda or t.dtype != torch.bfloat16 or t.stride(-1) != 1:
                raise RuntimeError(f"attention_decode {nm}: need a ROCm bf16 tensor with a dense last dim")
        Tk = k.shape[1]
        if n_splits is None:                     # one split per WAVE of the streaming kernel, whole workgroups of four, at most
            # one per 64-key block.  Measured on MI355X (tools/experiments/attn_decode_bench.py, round 6: the kernel keeps the next 32-key
            # half's requests in flight while it reduces the current one): 32 is best from 8 k to 131 k keys at batch 1-3 (29.5 / 78.7 /
            # 102 / 384 us at 1 x 8 k / 3 x 8 k / 32 k / 131 k; 64 splits: 31.7 / 81.3 / 107 / 385; the load-then-reduce loop of rounds
            # 3-5 wanted 64 and took 36.8 / 94.5 / 118 / 428)
            n_splits = min(((Tk + 63) // 64 + 3) // 4 * 4, 32)
        o = torch.empty(B, 1, H, hd, dtype=torch.bfloat16, device=q.device)
        part_o = torch.empty(B, H, n_splits, hd, dtype=torch.float32, device=q.device)
        part_ml = torch.empty(B, H, n_splits, 2, dtype=torch.float32, device=q.device)
        with self._t("attn_decode"):
            _check(self.lib.evo_attn_decode_bf16(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, Tk, q.stride(0), q.stride(2),
                k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2), _ptr(pos),
                part_o.data_ptr(), part_ml.data_ptr(), n_splits, 0.0 if prescaled else 1.0 / math.sqrt(hd), _stream()),
                "evo_attn_decode_bf16")
        return o

    def norm_linear(self, x: torch.Tensor, scale: torch.Tensor, eps: float, w: torch.Tensor,
                    b: Optional[torch.Tensor] = None, mfma: bool = False) -> torch.Tensor:
        """linear(rmsnorm(x) * scale, w, b).  Decode-sized batches (M <= 4) with a wide layer take ONE weight-streaming
        launch that rebuilds the normalised row on the fly; everything else is the two kernels."""
        M, K = x.shape
        if ((1 <= M <= 4 or (M <= 8 and K == 4096)) and w.shape[0] > 4096 and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
                and scale.dtype == torch.bfloat16 and x.is_contiguous() and w.is_contiguous() and scale.is_contiguous()
                and K % 8 == 0):
            y = torch.empty(M, w.shape[0], dtype=torch.bfloat16, device=x.device)
            with self._t("gemv_norm"):
                _check(self.lib.evo_norm_linear_small_m_bf16(x.data_ptr(), scale.data_ptr(), w.data_ptr(), _ptr(b),
                                                             y.data_ptr(), M, w.shape[0], K, float(eps), _stream()),
                       "evo_norm_linear_small_m_bf16")
            return y
        return self.linear(self.rmsnorm(x, None, scale, eps), w, b, mfma=mfma)

    @staticmethod
    def pack_gate_weights(w12: torch.Tensor) -> torch.Tensor:
        """[W1; W2] ([2 I, K]) -> the row order evo_mlp_gate_mfma_bf16 wants: blocks of 64 rows = rows 32 q .. 32 q + 31 of W1
        followed by the same rows of W2, so that an output tile of the dense layer holds z1 and z2 of the same gated columns."""
        I2, K = w12.shape
        I = I2 // 2
        if I % 32:
            raise ValueError(f"pack_gate_weights: inner size {I} is not a multiple of 32")
        return torch.stack([w12[:I].view(I // 32, 32, K), w12[I:].view(I // 32, 32, K)], 1).reshape(I2, K).contiguous()

    def mlp_gate_fused_ok(self, x: torch.Tensor, w12g: Optional[torch.Tensor]) -> bool:
        """The one-launch form of the gated MLP's first half (csrc/gemm.hip, gated epilogue) takes prefill-sized batches."""
        if w12g is None or not self.mlp_gate_fused:
            return False
        M, K = x.shape
        return (M >= 256 and w12g.shape[0] % 256 == 0 and K % 64 == 0 and K >= 128 and x.is_cuda and x.dtype == torch.bfloat16
                and w12g.dtype == torch.bfloat16 and x.is_contiguous() and w12g.is_contiguous()
                and M * K * 2 < 0xffffffff and w12g.shape[0] * K * 2 < 0xffffffff)

    def mlp_gate(self, x: torch.Tensor, w12: Optional[torch.Tensor], norm_scale: Optional[torch.Tensor] = None,
                 eps: float = 0.0, w12g: Optional[torch.Tensor] = None) -> torch.Tensor:
        """a [M, I] = gelu(x' @ W1^T) * (x' @ W2^T), w12 = [W1; W2] ([2I, K]); x' = x, or rmsnorm(x) * norm_scale when
        `norm_scale` is given.  Decode-sized batches (M <= 4) take ONE weight-streaming launch; prefill-sized ones the
        matrix-core dense layer with the gate in its epilogue when `w12g` (pack_gate_weights(w12)) is given -- the
        [M, 2 I] intermediate is never written; everything else is (norm,) dense layer and gate kernel.
        `w12 = None` (round 6, one weight set: StripedHyena.fold_norms_): only the regrouped copy w12g exists -- the weight-streaming
        launches read it in that row order (`grouped`), the unfused fallback un-groups the dense layer's columns."""
        M, K = x.shape
        wsrc = w12 if w12 is not None else w12g
        grouped = 1 if w12 is None else 0
        I = wsrc.shape[0] // 2
        if self.mlp_gate_fused_ok(x, w12g):
            if norm_scale is not None:
                x = self.rmsnorm(x, None, norm_scale, eps)
            a = torch.empty(M, I, dtype=torch.bfloat16, device=x.device)
            r = self._tail_rows(x, wsrc)                          # the BOS sliver (M % 256 <= 16) goes through the small-M path
            with self._t("gemm_gate"):
                _check(self.lib.evo_mlp_gate_mfma_bf16(x.data_ptr(), w12g.data_ptr(), a.data_ptr(), M - r, I, K, _stream()),
                       "evo_mlp_gate_mfma_bf16")
            if r:
                a[M - r:] = self.mlp_gate(x[M - r:], w12, w12g=w12g if w12 is None else None)
            return a
        ok_mem = (x.is_cuda and x.dtype == torch.bfloat16 and wsrc.dtype == torch.bfloat16 and x.is_contiguous() and wsrc.is_contiguous()
                  and (norm_scale is None or (norm_scale.dtype == torch.bfloat16 and norm_scale.is_contiguous())))
        # (5-8 rows WITH a norm keep the dot2 launch that norms for itself: one launch -- the BOS sliver of a scoring batch -- against norm pass + 44 us)
        if self.gate_small_m_mfma and 5 <= M <= 64 and (M > 8 or norm_scale is None) and K % 256 == 0 and I % 32 == 0 and ok_mem:
            # round 6: 5-64 rows on the MFMA weight-streaming form with the gate in its epilogue (csrc/gemv.hip skinny_nw_kernel GATE; 512-byte weight
            # requests, x shared by the workgroup: 5.4 TB/s of weights at 8 rows where the dot2 launch runs 3.6); the norm as its own small pass in front
            if norm_scale is not None:
                x = self.rmsnorm(x, None, norm_scale, eps)
            a = torch.empty(M, I, dtype=torch.bfloat16, device=x.device)
            with self._t("gemv_gate"):
                _check(self.lib.evo_mlp_gate_small_m_bf16(x.data_ptr(), wsrc.data_ptr(), a.data_ptr(), M, I, K, grouped, _stream()),
                       "evo_mlp_gate_small_m_bf16")
            return a
        if ((1 <= M <= 4 or (M <= 8 and K == 4096 and norm_scale is not None)) and x.is_cuda and x.dtype == torch.bfloat16 and wsrc.dtype == torch.bfloat16 and x.is_contiguous()
                and wsrc.is_contiguous() and K % 8 == 0 and I % 2 == 0 and (not grouped or I % 32 == 0)
                and (norm_scale is None or (norm_scale.dtype == torch.bfloat16 and norm_scale.is_contiguous()))):
            a = torch.empty(M, I, dtype=torch.bfloat16, device=x.device)
            with self._t("gemv_gate"):
                if norm_scale is None:
                    _check(self.lib.evo_mlp_gate_small_m_bf16(x.data_ptr(), wsrc.data_ptr(), a.data_ptr(), M, I, K, grouped,
                                                              _stream()), "evo_mlp_gate_small_m_bf16")
                else:
                    _check(self.lib.evo_norm_mlp_gate_small_m_bf16(x.data_ptr(), norm_scale.data_ptr(), wsrc.data_ptr(),
                                                                   a.data_ptr(), M, I, K, float(eps), grouped, _stream()),
                           "evo_norm_mlp_gate_small_m_bf16")
            return a
        if norm_scale is not None:
            x = self.rmsnorm(x, None, norm_scale, eps)
        g = self.linear(x, wsrc, None)
        if grouped:                                               # columns come out as [32 of z1 | the same 32 of z2] per block: back to [z1 | z2]
            g = g.view(M, I // 32, 2, 32).permute(0, 2, 1, 3).reshape(M, 2 * I).contiguous()
        return self.gelu_gate(g)

    def gelu_gate(self, g: torch.Tensor) -> torch.Tensor:
        self._need(g, torch.bfloat16, "gelu_gate g")
        M, I2 = g.shape
        a = torch.empty(M, I2 // 2, dtype=torch.bfloat16, device=g.device)
        with self._t("gelu_gate"):
            _check(self.lib.evo_gelu_gate_bf16(g.data_ptr(), a.data_ptr(), M, I2 // 2, _stream()),
                   "evo_gelu_gate_bf16")
        return a

    def logprob_entropy(self, logits: torch.Tensor, target: Optional[torch.Tensor], want_logprob=True,
                        want_entropy=False):
        """logits [M,V] bf16|f32, target [M] int64 -> (logprob [M] f32 | None, entropy [M] f32 | None)."""
        if logits.dtype not in (torch.bfloat16, torch.float32):
            raise RuntimeError(f"logprob logits: expected bf16 or f32, got {logits.dtype}")
        self._need(logits, logits.dtype, "logprob logits")
        M, V = logits.shape
        lp = torch.empty(M, dtype=torch.float32, device=logits.device) if want_logprob else None
        en = torch.empty(M, dtype=torch.float32, device=logits.device) if want_entropy else None
        if target is not None:
            target = target.reshape(-1).to(torch.int64).contiguous()
            if self.validate_ids and not torch.cuda.is_current_stream_capturing() and target.numel() \
                    and int(target.max().item()) >= V:
                raise IndexError(f"logprob target ids must be < {V} (negative = masked position)")
        _check(self.lib.evo_logprob_entropy(logits.data_ptr(), int(logits.dtype == torch.float32), _ptr(target),
                                            _ptr(lp), _ptr(en), M, V, _stream()),
               "evo_logprob_entropy")
        return lp, en


    def unembed_logprob_ok(self, h: torch.Tensor, emb: torch.Tensor) -> bool:
        return (h.is_cuda and h.dtype == torch.bfloat16 and emb.dtype == torch.bfloat16 and h.is_contiguous()
                and emb.is_contiguous() and emb.shape[0] == 512 and h.shape[1] % 32 == 0
                and os.environ.get("EVO_AMD_FUSED_TAIL", "1") != "0")

    def unembed_logprob(self, h: torch.Tensor, emb: torch.Tensor, target: Optional[torch.Tensor],
                        want_logprob=True, want_entropy=False):
        """Fused scoring tail: h [M,K] bf16 (final-norm output) x emb[512,K]^T -> (logprob [M] f32 | None,
        entropy [M] f32 | None) without materialising the [M, 512] logits."""
        self._need(h, torch.bfloat16, "unembed_logprob h")
        self._need(emb, torch.bfloat16, "unembed_logprob emb")
        M, K = h.shape
        V = emb.shape[0]
        lp = torch.empty(M, dtype=torch.float32, device=h.device) if want_logprob else None
        en = torch.empty(M, dtype=torch.float32, device=h.device) if want_entropy else None
        if target is not None:
            target = target.reshape(-1).to(torch.int64).contiguous()
            assert target.numel() == M
        with self._t("unembed_logprob"):
            _check(self.lib.evo_unembed_logprob_bf16(h.data_ptr(), emb.data_ptr(), _ptr(target), _ptr(lp), _ptr(en),
                                                     M, V, K, _stream()), "evo_unembed_logprob_bf16")
        return lp, en


_DEFAULT_OPS = None


def default_ops() -> HipOps:
    global _DEFAULT_OPS
    if _DEFAULT_OPS is None:
        _DEFAULT_OPS = HipOps()
    return _DEFAULT_OPS
