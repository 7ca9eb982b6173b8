"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED.**

A plain-PyTorch (CPU) restatement of the StripedHyena forward that evo-design/evo
reaches through `stripedhyena==0.2.2` + FlashAttention-2.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker -- never as the thing measured or shipped.  The
product (`evo_amd/`) never imports anything under `oracle/`.

Why "parity unpinned": the arithmetic lives in third-party packages that are NOT
under /root/reference and are not installable here (`stripedhyena==0.2.2`,
[REF requirements.txt:1]; `flash-attn` [REF README.md:47-50]).  The reference
repo holds no tests, golden vectors or fixtures for this path,
and `import evo` fails without `stripedhyena`.  This file therefore restates the
published algorithm of those packages, anchored on the reference's own call
sites and configs:

  * model construction / weight dtype policy ........ [REF evo/models.py:141-150]
  * forward call signature `model(ids) -> (logits, cache)` [REF evo/scoring.py:81]
  * cache object layout ............................. [REF evo/generation.py:105-155]
  * hyper-parameters ................................ [REF evo/configs/evo-1-8k-base_inference.yml:1-38]
  * rotary interpolation (131k) ..................... [REF evo/configs/evo-1-131k-base_inference.yml:39-40]

What IS pinned against the real reference: the host-side pieces that can be
imported here (tokenizer, prepare_batch, logits_to_logprobs, the generation
loop) -- see tests/golden/make_golden.py.

Three numeric modes:
  * "bf16" : bf16 weights/activations with a rounding after every torch op,
             as the eager upstream modules do (fp32 poles/residues/filter/FFT)
             -- this is the noise floor a bf16 pipeline carries;
  * "fp32" : bf16-rounded weights up-cast to fp32, all math fp32;
  * "fp64" : the same weights in fp64 -- the ground truth the HIP path is
             compared with.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- config

@dataclass
class RefConfig:
    """Derived constants of SURVEY.md A.1 [REF evo/configs/evo-1-8k-base_inference.yml]."""
    vocab_size: int = 512            # yml:1
    hidden_size: int = 4096          # yml:2
    num_layers: int = 32             # yml:7
    attn_layer_idxs: Tuple[int, ...] = (8, 16, 24)   # yml:5
    short_filter_length: int = 3     # yml:8
    num_attention_heads: int = 32    # yml:9
    eps: float = 1e-6                # yml:13
    state_size: int = 8              # yml:14
    inner_size_multiple_of: int = 16  # yml:15
    inner_mlp_size: Optional[int] = None  # yml:25
    rotary_emb_base: float = 10000.0
    rotary_emb_scaling_factor: float = 1.0   # 131k yml:40 -> 16
    use_interpolated_rotary_pos_emb: bool = False  # 131k yml:39
    max_seqlen: int = 8192

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def inner_size(self) -> int:
        if self.inner_mlp_size is not None:
            return int(self.inner_mlp_size)
        m = self.inner_size_multiple_of
        inner = int(2 * self.hidden_size * 4 / 3)
        return m * ((inner + m - 1) // m)

    @property
    def hyena_layer_idxs(self) -> Tuple[int, ...]:
        return tuple(i for i in range(self.num_layers) if i not in self.attn_layer_idxs)

    @staticmethod
    def from_dict(d: dict) -> "RefConfig":
        g = d.get
        return RefConfig(
            vocab_size=g("vocab_size", 512), hidden_size=g("hidden_size", 4096),
            num_layers=g("num_layers", 32), attn_layer_idxs=tuple(g("attn_layer_idxs", (8, 16, 24))),
            short_filter_length=g("short_filter_length", 3),
            num_attention_heads=g("num_attention_heads", 32), eps=float(g("eps", 1e-6)),
            state_size=g("state_size", 8), inner_size_multiple_of=g("inner_size_multiple_of", 16),
            inner_mlp_size=g("inner_mlp_size", None),
            rotary_emb_base=float(g("rotary_emb_base", None) or 10000.0),
            rotary_emb_scaling_factor=float(g("rotary_emb_scaling_factor", None) or 1.0),
            use_interpolated_rotary_pos_emb=bool(g("use_interpolated_rotary_pos_emb", False)),
            max_seqlen=int(g("max_seqlen", None) or 8192),
        )


# --------------------------------------------------------------------------- synthetic weights

def make_synthetic_state_dict(cfg: RefConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """SURVEY.md A.6 weight factory: real shapes (section B schema), bf16 except poles/residues.

    Poles sit inside the unit circle with long memory (|p| log-uniform in 1-|p| over
    [1e-5, 1e-1]) so the long convolution is actually exercised at T = 131k.
    """
    g = torch.Generator().manual_seed(seed)
    D, V, I, S = cfg.hidden_size, cfg.vocab_size, cfg.inner_size, cfg.state_size
    L = cfg.num_layers
    out_scale = 1.0 / math.sqrt(2.0 * L)

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    sd: Dict[str, torch.Tensor] = {}
    emb = rn(V, D, std=2.0 / math.sqrt(D))
    sd["embedding_layer.weight"] = emb
    sd["unembed.weight"] = emb                       # tied [REF evo/models.py:136-137]
    sd["norm.scale"] = 1.0 + rn(D)
    for i in range(L):
        pre = f"blocks.{i}."
        sd[pre + "pre_norm.scale"] = 1.0 + rn(D)
        sd[pre + "post_norm.scale"] = 1.0 + rn(D)
        sd[pre + "mlp.l1.weight"] = rn(I, D)
        sd[pre + "mlp.l2.weight"] = rn(I, D)
        sd[pre + "mlp.l3.weight"] = rn(D, I) * out_scale * 4.0
        if i in cfg.attn_layer_idxs:
            sd[pre + "inner_mha_cls.Wqkv.weight"] = rn(3 * D, D, std=0.04)
            sd[pre + "inner_mha_cls.Wqkv.bias"] = rn(3 * D)
            sd[pre + "inner_mha_cls.out_proj.weight"] = rn(D, D) * out_scale * 4.0
            sd[pre + "inner_mha_cls.out_proj.bias"] = rn(D)
            hd = cfg.head_dim
            sd[pre + "inner_mha_cls.rotary_emb.inv_freq"] = 1.0 / (
                cfg.rotary_emb_base ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        else:
            sd[pre + "projections.weight"] = rn(3 * D, D, std=0.04)
            sd[pre + "projections.bias"] = rn(3 * D)
            sd[pre + "filter.short_filter_weight"] = rn(3 * D, 1, cfg.short_filter_length, std=0.3)
            sd[pre + "filter.short_filter_bias"] = rn(3 * D)
            sd[pre + "filter.D"] = rn(D, std=0.5)
            u = torch.rand(D, S, generator=g)
            one_minus = 10.0 ** (-5.0 + 4.0 * u)             # log-uniform in [1e-5, 1e-1]
            mag = 1.0 - one_minus
            ang = (torch.rand(D, S, generator=g) * 2.0 - 1.0) * math.pi
            poles = torch.stack([mag * torch.cos(ang), mag * torch.sin(ang)], dim=-1)
            # residues scaled by (1-|p|)^(1/2) so the filter's energy stays O(1) per mode
            res = torch.randn(D, S, 2, generator=g) * math.sqrt(1.0 / (2 * S))
            res = res * torch.sqrt(one_minus).unsqueeze(-1) * 4.0
            sd[pre + "filter.poles"] = poles.reshape(D, S, 1, 2).float()
            sd[pre + "filter.residues"] = res.reshape(D, S, 1, 2).float()
            sd[pre + "out_filter_dense.weight"] = rn(D, D) * out_scale * 4.0
            sd[pre + "out_filter_dense.bias"] = rn(D)
    # dtype policy of to_bfloat16_except_poles_residues [REF evo/models.py:148]
    for k in list(sd.keys()):
        if k.endswith("poles") or k.endswith("residues") or k.endswith("inv_freq"):
            sd[k] = sd[k].float()
        else:
            sd[k] = sd[k].to(torch.bfloat16)
    sd["unembed.weight"] = sd["embedding_layer.weight"]
    return sd


# --------------------------------------------------------------------------- caches (upstream cache.py)

@dataclass
class RefInferenceParams:
    """Attention KV cache params [REF evo/generation.py:109-110,117-118,140-146]."""
    max_seqlen: int
    max_batch_size: int
    seqlen_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)
    lengths_per_sample: Optional[torch.Tensor] = None


@dataclass
class RefRecurrentInferenceParams:
    """Hyena recurrent cache params [REF evo/generation.py:111-114,119,143,147]."""
    fir_filter_length: int = 3
    state_dim: int = 8
    seqlen_offset: int = 0
    fir_state_dict: dict = field(default_factory=dict)
    state_dict: dict = field(default_factory=dict)
    max_batch_size: int = 1


# --------------------------------------------------------------------------- model

class RefStripedHyena:
    """Restatement of stripedhyena.model.StripedHyena.forward."""

    def __init__(self, cfg: RefConfig, state_dict: Dict[str, torch.Tensor], mode: str = "fp32",
                 rotary_table_bf16: bool = True, device=None):
        """`device`: where the eager torch ops of this restatement run (default: CPU, as everywhere in the CPU suite).  The
        GPU tests at BASELINE configs[3] / configs[4] sizes pass "cuda:0": the SAME statements then execute on torch's own
        eager kernels (rocBLAS / rocFFT) -- still test infrastructure, no kernel of libevo_mi355x.so is involved; pinned to
        the CPU execution by tests/test_gpu_parity_r4.py::test_oracle_on_the_gpu_is_the_cpu_oracle."""
        assert mode in ("bf16", "fp32", "fp64")
        self.cfg = cfg
        self.mode = mode
        self.dev = torch.device("cpu" if device is None else device)
        self.rotary_table_bf16 = rotary_table_bf16
        self.attn_chunk_elems = 1 << 24      # score-tile bound of `attention` (elements); GPU executions at long T raise it
        self.chan_chunk = None               # channels per filter / FFT evaluation in `hyena_filter_parallel` (None: all at once)
        self.act = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp64": torch.float64}[mode]
        self.hi = torch.float64 if mode == "fp64" else torch.float32   # filter / FFT / state precision
        self.w: Dict[str, torch.Tensor] = {}
        for k, v in state_dict.items():
            v = v.to(self.dev)
            if k.endswith("poles") or k.endswith("residues"):
                self.w[k] = v.to(self.hi)
            elif k.endswith("inv_freq"):
                self.w[k] = v.float()
            else:
                self.w[k] = v.to(torch.bfloat16).to(self.act)   # bf16-rounded values in every mode

    # ---- small pieces -----------------------------------------------------
    def rmsnorm(self, x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
        """upstream layers.RMSNorm: scale * x / (||x||_2 * D^-1/2 + eps); eps OUTSIDE the root."""
        D = x.shape[-1]
        n = torch.linalg.vector_norm(x.float() if self.mode == "bf16" else x, dim=-1, keepdim=True).to(x.dtype)
        den = n * (D ** -0.5) + self.cfg.eps
        return scale * (x / den)

    def linear(self, x, w, b=None):
        return F.linear(x, w, b)

    def mlp(self, x, pre):
        """upstream layers.ParallelGatedMLP: l3(gelu(l1 x) * l2 x), exact-erf GELU."""
        z1 = self.linear(x, self.w[pre + "mlp.l1.weight"])
        z2 = self.linear(x, self.w[pre + "mlp.l2.weight"])
        return self.linear(F.gelu(z1) * z2, self.w[pre + "mlp.l3.weight"])

    def poles_residues(self, pre):
        p = torch.view_as_complex(self.w[pre + "filter.poles"].reshape(-1, self.cfg.state_size, 2).contiguous())
        r = torch.view_as_complex(self.w[pre + "filter.residues"].reshape(-1, self.cfg.state_size, 2).contiguous())
        return p, r

    def compute_filter(self, pre, T: int, c0: int = 0, c1: Optional[int] = None) -> torch.Tensor:
        """upstream ParallelHyenaFilter.compute_filter: h[d,t] = Re sum_s R[d,s] * exp(t*log p[d,s])  (channels c0..c1)."""
        p, r = self.poles_residues(pre)
        p, r = p[c0:c1], r[c0:c1]
        t = torch.arange(T, dtype=self.hi, device=self.dev)
        logp = torch.log(p)
        h = (r[..., None] * torch.exp(logp[..., None] * t)).real.sum(1)      # [D,T]
        return h

    # ---- Hyena operator ----------------------------------------------------
    def fir(self, z_cl: torch.Tensor, pre) -> torch.Tensor:
        """upstream engine.parallel_fir: z [B,T,3D] -> zc [B,3D,T] (causal depthwise conv k=3 + bias)."""
        w = self.w[pre + "filter.short_filter_weight"]
        b = self.w[pre + "filter.short_filter_bias"]
        T = z_cl.shape[1]
        zt = z_cl.transpose(1, 2)
        K = w.shape[-1]
        zc = F.conv1d(zt, w, bias=None, stride=1, padding=K - 1, groups=w.shape[0])[..., :T]
        return zc + b[None, :, None]

    def column_split(self, zc: torch.Tensor):
        """channel c = h*3*hd + g*hd + j, g in {0:x2, 1:x1, 2:v}  (SURVEY.md A.3)."""
        B, C3, T = zc.shape
        H, hd = self.cfg.num_attention_heads, self.cfg.head_dim
        z4 = zc.reshape(B, H, 3 * hd, T)
        x2 = z4[:, :, :hd].reshape(B, H * hd, T)
        x1 = z4[:, :, hd:2 * hd].reshape(B, H * hd, T)
        v = z4[:, :, 2 * hd:].reshape(B, H * hd, T)
        return x2, x1, v

    def fftconv(self, x1v: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
        """upstream engine.parallel_iir FFT branch (prefill_style fft, use_flashfft False):
        irfft(fft(x, 2T)[:T+1] * rfft(h, 2T)/2T, 2T, norm='forward')[:T]  == causal conv."""
        T = x1v.shape[-1]
        n = 2 * T
        H = torch.fft.rfft(h.to(self.hi), n=n) / n
        X = torch.fft.fft(x1v.to(self.hi), n=n)[..., : T + 1]
        y = torch.fft.irfft(X * H, n=n, norm="forward")[..., :T]
        return y

    def hyena_filter_parallel(self, z_cl, pre, want_state: bool, padding_mask=None):
        zc = self.fir(z_cl, pre)
        if padding_mask is not None:              # upstream engine.parallel_fir: z_pre * padding_mask[:, None]
            zc = zc * padding_mask[:, None, :].to(zc.dtype)
        x2, x1, v = self.column_split(zc)
        x1v = x1 * v
        T = x1v.shape[-1]
        if self.chan_chunk is None:
            h = self.compute_filter(pre, T)
            y = self.fftconv(x1v, h).to(x1v.dtype)
        else:
            # the same two statements, `chan_chunk` channels at a time: the filter [D, 8, T] complex is 34 GB at T = 131,073 in
            # fp32 with three temporaries of that size around it -- long inputs on the GPU bound it this way
            y = torch.empty_like(x1v)
            for c0 in range(0, x1v.shape[1], self.chan_chunk):
                c1 = min(x1v.shape[1], c0 + self.chan_chunk)
                y[:, c0:c1] = self.fftconv(x1v[:, c0:c1], self.compute_filter(pre, T, c0, c1)).to(x1v.dtype)
        Dskip = self.w[pre + "filter.D"]
        y = (y + x1v * Dskip[None, :, None]) * x2
        state = None
        if want_state:
            state = self.prefill_state_recurrence(x1v, pre)
        return y.transpose(1, 2), state

    def prefill_state_recurrence(self, x1v, pre):
        """S_{T-1} with S_t = p*S_{t-1} + x1v_t  (== upstream prefill_via_modal_fft)."""
        p, _ = self.poles_residues(pre)
        T = x1v.shape[-1]
        t = torch.arange(T - 1, -1, -1, dtype=self.hi, device=self.dev)
        pw = torch.exp(torch.log(p)[..., None] * t)                   # [D,S,T] p^(T-1-j)
        cdt = torch.complex128 if self.hi == torch.float64 else torch.complex64
        return torch.einsum("bdt,dst->bds", x1v.to(self.hi).to(cdt), pw.to(cdt))

    def hyena_filter_step(self, z_t, pre, fir_state, iir_state):
        """upstream step_fir + step_iir; z_t [B,3D]; fir_state [B,3D,2]; iir_state [B,D,S] complex."""
        w = self.w[pre + "filter.short_filter_weight"][:, 0, :]          # [3D,3]
        b = self.w[pre + "filter.short_filter_bias"]
        zc = w[:, 2] * z_t + (w[None, :, :2] * fir_state).sum(-1) + b
        new_fir = torch.cat([fir_state[..., 1:], z_t[..., None]], dim=-1)
        x2, x1, v = self.column_split(zc[..., None])
        x2, x1, v = x2[..., 0], x1[..., 0], v[..., 0]
        x1v = x1 * v
        p, r = self.poles_residues(pre)
        new_state = p[None] * iir_state + x1v.to(self.hi)[..., None]
        yr = (r[None] * new_state).real.sum(-1)
        Dskip = self.w[pre + "filter.D"]
        y = x2 * (yr.to(x1v.dtype) + Dskip * x1v)
        return y[:, None, :], new_fir, new_state

    # ---- attention -----------------------------------------------------------
    def rotary_table(self, T0: int, T1: int):
        hd = self.cfg.head_dim
        inv_freq = 1.0 / (self.cfg.rotary_emb_base ** (torch.arange(0, hd, 2, dtype=torch.float32, device=self.dev) / hd))
        t = torch.arange(T0, T1, dtype=torch.float32, device=self.dev)
        if self.cfg.use_interpolated_rotary_pos_emb:
            t = t / self.cfg.rotary_emb_scaling_factor
        freqs = torch.outer(t, inv_freq)
        cos, sin = torch.cos(freqs), torch.sin(freqs)
        if self.rotary_table_bf16:       # flash-attn caches cos/sin in the activation dtype (bf16)
            cos, sin = cos.bfloat16().float(), sin.bfloat16().float()
        return cos, sin

    def rope(self, x: torch.Tensor, cos, sin) -> torch.Tensor:
        """NeoX / non-interleaved: pairs (i, i+hd/2).  x [B,T,H,hd]."""
        hd = x.shape[-1]
        x0, x1 = x[..., : hd // 2], x[..., hd // 2:]
        c = cos[None, :, None, :].to(self.hi)
        s = sin[None, :, None, :].to(self.hi)
        x0h, x1h = x0.to(self.hi), x1.to(self.hi)
        out = torch.cat([x0h * c - x1h * s, x0h * s + x1h * c], dim=-1)
        return out.to(x.dtype)

    def attention(self, q, k, v, q_pos0: int) -> torch.Tensor:
        """causal softmax(q k^T / sqrt(hd)) v per head; fp32 scores/softmax, P cast to the
        activation dtype before P@V (FlashAttention-2 numerics).  q [B,Tq,H,hd], k/v [B,Tk,H,hd];
        query i sits at absolute position q_pos0+i, key j at position j."""
        B, Tq, H, hd = q.shape
        Tk = k.shape[1]
        out = torch.empty_like(q)
        qi = torch.arange(Tq, device=self.dev)[:, None] + q_pos0
        kj = torch.arange(Tk, device=self.dev)[None, :]
        # query rows are taken in chunks of `attn_chunk_elems / Tk` (bounds the [chunk, Tk] score tile); a chunk only looks at the keys
        # up to its last row's position (the keys beyond are masked to exp(-inf) = 0 exactly: same softmax, half the work at long T)
        chunk = max(1, min(Tq, self.attn_chunk_elems // max(1, Tk)))
        for h in range(H):
            for b in range(B):
                kk = k[b, :, h].to(self.hi)
                vv = v[b, :, h]
                for s0 in range(0, Tq, chunk):
                    s1 = min(Tq, s0 + chunk)
                    kmax = min(Tk, q_pos0 + s1)
                    sc = (q[b, s0:s1, h].to(self.hi) @ kk[:kmax].T) / math.sqrt(hd)
                    sc = sc.masked_fill(kj[:, :kmax] > qi[s0:s1], float("-inf"))
                    pr = torch.softmax(sc, dim=-1)
                    if self.mode == "bf16":
                        o = (pr.to(torch.bfloat16).float() @ vv[:kmax].float()).to(q.dtype)
                    else:
                        o = (pr @ vv[:kmax].to(self.hi)).to(q.dtype)
                    out[b, s0:s1, h] = o
        return out

    # ---- blocks ----------------------------------------------------------------
    def hyena_block(self, u, i, cache: Optional[RefRecurrentInferenceParams], padding_mask=None):
        """upstream ParallelGatedConvBlock.forward; with a padding_mask [B,T] the projections output, the FIR output and
        (mixer output + residual) are multiplied by it [UPSTREAM-RECALLED]."""
        pre = f"blocks.{i}."
        z = self.linear(self.rmsnorm(u, self.w[pre + "pre_norm.scale"]),
                        self.w[pre + "projections.weight"], self.w[pre + "projections.bias"])
        pm = None if padding_mask is None else padding_mask[..., None].to(u.dtype)
        if pm is not None:
            z = z * pm
        if cache is not None and i in cache.fir_state_dict:
            y, nf, ns = self.hyena_filter_step(z[:, 0], pre, cache.fir_state_dict[i], cache.state_dict[i])
            cache.fir_state_dict[i] = nf
            cache.state_dict[i] = ns
        else:
            y, state = self.hyena_filter_parallel(z, pre, want_state=cache is not None, padding_mask=padding_mask)
            if cache is not None:
                zt = z.transpose(1, 2)
                K1 = self.cfg.short_filter_length - 1
                fs = zt[..., -K1:]
                if fs.shape[-1] < K1:
                    fs = F.pad(fs, (K1 - fs.shape[-1], 0))
                cache.fir_state_dict[i] = fs.clone()
                cache.state_dict[i] = state
        u2 = self.linear(y, self.w[pre + "out_filter_dense.weight"], self.w[pre + "out_filter_dense.bias"]) + u
        if pm is not None:
            u2 = u2 * pm
        return self.mlp(self.rmsnorm(u2, self.w[pre + "post_norm.scale"]), pre) + u2

    def attn_block(self, u, i, cache: Optional[RefInferenceParams], padding_mask=None):
        """upstream AttentionBlock.forward; a padding_mask multiplies u before and after the mixer [UPSTREAM-RECALLED]."""
        pre = f"blocks.{i}."
        pm = None if padding_mask is None else padding_mask[..., None].to(u.dtype)
        if pm is not None:
            u = u * pm
        B, T, D = u.shape
        H, hd = self.cfg.num_attention_heads, self.cfg.head_dim
        qkv = self.linear(self.rmsnorm(u, self.w[pre + "pre_norm.scale"]),
                          self.w[pre + "inner_mha_cls.Wqkv.weight"], self.w[pre + "inner_mha_cls.Wqkv.bias"])
        qkv = qkv.reshape(B, T, 3, H, hd)
        off = cache.seqlen_offset if cache is not None else 0
        cos, sin = self.rotary_table(off, off + T)
        q = self.rope(qkv[:, :, 0], cos, sin)
        k = self.rope(qkv[:, :, 1], cos, sin)
        v = qkv[:, :, 2]
        if cache is not None:
            if i not in cache.key_value_memory_dict:
                cache.key_value_memory_dict[i] = torch.zeros(
                    cache.max_batch_size, cache.max_seqlen, 2, H, hd, dtype=u.dtype, device=u.device)
            kv = cache.key_value_memory_dict[i]
            kv[:B, off:off + T, 0] = k
            kv[:B, off:off + T, 1] = v
            k = kv[:B, : off + T, 0]
            v = kv[:B, : off + T, 1]
        a = self.attention(q, k, v, q_pos0=off).reshape(B, T, D)
        u2 = self.linear(a, self.w[pre + "inner_mha_cls.out_proj.weight"],
                         self.w[pre + "inner_mha_cls.out_proj.bias"]) + u
        if pm is not None:
            u2 = u2 * pm
        return self.mlp(self.rmsnorm(u2, self.w[pre + "post_norm.scale"]), pre) + u2

    # ---- top level -----------------------------------------------------------------
    def initialize_inference_params(self):
        return {
            "mha": RefInferenceParams(max_seqlen=self.cfg.max_seqlen, max_batch_size=1, seqlen_offset=0),
            "hyena": RefRecurrentInferenceParams(
                fir_filter_length=self.cfg.short_filter_length, state_dim=self.cfg.state_size, seqlen_offset=0),
        }

    @torch.no_grad()
    def forward(self, ids: torch.Tensor, inference_params_dict=None, return_hidden: bool = False, padding_mask=None):
        x = self.w["embedding_layer.weight"][ids.long().to(self.dev)]
        for i in range(self.cfg.num_layers):
            if i in self.cfg.attn_layer_idxs:
                x = self.attn_block(x, i, inference_params_dict["mha"] if inference_params_dict else None, padding_mask)
            else:
                x = self.hyena_block(x, i, inference_params_dict["hyena"] if inference_params_dict else None, padding_mask)
        x = self.rmsnorm(x, self.w["norm.scale"])
        if return_hidden:
            return x
        logits = self.linear(x, self.w["unembed.weight"])
        return logits, inference_params_dict

    __call__ = forward


# --------------------------------------------------------------------------- alternative formulations
# (self-consistency checks of SURVEY.md section 4.1: FFT conv == direct causal conv == modal recurrence)

def direct_causal_conv(x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """y[d,t] = sum_{j<=t} h[d,t-j] x[d,j]; O(T^2), small cases only.  x [B,D,T], h [D,T]."""
    B, D, T = x.shape
    y = torch.zeros_like(x)
    for t in range(T):
        y[..., t] = (x[..., : t + 1] * h[:, : t + 1].flip(-1)[None]).sum(-1)
    return y


def modal_recurrence(x: torch.Tensor, p: torch.Tensor, r: torch.Tensor, s0: Optional[torch.Tensor] = None):
    """S_t = p S_{t-1} + x_t ; y_t = Re sum_s R_s S_t.  x [B,D,T] real, p,r [D,S] complex."""
    B, D, T = x.shape
    S = torch.zeros(B, D, p.shape[1], dtype=p.dtype) if s0 is None else s0.clone()
    y = torch.zeros(B, D, T, dtype=x.dtype)
    for t in range(T):
        S = p[None] * S + x[..., t, None]
        y[..., t] = (r[None] * S).real.sum(-1)
    return y, S


def sample(logits: torch.Tensor, top_k: int = 1, top_p: float = 0.0, temperature: float = 1.0,
           generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """upstream sample.sample: greedy if top_k==1, else top-k -> /temperature -> top-p -> multinomial."""
    logits = logits.float()
    if top_k == 1:
        return logits.argmax(dim=-1)
    if top_k > 0:
        top_k = min(top_k, logits.size(-1))
        kth = torch.topk(logits, top_k, dim=-1)[0][..., -1, None]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if temperature != 1.0 and temperature > 0:
        logits = logits / temperature
    if 0.0 < top_p < 1.0:
        sl, si = torch.sort(logits, descending=False)
        cp = sl.softmax(dim=-1).cumsum(dim=-1)
        rm = cp <= (1 - top_p)
        rm = rm.scatter(1, si, rm)
        logits = logits.masked_fill(rm, float("-inf"))
    return torch.multinomial(torch.softmax(logits, dim=-1), num_samples=1, generator=generator).squeeze(-1)


# --------------------------------------------------------------------------- op-level oracle
# The same arithmetic as RefStripedHyena's methods, as free functions over explicit tensors, computed in
# `hi` precision (fp64 by default) -- the per-kernel checker for the C-ABI entry points of
# include/evo_mi355x.h, and the compute backend tests inject into the host model to run it on CPU.

def op_rmsnorm(x, scale, eps, bias=None, hi=torch.float64):
    """evo_rmsnorm_bf16: returns (updated_x, normed) in `hi`."""
    xh = x.to(hi)
    if bias is not None:
        xh = xh + bias.to(hi)
    n = torch.linalg.vector_norm(xh, dim=-1, keepdim=True)
    den = n * (xh.shape[-1] ** -0.5) + eps
    return xh, scale.to(hi) * (xh / den)


def op_hyena(z, fir_w, fir_b, poles, residues, dskip, n_heads, z_halo=None, s0=None, hi=torch.float64, mask=None):
    """evo_hyena_{seg_state,carry_scan,apply}: z [B,T,3D] -> (y [B,T,D], state [B,D,S] complex) in `hi`.
    poles/residues [D,S,2]; fir_w [3D,K]; z_halo [B,K-1,3D] (rows before t=0) or None; s0 complex or None."""
    B, T, D3 = z.shape
    D = D3 // 3
    K = fir_w.shape[-1]
    cdt = torch.complex128 if hi == torch.float64 else torch.complex64
    zt = z.to(hi).transpose(1, 2)                                   # [B,3D,T]
    left = z_halo.to(hi).transpose(1, 2) if z_halo is not None else zt.new_zeros(B, D3, K - 1)
    zp = torch.cat([left, zt], dim=-1)
    w = fir_w.to(hi)
    zc = sum(w[None, :, k, None] * zp[..., k:k + T] for k in range(K)) + fir_b.to(hi)[None, :, None]
    if mask is not None:                                            # padding_mask on the FIR output
        zc = zc * (mask != 0).to(hi)[:, None, :]
    hd = D // n_heads
    z4 = zc.reshape(B, n_heads, 3 * hd, T)
    x2 = z4[:, :, :hd].reshape(B, D, T)
    x1 = z4[:, :, hd:2 * hd].reshape(B, D, T)
    v = z4[:, :, 2 * hd:].reshape(B, D, T)
    x1v = x1 * v
    p = torch.view_as_complex(poles.to(hi).contiguous())            # [D,S]
    r = torch.view_as_complex(residues.to(hi).contiguous())
    t = torch.arange(T, dtype=hi)
    logp = torch.log(p)
    pw = torch.exp(logp[..., None] * t)                             # [D,S,T] p^t
    h = (r[..., None] * pw).real.sum(1)                             # [D,T]
    n = 2 * T
    y = torch.fft.irfft(torch.fft.rfft(x1v, n=n) * torch.fft.rfft(h, n=n), n=n)[..., :T]
    state = torch.einsum("bdt,dst->bds", x1v.to(cdt), pw.flip(-1).to(cdt))          # sum_j p^(T-1-j) x_j
    if s0 is not None:
        s0c = s0.to(cdt)
        carry = (r[None, :, :, None] * (pw * p[..., None])[None] * s0c[..., None]).real.sum(2)   # Re sum R p^(t+1) S0
        y = y + carry
        state = state + (pw[..., -1] * p)[None] * s0c
    out = (y + x1v * dskip.to(hi)[None, :, None]) * x2
    return out.transpose(1, 2).contiguous(), state


def op_hyena_step(z_t, fir_state, iir_state, fir_w, fir_b, poles, residues, dskip, n_heads, hi=torch.float64):
    """evo_hyena_step: returns (y [B,D], new_fir_state [B,3D,K-1], new_iir_state [B,D,S]) in `hi`."""
    B, D3 = z_t.shape
    D = D3 // 3
    cdt = torch.complex128 if hi == torch.float64 else torch.complex64
    w = fir_w.to(hi)
    zt = z_t.to(hi)
    fs = fir_state.to(hi)
    zc = w[:, -1] * zt + (w[None, :, :-1] * fs).sum(-1) + fir_b.to(hi)
    new_fs = torch.cat([fs[..., 1:], zt[..., None]], dim=-1)
    hd = D // n_heads
    z4 = zc.reshape(B, n_heads, 3 * hd)
    x2 = z4[:, :, :hd].reshape(B, D)
    x1 = z4[:, :, hd:2 * hd].reshape(B, D)
    v = z4[:, :, 2 * hd:].reshape(B, D)
    x1v = x1 * v
    p = torch.view_as_complex(poles.to(hi).contiguous())
    r = torch.view_as_complex(residues.to(hi).contiguous())
    ns = p[None] * iir_state.to(cdt) + x1v.to(cdt)[..., None]
    y = x2 * ((r[None] * ns).real.sum(-1) + dskip.to(hi) * x1v)
    return y, new_fs, ns


def op_rope(qkv, cos, sin, hi=torch.float64):
    """evo_rope_qk_bf16: qkv [B,T,3,H,hd] -> rotated copy in `hi` (v third untouched)."""
    out = qkv.to(hi).clone()
    hd = qkv.shape[-1]
    c = cos.to(hi)[None, :, None, :]
    s = sin.to(hi)[None, :, None, :]
    for w in (0, 1):
        x0 = out[:, :, w, :, : hd // 2].clone()
        x1 = out[:, :, w, :, hd // 2:].clone()
        out[:, :, w, :, : hd // 2] = x0 * c - x1 * s
        out[:, :, w, :, hd // 2:] = x0 * s + x1 * c
    return out


def op_attention(q, k, v, q_pos0, hi=torch.float64):
    """evo_attn_fwd_causal_bf16: q [B,Tq,H,hd], k/v [B,Tk,H,hd] -> o [B,Tq,H,hd] in `hi`."""
    B, Tq, H, hd = q.shape
    Tk = k.shape[1]
    qh, kh, vh = q.to(hi), k.to(hi), v.to(hi)
    mask = torch.arange(Tk)[None, :] > (torch.arange(Tq)[:, None] + q_pos0)
    out = torch.empty(B, Tq, H, hd, dtype=hi)
    chunk = max(1, min(Tq, (1 << 23) // max(1, Tk)))
    for b in range(B):
        for h in range(H):
            for s0 in range(0, Tq, chunk):
                s1 = min(Tq, s0 + chunk)
                sc = (qh[b, s0:s1, h] @ kh[b, :, h].T) / math.sqrt(hd)
                sc = sc.masked_fill(mask[s0:s1], float("-inf"))
                out[b, s0:s1, h] = torch.softmax(sc, dim=-1) @ vh[b, :, h]
    return out


def op_gelu_gate(g, hi=torch.float64):
    I = g.shape[-1] // 2
    gh = g.to(hi)
    return F.gelu(gh[..., :I]) * gh[..., I:]


def op_logprob_entropy(logits, target, hi=torch.float64):
    lsm = torch.log_softmax(logits.to(hi), dim=-1)
    ent = -(lsm.exp() * lsm).sum(-1)
    lp = None
    if target is not None:
        lp = lsm.gather(-1, target.clamp_min(0).long().unsqueeze(-1)).squeeze(-1)
        lp = torch.where(target >= 0, lp, torch.zeros_like(lp))
    return lp, ent
