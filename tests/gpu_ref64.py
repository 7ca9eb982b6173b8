"""TEST INFRASTRUCTURE (never imported by the product): fp64 restatements of the oracle's operators and blocks
ON THE GPU, for the BASELINE sizes the CPU oracle (oracle/stripedhyena_ref.py) cannot finish in seconds -- 8 x 8,193 and
1 x 131,073 tokens at D = 4096.  Everything here is eager torch in float64 (rocFFT / rocBLAS fp64): no kernel of
libevo_mi355x.so is called.  Each function mirrors the oracle function it names line for line (the oracle in turn cites
the reference: /root/reference/evo/scoring.py:81 -> stripedhyena.model.StripedHyena.forward), and
tests/test_gpu_fulldepth.py::test_gpu_ref64_blocks_agree_with_the_cpu_oracle pins these restatements to the CPU oracle at
BASELINE configs[0] size before they are used as the yardstick at configs[1] / configs[2] sizes.
"""
import math

import torch


def _bf(x):
    """One bf16 rounding of an fp64 tensor (value kept in fp64)."""
    return x.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def gpu_fft_hyena(z, fir_w, fir_b, poles, residues, dskip, H, chunk=128, z_halo=None, s0=None, ref_rounding=False,
                  want_state=True, want_scale=False):
    """fp64 GPU restatement of oracle.op_hyena / RefStripedHyena.hyena_filter_parallel (FIR + split + x1*v + FFT long
    convolution + D skip + gate) on device tensors, evaluated per batch row in chunks of `chunk` channels of one head so
    that the complex128 FFT buffers stay ~1 GB.  z [B,T,3D] (any float dtype, reference column order).
    `ref_rounding=True` rounds to bf16 where the reference's eager bf16 pipeline does (oracle mode "bf16": after the FIR,
    after x1*v, after the convolution, after every op of (y + x1v*D) * x2) -- the accuracy of the reference's own
    arithmetic against fp64, i.e. the floor an engine output is judged against.
    `z_halo` [B,2,3D]: the two rows before the first; `s0` [B,D,8] complex: modal state before the first row.
    Returns y [B,T,D] float64 and the end state [B,D,8] complex128 (None unless want_state); with `want_scale` also [D]: per channel the
    largest (|conv| + |x1v * D|) * |x2| over the batch -- the size of the TERMS an output is the sum of (equal to the output's own scale
    unless the convolution cancels the skip term)."""
    B, T, D3 = z.shape
    D = D3 // 3
    hd = D // H
    n = 1 << int(math.ceil(math.log2(2 * T - 1)))               # any n >= 2T-1 gives the same linear convolution
    rnd = _bf if ref_rounding else (lambda x: x)
    y = torch.empty(B, T, D, dtype=torch.float64, device=z.device)
    st = torch.empty(B, D, 8, dtype=torch.complex128, device=z.device) if want_state else None
    nat = torch.zeros(D, dtype=torch.float64, device=z.device) if want_scale else None
    t = torch.arange(T, dtype=torch.float64, device=z.device)
    w = fir_w.double()
    for h in range(H):
        for c0 in range(0, hd, chunk):
            dsl = slice(h * hd + c0, h * hd + c0 + chunk)
            p = torch.view_as_complex(poles[dsl].double().contiguous())
            r = torch.view_as_complex(residues[dsl].double().contiguous())
            pw = torch.exp(torch.log(p)[..., None] * t)                         # [c,8,T]  p^t
            hf = torch.fft.rfft((r[..., None] * pw).real.sum(1), n=n)           # [c, n/2+1]
            for b in range(B):
                f = []
                for g in range(3):
                    col = slice(h * 3 * hd + g * hd + c0, h * 3 * hd + g * hd + c0 + chunk)
                    zz = torch.nn.functional.pad(z[b, :, col].double().t(), (2, 0))      # [c, T+2]
                    if z_halo is not None:
                        zz[:, :2] = z_halo[b, :, col].double().t()
                    wc = w[col]
                    f.append(rnd(rnd(wc[:, 0:1] * zz[:, 0:T] + wc[:, 1:2] * zz[:, 1:T + 1] + wc[:, 2:3] * zz[:, 2:T + 2])
                                 + fir_b[col].double()[:, None]))
                x2, x1, v = f
                x1v = rnd(x1 * v)
                conv = torch.fft.irfft(torch.fft.rfft(x1v, n=n) * hf, n=n)[:, :T]
                if s0 is not None:                                              # y_t += Re sum_s R p^(t+1) S0
                    s0c = s0[b, dsl].to(torch.complex128)
                    conv = conv + torch.einsum("cs,cst->ct", r * p * s0c, pw).real
                conv = rnd(conv)
                if want_scale:
                    nat[dsl] = torch.maximum(nat[dsl], ((conv.abs() + (x1v * dskip[dsl].double()[:, None]).abs()) * x2.abs()).amax(-1))
                y[b, :, dsl] = rnd(rnd(conv + rnd(x1v * dskip[dsl].double()[:, None])) * x2).t()
                if want_state:
                    e = torch.einsum("ct,cst->cs", x1v.to(torch.complex128), pw.flip(-1))
                    if s0 is not None:
                        e = e + s0c * pw[..., -1] * p                           # p^T S0
                    st[b, dsl] = e
                del f, x2, x1, v, x1v, conv
            del pw, hf
    return (y, st, nat) if want_scale else (y, st)


# ---- blocks (RefStripedHyena.hyena_block / attn_block, mode "fp64") on a SUBSET of output rows -----------------------------
def rmsnorm64(x, scale, eps):
    """oracle RefStripedHyena.rmsnorm: scale * x / (||x|| D^-1/2 + eps), eps outside the root.  x [n,D] fp64."""
    den = torch.linalg.vector_norm(x, dim=-1, keepdim=True) * (x.shape[-1] ** -0.5) + eps
    return scale.double() * (x / den)


def _linear64(x, w, b=None, chunk=16384):
    """x [n,K] fp64 @ w[N,K]^T (+ b), row-chunked so that the fp64 copy of the output stays the only large buffer."""
    wd = w.double().t().contiguous()
    out = torch.empty(x.shape[0], w.shape[0], dtype=torch.float64, device=x.device)
    for i in range(0, x.shape[0], chunk):
        out[i:i + chunk] = x[i:i + chunk] @ wd
        if b is not None:
            out[i:i + chunk] += b.double()
    return out


def _mlp_tail64(u2, sd, pre, eps):
    """oracle: mlp(rmsnorm(u2, post_norm)) + u2 with the exact-erf GELU."""
    n2 = rmsnorm64(u2, sd[pre + "post_norm.scale"], eps)
    a = torch.nn.functional.gelu(_linear64(n2, sd[pre + "mlp.l1.weight"])) * _linear64(n2, sd[pre + "mlp.l2.weight"])
    return _linear64(a, sd[pre + "mlp.l3.weight"]) + u2


def hyena_block64(u, sd, i, cfg, rows):
    """oracle RefStripedHyena.hyena_block(u, i, None) in fp64 for ONE sequence u [T,D] (the engine's tapped residual
    stream, bf16 values); returns the block output at `rows` (LongTensor) as [len(rows), D] fp64.  The projection and the
    long convolution run over the whole sequence, the row-local tail (out_filter_dense, post-norm, MLP) only on `rows`."""
    pre = f"blocks.{i}."
    T, D = u.shape
    H = cfg["num_attention_heads"]
    eps = cfg.get("eps", 1e-6)
    z = torch.empty(T, 3 * D, dtype=torch.float64, device=u.device)
    wp = sd[pre + "projections.weight"].double().t().contiguous()
    for j in range(0, T, 16384):
        z[j:j + 16384] = rmsnorm64(u[j:j + 16384].double(), sd[pre + "pre_norm.scale"], eps) @ wp
        z[j:j + 16384] += sd[pre + "projections.bias"].double()
    del wp
    fw = sd[pre + "filter.short_filter_weight"].reshape(3 * D, -1)
    y, _ = gpu_fft_hyena(z[None], fw, sd[pre + "filter.short_filter_bias"], sd[pre + "filter.poles"].reshape(D, 8, 2),
                         sd[pre + "filter.residues"].reshape(D, 8, 2), sd[pre + "filter.D"], H, want_state=False)
    del z
    yr = y[0, rows]
    del y
    u2 = _linear64(yr, sd[pre + "out_filter_dense.weight"], sd[pre + "out_filter_dense.bias"]) + u[rows].double()
    return _mlp_tail64(u2, sd, pre, eps)


def rotary_table64(cfg, T, device):
    """oracle rotary_table: fp32 angles (positions / scaling factor for the 131k yml), cos / sin rounded to bf16 as
    flash-attn caches them in the activation dtype."""
    hd = cfg["hidden_size"] // cfg["num_attention_heads"]
    inv_freq = 1.0 / (cfg.get("rotary_emb_base", 10000.0) ** (torch.arange(0, hd, 2, dtype=torch.float32, device=device) / hd))
    t = torch.arange(T, dtype=torch.float32, device=device)
    if cfg.get("use_interpolated_rotary_pos_emb", False):
        t = t / float(cfg.get("rotary_emb_scaling_factor", 1.0))
    fr = torch.outer(t, inv_freq)
    return torch.cos(fr).bfloat16().double(), torch.sin(fr).bfloat16().double()


def _rope64(x, cos, sin):
    """NeoX pairs (i, i + hd/2); x [n,H,hd] fp64, cos / sin [n, hd/2]."""
    hd = x.shape[-1]
    x0, x1 = x[..., : hd // 2], x[..., hd // 2:]
    c, s = cos[:, None, :], sin[:, None, :]
    return torch.cat([x0 * c - x1 * s, x0 * s + x1 * c], dim=-1)


def causal_attention64(q, k, v, q_pos, chunk=256):
    """softmax(q k^T / sqrt(hd)) v in fp64 for query rows at absolute positions q_pos (LongTensor [n]) against keys
    0..Tk-1; q [n,H,hd], k / v [Tk,H,hd] (any float dtype)."""
    n, H, hd = q.shape
    Tk = k.shape[0]
    out = torch.empty(n, H, hd, dtype=torch.float64, device=q.device)
    kj = torch.arange(Tk, device=q.device)[None, :]
    for h in range(H):
        kk, vv = k[:, h].double(), v[:, h].double()
        for s0 in range(0, n, chunk):
            sc = (q[s0:s0 + chunk, h].double() @ kk.t()) / math.sqrt(hd)
            sc.masked_fill_(kj > q_pos[s0:s0 + chunk, None], float("-inf"))
            out[s0:s0 + chunk, h] = torch.softmax(sc, -1) @ vv
            del sc
    return out


def attn_block64(u, sd, i, cfg, rows):
    """oracle RefStripedHyena.attn_block(u, i, None) in fp64 for one sequence u [T,D]; output at `rows` [len(rows), D]."""
    pre = f"blocks.{i}."
    T, D = u.shape
    H = cfg["num_attention_heads"]
    hd = D // H
    eps = cfg.get("eps", 1e-6)
    w = sd[pre + "inner_mha_cls.Wqkv.weight"]
    b = sd[pre + "inner_mha_cls.Wqkv.bias"]
    cos, sin = rotary_table64(cfg, T, u.device)
    # K and V of every position, Q of the wanted rows only
    kv = torch.empty(T, 2, H, hd, dtype=torch.float64, device=u.device)
    wkv = w[D:].double().t().contiguous()
    for j in range(0, T, 16384):
        n_ = rmsnorm64(u[j:j + 16384].double(), sd[pre + "pre_norm.scale"], eps)
        kv[j:j + 16384] = (n_ @ wkv + b[D:].double()).view(-1, 2, H, hd)
        kv[j:j + 16384, 0] = _rope64(kv[j:j + 16384, 0], cos[j:j + 16384], sin[j:j + 16384])
    del wkv
    q = (rmsnorm64(u[rows].double(), sd[pre + "pre_norm.scale"], eps) @ w[:D].double().t() + b[:D].double()).view(-1, H, hd)
    q = _rope64(q, cos[rows], sin[rows])
    a = causal_attention64(q, kv[:, 0], kv[:, 1], rows).reshape(-1, D)
    del kv
    u2 = _linear64(a, sd[pre + "inner_mha_cls.out_proj.weight"], sd[pre + "inner_mha_cls.out_proj.bias"]) + u[rows].double()
    return _mlp_tail64(u2, sd, pre, eps)
