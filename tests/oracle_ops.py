"""TEST INFRASTRUCTURE: an `ops` backend for evo_amd.sh.model.StripedHyena built on the CPU oracle
(oracle/stripedhyena_ref.py).  It exists so the HOST logic (block sequencing, caches, generation loop,
sequence-parallel exchange) can run in `-m "not gpu"` tests; the product never imports it."""
import math

import torch

from oracle import stripedhyena_ref as R


class OracleOps:
    name = "oracle-cpu"

    def __init__(self, act=torch.float64):
        self.act = act              # dtype tensors are returned in (fp64: exact; bf16: rounding per op)

    def _o(self, t):
        return t.to(self.act)

    def linear(self, x, w, b=None, mfma=False):
        y = x.double() @ w.double().t()
        if b is not None:
            y = y + b.double()
        return self._o(y)

    def linear_residual_(self, res, x, w, mfma=False, bias=None):
        y = res.double() + x.double() @ w.double().t()
        if bias is not None:
            y = y + bias.double()
        res.copy_(self._o(y))
        return res

    def embed(self, ids, weight, validate=True):
        return self._o(weight[ids.reshape(-1).long()])

    def rmsnorm(self, x, bias, scale, eps):
        xh, n = R.op_rmsnorm(x, scale, eps, bias)
        if bias is not None:
            x.copy_(self._o(xh))
            xh, n = R.op_rmsnorm(x, scale, eps, None)
        return self._o(n)

    def hyena_prefill(self, z, fir_w, fir_b, poles, residues, dskip, n_heads, z_halo=None, s0=None,
                      want_state=False, seg_len=None, mask=None):
        y, st = R.op_hyena(z, fir_w, fir_b, poles, residues, dskip, n_heads, z_halo, s0, mask=mask)
        return self._o(y), (st.to(torch.complex64 if self.act != torch.float64 else torch.complex128)
                            if want_state else None)

    def hyena_end_state(self, z, fir_w, fir_b, poles, n_heads, z_halo=None, seg_len=None):
        D = z.shape[-1] // 3
        _, st = R.op_hyena(z, fir_w, fir_b, poles, torch.zeros_like(poles), torch.zeros(D), n_heads, z_halo, None)
        return st

    def hyena_step(self, z_t, fir_state, iir_state, fir_w, fir_b, poles, residues, dskip, n_heads):
        y, nf, ns = R.op_hyena_step(z_t, fir_state, iir_state, fir_w, fir_b, poles, residues, dskip, n_heads)
        fir_state.copy_(nf.to(fir_state.dtype))
        iir_state.copy_(ns.to(iir_state.dtype))
        return self._o(y)

    def rope_(self, qkv, cos, sin):
        qkv.copy_(self._o(R.op_rope(qkv, cos, sin)))
        return qkv

    def attention(self, q, k, v, q_pos0):
        return self._o(R.op_attention(q, k, v, q_pos0))

    def norm_linear(self, x, scale, eps, w, b=None, mfma=False):
        return self.linear(self.rmsnorm(x, None, scale, eps), w, b)

    def hyena_decode_fused(self, x, norm_scale, eps, proj_w, proj_b, fir_state, iir_state, fir_w, fir_b, poles, residues,
                           dskip, n_heads):
        z = self.norm_linear(x, norm_scale, eps, proj_w, proj_b)
        return self.hyena_step(z, fir_state, iir_state, fir_w, fir_b, poles, residues, dskip, n_heads)

    def mlp_gate(self, x, w12, norm_scale=None, eps=0.0, w12g=None):
        if norm_scale is not None:
            x = self.rmsnorm(x, None, norm_scale, eps)
        return self.gelu_gate(self.linear(x, w12, None))

    def gelu_gate(self, g):
        return self._o(R.op_gelu_gate(g))

    def logprob_entropy(self, logits, target, want_logprob=True, want_entropy=False):
        lp, en = R.op_logprob_entropy(logits, target)
        return (lp.float() if want_logprob and lp is not None else None), (en.float() if want_entropy else None)

    # stage-wise form used by the sequence-parallel engine
    def hyena_stage1(self, z, fir_w, fir_b, poles, n_heads, z_halo=None, seg_len=None):
        return None, self.hyena_end_state(z, fir_w, fir_b, poles, n_heads, z_halo)

    def hyena_stage2(self, z, fir_w, fir_b, poles, residues, dskip, n_heads, stage1, z_halo=None, s0=None):
        y, _ = R.op_hyena(z, fir_w, fir_b, poles, residues, dskip, n_heads, z_halo, s0)
        return self._o(y)

    def attention_decode(self, q, k, v, pos=None, n_splits=None):
        if pos is None or pos.numel() == 1:
            Tk = k.shape[1] if pos is None else int(pos.item()) + 1
            return self._o(R.op_attention(q, k[:, :Tk], v[:, :Tk], Tk - 1))
        rows = []                                  # one position per row (continuous batching)
        for b in range(q.shape[0]):
            Tk = int(pos[b].item()) + 1
            rows.append(R.op_attention(q[b:b + 1], k[b:b + 1, :Tk], v[b:b + 1, :Tk], Tk - 1))
        return self._o(torch.cat(rows, 0))
