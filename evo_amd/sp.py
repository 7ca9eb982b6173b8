"""Sequence-parallel scoring: the token dimension of every sequence is cut into R contiguous shards, one per
GPU (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).  The reference has no
multi-GPU path at all, so this is new design, not a translation.

Rank r owns tokens [r*Tl, min(T, (r+1)*Tl)), Tl = ceil(T/R), of ALL batch rows.  Embedding, RMSNorm, every
GEMM, the MLP, the unembedding and the scoring tail are token-local -> no communication.  Per layer:

  Hyena block   (1) 2-row halo of z from the previous rank (for the k=3 FIR): ONE neighbour send/recv of [B,2,3D].
                    The two tail rows are projected first by the weight-streaming kernel (M = 2B <= 16) and sent
                    while the shard's big projection GEMM runs, so the exchange is off the critical path.
                (2) stage 1 on the shard (zero carry-in) -> end state E_r [B,D,8] c64
                (3) all-gather of the R end states (262 KB * B per rank: latency-, not bandwidth-bound)
                (4) S_in(r) = sum_{q<r} p^{Tl*(r-1-q)} E_q   (exact: the filter is a finite sum of modes)
                (5) stage 2: carry-add + apply kernel
                Batch rows are cut into two groups: group A's all-gather (3) runs on RCCL's stream under group B's
                stage-1 kernels, group B's under group A's stage 2.
  Attention     head <-> sequence all-to-all ("Ulysses") when n_heads % R == 0: every rank sends, per batch row,
                the q,k,v of its token shard for head group g to rank g and receives the FULL sequence for its
                own H/R heads; it runs ordinary causal attention on them (perfectly balanced: no causal
                last-rank penalty) and a second all-to-all returns the outputs to the token owners.  Rows are
                pipelined: all forward exchanges are issued asynchronously up front, row b's attention
                overlaps the transfers of rows b+1.., and its return exchange overlaps row b+1's attention.
                Fallback (n_heads % R != 0): all-gather of K,V in token-major layout, queries attend keys <=
                their position.

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so volume per rank matters: per attention layer and
batch row the all-to-all moves 7/8 * (3+1) * Tl * D * 2 B = 0.47 GB (T = 131k, R = 8) against 1.88 GB received by
a K/V all-gather; a Hyena layer moves < 0.4 MB per row.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .sh.model import StripedHyena, _AttentionBlock


class _Done:
    def wait(self):
        return True


class DistComm:
    """all-gather over torch.distributed (RCCL when the backend is "nccl"; gloo in the CPU tests)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)

    def all_gather(self, t: torch.Tensor, async_op: bool = False):
        """Equal-shaped tensors from every rank, stacked on a new leading dim -> ([R, *shape], work)."""
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if dist.get_backend(self.group) == "nccl":
            w = dist.all_gather_into_tensor(out.view(self.world * t.shape[0], *t.shape[1:]), t, group=self.group,
                                            async_op=async_op)
        else:
            w = dist.all_gather(list(out.unbind(0)), t, group=self.group, async_op=async_op)
        return out, (w if async_op else _Done())

    def all_to_all(self, t: torch.Tensor, async_op: bool = False, out: Optional[torch.Tensor] = None):
        """t [R, ...]: slice g goes to rank g; returns ([R, ...] with slice q received from rank q, work)."""
        t = t.contiguous()
        out = torch.empty_like(t) if out is None else out
        w = dist.all_to_all_single(out, t, group=self.group, async_op=async_op)
        return out, (w if async_op else _Done())

    def shift_from_prev(self, t: torch.Tensor, async_op: bool = False):
        """Point-to-point ring step without wrap-around: this rank's `t` goes to rank+1, the return value is rank-1's
        `t` (None on rank 0).  One send + one recv per rank instead of an R-way gather."""
        rank = dist.get_rank(self.group)
        t = t.contiguous()
        ops, out = [], None
        if rank + 1 < self.world:
            ops.append(dist.P2POp(dist.isend, t, dist.get_global_rank(self.group, rank + 1) if self.group else rank + 1,
                                  self.group))
        if rank > 0:
            out = torch.empty_like(t)
            ops.append(dist.P2POp(dist.irecv, out, dist.get_global_rank(self.group, rank - 1) if self.group else rank - 1,
                                  self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        if not async_op:
            for w in works:
                w.wait()
            return out, _Done()
        return out, _Works(works, keep=t)


class HostStagedComm:
    """The DistComm interface over a CPU process group (gloo): device tensor -> host copy -> the collective on the host ->
    device copy on `wait()`.  For ranks that cannot reach each other through RCCL -- two processes that share ONE GPU (the
    two-rank test of tests/test_gpu_sp_two_procs.py: real async work handles, real HIP kernels, one device) or a box without
    peer access.  The exchange is asynchronous the way RCCL's is: `all_gather(..., async_op=True)` returns once the host
    collective is posted; the results reach the device, in stream order, when the handle is waited for."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)

    class _Handle:
        def __init__(self, works, finish, keep=None):
            self.works, self.finish, self.keep = works, finish, keep

        def wait(self):
            for w in self.works:
                w.wait()
            if self.finish is not None:
                self.finish()                                # host -> device, ordered on the current stream
                self.finish = None
            return True

    @staticmethod
    def _to_host(t):
        return t.detach().contiguous().to("cpu")             # waits for the kernels that produced t (current stream)

    def all_gather(self, t, async_op=False):
        host = self._to_host(t)
        parts = [torch.empty_like(host) for _ in range(self.world)]
        w = dist.all_gather(parts, host, group=self.group, async_op=True)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        h = HostStagedComm._Handle([w], lambda: out.copy_(torch.stack(parts, 0), non_blocking=False), keep=host)
        if not async_op:
            h.wait()
            return out, _Done()
        return out, h

    def all_to_all(self, t, async_op=False, out=None):
        host = self._to_host(t)
        recv = torch.empty_like(host)
        w = dist.all_to_all(list(recv.unbind(0)), list(host.unbind(0)), group=self.group, async_op=True) \
            if dist.get_backend(self.group) != "gloo" else None
        if w is None:                                        # gloo has no all_to_all: R isend / irecv pairs
            rank = dist.get_rank(self.group)
            ops = []
            for q in range(self.world):
                if q == rank:
                    recv[q].copy_(host[q])
                    continue
                ops.append(dist.P2POp(dist.isend, host[q], q, self.group))
                ops.append(dist.P2POp(dist.irecv, recv[q], q, self.group))
            works = dist.batch_isend_irecv(ops) if ops else []
        else:
            works = [w]
        out = torch.empty_like(t) if out is None else out
        h = HostStagedComm._Handle(works, lambda: out.copy_(recv, non_blocking=False), keep=host)
        if not async_op:
            h.wait()
            return out, _Done()
        return out, h

    def shift_from_prev(self, t, async_op=False):
        rank = dist.get_rank(self.group)
        host = self._to_host(t)
        ops, recv = [], None
        if rank + 1 < self.world:
            ops.append(dist.P2POp(dist.isend, host, rank + 1, self.group))
        if rank > 0:
            recv = torch.empty_like(host)
            ops.append(dist.P2POp(dist.irecv, recv, rank - 1, self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        out = torch.empty_like(t) if rank > 0 else None
        h = HostStagedComm._Handle(works, (lambda: out.copy_(recv, non_blocking=False)) if rank > 0 else None, keep=host)
        if not async_op:
            h.wait()
            return out, _Done()
        return out, h


class StubComm:
    """NOT a data path: a communicator whose exchanges return resident buffers of the right shape at once (what this rank sent
    stands in for what it would receive).  `SequenceParallelScorer(model, rank, world, comm=StubComm(world))` then executes exactly
    the kernels one rank of a `world`-rank job executes -- same shard length, same row groups, the carried-state arithmetic, Ulysses
    attention over the full sequence for H / world heads -- on ONE GPU: the per-rank compute time behind bench.py's
    `scaling_131k_predicted` (tools/sp_predict.py).  The values it produces mean nothing."""

    def __init__(self, world: int):
        self.world = world
        self._g = {}
        self.bytes = {"all_gather": 0, "all_to_all": 0, "shift": 0}      # what a real communicator would have received, per call site

    def all_gather(self, t, async_op=False):
        key = (tuple(t.shape), t.dtype)
        out = self._g.get(key)
        if out is None or out.device != t.device:
            out = self._g[key] = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        out.copy_(t.unsqueeze(0).expand_as(out))
        self.bytes["all_gather"] += (self.world - 1) * t.numel() * t.element_size()
        return out, _Done()

    def all_to_all(self, t, async_op=False, out=None):
        self.bytes["all_to_all"] += (self.world - 1) * (t.numel() // self.world) * t.element_size()
        if out is not None:
            out.copy_(t)
            return out, _Done()
        return t.contiguous(), _Done()

    def shift_from_prev(self, t, async_op=False):
        self.bytes["shift"] += t.numel() * t.element_size()
        return t.contiguous(), _Done()


class _Works:
    def __init__(self, works, keep=None):
        self.works, self.keep = works, keep            # `keep`: the send buffer must outlive the transfer

    def wait(self):
        for w in self.works:
            w.wait()
        return True


class SequenceParallelScorer:
    def __init__(self, model: StripedHyena, rank: int, world: int, group=None, comm=None):
        self.m = model
        self.rank, self.world = rank, world
        self.comm = comm if comm is not None else DistComm(group)
        self.attn_mode = "auto"          # "auto": Ulysses when n_heads % world == 0, else K/V all-gather
        self.row_groups = 2              # Hyena end-state exchange is pipelined over this many groups of batch rows
        self.attn_row_groups = 2         # Ulysses: rows per all-to-all / attention launch = B / attn_row_groups
        self._pow_cache = {}
        self._bufs = {}
        # comm_profile: every exchange is waited for right where it is posted and bracketed by device events, so the
        # numbers are raw collective durations (bench.py's instrumented pass); off in the timed region
        self.comm_profile = False
        self.comm_events = {}

    # ------------------------------------------------------------------ communication helpers
    def _post(self, name, fn, *args, **kw):
        """Post an exchange asynchronously; in comm_profile mode time it (post -> complete) on the device."""
        if self.comm_profile and torch.cuda.is_available():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out, w = fn(*args, **kw)
            w.wait()
            b.record()
            self.comm_events.setdefault(name, []).append((a, b))
            return out, _Done()
        return fn(*args, **kw)

    def comm_summary(self):
        """{name: (count, mean_ms)} of the exchanges timed in comm_profile mode (call after a synchronize)."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.comm_events.items()}

    def _gather0(self, t: torch.Tensor, async_op: bool = False, name: str = "all_gather"):
        return self._post(name, self.comm.all_gather, t, async_op=async_op)

    def _shift(self, t: torch.Tensor):
        """rank-1's `t` (None on rank 0), asynchronously."""
        if hasattr(self.comm, "shift_from_prev"):
            return self._post("halo_sendrecv", self.comm.shift_from_prev, t, async_op=True)
        g, w = self._post("halo_allgather", self.comm.all_gather, t, async_op=True)   # communicators without P2P

        class _Pick:
            def __init__(s, g, w, r):
                s.g, s.w, s.r = g, w, r

            def wait(s):
                return s.w.wait()
        return (g[self.rank - 1] if self.rank > 0 else None), _Pick(g, w, self.rank)

    def _buf(self, key, shape, like):
        b = self._bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != like.dtype or b.device != like.device:
            b = self._bufs[key] = torch.empty(shape, dtype=like.dtype, device=like.device)
        return b

    # ------------------------------------------------------------------ shard geometry
    def shard(self, T: int):
        Tl = (T + self.world - 1) // self.world
        t0 = min(T, self.rank * Tl)
        t1 = min(T, t0 + Tl)
        return Tl, t0, t1

    def check_geometry(self, T: int):
        """Identical verdict on EVERY rank (before any collective): all shards non-empty, and every shard that feeds a
        successor holds at least the 2 rows of the FIR halo."""
        Tl = (T + self.world - 1) // self.world
        if (self.world - 1) * Tl >= T or (self.world > 1 and Tl < 2):
            raise ValueError(f"cannot shard {T} tokens over {self.world} ranks: ceil(T/R) = {Tl} leaves an empty or "
                             f"sub-halo shard; use fewer ranks")
        return Tl

    def _pole_powers(self, poles: torch.Tensor, Tl: int) -> torch.Tensor:
        """[R, D, 8] complex128: p^(Tl*k) for k = 0..R-1 by integer exponentiation in fp64 (exact for p = 0 too;
        the carry must stay exact over the whole context)."""
        key = (poles.data_ptr(), Tl)
        hit = self._pow_cache.get(key)
        if hit is None:
            p = torch.view_as_complex(poles.double().contiguous())
            base, n = p.clone(), int(Tl)
            step = torch.ones_like(p)                        # p^Tl by binary exponentiation
            while n > 0:
                if n & 1:
                    step = step * base
                base = base * base
                n >>= 1
            rows = [torch.ones_like(p)]
            for _ in range(1, self.world):
                rows.append(rows[-1] * step)
            hit = torch.stack(rows, 0)
            if len(self._pow_cache) > 64:
                self._pow_cache.clear()
            self._pow_cache[key] = hit
        return hit

    # ------------------------------------------------------------------ blocks
    def _hyena_block(self, blk, x2d, B, Tloc, Tl, T, rs=None):
        """`rs`: 1 / (rms + eps) of every row of x2d from the epilogue of the dense layer that wrote it (round 6: the shards fold their
        RMSNorm passes into the dense layers exactly as the single-GPU forward does, sh/model.py `_nf_ok`), or None: this block norms for
        itself.  Returns the next block's `rs` (or None)."""
        m, ops = self.m, self.m.ops
        D, H = m.hidden_size, m.num_heads
        f = blk.filter
        # The single-pass matrix-core operator (csrc/hyena_ct.hip, the kernel of the scoring path) serves both stages when the shards fit
        # its launch contract: stage 1 = its state-only walk (end state from a zero carry), stage 2 = the full pass seeded with the
        # carried state, on CHANNEL-MAJOR z^T written by the projection's dense layer (the weight as it is: no regrouped copy per rank;
        # the halo rows travel in the REFERENCE's column order).  Other backends / shapes: the modal three-launch form (seg_state +
        # carry_scan, then carry_add + apply) on token-major z.
        # The choice must be the SAME on every rank -- it is a function of (B, Tl, t_min = the last shard's length, world) only; the
        # launches run per row group (nb_max / nb_min rows), so the contract is evaluated on BOTH extremes.
        t_min = T - (self.world - 1) * Tl
        G_ = max(1, min(self.row_groups, B))
        nb_max, nb_min = (B + G_ - 1) // G_, max(1, B // G_)
        ztm = getattr(ops, "hyena_mfma", False) and getattr(ops, "hyena_ct_flag", False) and hasattr(ops, "hyena_ct") \
            and m._mfma_hyena_ok(nb_max, Tl) and m._mfma_hyena_ok(nb_min, t_min) and m._mfma_hyena_ok(nb_max, t_min) \
            and m._mfma_hyena_ok(nb_min, Tl) and ops.zt_shape_ok(B, Tl, 3 * D, D) and ops.zt_shape_ok(B, t_min, 3 * D, D) and t_min >= 2 \
            and m._table_ok(blk)                              # (the filter fits the operand tables: a property of the weights, the same on every rank)
        w_p, b_p = blk.projections.weight.data, (None if blk.projections.bias is None else blk.projections.bias.data)
        table = m._mfma_table(blk) if ztm else None
        # norms folded into the dense layers (a rank-local choice: no collective depends on it -- the last, shorter shard may differ)
        nf = ztm and m._nf_ok(B * Tloc, None)
        stream_rows = nf and rs is not None and ops.zt_stream_rows_ok(B, Tloc)   # the projection reads the residual stream itself
        xp = n1 = None
        if stream_rows:
            pass
        elif ztm:
            xp = ops.rmsnorm_rows(x2d, blk.pre_norm.scale, m.eps, B, Tloc)      # rows in z^T's position order
        else:
            n1 = ops.rmsnorm(x2d, None, blk.pre_norm.scale, m.eps)
        # (1) halo: the shard's last two rows go to the next rank.  They are projected on their own first (2B rows:
        #     the weight-streaming kernel) so that the send/recv flies under the big projection GEMM.
        halo, halo_w, tail = None, _Done(), None
        if self.world > 1:
            if Tloc >= 2:
                if ztm:
                    # (norm + projection of the 2 B halo rows: ONE weight-streaming launch up to 8 rows at D = 4096, else norm + launch)
                    xh = x2d.view(B, Tloc, D)[:, -2:, :].reshape(B * 2, D).contiguous()
                    parts = [ops.norm_linear(xh[i:i + 8], blk.pre_norm.scale, m.eps, w_p, b_p) for i in range(0, 2 * B, 8)]   # (8 rows per fused launch)
                    tail = (parts[0] if len(parts) == 1 else torch.cat(parts, 0)).view(B, 2, 3 * D)
                else:
                    tail = ops.linear(n1.view(B, Tloc, D)[:, -2:, :].reshape(B * 2, D), w_p, b_p).view(B, 2, 3 * D)
                halo, halo_w = self._shift(tail)
            else:                                            # (last rank only, see check_geometry: nobody reads it)
                halo, halo_w = self._shift(x2d.new_zeros(B, 2, 3 * D))
        if stream_rows:
            wp_f = m._folded(blk, "_wp_f", blk.projections.weight, blk.pre_norm.scale)
            z = ops.linear_t_rs(x2d, rs, wp_f, b_p, w_p, blk.pre_norm.scale, m.eps, B, Tloc)
        elif ztm:
            z = ops.linear_t(xp, w_p, b_p, B, Tloc)          # z^T [blocks, 3 D, 256]
        else:
            z = ops.linear(n1, w_p, b_p).view(B, Tloc, 3 * D)
        if tail is not None:
            # the two rows the next rank convolves with came out of the weight-streaming kernel, this rank's own copy of them
            # out of the tile GEMM (another summation order: up to one bf16 ulp apart) -- use the SENT values here too, so
            # that both sides of a shard boundary see the same z
            if ztm:
                bb = torch.arange(B, device=z.device)[:, None].expand(B, 2)
                tt = torch.arange(Tloc - 2, Tloc, device=z.device)[None, :].expand(B, 2)
                pos = ops.zt_positions(B, Tloc, bb, tt).reshape(-1)
                z[pos // 256, :, pos % 256] = tail.reshape(B * 2, 3 * D)
            else:
                z[:, -2:, :] = tail
        halo_w.wait()
        if halo is not None:
            halo = halo.contiguous()
        # (2)-(5), pipelined over groups of batch rows
        G = max(1, min(self.row_groups, B))
        bounds = [(g * B) // G for g in range(G + 1)]
        st1, ends, works = [None] * G, [None] * G, [None] * G
        for g in range(G):
            b0, b1 = bounds[g], bounds[g + 1]
            hg = halo[b0:b1] if halo is not None else None
            if ztm:
                e_r = ops.hyena_ct(z, b1 - b0, Tloc, f._fir_w, f.short_filter_bias, table, H, z_halo=hg, poles=f._poles,
                                   state_only=True, b_first=b0, b_total=B)
            else:
                st1[g], e_r = ops.hyena_stage1(z[b0:b1], f._fir_w, f.short_filter_bias, f._poles, H, z_halo=hg)
            ends[g], works[g] = self._gather0(torch.view_as_real(e_r.to(torch.complex64)), async_op=True,
                                              name="state_allgather")
        y = ops.yblk_empty(B * Tloc, D, z.device) if ztm else \
            (torch.empty(B, Tloc, D, dtype=z.dtype, device=z.device) if G > 1 else None)
        for g in range(G):
            b0, b1 = bounds[g], bounds[g + 1]
            works[g].wait()
            s0 = None
            if self.rank > 0:                                # (4) carry entering this shard
                pw = self._pole_powers(f._poles, Tl)                               # [R, D, 8]
                e = torch.view_as_complex(ends[g][: self.rank].double().contiguous())  # [r, b, D, 8]
                idx = torch.arange(self.rank - 1, -1, -1, device=e.device)         # exponent index r-1-q
                s0 = (pw[idx][:, None] * e).sum(0).to(torch.complex64)
            hg = halo[b0:b1] if halo is not None else None
            if ztm:
                ops.hyena_ct(z, b1 - b0, Tloc, f._fir_w, f.short_filter_bias, table, H, z_halo=hg, s0=s0, b_first=b0, b_total=B,
                             y_blk=y, y_row0=b0 * Tloc)       # (blocked y, all row groups into one tensor)
                continue
            yg = ops.hyena_stage2(z[b0:b1], f._fir_w, f.short_filter_bias, f._poles, f._residues, f.D, H, st1[g],
                                  z_halo=hg, s0=s0)
            if G > 1:
                y[b0:b1] = yg
            else:
                y = yg
        if nf:
            # output projection with the post-norm's statistic out of its epilogue, gated MLP with the factor in its epilogue, l3 with the
            # NEXT block's statistic: no RMSNorm pass, no gate kernel (sh/model.py: _mixer_out_rs_ / _mlp_residual_rs_)
            return m._mlp_residual_rs_(blk, x2d, m._mixer_out_rs_(blk, x2d, y, blk.out_filter_dense.weight, blk.out_filter_dense.bias))
        if ztm:
            ops.linear_residual_yblk_(x2d, y, blk.out_filter_dense.weight, bias=blk.out_filter_dense.bias)
        else:
            ops.linear_residual_(x2d, y.view(B * Tloc, D), blk.out_filter_dense.weight, bias=blk.out_filter_dense.bias)
        m._mlp_residual_(blk, x2d, None)                     # (the bias went into the output projection's epilogue, as in model.py)
        return None

    def _attn_block(self, blk, x2d, B, Tloc, Tl, t0, T, rs=None):
        m, ops = self.m, self.m.ops
        D, H, hd = m.hidden_size, m.num_heads, m.head_dim
        mha = blk.inner_mha_cls
        nf = m._nf_ok(B * Tloc, None)
        if nf and rs is not None:
            wqkv_f = m._folded(mha, "_wqkv_f", mha.Wqkv.weight, blk.pre_norm.scale)
            qkv = ops.linear_rs(x2d, rs, wqkv_f, mha.Wqkv.bias, mha.Wqkv.weight, blk.pre_norm.scale, m.eps).view(B, Tloc, 3, H, hd)
        else:
            n1 = ops.rmsnorm(x2d, None, blk.pre_norm.scale, m.eps)
            qkv = ops.linear(n1, mha.Wqkv.weight, mha.Wqkv.bias, mfma=True).view(B, Tloc, 3, H, hd)
        cos, sin = m._rotary(t0, Tloc, x2d.device)
        pre = bool(getattr(ops, "attn_prescale", False)) and hasattr(ops, "attn_q_scale")     # (sh/model.py _attn_block)
        self._ak = {"prescaled": True} if pre else {}
        ops.rope_(qkv, cos, sin, **({"q_scale": ops.attn_q_scale(hd)} if pre else {}))
        if self.attn_mode != "allgather" and H % self.world == 0 and hasattr(self.comm, "all_to_all"):
            a = self._attn_ulysses(qkv, B, Tloc, Tl, T)
        else:
            a = self._attn_allgather(qkv, B, Tloc, Tl, t0)
        if nf:
            return m._mlp_residual_rs_(blk, x2d, m._mixer_out_rs_(blk, x2d, a.view(B * Tloc, D), mha.out_proj.weight, mha.out_proj.bias))
        ops.linear_residual_(x2d, a.view(B * Tloc, D), mha.out_proj.weight, mfma=True, bias=mha.out_proj.bias)
        m._mlp_residual_(blk, x2d, None)
        return None

    def _attn_ulysses(self, qkv, B, Tloc, Tl, T):
        """Batch rows travel in `attn_row_groups` groups: ONE all-to-all and ONE attention launch (rows x H/R heads) per group
        and direction; group g+1's exchange flies under group g's attention, its return exchange under group g+1's."""
        ops, R = self.m.ops, self.world
        H, hd = self.m.num_heads, self.m.head_dim
        Hr = H // R
        G = max(1, min(self.attn_row_groups, B))
        bounds = [(g * B) // G for g in range(G + 1)]
        fwd = []
        for g in range(G):                                   # [nb, Tloc, 3, R, Hr, hd] -> [R, nb, Tl, 3, Hr, hd]: slice r -> rank r
            b0, b1 = bounds[g], bounds[g + 1]
            nb = b1 - b0
            send = self._buf(("a2a_send", g), (R, nb, Tl, 3, Hr, hd), qkv)     # reused by every attention layer
            send[:, :, :Tloc].copy_(qkv[b0:b1].view(nb, Tloc, 3, R, Hr, hd).permute(3, 0, 1, 2, 4, 5))
            if Tloc < Tl:
                send[:, :, Tloc:].zero_()                    # ragged last shard: the pad rows are never attended to
            fwd.append(self._post("a2a_qkv", self.comm.all_to_all, send, async_op=True))
        back = []
        for g in range(G):
            nb = bounds[g + 1] - bounds[g]
            recv, w = fwd[g]
            w.wait()
            if nb == 1:                                      # [R, 1, Tl, ...] IS the (padded) sequence of the row
                full = recv.view(1, R * Tl, 3, Hr, hd)
            else:                                            # source-rank-major -> row-major
                full = self._buf(("a2a_full", g), (nb, R, Tl, 3, Hr, hd), qkv)
                full.copy_(recv.permute(1, 0, 2, 3, 4, 5))
                full = full.view(nb, R * Tl, 3, Hr, hd)
            o = ops.attention(full[:, :T, 0], full[:, :T, 1], full[:, :T, 2], 0, **self._ak)   # [nb, T, Hr, hd], this rank's heads
            ret = self._buf(("a2a_ret", g), (R, nb, Tl, Hr, hd), o)
            if R * Tl == T:
                ret.copy_(o.view(nb, R, Tl, Hr, hd).permute(1, 0, 2, 3, 4))
            else:
                pad = self._buf(("a2a_pad", g), (nb, R * Tl, Hr, hd), o)
                pad[:, :T].copy_(o)
                pad[:, T:].zero_()
                ret.copy_(pad.view(nb, R, Tl, Hr, hd).permute(1, 0, 2, 3, 4))
            back.append(self._post("a2a_out", self.comm.all_to_all, ret, async_op=True))
            fwd[g] = None
        a = torch.empty(B, Tloc, H, hd, dtype=qkv.dtype, device=qkv.device)
        for g in range(G):
            b0, b1 = bounds[g], bounds[g + 1]
            recv, w = back[g]                                # [R(head group), nb, Tl, Hr, hd]
            w.wait()
            a[b0:b1].view(b1 - b0, Tloc, R, Hr, hd).copy_(recv[:, :, :Tloc].permute(1, 2, 0, 3, 4))
        return a

    def _attn_allgather(self, qkv, B, Tloc, Tl, t0):
        ops = self.m.ops
        H, hd = self.m.num_heads, self.m.head_dim
        # K,V per batch row, padded to Tl tokens, gathered asynchronously: [R, Tl, 2, H, hd] == tokens 0..R*Tl-1
        works, bufs = [], []
        for b in range(B):
            kv = qkv[b, :, 1:3]
            if Tloc < Tl:                                    # ragged last shard: pad to Tl in a reused buffer
                pad = self._buf(("kv_pad", b), (Tl, 2, H, hd), qkv)
                pad[:Tloc].copy_(kv)
                pad[Tloc:].zero_()
                kv = pad
            g, w = self._gather0(kv, async_op=True, name="kv_allgather")
            bufs.append(g)
            works.append(w)
        a = torch.empty(B, Tloc, H, hd, dtype=qkv.dtype, device=qkv.device)
        n_keys = t0 + Tloc                                                     # causal: nothing past our last token
        for b in range(B):
            works[b].wait()
            kvg = bufs[b].view(self.world * Tl, 2, H, hd)[:n_keys]
            a[b:b + 1] = ops.attention(qkv[b:b + 1, :, 0], kvg[None, :, 0], kvg[None, :, 1], t0, **self._ak)
        return a

    # ------------------------------------------------------------------ forward / scoring
    @torch.no_grad()
    def hidden_local(self, ids_full: torch.Tensor) -> torch.Tensor:
        """ids_full [B, T] (same on every rank) -> this rank's final-norm hidden states [B * Tloc, D]."""
        m = self.m
        if not m._packed:
            m._pack()
        B, T = ids_full.shape
        self.check_geometry(T)                               # same verdict on every rank, before any collective
        # ... and the same for the token ids: every rank holds ALL of them, so every rank checks ALL of them here.  (The
        # embedding kernel's own check sees one shard: a bad id would raise on that rank only and leave the others
        # hanging in the first halo exchange.)
        if ids_full.numel() and (int(ids_full.min()) < 0 or int(ids_full.max()) >= m.vocab_size):
            raise IndexError(f"input_ids contain values outside [0, {m.vocab_size}) (embedding table has {m.vocab_size} rows)")
        Tl, t0, t1 = self.shard(T)
        Tloc = t1 - t0
        ops = m.ops
        # (checked above, on every rank: no second check -- and no shared flag toggled, ranks may be threads of one process in tests)
        h = ops.embed(ids_full[:, t0:t1].contiguous().to(m.device), m.embedding_layer.weight, validate=False)
        rs = None                                           # row factors handed from block to block (None: the block norms for itself)
        for blk in m.blocks:
            if isinstance(blk, _AttentionBlock):
                rs = self._attn_block(blk, h, B, Tloc, Tl, t0, T, rs)
            else:
                rs = self._hyena_block(blk, h, B, Tloc, Tl, T, rs)
        if m.norm is not None:
            h = ops.rmsnorm(h, None, m.norm.scale, m.eps)
        return h

    @torch.no_grad()
    def forward_local(self, ids_full: torch.Tensor) -> torch.Tensor:
        """ids_full [B, T] (same on every rank) -> this rank's logits [B, Tloc, V]."""
        B, T = ids_full.shape
        h = self.hidden_local(ids_full)
        return self.m.ops.linear(h, self.m.unembed.weight, None).view(B, h.shape[0] // B, self.m.vocab_size)

    @torch.no_grad()
    def score_logprobs(self, ids_full: torch.Tensor) -> torch.Tensor:
        """Log-prob of each next token for this rank's positions: [B, n_local] f32, where global position t
        (t0 <= t < min(t1, T-1)) scores token t+1 -- the rank-local part of evo.scoring.logits_to_logprobs.  On the HIP
        backend the unembedding, the log-softmax and the gather are ONE kernel (evo_unembed_logprob_bf16): the
        [B, Tloc, 512] logits are never written."""
        m, ops = self.m, self.m.ops
        B, T = ids_full.shape
        _, t0, t1 = self.shard(T)
        h = self.hidden_local(ids_full)                      # [B * Tloc, D]
        Tloc = t1 - t0
        n = min(t1, T - 1) - t0
        if n <= 0:
            return torch.zeros(B, 0, dtype=torch.float32, device=h.device)
        tgt = ids_full[:, t0 + 1: t0 + 1 + n].to(h.device)
        hn = h if n == Tloc else h.view(B, Tloc, -1)[:, :n].reshape(B * n, -1).contiguous()
        emb = m.unembed.weight
        if hasattr(ops, "unembed_logprob_ok") and ops.unembed_logprob_ok(hn, emb):
            lp, _ = ops.unembed_logprob(hn, emb, tgt.reshape(-1))
        else:
            lp, _ = ops.logprob_entropy(ops.linear(hn, emb, None), tgt.reshape(-1))
        return lp.view(B, n)

    def gather_logprobs(self, local: torch.Tensor, T: int) -> torch.Tensor:
        """All ranks -> full [B, T-1] log-prob matrix (for score_sequences-style reductions)."""
        Tl, _, _ = self.shard(T)
        pad = torch.zeros(local.shape[0], Tl, dtype=local.dtype, device=local.device)
        pad[:, : local.shape[1]] = local
        g, _ = self._gather0(pad)
        return g.permute(1, 0, 2).reshape(local.shape[0], -1)[:, : T - 1]
