"""Per-channel constant tables of the blocked (matrix-core) form of the Hyena long convolution.

The filter of channel d is a finite sum of modes, h_k = Re sum_s R_s p_s^k.  Time is cut into blocks of L = 32 steps
and tiles of NB = 16 blocks.  With x = x1*v and S the 8 complex modal states (16 reals, S_t = p S_{t-1} + x_t):

    within a block      y0[i]  = sum_{j<=i} h[i-j] x[j]                      T0 [L x L]  lower-triangular Toeplitz
    block aggregate     E[m]   = sum_j  (p^(L-1-j))_m x[j]                   W  [16 x L] (m = 2s: re, 2s+1: im)
    block scan          S_end[a] = p^L S_end[a-1] + E[a]                     P  powers p^(L 2^k), k = 0..3 (Kogge-Stone)
    carry into a block  yc[i]  = sum_m G[i][m] S_start[m]                    G  [L x 16]:  Re(R p^(i+1)), -Im(R p^(i+1))

T0 and W multiply bf16 data on the bf16 matrix cores, so they are stored as sums of bf16 terms (T0: hi + lo, W: hi + mid +
lo -- 2^-17 and 2^-25 relative).  G multiplies the fp32 block states: both are split hi + lo in bf16 (the kernel splits the
states on the fly), G_hi S_hi + G_hi S_lo + G_lo S_hi on the bf16 matrix cores (2^-17; the fp32 matrix cores would be exact
but 8x slower per product).  P stays fp32 (VALU).  Everything is evaluated in fp64 here,
once per model load, and laid out in the MFMA operand order the kernel reads (csrc/hyena_ct.hip).  The same numbers
drive the CPU emulation in tests/test_hyena_blocked.py.
"""
from __future__ import annotations

import torch

L = 32          # steps per block  (= K of v_mfma_f32_16x16x32_bf16)
NB = 16         # blocks per tile  (= N of the 16x16 MFMA tile)
NS = 8          # modes


def _split_bf16(x: torch.Tensor, n: int):
    """x (fp64) -> n bf16 tensors whose fp64 sum approximates x to 2^-(8n+1) relative."""
    parts, r = [], x.clone()
    for _ in range(n):
        p = r.to(torch.float32).to(torch.bfloat16)
        parts.append(p)
        r = r - p.to(torch.float64)
    return parts


def blocked_constants(poles: torch.Tensor, residues: torch.Tensor, dskip: torch.Tensor = None):
    """poles, residues [D, 8, 2] fp32 (+ dskip [D]: filter.D, added to T0's diagonal so that T0 . X = y_conv + x1v * D within
    a block -- the skip term costs nothing) -> dict of per-channel constants (math layout):
         T0 [2][D, L, L] bf16 (hi, lo)      W [3][D, 16, L] bf16 (hi, mid, lo)
         G [D, L, 16] f32                   P [D, 4, 16] f32: (re, im) of p_s^(L 2^k) as [k][2s], [k][2s+1]"""
    D = poles.shape[0]
    p = torch.view_as_complex(poles.double().contiguous())            # [D, 8]
    r = torch.view_as_complex(residues.double().contiguous())
    k = torch.arange(L + 1, dtype=torch.float64, device=poles.device)
    # integer powers by cumulative product in fp64 (exact enough: 33 steps)
    pw = torch.ones(D, NS, L + 1, dtype=torch.complex128, device=poles.device)
    for i in range(1, L + 1):
        pw[..., i] = pw[..., i - 1] * p
    h = (r[..., None] * pw[..., :L]).real.sum(1)                      # [D, L]  h_k, k < L
    idx = torch.arange(L, device=poles.device)
    lag = idx[:, None] - idx[None, :]                                 # i - j
    T0 = torch.where(lag[None] >= 0, h[:, lag.clamp_min(0)], torch.zeros((), dtype=torch.float64, device=poles.device))
    if dskip is not None:
        T0 = T0 + torch.diag_embed(dskip.to(torch.float64)[:, None].expand(D, L))
    Wc = pw[..., :L].flip(-1)                                         # [D, 8, L]: p^(L-1-j)
    W = torch.stack([Wc.real, Wc.imag], 2).reshape(D, 2 * NS, L)      # m = 2s (re), 2s+1 (im)
    Gc = r[..., None] * pw[..., 1:L + 1]                              # [D, 8, L]: R p^(i+1)
    G = torch.stack([Gc.real, -Gc.imag], 2).reshape(D, 2 * NS, L).transpose(1, 2).contiguous()   # [D, L, 16]
    P = []
    q = pw[..., L]                                                    # p^L
    for _ in range(4):
        P.append(torch.stack([q.real, q.imag], -1).reshape(D, 2 * NS))
        q = q * q
    P = torch.stack(P, 1)                                             # [D, 4, 16]
    return {"T0": _split_bf16(T0, 2), "W": _split_bf16(W, 3), "G": G.float(), "Gs": _split_bf16(G, 2), "P": P.float(), "h": h}


# ---- what the bf16 hi / lo operands can hold -------------------------------------------------------------------------------------------
# The carry product  yc[i] = sum_m G[i][m] S_start[m]  runs mode by mode on bf16 hi + lo splits of G and of the fp32 block states (each
# 2^-17 relative), and the block states themselves come out of bf16-split aggregates (2^-17 again).  That is ample while the modes'
# contributions |G| |S| are of the size of their sum -- the synthetic law of the bench, upstream's init, every "plain" regime of
# tests/test_gpu_parity_r6.py.  A filter whose modes CANCEL (large residues of opposite sign on nearly equal poles: the sum is a small
# difference of large terms) amplifies the per-mode 2^-16 by the cancellation factor; at 1 % cancellation the error before the output
# rounding reaches 1e-3 of the channel's output scale (measured, CPU emulation and MI355X) -- the size of a bf16 rounding.  Real
# checkpoints keep poles / residues in fp32 for a reason [REF evo/models.py:146-148], so this is decided per layer when its table is
# built: table_precision() predicts, from the filter alone, the carry term's split error against the output's scale under a unit-variance
# white input; a layer with any channel above TABLE_TOL is routed to the modal kernels (csrc/hyena.hip: fp32 states, no split), which hold
# every regime (tests/test_gpu_parity_r6.py proves both the bound and that the guard fires).
TABLE_TOL = 5.0e-4       # (at 1e-3 a channel sits 1.2e-4 of its scale outside the bf16 bound on MI355X: tests/test_gpu_parity_r6.py)
T_CAP = 131072.0          # longest context the variance bounds are taken over (|p| = 1 states grow like sqrt(T))


def table_precision(poles: torch.Tensor, residues: torch.Tensor, dskip: torch.Tensor = None) -> torch.Tensor:
    """[D] fp64: predicted (split error of the carry product) / (output scale) per channel.
         error_i  = 2^-16 sqrt( sum_s |R_s p_s^(i+1)|^2 V_s ),  V_s = min(1 / (1 - |p_s|^2), T_CAP)   (state variance per unit input variance)
         scale    = sqrt( sum_k h_k^2 + D^2 )                   (closed form over mode pairs, |1 / (1 - q)| capped at T_CAP; at least the
                                                                  energy of the first L taps, which the Toeplitz table holds exactly)
       Over-predicts the measured rms error by ~10x (errors of different modes partly cancel) and tracks the measured worst element."""
    D = poles.shape[0]
    p = torch.view_as_complex(poles.double().contiguous())
    r = torch.view_as_complex(residues.double().contiguous())
    pw = torch.ones(D, NS, L + 1, dtype=torch.complex128, device=poles.device)
    for i in range(1, L + 1):
        pw[..., i] = pw[..., i - 1] * p
    a2 = (p.abs() ** 2).clamp(max=1.0 - 1.0 / T_CAP)
    V = 1.0 / (1.0 - a2)                                                        # [D, 8]
    Gc = r[..., None] * pw[..., 1:L + 1]
    err = (2.0 ** -16) * torch.sqrt(((Gc.real ** 2 + Gc.imag ** 2) * V[..., None]).sum(1)).amax(-1)

    def inv_capped(q):                                                          # 1 / (1 - q), its modulus capped at T_CAP (phase kept)
        d = 1.0 - q
        return torch.where(d.abs() >= 1.0 / T_CAP, 1.0 / d, d.conj() / d.abs().clamp_min(1e-300) * T_CAP)
    pp, pc = p[:, :, None] * p[:, None, :], p[:, :, None] * p[:, None, :].conj()
    rr, rc = r[:, :, None] * r[:, None, :], r[:, :, None] * r[:, None, :].conj()
    eye = torch.eye(NS, dtype=torch.bool, device=poles.device)[None]
    cross = torch.where(eye, torch.zeros((), dtype=torch.complex128, device=poles.device), rc * inv_capped(pc))
    e = 0.5 * ((rr * inv_capped(pp)).sum((1, 2)).real + cross.sum((1, 2)).real + ((r.abs() ** 2) * V).sum(1))
    h = (r[..., None] * pw[..., :L]).real.sum(1)                                # the first L taps
    e = torch.maximum(e, (h ** 2).sum(-1))
    if dskip is not None:
        e = e + dskip.double() ** 2
    return err / torch.sqrt(e).clamp_min(1e-300)


def table_ok(poles: torch.Tensor, residues: torch.Tensor, dskip: torch.Tensor = None, tol: float = TABLE_TOL) -> bool:
    """True when every channel's predicted split error stays below `tol` of its output scale: the layer may run csrc/hyena_ct.hip."""
    return bool((table_precision(poles, residues, dskip) <= tol).all())


# ---- MFMA operand order -------------------------------------------------------------------------------------------------
# One table row per channel: 52 dwords per lane x 64 lanes, in the order csrc/hyena_ct.hip keeps them in registers.
#   v_mfma_f32_16x16x32_bf16  A operand: lane l holds A[row = l & 15][k = 8 (l >> 4) + 0..7]  (4 dwords = 8 bf16)
#   (B operands come from the data; C/D: lane l holds D[row = 4 (l >> 4) + r][col = l & 15], r = 0..3)
TAB_T0 = 0          # [mt 2][split 2][4 dwords]   T0[16 mt + row][k]
TAB_W = 16          # [split 3][4 dwords]         W[row = m][k = j]
TAB_G = 28          # [mt 2][4 dwords]            carry product, K = 32: per k-group kg the B operand is [S_hi(4 kg .. 4 kg + 3) | S_lo(same)]
                    #                             words: G_hi(row, 4 kg + {0,1}), G_hi(.. {2,3}), G_lo(.. {0,1}), G_lo(.. {2,3}); the kernel
                    #                             forms the A operands [G_hi | G_hi] and [G_lo | 0] from them
TAB_P = 36          # [k 4][r 4]                  P[k][4 q + r],  q = l >> 4 (the lane's two modes: re, im, re, im);
                    #                             the kernel keeps these 16 words in LDS, not in registers
TAB_WORDS = 52


def _pack_bf16_pairs(x: torch.Tensor) -> torch.Tensor:
    """[..., 2n] bf16 -> [..., n] int32, low half = even element."""
    u = x.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    return (u[..., 0::2] | (u[..., 1::2] << 16)).to(torch.int32)


def mfma_operand_table(poles: torch.Tensor, residues: torch.Tensor, dskip: torch.Tensor) -> torch.Tensor:
    """[D, 52, 64] int32: the per-lane constant registers of every channel (13,312 B per channel); filter.D is folded into
    the block-Toeplitz diagonal."""
    C = blocked_constants(poles, residues, dskip)
    D = poles.shape[0]
    dev = poles.device
    lane = torch.arange(64, device=dev)
    row, kg = lane & 15, lane >> 4
    tab = torch.empty(D, TAB_WORDS, 64, dtype=torch.int32, device=dev)
    kk = (8 * kg)[:, None] + torch.arange(8, device=dev)[None, :]                 # [64, 8] k indices of a bf16 A fragment
    for mt in range(2):
        for sp in range(2):
            frag = C["T0"][sp][:, (16 * mt + row)[:, None], kk]                   # [D, 64, 8] bf16
            base = TAB_T0 + (mt * 2 + sp) * 4
            tab[:, base:base + 4, :] = _pack_bf16_pairs(frag).transpose(1, 2)
    for sp in range(3):
        frag = C["W"][sp][:, row[:, None], kk]                                    # [D, 64, 8]
        base = TAB_W + sp * 4
        tab[:, base:base + 4, :] = _pack_bf16_pairs(frag).transpose(1, 2)
    P = C["P"]                                                                    # [D, 4, 16]
    for k in range(4):
        for r in range(4):
            tab[:, TAB_P + 4 * k + r, :] = P[:, k, 4 * kg + r].contiguous().view(torch.int32)
    Gh, Gl = C["Gs"]                                                              # [D, L, 16] bf16
    cc = (4 * kg)[:, None] + torch.arange(4, device=dev)[None, :]                 # [64, 4] components of the lane's k-group
    for mt in range(2):
        gh = Gh[:, (16 * mt + row)[:, None], cc]                                  # [D, 64, 4]
        gl = Gl[:, (16 * mt + row)[:, None], cc]
        base = TAB_G + 4 * mt
        tab[:, base:base + 4, :] = _pack_bf16_pairs(torch.cat([gh, gl], -1)).transpose(1, 2)
    return tab.contiguous()


# ---- grouped z layout ------------------------------------------------------------------------------------------------------
GROUP = 16      # channels per workgroup of csrc/hyena_ct.hip


def group_permutation(D: int, n_heads: int, device=None) -> torch.Tensor:
    """perm [3D] int64: grouped column r = cg * 48 + g * 16 + j  <-  reference column  c = h * 3 hd + g * hd + (cg % (hd/16)) * 16 + j
    (h = cg // (hd / 16), hd = D / n_heads, g in {0: x2, 1: x1, 2: v}).  `z_grouped = z[..., perm]`; applied once to the ROWS
    of the projection weight / bias, the GEMM writes the grouped layout directly and the 96 bytes a workgroup needs of a z
    row are contiguous."""
    hd = D // n_heads
    gph = hd // GROUP
    cg = torch.arange(D // GROUP, device=device)
    g = torch.arange(3, device=device)
    j = torch.arange(GROUP, device=device)
    h = cg // gph
    c = (h * 3 * hd)[:, None, None] + (g * hd)[None, :, None] + ((cg % gph) * GROUP)[:, None, None] + j[None, None, :]
    return c.reshape(-1)
