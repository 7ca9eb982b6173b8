"""CPU: the index maps the round-3 kernels rely on, restated in Python and checked exhaustively (no GPU, no library call).

* csrc/gemm.hip: the relabelling of W rows inside the LDS slab that makes a lane's accumulators of an n-tile pair eight
  consecutive output columns; the chunk -> (group, place) map of the group-major epilogue; the gated-MLP weight regrouping.
* csrc/hyena_mfma.hip: the two plane layouts as ONE formula (unit stride US, lo-term offset LO) -- every (channel, step, term)
  gets its own bytes inside the channel's XTCH bytes, and the fp32 (y + x1v D)^T quads lie over exactly the bytes of their steps.
"""
import itertools


def lds_row_to_w_row(rho):
    """LDS row rho of a 256-row W slab holds W row ... (gemm.hip: `wrow` of the DMA plan, blocks of 32 rows)."""
    blk, r0 = divmod(rho, 32)
    return 32 * blk + 8 * ((r0 >> 2) & 3) + 4 * ((r0 >> 4) & 1) + (r0 & 3)


def test_w_row_relabelling_is_a_permutation_and_gives_eight_consecutive_columns():
    rows = [lds_row_to_w_row(r) for r in range(256)]
    assert sorted(rows) == list(range(256))
    # MFMA D layout of a 16 x 16 tile: lane (q = lane >> 4) register r holds row position p = 4 q + r of n tile i, whose A
    # fragment row p was read from LDS row 16 i + p (within the wave's 128 rows)
    for wn_base in (0, 128):
        for b in range(4):                       # strip b = n tiles 2 b, 2 b + 1
            for q in range(4):
                cols = [lds_row_to_w_row(wn_base + 16 * (2 * b + t) + 4 * q + r) for t in (0, 1) for r in range(4)]
                assert cols == list(range(wn_base + 32 * b + 8 * q, wn_base + 32 * b + 8 * q + 8)), (b, q, cols)


def test_group_major_chunk_map():
    """Epilogue mode 2: chunk cc (8 columns) of the grouped projection output -> group cc // 6, 16-byte place cc % 6 of its 96-byte
    row; the kernel divides by a multiply-shift."""
    for cc in range(8192):
        grp = (cc * 10923) >> 16
        assert grp == cc // 6 and 0 <= cc - 6 * grp < 6
    # largest byte offset the kernel forms in 32 bits: (group * Mtot + m) * 96 + place * 16 at the largest supported problem
    n_groups, mtot = 12288 // 48, 131073
    assert ((n_groups - 1) * mtot + (mtot - 1)) * 96 + 5 * 16 < 0xfffffff0
    assert mtot * 12288 * 2 < 0xfffffff0             # the descriptor's num_records


def test_gate_weight_regrouping_puts_matching_columns_in_one_wave_tile():
    """pack_gate_weights + the relabelling: strip 2 p of a wave holds z1 and strip 2 p + 1 holds z2 of the same 32 gated columns."""
    I = 256                                          # inner size (a toy multiple of 128: one 256-row tile covers 128 gated columns)
    def packed_row_source(r):                        # row r of the regrouped weight = (which matrix, its row)
        q, k = divmod(r, 64)
        return (0, 32 * q + k) if k < 32 else (1, 32 * q + k - 32)
    for n0 in range(0, 2 * I, 256):
        for wn in (0, 1):
            for p in (0, 1):
                for lq in range(4):
                    z1 = [packed_row_source(n0 + c) for c in range(wn * 128 + 64 * p + 8 * lq, wn * 128 + 64 * p + 8 * lq + 8)]
                    z2 = [packed_row_source(n0 + c) for c in range(wn * 128 + 64 * p + 32 + 8 * lq, wn * 128 + 64 * p + 32 + 8 * lq + 8)]
                    assert all(m == 0 for m, _ in z1) and all(m == 1 for m, _ in z2)
                    assert [c for _, c in z1] == [c for _, c in z2]
                    gated = n0 // 2 + wn * 64 + 32 * p + 8 * lq          # the kernel's output column (ep_voff_g + soffset)
                    assert [c for _, c in z1] == list(range(gated, gated + 8))


def plane_byte(step, lo, US, LO):
    return (step >> 3) * US + (step & 7) * 2 + lo * LO


def quad_byte(q, US, LO):
    return (q >> 1) * US + (q & 1) * LO


def test_plane_layouts_are_one_formula():
    for (US, LO, XTCH) in ((32, 16, 2064), (16, 1056, 2192)):
        used = {}
        for step, lo in itertools.product(range(512), (0, 1)):
            b = plane_byte(step, lo, US, LO)
            assert 0 <= b and b + 2 <= XTCH
            for byte in (b, b + 1):
                assert byte not in used, (US, step, lo, used.get(byte))
                used[byte] = (step, lo)
        assert len(used) == 2048                     # 512 steps x 2 terms x 2 bytes, no overlap
        # the fp32 result quad q (steps 4 q .. 4 q + 3, 16 bytes) overwrites bytes of its OWN unit's operands only: unit u = q >> 1
        # (8 steps) owns 16 bytes of hi terms and 16 bytes of lo terms -- stage 3's thread reads what stage 1's thread wrote
        for q in range(128):
            b = quad_byte(q, US, LO)
            assert b % 16 == 0 and b + 16 <= XTCH
            owners = {used[x][0] >> 3 for x in range(b, b + 16)}
            assert owners == {q >> 1}, (US, q, owners)
        assert XTCH % 16 == 0


# ---- round 4: csrc/hyena_cs.hip and the blocked-X dense layer ----------------------------------------------------------------
def test_hyena_cs_window_image_and_bank_spread():
    """The LDS image of a tile's z window: 16 blocks of 32 rows x 96 B + a 16-byte pad, filled by 49 one-KiB DMA pieces (lane l of
    piece p writes LDS byte 1024 p + 16 l).  Every (row, signal, quad) a lane reads must hold the stream bytes the kernel thinks it
    holds; a ds_read_b64 of the 32 lanes of a group lands on all 64 banks twice (2-way: the floor for 8 bytes out of 16-byte granules)."""
    import collections
    BLKB = 32 * 96 + 16

    def dma_off(o):                                       # hyena_cs.hip: dma_off[] (source byte of the lane's chunk within the tile)
        blk, within = divmod(o, BLKB)
        return 32 * 96 * blk + (within if within < 32 * 96 else 32 * 96 - 16)
    src = {1024 * p + 16 * l: dma_off(1024 * p + 16 * l) for p in range(49) for l in range(64)}
    for r in range(512):
        for s in range(3):
            for q4 in range(4):
                a = r * 96 + (r >> 5) * 16 + 32 * s + 8 * q4
                assert src[a & ~15] + (a & 15) == r * 96 + 32 * s + 8 * q4
    assert max(src) + 16 <= 49 * 1024
    for la in range(16):                                  # a lane's ten rows: two of history, eight of its own
        for lq in range(4):
            main = (32 * la + 8 * lq) * 96 + la * 16
            hist = main - 192 if lq else (main - 192 - 16 if la else None)
            for i in range(10):
                r = 32 * la + 8 * lq + i - 2
                if r < 0:
                    continue                              # (block 0: the halo slot)
                addr = (hist + i * 96) if i < 2 else main + (i - 2) * 96
                assert addr == r * 96 + (r >> 5) * 16
    for i in range(2, 10):
        for grp in range(2):
            banks = collections.Counter()
            for lane in range(32 * grp, 32 * grp + 32):
                la, lq = lane & 15, lane >> 4
                a = (32 * la + 8 * lq) * 96 + la * 16 + (i - 2) * 96
                for d in range(2):
                    banks[(a // 4 + d) % 64] += 1
            assert max(banks.values()) == 2


def test_hyena_cs_row_permutation_of_the_operand_tables():
    """T0 / G rows as hyena_cs.hip loads them: logical row 16 mt + la (la = 4 q + r) of the kernel = step 8 q + 4 mt + r of the block =
    row 8 (q & 1) + 4 mt + r of the table's M tile q >> 1.  Then the accumulators of lane (block, lq) -- rows 4 lq + r of both M tiles --
    are its own eight steps 8 lq + 4 mt + r, and the map is a permutation of the 32 steps."""
    steps = []
    for mt in range(2):
        for la in range(16):
            q, r = la >> 2, la & 3
            src_mt, src_row = q >> 1, 8 * (q & 1) + 4 * mt + r
            step = 16 * src_mt + src_row
            assert step == 8 * q + 4 * mt + r
            steps.append(step)
    assert sorted(steps) == list(range(32))
    for lq in range(4):
        own = sorted(8 * lq + 4 * mt + r for mt in range(2) for r in range(4))
        assert own == list(range(8 * lq, 8 * lq + 8))


def test_blocked_y_layout_and_the_dense_layers_gather():
    """y blocked = [row block of 128][K / 16 groups][128 rows][16 channels].  (1) the byte hyena_cs.hip stores row R, group cg at;
    (2) HipOps.yblk_to_rows / zg_rows / zg_set_rows invert the layouts; (3) csrc/gemm.hip (XB): the lane that fills 16-byte granule
    s of LDS row r of an X slab (rows m0 .. m0 + 255, channels 64 k .. 64 k + 63) fetches exactly that row's channels 8 s' .. 8 s' + 7
    (s' = the swizzled granule), with voffset = lane part, soffset = m0 * K * 2 + k * 16,384."""
    import torch
    from evo_amd.ops import HipOps
    K, M = 256, 700
    G = K // 16
    y = torch.arange(M * K, dtype=torch.float32).view(M, K)
    nrb = (M + 127) // 128
    pad = torch.zeros(nrb * 128, K)
    pad[:M] = y
    y_blk = pad.view(nrb, 128, G, 16).permute(0, 2, 1, 3).contiguous()
    assert torch.equal(HipOps.yblk_to_rows(y_blk, M), y)
    flat = y_blk.reshape(-1)
    for R in (0, 1, 127, 128, 300, 699):
        for cg in (0, 3, G - 1):
            off = ((R // 128) * G + cg) * (128 * 16) + (R % 128) * 16          # element offset of (row R, channel 16 cg)
            assert flat[off].item() == y[R, 16 * cg].item() and flat[off + 15].item() == y[R, 16 * cg + 15].item()
    # the dense layer's X-tile gather (bytes; bf16 = 2 B per element)
    kb = K * 2
    for m0 in (0, 256):
        for k in range(K // 64):
            soff = m0 * kb + k * 16384
            for wave in range(4):
                for lane in range(64):
                    r0 = 8 * wave + (lane >> 3)
                    xs = ((lane & 7) ^ r0) & 7
                    voff0 = r0 * 32 + (xs >> 1) * 4096 + (xs & 1) * 16
                    for jj in range(8):
                        voff = voff0 + (jj & 3) * 1024 + (jj >> 2) * 128 * kb
                        row, ch = m0 + r0 + 32 * jj, 64 * k + 8 * xs                # what LDS row r0 + 32 jj, granule slot lane & 7 must hold
                        want = (((row // 128) * G + ch // 16) * (128 * 16) + (row % 128) * 16 + ch % 16) * 2
                        assert soff + voff == want, (m0, k, wave, lane, jj)
    # group-major z helpers
    B, T = 3, 5
    zt = torch.arange(B * T * 3 * 32, dtype=torch.float32).view(B, T, 96)
    zg = zt.view(B * T, 2, 48).transpose(0, 1).contiguous()
    assert torch.equal(HipOps.zg_rows(zg, B, T, 3, 2), zt[:, 3:5])
    z2 = torch.zeros_like(zg)
    HipOps.zg_set_rows(z2, B, T, 0, zt)
    assert torch.equal(z2, zg)


def test_ct_lane_positions_history_lanes_and_zt_layout():
    """csrc/hyena_ct.hip: the index arithmetic of the channel-major form, replayed on the host.  Lane (la = lane & 15, lq = lane >> 4)
    of a tile loads eight consecutive positions of a column of z^T (16 bytes): in the main area position b Tp + 512 tile + 32 la + 8 lq,
    in the tail form (T = 512 k + r, r <= 8: HipOps.zt_layout) the ragged last tile loads position tail0 + 8 b + 32 la + 8 lq; the two
    steps before a lane's first are the last pair of lane `hist_src` (lane - 16, or lane + 47 for lq = 0), for lane 0 the last pair of
    lane 63 of the previous tile, at the start of a row the halo / zeros.  Reassembling every channel's stream from these pieces must
    give back the sequence -- ragged T, padded rows, several batch rows, both forms."""
    import torch
    from evo_amd.ops import HipOps
    forms = set()
    for (B, T) in ((3, 1100), (2, 513), (1, 37), (4, 1024), (8, 1025), (2, 1032), (3, 520), (16, 513), (17, 513)):
        Tm, Tp, Mp, r = HipOps.zt_layout(B, T)
        forms.add(r > 0)
        assert Tp % 8 == 0 and 0 <= Tp - Tm < HipOps.ZT_ALIGN and Mp % 256 == 0 and Mp >= B * Tp
        assert (r == 0 and Tm == T) or (1 <= r <= 8 and Tm == T - r and Tm % 512 == 0 and B * r <= 16)
        assert HipOps.zt_geometry(B, T) == (Tp, Mp)
        P = Mp + (256 if r else 0)
        z = torch.arange(1, B * T + 1, dtype=torch.float32).view(B, T, 1)          # one column: value = 1 + flat index (0 = "nothing")
        col = HipOps.zt_from_rows(z, B, T, pad_value=-1.0).permute(1, 0, 2).reshape(P)   # pad positions: a value that must never be used
        bb, tt = torch.meshgrid(torch.arange(B), torch.arange(T), indexing="ij")
        assert torch.equal(col[HipOps.zt_positions(B, T, bb, tt)], z[..., 0])       # zt_positions is the layout zt_from_rows writes
        pos_max = P - 8
        n_tiles = (T + 511) // 512
        for b in range(B):
            carry = None
            for tile in range(n_tiles):
                rw = {}
                for lane in range(64):
                    la, lq = lane & 15, lane >> 4
                    t0 = tile * 512
                    base = Mp + 8 * b + (t0 - Tm) if (r and t0 >= Tm) else b * Tp + t0
                    p = min(base + 32 * la + 8 * lq, pos_max)
                    rw[lane] = col[p:p + 8]
                for lane in range(64):
                    la, lq = lane & 15, lane >> 4
                    t0 = tile * 512 + 32 * la + 8 * lq
                    if lane == 0:
                        hist = torch.zeros(2) if tile == 0 else carry
                    else:
                        src = lane - 16 if lq > 0 else lane + 47
                        hist = rw[src][6:8]
                    for i in range(-2, 8):
                        t = t0 + i
                        if t >= T or t0 >= T:
                            continue                                               # masked steps (n_valid): whatever was loaded
                        got = (hist[i + 2] if i < 0 else rw[lane][i]).item()
                        want = z[b, t, 0].item() if t >= 0 else 0.0
                        assert got == want, (B, T, b, tile, lane, i)
                carry = rw[63][6:8]
    assert forms == {True, False}
    # zt_rows: token-major rows out of z^T; the kernel's byte offset of position p of column c
    for (B, T, C) in ((3, 13, 6), (2, 515, 4)):
        Tm, Tp, Mp, r = HipOps.zt_layout(B, T)
        zz = torch.randn(B, T, C)
        zt = HipOps.zt_from_rows(zz, B, T)
        assert tuple(zt.shape) == (Mp // 256 + (1 if r else 0), C, 256)
        assert torch.equal(HipOps.zt_rows(zt, B, T, T - 2, 2), zz[:, T - 2:])
        assert torch.equal(HipOps.zt_rows(zt, B, T, 0, T), zz)
        flat = zt.reshape(-1)
        for b in range(B):
            for t in (0, 5, T - 4, T - 1):
                p = int(HipOps.zt_positions(B, T, torch.tensor(b), torch.tensor(t)))
                for c in (0, C - 1):
                    off = c * 512 + (p // 256) * C * 512 + (p % 256) * 2
                    assert flat[off // 2].item() == zz[b, t, c].item()
    # the pre-norm's row map (csrc/elementwise.hip rms_out_row) = the positions of the main area + compact tail rows
    for (B, T) in ((8, 1025), (3, 1100), (4, 1028)):
        Tm, Tp, Mp, r = HipOps.zt_layout(B, T)
        for row in range(B * T):
            b, t = divmod(row, T)
            out = b * Tp + t if t < Tm else Mp + b * (T - Tm) + (t - Tm)
            if t < Tm:
                assert out == int(HipOps.zt_positions(B, T, torch.tensor(b), torch.tensor(t)))
            else:
                assert Mp <= out < Mp + 16


def test_rmsnorm_rows_workspace_is_per_thread_and_per_shape():
    """HipOps._xpad_buffer: the same tensor for the same shape on one thread, a fresh zeroed one for another shape, and never the
    tensor another thread is using."""
    import threading
    import torch
    from evo_amd.ops import HipOps
    ops = HipOps.__new__(HipOps)                              # (no library needed for this helper)
    ops._xpad = threading.local()
    a = ops._xpad_buffer(8, 16, "cpu")
    assert a.shape == (8, 16) and a.dtype == torch.bfloat16 and bool((a == 0).all())
    a.fill_(1.0)
    assert ops._xpad_buffer(8, 16, "cpu") is a                # cached
    b = ops._xpad_buffer(12, 16, "cpu")
    assert b is not a and bool((b == 0).all())                # another shape: a new zeroed buffer ...
    c = ops._xpad_buffer(8, 16, "cpu")
    assert c is not a and bool((c == 0).all())                # ... and one shape at a time
    seen = []
    t = threading.Thread(target=lambda: seen.append(ops._xpad_buffer(8, 16, "cpu")))
    t.start(); t.join()
    assert seen[0] is not c and bool((seen[0] == 0).all())    # another thread: its own buffer
