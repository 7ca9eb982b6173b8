"""CPU: the index maps the kernels rely on, restated in Python and checked exhaustively (no GPU, no library call).

* csrc/gemm.hip: the relabelling of W rows inside the LDS slab that makes a lane's accumulators of an n-tile pair eight
  consecutive output columns; the gated-MLP weight regrouping; the X-tile gather from the Hyena operator's blocked y.
* csrc/hyena_ct.hip: lane positions / history lanes / the z^T layout in its two forms.
* csrc/attn_w64.hip: the key order of the S^T accumulator registers under the K-row swap, the V^T tile image and its DMA plan.
(The maps of the retired csrc/hyena_mfma.hip / hyena_cs.hip -- plane layouts, window image, group-major chunks -- went with them in round 5.)
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


def test_blocked_y_layout_and_the_dense_layers_gather():
    """y blocked = [row block of 128][K / 16 groups][128 rows][16 channels].  (1) the byte hyena_ct.hip stores row R, group cg at;
    (2) HipOps.yblk_to_rows inverts the layout; (3) csrc/gemm.hip (XB): the lane that fills 16-byte granule
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


def test_rmsnorm_rows_workspace_is_per_thread_and_per_layout():
    """HipOps._xpad_buffer: the same tensor for the same (B, T) on one thread, a fresh zeroed one for another layout -- also one with the
    SAME padded row count (1 x 513 in tail form, then 1 x 500 padded) --, and never the tensor another thread is using."""
    import threading
    import torch
    from evo_amd.ops import HipOps
    ops = HipOps.__new__(HipOps)                              # (no library needed for this helper)
    ops._xpad = threading.local()
    assert HipOps.zt_layout(1, 513)[2] == HipOps.zt_layout(1, 500)[2] == 512
    a = ops._xpad_buffer(1, 513, 16, "cpu")
    assert a.shape == (512 + 16, 16) and a.dtype == torch.bfloat16 and bool((a == 0).all())
    a.fill_(1.0)
    assert ops._xpad_buffer(1, 513, 16, "cpu") is a           # cached
    b = ops._xpad_buffer(1, 500, 16, "cpu")
    assert b is not a and b.shape == a.shape and bool((b == 0).all())   # another layout with the same rows: a new zeroed buffer ...
    c = ops._xpad_buffer(1, 513, 16, "cpu")
    assert c is not a and bool((c == 0).all())                # ... and one layout at a time
    seen = []
    t = threading.Thread(target=lambda: seen.append(ops._xpad_buffer(1, 513, 16, "cpu")))
    t.start(); t.join()
    assert seen[0] is not c and bool((seen[0] == 0).all())    # another thread: its own buffer
    ops.release_workspaces()
    assert ops._xpad_buffer(1, 513, 16, "cpu") is not c


def test_attn_w64_key_order_vt_image_and_dma_plan():
    """csrc/attn_w64.hip, replayed on the host.  (1) K rows are read with bits 2 and 3 of the key index swapped; with the MFMA C/D layout
    (accumulator register r of lane (col, half) = row (r & 3) + 8 (r >> 2) + 4 half) register r then holds key 16 (r >> 3) + 8 half +
    (r & 7) of the 32-key half: the eight keys of a P^T fragment (registers 8 u .. 8 u + 7) are CONTIGUOUS, so a V^T fragment is one
    16-byte read.  (2) V^T tile image [128 d][128 B]: chunk c of row d sits at slot c ^ ((d >> 1) & 7); the 16 lanes of every
    ds_read_b128 lane group hit 16 distinct 16-byte bank windows.  (3) the DMA plan: piece j (8 rows), lane L writes LDS bytes
    j * 1024 + 16 L = row 8 j + (L >> 3), slot L & 7, and must fetch chunk (L & 7) ^ ((row >> 1) & 7) of that row -- the kernel's
    per-lane constant uses ((4 (wave & 1) + (L >> 4)) & 7) for every piece j = wave + 4 jj of the wave."""
    swap = lambda m: (m & 0x13) | ((m & 4) << 1) | ((m & 8) >> 1)          # noqa: E731  (the kernel's `krow`)
    assert sorted(swap(m) for m in range(32)) == list(range(32))
    for half in (0, 1):
        for r in range(16):
            row = (r & 3) + 8 * (r >> 2) + 4 * half                          # accumulator row of register r
            assert swap(row) == 16 * (r >> 3) + 8 * half + (r & 7)           # = the key that row's K fragment was read from
    # K reads stay conflict-free: a b128 lane group reads 16 rows that are distinct mod 16 (rows of 272 bytes)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for g in groups:
        assert len({swap(l) % 16 for l in g}) == 16
        for chunk in range(8):                                              # V^T: slot16 = (d & 1) * 8 + (chunk ^ ((d >> 1) & 7))
            assert len({((d & 1) * 8 + (chunk ^ ((d >> 1) & 7))) for d in g}) == 16
    for wave in range(4):
        for jj in range(4):
            j = wave + 4 * jj
            for L in range(64):
                row = 8 * j + (L >> 3)
                assert ((row >> 1) & 7) == ((4 * (wave & 1) + (L >> 4)) & 7)


def test_norm_fold_host_maps_and_the_stream_row_remap_of_the_transposed_projection():
    """Round 5 (csrc/gemm.hip NF): the host-side pieces of the RMSNorm folding, restated and checked without a GPU.
    * HipOps.fold_norm_scale = bf16(W diag(g)) with an fp32 product; * which rows the persistent launch takes and which go to the
    weight-streaming launches (_nf_main_rows); * which z^T layouts can be filled straight from the stream (zt_stream_rows_ok);
    * the kernel's row remap of mode 3 (`src_row`: position tile origin n0 -> n0 + (n0 / 256 / tm_tiles) * row_skip) lands every
    position of the main area on the stream row HipOps.zt_positions assigns it, and a 256-position tile never straddles two batch rows;
    * the statistic's geometry: 16 lanes x 8 m tiles x 2 wave rows cover a tile's 256 rows once per 128-column strip."""
    import torch
    from evo_amd.ops import HipOps
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 32, generator=g).to(torch.bfloat16)
    s = (1 + 0.1 * torch.randn(32, generator=g)).to(torch.bfloat16)
    f = HipOps.fold_norm_scale(w, s)
    assert f.dtype == torch.bfloat16 and f.is_contiguous() and torch.equal(f, (w.float() * s.float()[None, :]).to(torch.bfloat16))
    assert torch.equal(HipOps.fold_norm_scale(w, torch.ones(32, dtype=torch.bfloat16)), w)          # g = 1: the weight itself
    main = lambda M: HipOps._nf_main_rows(HipOps, M)
    assert [main(M) for M in (65544, 131073, 8192, 2100, 2050, 272, 273)] == [65536, 131072, 8192, 2100, 2048, 256, 273]
    ok = lambda B, T: HipOps.zt_stream_rows_ok(HipOps, B, T)
    assert ok(8, 8193) and ok(1, 131073) and ok(2, 1026) and ok(1, 8192) and ok(4, 1280)
    assert not ok(3, 700) and not ok(3, 5003) and not ok(1, 8210)                                     # padded rows: rmsnorm_rows writes the copy
    for B, T in ((8, 8193), (2, 1026), (3, 1025), (1, 131073)):
        Tm, Tp, Mp, r = HipOps.zt_layout(B, T)
        assert r > 0 and Tm % 256 == 0 and Mp == B * Tm
        tm_tiles = Tm // 256
        bb = torch.arange(B)[:, None].expand(B, Tm).reshape(-1)
        tt = torch.arange(Tm)[None, :].expand(B, Tm).reshape(-1)
        pos = HipOps.zt_positions(B, T, bb, tt)                                                       # position of (b, t), t < Tm
        n0 = pos // 256 * 256
        src = n0 + (n0 // 256 // tm_tiles) * r + pos % 256                                            # the kernel's source row of that position
        assert torch.equal(src, bb * T + tt)
        assert torch.equal(n0 // Tm, (n0 + 255) // Tm)                                                # one batch row per tile
        assert int(src.max()) < B * T
    rows = sorted(wm * 128 + 16 * j + l15 for wm in range(2) for j in range(8) for l15 in range(16))
    assert rows == list(range(256))
