// Hyena operator, single pass over GROUP-MAJOR z, CHANNEL-STATIONARY waves (gfx950).  Round 4.
//
// Same operator and the same blocked arithmetic as csrc/hyena_mfma.hip -- FIR(k = 3) + bias on x2 | x1 | v, x = x1 * v, long
// convolution with h_k = Re sum_s R_s p_s^k as  y0 = T0 . X,  E = W . X  (v_mfma_f32_16x16x32_bf16, operands split into bf16
// terms), a Kogge-Stone scan of the 16 block aggregates (fp32 VALU, DPP row shifts),  y = y0 + G . S,  (y + x1v D) * x2 --
// but a different division of labour.  hyena_mfma.hip has every wave PRODUCE 64 steps of all 16 channels (FIR, x1 * v, bf16
// planes in LDS), then CONSUME two channels (planes back from LDS into the matrix cores), then gate and store a third time
// slice: three LDS round trips per value (planes, parked x2, fp32 y^T) and an 8 x 8 transposition by v_perm; measured
// issue-bound (profiles/r03_hyena_mfma_notes.txt: ~590 instructions per wave and tile, ~400 KB through the LDS pipe per tile
// and CU, 0.44-0.49 of HBM).  Here a wave OWNS its channels for the whole tile:
//
//   workgroup = one 16-channel group of one batch row stream, HC_NW waves; wave = HC_CPW = 16 / HC_NW channels (4 | 2);
//   lane (la = lane & 15, lq = lane >> 4) = block la (32 steps), steps 8 lq .. 8 lq + 7 of it -- which is exactly the lane
//   that holds those steps in the B operand of the 16x16x32 MFMA (k = 8 lq + j, n = la).  So the FIR outputs of a lane,
//   x = x1 * v split into bf16 hi + lo, ARE its B-operand registers: no planes, no transposition.  The D operand of
//   T0 . X has rows (mt, 4 lq + r): with the ROWS of T0 and G permuted at load time (logical row 16 mt + 4 q + r <- step
//   8 q + 4 mt + r) a lane's eight accumulators are the outputs of its own eight steps, whose FIR'd x2 it also holds: the
//   gate is in-lane, no parked x2, no fp32 y^T.  What goes through LDS: the z window (DMA in, one 8-byte read per row,
//   signal and 4 channels) and the bf16 outputs (32-byte rows staged for 16-byte stores).
//
//   window  [2 buffers][16 blocks x (32 rows x 96 B + 16 B pad)]: tile k + 1 is fetched (buffer_load ... lds, bounded
//           descriptor: rows past the end of the stream read as zeros) while tile k is computed; the pad puts the 16 blocks
//           a ds_read_b64 touches on 16 distinct bank quads (a column read of 8 B out of 16-byte DMA granules cannot do
//           better than 2-way: 4 LDS cycles per read).  The two rows before a tile (FIR history) are kept by the lane that
//           read them last (lane 63: rows 510, 511) in a 2 x 192 B halo slot.
//   ONE barrier per tile: it publishes window(k) (every wave waited for its own DMA pieces) and staging(k - 1).
//
// Per tile and wave (HC_CPW = 4, X_lo kept): ~720 VALU, 52 MFMA, 30 + 20 LDS reads, 8 LDS writes, 13 DMA pieces, 4 stores
// -- against 2 x 590 instructions per SIMD in hyena_mfma.hip; LDS traffic 1.2 k instead of 2.9 k+ cycles per tile.
// Entry point and reference citation: include/evo_mi355x.h (evo_hyena_cs_zg).
#include "common.h"
#include "../../include/evo_mi355x.h"

#ifndef HC_XLO
#define HC_XLO 1                            // 1: X = x1 * v as bf16 hi + lo; 0: one bf16 term (where the reference rounds it)
#endif
#ifndef HC_NW
#define HC_NW 8                             // waves per workgroup: 8 (two per SIMD, 2 channels each) | 4 (one per SIMD, 4 channels each)
#endif
#define HC_CH 16
#define HC_CPW (HC_CH / HC_NW)              // channels per wave: 4 | 2
#define HC_NPAIR (HC_CPW / 2)               // channel pairs per wave: 2 | 1
#define HC_THREADS (64 * HC_NW)
#define HC_TT 512                           // steps per tile = 16 blocks of 32
#define HC_ROWB 96                          // bytes of a z row of one group: x2 16 | x1 16 | v 16 bf16
#define HC_BLKB (32 * HC_ROWB + 16)         // LDS image of a block: 32 rows + 16 B pad = 3,088
#define HC_NPIECE 49                        // 1 KiB DMA pieces per window (16 x 3,088 = 49,408 B -> 48.25)
#define HC_WINB (HC_NPIECE * 1024)          // 50,176
#define HC_PPW ((HC_NPIECE + HC_NW - 1) / HC_NW)        // DMA pieces per wave and tile: 13 | 7
#define HC_STGB (HC_TT * 32 + 16 * 16)      // staged outputs: 512 rows of 32 B, 16 B pad after every 32 rows = 16,640
#define HC_RW (HC_TT / HC_NW)               // rows a wave stores: 128 | 64
#define HC_NST (HC_RW / 32)                 // 16-byte stores per lane and tile: 4 | 2
#define HC_OFF_WIN 0
#define HC_OFF_STG (2 * HC_WINB)
#define HC_OFF_HALO (HC_OFF_STG + 2 * HC_STGB)          // [2 parities][2 rows][96 B]
#define HC_OFF_PW (HC_OFF_HALO + 2 * 2 * HC_ROWB)       // scan powers [16 ch][4 k][16 components] f32 = 4 KiB
#define HC_OFF_FW (HC_OFF_PW + HC_CH * 4 * 16 * 4)      // FIR taps + bias, fp32 pairs: [8 pairs][3 signals][4] x 8 B = 768 B
#define HC_OFF_XS (HC_OFF_FW + 768)                     // end-state scratch: per wave the hi | lo planes of one channel (2 KiB)
#define HC_LDS (HC_OFF_XS + HC_NW * 2048)
static_assert(HC_LDS <= 160 * 1024, "LDS");
#ifndef HC_PROFILE
#define HC_PROFILE 0
#endif
#ifndef HC_SPREAD
#define HC_SPREAD 2                         // where a wave issues its window pieces: 2 (default): inside the FIR loops, the first ~40 % of the tile's
                                            // arithmetic (0.513 / 0.549 of 8 TB/s at 8 x 8,193 / 1 x 131,073 with the blocked y); 0: all right behind
                                            // the barrier (0.483 / 0.518); 1: spread over the whole tile (they land late: 0.49 / 0.52)
#endif
#define HC_YBLK 128                         // rows per block of the BLOCKED y layout (HcArgs.y_blk): [row block][group][128 rows][16 channels]
#define HC_TABW 52                          // dwords per lane of a channel's operand table (evo_amd/hyena_tables.py)

typedef float hc_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t hc_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t hc_u32x2 __attribute__((ext_vector_type(2)));
typedef int hc_srd __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2_t hc_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t hc_bf2(uint32_t w) { f32x2_t r = {bf_lo(w), bf_hi(w)}; return r; }
__device__ __forceinline__ hc_u32x4 hc_u4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { hc_u32x4 r = {a, b, c, d}; return r; }
template <int D_>
__device__ __forceinline__ float hc_shr(float v) {          // value of lane (a - D_) of the 16-lane row, 0 where a < D_
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + D_, 0xf, 0xf, true));
}

// With two waves per SIMD sharing the matrix pipe hipcc under-pads MFMA -> consumer distances (hyena_mfma.hip, round 2): the
// bursts are fenced there.  One wave per SIMD (HC_NW = 4) is the case its hazard recogniser models: no fences.
#ifndef HC_NOFENCE
#define HC_NOFENCE 0
#endif
#ifndef HC_PRIO
#define HC_PRIO 0                           // 1: the second half of the waves (the losers of the age-based issue arbitration) run at priority 1
#endif
#if HC_NW == 8 && !HC_NOFENCE
#define HC_FENCE_NOP() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HC_FENCE_NOP() do { } while (0)
#endif

struct hc_false { static constexpr bool value = false; };
struct hc_true { static constexpr bool value = true; };

struct HcArgs {
    const unsigned char* z; const uint32_t* z_halo; const uint16_t* fir_w; const uint16_t* fir_b;
    const uint32_t* tab; unsigned char* y; const float* s0; float* s_out; const float* poles;
    int B; int T; int D; int n_tiles; int n_groups; int nb_split;
    int64_t z_group_rows;                                   // rows between two groups' streams in z (>= B * T)
    int64_t y_rowbytes;
    int y_blk;                                              // y is [ceil(rows / 128)][D / 16][128][16] bf16 instead of [rows][D] (see vm_store)
    int64_t y_row0, y_rows;                                 // blocked y: row of batch row 0 / total rows of the [rows, D] matrix it stands for
};

// words of a channel's table kept in registers: T0 [mt 2][hi, lo][4] = 0..15, W [hi, mid][4] = 16..23, G [mt 2][4] = 24..31
#define HC_NTB 32

// SO = "state only": the same walk, nothing written but the end state (stage 1 of a sequence-parallel shard).
// HALF: with HC_CPW = 2 the wave takes dword HALF of every 8-byte (4-channel) window read; unused with HC_CPW = 4.
// WS = "want state": the instantiation that finishes the state after the last token (`s_out`); scoring launches carry no trace of it.
template <bool SO, bool WS, int HALF>
__device__ __forceinline__ void hc_run(const HcArgs& a, unsigned char* smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int la = lane & 15, lq = lane >> 4;
    const int q4 = HC_CPW == 4 ? wave : wave >> 1;          // the 4-channel quad of the group this wave reads
    const int ch0 = 4 * q4 + (HC_CPW == 4 ? 0 : 2 * HALF);   // first channel of the wave within the group
    int b0, cg;
    {
        const int bid = blockIdx.x, total = gridDim.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int per_xcd = total >> 3;                     // host guarantees total % 8 == 0
        const int s = xcd * per_xcd + slot;                 // contiguous stream ids per XCD: the four groups of a y line together
        b0 = s / a.n_groups;
        cg = s - b0 * a.n_groups;
    }
    const int h = cg >> 3, cw0 = (cg & 7) * HC_CH;
    const int d0 = h * 128 + cw0;                           // first output channel of the group
    const int Ti = a.T;
    const int n_rows = (a.B - b0 + a.nb_split - 1) / a.nb_split;
    const int n_steps = n_rows * a.n_tiles;
    struct Cur { int b, tile; };
    auto advance = [&](Cur& c) { if (++c.tile == a.n_tiles) { c.tile = 0; c.b += a.nb_split; } };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- constants: FIR taps / bias of the wave's channel pairs (registers), scan powers (LDS), MFMA operand tables (registers)
    // FIR taps + bias as fp32 pairs in LDS, [wave][pair][signal x2, x1, v][tap 0, 1, 2, bias] (wave-uniform values: left to the
    // compiler they go to SGPRs, 48 per wave, which spill and bind the one-scalar-operand limit of the packed FMAs)
    f32x2_t* fwl = (f32x2_t*)(smem + HC_OFF_FW) + wave * (HC_NPAIR * 12);
    if (lane < HC_NPAIR * 12) {
        const int pp = lane / 12, rem = lane - 12 * pp, g = rem >> 2, k = rem & 3;
        const int c = h * 384 + g * 128 + cw0 + ch0 + 2 * pp;
        f32x2_t v;
        if (k < 3) { v[0] = bf_to_f(a.fir_w[c * 3 + k]); v[1] = bf_to_f(a.fir_w[(c + 1) * 3 + k]); }
        else { v[0] = bf_to_f(a.fir_b[c]); v[1] = bf_to_f(a.fir_b[c + 1]); }
        fwl[lane] = v;
    }
    float* pwl = (float*)(smem + HC_OFF_PW);                 // [ch][k][16] f32
    if (tid < HC_CH * 16) {
        const int c = tid >> 4, m = tid & 15;               // component m = 4 q + r sits in table word 36 + 4 k + r of the lanes with lq = q
        const uint32_t* tp = a.tab + ((int64_t)(d0 + c) * HC_TABW) * 64 + (m >> 2) * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) pwl[(c * 4 + k) * 16 + m] = __builtin_bit_cast(float, tp[(36 + 4 * k + (m & 3)) * 64]);
    }
    uint32_t tb[HC_CPW][HC_NTB];
    {
        // row permutation of T0 and G: logical row 16 mt + la (la = 4 q + r) of this kernel is step 8 q + 4 mt + r of the block,
        // i.e. row 8 (q & 1) + 4 mt + r of the table's M tile q >> 1 -- a lane's accumulators (rows 4 lq + r of both M tiles) are
        // then its own eight steps 8 lq + 4 mt + r.  W (rows = state components) and the K order stay as the table has them.
        const int q = la >> 2, r = la & 3;
#pragma unroll
        for (int cc = 0; cc < HC_CPW; ++cc) {
            const uint32_t* tp = a.tab + ((int64_t)(d0 + ch0 + cc) * HC_TABW) * 64;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int src_mt = q >> 1, src_lane = (8 * (q & 1) + 4 * mt + r) + 16 * lq;
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int w = 0; w < 4; ++w) tb[cc][8 * mt + 4 * sp + w] = tp[((src_mt * 2 + sp) * 4 + w) * 64 + src_lane];
#pragma unroll
                for (int w = 0; w < 4; ++w) tb[cc][24 + 4 * mt + w] = tp[(28 + 4 * src_mt + w) * 64 + src_lane];
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) tb[cc][16 + w] = tp[(16 + w) * 64 + lane];
        }
        // waited for HERE: a load whose first use sits in the tile loop would put the compiler's s_waitcnt vmcnt(0) there
#pragma unroll
        for (int cc = 0; cc < HC_CPW; ++cc)
#pragma unroll
            for (int w = 0; w < HC_NTB; ++w) asm volatile("" : "+v"(tb[cc][w]));
    }

    // ---- window DMA: piece p covers bytes [1024 p, 1024 p + 1024) of the padded image; lane l carries the 16 bytes at
    //      1024 p + 16 l = block * 3,088 + within  <-  stream byte  block * 3,072 + within  (pad chunks re-fetch the block's last
    //      chunk and are never read).  This wave issues pieces wave, wave + HC_NW, ... (past the last piece: the last piece again,
    //      same bytes to the same place, so that every wave issues HC_PPW pieces and the vmcnt arithmetic is one constant).
    uint32_t dma_off[HC_PPW];
#pragma unroll
    for (int i = 0; i < HC_PPW; ++i) {
        int p = wave + HC_NW * i;
        p = p < HC_NPIECE ? p : HC_NPIECE - 1;
        const int o = 1024 * p + 16 * lane;
        const int blk = o / HC_BLKB, within = o - HC_BLKB * blk;
        dma_off[i] = (uint32_t)(32 * HC_ROWB * blk + (within < 32 * HC_ROWB ? within : 32 * HC_ROWB - 16));
    }
    // The pieces of a window are NOT issued in a burst: the CU issues one 1 KiB LDS-DMA piece per ~60 clocks, so the 49 pieces of
    // a tile behind a barrier keep the last wave waiting ~2.9 k clocks at the vector-memory queue before it computes anything
    // (tools/hc_stage_profile.py, round 4: per tile 8.4 k clocks = 2.9 k DMA issue + 0.6 k stores + 1.1 k window reads + 3.5 k
    // compute, one after the other).  `vm_prepare` builds the tile's descriptor, `vm_piece` issues ONE piece; `compute` calls it
    // at points spread over the tile's arithmetic, so that a wave's memory instructions queue while its partner on the SIMD computes.
    struct Vm { hc_srd d; uint32_t base; bool dma; bool st; Cur cst; int sbuf; hc_u32x4 sdat[HC_NST]; };
    auto vm_prepare = [&](Vm& v, const Cur& c, int buf) {
        // descriptor of this tile of the (group, batch row) stream: base = first row of the tile, num_records = bytes up to the
        // end of the row's T tokens (the hardware returns zeros beyond: the ragged last tile needs no clamping)
        const int64_t row0 = (int64_t)cg * a.z_group_rows + (int64_t)c.b * Ti + (int64_t)c.tile * HC_TT;
        const uint64_t a64 = (uint64_t)(a.z + row0 * HC_ROWB);
        const int64_t left = ((int64_t)Ti - (int64_t)c.tile * HC_TT) * HC_ROWB;
        v.d[0] = (int)(uint32_t)a64;
        v.d[1] = (int)(uint32_t)(a64 >> 32);
        // (no next tile: num_records = 0 -- the pieces are still issued, as zeros into the free buffer, so that every interval
        //  carries the same instructions: no branch around a piece, constant vmcnt arithmetic)
        v.d[2] = (int)(v.dma && left > 0 ? (left < 0x7fffffff ? left : 0x7fffffff) : 0);
        v.d[3] = 0x00020000;
        v.base = lds0 + HC_OFF_WIN + buf * HC_WINB;
    };
    auto vm_piece = [&](const Vm& v, const int i) {
        int p = wave + HC_NW * i;
        p = p < HC_NPIECE ? p : HC_NPIECE - 1;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     ::"s"(v.base + 1024 * p), "v"(dma_off[i]), "s"(v.d) : "memory", "m0");
    };
    auto dma_win = [&](const Cur& c, int buf) {              // (the whole window at once: prologue only)
        Vm v;
        v.dma = true;
        vm_prepare(v, c, buf);
#pragma unroll
        for (int i = 0; i < HC_PPW; ++i) vm_piece(v, i);
    };

    // ---- per-lane LDS addresses (buffer 0; + HC_WINB / + HC_STGB / + 192 for buffer 1)
    //   window row of step 8 lq + i of block la: (32 la + 8 lq + i) * 96 + la * 16; the two history rows of the lanes with lq = 0
    //   lie in the previous block (16 B pad in between) or, for block 0, in the halo slot
    const uint32_t win_main = HC_OFF_WIN + (32 * la + 8 * lq) * HC_ROWB + la * 16 + q4 * 8 + (HC_CPW == 2 ? 4 * HALF : 0);
    const uint32_t win_hist0 = lq > 0 ? win_main - 2 * HC_ROWB : (la > 0 ? win_main - 2 * HC_ROWB - 16 : 0u);   // (la = 0, lq = 0: halo)
    const uint32_t halo_rd = HC_OFF_HALO + q4 * 8 + (HC_CPW == 2 ? 4 * HALF : 0);
    const bool from_halo = lane == 0;
    const uint32_t stg_wr = HC_OFF_STG + (32 * la + 8 * lq) * 32 + la * 16 + q4 * 8 + (HC_CPW == 2 ? 4 * HALF : 0);
    const uint64_t y64 = (uint64_t)a.y;
    const hc_u32x4 ysrd = {(uint32_t)y64, (uint32_t)(y64 >> 32) & 0xffffu,
                           (uint32_t)((a.y_blk ? (a.y_rows + HC_YBLK - 1) / HC_YBLK * HC_YBLK : (int64_t)a.B * Ti) * a.y_rowbytes), 0x00020000u};
    const uint32_t yrb = (uint32_t)a.y_rowbytes;

    float carry[HC_CPW][4];                                  // tile-entering state: components 4 lq .. 4 lq + 3, valid in lanes la = 0
    const float first_blk = la == 0 ? 1.f : 0.f;

    // ---- store of tile `c` from staging buffer `buf`: the wave's HC_RW rows x 32 B, 16 B per lane
    // the staged outputs of tile k - 1 go to registers right behind the barrier (with the window reads: one LDS round trip for both);
    // the stores themselves are issued later, between the arithmetic
    auto vm_store_fetch = [&](Vm& v) {
        if (SO) return;
#pragma unroll
        for (int hs = 0; hs < HC_NST; ++hs) {
            const int row = HC_RW * wave + 32 * hs + (lane >> 1);
            v.sdat[hs] = *(const hc_u32x4*)(smem + HC_OFF_STG + v.sbuf * HC_STGB + row * 32 + (row >> 5) * 16 + (lane & 1) * 16);
        }
    };
    auto vm_store = [&](const Vm& v, const int hs) {
        if (SO) return;
        const Cur& c = v.cst;
        const int t0 = c.tile * HC_TT;
        const bool full = t0 + HC_TT <= Ti;
        const uint32_t row0 = (uint32_t)(((int64_t)c.b * Ti + t0) * a.y_rowbytes + d0 * 2);
        const int row = HC_RW * wave + 32 * hs + (lane >> 1);
        // bounds-checked buffer store: rows past the end get an offset beyond num_records and are dropped, so that the VM
        // counter sees exactly HC_NST stores per interval (also dropped that way: the stores of the first interval, which has
        // no previous tile)
        // Row-major y costs a store instruction ~270 clocks at the vector-memory unit (its 32 rows are 32 different cache lines,
        // 32 B of each) -- 16 stores per tile and CU were ~4 k of the ~8.4 k clocks a tile took.  The BLOCKED form keeps a group's
        // 16 channels of 128 consecutive rows together (4 KiB): a store instruction covers 8 whole lines, +13...21 % on the kernel
        // (profiles/r04_hyena_cs_notes.txt); the output projection's dense layer gathers it (csrc/gemm.hip, XB).
        const uint32_t R = (uint32_t)(a.y_row0 + (int64_t)c.b * Ti + t0 + row);  // row of the [rows, D] matrix
        const uint32_t yb = ((R / HC_YBLK) * (uint32_t)a.n_groups + (uint32_t)cg) * (HC_YBLK * 32) + (R % HC_YBLK) * 32 + (lane & 1) * 16;
        const uint32_t off = !(v.st && (full || t0 + row < Ti)) ? 0xfffffff0u : (a.y_blk ? yb : row0 + (uint32_t)row * yrb + (lane & 1) * 16);
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v.sdat[hs]), "v"(off), "s"(ysrd) : "memory");
    };

    // ---- the window of a tile into registers: ten rows (two of history) x three signals, 8 bytes = 4 channels each (the compiler
    //      places the waits where the values are used: the FIR of v and x1 starts while the x2 rows are still on their way)
    constexpr int NQ = HC_CPW == 4 ? 2 : 1;                  // dwords per read
    uint32_t raw[3][10][NQ];
    auto read_win = [&](uint32_t (&dst)[3][10][NQ], int buf) {
        const uint32_t wm = win_main + buf * HC_WINB;
        const uint32_t wh = from_halo ? halo_rd + buf * (2 * HC_ROWB) : win_hist0 + buf * HC_WINB;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                if (SO && g == 0) { for (int e = 0; e < NQ; ++e) dst[g][i][e] = 0u; continue; }
                const unsigned char* p = smem + (i < 2 ? wh + i * HC_ROWB : wm + (i - 2) * HC_ROWB) + 32 * g;
                if (HC_CPW == 4) { const hc_u32x2 v = *(const hc_u32x2*)p; dst[g][i][0] = v[0]; dst[g][i][NQ - 1] = v[1]; }
                else dst[g][i][0] = *(const uint32_t*)p;
            }
    };
    // rows 510, 511 of a tile are the history of the next one: lane 63 holds them (its rows 8, 9) and leaves them in halo slot `slot`
    auto put_history = [&](const uint32_t (&src)[3][10][NQ], int slot) {
        if (lane == 63) {
            unsigned char* hp = smem + HC_OFF_HALO + slot * (2 * HC_ROWB) + q4 * 8 + (HC_CPW == 2 ? 4 * HALF : 0);
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (HC_CPW == 4) { hc_u32x2 v = {src[g][8 + i][0], src[g][8 + i][NQ - 1]}; *(hc_u32x2*)(hp + i * HC_ROWB + 32 * g) = v; }
                    else *(uint32_t*)(hp + i * HC_ROWB + 32 * g) = src[g][8 + i][0];
                }
        }
    };
    // FIR history of a new batch row (or zeros) into halo slot `slot` (wave 0)
    auto seed_history = [&](int b, int slot) {
        if (wave == 0 && lane < 48) {
            const int r = lane / 24, wq = lane - 24 * r;     // 24 dwords per row: x2 | x1 | v
            uint32_t v = 0u;
            if (a.z_halo) {
                const uint32_t* hp = a.z_halo + ((int64_t)b * 2 + r) * (a.D * 6 / 4) + cg * (HC_ROWB / 4) + wq;
                asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(hp) : "memory");
            }
            *(uint32_t*)(smem + HC_OFF_HALO + slot * (2 * HC_ROWB) + r * HC_ROWB + wq * 4) = v;
        }
    };

#if HC_SPREAD == 1
#define HC_VMP(I) vm_piece(vm, (I))
#define HC_VMS(I) vm_store(vm, (I))
#define HC_VMF(I)
#elif HC_SPREAD == 2      // the pieces inside the FIR loops (the first ~40 % of the tile's arithmetic: they land before the next barrier)
#define HC_VMP(I)
#define HC_VMS(I) vm_store(vm, (I))
#define HC_VMF(I) { if ((I) < HC_PPW) vm_piece(vm, (I)); }
#else
#define HC_VMP(I)
#define HC_VMS(I)
#define HC_VMF(I)
#endif
    // ---- one tile of this wave's channels, from the registers read_win filled
    auto compute = [&](const Cur& c, int buf, const Vm& vm, auto ragged_t) {
        constexpr bool RAGGED = decltype(ragged_t)::value;
        const int t0 = c.tile * HC_TT;
        const bool last_tile = c.tile == a.n_tiles - 1;
        if (c.tile == 0) {                                   // a new sequence: zero state or the carried one
#pragma unroll
            for (int cc = 0; cc < HC_CPW; ++cc) {
                hc_f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
                if (a.s0) {                                  // (inline asm: a visible load would put s_waitcnt vmcnt(0) into every tile)
                    const float* sp = a.s0 + ((int64_t)c.b * a.D + d0 + ch0 + cc) * 16 + 4 * lq;
                    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(c4) : "v"(sp) : "memory");
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) carry[cc][r] = c4[r];
            }
        }
        const int n_valid = RAGGED ? Ti - (t0 + 32 * la + 8 * lq) : 8;         // steps of this lane inside the sequence
        static_assert(HC_PPW == 1 + 2 * HC_NPAIR + 2 * HC_CPW && HC_NST == HC_CPW, "issue points of the window pieces / stores");
        HC_VMP(0);

        bf16x8_t xh[HC_CPW];
#if HC_XLO
        bf16x8_t xl[HC_CPW];
#endif
        f32x2_t x2f[HC_NPAIR][8];
#pragma unroll
        for (int pp = 0; pp < HC_NPAIR; ++pp) {
            // FIR of x1 and v, x = x1 * v; the lane's eight steps of both channels -> the two channels' B operands
            f32x2_t x[8];
            {
                const f32x2_t* fp_ = fwl + pp * 12;
                const f32x2_t w10 = fp_[4], w11 = fp_[5], w12 = fp_[6], b1 = fp_[7];
                const f32x2_t w20 = fp_[8], w21 = fp_[9], w22 = fp_[10], b2 = fp_[11];
                f32x2_t m2a = hc_bf2(raw[1][0][pp]), m1a = hc_bf2(raw[1][1][pp]), m2b = hc_bf2(raw[2][0][pp]), m1b = hc_bf2(raw[2][1][pp]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2_t ca = hc_bf2(raw[1][i + 2][pp]), cb = hc_bf2(raw[2][i + 2][pp]);
                    const f32x2_t x1c = hc_fma(w12, ca, hc_fma(w11, m1a, hc_fma(w10, m2a, b1)));
                    const f32x2_t vc = hc_fma(w22, cb, hc_fma(w21, m1b, hc_fma(w20, m2b, b2)));
                    x[i] = x1c * vc;
                    if (RAGGED && i >= n_valid) { x[i][0] = 0.f; x[i][1] = 0.f; }      // past the end: nothing enters the modes
                    m2a = m1a; m1a = ca; m2b = m1b; m1b = cb;
                    if ((i & 1) == 0) HC_VMF(7 * pp + (i >> 1));
                }
            }
            HC_VMP(1 + 2 * pp);
            if (!SO) {
                const f32x2_t* fp_ = fwl + pp * 12;
                const f32x2_t w00 = fp_[0], w01 = fp_[1], w02 = fp_[2], b0f = fp_[3];
                f32x2_t m2 = hc_bf2(raw[0][0][pp]), m1 = hc_bf2(raw[0][1][pp]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2_t cx = hc_bf2(raw[0][i + 2][pp]);
                    x2f[pp][i] = hc_fma(w02, cx, hc_fma(w01, m1, hc_fma(w00, m2, b0f)));
                    m2 = m1; m1 = cx;
                    if (i % 3 == 0) HC_VMF(7 * pp + 4 + i / 3);
                }
            }
            HC_VMP(2 + 2 * pp);
            if (SO) {                                        // (no x2 loop in the state-only walk: its pieces go out here)
#pragma unroll
                for (int i = 0; i < 3; ++i) HC_VMF(7 * pp + 4 + i);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                uint32_t hw[4];
#if HC_XLO
                uint32_t lw[4];
#endif
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hw[j] = pack_bf2(x[2 * j][e], x[2 * j + 1][e]);
#if HC_XLO
                    lw[j] = pack_bf2(x[2 * j][e] - bf_lo(hw[j]), x[2 * j + 1][e] - bf_hi(hw[j]));
#endif
                }
                xh[2 * pp + e] = __builtin_bit_cast(bf16x8_t, hc_u4(hw[0], hw[1], hw[2], hw[3]));
#if HC_XLO
                xl[2 * pp + e] = __builtin_bit_cast(bf16x8_t, hc_u4(lw[0], lw[1], lw[2], lw[3]));
#endif
            }
        }

        // ---- per channel: E = W . X and y0 = T0 . X on the matrix cores, the block scan, y += G . S
        hc_f32x4 yv[HC_CPW][2];
#pragma unroll
        for (int cc = 0; cc < HC_CPW; ++cc) {
            const uint32_t* t_ = tb[cc];
            // the four scan powers of the channel, requested HERE: the scan's levels are asm blocks the compiler keeps in order, and a
            // load placed next to its level waits out the whole LDS latency four times per channel (round 4: ~2 k clocks per tile)
            const hc_f32x4* pwc = (const hc_f32x4*)(pwl + (ch0 + cc) * 64) + lq;
            const hc_f32x4 pw4[4] = {pwc[0], pwc[4], pwc[8], pwc[12]};
            __builtin_amdgcn_sched_barrier(0);               // (the loads stay in front of the MFMA burst: its ~250 clocks cover their latency)
#define HC_FRAG(BASE) __builtin_bit_cast(bf16x8_t, hc_u4(t_[(BASE)], t_[(BASE) + 1], t_[(BASE) + 2], t_[(BASE) + 3]))
            const hc_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            hc_f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HC_FRAG(20), xh[cc], zero4, 0, 0, 0);       // W_mid . X_hi
#if HC_XLO
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HC_FRAG(16), xl[cc], e, 0, 0, 0);                    // W_hi . X_lo
#endif
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HC_FRAG(16), xh[cc], e, 0, 0, 0);                    // W_hi . X_hi
            if (!SO)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    hc_f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HC_FRAG(8 * mt + 4), xh[cc], zero4, 0, 0, 0);
#if HC_XLO
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HC_FRAG(8 * mt), xl[cc], acc, 0, 0, 0);
#endif
                    yv[cc][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HC_FRAG(8 * mt), xh[cc], acc, 0, 0, 0);
                }
            HC_FENCE_NOP();
            HC_VMP(1 + 2 * HC_NPAIR + 2 * cc);
            // Kogge-Stone scan of the 16 block aggregates -> state entering every block; the tile's end state
            float sv[4] = {e[0], e[1], e[2], e[3]};
            {
                const hc_f32x4 P = pw4[0];
                const float c0 = first_blk * carry[cc][0], c1 = first_blk * carry[cc][1], c2 = first_blk * carry[cc][2], c3 = first_blk * carry[cc][3];
                sv[0] = fmaf(-P[1], c1, fmaf(P[0], c0, sv[0]));
                sv[1] = fmaf(P[1], c0, fmaf(P[0], c1, sv[1]));
                sv[2] = fmaf(-P[3], c3, fmaf(P[2], c2, sv[2]));
                sv[3] = fmaf(P[3], c2, fmaf(P[2], c3, sv[3]));
            }
            // One Kogge-Stone level on the lane's two modes, (re, im) += P * (re, im) of the lane SH blocks to the left (0 beyond the
            // row's start): the DPP shift is an operand of the FMA (v_fmac_f32_dpp) -- five instructions per mode and level instead of
            // two DPP moves + four FMAs.  im' is built in a scratch register (the old im is still needed for re'), so the im registers
            // alternate from level to level.  Wait states: a VGPR written by a VALU instruction may be read through DPP two
            // instructions later at the earliest -- inside a level and from level to level the order below keeps that distance
            // (hand-written: the compiler does not look into inline asm); the leading / trailing s_nop cover the compiler's own
            // instructions around the block.
#define HC_LEVEL(KK, SH, PRE, POST)                                                                                       \
            {                                                                                                             \
                const hc_f32x4 P = pw4[(KK)];                                                                             \
                float t1, t3;                                                                                             \
                asm volatile(PRE                                                                                          \
                             "v_mov_b32 %2, %4\n\t"                                                                       \
                             "v_fmac_f32_dpp %2, %4, %6 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
                             "v_fmac_f32_dpp %2, %0, %7 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
                             "v_fmac_f32_dpp %0, %0, %6 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
                             "v_fmac_f32_dpp %0, %4, -%7 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
                             "v_mov_b32 %3, %5\n\t"                                                                       \
                             "v_fmac_f32_dpp %3, %5, %8 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
                             "v_fmac_f32_dpp %3, %1, %9 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
                             "v_fmac_f32_dpp %1, %1, %8 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
                             "v_fmac_f32_dpp %1, %5, -%9 row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1" POST     \
                             : "+v"(sv[0]), "+v"(sv[2]), "=&v"(t1), "=&v"(t3)                                             \
                             : "v"(sv[1]), "v"(sv[3]), "v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]));                       \
                sv[1] = t1;                                                                                               \
                sv[3] = t3;                                                                                               \
            }
            HC_LEVEL(0, 1, "s_nop 1\n\t", "") HC_LEVEL(1, 2, "", "") HC_LEVEL(2, 4, "", "") HC_LEVEL(3, 8, "", "\n\ts_nop 1")
#undef HC_LEVEL
            float st[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) st[r] = hc_shr<1>(sv[r]) + first_blk * carry[cc][r];                 // state ENTERING block la
#pragma unroll
            for (int r = 0; r < 4; ++r)
                carry[cc][r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv[r]), 0x121, 0xf, 0xf, false));
            HC_VMP(2 + 2 * HC_NPAIR + 2 * cc);
            if (WS && last_tile) {
                // state after the last token T - 1, which sits in block a_ at local step r_ - 1: the recurrence over the block's first
                // r_ steps from the state entering it.  Lane s (< 8) takes mode s.  The x values are the very bf16 terms the matrix
                // cores consumed: the channel's fragments go through the wave's scratch planes (hi | lo, time-contiguous).
                unsigned char* xs = smem + HC_OFF_XS + wave * 2048;
                *(bf16x8_t*)(xs + (4 * la + lq) * 16) = xh[cc];
#if HC_XLO
                *(bf16x8_t*)(xs + 1024 + (4 * la + lq) * 16) = xl[cc];
#endif
                const int tin = Ti - t0;                     // 1..512 valid steps of this tile
                const int a_ = (tin - 1) >> 5, r_ = tin - 32 * a_;
                const int src = (16 * ((lane & 7) >> 1) + a_) * 4;
                float g4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) g4[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, st[r])));
                float sre = (lane & 1) ? g4[2] : g4[0], sim = (lane & 1) ? g4[3] : g4[1];
                const int dch = d0 + ch0 + cc;
                f32x2_t pp2;
                {
                    const float* qp = a.poles + ((int64_t)dch * 8 + (lane & 7)) * 2;
                    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(pp2) : "v"(qp) : "memory");
                }
                const float pre = pp2[0], pim = pp2[1];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int j = 0; j < r_; ++j) {
                    const unsigned char* up = xs + (4 * a_ + (j >> 3)) * 16 + (j & 7) * 2;
                    float xv = bf_to_f(*(const uint16_t*)up);
#if HC_XLO
                    xv += bf_to_f(*(const uint16_t*)(up + 1024));
#endif
                    const float nre = fmaf(pre, sre, fmaf(-pim, sim, xv));
                    sim = fmaf(pre, sim, pim * sre);
                    sre = nre;
                }
                if (lane < 8) {
                    float* so = a.s_out + ((int64_t)c.b * a.D + dch) * 16 + 2 * lane;
                    const f32x2_t sv2 = {sre, sim};
                    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(so), "v"(sv2) : "memory");
                }
            }
            if (!SO) {
                // y += G . S_start with the block states split hi + lo on the fly
                const uint32_t h01 = pack_bf2(st[0], st[1]), h23 = pack_bf2(st[2], st[3]);
                const uint32_t l01 = pack_bf2(st[0] - bf_lo(h01), st[1] - bf_hi(h01));
                const uint32_t l23 = pack_bf2(st[2] - bf_lo(h23), st[3] - bf_hi(h23));
                const bf16x8_t sb = __builtin_bit_cast(bf16x8_t, hc_u4(h01, h23, l01, l23));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const uint32_t* g_ = t_ + 24 + 4 * mt;
                    yv[cc][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, hc_u4(g_[0], g_[1], g_[0], g_[1])),
                                                                       sb, yv[cc][mt], 0, 0, 0);
                    yv[cc][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, hc_u4(g_[2], g_[3], 0u, 0u)),
                                                                       sb, yv[cc][mt], 0, 0, 0);
                }
                HC_FENCE_NOP();
            }
            HC_VMS(cc);                                      // (tile k - 1, staged before the barrier; one 16-byte store per lane)
#undef HC_FRAG
        }
        // ---- gate and stage: accumulator (mt, r) of a lane is its step 4 mt + r; one dword (two channels) per step and pair
        if (!SO) {
            unsigned char* sp = smem + stg_wr + buf * HC_STGB;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t o[HC_NPAIR];
#pragma unroll
                for (int pp = 0; pp < HC_NPAIR; ++pp)
                    o[pp] = pack_bf2(yv[2 * pp][i >> 2][i & 3] * x2f[pp][i][0], yv[2 * pp + 1][i >> 2][i & 3] * x2f[pp][i][1]);
                if (HC_CPW == 4) { hc_u32x2 v = {o[0], o[HC_NPAIR - 1]}; *(hc_u32x2*)(sp + i * 32) = v; }
                else *(uint32_t*)(sp + i * 32) = o[0];
            }
        }
    };

    // ---- the pipeline: one barrier per tile.  VM queue of a wave per interval, in issue order: the HC_PPW pieces of window(k + 1) and
    //      the HC_NST stores of tile k - 1, interleaved with the arithmetic; the LAST memory instruction of an interval is a store and
    //      the last piece precedes it directly, so before the barrier of interval k + 1 "window(k + 1) landed" is vmcnt(1).
#if HC_PROFILE      // -DHC_PROFILE=1: every wave accumulates shader-clock deltas per phase and writes 16 floats at y + 64 B * (HC_NW * workgroup
    //                 + wave) (tools/hc_stage_profile.py; a timing build: it overwrites y)
    uint64_t tprof[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
    const uint64_t rt0 = __builtin_amdgcn_s_memrealtime(), ck0 = __builtin_readcyclecounter();
#define HC_STAMP(K) { const uint64_t now_ = __builtin_readcyclecounter(); tprof[K] += now_ - tlast; tlast = now_; }
#else
#define HC_STAMP(K)
#endif
    Cur c_cmp = {b0, 0}, c_dma = {b0, 0}, c_st = {b0, 0};
#if HC_PRIO
    if (wave >= HC_NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    if (n_steps > 0) { dma_win(c_dma, 0); advance(c_dma); }
#if HC_PROFILE
    tlast = __builtin_readcyclecounter();
#endif
    for (int k = 0; k <= n_steps; ++k) {
        const int buf = k & 1;
        Cur nx = c_cmp;
        advance(nx);
        const bool next_row_start = nx.tile == 0;
        if (k < n_steps) {
            if (c_cmp.tile == 0) seed_history(c_cmp.b, buf);  // a new row: its FIR history (or zeros) into this tile's halo slot
            // interval k - 1 issued, in this order: ..., the last piece of window(k), ONE more store (of tile k - 2; dropped when k = 1)
            if (!SO && k >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HC_SPREAD == 1 ? 1 : HC_NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        HC_STAMP(0);
        __syncthreads();                                     // window(k) and the halo slot -> everybody; staging(k - 1) complete
        HC_STAMP(1);
        Vm vm;
        vm.dma = k + 1 < n_steps;
        vm_prepare(vm, c_dma, buf ^ 1);
        if (vm.dma) advance(c_dma);
        vm.st = !SO && k >= 1;
        vm.cst = c_st;
        vm.sbuf = buf ^ 1;
        if (vm.st) advance(c_st);
        vm_store_fetch(vm);
#if HC_SPREAD == 0
#pragma unroll
        for (int i = 0; i < HC_PPW; ++i) vm_piece(vm, i);
        HC_STAMP(2);
        if (k < n_steps) {
#pragma unroll
            for (int hs = 0; hs < HC_NST; ++hs) vm_store(vm, hs);
        }
        HC_STAMP(3);
#endif
        if (k < n_steps) {
            read_win(raw, buf);
            if (!next_row_start) put_history(raw, buf ^ 1);
            HC_STAMP(5);
            if (c_cmp.tile * HC_TT + HC_TT <= Ti) compute(c_cmp, buf, vm, hc_false{});
            else compute(c_cmp, buf, vm, hc_true{});
            c_cmp = nx;
        } else {
#pragma unroll
            for (int hs = 0; hs < HC_NST; ++hs) vm_store(vm, hs);      // the last tile's outputs
        }
        HC_STAMP(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if HC_PROFILE
    if (lane == 0) {
        float* o = (float*)a.y + 16 * (blockIdx.x * HC_NW + wave);
        for (int k = 0; k < 5; ++k) o[k] = (float)tprof[k];
        o[8] = (float)tprof[5];
        o[5] = (float)n_steps;
        o[6] = (float)(__builtin_amdgcn_s_memrealtime() - rt0);      // 100 MHz ticks, whole workgroup
        o[7] = (float)(__builtin_readcyclecounter() - ck0);          // shader clocks, whole workgroup
    }
#endif
#undef HC_STAMP
}

template <bool SO, bool WS>
__global__ __launch_bounds__(HC_THREADS, 1) void hyena_cs_kernel(HcArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[HC_LDS];      // the only LDS object
#if HC_CPW == 4
    hc_run<SO, WS, 0>(a, smem);
#else
    if (threadIdx.x & 64) hc_run<SO, WS, 1>(a, smem);            // (wave-uniform: the two waves of a quad take its two channel pairs)
    else hc_run<SO, WS, 0>(a, smem);
#endif
}

extern "C" int evo_hyena_cs_zg(const void* z, const void* z_halo, const void* fir_w, const void* fir_b, const void* table, void* y,
                               const float* s0, float* s_out, const float* poles, int64_t B, int64_t T, int64_t D, int64_t n_heads,
                               int64_t z_group_rows, int64_t state_only, int64_t y_blocked_rows, int64_t y_row0, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || n_heads <= 0 || D != n_heads * 128) return -1;
    const int64_t yrb = D * 2;
    if (B * T * yrb >= 0xfffffff0ll || T * HC_ROWB >= 0x7fffffffll) return -1;     // 32-bit offsets inside y / one stream of z
    if (y_blocked_rows && (y_row0 < 0 || y_row0 + B * T > y_blocked_rows || (y_blocked_rows + HC_YBLK) * yrb >= 0xfffffff0ll)) return -1;
    if (z_group_rows < B * T) return -1;
    if (s_out && !poles) return -1;
    if (state_only ? !s_out : !y) return -1;
    const int64_t groups = D / HC_CH;
    // workgroups = groups x nb_split, ~one per CU: a workgroup walks batch rows b0, b0 + nb_split, ... of its group
    int64_t nb_split = (256 + groups - 1) / groups;
    if (nb_split > B) nb_split = B;
    const int64_t streams = groups * nb_split;
    if (streams % 8 != 0 || B * groups > 0x7fffffff) return -1;                      // equal runs of streams per XCD
    HcArgs a;
    a.z = (const unsigned char*)z; a.z_halo = (const uint32_t*)z_halo; a.fir_w = (const uint16_t*)fir_w; a.fir_b = (const uint16_t*)fir_b;
    a.tab = (const uint32_t*)table; a.y = (unsigned char*)y; a.s0 = s0; a.s_out = s_out; a.poles = poles;
    a.B = (int)B; a.T = (int)T; a.D = (int)D; a.n_tiles = (int)((T + HC_TT - 1) / HC_TT); a.n_groups = (int)groups;
    a.nb_split = (int)nb_split; a.z_group_rows = z_group_rows; a.y_rowbytes = yrb;
    a.y_blk = y_blocked_rows ? 1 : 0; a.y_row0 = y_row0; a.y_rows = y_blocked_rows;
    if (state_only) hipLaunchKernelGGL((hyena_cs_kernel<true, true>), dim3((unsigned)streams), dim3(HC_THREADS), 0, (hipStream_t)stream, a);
    else if (s_out) hipLaunchKernelGGL((hyena_cs_kernel<false, true>), dim3((unsigned)streams), dim3(HC_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((hyena_cs_kernel<false, false>), dim3((unsigned)streams), dim3(HC_THREADS), 0, (hipStream_t)stream, a);
    return evo_launch_status();
}
