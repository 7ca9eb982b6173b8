// Hyena operator, single pass over CHANNEL-MAJOR z (z^T), channel-stationary waves, no input window in LDS (gfx950).  Round 4; since
// round 5 the ONLY single-pass form (its predecessors -- token-major hyena_mfma.hip of rounds 2-3, group-major hyena_cs.hip of round 4
// -- are retired; their measurement notes stay under profiles/r03_*, profiles/r04_hyena_cs_notes.txt).
//
// Division of labour: a workgroup owns 16 channels of one batch row and walks the sequence in tiles of 512 steps; a wave owns two
// channels for the whole tile.  Lane (la = lane & 15, lq = lane >> 4) holds steps 8 lq .. 8 lq + 7 of 32-step block la: its FIR outputs
// ARE its column of the B operand of v_mfma_f32_16x16x32_bf16 (block-Toeplitz product T0 and block aggregates G, operands split into
// bf16 hi + lo terms, fp32 accumulation; the rows of T0 / G are permuted at load time so that a lane's accumulators are its own eight
// steps and meet the FIR'd x2 of the same lane), the 16 blocks' modal states meet in a DPP scan in fp32, the carry product runs on
// the matrix cores again with hi/lo-split states.  No bf16 planes, no parked x2, no fp32 y^T travel through LDS.
// Where z comes from is what makes this form: the group-major predecessor DMA'd a tile's window into LDS and read it back as 30
// conflicted ds_read_b64 per wave and tile (window reads 1.5-2.4 k and DMA waits 0.6-0.7 k of the 7.7 k clocks a tile took).
// Here z arrives TRANSPOSED, z^T [3 D columns][time] (stored in blocks of 256 positions): the projection's dense layer is launched
// with its operands swapped (evo_linear_t_mfma_bf16, csrc/gemm.hip: out[n][m] = W . x^T, the same kernel, whole-line stores, an
// output tile = one contiguous 128 KiB block), so a lane's eight steps of one
// channel and signal are 16 consecutive bytes: one global_load_dwordx4 straight into the registers the FIR reads, a wave's load =
// 1 KiB contiguous.  No window, no DMA descriptor, no wait for other waves' pieces; the two steps of FIR history come from the
// neighbouring lane (ds_bpermute) or, for lane 0, from lane 63 of the previous tile (v_readlane).  LDS carries only the staged bf16
// outputs (32-byte rows for the blocked y) and constants; the one barrier per tile publishes the staging buffer and nothing else.
//
//   loads of tile k + 1 are issued right after the FIR of tile k has consumed its raw values (~35 % into the tile); the compiler
//   places the wait at their first use, the top of the next iteration.
//
// Alignment: 16-byte loads need (position of step 0 of a batch row) % 8 == 0 -- the caller pads every batch row to a multiple of
// 8 positions (HipOps.rmsnorm_rows writes the normalised rows at b * Tp + t; the pad positions hold anything: they are
// masked exactly like the ragged end of the last tile).  Tail form (HtArgs.tail_T): for T = 512 k + r, r <= 8, the rows hold 512 k
// positions (no padding) and the last r tokens of a row sit in a tail block behind the main area -- they are the whole ragged last
// tile, which loads them from there (voff_of); the projection's dense layer then covers B * 512 k rows, a whole number of tiles.
// Entry point and reference citation: include/evo_mi355x.h (evo_hyena_ct).
#include "common.h"
#include "../../include/evo_mi355x.h"

#ifndef HT_XLO
#define HT_XLO 1                            // 1: X = x1 * v as bf16 hi + lo (profiles/r04_hyena_cs_notes.txt: the closed X_lo question)
#endif
#define HT_NW 8                             // waves per workgroup, two per SIMD
#define HT_CH 16
#define HT_CPW 2                            // channels per wave
#define HT_THREADS (64 * HT_NW)
#define HT_TT 512                           // steps per tile = 16 blocks of 32
#define HT_STGB (HT_TT * 32 + 16 * 16)      // staged outputs: 512 rows of 32 B, 16 B pad after every 32 rows = 16,640
#define HT_RW (HT_TT / HT_NW)               // rows a wave stores: 64
#define HT_NST (HT_RW / 32)                 // 16-byte stores per lane and tile: 2
#define HT_OFF_STG 0
#define HT_OFF_PW (2 * HT_STGB)                         // scan powers [16 ch][4 k][16 components] f32 = 4 KiB
#define HT_OFF_FW (HT_OFF_PW + HT_CH * 4 * 16 * 4)      // FIR taps + bias, fp32 pairs: [8 waves][3 signals][4] x 8 B = 768 B
#define HT_OFF_XS (HT_OFF_FW + 768)                     // end-state scratch: per wave the hi | lo planes of one channel (2 KiB)
#define HT_LDS (HT_OFF_XS + HT_NW * 2048)
#ifndef HT_PROFILE
#define HT_PROFILE 0
#endif
#define HT_ZBLK 256                         // positions per block of z^T: [position block][3 D][256]
#define HT_YBLK 128                         // rows per block of the BLOCKED y layout: [row block][group][128 rows][16 channels]
#define HT_TABW 52                          // dwords per lane of a channel's operand table (evo_amd/hyena_tables.py)
#define HT_NTB 32                           // of which in registers: T0 [mt 2][hi, lo][4] = 0..15, W [hi, mid][4] = 16..23, G [mt 2][4] = 24..31

typedef float ht_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t ht_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2_t ht_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ ht_u32x4 ht_u4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { ht_u32x4 r = {a, b, c, d}; return r; }
template <int D_>
__device__ __forceinline__ float ht_shr(float v) {          // value of lane (a - D_) of the 16-lane row, 0 where a < D_
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + D_, 0xf, 0xf, true));
}
// two waves per SIMD share the matrix pipe: hipcc under-pads MFMA -> consumer distances (found on the round-2 kernel); bursts are fenced
#define HT_FENCE_NOP() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

struct ht_false { static constexpr bool value = false; };
struct ht_true { static constexpr bool value = true; };

struct HtArgs {
    const unsigned char* zt; const uint16_t* z_halo; const uint16_t* fir_w; const uint16_t* fir_b;
    const uint32_t* tab; unsigned char* y; const float* s0; float* s_out; const float* poles;
    int B; int T; int D; int n_tiles; int n_groups; int nb_split;
    int64_t zt_pitch;                                       // positions z^T holds per column (% 256 == 0: whole blocks)
    int64_t row_pitch;                                      // positions between two batch rows (>= T, % 8 == 0)
    int64_t zt_row0;                                        // position of batch row 0, step 0 (% 8 == 0)
    int64_t tail_T, tail_pos0;                              // tail form (0: none): steps >= tail_T of batch row b sit at tail_pos0 + 8 b + (t - tail_T)
    int64_t y_rowbytes;
    int y_blk;                                              // y is [ceil(rows / 128)][D / 16][128][16] bf16 instead of [rows][D]
    int64_t y_row0, y_rows;                                 // blocked y: row of batch row 0 / total rows of the [rows, D] matrix it stands for
    int64_t y_pitch;                                        // rows of y between two batch rows (>= T; = T unless the caller keeps rows of its own behind a batch row's T)
};

// SO = "state only": the same walk, nothing written but the end state (stage 1 of a sequence-parallel shard; x2 is not even read).
// WS = "want state": the instantiation that finishes the state after the last token (`s_out`).
template <bool SO, bool WS>
__global__ __launch_bounds__(HT_THREADS, 1) void hyena_ct_kernel(HtArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[HT_LDS];      // the only LDS object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int la = lane & 15, lq = lane >> 4;
    const int ch0 = 2 * wave;                               // first channel of the wave within the group
    int b0, cg;
    {
        const int bid = blockIdx.x, total = gridDim.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int per_xcd = total >> 3;                     // host guarantees total % 8 == 0
        const int s = xcd * per_xcd + slot;                 // contiguous stream ids per XCD: the four groups of a y line together
        b0 = s / a.n_groups;
        cg = s - b0 * a.n_groups;
    }
    const int h = cg >> 3, cw0 = (cg & 7) * HT_CH;
    const int d0 = h * 128 + cw0;                           // first output channel of the group
    const int Ti = a.T;
    const int n_rows = (a.B - b0 + a.nb_split - 1) / a.nb_split;
    const int n_steps = n_rows * a.n_tiles;
    struct Cur { int b, tile; };
    auto advance = [&](Cur& c) { if (++c.tile == a.n_tiles) { c.tile = 0; c.b += a.nb_split; } };

    // ---- constants: FIR taps / bias of the wave's channel pair (LDS: wave-uniform values left to the compiler go to SGPRs, 48 per
    //      wave, which spill and bind the one-scalar-operand limit of the packed FMAs), scan powers (LDS), operand tables (registers)
    f32x2_t* fwl = (f32x2_t*)(smem + HT_OFF_FW) + wave * 12;     // [signal x2, x1, v][tap 0, 1, 2, bias]
    if (lane < 12) {
        const int g = lane >> 2, k = lane & 3;
        const int c = h * 384 + g * 128 + cw0 + ch0;        // column of z / row of the FIR weights [REF model.py: x2 | x1 | v per head]
        f32x2_t v;
        if (k < 3) { v[0] = bf_to_f(a.fir_w[c * 3 + k]); v[1] = bf_to_f(a.fir_w[(c + 1) * 3 + k]); }
        else { v[0] = bf_to_f(a.fir_b[c]); v[1] = bf_to_f(a.fir_b[c + 1]); }
        fwl[lane] = v;
    }
    float* pwl = (float*)(smem + HT_OFF_PW);                 // [ch][k][16] f32
    if (tid < HT_CH * 16) {
        const int c = tid >> 4, m = tid & 15;               // component m = 4 q + r sits in table word 36 + 4 k + r of the lanes with lq = q
        const uint32_t* tp = a.tab + ((int64_t)(d0 + c) * HT_TABW) * 64 + (m >> 2) * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) pwl[(c * 4 + k) * 16 + m] = __builtin_bit_cast(float, tp[(36 + 4 * k + (m & 3)) * 64]);
    }
    uint32_t tb[HT_CPW][HT_NTB];
    {
        // row permutation of T0 and G: logical row 16 mt + la (la = 4 q + r) of this kernel is step 8 q + 4 mt + r of the block,
        // i.e. row 8 (q & 1) + 4 mt + r of the table's M tile q >> 1 -- a lane's accumulators (rows 4 lq + r of both M tiles) are
        // then its own eight steps 8 lq + 4 mt + r.  W (rows = state components) and the K order stay as the table has them.
        const int q = la >> 2, r = la & 3;
#pragma unroll
        for (int cc = 0; cc < HT_CPW; ++cc) {
            const uint32_t* tp = a.tab + ((int64_t)(d0 + ch0 + cc) * HT_TABW) * 64;
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
        for (int cc = 0; cc < HT_CPW; ++cc)
#pragma unroll
            for (int w = 0; w < HT_NTB; ++w) asm volatile("" : "+v"(tb[cc][w]));
    }

    // ---- the six input streams of the wave: column (signal g, channel ch0 + cc) of z^T; a lane's eight steps are 16 bytes at
    //      2 * (position of the tile's step 0 + 32 la + 8 lq) -- across the wave 1 KiB contiguous
    const unsigned char* zs[3][HT_CPW];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int cc = 0; cc < HT_CPW; ++cc) zs[g][cc] = a.zt + (int64_t)(h * 384 + g * 128 + cw0 + ch0 + cc) * (HT_ZBLK * 2);
    // z^T is stored in blocks of HT_ZBLK = 256 positions, [position block][3 D columns][256] (the dense layer's output tiles are then
    // contiguous: csrc/gemm.hip, MODE 3): position p of a column sits at byte (p / 256) * 3 D * 512 + (p % 256) * 2 behind the column's base
    const uint32_t pos_max = (uint32_t)(a.zt_pitch - 8);     // (positions past the end: clamped -- they are masked steps)
    const uint32_t blk_bytes = (uint32_t)a.D * 3u * (HT_ZBLK * 2);
    auto voff_of = [&](const Cur& c) -> uint32_t {
        // (tail form: T = 512 k + r, r <= 8 -- the row's last r tokens came through the weight-streaming kernel into the tail block of z^T
        //  (HipOps.zt_layout); they are the whole ragged last tile, whose lane 0 loads them; every other lane of that tile is masked)
        const int64_t t0 = (int64_t)c.tile * HT_TT;
        const int64_t p = (a.tail_T && t0 >= a.tail_T ? a.tail_pos0 + (int64_t)c.b * 8 + (t0 - a.tail_T)
                                                      : a.zt_row0 + (int64_t)c.b * a.row_pitch + t0) + 32 * la + 8 * lq;
        const uint32_t pc = p < (int64_t)pos_max ? (uint32_t)p : pos_max;
        return (pc / HT_ZBLK) * blk_bytes + (pc % HT_ZBLK) * 2u;
    };
    ht_u32x4 rw[3][HT_CPW];                                  // the tile's raw bf16 pairs: steps (0,1) (2,3) (4,5) (6,7) of this lane
    uint32_t carry_h[3][HT_CPW];                             // lane 63's last pair of the previous tile (wave-uniform)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int cc = 0; cc < HT_CPW; ++cc) { rw[g][cc] = ht_u4(0u, 0u, 0u, 0u); carry_h[g][cc] = 0u; }
    // (plain loads, visible to the compiler: its s_waitcnt pass waits for them at their first use -- the top of the next tile.  It
    //  does not count the inline-asm stores issued after them, so its vmcnt(n) can only over-wait (n known later operations + the
    //  stores: "at most n outstanding" still means this load has landed).  As inline asm with a hand-counted wait the loop-carried
    //  registers were copied by the register allocator BEFORE the wait statement -- reading registers still in flight.)
    auto issue_loads = [&](const uint32_t voff, const int g) {
        if (SO && g == 0) return;
#pragma unroll
        for (int cc = 0; cc < HT_CPW; ++cc) rw[g][cc] = *(const ht_u32x4*)(zs[g][cc] + voff);
    };
    // the lane that holds the two steps before this lane's first: (la, lq - 1) = lane - 16, or (la - 1, 3) = lane + 47 for lq = 0
    const int hist_src = 4 * (lq > 0 ? lane - 16 : (la > 0 ? lane + 47 : 0));

    // ---- per-lane LDS addresses (staging buffer 0; + HT_STGB for buffer 1)
    const uint32_t stg_wr = HT_OFF_STG + (32 * la + 8 * lq) * 32 + la * 16 + ch0 * 2;
    const uint64_t y64 = (uint64_t)a.y;
    const ht_u32x4 ysrd = {(uint32_t)y64, (uint32_t)(y64 >> 32) & 0xffffu,
                           (uint32_t)((a.y_blk ? (a.y_rows + HT_YBLK - 1) / HT_YBLK * HT_YBLK : (int64_t)(a.B - 1) * a.y_pitch + Ti) * a.y_rowbytes), 0x00020000u};
    const uint32_t yrb = (uint32_t)a.y_rowbytes;

    float carry[HT_CPW][4];                                  // tile-entering state: components 4 lq .. 4 lq + 3, valid in lanes la = 0
    const float first_blk = la == 0 ? 1.f : 0.f;

    // ---- store of tile `cst` from staging buffer `sbuf`: the wave's 64 rows x 32 B, 16 B per lane; the staged outputs of tile
    //      k - 1 go to registers right behind the barrier, the stores themselves are issued between the arithmetic
    struct Vm { bool st; Cur cst; int sbuf; ht_u32x4 sdat[HT_NST]; uint32_t nvoff; };
    auto vm_store_fetch = [&](Vm& v) {
        if (SO) return;
#pragma unroll
        for (int hs = 0; hs < HT_NST; ++hs) {
            const int row = HT_RW * wave + 32 * hs + (lane >> 1);
            v.sdat[hs] = *(const ht_u32x4*)(smem + HT_OFF_STG + v.sbuf * HT_STGB + row * 32 + (row >> 5) * 16 + (lane & 1) * 16);
        }
    };
    auto vm_store = [&](const Vm& v, const int hs) {
        if (SO) return;
        const Cur& c = v.cst;
        const int t0 = c.tile * HT_TT;
        const bool full = t0 + HT_TT <= Ti;
        const uint32_t row0 = (uint32_t)(((int64_t)c.b * a.y_pitch + t0) * a.y_rowbytes + d0 * 2);
        const int row = HT_RW * wave + 32 * hs + (lane >> 1);
        // bounds-checked buffer store: rows past the end (and the stores of the first interval, which has no previous tile) get an
        // offset beyond num_records and are dropped, so that the VM counter sees exactly HT_NST stores per interval.
        // BLOCKED y (round 4): a group's 16 channels of 128 consecutive rows are 4 KiB: a store covers 8 whole lines.
        const uint32_t R = (uint32_t)(a.y_row0 + (int64_t)c.b * a.y_pitch + t0 + row);  // row of the [rows, D] matrix
        const uint32_t yb = ((R / HT_YBLK) * (uint32_t)a.n_groups + (uint32_t)cg) * (HT_YBLK * 32) + (R % HT_YBLK) * 32 + (lane & 1) * 16;
        // (the in-range offset is computed by EVERY lane and pinned before the select: left to itself hipcc wraps the ten address instructions in an
        //  s_and_saveexec / s_or exec region for the lanes in range -- six EXEC writes per tile in the middle of the MFMA stream, each of which
        //  waits for the matrix pipe to drain; round 6)
        uint32_t in_range = a.y_blk ? yb : row0 + (uint32_t)row * yrb + (lane & 1) * 16;
#ifndef HT_OLD_SELECT                   /* A/B knob: the round-5 form (EXEC regions) */
        asm volatile("" : "+v"(in_range));
#endif
        const uint32_t off = (v.st && (full || t0 + row < Ti)) ? in_range : 0xfffffff0u;
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v.sdat[hs]), "v"(off), "s"(ysrd) : "memory");
    };

#if HT_PROFILE      // -DHT_PROFILE=1: every wave accumulates shader-clock deltas per phase and writes 16 floats at y + 64 B * (HT_NW * workgroup + wave)
    //                 (tools/ht_stage_profile.py; a timing build: it overwrites y).  Phases: 0 wait for the tile's loads (+ the loop-carried
    //                 copies), 1 barrier, 5 staged outputs + history, 2 FIR, 3 matrix cores + scan + stores, 4 gate + stage
    uint64_t tprof[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
    const uint64_t rt0 = __builtin_amdgcn_s_memrealtime(), ck0 = __builtin_readcyclecounter();
#define HT_STAMP(K) { const uint64_t now_ = __builtin_readcyclecounter(); tprof[K] += now_ - tlast; tlast = now_; }
#else
#define HT_STAMP(K)
#endif
    // ---- one tile of this wave's channels, from rw (this tile's steps) and hist (the two steps before them)
    auto compute = [&](const Cur& c, int buf, const Vm& vm, const uint32_t (&hist)[3][HT_CPW], auto ragged_t) {
        constexpr bool RAGGED = decltype(ragged_t)::value;
        const int t0 = c.tile * HT_TT;
        const bool last_tile = c.tile == a.n_tiles - 1;
        if (c.tile == 0) {                                   // a new sequence: zero state or the carried one
#pragma unroll
            for (int cc = 0; cc < HT_CPW; ++cc) {
                ht_f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
                if (a.s0) {                                  // (inline asm: a visible load would put s_waitcnt vmcnt(0) into every tile)
                    const float* sp = a.s0 + ((int64_t)c.b * a.D + d0 + ch0 + cc) * 16 + 4 * lq;
                    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(c4) : "v"(sp) : "memory");
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) carry[cc][r] = c4[r];
            }
        }
        const int n_valid = RAGGED ? Ti - (t0 + 32 * la + 8 * lq) : 8;         // steps of this lane inside the sequence
        // value of signal g, step i - 2 (i = 0..9) of both channels as a pair
        auto P = [&](const int g, const int i) -> f32x2_t {
            const int j = i < 2 ? 0 : (i - 2) >> 1;
            const uint32_t w0 = i < 2 ? hist[g][0] : rw[g][0][j];
            const uint32_t w1 = i < 2 ? hist[g][1] : rw[g][1][j];
            f32x2_t r;
            r[0] = (i & 1) ? bf_hi(w0) : bf_lo(w0);
            r[1] = (i & 1) ? bf_hi(w1) : bf_lo(w1);
            return r;
        };

        bf16x8_t xh[HT_CPW];
#if HT_XLO
        bf16x8_t xl[HT_CPW];
#endif
        f32x2_t x2f[8];
        {
            // FIR of x1 and v, x = x1 * v; the lane's eight steps of both channels -> the two channels' B operands
            f32x2_t x[8];
            {
                const f32x2_t w10 = fwl[4], w11 = fwl[5], w12 = fwl[6], b1 = fwl[7];
                const f32x2_t w20 = fwl[8], w21 = fwl[9], w22 = fwl[10], b2 = fwl[11];
                f32x2_t m2a = P(1, 0), m1a = P(1, 1), m2b = P(2, 0), m1b = P(2, 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2_t ca = P(1, i + 2), cb = P(2, i + 2);
                    const f32x2_t x1c = ht_fma(w12, ca, ht_fma(w11, m1a, ht_fma(w10, m2a, b1)));
                    const f32x2_t vc = ht_fma(w22, cb, ht_fma(w21, m1b, ht_fma(w20, m2b, b2)));
                    x[i] = x1c * vc;
                    if (RAGGED && i >= n_valid) { x[i][0] = 0.f; x[i][1] = 0.f; }      // past the end: nothing enters the modes
                    m2a = m1a; m1a = ca; m2b = m1b; m1b = cb;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            issue_loads(vm.nvoff, 1);                        // x1, v of the next tile into the registers just consumed
            issue_loads(vm.nvoff, 2);
            __builtin_amdgcn_sched_barrier(0);
            if (!SO) {
                const f32x2_t w00 = fwl[0], w01 = fwl[1], w02 = fwl[2], b0f = fwl[3];
                f32x2_t m2 = P(0, 0), m1 = P(0, 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2_t cx = P(0, i + 2);
                    x2f[i] = ht_fma(w02, cx, ht_fma(w01, m1, ht_fma(w00, m2, b0f)));
                    m2 = m1; m1 = cx;
                }
                __builtin_amdgcn_sched_barrier(0);
                issue_loads(vm.nvoff, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                uint32_t hw[4];
#if HT_XLO
                uint32_t lw[4];
#endif
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hw[j] = pack_bf2(x[2 * j][e], x[2 * j + 1][e]);
#if HT_XLO
                    lw[j] = pack_bf2(x[2 * j][e] - bf_lo(hw[j]), x[2 * j + 1][e] - bf_hi(hw[j]));
#endif
                }
                xh[e] = __builtin_bit_cast(bf16x8_t, ht_u4(hw[0], hw[1], hw[2], hw[3]));
#if HT_XLO
                xl[e] = __builtin_bit_cast(bf16x8_t, ht_u4(lw[0], lw[1], lw[2], lw[3]));
#endif
            }
        }

        HT_STAMP(2);
        // ---- per channel: E = W . X and y0 = T0 . X on the matrix cores, the block scan, y += G . S
        ht_f32x4 yv[HT_CPW][2];
#pragma unroll
        for (int cc = 0; cc < HT_CPW; ++cc) {
            const uint32_t* t_ = tb[cc];
            // the four scan powers of the channel, requested HERE: the scan's levels are asm blocks the compiler keeps in order, and a
            // load placed next to its level waits out the whole LDS latency four times per channel
            const ht_f32x4* pwc = (const ht_f32x4*)(pwl + (ch0 + cc) * 64) + lq;
            const ht_f32x4 pw4[4] = {pwc[0], pwc[4], pwc[8], pwc[12]};
            __builtin_amdgcn_sched_barrier(0);               // (the loads stay in front of the MFMA burst: its ~250 clocks cover their latency)
#define HT_FRAG(BASE) __builtin_bit_cast(bf16x8_t, ht_u4(t_[(BASE)], t_[(BASE) + 1], t_[(BASE) + 2], t_[(BASE) + 3]))
            const ht_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            ht_f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HT_FRAG(20), xh[cc], zero4, 0, 0, 0);       // W_mid . X_hi
#if HT_XLO
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HT_FRAG(16), xl[cc], e, 0, 0, 0);                    // W_hi . X_lo
#endif
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HT_FRAG(16), xh[cc], e, 0, 0, 0);                    // W_hi . X_hi
            if (!SO)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    ht_f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HT_FRAG(8 * mt + 4), xh[cc], zero4, 0, 0, 0);
#if HT_XLO
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HT_FRAG(8 * mt), xl[cc], acc, 0, 0, 0);
#endif
                    yv[cc][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HT_FRAG(8 * mt), xh[cc], acc, 0, 0, 0);
                }
            HT_FENCE_NOP();
            // Kogge-Stone scan of the 16 block aggregates -> state entering every block; the tile's end state
            float sv[4] = {e[0], e[1], e[2], e[3]};
            {
                const ht_f32x4 Pw = pw4[0];
                const float c0 = first_blk * carry[cc][0], c1 = first_blk * carry[cc][1], c2 = first_blk * carry[cc][2], c3 = first_blk * carry[cc][3];
                sv[0] = fmaf(-Pw[1], c1, fmaf(Pw[0], c0, sv[0]));
                sv[1] = fmaf(Pw[1], c0, fmaf(Pw[0], c1, sv[1]));
                sv[2] = fmaf(-Pw[3], c3, fmaf(Pw[2], c2, sv[2]));
                sv[3] = fmaf(Pw[3], c2, fmaf(Pw[2], c3, sv[3]));
            }
            // One Kogge-Stone level on the lane's two modes, (re, im) += P * (re, im) of the lane SH blocks to the left (0 beyond the
            // row's start): the DPP shift is an operand of the FMA (v_fmac_f32_dpp).  im' is built in a scratch register (the old im is
            // still needed for re'), so the im registers alternate from level to level.  Wait states: a VGPR written by a VALU
            // instruction may be read through DPP two instructions later at the earliest -- inside a level and from level to level
            // the order below keeps that distance (hand-written: the compiler does not look into inline asm); the leading / trailing
            // s_nop cover the compiler's own instructions around the block.  
#define HT_LEVEL(KK, SH, PRE, POST)                                                                                       \
            {                                                                                                             \
                const ht_f32x4 Pw = pw4[(KK)];                                                                            \
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
                             : "v"(sv[1]), "v"(sv[3]), "v"(Pw[0]), "v"(Pw[1]), "v"(Pw[2]), "v"(Pw[3]));                   \
                sv[1] = t1;                                                                                               \
                sv[3] = t3;                                                                                               \
            }
            HT_LEVEL(0, 1, "s_nop 1\n\t", "") HT_LEVEL(1, 2, "", "") HT_LEVEL(2, 4, "", "") HT_LEVEL(3, 8, "", "\n\ts_nop 1")
#undef HT_LEVEL
            float st[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) st[r] = ht_shr<1>(sv[r]) + first_blk * carry[cc][r];                 // state ENTERING block la
#pragma unroll
            for (int r = 0; r < 4; ++r)
                carry[cc][r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv[r]), 0x121, 0xf, 0xf, false));
            if (WS && last_tile && Ti - t0 == HT_TT) {
                // a FULL last tile (T a multiple of 512: the scoring path's main tokens, an 8,192-token prompt): the state after its last
                // step IS the carry the next tile would enter with -- already in the la = 0 lanes, components 4 lq .. 4 lq + 3, the layout
                // of s0 / s_out.  (The general path below walks the last block's 32 steps serially: ~1.6 us per row and channel pair,
                // half a tile's time -- measured as 20 us of a 440 us launch at 8 x 8,192.)
                if (la == 0) {                               // (four dword stores from the carry registers themselves: a 16-byte store wants a copy in
                    //                                              four consecutive registers, and this instantiation has none to spare)
                    const float* so = a.s_out + ((int64_t)c.b * a.D + d0 + ch0 + cc) * 16;      // wave-uniform
                    const uint32_t vo = (uint32_t)lq * 16u;
                    asm volatile("global_store_dword %0, %1, %5\n\tglobal_store_dword %0, %2, %5 offset:4\n\t"
                                 "global_store_dword %0, %3, %5 offset:8\n\tglobal_store_dword %0, %4, %5 offset:12\n\ts_nop 1"
                                 :: "v"(vo), "v"(carry[cc][0]), "v"(carry[cc][1]), "v"(carry[cc][2]), "v"(carry[cc][3]), "s"(so) : "memory");
                }
            } else if (WS && last_tile) {
                // state after the last token T - 1, which sits in block a_ at local step r_ - 1: the recurrence over the block's first
                // r_ steps from the state entering it.  Lane s (< 8) takes mode s.  The x values are the very bf16 terms the matrix
                // cores consumed: the channel's fragments go through the wave's scratch planes (hi | lo, time-contiguous).
                unsigned char* xs = smem + HT_OFF_XS + wave * 2048;
                *(bf16x8_t*)(xs + (4 * la + lq) * 16) = xh[cc];
#if HT_XLO
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
#if HT_XLO
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
                const bf16x8_t sb = __builtin_bit_cast(bf16x8_t, ht_u4(h01, h23, l01, l23));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const uint32_t* g_ = t_ + 24 + 4 * mt;
                    yv[cc][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ht_u4(g_[0], g_[1], g_[0], g_[1])),
                                                                       sb, yv[cc][mt], 0, 0, 0);
                    yv[cc][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ht_u4(g_[2], g_[3], 0u, 0u)),
                                                                       sb, yv[cc][mt], 0, 0, 0);
                }
                HT_FENCE_NOP();
            }
            vm_store(vm, cc);                                // (tile k - 1, staged before the barrier; one 16-byte store per lane)
#undef HT_FRAG
        }
        HT_STAMP(3);
        // ---- gate and stage: accumulator (mt, r) of a lane is its step 4 mt + r; one dword (the wave's two channels) per step
        if (!SO) {
            unsigned char* sp = smem + stg_wr + buf * HT_STGB;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *(uint32_t*)(sp + i * 32) = pack_bf2(yv[0][i >> 2][i & 3] * x2f[i][0], yv[1][i >> 2][i & 3] * x2f[i][1]);
        }
    };

    // ---- the pipeline: one barrier per tile (staging only).  VM queue of a wave per interval, in issue order: the loads of tile
    //      k + 1 (6; 4 in the state-only walk), then the HT_NST stores of tile k - 1.
    Cur c_cmp = {b0, 0}, c_st = {b0, 0};
    if (n_steps > 0) {
        const uint32_t v0 = voff_of(c_cmp);
        issue_loads(v0, 1); issue_loads(v0, 2); issue_loads(v0, 0);
    }
#if HT_PROFILE
    tlast = __builtin_readcyclecounter();
#endif
    for (int k = 0; k <= n_steps; ++k) {
        const int buf = k & 1;
        Cur nx = c_cmp;
        advance(nx);
        HT_STAMP(0);
        __syncthreads();                                     // staging(k - 1) complete; staging(k - 2) read by everybody
        HT_STAMP(1);
        Vm vm;
        vm.st = !SO && k >= 1;
        vm.cst = c_st;
        vm.sbuf = buf ^ 1;
        if (vm.st) advance(c_st);
        vm_store_fetch(vm);
        if (k < n_steps) {
            vm.nvoff = voff_of(k + 1 < n_steps ? nx : c_cmp);    // (no next tile: the same positions again -- constant instruction counts)
            // the two steps before the lane's first, per stream
            uint32_t hist[3][HT_CPW];
            const bool row_start = c_cmp.tile == 0;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int cc = 0; cc < HT_CPW; ++cc) {
                    if (SO && g == 0) { hist[g][cc] = 0u; continue; }
                    uint32_t first = carry_h[g][cc];         // lane 0: lane 63 of the previous tile ...
                    if (row_start) {                         // ... or, at the start of a row, the halo (zeros without one)
                        first = 0u;
                        if (a.z_halo) {
                            const uint16_t* hp = a.z_halo + (int64_t)c_cmp.b * 2 * (3 * a.D) + (h * 384 + g * 128 + cw0 + ch0 + cc);
                            uint32_t lo_, hi_;
                            asm volatile("global_load_ushort %0, %2, off\n\tglobal_load_ushort %1, %3, off\n\ts_waitcnt vmcnt(0)"
                                         : "=&v"(lo_), "=&v"(hi_) : "v"(hp), "v"(hp + 3 * a.D) : "memory");
                            first = lo_ | (hi_ << 16);
                        }
                    }
                    const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute(hist_src, (int)rw[g][cc][3]);
                    hist[g][cc] = lane == 0 ? first : nb;
                    carry_h[g][cc] = (uint32_t)__builtin_amdgcn_readlane((int)rw[g][cc][3], 63);
                }
            HT_STAMP(5);
            if (c_cmp.tile * HT_TT + HT_TT <= Ti) compute(c_cmp, buf, vm, hist, ht_false{});
            else compute(c_cmp, buf, vm, hist, ht_true{});
            c_cmp = nx;
        } else {
#pragma unroll
            for (int hs = 0; hs < HT_NST; ++hs) vm_store(vm, hs);      // the last tile's outputs
        }
        HT_STAMP(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if HT_PROFILE
    if (lane == 0) {
        float* o = (float*)a.y + 16 * (blockIdx.x * HT_NW + wave);
        for (int k = 0; k < 5; ++k) o[k] = (float)tprof[k];
        o[8] = (float)tprof[5];
        o[5] = (float)n_steps;
        o[6] = (float)(__builtin_amdgcn_s_memrealtime() - rt0);      // 100 MHz ticks, whole workgroup
        o[7] = (float)(__builtin_readcyclecounter() - ck0);          // shader clocks, whole workgroup
    }
#endif
#undef HT_STAMP
}

extern "C" int evo_hyena_ct(const void* zt, const void* z_halo, const void* fir_w, const void* fir_b, const void* table, void* y,
                            const float* s0, float* s_out, const float* poles, int64_t B, int64_t T, int64_t D, int64_t n_heads,
                            int64_t zt_pitch, int64_t row_pitch, int64_t zt_row0, int64_t tail_T, int64_t tail_pos0, int64_t state_only,
                            int64_t y_blocked_rows, int64_t y_row0, int64_t y_row_pitch, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || n_heads <= 0 || D != n_heads * 128) return -1;
    const int64_t yrb = D * 2;
    if (y_row_pitch == 0) y_row_pitch = T;
    if (y_row_pitch < T) return -1;
    const int64_t y_span = (B - 1) * y_row_pitch + T;                                // rows of y from batch row 0's first to the last row's last
    if (y_span * yrb >= 0xfffffff0ll) return -1;                                     // 32-bit offsets inside y
    if (y_blocked_rows && (y_row0 < 0 || y_row0 + y_span > y_blocked_rows || (y_blocked_rows + HT_YBLK) * yrb >= 0xfffffff0ll)) return -1;
    // z^T: 16-byte loads -> every batch row starts at a multiple of 8 positions; 32-bit byte offsets inside a column
    if ((!tail_T && row_pitch < T) || row_pitch % 8 != 0 || zt_row0 < 0 || zt_row0 % 8 != 0 || zt_pitch % HT_ZBLK != 0 || zt_pitch < HT_ZBLK) return -1;
    if (zt_pitch * 3 * D * 2 >= 0xfffffff0ll) return -1;                                                    // 32-bit byte offsets inside z^T
    if (tail_T) {            // the last T - tail_T <= 8 tokens of every row in the tail block: the whole ragged last tile
        if (tail_T % HT_TT != 0 || T <= tail_T || T - tail_T > 8 || row_pitch < tail_T || tail_pos0 % 8 != 0) return -1;
        if (zt_row0 + (B - 1) * row_pitch + tail_T > tail_pos0 || tail_pos0 + B * 8 > zt_pitch) return -1;
    } else if (zt_row0 + (B - 1) * row_pitch + T > zt_pitch) return -1;
    if (((uintptr_t)zt & 15) != 0) return -1;
    if (s_out && !poles) return -1;
    if (state_only ? !s_out : !y) return -1;
    const int64_t groups = D / HT_CH;
    // workgroups = groups x nb_split, ~one per CU: a workgroup walks batch rows b0, b0 + nb_split, ... of its group
    int64_t nb_split = (256 + groups - 1) / groups;
    if (nb_split > B) nb_split = B;
    const int64_t streams = groups * nb_split;
    if (streams % 8 != 0 || B * groups > 0x7fffffff) return -1;                      // equal runs of streams per XCD
    HtArgs a;
    a.zt = (const unsigned char*)zt; a.z_halo = (const uint16_t*)z_halo; a.fir_w = (const uint16_t*)fir_w; a.fir_b = (const uint16_t*)fir_b;
    a.tab = (const uint32_t*)table; a.y = (unsigned char*)y; a.s0 = s0; a.s_out = s_out; a.poles = poles;
    a.B = (int)B; a.T = (int)T; a.D = (int)D; a.n_tiles = (int)((T + HT_TT - 1) / HT_TT); a.n_groups = (int)groups;
    a.nb_split = (int)nb_split; a.zt_pitch = zt_pitch; a.row_pitch = row_pitch; a.zt_row0 = zt_row0; a.tail_T = tail_T; a.tail_pos0 = tail_pos0; a.y_rowbytes = yrb;
    a.y_blk = y_blocked_rows ? 1 : 0; a.y_row0 = y_row0; a.y_rows = y_blocked_rows; a.y_pitch = y_row_pitch;
    if (state_only) hipLaunchKernelGGL((hyena_ct_kernel<true, true>), dim3((unsigned)streams), dim3(HT_THREADS), 0, (hipStream_t)stream, a);
    else if (s_out) hipLaunchKernelGGL((hyena_ct_kernel<false, true>), dim3((unsigned)streams), dim3(HT_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((hyena_ct_kernel<false, false>), dim3((unsigned)streams), dim3(HT_THREADS), 0, (hipStream_t)stream, a);
    return evo_launch_status();
}
