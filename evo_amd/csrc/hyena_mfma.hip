// Hyena operator on the matrix cores, SINGLE PASS over z (gfx950).
//
// The two-pass modal recurrence of hyena.hip is bound by fp32 VALU issue (60 v_pk_fma_f32 per step and wave in `apply`,
// 39 in `seg_state`: rocprofv3 shows both at 75-88 % of the packed-FMA pipe rate, profiles/r02_hyena_sq_pmc.txt) and
// reads the x1|v thirds of z twice.  This kernel evaluates the same operator -- FIR(k=3) + bias, x1*v, long convolution
// with h_k = Re sum_s R_s p_s^k, (y + x1v*D)*x2 -- in ONE pass with the heavy arithmetic on MFMA:
//
//   workgroup = (batch rows b0, b0 + nb_split, ..., 16 channels), 8 waves, walks the sequence tile by tile (512 steps = 16
//   blocks of 32), carrying the 16 real modal states of its channels in registers -- sequential in time, parallel over
//   channels, so there is no segment pass and no carry workspace: z is read once, y written once (32,768 B/token/layer).
//   Per tile and channel (constants from evo_amd/hyena_tables.py, math pinned on the CPU by tests/test_hyena_blocked.py):
//     y0 = T0 . X          block Toeplitz (32 x 32 lower triangular, filter.D on the diagonal) x 16 blocks   6 x v_mfma_f32_16x16x32_bf16
//     E  = W  . X          block aggregates: 16 state components x 16 blocks         5 x v_mfma_f32_16x16x32_bf16
//     S  = scan(E)         Kogge-Stone over the 16 blocks with p^32, p^64, p^128, p^256 (DPP row shifts, fp32 VALU)
//     y  = y0 + G . S      contribution of the state entering each block              4 x v_mfma_f32_16x16x32_bf16 (G, S split hi + lo)
//   X = x1*v is split into bf16 hi + lo (2^-17), T0 into 2 and W into 3 bf16 terms, accumulation is fp32: before the one
//   bf16 rounding of the output the result is within 2e-5 of the fp64 oracle (1e-4 at T = 131,073).
//
// z layout: GROUPED -- the projection's output columns are ordered [group][x2 16 | x1 16 | v 16] (hyena_tables.
// group_permutation applied to the rows of the projection weight at load time), so that the 96 bytes a workgroup needs of
// a row are contiguous (with the reference's column order the kernel was bound by the L1's tag rate at 2 TB/s with no
// arithmetic at all: profiles/r02_hyena_mfma_notes.txt).
//
// Three stages per tile, SOFTWARE-PIPELINED over three consecutive tiles, ONE barrier per tile:
//   S1(t)  all threads (channel pair x 8 steps): FIR of x1 and v, x = x1*v, bf16 hi | lo "planes" [channel][time]; x2 rows parked
//   S2(t)  wave = 2 channels: the MFMAs and the scan; leaves (y + x1v D)^T (fp32) in place of its channels' planes
//   S3(t)  all threads: FIR of x2, gate, 16-byte y stores
// In the interval between two barriers a wave runs S3(k-1), S1(k+1) and S2(k) -- waves 0-3 in this order, waves 4-7 with S2
// first, so that on every SIMD one wave is in the MFMA stage while the other runs the VALU stages.  What makes one
// barrier enough:
//   * the z rows a wave consumes are exactly the rows it fetched (64 steps + 2 rows of FIR history, global->LDS DMA): its
//     window is wave-private (single buffer, consumed by S1 and refilled right behind it) -- no barrier, only this wave's
//     counted vmcnt.  S1 also copies the window's x2 dwords into a two-tile ring of the same wave for S3 two intervals
//     later -- fetching x2 a second time when S3 needs it cost +21 % / +63 % L2-miss read traffic at 8 x 8,193 / 131 k
//     (the lines had left the XCD's 4 MiB L2 by then);
//   * the planes are double-buffered, and a thread's plane unit (8 steps: 16 B hi | 16 B lo) is byte for byte the unit of
//     y^T (8 fp32) it reads in S3: S3(k-1) and S1(k+1) touch the same bytes of the same buffer from the same thread, in
//     program order; S2(k) works on the other buffer, on the wave's own two channels.
//   The barrier at the end of interval k publishes planes(k+1) to S2(k+1) and y^T(k) to S3(k).
// The first version ran the three stages strictly one after the other behind three barriers: 9.5-11.4 k cycles per tile, of
// which 2.3-3.3 k barrier skew, against an HBM floor of 6.3 k.
// Entry point and reference citation: include/evo_mi355x.h.
#include "common.h"
#include "../../include/evo_mi355x.h"

#define HM_CH 16                            // channels per workgroup
#define HM_L 32                             // steps per block
#define HM_NB 16                            // blocks per tile
#define HM_TT (HM_L * HM_NB)                // 512 steps per tile
#define HM_ROWB (3 * HM_CH * 2)             // 96 B of a z row per workgroup: x2 | x1 | v of the group, CONTIGUOUS (grouped z layout)
#define HM_WROWS 66                         // rows of a wave's window: its 64 steps + 2 rows of FIR history
// LDS layouts against bank conflicts: a thread of S1 owns 8 consecutive rows, so neighbouring time phases would sit a
// multiple of 128 B apart; a 32-byte gap after every 8 rows (800 B = 200 dwords = 8 mod 32) puts the eight phases of a wave
// on 4 x 2 distinct bank groups.
#define HM_WIN_GROUP (8 * HM_ROWB + 32)     // window: 8 rows of 96 B + gap
#define HM_WIN_WAVE 7168                    // 7 one-KiB DMA pieces (8 groups + 2 rows = 6592 B)
#define HM_WIN_ROW(R) (((R) >> 3) * HM_WIN_GROUP + ((R) & 7) * HM_ROWB)
// parked x2 rows (66 rows of 32 B per wave and tile, later the staged outputs): row r = 8 phase + i sits in slot 8 i + phase
// (rows 64, 65 in slots 64, 65), so that the eight phases of a wave read / write 256 contiguous bytes per access
#define HM_X2P_TILE (HM_WROWS * 32)         // 2,112 B
#define HM_X2P_SLOT(R) ((R) < 64 ? (((R) & 7) * 8 + ((R) >> 3)) : (R))
#define HM_UNIT 32                          // plane unit: 8 steps = [16 B hi | 16 B lo]; later 8 fp32 of (y + x1v D)
#define HM_XTCH (64 * HM_UNIT + 16)         // 2,064 B per channel (odd multiple of 16)
#define HM_OFF_WIN 0
#define HM_OFF_X2P (8 * HM_WIN_WAVE)                        // 57,344
#define HM_OFF_P (HM_OFF_X2P + 8 * 2 * HM_X2P_TILE)         // 91,136
#define HM_OFF_FIR (HM_OFF_P + 2 * HM_CH * HM_XTCH)         // 157,184
#define HM_FIRB (8 * 3 * 4 * 8 + 64)                        // FIR taps + bias of the 8 channel pairs as f32x2 (768 B) + pad
#define HM_OFF_PW (HM_OFF_FIR + HM_FIRB)                    // 158,016
#define HM_PWB (HM_CH * 4 * 16 * 4)                         // p^32, p^64, p^128, p^256 of the 16 channels: [ch][k][16 components] f32, 4 KiB
#define HM_LDS (HM_OFF_PW + HM_PWB)                         // 162,112 B
#define HM_TABW 52
#ifndef HM_PROFILE
#define HM_PROFILE 0
#endif
// Fences around the MFMA bursts of stage 2.  Without them hipcc interleaves the scan's LDS loads and the y^T stores with the
// bursts and pads the MFMA -> consumer distances for an idle matrix pipe (7-8 wait states for these 4-pass MFMAs); with two
// waves per SIMD sharing the pipe, results were read before they were written: run-to-run differing outputs on ~10 % of
// the elements at 8 x 8,193 x 4096 (tools/hm_determinism.py; the variants that pin the schedule are bit-stable over
// hundreds of launches, tests/test_gpu_kernels.py::test_hyena_mfma_is_bit_reproducible).  sched_barrier pins the order,
// the s_nop 7 adds 8 wait states on top of the compiler's own padding.
#define HM_FENCE_NOP() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define HM_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

typedef float hm_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t hm_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2_t hm_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t bf2_f(uint32_t w) { f32x2_t r = {bf_lo(w), bf_hi(w)}; return r; }
__device__ __forceinline__ hm_u32x4 hm_u4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { hm_u32x4 r = {a, b, c, d}; return r; }

struct HmArgs {
    const unsigned char* z; const uint32_t* z_halo; const uint16_t* fir_w; const uint16_t* fir_b; const uint16_t* dskip;
    const uint32_t* tab; uint32_t* y;
    int B; int64_t T; int D; int H; int n_tiles; int n_groups; int nb_split;
};

__device__ __forceinline__ float hm_dpp_shr(float v, const int d) {
    // value of lane (a - d) within the 16-lane row, 0 where a < d
    switch (d) {
        case 1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
        case 2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
        case 4: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
        default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    }
}

__global__ __launch_bounds__(512, 1) void hyena_mfma_kernel(HmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[HM_LDS];      // the only LDS object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- which (batch rows, 16-channel group): block i runs on XCD i % 8; the four groups that share a 128-byte line of
    //      z / y are the four consecutive slots of one XCD.  A workgroup keeps its channel group and walks batch rows b0,
    //      b0 + nb_split, ...: the 13 KiB of MFMA constants per channel are loaded once per workgroup.
    int b0, cg;
    {
        const int bid = blockIdx.x, total = gridDim.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int per_xcd = total >> 3;                     // host guarantees total % 8 == 0
        const int s = xcd * per_xcd + slot;                 // contiguous stream ids per XCD
        b0 = s / a.n_groups;
        cg = s - b0 * a.n_groups;
    }
    const int h = cg >> 3, cw0 = (cg & 7) * HM_CH;          // head, first channel within the head
    const int d0 = h * 128 + cw0;                           // first output channel
    const int64_t rowbytes = (int64_t)a.D * 6;
    unsigned char* win = smem + HM_OFF_WIN + wave * HM_WIN_WAVE;          // this wave's window of z rows
    unsigned char* x2pw = smem + HM_OFF_X2P + wave * (2 * HM_X2P_TILE);   // this wave's parked-x2 ring (two tiles)
    unsigned char* pl = smem + HM_OFF_P;                                  // planes / y^T: [2][16 channels][HM_XTCH]

    // ---- DMA plan: the wave's window = rows (tile start + 64 wave - 2) + 0..65.  Chunk c = 64 i + lane of piece i is 16
    //      bytes of the LDS image (lane-linear); gap and tail chunks re-fetch a valid chunk (never read back).
    int w_off[7], w_row[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int c = 64 * i + lane;
        const int g8 = c / 50, rem = c - 50 * g8;           // 50 chunks per group: 48 of data + 2 of gap
        int row = 8 * g8 + (rem < 48 ? rem / 6 : 7);
        const int col = rem < 48 ? (rem % 6) * 16 : 80;
        if (row > HM_WROWS - 1) row = HM_WROWS - 1;
        w_row[i] = row;
        w_off[i] = row * (int)rowbytes + cg * HM_ROWB + col;                // < 2^31: 66 rows of <= 3 MiB
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int n_rows = (a.B - b0 + a.nb_split - 1) / a.nb_split;
    const int n_steps = n_rows * a.n_tiles;                 // global step = (batch row of this workgroup, tile)
    struct StepInfo { int b; int tile; int64_t t0; };
    auto step_info = [&](int step) {
        StepInfo s;
        const int ri = step / a.n_tiles;
        s.tile = step - ri * a.n_tiles;
        s.b = b0 + ri * a.nb_split;
        s.t0 = (int64_t)s.tile * HM_TT;
        return s;
    };
#define HM_DMA(LDSADDR, SRC)                                                                                  \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(LDSADDR), "v"(SRC) : "memory", "m0")
    auto dma_win = [&](int step) {                          // z rows of `step` -> the wave's window (7 pieces)
        const StepInfo s = step_info(step);
        const unsigned char* zb = a.z + (int64_t)s.b * a.T * rowbytes;
        const int64_t t_first = s.t0 + 64 * wave - 2;
        const bool interior = t_first >= 0 && t_first + HM_WROWS <= a.T;
        const unsigned char* base = zb + t_first * rowbytes;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const unsigned char* src;
            if (interior) src = base + w_off[i];
            else {
                int64_t t = t_first + w_row[i];
                t = t < 0 ? (int64_t)0 : (t > a.T - 1 ? a.T - 1 : t);
                src = zb + t * rowbytes + (w_off[i] - w_row[i] * (int)rowbytes);
            }
            HM_DMA(lds0 + HM_OFF_WIN + wave * HM_WIN_WAVE + i * 1024, src);
        }
    };

    // ---- stage 1 / 3 thread mapping: channel pair p (channels 2p, 2p+1 of the group) x time phase ph (8 steps); the wave's
    //      eight phases phl = 0..7 are its 64 steps
    const int p = tid & 7, ph = tid >> 3, phl = ph & 7;
    // FIR taps / bias live in LDS ([pair][group][tap 0..2, bias] as f32x2) and are read at the head of stages 1 and 3: held in
    // registers they pushed stage 2 over the 256-VGPR budget, and a scratch reload inside the tile loop is a VMEM load whose
    // compiler-placed vmcnt(0) would drain the DMA in flight.
    f32x2_t* firl = (f32x2_t*)(smem + HM_OFF_FIR);
    if (tid < 8 * 3 * 4) {
        const int pp = tid / 12, rem = tid - 12 * pp, g = rem >> 2, k = rem & 3;
        const int c = h * 384 + g * 128 + cw0 + 2 * pp;
        f32x2_t v;
        if (k < 3) { v[0] = bf_to_f(a.fir_w[c * 3 + k]); v[1] = bf_to_f(a.fir_w[(c + 1) * 3 + k]); }
        else { v[0] = bf_to_f(a.fir_b[c]); v[1] = bf_to_f(a.fir_b[c + 1]); }
        firl[tid] = v;
    }
    const f32x2_t* firp = firl + p * 12;                     // this thread's pair: [g][tap 0, 1, 2, bias]

    // ---- stage 2 constants: this wave's two channels (hyena_tables.mfma_operand_table: 52 dwords per lane and channel; the
    //      first 36 -- the MFMA A operands T0, W, G -- stay in registers, the 16 scan powers go to LDS)
    uint32_t tb[2][36];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const uint32_t* tp = a.tab + ((int64_t)(d0 + 2 * wave + cc) * HM_TABW) * 64 + lane;
#pragma unroll
        for (int w = 0; w < 36; ++w) tb[cc][w] = tp[w * 64];
    }
    float* pwl = (float*)(smem + HM_OFF_PW);                 // [ch][k][16] f32
    if (tid < HM_CH * 16) {
        const int c = tid >> 4, m = tid & 15;               // component m = 4 q + r sits in table word 36 + 4 k + r of lanes with q
        const uint32_t* tp = a.tab + ((int64_t)(d0 + c) * HM_TABW) * 64 + (m >> 2) * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) pwl[(c * 4 + k) * 16 + m] = __builtin_bit_cast(float, tp[(36 + 4 * k + (m & 3)) * 64]);
    }
    // (NO compiler-visible VMEM access may sit inside the tile loop: a load gets an s_waitcnt vmcnt(0), which drains the DMA
    //  of the next tile in the middle of the step, and the same goes for scratch reloads -- hence the LDS-resident
    //  constants.  The y stores are inline asm too, and bounds-checked buffer stores so that EVERY S3 issues exactly two.)
    float carry[2][4];                                       // tile-entering state: components 4q..4q+3, valid in lanes a = 0
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int r = 0; r < 4; ++r) carry[cc][r] = 0.f;
    const int la = lane & 15, lq = lane >> 4;
    const float first_blk = la == 0 ? 1.f : 0.f;
    const uint64_t y64 = (uint64_t)a.y;
    const hm_u32x4 ysrd = {(uint32_t)y64, (uint32_t)(y64 >> 32) & 0xffffu, (uint32_t)((int64_t)a.B * a.T * a.D * 2), 0x00020000u};

    // ================= stage 1: FIR (x1, v), x = x1 * v, bf16 hi | lo plane units; the x2 dwords parked for S3 =================
    auto stage1 = [&](int step) {
        const StepInfo s = step_info(step);
        if (s.tile == 0 && wave == 0) {                      // rows -2, -1: the halo (or zeros) instead of the clamped row 0
            if (lane < 48) {
                const int r = lane / 24, wq = lane - 24 * r; // 24 dwords per row: x2 | x1 | v
                uint32_t v = 0u;
                if (a.z_halo) v = a.z_halo[((int64_t)s.b * 2 + r) * (rowbytes / 4) + cg * (HM_ROWB / 4) + wq];
                *(uint32_t*)(win + HM_WIN_ROW(r) + wq * 4) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const f32x2_t w10 = firp[4], w11 = firp[5], w12 = firp[6], b1 = firp[7];
        const f32x2_t w20 = firp[8], w21 = firp[9], w22 = firp[10], b2 = firp[11];
        // window row r <-> step 64 wave - 2 + r of the tile.  This thread reads rows 8 phl .. 8 phl + 9
        const unsigned char* zr = win + p * 4;               // x2 +0, x1 +32, v +64
        unsigned char* xp = x2pw + (step & 1) * HM_X2P_TILE + p * 4;
#define HM_WR(I) HM_WIN_ROW(8 * phl + (I))
        if (phl == 0) {                                      // the wave's two history rows of x2
            *(uint32_t*)(xp + HM_X2P_SLOT(0) * 32) = *(const uint32_t*)(zr + HM_WIN_ROW(0));
            *(uint32_t*)(xp + HM_X2P_SLOT(1) * 32) = *(const uint32_t*)(zr + HM_WIN_ROW(1));
        }
        f32x2_t m2a = bf2_f(*(const uint32_t*)(zr + HM_WR(0) + 32)), m2b = bf2_f(*(const uint32_t*)(zr + HM_WR(0) + 64));
        f32x2_t m1a = bf2_f(*(const uint32_t*)(zr + HM_WR(1) + 32)), m1b = bf2_f(*(const uint32_t*)(zr + HM_WR(1) + 64));
        uint32_t hi8[2][4], lo8[2][4];                       // this thread's 8 steps of both channels, bf16 pairs
        const bool full1 = s.t0 + HM_TT <= a.T;
        const int n_valid = full1 ? 8 : (int)(a.T - s.t0 - 8 * ph);      // steps of this thread inside the sequence
        uint32_t hprev = 0u, lprev = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t cx = *(const uint32_t*)(zr + HM_WR(i + 2));
            const f32x2_t ca = bf2_f(*(const uint32_t*)(zr + HM_WR(i + 2) + 32));
            const f32x2_t cb = bf2_f(*(const uint32_t*)(zr + HM_WR(i + 2) + 64));
            // window row 8 phl + i + 2 -> parked slot: i < 6: 8 (i + 2) + phl; i = 6, 7: row 8 (phl + 1) + (i - 6)
            {
                const int slot = i < 6 ? 8 * (i + 2) + phl : (phl < 7 ? 8 * (i - 6) + phl + 1 : 64 + (i - 6));
                *(uint32_t*)(xp + slot * 32) = cx;
            }
            const f32x2_t x1c = hm_fma(w12, ca, hm_fma(w11, m1a, hm_fma(w10, m2a, b1)));
            const f32x2_t vc = hm_fma(w22, cb, hm_fma(w21, m1b, hm_fma(w20, m2b, b2)));
            f32x2_t x = x1c * vc;
            if (!full1 && i >= n_valid) { x[0] = 0.f; x[1] = 0.f; }      // past the end: nothing enters the modes
            const uint32_t hi = pack_bf2(x[0], x[1]);
            const uint32_t lo = pack_bf2(x[0] - bf_lo(hi), x[1] - bf_hi(hi));
            // transpose the (channel pair) x (8 steps) block in registers: word i/2 of channel e = steps i-1, i of e
            if (i & 1) {                                     // v_perm_b32: bytes of {odd step, even step}
                hi8[0][i >> 1] = __builtin_amdgcn_perm(hi, hprev, 0x05040100u);
                hi8[1][i >> 1] = __builtin_amdgcn_perm(hi, hprev, 0x07060302u);
                lo8[0][i >> 1] = __builtin_amdgcn_perm(lo, lprev, 0x05040100u);
                lo8[1][i >> 1] = __builtin_amdgcn_perm(lo, lprev, 0x07060302u);
            } else {
                hprev = hi;
                lprev = lo;
            }
            m2a = m1a; m1a = ca; m2b = m1b; m1b = cb;
        }
#undef HM_WR
        // unit ph of both channels: [hi | lo]
        unsigned char* x0 = pl + ((step & 1) * HM_CH + 2 * p) * HM_XTCH + ph * HM_UNIT;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            *(hm_u32x4*)(x0 + e * HM_XTCH) = hm_u4(hi8[e][0], hi8[e][1], hi8[e][2], hi8[e][3]);
            *(hm_u32x4*)(x0 + e * HM_XTCH + 16) = hm_u4(lo8[e][0], lo8[e][1], lo8[e][2], lo8[e][3]);
        }
        // every read of the window has returned before the caller refills it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // ================= stage 2: per channel  E = W.X, y0 = T0.X, block scan, y = y0 + G.S =================
    auto stage2 = [&](int step) {
        const StepInfo s = step_info(step);
        if (s.tile == 0) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 4; ++r) carry[cc][r] = 0.f;        // a new sequence starts from a zero state
        }
        // (the two channels one after the other: taking both through the phases together -- two independent scan chains per
        //  phase -- measured 4 % / 20 % SLOWER at 8 x 8,193 / 131 k, profiles/r02_hyena_mfma_notes.txt)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            unsigned char* xc = pl + ((step & 1) * HM_CH + 2 * wave + cc) * HM_XTCH;
            const bf16x8_t xh = *(const bf16x8_t*)(xc + (4 * la + lq) * HM_UNIT);
            const bf16x8_t xl = *(const bf16x8_t*)(xc + (4 * la + lq) * HM_UNIT + 16);
            const uint32_t* t_ = tb[cc];
#define HM_FRAG(BASE) __builtin_bit_cast(bf16x8_t, hm_u4(t_[(BASE)], t_[(BASE) + 1], t_[(BASE) + 2], t_[(BASE) + 3]))
            const hm_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            hm_f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16 + 8), xh, zero4, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16 + 4), xl, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16 + 4), xh, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16), xl, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16), xh, e, 0, 0, 0);
            hm_f32x4 yv[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                hm_f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(8 * mt + 4), xh, zero4, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(8 * mt), xl, acc, 0, 0, 0);
                yv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(8 * mt), xh, acc, 0, 0, 0);
            }
#undef HM_FRAG
            HM_FENCE_NOP();
            float sv[4] = {e[0], e[1], e[2], e[3]};
            const hm_f32x4* pwc = (const hm_f32x4*)(pwl + (2 * wave + cc) * 64) + lq;
            {
                const hm_f32x4 P = pwc[0];
                sv[0] += first_blk * (P[0] * carry[cc][0] - P[1] * carry[cc][1]);
                sv[1] += first_blk * (P[0] * carry[cc][1] + P[1] * carry[cc][0]);
                sv[2] += first_blk * (P[2] * carry[cc][2] - P[3] * carry[cc][3]);
                sv[3] += first_blk * (P[2] * carry[cc][3] + P[3] * carry[cc][2]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const hm_f32x4 P = pwc[4 * k];
                const float u0 = hm_dpp_shr(sv[0], 1 << k), u1 = hm_dpp_shr(sv[1], 1 << k);
                const float u2 = hm_dpp_shr(sv[2], 1 << k), u3 = hm_dpp_shr(sv[3], 1 << k);
                sv[0] += P[0] * u0 - P[1] * u1;
                sv[1] += P[0] * u1 + P[1] * u0;
                sv[2] += P[2] * u2 - P[3] * u3;
                sv[3] += P[2] * u3 + P[3] * u2;
            }
            float st[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) st[r] = hm_dpp_shr(sv[r], 1) + first_blk * carry[cc][r];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                carry[cc][r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv[r]), 0x121, 0xf, 0xf, false));
            {
                const uint32_t h01 = pack_bf2(st[0], st[1]), h23 = pack_bf2(st[2], st[3]);
                const uint32_t l01 = pack_bf2(st[0] - bf_lo(h01), st[1] - bf_hi(h01));
                const uint32_t l23 = pack_bf2(st[2] - bf_lo(h23), st[3] - bf_hi(h23));
                const bf16x8_t sb = __builtin_bit_cast(bf16x8_t, hm_u4(h01, h23, l01, l23));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const uint32_t* g_ = t_ + 28 + 4 * mt;
                    yv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, hm_u4(g_[0], g_[1], g_[0], g_[1])),
                                                                    sb, yv[mt], 0, 0, 0);
                    yv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, hm_u4(g_[2], g_[3], 0u, 0u)),
                                                                    sb, yv[mt], 0, 0, 0);
                }
            }
            HM_FENCE_NOP();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) *(hm_f32x4*)(xc + (4 * la + 2 * mt + (lq >> 1)) * HM_UNIT + (lq & 1) * 16) = yv[mt];
            HM_FENCE();
        }
    };

    // ================= stage 3: FIR (x2), gate, store =================
    auto stage3 = [&](int step) {
        const StepInfo s = step_info(step);
        unsigned char* x2b = x2pw + (step & 1) * HM_X2P_TILE;
        const f32x2_t w00 = firp[0], w01 = firp[1], w02 = firp[2], b0f = firp[3];
        unsigned char* zr = x2b + p * 4;
        // parked row 8 phl + i of this wave (S1 of the same wave put it there two intervals ago; rows 0, 1 are the history)
#define HM_X2R(I) (((I) < 8 ? 8 * (I) + phl : (phl < 7 ? 8 * ((I) - 8) + phl + 1 : 64 + ((I) - 8))) * 32)
        // all ten x2 rows of this thread FIRST: the outputs below are staged in these very rows (rows 8 phl + 8, + 9 are the
        // next phase's first two output rows; one wave's LDS operations execute in order)
        uint32_t xr[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) xr[i] = *(const uint32_t*)(zr + HM_X2R(i));
        // (y + x1v D)^T of this thread's 8 steps: unit ph of both channels
        hm_f32x4 yq[2][2];
        const unsigned char* y0 = pl + ((step & 1) * HM_CH + 2 * p) * HM_XTCH + ph * HM_UNIT;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < 2; ++k) yq[e][k] = *(const hm_f32x4*)(y0 + e * HM_XTCH + k * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]),
                     "+v"(xr[6]), "+v"(xr[7]), "+v"(xr[8]), "+v"(xr[9]) :: "memory");
        f32x2_t m2 = bf2_f(xr[0]), m1 = bf2_f(xr[1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2_t c = bf2_f(xr[i + 2]);
            const f32x2_t x2f = hm_fma(w02, c, hm_fma(w01, m1, hm_fma(w00, m2, b0f)));
            m2 = m1;
            m1 = c;
            const f32x2_t yc = {yq[0][i >> 2][i & 3], yq[1][i >> 2][i & 3]};      // y_conv + x1v * D
            const f32x2_t o = yc * x2f;
            // staged in the x2 slot of the step's own row: the 8 pairs of a wave complete the row's 32 output bytes
            *(uint32_t*)(zr + HM_X2R(i + 2)) = pack_bf2(o[0], o[1]);
        }
#undef HM_X2R
        // the wave's 64 rows x 32 B, 16 B per lane: two 16-byte stores per wave and tile (the y tail of the first version was
        // store-ISSUE bound with eight dword stores).  Same wave wrote the staging rows: LDS executes a wave's operations in
        // order; the fence keeps the differently typed accesses to the same bytes ordered for the compiler.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool full = s.t0 + HM_TT <= a.T;
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            const int rr = hs * 32 + (lane >> 1);            // local step 0..63 of the wave, half (lane & 1)
            const int row = rr + 2;
            const hm_u32x4 v = *(const hm_u32x4*)(x2b + HM_X2P_SLOT(row) * 32 + (lane & 1) * 16);
            const int64_t t = s.t0 + 64 * wave + rr;
            // bounds-checked buffer store: rows past the end of the sequence get an offset beyond num_records and are dropped,
            // so that the VM counter sees exactly two stores per S3
            const uint32_t off = (full || t < a.T) ? (uint32_t)((((int64_t)s.b * a.T + t) * a.D + d0) * 2 + (lane & 1) * 16) : 0xfffffff0u;
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(off), "s"(ysrd) : "memory");
        }
    };

    // ---- the pipeline.  VM queue of a wave per interval, in issue order: 2 y stores (S3), 7 DMA pieces of window(k+2); it
    //      retires in order.  Before S1(k+1): window(k+1), issued last in the previous interval, must have landed -- this
    //      interval's 2 stores may be in flight.  At the head of the stream (no S3 yet) the count does not hold: wait for all.
#if HM_PROFILE      // -DHM_PROFILE=1: wave 0 of workgroup 0 accumulates shader-clock deltas per role (tools/hm_stage_profile.py)
#ifndef HM_PROF_WAVE
#define HM_PROF_WAVE 0
#endif
    const bool prof = blockIdx.x == 0 && wave == HM_PROF_WAVE;
    uint64_t tprof[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#define HM_STAMP(K) if (prof) { const uint64_t now_ = __builtin_readcyclecounter(); tprof[K] += now_ - tlast; tlast = now_; }
#else
#define HM_STAMP(K)
#endif
    dma_win(0);
    __syncthreads();                                         // FIR taps and scan powers are in LDS
#if HM_PROFILE
    if (prof) tlast = __builtin_readcyclecounter();
#endif
#ifndef HM_ORDER
#define HM_ORDER 0                          // 0: waves 4-7 run S2 first (default), 1: no wave does, 2: all do (measurement builds)
#endif
    const bool mfma_first = HM_ORDER == 0 ? wave >= 4 : HM_ORDER == 2;
    for (int k = -1; k <= n_steps; ++k) {
        if (mfma_first && k >= 0 && k < n_steps) { stage2(k); HM_STAMP(3); }
        if (k >= 1) { stage3(k - 1); HM_STAMP(4); }
        if (k + 1 < n_steps) {
            if (k >= 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            HM_STAMP(0);
            stage1(k + 1);
            if (k + 2 < n_steps) dma_win(k + 2);
            HM_STAMP(2);
        }
        if (!mfma_first && k >= 0 && k < n_steps) { stage2(k); HM_STAMP(3); }
        __syncthreads();                                     // planes(k+1) -> S2(k+1), y^T(k) -> S3(k)
        HM_STAMP(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if HM_PROFILE
    if (prof && lane == 0) {                                 // (timing build only: overwrites the first words of y)
        for (int k = 0; k < 5; ++k) ((float*)a.y)[k] = (float)tprof[k];
        ((float*)a.y)[5] = (float)n_steps;
    }
#endif
#undef HM_STAMP
}

extern "C" int evo_hyena_mfma(const void* z, const void* z_halo, const void* fir_w, const void* fir_b, const void* dskip,
                              const void* table, void* y, int64_t B, int64_t T, int64_t D, int64_t n_heads, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || n_heads <= 0 || D != n_heads * 128) return -1;
    if (B * T * D * 2 >= 0xfffffff0ll) return -1;                       // y goes through a 32-bit bounded buffer descriptor
    const int64_t groups = D / HM_CH;
    // workgroups = groups x nb_split, ~one per CU: a workgroup walks batch rows b0, b0 + nb_split, ... of its channels
    int64_t nb_split = (256 + groups - 1) / groups;
    if (nb_split > B) nb_split = B;
    const int64_t streams = groups * nb_split;
    if (streams % 8 != 0 || B * groups > 0x7fffffff) return -1;         // equal runs of streams per XCD
    HmArgs a;
    a.z = (const unsigned char*)z; a.z_halo = (const uint32_t*)z_halo; a.fir_w = (const uint16_t*)fir_w;
    a.fir_b = (const uint16_t*)fir_b; a.dskip = (const uint16_t*)dskip; a.tab = (const uint32_t*)table; a.y = (uint32_t*)y;
    a.B = (int)B; a.T = T; a.D = (int)D; a.H = (int)n_heads; a.n_tiles = (int)((T + HM_TT - 1) / HM_TT); a.n_groups = (int)groups;
    a.nb_split = (int)nb_split;
    hipLaunchKernelGGL(hyena_mfma_kernel, dim3((unsigned)streams), dim3(512), 0, (hipStream_t)stream, a);
    return evo_launch_status();
}
