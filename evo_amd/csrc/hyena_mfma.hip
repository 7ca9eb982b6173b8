// Hyena operator on the matrix cores, SINGLE PASS over z (gfx950).
//
// The two-pass modal recurrence of hyena.hip is bound by fp32 VALU issue (60 v_pk_fma_f32 per step and wave in `apply`,
// 39 in `seg_state`: rocprofv3 shows both at 75-88 % of the packed-FMA pipe rate, profiles/r02_hyena_sq_pmc.txt) and
// reads the x1|v thirds of z twice.  This kernel evaluates the same operator -- FIR(k=3) + bias, x1*v, long convolution
// with h_k = Re sum_s R_s p_s^k, (y + x1v*D)*x2 -- in ONE pass with the heavy arithmetic on MFMA:
//
//   workgroup = (batch row b, 16 channels), 8 waves, walks the sequence tile by tile (512 steps = 16 blocks of 32),
//   carrying the 16 real modal states of its channels in registers -- sequential in time, parallel over channels, so
//   there is no segment pass and no carry workspace: z is read once, y written once (32,768 B/token/layer).
//   Per tile and channel (constants from evo_amd/hyena_tables.py, math pinned on the CPU by tests/test_hyena_blocked.py):
//     y0 = T0 . X          block Toeplitz (32 x 32 lower triangular) x 16 blocks     6 x v_mfma_f32_16x16x32_bf16
//     E  = W  . X          block aggregates: 16 state components x 16 blocks         5 x v_mfma_f32_16x16x32_bf16
//     S  = scan(E)         Kogge-Stone over the 16 blocks with p^32, p^64, p^128, p^256 (DPP row shifts, fp32 VALU)
//     y  = y0 + G . S      contribution of the state entering each block              8 x v_mfma_f32_16x16x4_f32 (exact fp32)
//   X = x1*v is split into bf16 hi + lo (2^-17), T0 into 2 and W into 3 bf16 terms, accumulation is fp32: before the one
//   bf16 rounding of the output the result is within 2e-5 of the fp64 oracle (1e-4 at T = 131,073).
//
// Data path: the tile's rows (x2|x1|v, 3 x 32 B of each 24 KiB z row) arrive by global->LDS DMA into two rings of two
// buffers: the x1|v thirds (dead after stage 1) are fetched TWO tiles ahead, the x2 third (needed by stage 3) one tile
// ahead, every wait a counted vmcnt (the VM counter retires in order: DMA pieces and y stores are counted together);
// stage 1 (all 512 threads, lanes over channel pairs x time) computes FIR and x1*v and writes
// the bf16 planes TRANSPOSED ([channel][time], what the MFMA B operand wants); stage 2 (wave = 2 channels) runs the
// MFMAs and the scan and leaves (y + x1v D)^T (fp32) in place of its channels' planes; stage 3 (all threads) runs the x2 FIR,
// applies the x2 gate and stores y.  The four workgroups that share a 128-byte line of z (and of y) are numbered onto one XCD.
// Entry point and reference citation: include/evo_mi355x.h.
#include <stdlib.h>
#include "common.h"
#include "../../include/evo_mi355x.h"

#define HM_CH 16                            // channels per workgroup
#define HM_L 32                             // steps per block
#define HM_NB 16                            // blocks per tile
#define HM_TT (HM_L * HM_NB)                // 512 steps per tile
#define HM_ROWS (HM_TT + 2)                 // + 2 rows of FIR history
#define HM_XVROW (2 * HM_CH * 2)            // 64 B per row of the x1|v buffer
#define HM_X2ROW (HM_CH * 2)                // 32 B per row of the x2 buffer
#define HM_NXV 33                           // ceil(514 * 64 / 1024) one-KiB DMA pieces
#define HM_NX2 17                           // ceil(514 * 32 / 1024)
#define HM_XVBUF (HM_NXV * 1024)            // 33,792 B
#define HM_X2BUF (HM_NX2 * 1024)            // 17,408 B
#define HM_PLANE (HM_TT * 2 + 16)           // 1,040 B: one bf16 plane of one channel (+ pad; keeps 16-byte alignment)
#define HM_XTCH (2 * HM_PLANE)              // 2,080 B per channel: hi | lo planes, later y^T fp32 [512]
#define HM_LDS (2 * HM_XVBUF + 2 * HM_X2BUF + HM_CH * HM_XTCH)   // 135,680 B: [xv0][xv1][x20][x21][planes]
#define HM_TABW 52

typedef float hm_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t hm_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2_t hm_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t bf2_f(uint32_t w) { f32x2_t r = {bf_lo(w), bf_hi(w)}; return r; }
__device__ __forceinline__ hm_u32x4 hm_u4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { hm_u32x4 r = {a, b, c, d}; return r; }

struct HmArgs {
    const unsigned char* z; const uint32_t* z_halo; const uint16_t* fir_w; const uint16_t* fir_b; const uint16_t* dskip;
    const uint32_t* tab; uint32_t* y;
    int B; int64_t T; int D; int H; int n_tiles; int n_groups; int nb_split; int dbg;
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
    // ---- which (batch row, 16-channel group): block i runs on XCD i % 8; the four groups that share a 128-byte line of
    //      z / y are the four consecutive slots of one XCD
    //      A workgroup keeps its channel group and walks batch rows b0, b0 + nb_split, ...: the 13 KiB of MFMA constants
    //      per channel are loaded once per workgroup.
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
    unsigned char* xt = smem + 2 * HM_XVBUF + 2 * HM_X2BUF;

    // ---- DMA plan: piece i (wave, wave + 8, ...) = 1 KiB of LDS; chunk c = 64 i + lane is 16 bytes = half of one
    //      (row, group) piece.  x1|v buffer: row = c / 4, chunk j = c % 4 -> group 1 + j/2, half j%2; x2 buffer: row = c / 2.
    int xv_row[5], xv_col[5], x2_row[3], x2_col[3];
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
        int c = (wave + 8 * jj) * 64 + lane;
        if (c > HM_ROWS * 4 - 1) c = HM_ROWS * 4 - 1;       // tail of the last piece: re-fetch the last chunk (pad space)
        const int row = c >> 2, j = c & 3;
        xv_row[jj] = row;
        xv_col[jj] = (h * 384 + (1 + (j >> 1)) * 128 + cw0) * 2 + (j & 1) * 16;
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        int c = (wave + 8 * jj) * 64 + lane;
        if (c > HM_ROWS * 2 - 1) c = HM_ROWS * 2 - 1;
        x2_row[jj] = c >> 1;
        x2_col[jj] = (h * 384 + cw0) * 2 + (c & 1) * 16;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int n_rows = (a.B - b0 + a.nb_split - 1) / a.nb_split;
    const int n_steps = n_rows * a.n_tiles;                 // global step = (batch row of this workgroup, tile)
    auto row_base = [&](int step) {
        const int ri = step / a.n_tiles;
        return a.z + (int64_t)(b0 + ri * a.nb_split) * a.T * rowbytes;
    };
    auto clamp_t = [&](int64_t t) { return t < 0 ? (int64_t)0 : (t > a.T - 1 ? a.T - 1 : t); };
    auto dma_xv = [&](int step) {                           // x1|v thirds of `step` -> xv buffer step & 1
        const unsigned char* zb = row_base(step);
        const int64_t t_first = (int64_t)(step % a.n_tiles) * HM_TT - 2;
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int i = wave + 8 * jj;
            if (i < HM_NXV) {
                const unsigned char* src = zb + clamp_t(t_first + xv_row[jj]) * rowbytes + xv_col[jj];
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             ::"s"(lds0 + (step & 1) * HM_XVBUF + i * 1024), "v"(src) : "memory", "m0");
            }
        }
    };
    auto dma_x2 = [&](int step) {                           // x2 third of `step` -> x2 buffer step & 1
        const unsigned char* zb = row_base(step);
        const int64_t t_first = (int64_t)(step % a.n_tiles) * HM_TT - 2;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int i = wave + 8 * jj;
            if (i < HM_NX2) {
                const unsigned char* src = zb + clamp_t(t_first + x2_row[jj]) * rowbytes + x2_col[jj];
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             ::"s"(lds0 + 2 * HM_XVBUF + (step & 1) * HM_X2BUF + i * 1024), "v"(src) : "memory", "m0");
            }
        }
    };

    // ---- stage 1 / 3 thread mapping: channel pair p (channels 2p, 2p+1 of the group) x 64 time phases of 8 steps
    const int p = tid & 7, ph = tid >> 3;
    f32x2_t fw[3][3], fb[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int c = h * 384 + g * 128 + cw0 + 2 * p;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            f32x2_t v = {bf_to_f(a.fir_w[c * 3 + k]), bf_to_f(a.fir_w[(c + 1) * 3 + k])};
            fw[g][k] = v;
        }
        f32x2_t bb = {bf_to_f(a.fir_b[c]), bf_to_f(a.fir_b[c + 1])};
        fb[g] = bb;
    }

    // ---- stage 2 constants: this wave's two channels (52 dwords per lane each, hyena_tables.mfma_operand_table)
    uint32_t tb[2][HM_TABW];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const uint32_t* tp = a.tab + ((int64_t)(d0 + 2 * wave + cc) * HM_TABW) * 64 + lane;
#pragma unroll
        for (int w = 0; w < HM_TABW; ++w) tb[cc][w] = tp[w * 64];
    }
    float carry[2][4];                                       // tile-entering state: components 4q..4q+3 (same in every lane a)
    const int la = lane & 15, lq = lane >> 4;
    const float first_blk = la == 0 ? 1.f : 0.f;
    const bool no1 = a.dbg & 1, no2 = a.dbg & 2, no3 = a.dbg & 4;       // timing ablations (EVO_HM_DBG; wrong results)

    // VM-counter bookkeeping.  Issue order per step s:  [top] x2(s+1)  ...  [after stage 1] xv(s+2)  ...  [stage 3] 8 y stores.
    // At the top of step s the tile needs xv(s) (issued in step s-2) and x2(s) (issued at the top of step s-1): everything but
    // the youngest  (xv(s+1) pieces of this wave) + (8 stores of step s-1)  must have retired.  The stores only count when
    // step s-1 was a full tile (otherwise they are conditional: wait for them too).
    const int nxv_wave = wave == 0 ? 5 : 4;                  // pieces 0, 8, 16, 24, 32 vs w, w+8, w+16, w+24
    dma_xv(0);
    dma_x2(0);
    if (n_steps > 1) dma_xv(1);
    for (int step = 0; step < n_steps; ++step) {
        const int ri = step / a.n_tiles, tile = step - ri * a.n_tiles;
        const int b = b0 + ri * a.nb_split;
        unsigned char* xv = smem + (step & 1) * HM_XVBUF;
        unsigned char* x2b = smem + 2 * HM_XVBUF + (step & 1) * HM_X2BUF;
        {
            const bool xv_young = step + 1 < n_steps;                              // xv(step+1) is in flight behind what we need
            const bool st_young = step > 0 && tile != 0;                           // step-1 was a full tile of the same row
            const int young = (xv_young ? nxv_wave : 0) + (st_young ? 8 : 0);
            switch (young) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
            }
        }
        __syncthreads();                                     // ... everyone's: the tile is visible, step-1 fully consumed
        if (step + 1 < n_steps) dma_x2(step + 1);            // its buffer was last read by stage 3 of step-1
        if (tile == 0) {                                     // rows -2, -1: the halo (or zeros) instead of the clamped row 0
            if (tid < 2 * 24) {
                const int r = tid / 24, wq = tid - 24 * r;   // 24 dwords per row: x2 | x1 | v
                const int g = wq >> 3, wd = wq & 7;
                uint32_t v = 0u;
                if (a.z_halo) v = a.z_halo[((int64_t)b * 2 + r) * (rowbytes / 4) + (h * 384 + g * 128 + cw0) / 2 + wd];
                if (g == 0) *(uint32_t*)(x2b + r * HM_X2ROW + wd * 4) = v;
                else *(uint32_t*)(xv + r * HM_XVROW + (g - 1) * 32 + wd * 4) = v;
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 4; ++r) carry[cc][r] = 0.f;        // a new sequence starts from a zero state
            __syncthreads();
        }
        const int64_t t0 = (int64_t)tile * HM_TT;

        // ================= stage 1: FIR (x1, v), x = x1 * v, bf16 hi / lo planes written transposed =================
        if (!no1) {
            const int tl0 = ph * 8;
            const unsigned char* zr = xv + tl0 * HM_XVROW + p * 4;      // buffer row r <-> local step r - 2; x1 at +0, v at +32
            f32x2_t m2a = bf2_f(*(const uint32_t*)zr), m2b = bf2_f(*(const uint32_t*)(zr + 32));
            f32x2_t m1a = bf2_f(*(const uint32_t*)(zr + HM_XVROW)), m1b = bf2_f(*(const uint32_t*)(zr + HM_XVROW + 32));
            unsigned char* x0 = xt + (2 * p) * HM_XTCH + tl0 * 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2_t ca = bf2_f(*(const uint32_t*)(zr + (i + 2) * HM_XVROW));
                const f32x2_t cb = bf2_f(*(const uint32_t*)(zr + (i + 2) * HM_XVROW + 32));
                const f32x2_t x1c = hm_fma(fw[1][2], ca, hm_fma(fw[1][1], m1a, hm_fma(fw[1][0], m2a, fb[1])));
                const f32x2_t vc = hm_fma(fw[2][2], cb, hm_fma(fw[2][1], m1b, hm_fma(fw[2][0], m2b, fb[2])));
                f32x2_t x = x1c * vc;
                if (t0 + tl0 + i >= a.T) { x[0] = 0.f; x[1] = 0.f; }     // past the end: nothing enters the modes
                const uint32_t hi = pack_bf2(x[0], x[1]);
                const uint32_t lo = pack_bf2(x[0] - bf_lo(hi), x[1] - bf_hi(hi));
                *(uint16_t*)(x0 + i * 2) = (uint16_t)(hi & 0xffffu);
                *(uint16_t*)(x0 + HM_PLANE + i * 2) = (uint16_t)(lo & 0xffffu);
                *(uint16_t*)(x0 + HM_XTCH + i * 2) = (uint16_t)(hi >> 16);
                *(uint16_t*)(x0 + HM_XTCH + HM_PLANE + i * 2) = (uint16_t)(lo >> 16);
                m2a = m1a; m1a = ca; m2b = m1b; m1b = cb;
            }
        }
        __syncthreads();
        if (step + 2 < n_steps) dma_xv(step + 2);            // this step's x1|v buffer is dead from here on

        // ================= stage 2: per channel  E = W.X, y0 = T0.X, block scan, y = y0 + G.S =================
        if (!no2)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            unsigned char* xc = xt + (2 * wave + cc) * HM_XTCH;
            const bf16x8_t xh = *(const bf16x8_t*)(xc + (HM_L * la + 8 * lq) * 2);
            const bf16x8_t xl = *(const bf16x8_t*)(xc + HM_PLANE + (HM_L * la + 8 * lq) * 2);
            const uint32_t* t_ = tb[cc];
#define HM_FRAG(BASE) __builtin_bit_cast(bf16x8_t, hm_u4(t_[(BASE)], t_[(BASE) + 1], t_[(BASE) + 2], t_[(BASE) + 3]))
            const hm_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            // block aggregates (W = hi + mid + lo, X = hi + lo; the lo*lo term is below 2^-33)
            hm_f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16 + 8), xh, zero4, 0, 0, 0);      // W_lo  . X_hi
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16 + 4), xl, e, 0, 0, 0);                   // W_mid . X_lo
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16 + 4), xh, e, 0, 0, 0);                   // W_mid . X_hi
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16), xl, e, 0, 0, 0);                       // W_hi  . X_lo
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(16), xh, e, 0, 0, 0);                       // W_hi  . X_hi
            // block Toeplitz, two 16-row tiles (T0 = hi + lo)
            hm_f32x4 yv[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                hm_f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(8 * mt + 4), xh, zero4, 0, 0, 0);   // T0_lo . X_hi
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(8 * mt), xl, acc, 0, 0, 0);                  // T0_hi . X_lo
                yv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HM_FRAG(8 * mt), xh, acc, 0, 0, 0);               // T0_hi . X_hi
            }
#undef HM_FRAG
            // scan over the 16 blocks (lanes a = lane & 15 of each 16-lane row; this lane: modes 2q, 2q+1 as re, im, re, im)
            float s[4] = {e[0], e[1], e[2], e[3]};
#define HM_F(IDX) __builtin_bit_cast(float, t_[(IDX)])
            {   // the state entering the tile goes into block 0's aggregate: E[0] += p^32 * carry
                const float P[4] = {HM_F(28), HM_F(29), HM_F(30), HM_F(31)};
                s[0] += first_blk * (P[0] * carry[cc][0] - P[1] * carry[cc][1]);
                s[1] += first_blk * (P[0] * carry[cc][1] + P[1] * carry[cc][0]);
                s[2] += first_blk * (P[2] * carry[cc][2] - P[3] * carry[cc][3]);
                s[3] += first_blk * (P[2] * carry[cc][3] + P[3] * carry[cc][2]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float P[4] = {HM_F(28 + 4 * k), HM_F(29 + 4 * k), HM_F(30 + 4 * k), HM_F(31 + 4 * k)};
                const float u0 = hm_dpp_shr(s[0], 1 << k), u1 = hm_dpp_shr(s[1], 1 << k);
                const float u2 = hm_dpp_shr(s[2], 1 << k), u3 = hm_dpp_shr(s[3], 1 << k);
                s[0] += P[0] * u0 - P[1] * u1;
                s[1] += P[0] * u1 + P[1] * u0;
                s[2] += P[2] * u2 - P[3] * u3;
                s[3] += P[2] * u3 + P[3] * u2;
            }
            // state ENTERING each block: the inclusive scan shifted by one block, the tile's entering state in block 0
            float st[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) st[r] = hm_dpp_shr(s[r], 1) + first_blk * carry[cc][r];
            // next tile's entering state = inclusive value of block 15, for every lane of the row
#pragma unroll
            for (int r = 0; r < 4; ++r) carry[cc][r] = __shfl(s[r], (lane & 48) | 15, 64);
            // y += G . S_start on the fp32 matrix core (k order: virtual component 4 k + ks, both operands alike)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    yv[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(HM_F(44 + 4 * mt + ks), st[ks], yv[mt], 0, 0, 0);
#undef HM_F
            // + x * D (the skip term) while x is at hand in this layout: x = hi + lo of the planes (2^-17), so that stage 3
            // needs neither the x1 / v thirds nor their FIR again
            const float dkc = bf_to_f(a.dskip[d0 + 2 * wave + cc]);
            uint2 xh4[2], xl4[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                xh4[mt] = *(const uint2*)(xc + (HM_L * la + 16 * mt + 4 * lq) * 2);
                xl4[mt] = *(const uint2*)(xc + HM_PLANE + (HM_L * la + 16 * mt + 4 * lq) * 2);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                yv[mt][0] = fmaf(bf_lo(xh4[mt].x) + bf_lo(xl4[mt].x), dkc, yv[mt][0]);
                yv[mt][1] = fmaf(bf_hi(xh4[mt].x) + bf_hi(xl4[mt].x), dkc, yv[mt][1]);
                yv[mt][2] = fmaf(bf_lo(xh4[mt].y) + bf_lo(xl4[mt].y), dkc, yv[mt][2]);
                yv[mt][3] = fmaf(bf_hi(xh4[mt].y) + bf_hi(xl4[mt].y), dkc, yv[mt][3]);
            }
            // (y + x D)^T over this channel's planes: lane (a, q) holds steps 32 a + 16 mt + 4 q + 0..3.  Every plane read of
            // this wave precedes these stores in program order, and one wave's LDS operations execute in order.
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) *(hm_f32x4*)(xc + (HM_L * la + 16 * mt + 4 * lq) * 4) = yv[mt];
        }
        __syncthreads();

        // ================= stage 3: FIR (x2), gate, store =================
        if (!no3) {
            const int tl0 = ph * 8;
            const unsigned char* zr = x2b + tl0 * HM_X2ROW + p * 4;
            f32x2_t m2 = bf2_f(*(const uint32_t*)zr), m1 = bf2_f(*(const uint32_t*)(zr + HM_X2ROW));
            const float* y0p = (const float*)(xt + (2 * p) * HM_XTCH) + tl0;
            const float* y1p = (const float*)(xt + (2 * p + 1) * HM_XTCH) + tl0;
            uint32_t* yo = a.y + (((int64_t)b * a.T + t0 + tl0) * a.D + d0) / 2 + p;
            const bool full = t0 + HM_TT <= a.T;                        // (wave-uniform: the usual case is branch-free)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2_t c = bf2_f(*(const uint32_t*)(zr + (i + 2) * HM_X2ROW));
                const f32x2_t x2f = hm_fma(fw[0][2], c, hm_fma(fw[0][1], m1, hm_fma(fw[0][0], m2, fb[0])));
                m2 = m1;
                m1 = c;
                const f32x2_t yc = {y0p[i], y1p[i]};                    // y_conv + x1v * D
                const f32x2_t o = yc * x2f;
                if (full || t0 + tl0 + i < a.T) yo[(int64_t)i * (a.D / 2)] = pack_bf2(o[0], o[1]);
            }
        }
    }
}

extern "C" int evo_hyena_mfma(const void* z, const void* z_halo, const void* fir_w, const void* fir_b, const void* dskip,
                              const void* table, void* y, int64_t B, int64_t T, int64_t D, int64_t n_heads, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || n_heads <= 0 || D != n_heads * 128) return -1;
    const int64_t groups = D / HM_CH;
    // workgroups = groups x nb_split, ~one per CU: a workgroup walks batch rows b0, b0 + nb_split, ... of its channels
    int64_t nb_split = (256 + groups - 1) / groups;
    if (nb_split > B) nb_split = B;
    const int64_t streams = groups * nb_split;
    if (streams % 8 != 0 || B * groups > 0x7fffffff) return -1;         // equal runs of streams per XCD
    static const int dbg = [] { const char* e = getenv("EVO_HM_DBG"); return e ? atoi(e) : 0; }();
    HmArgs a;
    a.z = (const unsigned char*)z; a.z_halo = (const uint32_t*)z_halo; a.fir_w = (const uint16_t*)fir_w;
    a.fir_b = (const uint16_t*)fir_b; a.dskip = (const uint16_t*)dskip; a.tab = (const uint32_t*)table; a.y = (uint32_t*)y;
    a.B = (int)B; a.T = T; a.D = (int)D; a.H = (int)n_heads; a.n_tiles = (int)((T + HM_TT - 1) / HM_TT); a.n_groups = (int)groups;
    a.nb_split = (int)nb_split; a.dbg = dbg;
    hipLaunchKernelGGL(hyena_mfma_kernel, dim3((unsigned)streams), dim3(512), 0, (hipStream_t)stream, a);
    return evo_launch_status();
}
