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
//     y  = y0 + G . S      contribution of the state entering each block              4 x v_mfma_f32_16x16x32_bf16 (G, S split hi + lo)
//   X = x1*v is split into bf16 hi + lo (2^-17), T0 into 2 and W into 3 bf16 terms, accumulation is fp32: before the one
//   bf16 rounding of the output the result is within 2e-5 of the fp64 oracle (1e-4 at T = 131,073).
//
// z layout: GROUPED -- the projection's output columns are ordered [group][x2 16 | x1 16 | v 16] (hyena_tables.
// group_permutation applied to the rows of the projection weight at load time), so that the 96 bytes a workgroup needs of
// a row are contiguous.  With the reference's column order (x2 | x1 | v blocks of 128 per head) they are three 32-byte
// pieces in three cache lines, and the kernel was bound by the L1's tag rate: a 1-KiB LDS-DMA instruction that touches 32
// lines costs ~170 cycles (measured: 2 TB/s with NO arithmetic at identical HBM traffic; TCC requests 3x those of
// hyena_apply -- profiles/r02_hyena_mfma_notes.txt).
// Data path: the tile's rows arrive by global->LDS DMA, one tile ahead, into two alternating buffers, every wait a counted
// vmcnt (the VM counter retires in order: DMA pieces and y stores are counted together); stage 1 (all 512 threads, lanes over channel pairs x time) computes FIR and x1*v and writes
// the bf16 planes TRANSPOSED ([channel][time], what the MFMA B operand wants); stage 2 (wave = 2 channels) runs the
// MFMAs and the scan and leaves (y + x1v D)^T (fp32) in place of its channels' planes; stage 3 (all threads) runs the x2 FIR,
// applies the x2 gate and stores y.  The four workgroups that share a 128-byte line of z (and of y) are numbered onto one XCD.
// Entry point and reference citation: include/evo_mi355x.h.
#include "common.h"
#include "../../include/evo_mi355x.h"

#define HM_CH 16                            // channels per workgroup
#define HM_L 32                             // steps per block
#define HM_NB 16                            // blocks per tile
#define HM_TT (HM_L * HM_NB)                // 512 steps per tile
#define HM_ROWS (HM_TT + 2)                 // + 2 rows of FIR history
#define HM_ROWB (3 * HM_CH * 2)             // 96 B per row: x2 | x1 | v of the group, CONTIGUOUS in the grouped z layout
// LDS layouts are chosen against bank conflicts (the first version spent 80 % of its LDS cycles in conflicts,
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, and was LDS-bound):
//  * z image: row r sits in slot r + r/8 (a gap slot after every 8 rows): the threads of stages 1 / 3 own 8 consecutive rows
//    each, so neighbouring time phases are 9 slots = 216 dwords = 24 (mod 32) banks apart instead of 0;
//  * planes: the 16-byte chunk m (8 steps) of a channel sits at m ^ ((m >> 4) & 3); y^T: chunk k (4 steps) at k ^ ((k >> 3) & 7);
//    the per-channel pitch is an odd number of 16-byte units.
#define HM_SLOTS (HM_ROWS + HM_ROWS / 8)    // 578 slots of 96 B
#define HM_NDMA 55                          // ceil(578 * 96 / 1024) one-KiB DMA pieces per tile
#define HM_ZBUF (HM_NDMA * 1024)            // 56,320 B
#define HM_PLANE (HM_TT * 2 + 16)           // 1,040 B: one bf16 plane of one channel (+ pad; keeps 16-byte alignment)
#define HM_XTCH (2 * HM_PLANE + 16)         // 2,096 B per channel (16 x 131): hi | lo planes, later y^T fp32 [512]
#define HM_FIRB (8 * 3 * 4 * 8 + HM_CH * 4)  // FIR taps + bias of the 8 channel pairs as f32x2 (768 B) + D of the 16 channels
#define HM_PWB (HM_CH * 4 * 16 * 4)          // p^32, p^64, p^128, p^256 of the 16 channels: [ch][k][16 components] f32, 4 KiB
#define HM_LDS (2 * HM_ZBUF + HM_CH * HM_XTCH + HM_FIRB + HM_PWB)   // 151,104 B: [z0][z1][planes][fir][D][powers]
#define HM_SLOT(R) ((R) + ((R) >> 3))       // LDS slot of buffer row R
#define HM_PCH(M) ((M) ^ (((M) >> 4) & 3))  // stored position of plane chunk M (16 B = 8 steps)
#define HM_YCH(K) ((K) ^ (((K) >> 3) & 7))  // stored position of y^T chunk K (16 B = 4 steps)
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
    unsigned char* xt = smem + 2 * HM_ZBUF;

    // ---- DMA plan: piece i (wave, wave + 8, ...) = 1 KiB of the tile buffer; chunk c = 64 i + lane is 16 bytes of row c / 6
    //      (slot 9 m + 8 is a gap: those lanes re-fetch the neighbouring row into it)
    //      Interior tiles (every row of the buffer inside the sequence) address with one precomputed per-lane byte offset per
    //      piece; the first / last tile of a row clamp row by row.
    int dma_off[7], dma_row[7];
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) {
        int c = (wave + 8 * jj) * 64 + lane;
        if (c > HM_SLOTS * 6 - 1) c = HM_SLOTS * 6 - 1;     // tail of the last piece: re-fetch the last chunk (pad space)
        const int slot = c / 6;
        int row = slot - slot / 9;                          // slots 9m .. 9m+7 hold rows 8m .. 8m+7; 9m+8 is the gap
        if (row > HM_ROWS - 1) row = HM_ROWS - 1;
        dma_row[jj] = row;
        dma_off[jj] = row * (int)rowbytes + cg * HM_ROWB + (c - 6 * slot) * 16;      // < 2^31: 514 rows of <= 3 MiB
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int n_rows = (a.B - b0 + a.nb_split - 1) / a.nb_split;
    const int n_steps = n_rows * a.n_tiles;                 // global step = (batch row of this workgroup, tile)
    auto dma_tile = [&](int step) {                         // rows of `step` -> buffer step & 1
        const unsigned char* zb = a.z + (int64_t)(b0 + (step / a.n_tiles) * a.nb_split) * a.T * rowbytes;
        const int64_t t_first = (int64_t)(step % a.n_tiles) * HM_TT - 2;
        const bool interior = t_first >= 0 && t_first + HM_ROWS <= a.T;
        const unsigned char* base = zb + t_first * rowbytes;
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) {
            const int i = wave + 8 * jj;
            if (i < HM_NDMA) {
                const unsigned char* src;
                if (interior) {
                    src = base + dma_off[jj];
                } else {
                    int64_t t = t_first + dma_row[jj];
                    t = t < 0 ? (int64_t)0 : (t > a.T - 1 ? a.T - 1 : t);
                    src = zb + t * rowbytes + (dma_off[jj] - dma_row[jj] * (int)rowbytes);
                }
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             ::"s"(lds0 + (step & 1) * HM_ZBUF + i * 1024), "v"(src) : "memory", "m0");
            }
        }
    };

    // ---- stage 1 / 3 thread mapping: channel pair p (channels 2p, 2p+1 of the group) x 64 time phases of 8 steps
    const int p = tid & 7, ph = tid >> 3;
    // FIR taps / bias live in LDS ([pair][group][tap 0..2, bias] as f32x2) and are read at the head of stages 1 and 3: held in
    // registers they pushed stage 2 over the 256-VGPR budget, and a scratch reload inside the tile loop is a VMEM load whose
    // compiler-placed vmcnt(0) would drain the DMA in flight.
    f32x2_t* firl = (f32x2_t*)(smem + 2 * HM_ZBUF + HM_CH * HM_XTCH);
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
    float* pwl = (float*)(firl + 96) + HM_CH;                                // [ch][k][16] f32
    if (tid < HM_CH * 16) {
        const int c = tid >> 4, m = tid & 15;               // component m = 4 q + r sits in table word 36 + 4 k + r of lanes with q
        const uint32_t* tp = a.tab + ((int64_t)(d0 + c) * HM_TABW) * 64 + (m >> 2) * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) pwl[(c * 4 + k) * 16 + m] = __builtin_bit_cast(float, tp[(36 + 4 * k + (m & 3)) * 64]);
    }
    // (NO compiler-visible VMEM load may sit inside the tile loop: it gets an s_waitcnt vmcnt(0), which drains the DMA of
    //  the next tile in the middle of the step -- an early version lost ~40 % to a dskip load in stage 2; the same goes
    //  for scratch reloads, hence the LDS-resident constants)
    float carry[2][4];                                       // tile-entering state: components 4q..4q+3, valid in lanes a = 0
    const int la = lane & 15, lq = lane >> 4;
    const float first_blk = la == 0 ? 1.f : 0.f;

    // VM-counter bookkeeping.  Issue order per step s:  [top] z(s+1)  ...  [stage 3] 2 y stores.  At the top of step s the
    // tile z(s) (issued at the top of s-1) must have landed: everything but the 2 stores of step s-1 has to retire -- they
    // only count when step s-1 was a full tile (otherwise they are conditional: wait for them too).
#if HM_PROFILE      // -DHM_PROFILE=1: wave 0 of workgroup 0 accumulates shader-clock deltas per stage (tools/hm_stage_profile.py)
    const bool prof = blockIdx.x == 0 && wave == 0;
    uint64_t tprof[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#define HM_STAMP(K) if (prof) { const uint64_t now_ = __builtin_readcyclecounter(); tprof[K] += now_ - tlast; tlast = now_; }
#else
#define HM_STAMP(K)
#endif
    dma_tile(0);
#if HM_PROFILE
    if (prof) tlast = __builtin_readcyclecounter();
#endif
    for (int step = 0; step < n_steps; ++step) {
        const int ri = step / a.n_tiles, tile = step - ri * a.n_tiles;
        const int b = b0 + ri * a.nb_split;
        unsigned char* zt = smem + (step & 1) * HM_ZBUF;
        if (step > 0 && tile != 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HM_STAMP(0);                                         // [0] waiting for the tile's DMA (and old stores)
        __syncthreads();                                     // ... everyone's: the tile is visible, step-1 fully consumed
        HM_STAMP(1);                                         // [1] barrier skew
        if (step + 1 < n_steps) dma_tile(step + 1);          // its buffer was last read by stage 3 of step-1
        if (tile == 0) {                                     // rows -2, -1: the halo (or zeros) instead of the clamped row 0
            if (tid < 2 * 24) {
                const int r = tid / 24, wq = tid - 24 * r;   // 24 dwords per row: x2 | x1 | v
                uint32_t v = 0u;
                if (a.z_halo) v = a.z_halo[((int64_t)b * 2 + r) * (rowbytes / 4) + cg * (HM_ROWB / 4) + wq];
                *(uint32_t*)(zt + r * HM_ROWB + wq * 4) = v;
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 4; ++r) carry[cc][r] = 0.f;        // a new sequence starts from a zero state
            __syncthreads();
        }
        const int64_t t0 = (int64_t)tile * HM_TT;

        // ================= stage 1: FIR (x1, v), x = x1 * v, bf16 hi / lo planes written transposed =================
        {
            const int tl0 = ph * 8;
            const f32x2_t w10 = firp[4], w11 = firp[5], w12 = firp[6], b1 = firp[7];
            const f32x2_t w20 = firp[8], w21 = firp[9], w22 = firp[10], b2 = firp[11];
            // buffer row r <-> local step r - 2.  This thread reads rows 8 ph .. 8 ph + 9: slots 9 ph + {0..7}, 9 ph + {9, 10}
            const unsigned char* zr = zt + (9 * ph) * HM_ROWB + p * 4;  // x2 +0, x1 +32, v +64
#define HM_ROWOFF(I) (((I) < 8 ? (I) : (I) + 1) * HM_ROWB)              /* byte offset of the thread's I-th row, I = 0..9 */
            f32x2_t m2a = bf2_f(*(const uint32_t*)(zr + HM_ROWOFF(0) + 32)), m2b = bf2_f(*(const uint32_t*)(zr + HM_ROWOFF(0) + 64));
            f32x2_t m1a = bf2_f(*(const uint32_t*)(zr + HM_ROWOFF(1) + 32)), m1b = bf2_f(*(const uint32_t*)(zr + HM_ROWOFF(1) + 64));
            uint32_t hi8[2][4], lo8[2][4];                              // this thread's 8 steps of both channels, bf16 pairs
            const bool full1 = t0 + HM_TT <= a.T;
            const int n_valid = full1 ? 8 : (int)(a.T - t0 - tl0);      // steps of this thread inside the sequence
            uint32_t hprev = 0u, lprev = 0u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2_t ca = bf2_f(*(const uint32_t*)(zr + HM_ROWOFF(i + 2) + 32));
                const f32x2_t cb = bf2_f(*(const uint32_t*)(zr + HM_ROWOFF(i + 2) + 64));
                const f32x2_t x1c = hm_fma(w12, ca, hm_fma(w11, m1a, hm_fma(w10, m2a, b1)));
                const f32x2_t vc = hm_fma(w22, cb, hm_fma(w21, m1b, hm_fma(w20, m2b, b2)));
                f32x2_t x = x1c * vc;
                if (!full1 && i >= n_valid) { x[0] = 0.f; x[1] = 0.f; }  // past the end: nothing enters the modes
                const uint32_t hi = pack_bf2(x[0], x[1]);
                const uint32_t lo = pack_bf2(x[0] - bf_lo(hi), x[1] - bf_hi(hi));
                // transpose the (channel pair) x (8 steps) block in registers: word i/2 of channel e = steps i-1, i of e
                if (i & 1) {                                            // v_perm_b32: bytes of {odd step, even step}
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
            // one 16-byte chunk (8 steps) per channel and plane, at its swizzled position
            unsigned char* x0 = xt + (2 * p) * HM_XTCH + HM_PCH(ph) * 16;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                *(hm_u32x4*)(x0 + e * HM_XTCH) = hm_u4(hi8[e][0], hi8[e][1], hi8[e][2], hi8[e][3]);
                *(hm_u32x4*)(x0 + e * HM_XTCH + HM_PLANE) = hm_u4(lo8[e][0], lo8[e][1], lo8[e][2], lo8[e][3]);
            }
        }
        HM_STAMP(2);                                         // [2] DMA issue + stage 1
        __syncthreads();
        HM_STAMP(1);

        // ================= stage 2: per channel  E = W.X, y0 = T0.X, block scan, y = y0 + G.S =================
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            unsigned char* xc = xt + (2 * wave + cc) * HM_XTCH;
            // B operands: lane (a, kg) holds steps 32 a + 8 kg .. + 7 = plane chunk 4 a + kg
            const bf16x8_t xh = *(const bf16x8_t*)(xc + HM_PCH(4 * la + lq) * 16);
            const bf16x8_t xl = *(const bf16x8_t*)(xc + HM_PLANE + HM_PCH(4 * la + lq) * 16);
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
            HM_FENCE_NOP();
            float s[4] = {e[0], e[1], e[2], e[3]};
            const hm_f32x4* pwc = (const hm_f32x4*)(pwl + (2 * wave + cc) * 64) + lq;      // [k] -> + 4 k
            {   // the state entering the tile goes into block 0's aggregate: E[0] += p^32 * carry
                const hm_f32x4 P = pwc[0];
                s[0] += first_blk * (P[0] * carry[cc][0] - P[1] * carry[cc][1]);
                s[1] += first_blk * (P[0] * carry[cc][1] + P[1] * carry[cc][0]);
                s[2] += first_blk * (P[2] * carry[cc][2] - P[3] * carry[cc][3]);
                s[3] += first_blk * (P[2] * carry[cc][3] + P[3] * carry[cc][2]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const hm_f32x4 P = pwc[4 * k];
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
            // next tile's entering state = inclusive value of block 15.  Only lane a = 0 of a row ever uses it (first_blk masks the
            // others): a row rotate by one lane puts block 15's value there -- plain DPP, no LDS crossbar round trip
#pragma unroll
            for (int r = 0; r < 4; ++r)
                carry[cc][r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s[r]), 0x121, 0xf, 0xf, false));
            // y += G . S_start on the bf16 matrix core with both operands split hi + lo (G_hi S_hi + G_hi S_lo + G_lo S_hi,
            // 2^-17): K = 32 = per k-group [S_hi of components 4 kg .. 4 kg + 3 | S_lo of the same]
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
            // (y + x1v D)^T (filter.D sits on T0's diagonal) over this channel's planes: lane (a, q) holds steps
            // 32 a + 16 mt + 4 q + 0..3.  Every plane read of
            // this wave precedes these stores in program order, and one wave's LDS operations execute in order.
            HM_FENCE_NOP();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) *(hm_f32x4*)(xc + HM_YCH(8 * la + 4 * mt + lq) * 16) = yv[mt];
            HM_FENCE();
        }
        HM_STAMP(3);                                         // [3] stage 2
        __syncthreads();
        HM_STAMP(1);

        // ================= stage 3: FIR (x2), gate, store =================
        {
            const f32x2_t w00 = firp[0], w01 = firp[1], w02 = firp[2], b0f = firp[3];
            const unsigned char* zr = zt + (9 * ph) * HM_ROWB + p * 4;
            f32x2_t m2 = bf2_f(*(const uint32_t*)(zr + HM_ROWOFF(0))), m1 = bf2_f(*(const uint32_t*)(zr + HM_ROWOFF(1)));
            // (y + x1v D)^T of this thread's 8 steps: chunks 2 ph, 2 ph + 1 of both channels
            hm_f32x4 yq[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    yq[e][k] = *(const hm_f32x4*)(xt + (2 * p + e) * HM_XTCH + HM_YCH(2 * ph + k) * 16);
            const bool full = t0 + HM_TT <= a.T;                        // (wave-uniform: the usual case is branch-free)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2_t c = bf2_f(*(const uint32_t*)(zr + HM_ROWOFF(i + 2)));
                const f32x2_t x2f = hm_fma(w02, c, hm_fma(w01, m1, hm_fma(w00, m2, b0f)));
                m2 = m1;
                m1 = c;
                const f32x2_t yc = {yq[0][i >> 2][i & 3], yq[1][i >> 2][i & 3]};      // y_conv + x1v * D
                const f32x2_t o = yc * x2f;
                // staged in the x1 slot of the row (dead since stage 1; stage 3 reads only x2 slots): the 8 pairs of a wave
                // complete the row's 32 output bytes
                *(uint32_t*)(const_cast<unsigned char*>(zr) + HM_ROWOFF(i + 2) + 32) = pack_bf2(o[0], o[1]);
            }
            // the wave's 64 rows x 32 B, 16 B per lane: two global_store_dwordx4 per wave instead of eight dword stores
            // (the y tail was store-ISSUE bound).  Same wave wrote the staging slots: LDS executes a wave's operations in order.
            // (compiler fence: the dword stores above and the 16-byte loads below are differently typed accesses to the same
            //  bytes -- without it the loads may be scheduled first)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int wrow0 = (tid >> 6) * 64;                           // first local step of this wave's rows
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                const int rr = hs * 32 + (lane >> 1);                    // local row 0..63 of the wave, half (lane & 1)
                const int r = wrow0 + rr + 2;                            // buffer row
                const hm_u32x4 v = *(const hm_u32x4*)(zt + HM_SLOT(r) * HM_ROWB + 32 + (lane & 1) * 16);
                const int64_t t = t0 + wrow0 + rr;
                if (full || t < a.T)
                    *(hm_u32x4*)((unsigned char*)a.y + (((int64_t)b * a.T + t) * a.D + d0) * 2 + (lane & 1) * 16) = v;
            }
        }
        HM_STAMP(4);                                         // [4] stage 3
    }
#if HM_PROFILE
    if (prof && lane == 0) {                                 // (timing build only: overwrites the first words of y)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = 0; k < 5; ++k) ((float*)a.y)[k] = (float)tprof[k];
        ((float*)a.y)[5] = (float)n_steps;
    }
#endif
#undef HM_STAMP
#undef HM_ROWOFF
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
    HmArgs a;
    a.z = (const unsigned char*)z; a.z_halo = (const uint32_t*)z_halo; a.fir_w = (const uint16_t*)fir_w;
    a.fir_b = (const uint16_t*)fir_b; a.dskip = (const uint16_t*)dskip; a.tab = (const uint32_t*)table; a.y = (uint32_t*)y;
    a.B = (int)B; a.T = T; a.D = (int)D; a.H = (int)n_heads; a.n_tiles = (int)((T + HM_TT - 1) / HM_TT); a.n_groups = (int)groups;
    a.nb_split = (int)nb_split;
    hipLaunchKernelGGL(hyena_mfma_kernel, dim3((unsigned)streams), dim3(512), 0, (hipStream_t)stream, a);
    return evo_launch_status();
}
