// Dense layer on the MFMA pipe, hand-written for gfx950:  Y[M,N] = X[M,K] . W[N,K]^T (+ bias[N]) (+ R[M,N])
// bf16 in, fp32 accumulate, one bf16 rounding.  Both operands are K-contiguous (activations row-major, nn.Linear
// weights [out, in]), so both tiles are plain row slabs.
//
// Workgroup = 8 waves = one 256(M) x 256(N) output tile, K walked in steps of 64; wave (wm, wn) of a 2 x 4 grid
// owns 128(M) x 64(N) = 4 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16, computed TRANSPOSED (A operand = W rows,
// B operand = X rows) so that a lane's accumulator registers run along N: the epilogue adds bias / residual and
// stores 4 consecutive bf16 (8 bytes) per register group instead of scattering single elements.
//
// Data path.  Operand slabs (256 rows x 128 B = 32 KiB each, one cache line per row) arrive by asynchronous
// global->LDS DMA (global_load_lds_dwordx4) into a ring of FIVE slab slots = all 160 KiB of the CU's LDS; the slab
// sequence is X0 W0 X1 W1 ...  Two slabs are being read, three are in flight.  One raw s_barrier per k-step sits
// between its 3rd and 4th 16-deep sub-step; fragments are double-buffered in registers, so the first fragments of
// the next stage are read right behind the barrier, under the last sub-step of the current one.  The DMA
// instructions are spread between the MFMAs (one per two MFMAs): a burst of 64 KiB right behind the barrier blocks
// every wave on the vector-memory issue queue for ~750 cycles per k-step (measured, independent of data latency).
// LDS rows are XOR-swizzled (granule g of row r at slot g ^ ((r >> 1) & 7)): ds_read_b128 fragment reads are
// bank-conflict free (SQ_LDS_BANK_CONFLICT = 0) and the DMA stays lane-linear on the LDS side.
//
// Everything the compiler would otherwise serialise is inline asm with hand-counted waits: it answers any visible
// LDS-DMA with s_waitcnt vmcnt(0) before the next LDS read, and puts an lgkmcnt wait in front of every other MFMA.
// Workgroups are numbered so that each XCD (private L2) sweeps a contiguous,
// group_m-rastered range of output tiles.
// Entry point and reference citation: include/evo_mi355x.h.
#include <stdlib.h>
#include "common.h"
#include "../../include/evo_mi355x.h"

#define GBM 256
#define GBN 256
#define GBK 64
#define G_ROW (GBK * 2)                      // 128 B per LDS row = 8 granules of 16 B
#define G_SLAB (GBM * G_ROW)                 // 32,768 B = 32 DMA pieces of 1 KiB -> 4 per wave
#define G_NSLOT 5                            // 163,840 B
#define G_NP 4                               // DMA pieces per wave per slab

typedef uint32_t g_u32x4 __attribute__((ext_vector_type(4)));

#define G_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define G_BARRIER() asm volatile("s_barrier" ::: "memory")

struct GemmArgs {
    const unsigned char* x; const unsigned char* w; const uint16_t* bias; const uint16_t* res; uint16_t* y;
    int64_t M; int N; int K;
    int tiles_n; int n_tiles; int tiles_m; int group_m;
    // RMSNorm folded into the dense layers around it (round 5; persistent kernel only, see gemmr_bf16_kernel's NF parameter)
    const float* rs = nullptr;               // SCALE: 1 / (rms + eps) per token row, applied to the accumulators in the epilogue
    float* ss = nullptr;                     // STATS: partial sums of squares of the stored rows, [N / 128 strips][ss_ld] fp32
    int64_t ss_ld = 0;
    int64_t w_rows = 0;                      // MODE 3 + SCALE: rows of the token matrix (the kernel's "W" operand): B T
    int tm_tiles = 0, row_skip = 0;          // MODE 3 + SCALE: a batch row = tm_tiles column tiles; its tokens start row_skip rows further per batch row
};

template <bool BIAS, bool RES>
__global__ __launch_bounds__(512, 1) void gemm_bf16_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[G_NSLOT * G_SLAB];   // the ONLY __shared__ object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;                     // 2 x 4 wave grid

    // XCD-aware tile id: block b runs on XCD b % 8; give every XCD a contiguous run of tiles (bijective for any count)
    int tile;
    {
        const int bid = blockIdx.x, nt = a.n_tiles;
        const int q = nt >> 3, r = nt & 7, xcd = bid & 7, slot = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    // grouped raster: consecutive tile ids walk group_m row tiles before the next column tile, so the ~32 tiles an XCD
    // runs at once share group_m X panels and 32/group_m W panels in its L2 instead of 1 + 32
    int tm, tn;
    {
        const int per_group = a.group_m * a.tiles_n;
        const int grp = tile / per_group, in_grp = tile - grp * per_group;
        const int first_m = grp * a.group_m;
        const int gsz = a.tiles_m - first_m < a.group_m ? a.tiles_m - first_m : a.group_m;
        tn = in_grp / gsz;
        tm = first_m + in_grp - tn * gsz;
    }
    const int64_t m0 = (int64_t)tm * GBM;
    const int n0 = tn * GBN;
    const int64_t kb = (int64_t)a.K * 2;                         // bytes per operand row
    const int nk = a.K / GBK;

    // ---- DMA plan: a slab is 32 one-KiB pieces (8 rows each), 4 per wave.  The LDS side of a DMA is lane-linear, so
    //      the swizzle is applied to the global source: the lane that fills slot s of row r fetches granule
    //      s ^ ((r >> 1) & 7) -- still the same 128-byte line.
    const unsigned char* srcx[G_NP];
    const unsigned char* srcw[G_NP];
#pragma unroll
    for (int jj = 0; jj < G_NP; ++jj) {
        const int pos = (wave + 8 * jj) * 1024 + 16 * lane;
        const int r = pos / G_ROW;
        const int col = ((((pos - r * G_ROW) >> 4) ^ (r >> 1)) & 7) * 16;
        int64_t m = m0 + r;
        if (m > a.M - 1) m = a.M - 1;                            // ragged last M tile: clamp (rows never stored)
        srcx[jj] = a.x + m * kb + col;
        srcw[jj] = a.w + (int64_t)(n0 + r) * kb + col;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t lds_dma = lds0 + wave * 1024;
    // one piece of slab `H` (even: X of stage H/2, odd: W of stage H/2) -> slot H % 5.  Past the last stage the same
    // instruction re-fetches the last stage into a slot nobody reads: the loop stays branch-free and the vmcnt counts
    // stay constant.
#define G_DMA(SRC, H, JJ)                                                                                     \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"                             \
                 ::"s"(lds_dma + ((H) % G_NSLOT) * G_SLAB + (JJ) * 8192),                                     \
                   "v"((SRC)[JJ] + (int64_t)(((H) >> 1) < nk ? ((H) >> 1) : nk - 1) * (GBK * 2)) : "memory", "m0")

    // fragment bases (within a slab): B operand = X rows (m), A operand = W rows (n).  Rows 32 apart share
    // (row >> 1) & 7, so one base per k-substep serves all of a wave's tiles (tile offsets are immediates).
    uint32_t x_rd[4], w_rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int xr = wm * 128 + l31, wr = wn * 64 + l31;
        x_rd[ks] = lds0 + (uint32_t)(xr * G_ROW + (((2 * ks + half) ^ (xr >> 1)) & 7) * 16);
        w_rd[ks] = lds0 + (uint32_t)(wr * G_ROW + (((2 * ks + half) ^ (wr >> 1)) & 7) * 16);
    }

    f32x16_t acc[2][4];                                          // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    g_u32x4 wf[2][2], xf[2][4];                                  // double-buffered fragments

#define G_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
    // fragments of sub-step KS of the stage whose X / W slabs sit at byte offsets XO / WO
#define G_READ(XO, WO, KS, BUF)                                                                               \
    {                                                                                                         \
        const uint32_t wa_ = (WO) + w_rd[KS], xa_ = (XO) + x_rd[KS];                                          \
        G_DSR(wf[BUF][0], wa_, 0); G_DSR(wf[BUF][1], wa_, 32 * G_ROW);                                        \
        G_DSR(xf[BUF][0], xa_, 0); G_DSR(xf[BUF][1], xa_, 32 * G_ROW);                                        \
        G_DSR(xf[BUF][2], xa_, 64 * G_ROW); G_DSR(xf[BUF][3], xa_, 96 * G_ROW);                               \
    }
    // one counted wait per sub-step, tied to the fragment registers it guards
#define G_LGKM(N, BUF)                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(wf[BUF][0]), "+v"(wf[BUF][1]), "+v"(xf[BUF][0]), "+v"(xf[BUF][1]), \
                 "+v"(xf[BUF][2]), "+v"(xf[BUF][3]) : "n"(N))
    // (the MFMA itself stays a builtin: the compiler must see it to place the MFMA -> accumulator-read wait states)
#define G_MFMA(I, J, BUF)                                                                                     \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[BUF][I]),             \
                                                        __builtin_bit_cast(bf16x8_t, xf[BUF][J]), acc[I][J], 0, 0, 0)
#define G_DMA_PIN(SRC, H, JJ)                                                                                 \
    {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        G_DMA(SRC, H, JJ);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    // eight MFMAs of one sub-step with up to four DMA pieces between them (piece index < 0: none)
#define G_MMA_DMA(BUF, SRC, H, JA, JB, JC, JD)                                                                \
    {                                                                                                         \
        G_MFMA(0, 0, BUF); G_MFMA(0, 1, BUF);                                                                 \
        if ((JA) >= 0) G_DMA_PIN(SRC, H, (JA) < 0 ? 0 : (JA));                                                    \
        G_MFMA(0, 2, BUF); G_MFMA(0, 3, BUF);                                                                 \
        if ((JB) >= 0) G_DMA_PIN(SRC, H, (JB) < 0 ? 0 : (JB));                                                    \
        G_MFMA(1, 0, BUF); G_MFMA(1, 1, BUF);                                                                 \
        if ((JC) >= 0) G_DMA_PIN(SRC, H, (JC) < 0 ? 0 : (JC));                                                    \
        G_MFMA(1, 2, BUF); G_MFMA(1, 3, BUF);                                                                 \
        if ((JD) >= 0) G_DMA_PIN(SRC, H, (JD) < 0 ? 0 : (JD));                                                    \
    }

    // ---- pipeline.  Slab h (even: X of stage h/2, odd: W) lives in slot h % 5.  Iteration k:
    //        sub-steps 0,1   MFMAs(k,0..1)  + DMA X(k+2) (slab 2k+4: its slot was freed by barrier k-1)
    //        sub-step  2     MFMAs(k,2)
    //        wait            own pieces of stage k+1 (slabs 2k+2, 2k+3) retired: vmcnt(4) leaves X(k+2) in flight   [RAW]
    //                        lgkmcnt(0): this wave's last reads of stage k retired                                  [WAR]
    //        barrier k       stage k+1 visible to all; slots of slabs 2k, 2k+1 free
    //        sub-step  3     first fragments of stage k+1 are read; MFMAs(k,3) + DMA W(k+2) (slab 2k+5 -> slot of X(k))
    //      A slab is re-filled only behind the barrier that follows the last read of its previous contents, and read
    //      only behind the barrier that follows its own counted vmcnt.
#pragma unroll
    for (int jj = 0; jj < G_NP; ++jj) G_DMA(srcx, 0, jj);
#pragma unroll
    for (int jj = 0; jj < G_NP; ++jj) G_DMA(srcw, 1, jj);
    if (nk > 1) {
#pragma unroll
        for (int jj = 0; jj < G_NP; ++jj) G_DMA(srcx, 2, jj);
#pragma unroll
        for (int jj = 0; jj < G_NP; ++jj) G_DMA(srcw, 3, jj);
        G_VMCNT(2 * G_NP);
    } else {
        G_VMCNT(0);
    }
    G_BARRIER();
    G_READ(0, G_SLAB, 0, 0);
    for (int k = 0; k + 1 < nk; ++k) {                           // (the last k-step is peeled)
        const uint32_t xo = ((2 * k) % G_NSLOT) * G_SLAB, wo = ((2 * k + 1) % G_NSLOT) * G_SLAB;
        G_READ(xo, wo, 1, 1);
        G_LGKM(6, 0);
        __builtin_amdgcn_sched_barrier(0);
        G_MMA_DMA(0, srcx, 2 * k + 4, -1, 0, -1, 1);
        __builtin_amdgcn_sched_barrier(0);
        G_READ(xo, wo, 2, 0);
        G_LGKM(6, 1);
        __builtin_amdgcn_sched_barrier(0);
        G_MMA_DMA(1, srcx, 2 * k + 4, -1, 2, -1, 3);
        __builtin_amdgcn_sched_barrier(0);
        G_READ(xo, wo, 3, 1);
        G_LGKM(6, 0);
        __builtin_amdgcn_sched_barrier(0);
        G_MMA_DMA(0, srcx, 0, -1, -1, -1, -1);
        __builtin_amdgcn_sched_barrier(0);
        G_LGKM(0, 1);                                            // sub-step 3's fragments: stage k is fully read
        G_VMCNT(G_NP);
        G_BARRIER();
        G_READ(((2 * k + 2) % G_NSLOT) * G_SLAB, ((2 * k + 3) % G_NSLOT) * G_SLAB, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        G_MMA_DMA(1, srcw, 2 * k + 5, 0, 1, 2, 3);
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        const int k = nk - 1;
        const uint32_t xo = ((2 * k) % G_NSLOT) * G_SLAB, wo = ((2 * k + 1) % G_NSLOT) * G_SLAB;
        G_READ(xo, wo, 1, 1);
        G_LGKM(6, 0);
        G_MMA_DMA(0, srcx, 0, -1, -1, -1, -1);
        G_READ(xo, wo, 2, 0);
        G_LGKM(6, 1);
        G_MMA_DMA(1, srcx, 0, -1, -1, -1, -1);
        G_READ(xo, wo, 3, 1);
        G_LGKM(6, 0);
        G_MMA_DMA(0, srcx, 0, -1, -1, -1, -1);
        G_LGKM(0, 1);
        G_MMA_DMA(1, srcx, 0, -1, -1, -1, -1);
        G_VMCNT(0);                                              // (tail re-fetches: nothing may land after the LDS is released)
    }

    // ---- epilogue.  D[n][m]: a lane holds column m = lane & 31 and rows n = (r&3) + 8(r>>2) + 4*half, i.e. runs of four
    // consecutive n.  The wave transposes its 128 x 64 tile through its own 18 KiB of the (now idle) LDS -- rows of
    // 128 B + 16 B pad -- so that global traffic moves whole 128-byte row segments, 16 B per lane and eight lanes per
    // output row (the direct 8-byte scatter cost ~10 % of the kernel in partial-line accesses).  The residual tile
    // takes the same way in; bias and residual are added in fp32 before the one rounding.
    G_BARRIER();                                                 // every wave is done with the operand slabs
    {
        constexpr int PITCH = 144;
        unsigned char* tr = smem + wave * (128 * PITCH);
        const int rsub = lane >> 3, cseg = lane & 7;             // 8 rows x 8 segments of 16 B per pass
        const int nseg = n0 + wn * 64 + cseg * 8;
        if (RES) {                                               // residual tile -> LDS, coalesced
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = it * 8 + rsub;
                const int64_t m = m0 + wm * 128 + row;
                uint4 r = make_uint4(0, 0, 0, 0);
                if (m < a.M) r = *(const uint4*)(a.res + m * a.N + nseg);
                *(uint4*)(tr + row * PITCH + cseg * 16) = r;
            }
        }
        asm volatile("" ::: "memory");       // (uint4 rows and uint2 cells are distinct types: keep the passes ordered)
        uint2 bpk[2][4];                                         // this lane's 32 bias values, packed
        if (BIAS) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) bpk[i][g] = *(const uint2*)(a.bias + n0 + wn * 64 + i * 32 + 8 * g + 4 * half);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)                              // one accumulator tile at a time (register pressure)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned char* cell = tr + (j * 32 + l31) * PITCH + (i * 32 + 8 * g + 4 * half) * 2;
                    float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (BIAS) {
                        v[0] += bf_lo(bpk[i][g].x); v[1] += bf_hi(bpk[i][g].x);
                        v[2] += bf_lo(bpk[i][g].y); v[3] += bf_hi(bpk[i][g].y);
                    }
                    if (RES) {
                        const uint2 r = *(const uint2*)cell;
                        v[0] += bf_lo(r.x); v[1] += bf_hi(r.x); v[2] += bf_lo(r.y); v[3] += bf_hi(r.y);
                    }
                    uint2 o;
                    o.x = pack_bf2(v[0], v[1]);                  // the one rounding
                    o.y = pack_bf2(v[2], v[3]);
                    *(uint2*)cell = o;
                }
        asm volatile("" ::: "memory");
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = it * 8 + rsub;
            const int64_t m = m0 + wm * 128 + row;
            const uint4 o = *(const uint4*)(tr + row * PITCH + cseg * 16);
            if (m < a.M) *(uint4*)(a.y + m * a.N + nseg) = o;
        }
    }
}


// ================================================================================================================
// PERSISTENT 4-wave form (the default).  One workgroup per CU walks a list of output tiles; the (tile, k-step) pairs
// form ONE stream of stages that flows through the five-slot LDS ring without a break, so the next tile's first operands
// are in flight while the current tile finishes and its epilogue runs (the one-tile-per-workgroup kernel above spends
// 26-40 k of its ~190 k cycles per tile on launch, pipeline fill and epilogue with the MFMA pipe idle).
// Wave (wm, wn) of a 2 x 2 grid owns 128(M) x 128(N) = 8 x 8 tiles of v_mfma_f32_16x16x32_bf16; the 256 accumulator
// registers live in the AGPR half of the unified register file (one wave per SIMD -> 512 registers per lane); per 32-deep
// half step a wave reads 8 + 8 fragments for 64 MFMAs.  (16x16x32 moves 4x fewer accumulator bytes per flop than 32x32x16:
// at equal cycle counts the 32x32x16 form of this loop clocked 1.57 GHz where hipBLASLt's 16x16x32 loop clocks 1.69.)
// With ONE wave per SIMD nothing hides a stalled wave, so the loop is written gap by gap (gap = the slot behind one MFMA,
// 128 per k-step), at most one memory instruction per gap.  Measured next to back-to-back v_mfma_f32_32x32x16_bf16
// (tools/probes/mfma_issue_probe.hip, profiles/r02_gemm_notes.txt): a ds_read_b128, SALU or VALU instruction per gap is
// free; a 16-byte-per-lane global load INTO REGISTERS costs ~24 cycles of MFMA issue each and a ds_write_b128 ~11 (the
// global -> VGPR -> LDS form of this loop ran at 38.5 instead of 32 cycles per MFMA); an LDS-DMA load costs the same
// when its M0 write (+ s_nop) sits in the same gap -- and ~6 cycles when M0 is written one gap earlier.  Hence:
// operands come by buffer_load_dwordx4 ... lds (resource = whole matrix, voffset = lane's row/granule, soffset = tile +
// k; rows past the end read as zeros, so the ragged last M tile needs no clamping), M0 is set in the preceding odd gap,
// fragment reads sit in the even gaps.
// A one-wave-per-SIMD stream issues one instruction per ~4 cycles, so a 16-cycle MFMA slot has room for three more: the scalar
// bookkeeping of the next k-step rides in the gaps in pieces of <= 3 instructions (round 3; as bursts of 14-20 it idled the pipe).
// Epilogue (round 3): the W rows sit in the LDS slab in a relabelled order that makes a lane's accumulators of an n-tile pair
// eight consecutive output columns; results go from registers to memory as whole 128-byte lines, no LDS, no barrier
// (profiles/r03_gemm_notes.txt: 97-99 % of hipBLASLt on the model's four layer shapes).
#ifndef GR_DGAP
#define GR_DGAP 8                            // gaps between the DMA pieces of a half step (8 pieces: 8 = spread over all 64 gaps, 4 = first 32)
#endif
#ifndef GE_ORDER
#define GE_ORDER 1                           // epilogue unit order: 1 = the four 64-byte quarters of a row pair back to back (L2 merges them into whole lines)
#endif
#ifndef GE_FULL
#define GE_FULL 1                            // 1: strips stored pairwise as whole 128-byte lines (needs GE_ORDER 1)
#endif
static_assert(!GE_FULL || GE_ORDER == 1, "GE_FULL pairs the strips of GE_ORDER 1");
// epilogue units: 32 per wave tile, unit = (strip b of 32 output columns, m tile j of 16 rows)
#if GE_ORDER == 0
#define GE_B(U) ((U) >> 3)                   /* unit U = (strip b = U >> 3, m tile j = U & 7) */
#define GE_J(U) ((U) & 7)
#else
#define GE_B(U) ((U) & 3)                    /* unit U = (m tile j = U >> 2, strip b = U & 3): the four 64-byte quarters of a row pair back to back */
#define GE_J(U) ((U) >> 2)
#endif

#ifndef GE_STPOL_ID
#define GE_STPOL_ID 0                        // cache policy of the epilogue's stores: 0 default, 1 nt, 2 sc1, 3 sc0 sc1
#endif
#if GE_STPOL_ID == 1
#define GE_STPOL " nt"
#elif GE_STPOL_ID == 2
#define GE_STPOL " sc1"
#elif GE_STPOL_ID == 3
#define GE_STPOL " sc0 sc1"
#else
#define GE_STPOL ""
#endif
#ifndef GR_MIDB
#define GR_MIDB 1                            // 1: the stage barrier behind the first MFMA of half 1; 0: in front of it
#endif
#ifndef GR_PAD
#define GR_PAD 0
#endif
#ifndef GR_ALIGN
#define GR_ALIGN 0
#endif
#ifndef GR_PROFILE
#define GR_PROFILE 0                         // 1: wave 0 of workgroup 0 accumulates cycles per loop segment, written over y (tools/gemm_stage_profile.py)
#endif

// XB (round 4): the X operand is the BLOCKED output of the channel-stationary Hyena operator (csrc/hyena_cs.hip),
//   [row block of 128][K / 16 groups][128 rows][16 channels] bf16 -- a 64-channel slab of a row is four 32-byte pieces 4 KiB apart
//   instead of one 128-byte piece; only the DMA's SOURCE addresses change (the lane that fills granule s of LDS row r fetches
//   32-byte piece s >> 1, half s & 1), the k-step is 16 KiB instead of 128 B; the tile origin m0 * K * 2 is the same number.
// NF (round 5): the RMSNorm passes around the dense layers, folded into their epilogues [REF stripedhyena/model.py: pre_norm / post_norm of
//   every block; layers.py RMSNorm].  The reference writes a normalised copy n = bf16(g * x / (rms(x) + eps)) of the residual stream and
//   multiplies it by the next layer's weight; here the layer that WRITES the stream also emits the statistic, and the layer that consumes it
//   reads the stream itself:  W n = rstd_row * ((W diag(g)) x)  -- the scale vector folded into a copy of the weight (bf16(W g): one rounding of
//   the weight instead of one of the activation), the row factor applied to the fp32 accumulators before bias / gate / the one rounding.
//   NF & 1, STATS (RES launches): the rounded rows this tile stores are squared and summed per row over the wave's 128 columns
//     (a lane holds 32 values of its row per m tile; two cross-lane steps join the four column groups) -> ss[n0 / 128 + wn][m]:
//     evo_rms_finalize_f32 adds the N / 128 partials in a fixed order (no atomics: bit-reproducible).  ~520 VALU per lane and tile under
//     the stores the epilogue is paced by.
//   NF & 2, SCALE (launches without a residual): the token's factor rs[row] multiplies the accumulators.  Tokens run along the LANES
//     in modes 0 / 1 (eight values per lane and tile: one per m tile) and along the REGISTERS in mode 3 (the swapped launch: 32 values per
//     lane, the eight columns of its four strips).  Mode 3 also reads its token rows from the residual stream itself: z^T position
//     p = b Tm + t is row p + b row_skip of x (tail form of z^T: rows of Tm = T - row_skip positions, a tile never straddles two batch rows).
template <bool BIAS, bool RES, int MODE, bool XB = false, int NF = 0>   // MODE 0: y [M, N]; 1: gated MLP, a [M, N / 2]; (2: retired with csrc/hyena_cs.hip in round 5);
                                                            // 3: y [M, N] with the bias indexed by ROW (the swapped-operand launch: evo_linear_t_mfma_bf16)
__global__ __launch_bounds__(256, 1) void gemmr_bf16_kernel(GemmArgs a) {
    constexpr bool STATS = (NF & 1) != 0, SCALE = (NF & 2) != 0;
    static_assert(!STATS || (RES && MODE == 0), "the statistic rides on the launches that write the residual stream");
    static_assert(!SCALE || !RES, "the row factor belongs to the launches that read the normalised stream");
    constexpr uint32_t XSTEP = XB ? 16384u : (uint32_t)(GBK * 2);   // bytes from one k-step's X slab to the next
    __shared__ __attribute__((aligned(16))) unsigned char smem[G_NSLOT * G_SLAB];   // the ONLY __shared__ object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                     // 2 x 2 wave grid
    const uint32_t kb = (uint32_t)a.K * 2;                       // bytes per operand row
    const int nk = a.K / GBK;

    // ---- this workgroup's tiles.  Block b runs on XCD b % 8; every XCD owns a contiguous run of (group_m-rastered) tile
    //      ids and its workgroups take them round-robin, so the tiles an XCD computes at one time are consecutive ids.
    int t_first, t_step, n_my;
    {
        const int bid = blockIdx.x, per = gridDim.x >> 3;        // host: gridDim.x % 8 == 0
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = a.n_tiles >> 3, r = a.n_tiles & 7;
        const int xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int xn = q + (xcd < r ? 1 : 0);
        t_first = xs + slot;
        t_step = per;
        n_my = slot < xn ? (xn - slot + per - 1) / per : 0;
    }
    if (n_my == 0) return;
    auto tile_origin = [&](int tile, int64_t& m0, int& n0) {
        if constexpr (MODE == 3) {
            // swapped operands: the rows (m) are the WEIGHT, the columns (n) the tokens -- the raster is mirrored too, groups of
            // group_m COLUMN tiles swept over all row tiles, so that the operand re-streamed once per group is again the weight
            // (100 MB: it stays in the 256 MB memory-side cache) and not the activations (0.5-1 GB).  With the m-grouped raster
            // this launch ran 4-5 % behind the group-major projection of the same shape (tools/hc_bench.py).
            const int per_group = a.group_m * a.tiles_m;
            const int grp = tile / per_group, in_grp = tile - grp * per_group;
            const int first_n = grp * a.group_m;
            const int gsz = a.tiles_n - first_n < a.group_m ? a.tiles_n - first_n : a.group_m;
            const int tm = in_grp / gsz;
            m0 = (int64_t)tm * GBM;
            n0 = (first_n + in_grp - tm * gsz) * GBN;
            return;
        }
        const int per_group = a.group_m * a.tiles_n;
        const int grp = tile / per_group, in_grp = tile - grp * per_group;
        const int first_m = grp * a.group_m;
        const int gsz = a.tiles_m - first_m < a.group_m ? a.tiles_m - first_m : a.group_m;
        const int tn = in_grp / gsz;
        m0 = (int64_t)(first_m + in_grp - tn * gsz) * GBM;
        n0 = tn * GBN;
    };

    // MODE 3 + SCALE: first row of the token matrix behind column origin n0 (a multiple of 256 positions)
    auto src_row = [&](int n0) -> uint32_t {
        if constexpr (MODE == 3 && SCALE) return (uint32_t)n0 + (uint32_t)((n0 / GBN) / a.tm_tiles) * (uint32_t)a.row_skip;
        else return (uint32_t)n0;
    };

    // ---- DMA plan: a slab is 32 one-KiB pieces (8 rows of 128 B each); wave w moves pieces w, w + 4, ..., w + 28.  The LDS
    //      side of a DMA is lane-linear (M0 + 16 lane), so the swizzle is applied to the source: the lane that fills granule
    //      slot s of row r fetches granule s ^ (r & 7) of that row (a fragment read touches 16 consecutive rows x 4 granules:
    //      8 consecutive rows hit 8 different slots = all 32 banks).  Rows of one lane are 32 apart: same swizzle term.
    const int r0 = 8 * wave + (lane >> 3);
    const uint32_t xs_ = (uint32_t)(((lane & 7) ^ r0) & 7);       // source granule of this lane's LDS slot
    const uint32_t voff0 = XB ? (uint32_t)r0 * 32u + (xs_ >> 1) * 4096u + (xs_ & 1) * 16u
                              : (uint32_t)r0 * kb + xs_ * 16;
    const uint32_t row32 = 32u * kb;                             // voffset of piece jj = voff0 + jj * row32
    uint32_t voff[8], wvoff[8];
    // W slabs: LDS row 32 b + 16 t + 4 q + r holds W row 32 b + 8 q + 4 t + r (a relabelling inside 32-row blocks: every row is
    // still one whole 128-byte line, the swizzle stays keyed on the LDS row) -- it makes the epilogue's stores 16 bytes per lane
    const int wrow = 8 * ((r0 >> 2) & 3) + 4 * ((r0 >> 4) & 1) + (r0 & 3);
    const uint32_t wvoff0 = (uint32_t)wrow * kb + (uint32_t)((((lane & 7) ^ r0) & 7) * 16);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        // (blocked X: rows r0 + 32 jj of the 256-row tile = row (r0 + 32 jj) & 127 of row block jj >> 2, 128 * kb bytes apart)
        voff[jj] = XB ? voff0 + (uint32_t)(jj & 3) * 1024u + (uint32_t)(jj >> 2) * 128u * kb : voff0 + jj * row32;
        wvoff[jj] = wvoff0 + jj * row32;
    }
    const uint64_t xa64 = (uint64_t)a.x, wa64 = (uint64_t)a.w;
    const g_u32x4 rx = {(uint32_t)xa64, (uint32_t)(xa64 >> 32) & 0xffffu,
                        (uint32_t)((XB ? (a.M + 127) / 128 * 128 : a.M) * (int64_t)kb), 0x00020000u};
    const g_u32x4 rw = {(uint32_t)wa64, (uint32_t)(wa64 >> 32) & 0xffffu, (uint32_t)((MODE == 3 && SCALE ? a.w_rows : (int64_t)a.N) * kb), 0x00020000u};
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t lds_dma = lds0 + wave * 1024;                 // + slot * G_SLAB + jj * 4096

    // fetch cursor: the stage whose slabs are issued next (soffsets: tile origin + k * 128 bytes)
    int f_i = 0, f_k = 0;
    uint32_t fxs, fws;
    {
        int64_t m0; int n0;
        tile_origin(t_first, m0, n0);
        fxs = (uint32_t)(m0 * kb);
        fws = src_row(n0) * kb;
    }
    auto fetch_advance = [&]() {                                 // (prologue only; past the last stage: stay)
        if (f_k + 1 < nk) { ++f_k; fxs += XSTEP; fws += GBK * 2; }
        else if (f_i + 1 < n_my) {
            ++f_i; f_k = 0;
            int64_t m0; int n0;
            tile_origin(t_first + f_i * t_step, m0, n0);
            fxs = (uint32_t)(m0 * kb);
            fws = src_row(n0) * kb;
        }
    };
    // Inside the stream the cursor moves WITHOUT branches (a taken branch in the middle of the MFMA stream is an instruction
    // refetch with the pipe idle; the first version of this loop had two per k-step): the soffsets of the first stage of the
    // tile after the fetch tile (nx0, nw0) are kept ready -- recomputed once per tile, in the epilogue -- and the step is
    // a chain of scalar selects.  Past the last stage the cursor stays (harmless re-fetches keep the vmcnt counts constant).
    uint32_t nx0 = 0, nw0 = 0, nfxs = 0, nfws = 0, fxp = 0, fwp = 0;
    int f_rem = 0;                                               // k-steps left in the fetch tile (set after the prologue)
    auto next_tile_origin = [&]() {
        if (f_i + 1 < n_my) {
            int64_t m0; int n0;
            tile_origin(t_first + (f_i + 1) * t_step, m0, n0);
            nx0 = (uint32_t)(m0 * kb);
            nw0 = src_row(n0) * kb;
        }
    };
    // The step itself is written as pieces of two or three scalar instructions, one piece per MFMA gap (GR_GAP): a one-wave-
    // per-SIMD stream issues one instruction per ~4 cycles, so a 16-cycle MFMA slot has room for three more -- a burst of 14-20
    // scalar instructions in one gap (the first form of this bookkeeping) idles the matrix pipe for 40-70 cycles.  Past the
    // last tile the cursor re-enters the last origin it knew (harmless re-fetches of valid rows keep the vmcnt counts constant).
#define GD_M0(V) asm volatile("s_mov_b32 m0, %0" ::"s"(V) : "memory", "m0")
#ifndef GR_ABL
#define GR_ABL 0                             // ablation bits (measurement builds only): 1 no in-loop DMA, 2 no in-loop barrier, 8 no in-loop fragment reads, 16 no M0 writes
#endif
#define GD_DMAX(JJ) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff[JJ]), "s"(rx), "s"(fxs) : "memory")
#define GD_DMAW(JJ) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(wvoff[JJ]), "s"(rw), "s"(fws) : "memory")

    // fragment bases (within a slab): v_mfma_f32_16x16x32_bf16 operands -- lane (row = l & 15, kg = l >> 4) holds the 8 bf16
    // k = 32 kh + 8 kg .. + 7 of its row.  A operand = W rows (n), B operand = X rows (m): D[n][m], a lane's four accumulator
    // registers run along n.  Tiles 16 rows apart share (row & 7): one base per k-half, tile offsets are immediates.
    const int l15 = lane & 15, lq = lane >> 4;
    uint32_t x_rd[2], w_rd[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int xr = wm * 128 + l15, wr = wn * 128 + l15;
        x_rd[kh] = lds0 + (uint32_t)(xr * G_ROW + (((4 * kh + lq) ^ xr) & 7) * 16);
        w_rd[kh] = lds0 + (uint32_t)(wr * G_ROW + (((4 * kh + lq) ^ wr) & 7) * 16);
    }

    // epilogue: this lane's byte offset inside an output tile's row block (row wm 128 + l15, column wn 128 + 8 lq); 16 rows further
    // (MODE 3 writes its result in column blocks of 256: [N / 256][M][256] -- an output tile is ONE contiguous 128 KiB piece; as a
    //  plain [M][N] matrix the 256 rows of a tile of z^T lie N * 2 = 130-260 KB apart, 17-33 different 2 MiB pages per tile, and the
    //  launch ran 3-4 % behind the group-major projection at 131 k)
    const int ldy = MODE == 3 ? GBN : a.N;
    const uint32_t ep_voff = (uint32_t)((wm * 128 + l15) * ldy + wn * 128 + 8 * lq) * 2u;
    const int ep_rows16 = 16 * ldy * 2;
    // whole-line form: row (l15 & 7) of an 8-row group, byte 64 (l15 >> 3) + 16 lq of the 128-byte line of a strip pair
    const uint32_t ep_voff_f = (uint32_t)((wm * 128 + (l15 & 7)) * ldy + wn * 128 + 8 * lq + 32 * (l15 >> 3)) * 2u;
    const uint32_t ep_boff = (uint32_t)(wn * 128 + 8 * lq) * 2u;        // bias: this lane's eight columns within a strip of the tile
    const uint32_t ep_rboff = (uint32_t)(wm * 128 + l15) * 2u;           // MODE 3 (row bias): this lane's row within the tile, m tile 0
    const uint32_t ep_rsoff = (uint32_t)(wm * 128 + l15) * 4u;           // SCALE, modes 0 / 1, and STATS: this lane's row within the tile (fp32 per row), m tile 0
    const uint32_t ep_rsoff3 = (uint32_t)(wn * 128 + 8 * lq) * 4u;       // SCALE, mode 3: this lane's eight columns (= tokens) within a strip of the tile
    // GATE: the output is [M, N / 2]; a wave's 128 tile columns = two gated strips of 32 columns = one 128-byte line per row
    const int ep_gn = a.N >> 1;
    const uint32_t ep_voff_g = (uint32_t)((wm * 128 + (l15 & 7)) * ep_gn + wn * 64 + 8 * lq + 32 * (l15 >> 3)) * 2u;
    const int ep_grows16 = 16 * ep_gn * 2;
    f32x4_t acc[8][8];                                           // [n tile][m tile]
// (the empty asm pins a zeroed quad in its AGPRs HERE, in program order with the other volatile asm statements: left free,
//  hipcc may sink the v_accvgpr_write next to the inline-asm MFMAs, whose hazards it does not model -- seen once as wrong
//  sums in the fourth register of every quad when a branch followed the epilogue)
#define GR_ZERO1(I, J) { _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) acc[I][J][r_] = 0.f; asm volatile("" : "+a"(acc[I][J])); }
#define GR_ZERO() _Pragma("unroll") for (int i = 0; i < 8; ++i) _Pragma("unroll") for (int j = 0; j < 8; ++j) GR_ZERO1(i, j)
    GR_ZERO();
    asm volatile("s_nop 7" ::: "memory");
    g_u32x4 wf[2][8], xf[2][8];                                  // double-buffered fragments of one k-half

#define GR_LGKM(N, BUF)                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(%16)" : "+v"(wf[BUF][0]), "+v"(wf[BUF][1]), "+v"(wf[BUF][2]), "+v"(wf[BUF][3]), \
                 "+v"(wf[BUF][4]), "+v"(wf[BUF][5]), "+v"(wf[BUF][6]), "+v"(wf[BUF][7]),                      \
                 "+v"(xf[BUF][0]), "+v"(xf[BUF][1]), "+v"(xf[BUF][2]), "+v"(xf[BUF][3]),                      \
                 "+v"(xf[BUF][4]), "+v"(xf[BUF][5]), "+v"(xf[BUF][6]), "+v"(xf[BUF][7]) : "n"(N))
    // (inline asm with the accumulator pinned to the AGPR file: as a builtin, hipcc splits the 64 four-register accumulators
    //  between VGPRs and AGPRs and shuffles ~200 v_accvgpr_read / write / mov through every k-step.  The hazards the compiler
    //  no longer sees: operands come from ds_read + s_waitcnt (no wait states needed); an accumulator is touched once per 64
    //  MFMAs; the zeroing before a tile and the reads after it are fenced with s_nop below.)
#define GR_MFMA(I, J, BUF)                                                                                    \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(wf[BUF][I]), "v"(xf[BUF][J]))
    // fragment read Q of k-half KH (X slab at byte offset XO, W slab at WO) into fragment buffer BUF: Q = 0..7 X, 8..15 W tiles
#define GR_RD1(XO, WO, KH, BUF, Q)                                                                            \
    { if ((Q) < 8) G_DSR(xf[BUF][(Q) & 7], (XO) + x_rd[KH], ((Q) & 7) * 16 * G_ROW);                          \
      else G_DSR(wf[BUF][(Q) & 7], (WO) + w_rd[KH], ((Q) & 7) * 16 * G_ROW); }
    // The gap behind MFMA number G (0..127) of a k-step, at most ONE memory instruction per gap:
    //   even gaps 0..30 of each half: one fragment read each (the next half's fragments: X0..X7, W0..W7 -- all of them are
    //                                 needed within the first MFMAs of that half, which waits for them with lgkmcnt(0); spreading
    //                                 half 1's reads over all 64 gaps with row-wise counted waits measured neutral)
    //   gaps 8j+1 / 8j+5 of half 0 (j = 0..7): M0 <- destination of X piece j of stage g+2 / its DMA
    //   gaps 8j+1 / 8j+5 of half 1 (behind the barrier): M0 <- destination of W piece j of stage g+2 / its DMA
#define GR_GAP(G, RXO, RWO, RKH, RBUF)                                                                        \
    {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        constexpr int g_ = (G), s_ = g_ & 63;                                                                 \
        /* the stage barrier sits BEHIND the first MFMA of half 1 (its operands are in registers): the matrix pipe works  */ \
        /* through that MFMA while the wave waits for the others                                                          */ \
        if constexpr (g_ == 64 && GR_MIDB) { G_VMCNT(8); if (!(GR_ABL & 2)) G_BARRIER(); }                    \
        if constexpr ((s_ & 1) == 0 && s_ < 32 && !(GR_ABL & 8)) { GR_RD1(RXO, RWO, RKH, RBUF, s_ >> 1); }    \
        if constexpr (g_ < 64 && s_ < 8 * GR_DGAP && (s_ % GR_DGAP) == 1 && !(GR_ABL & 16)) GD_M0(fxl + (s_ / GR_DGAP) * 4096);        \
        if constexpr (g_ < 64 && s_ < 8 * GR_DGAP && (s_ % GR_DGAP) == GR_DGAP / 2 + 1 && !(GR_ABL & 1)) GD_DMAX(s_ / GR_DGAP);       \
        if constexpr (g_ >= 64 && s_ < 8 * GR_DGAP && (s_ % GR_DGAP) == 1 && !(GR_ABL & 16)) GD_M0(fwl + (s_ / GR_DGAP) * 4096);       \
        if constexpr (g_ >= 64 && s_ < 8 * GR_DGAP && (s_ % GR_DGAP) == GR_DGAP / 2 + 1 && !(GR_ABL & 1)) GD_DMAW(s_ / GR_DGAP);      \
        /* scalar bookkeeping of the NEXT k-step, <= 3 instructions per gap, in gaps that carry no memory instruction and no   */ \
        /* M0 write (half 1: this k-step's X pieces are out; the W pieces end at gap 125; slot offsets are dead once read).    */ \
        if constexpr (g_ == 67) { fxp = fxs + XSTEP; fwp = fws + GBK * 2; GS_PIN2(fxp, fwp); }              \
        if constexpr (g_ == 71) { f_rem -= 1; GS_PIN1(f_rem); }                                               \
        if constexpr (g_ == 75) { nfxs = f_rem == 0 ? nx0 : fxp; GS_PIN1(nfxs); }                             \
        if constexpr (g_ == 87) { nfws = f_rem == 0 ? nw0 : fwp; GS_PIN1(nfws); }                             \
        if constexpr (g_ == 110) { c_k -= 1; GS_PIN1(c_k); }                                                  \
        if constexpr (g_ == 91) { const bool z_ = f_rem == 0; f_rem = z_ ? nk : f_rem; f_i += z_ ? 1 : 0; GS_PIN2(f_rem, f_i); } \
        if constexpr (g_ == 83) { fxs = nfxs; GS_PIN1(fxs); }                                                 \
        /* the five-slot ring advances by two slots per stage: {xo, wo, nxo, nwo, fxo} <- {nxo, nwo, fxo, xo, wo}             */ \
        if constexpr (g_ == 96) { r_t0 = xo; xo = nxo; GS_PIN2(r_t0, xo); }                                   \
        if constexpr (g_ == 98) { r_t1 = wo; wo = nwo; GS_PIN2(r_t1, wo); }                                   \
        if constexpr (g_ == 100) { nxo = fxo; nwo = r_t0; GS_PIN2(nxo, nwo); }                                \
        if constexpr (g_ == 102) { fxo = r_t1; fxl = lds_dma + r_t1; GS_PIN2(fxo, fxl); }                     \
        if constexpr (g_ == 126) { fws = nfws; fwl = lds_dma + xo; GS_PIN2(fws, fwl); }                       \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    // eight MFMAs of fragment row I of half SS on fragment buffer BUF, each followed by its gap
#define GR_ROW8(SS, I, BUF, RXO, RWO, RKH, RBUF)                                                              \
    {                                                                                                         \
        GR_MFMA(I, 0, BUF); GR_GAP(64 * (SS) + 8 * (I) + 0, RXO, RWO, RKH, RBUF); GR_MFMA(I, 1, BUF); GR_GAP(64 * (SS) + 8 * (I) + 1, RXO, RWO, RKH, RBUF); \
        GR_MFMA(I, 2, BUF); GR_GAP(64 * (SS) + 8 * (I) + 2, RXO, RWO, RKH, RBUF); GR_MFMA(I, 3, BUF); GR_GAP(64 * (SS) + 8 * (I) + 3, RXO, RWO, RKH, RBUF); \
        GR_MFMA(I, 4, BUF); GR_GAP(64 * (SS) + 8 * (I) + 4, RXO, RWO, RKH, RBUF); GR_MFMA(I, 5, BUF); GR_GAP(64 * (SS) + 8 * (I) + 5, RXO, RWO, RKH, RBUF); \
        GR_MFMA(I, 6, BUF); GR_GAP(64 * (SS) + 8 * (I) + 6, RXO, RWO, RKH, RBUF); GR_MFMA(I, 7, BUF); GR_GAP(64 * (SS) + 8 * (I) + 7, RXO, RWO, RKH, RBUF); \
    }
#define GR_SUB(SS, BUF, RXO, RWO, RKH, RBUF)                                                                  \
    {                                                                                                         \
        GR_ROW8(SS, 0, BUF, RXO, RWO, RKH, RBUF); GR_ROW8(SS, 1, BUF, RXO, RWO, RKH, RBUF);                   \
        GR_ROW8(SS, 2, BUF, RXO, RWO, RKH, RBUF); GR_ROW8(SS, 3, BUF, RXO, RWO, RKH, RBUF);                   \
        GR_ROW8(SS, 4, BUF, RXO, RWO, RKH, RBUF); GR_ROW8(SS, 5, BUF, RXO, RWO, RKH, RBUF);                   \
        GR_ROW8(SS, 6, BUF, RXO, RWO, RKH, RBUF); GR_ROW8(SS, 7, BUF, RXO, RWO, RKH, RBUF);                   \
    }
#define GS_PIN1(A) asm volatile("" : "+s"(A))
#define GS_PIN2(A, B) asm volatile("" : "+s"(A), "+s"(B))

    // ---- prologue: stages 0 and 1 (slabs 0..3 -> slots 0..3)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) { GD_M0(lds_dma + 0 * G_SLAB + jj * 4096); asm volatile("s_nop 0"); GD_DMAX(jj); }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) { GD_M0(lds_dma + 1 * G_SLAB + jj * 4096); asm volatile("s_nop 0"); GD_DMAW(jj); }
    fetch_advance();
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) { GD_M0(lds_dma + 2 * G_SLAB + jj * 4096); asm volatile("s_nop 0"); GD_DMAX(jj); }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) { GD_M0(lds_dma + 3 * G_SLAB + jj * 4096); asm volatile("s_nop 0"); GD_DMAW(jj); }
    fetch_advance();
    f_rem = nk - f_k;
    next_tile_origin();
    G_VMCNT(16);
    G_BARRIER();
    {
        const uint32_t x0_ = 0, w0_ = G_SLAB;
#pragma unroll
        for (int q = 0; q < 16; ++q) GR_RD1(x0_, w0_, 0, 0, q);
    }

#if GR_PROFILE
    const bool prof = blockIdx.x == 0 && wave == 0;
    uint64_t tp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter();
#define GR_STAMP(K) if (prof) { const uint64_t n_ = __builtin_readcyclecounter(); tp[K] += n_ - tl; tl = n_; }
#else
#define GR_STAMP(K)
#endif
    // ---- the stage stream.  Stage g = slabs 2g (X), 2g+1 (W) in slots (2g) % 5, (2g+1) % 5; sl = (2g) % 5.
    //        half 0 (k 0..31)   64 MFMAs + the second half's fragment reads + DMA X(g+2) -> slot (sl+4) % 5 (freed by barrier g-1)
    //        wait               vmcnt(8): everything but X(g+2) retired -> stage g+1 landed [RAW]; lgkmcnt(0): stage g read [WAR]
    //        barrier g          stage g+1 visible to all; slots of stage g free
    //        half 1 (k 32..63)  64 MFMAs + the first fragment reads of stage g+1 + DMA W(g+2) -> slot sl (held X(g))
    uint32_t xo = 0, wo = G_SLAB, nxo = 2 * G_SLAB, nwo = 3 * G_SLAB, fxo = 4 * G_SLAB;     // slot byte offsets of stage g: X, W; stage g+1: X, W; free
    uint32_t fxl = lds_dma + 4 * G_SLAB, fwl = lds_dma, r_t0 = 0, r_t1 = 0;
    // code-placement knobs (measurement builds; MI355X_MICROARCH.md "code-placement sensitivity of hand-written streams"): GR_PAD shifts the
    // whole stage stream by 4-byte s_nop's, GR_ALIGN pins the head of the tile loop to a 2^GR_ALIGN-byte boundary
#if GR_PAD
    asm volatile(".rept %0\n\ts_nop 0\n\t.endr" :: "n"(GR_PAD));
#endif
#if GR_ALIGN
    asm volatile(".p2align %0" :: "n"(GR_ALIGN));
#endif
    for (int c_i = 0; c_i < n_my; ++c_i) {
#define GR_KSTEP()                                                                                            \
        {                                                                                                     \
            GR_LGKM(0, 0);                                                                                    \
            GR_STAMP(5);                                                                                      \
            GR_SUB(0, 0, xo, wo, 1, 1);                                                                       \
            GR_STAMP(0);                                                                                      \
            GR_LGKM(0, 1);                                       /* the second half's fragments: stage g is fully read */ \
            GR_STAMP(1);                                                                                      \
            if (!GR_MIDB) { G_VMCNT(8); GR_STAMP(2); if (!(GR_ABL & 2)) G_BARRIER(); }                        \
            GR_STAMP(6);                                                                                      \
            GR_SUB(1, 1, nxo, nwo, 0, 0);                                                                     \
            GR_STAMP(3);                                                                                      \
        }
        for (int c_k = nk; c_k != 0;) GR_KSTEP();              // (the count-down sits in gap 110)
        {
            // ---- epilogue of tile c_i.  D[n][m] of a 16 x 16 tile: a lane holds column m = lane & 15 and the four rows
            // 4 (lane >> 4) + 0..3.  The W rows sit in LDS in a permuted order (DMA plan above): row position p = 4 q + r of n
            // tile 2 b + t is output column 32 b + 8 q + 4 t + r -- so a lane's values of the tile pair (2 b, 2 b + 1) are EIGHT
            // consecutive output columns of its row: 16 bytes, stored straight from registers (a store instruction covers 16
            // rows x 64 B).  No transposition through LDS, no scratch slot (the first form of this epilogue spent 6.8 k cycles
            // per tile moving 32-column strips through 8 KiB of LDS per wave).
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");    // the tile's last MFMAs (4 passes) -> accumulator reads
            int64_t m0; int n0;
            tile_origin(t_first + c_i * t_step, m0, n0);
            // bounded buffer descriptors over the tile's (<= 256) valid rows: rows past M are dropped / read as zero by the
            // hardware, so there is no branch per store; offset = per-lane part (row, column within the tile: ep_voff, the same
            // for every tile) + scalar part (n0, strip b, 16 j rows).  Every vector-memory instruction of the epilogue is inline
            // asm with hand-counted waits: a load the compiler can see here gets its s_waitcnt vmcnt placed INSIDE the k-loop
            // (at the first redefinition of the register), where it drains the operand DMA on every k-step (measured: -25 %).
            GR_STAMP(7);                                         // (tile origin arithmetic)
            if constexpr (MODE == 1) {
                // ---- gated MLP form [REF stripedhyena/layers.py ParallelGatedMLP: l3(gelu(l1 x) * l2 x)]: the weight rows come as
                // blocks of [32 rows of W1 | the matching 32 rows of W2] (HipOps.pack_gate_weights), so strip 2 p of a wave holds
                // z1 and strip 2 p + 1 holds z2 of the SAME 32 gated columns, eight per lane.  z1, z2 are rounded to bf16 (the
                // dense layers' outputs in the reference), the gate is evaluated in fp32 and rounded once -- the arithmetic of
                // evo_gelu_gate_bf16 on the unfused path.  The epilogue runs at the pace of its stores (~270 cycles per store
                // instruction): the ~25 packed VALU operations per output pair ride under them, the [M, 2 I] intermediate and the
                // gate kernel's pass over it (6 I bytes per token) disappear.
                const int rows_ok = a.M - m0 < GBM ? (int)(a.M - m0) : GBM;
                const uint64_t y64 = (uint64_t)(a.y + m0 * ep_gn);
                const g_u32x4 yd = {(uint32_t)y64, (uint32_t)(y64 >> 32) & 0xffffu, (uint32_t)(rows_ok * ep_gn * 2), 0x00020000u};
                g_u32x4 ost[8], gq[2];
                float rsj[8];                                      // SCALE: 1 / (rms + eps) of this lane's row in each of the eight m tiles
                if constexpr (SCALE) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        asm volatile("global_load_dword %0, %1, %2" : "=v"(rsj[j]) : "v"(ep_rsoff), "s"(a.rs + m0 + 16 * j) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rsj[0]), "+v"(rsj[1]), "+v"(rsj[2]), "+v"(rsj[3]), "+v"(rsj[4]), "+v"(rsj[5]),
                                 "+v"(rsj[6]), "+v"(rsj[7]) :: "memory");
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        __builtin_amdgcn_sched_barrier(0);
                        float z1[8], z2[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(z1[r]) : "a"(acc[4 * pp][j][r]));
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(z1[4 + r]) : "a"(acc[4 * pp + 1][j][r]));
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(z2[r]) : "a"(acc[4 * pp + 2][j][r]));
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(z2[4 + r]) : "a"(acc[4 * pp + 3][j][r]));
                        }
                        GR_ZERO1(4 * pp, j); GR_ZERO1(4 * pp + 1, j); GR_ZERO1(4 * pp + 2, j); GR_ZERO1(4 * pp + 3, j);
                        if constexpr (SCALE) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { z1[e] *= rsj[j]; z2[e] *= rsj[j]; }
                        }
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const uint32_t u1 = pack_bf2(z1[e], z1[e + 1]), u2 = pack_bf2(z2[e], z2[e + 1]);     // the dense layers' bf16 outputs
                            const f32x2_t uu = {bf_lo(u1), bf_hi(u1)}, ww = {bf_lo(u2), bf_hi(u2)};
                            const f32x2_t oo = gelu_gate2(uu, ww);
                            gq[pp][e >> 1] = pack_bf2(oo[0], oo[1]);
                        }
                    }
                    // whole 128-byte lines: lanes l15 < 8 and their partners l15 + 8 swap one quad (see the plain form below)
                    const bool low = l15 < 8;
                    g_u32x4 &s1 = ost[(2 * j) & 7], &s2 = ost[(2 * j + 1) & 7];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t give = low ? gq[1][d] : gq[0][d];
                        const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0x128, 0xf, 0xf, false);
                        s1[d] = low ? gq[0][d] : recv;
                        s2[d] = low ? recv : gq[1][d];
                    }
                    const int so = n0 + j * ep_grows16;            // (n0 / 2 gated columns x 2 bytes)
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" GE_STPOL :: "v"(s1), "v"(ep_voff_g), "s"(yd), "s"(so) : "memory");
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" GE_STPOL :: "v"(s2), "v"(ep_voff_g), "s"(yd), "s"(so + ep_grows16 / 2) : "memory");
                    if (j >= 3) { asm volatile("" :: "v"(ost[(2 * j + 2) & 7])); asm volatile("" :: "v"(ost[(2 * j + 3) & 7])); }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("" :: "v"(ost[k]));
                GR_STAMP(8);
            } else {
            const int rows_ok = a.M - m0 < GBM ? (int)(a.M - m0) : GBM;
            const uint64_t y64 = (uint64_t)(MODE == 3 ? a.y + ((int64_t)(n0 / GBN) * a.M + m0) * GBN : a.y + m0 * a.N);
            const uint64_t r64 = (uint64_t)((RES ? a.res : a.y) + m0 * a.N);
            const g_u32x4 yd = {(uint32_t)y64, (uint32_t)(y64 >> 32) & 0xffffu, (uint32_t)(rows_ok * ldy * 2), 0x00020000u};
            const g_u32x4 rd = {(uint32_t)r64, (uint32_t)(r64 >> 32) & 0xffffu, RES ? (uint32_t)(rows_ok * a.N * 2) : 0u, 0x00020000u};
#define GE_SOFF(U) ((MODE == 3 ? 0 : n0 * 2) + 64 * GE_B(U) + GE_J(U) * ep_rows16)
#define GE_NR 6                              /* residual requests in flight (a ring of GE_NR x 4 VGPRs; 8 spills fragment registers) */
#define GE_LOAD(U) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rr[(U) % GE_NR]) : "v"(ep_voff), "s"(rd), "s"(GE_SOFF(U)) : "memory")
            g_u32x4 rr[GE_NR], bq[4], ost[8], o_even;
            uint32_t rbq[8];                                     // MODE 3: the bias of this lane's row in each of the eight m tiles
            float rsj[8];                                        // SCALE, mode 0: the factor of this lane's row in each of the eight m tiles
            g_u32x4 rsq[4][2];                                   // SCALE, mode 3: the factors of the lane's eight columns (tokens) of every strip, as raw fp32
            if constexpr (SCALE && MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("global_load_dword %0, %1, %2" : "=v"(rsj[j]) : "v"(ep_rsoff), "s"(a.rs + m0 + 16 * j) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rsj[0]), "+v"(rsj[1]), "+v"(rsj[2]), "+v"(rsj[3]), "+v"(rsj[4]), "+v"(rsj[5]),
                             "+v"(rsj[6]), "+v"(rsj[7]) :: "memory");
            }
            if constexpr (SCALE && MODE == 3) {
                const float* rs0 = a.rs + src_row(n0);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rsq[b][0]) : "v"(ep_rsoff3), "s"(rs0 + b * 32) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rsq[b][1]) : "v"(ep_rsoff3), "s"(rs0 + b * 32 + 4) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rsq[0][0]), "+v"(rsq[0][1]), "+v"(rsq[1][0]), "+v"(rsq[1][1]), "+v"(rsq[2][0]),
                             "+v"(rsq[2][1]), "+v"(rsq[3][0]), "+v"(rsq[3][1]) :: "memory");
            }
            float ssq = 0.f, sst[4];                             // STATS: the running sum of squares of this lane's row in m tile j; stored totals
            if (BIAS && MODE != 3) {                             // the lane's eight columns of every strip
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[b]) : "v"(ep_boff), "s"(a.bias + n0 + b * 32) : "memory");
            }
            if (BIAS && MODE == 3) {
                // swapped operands: the rows of this launch are the dense layer's OUTPUT FEATURES, so the bias runs along m -- one
                // bf16 per lane and m tile (row m0 + wm 128 + 16 j + l15), the same value for the lane's eight columns
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("global_load_ushort %0, %1, %2" : "=v"(rbq[j]) : "v"(ep_rboff), "s"(a.bias + m0 + 16 * j) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rbq[0]), "+v"(rbq[1]), "+v"(rbq[2]), "+v"(rbq[3]), "+v"(rbq[4]), "+v"(rbq[5]),
                             "+v"(rbq[6]), "+v"(rbq[7]) :: "memory");
            }
            if (RES) {                                           // residual rows: a ring of eight requests ahead of the arithmetic
#pragma unroll
                for (int u = 0; u < GE_NR; ++u) GE_LOAD(u);
            }
            if (BIAS && !RES && MODE != 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) :: "memory");
#pragma unroll
            for (int u = 0; u < 32; ++u) {                       // unit (b, j): n tiles 2 b, 2 b + 1 x m tile j = 16 rows x 64 B
                const int b = GE_B(u), j = GE_J(u);
                __builtin_amdgcn_sched_barrier(0);               // (bounds the live ranges: 8 accumulator values at a time)
                if (RES) {
                    // loads return in order: "at most 7 outstanding" = everything up to unit u's request has landed (the stores in
                    // between can only make this wait longer); the first wait also covers the bias loads issued before the ring
                    if (u + GE_NR <= 32) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(rr[u % GE_NR]), "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) : "n"(GE_NR - 1) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(rr[u % GE_NR]) :: "memory");
                }
                // (accumulator reads as volatile asm: left to itself hipcc copies dozens of accumulators into VGPRs ahead of
                //  time and spills operand fragments to scratch to make room -- the reload's s_waitcnt then lands in the k-loop)
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[r]) : "a"(acc[2 * b][j][r]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[4 + r]) : "a"(acc[2 * b + 1][j][r]));
                }
                // the next tile's zeros, here: the epilogue runs at the pace of its stores (~270 cycles per store instruction and
                // wave, whatever it carries), the eight writes are free; behind the loop they cost 1.4 k cycles per tile
                GR_ZERO1(2 * b, j); GR_ZERO1(2 * b + 1, j);
                if constexpr (SCALE && MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= rsj[j];
                }
                if constexpr (SCALE && MODE == 3) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= __uint_as_float(rsq[b][e >> 2][e & 3]);
                }
                if (BIAS && MODE != 3) {
                    v[0] += bf_lo(bq[b][0]); v[1] += bf_hi(bq[b][0]); v[2] += bf_lo(bq[b][1]); v[3] += bf_hi(bq[b][1]);
                    v[4] += bf_lo(bq[b][2]); v[5] += bf_hi(bq[b][2]); v[6] += bf_lo(bq[b][3]); v[7] += bf_hi(bq[b][3]);
                }
                if (BIAS && MODE == 3) {
                    const float rb = bf_lo(rbq[j]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rb;
                }
                if (RES) {
                    const g_u32x4 r = rr[u % GE_NR];
                    v[0] += bf_lo(r[0]); v[1] += bf_hi(r[0]); v[2] += bf_lo(r[1]); v[3] += bf_hi(r[1]);
                    v[4] += bf_lo(r[2]); v[5] += bf_hi(r[2]); v[6] += bf_lo(r[3]); v[7] += bf_hi(r[3]);
                }
                // the one rounding
                g_u32x4 o;
                o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]); o[2] = pack_bf2(v[4], v[5]); o[3] = pack_bf2(v[6], v[7]);
                if constexpr (STATS) {
                    // squares of the ROUNDED values (what the next RMSNorm reads from memory); unit order u = 4 j + b: the four units of
                    // an m tile are consecutive, one running sum is live
                    static_assert(GE_ORDER == 1, "the statistic closes an m tile at b == 3");
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const float lo_ = bf_lo(o[d]), hi_ = bf_hi(o[d]);
                        ssq = fmaf(lo_, lo_, ssq);
                        ssq = fmaf(hi_, hi_, ssq);
                    }
                    if (b == 3) {
                        // the row's other three column groups sit in lanes l15 + 16, + 32, + 48
                        float t_ = ssq + __uint_as_float((uint32_t)__builtin_amdgcn_ds_swizzle((int)__float_as_uint(ssq), 0x401f));   // xor 16
                        t_ += __shfl_xor(t_, 32);
                        sst[j & 3] = t_;
                        // every lane group stores the same total to the same address (one instruction either way)
                        const float* sp_ = a.ss + ((int64_t)(n0 / 128 + wn) * a.ss_ld + m0 + 16 * j);
                        asm volatile("global_store_dword %0, %1, %2" :: "v"(ep_rsoff), "v"(sst[j & 3]), "s"(sp_) : "memory");
                        ssq = 0.f;
                    }
                }
#if GE_FULL
                // A store instruction costs the wave ~270 cycles whatever it carries (measured: 16 rows x 64 B and 16 rows x 32 B
                // alike) -- it is paid per row segment.  So two strips are stored together as WHOLE 128-byte lines, 8 rows per
                // instruction: lanes l15 < 8 and their partners l15 + 8 swap one quad (DPP row_ror:8), after which instruction 1
                // holds rows 0..7 x [strip b | strip b + 1] and instruction 2 rows 8..15.
                if ((u & 1) == 0) o_even = o;
                else {
                    const bool low = l15 < 8;
                    g_u32x4 &s1 = ost[(u - 1) & 7], &s2 = ost[u & 7];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t give = low ? o[d] : o_even[d];
                        const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0x128, 0xf, 0xf, false);
                        s1[d] = low ? o_even[d] : recv;
                        s2[d] = low ? recv : o[d];
                    }
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" GE_STPOL :: "v"(s1), "v"(ep_voff_f), "s"(yd), "s"(GE_SOFF(u - 1)) : "memory");
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" GE_STPOL :: "v"(s2), "v"(ep_voff_f), "s"(yd), "s"(GE_SOFF(u - 1) + ep_rows16 / 2) : "memory");
                    if (u >= 7) { asm volatile("" :: "v"(ost[(u + 1) & 7])); asm volatile("" :: "v"(ost[(u + 2) & 7])); }
                }
#else
                // A store reads its data registers when the memory pipeline gets to it, not at issue: the packed results rotate
                // through eight register quads (kept allocated by the empty asm)
                ost[u & 7] = o;
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" GE_STPOL :: "v"(ost[u & 7]), "v"(ep_voff), "s"(yd), "s"(GE_SOFF(u)) : "memory");
                if (u >= 7) asm volatile("" :: "v"(ost[(u + 1) & 7]));
#endif
                if (RES && u + GE_NR < 32) GE_LOAD(u + GE_NR);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" :: "v"(ost[k]));     // (the last eight quads stay distinct, too)
            if constexpr (STATS) asm volatile("" :: "v"(sst[0]), "v"(sst[1]), "v"(sst[2]), "v"(sst[3]));
            GR_STAMP(8);                                         // (accumulator reads, packing, stores)
#undef GE_LOAD
#undef GE_NR
#undef GE_SOFF
            }
            next_tile_origin();                                  // (the fetch cursor entered tile c_i + 1 two stages ago)
            GR_STAMP(9);
            asm volatile("s_nop 7" ::: "memory");              // accumulator writes -> the next tile's first MFMAs
            GR_STAMP(4);
        }
    }
    G_VMCNT(0);                                                  // (tail re-fetches: nothing may land after the LDS is released)
#if GR_PROFILE
    if (prof && lane == 0) {
        for (int k = 0; k < 12; ++k) ((float*)a.y)[k] = (float)tp[k];
        ((float*)a.y)[12] = (float)(n_my * nk);
        ((float*)a.y)[13] = (float)n_my;
    }
#endif
}

extern "C" int evo_linear_mfma_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y,
                                    int64_t M, int64_t N, int64_t K, void* stream) {
    return evo_linear_mfma_nf_bf16(x, w, bias, residual, y, nullptr, nullptr, 0, M, N, K, stream);
}

// The same dense layer with the RMSNorm around it folded in (gemmr_bf16_kernel, NF): `row_scale` [ceil(M / 256) * 256] fp32 multiplies the
// accumulators of row m before the bias (the consumer of a normalised stream: w is then W diag(g), no residual); `sumsq`
// [N / 128][ss_ld] fp32 (ss_ld >= ceil(M / 256) * 256) receives the sums of squares of the stored rows per 128-column strip (the
// producer of the stream: residual required).  Both need the persistent form's shape contract (K >= 128, operands < 4 GiB).
extern "C" int evo_linear_mfma_nf_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y,
                                       const float* row_scale, float* sumsq, int64_t ss_ld, int64_t M, int64_t N, int64_t K, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || N % GBN != 0 || K % GBK != 0 || N > 0x7fffffff / 2) return -1;
    if ((row_scale && (residual || sumsq)) || (sumsq && (!residual || ss_ld < (M + GBM - 1) / GBM * GBM))) return -1;
    if ((row_scale || sumsq) && !(K >= 2 * GBK && M * K * 2 < 0xffffffffll && N * K * 2 < 0xffffffffll)) return -1;
    GemmArgs a;
    a.rs = row_scale; a.ss = sumsq; a.ss_ld = ss_ld;
    a.x = (const unsigned char*)x; a.w = (const unsigned char*)w; a.bias = (const uint16_t*)bias;
    a.res = (const uint16_t*)residual; a.y = (uint16_t*)y;
    a.M = M; a.N = (int)N; a.K = (int)K;
    a.tiles_n = (int)(N / GBN);
    a.tiles_m = (int)((M + GBM - 1) / GBM);
    // raster width: the ~32 tiles an XCD runs at once cover group_m X panels x 32 / group_m W panels.  Measured on the four layer
    // shapes at M = 65,536 (tools/gemm_ab.py lib.so@G): N = 12,288 / 22,016: 8 is best (98.1 / 98.2 % of hipBLASLt against 97.6 /
    // 96.6 at 4, 88 at 16, 60 at 32); N = 4,096 (16 column tiles): 1-4 tie, 8 loses 1.5-2.5 %.
    const int group_m = (a.tiles_n >= 32 ? 8 : 4);
    a.group_m = group_m;
    const int64_t tiles = ((M + GBM - 1) / GBM) * a.tiles_n;
    if (tiles > 0x7fffffff) return -1;
    a.n_tiles = (int)tiles;
    const dim3 grid((unsigned)tiles), block(512);
    hipStream_t st = (hipStream_t)stream;
    static const int form = [] { const char* e = getenv("EVO_GEMM_FORM"); return e ? atoi(e) : 1; }();   // 1: persistent, 0: tile per workgroup
    if ((row_scale || sumsq) && form != 1) return -1;           // the folded norm exists in the persistent kernel only: never fall through to a launch that ignores it
    if (form == 1 && K >= 2 * GBK && M * K * 2 < 0xffffffffll && N * K * 2 < 0xffffffffll) {
        static const int n_cu = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            n &= ~7;
            return n < 8 ? 8 : n;
        }();
        const dim3 gridp((unsigned)n_cu), block4(256);
        if (row_scale) {
            if (bias) hipLaunchKernelGGL((gemmr_bf16_kernel<true, false, 0, false, 2>), gridp, block4, 0, st, a);
            else hipLaunchKernelGGL((gemmr_bf16_kernel<false, false, 0, false, 2>), gridp, block4, 0, st, a);
            return evo_launch_status();
        }
        if (sumsq) {
            if (bias) hipLaunchKernelGGL((gemmr_bf16_kernel<true, true, 0, false, 1>), gridp, block4, 0, st, a);
            else hipLaunchKernelGGL((gemmr_bf16_kernel<false, true, 0, false, 1>), gridp, block4, 0, st, a);
            return evo_launch_status();
        }
        if (bias && residual) hipLaunchKernelGGL((gemmr_bf16_kernel<true, true, 0>), gridp, block4, 0, st, a);
        else if (bias) hipLaunchKernelGGL((gemmr_bf16_kernel<true, false, 0>), gridp, block4, 0, st, a);
        else if (residual) hipLaunchKernelGGL((gemmr_bf16_kernel<false, true, 0>), gridp, block4, 0, st, a);
        else hipLaunchKernelGGL((gemmr_bf16_kernel<false, false, 0>), gridp, block4, 0, st, a);
        return evo_launch_status();
    }
    if (bias && residual) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, st, a);
    else if (bias) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, st, a);
    else if (residual) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, st, a);
    return evo_launch_status();
}

// y [M, N] = x . w^T (+ bias) (+ residual) with x in the BLOCKED layout the channel-stationary Hyena operator writes
// ([ceil(M / 128)][K / 16][128][16] bf16): the Hyena block's output projection [REF stripedhyena/model.py ParallelGatedConvBlock.forward:
// out_filter_dense].  M % 256 == 0 (the caller peels the BOS sliver), K % 64 == 0, K >= 128.
extern "C" int evo_linear_xblk_mfma_bf16(const void* x_blk, const void* w, const void* bias, const void* residual, void* y,
                                         int64_t M, int64_t N, int64_t K, void* stream) {
    return evo_linear_xblk_mfma_nf_bf16(x_blk, w, bias, residual, y, nullptr, 0, M, N, K, stream);
}

// ... and with the sums of squares of the stored rows (see evo_linear_mfma_nf_bf16; residual required when sumsq is given)
extern "C" int evo_linear_xblk_mfma_nf_bf16(const void* x_blk, const void* w, const void* bias, const void* residual, void* y,
                                            float* sumsq, int64_t ss_ld, int64_t M, int64_t N, int64_t K, void* stream) {
    if (sumsq && (!residual || ss_ld < M)) return -1;
    if (M <= 0 || M % GBM != 0 || N <= 0 || K <= 0 || N % GBN != 0 || K % GBK != 0 || K < 2 * GBK || N > 0x7fffffff / 2) return -1;
    if (M * K * 2 >= 0xffffffffll || N * K * 2 >= 0xffffffffll) return -1;
    GemmArgs a;
    a.x = (const unsigned char*)x_blk; a.w = (const unsigned char*)w; a.bias = (const uint16_t*)bias;
    a.res = (const uint16_t*)residual; a.y = (uint16_t*)y;
    a.ss = sumsq; a.ss_ld = ss_ld;
    a.M = M; a.N = (int)N; a.K = (int)K;
    a.tiles_n = (int)(N / GBN);
    a.tiles_m = (int)(M / GBM);
    a.group_m = a.tiles_n >= 32 ? 8 : 4;
    const int64_t tiles = (M / GBM) * a.tiles_n;
    if (tiles > 0x7fffffff) return -1;
    a.n_tiles = (int)tiles;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        n &= ~7;
        return n < 8 ? 8 : n;
    }();
    const dim3 gridp((unsigned)n_cu), block4(256);
    if (sumsq) {
        if (bias) hipLaunchKernelGGL((gemmr_bf16_kernel<true, true, 0, true, 1>), gridp, block4, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((gemmr_bf16_kernel<false, true, 0, true, 1>), gridp, block4, 0, (hipStream_t)stream, a);
        return evo_launch_status();
    }
    if (bias && residual) hipLaunchKernelGGL((gemmr_bf16_kernel<true, true, 0, true>), gridp, block4, 0, (hipStream_t)stream, a);
    else if (residual) hipLaunchKernelGGL((gemmr_bf16_kernel<false, true, 0, true>), gridp, block4, 0, (hipStream_t)stream, a);
    else if (bias) hipLaunchKernelGGL((gemmr_bf16_kernel<true, false, 0, true>), gridp, block4, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemmr_bf16_kernel<false, false, 0, true>), gridp, block4, 0, (hipStream_t)stream, a);
    return evo_launch_status();
}

// Gated MLP, first half: a[M, I] = gelu(x W1^T) * (x W2^T) in ONE launch of the persistent kernel -- the [M, 2 I] intermediate
// never reaches memory [REF stripedhyena/layers.py ParallelGatedMLP.forward].  w12g = the rows of [W1; W2] regrouped as blocks
// of 64: 32 rows of W1 followed by the same 32 rows of W2 (evo_amd/ops.py pack_gate_weights), 2 I rows in all.
extern "C" int evo_mlp_gate_mfma_bf16(const void* x, const void* w12g, void* a_out, int64_t M, int64_t I, int64_t K, void* stream) {
    return evo_mlp_gate_mfma_nf_bf16(x, nullptr, w12g, a_out, M, I, K, stream);
}

// ... with the post-mixer RMSNorm folded in: x = the residual stream itself, row_scale [ceil(M / 256) * 256] = 1 / (rms + eps) per row,
// w12g = the regrouped rows of [W1 diag(g); W2 diag(g)] (see evo_linear_mfma_nf_bf16)
extern "C" int evo_mlp_gate_mfma_nf_bf16(const void* x, const float* row_scale, const void* w12g, void* a_out, int64_t M, int64_t I, int64_t K,
                                         void* stream) {
    const int64_t N = 2 * I;
    if (M <= 0 || I <= 0 || K <= 0 || N % GBN != 0 || K % GBK != 0 || K < 2 * GBK || N > 0x7fffffff / 2) return -1;
    if (M * K * 2 >= 0xffffffffll || N * K * 2 >= 0xffffffffll) return -1;
    GemmArgs a;
    a.x = (const unsigned char*)x; a.w = (const unsigned char*)w12g; a.bias = nullptr; a.res = nullptr; a.y = (uint16_t*)a_out;
    a.rs = row_scale;
    a.M = M; a.N = (int)N; a.K = (int)K;
    a.tiles_n = (int)(N / GBN);
    a.tiles_m = (int)((M + GBM - 1) / GBM);
    a.group_m = (a.tiles_n >= 32 ? 8 : 4);
    const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
    if (tiles > 0x7fffffff) return -1;
    a.n_tiles = (int)tiles;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        n &= ~7;
        return n < 8 ? 8 : n;
    }();
    if (row_scale) hipLaunchKernelGGL((gemmr_bf16_kernel<false, false, 1, false, 2>), dim3((unsigned)n_cu), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemmr_bf16_kernel<false, false, 1>), dim3((unsigned)n_cu), dim3(256), 0, (hipStream_t)stream, a);
    return evo_launch_status();
}

// z^T = (x [Mp, K] . w [N, K]^T + bias [N])^T, stored in blocks of 256 positions, zt [Mp / 256][N][256] (element (feature c, position p) at
// ((p / 256) * N + c) * 256 + p % 256): the Hyena projection with a CHANNEL-MAJOR result for csrc/hyena_ct.hip --
// [REF stripedhyena/model.py ParallelGatedConvBlock.forward: projections].  The persistent kernel is launched with its operands
// SWAPPED (its "X" = w: the rows of the result are output features; its "W" = x: the columns are tokens), so a lane's eight
// consecutive output columns are eight consecutive TOKENS of one feature and the epilogue stores whole 128-byte lines of z^T as it
// does for y; the bias is indexed by row (MODE 3).  Bit-identical to evo_linear_mfma_bf16's result, transposed (same k order of
// the same products, one rounding).  Mp % 256 == 0 (the caller pads: every row of x is computed), N % 256 == 0, K % 64 == 0, K >= 128.
extern "C" int evo_linear_t_mfma_bf16(const void* x, const void* w, const void* bias, void* zt, int64_t Mp, int64_t N, int64_t K,
                                      void* stream) {
    return evo_linear_t_mfma_nf_bf16(x, nullptr, w, bias, zt, Mp, N, K, Mp, Mp, 0, stream);
}

// ... with the block's pre-norm folded in: x = the residual stream [x_rows, K] in (batch row, token) order, w = W diag(g), row_scale
// [x_rows] = 1 / (rms + eps) per row; position p = b Tm + t of z^T (Mp = B Tm positions, Tm % 256 == 0) is row p + b row_skip of x -- the
// tail form of z^T (HipOps.zt_layout: Tm = T - row_skip), whose tail tokens the caller projects separately.  row_scale == NULL:
// the plain launch (x_rows = Tm = Mp, row_skip = 0).
extern "C" int evo_linear_t_mfma_nf_bf16(const void* x, const float* row_scale, const void* w, const void* bias, void* zt, int64_t Mp, int64_t N,
                                         int64_t K, int64_t x_rows, int64_t Tm, int64_t row_skip, void* stream) {
    if (row_scale) {
        if (Tm <= 0 || Tm % GBN != 0 || Mp % Tm != 0 || row_skip < 0 || row_skip > 0x7fff || x_rows < Mp + (Mp / Tm - 1) * row_skip
            || x_rows * K * 2 >= 0xffffffffll) return -1;
    } else if (x_rows != Mp || Tm != Mp || row_skip != 0) return -1;
    if (Mp <= 0 || N <= 0 || K <= 0 || Mp % GBN != 0 || N % GBM != 0 || K % GBK != 0 || K < 2 * GBK || Mp > 0x7fffffff / 2) return -1;
    if (Mp * K * 2 >= 0xffffffffll || N * K * 2 >= 0xffffffffll) return -1;
    GemmArgs a;
    a.x = (const unsigned char*)w; a.w = (const unsigned char*)x; a.bias = (const uint16_t*)bias; a.res = nullptr; a.y = (uint16_t*)zt;
    a.M = N; a.N = (int)Mp; a.K = (int)K;
    a.rs = row_scale; a.w_rows = x_rows; a.tm_tiles = (int)(Tm / GBN); a.row_skip = (int)row_skip;
    a.tiles_n = (int)(Mp / GBN);
    a.tiles_m = (int)(N / GBM);
    a.group_m = (a.tiles_m >= 32 ? 8 : 4);      // (column tiles per raster group: see tile_origin, MODE 3)
    const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
    if (tiles > 0x7fffffff) return -1;
    a.n_tiles = (int)tiles;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        n &= ~7;
        return n < 8 ? 8 : n;
    }();
    if (row_scale) {
        if (bias) hipLaunchKernelGGL((gemmr_bf16_kernel<true, false, 3, false, 2>), dim3((unsigned)n_cu), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((gemmr_bf16_kernel<false, false, 3, false, 2>), dim3((unsigned)n_cu), dim3(256), 0, (hipStream_t)stream, a);
        return evo_launch_status();
    }
    if (bias) hipLaunchKernelGGL((gemmr_bf16_kernel<true, false, 3>), dim3((unsigned)n_cu), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemmr_bf16_kernel<false, false, 3>), dim3((unsigned)n_cu), dim3(256), 0, (hipStream_t)stream, a);
    return evo_launch_status();
}
