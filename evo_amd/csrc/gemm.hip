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

extern "C" int evo_linear_mfma_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y,
                                    int64_t M, int64_t N, int64_t K, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || N % GBN != 0 || K % GBK != 0 || N > 0x7fffffff / 2) return -1;
    GemmArgs a;
    a.x = (const unsigned char*)x; a.w = (const unsigned char*)w; a.bias = (const uint16_t*)bias;
    a.res = (const uint16_t*)residual; a.y = (uint16_t*)y;
    a.M = M; a.N = (int)N; a.K = (int)K;
    a.tiles_n = (int)(N / GBN);
    a.tiles_m = (int)((M + GBM - 1) / GBM);
    static const int group_m = [] {                              // raster width: 4 measured best (profiles/r01_gemm_notes.txt)
        const char* e = getenv("EVO_GEMM_GROUP_M");
        const int g = e ? atoi(e) : 4;
        return g < 1 ? 1 : g;
    }();
    a.group_m = group_m;
    const int64_t tiles = ((M + GBM - 1) / GBM) * a.tiles_n;
    if (tiles > 0x7fffffff) return -1;
    a.n_tiles = (int)tiles;
    const dim3 grid((unsigned)tiles), block(512);
    hipStream_t st = (hipStream_t)stream;
    if (bias && residual) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, st, a);
    else if (bias) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, st, a);
    else if (residual) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, st, a);
    return evo_launch_status();
}
