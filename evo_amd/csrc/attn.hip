// Causal multi-head attention forward for gfx950, head dim 128, bf16 in / fp32 softmax+accumulate /
// bf16 out -- the kernel that stands where the reference calls FlashAttention-2.
//
// Workgroup = 4 waves = 128 consecutive queries of one (batch, head); wave w owns 32 query rows.  Keys
// and values stream through LDS in tiles of 64 keys (next tile's global loads are issued before the
// current tile's math and written to LDS after it).  Both GEMMs run "swapped" on
// v_mfma_f32_32x32x16_bf16 so that the accumulator COLUMN is the query:
//     S^T[key][q] = K . Q^T      (A = K tile from LDS,   B = Q fragment held in registers)
//     O^T[d][q]   = V^T . P^T    (A = V^T tile from LDS, B = P^T built in-register from S^T)
// With C/D layout col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5), every lane holds 16 of the 32
// scores of ITS query per 32-key tile, so the online-softmax row reductions are 31 in-lane ops + one
// exchange with lane^32, the rescale of O^T is lane-local, and P never goes through LDS.  The key
// order inside an MFMA k-step is free as long as A and B agree, so P^T takes S^T's registers as they
// are and the V^T fragment is fetched in the matching order (two 4-key groups per lane).
//
// LDS images (both conflict-free for their reads):
//   Ks [64 keys][128 d] bf16, 16-byte slot XOR-swizzled by (key & 15)      -> ds_read_b128 A fragments
//   Vt [128 d][64 keys] bf16, 8-byte  slot XOR-swizzled by ((d >> 1) & 15) -> ds_read_b64  A fragments
// Entry point and reference citation: include/evo_mi355x.h.
#include <stdlib.h>
#include "common.h"
#include "../../include/evo_mi355x.h"

#include "attn_common.h"

// DECODE = true: one query row per (batch, head); blockIdx.x is a SPLIT of the key range ("flash-decoding"):
// every split streams its share of the KV cache and leaves an unnormalised partial (O, m, l) that
// attn_decode_combine_kernel merges.  The key count may come from device memory (dyn_pos), so the launch is
// position-independent and can sit inside a captured hipGraph.
template <bool DECODE>
__global__ __launch_bounds__(256, DECODE ? 1 : 2) void attn_fwd_kernel(AttnArgs a) {
    // two stages of {Ks 16 KiB, Vt 16 KiB}: tile t+1 is written while tile t is consumed -> one barrier per tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * KB * DH * 2];
    constexpr int STAGE_B = 2 * KB * DH * 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;

    int64_t Tk_ = a.Tk, q_pos0_ = a.q_pos0;
    if (DECODE && a.dyn_pos) { q_pos0_ = a.dyn_pos[blockIdx.z]; Tk_ = q_pos0_ + a.Tq; }
    int qb = 0, head = blockIdx.y, bat = blockIdx.z;
    if (!DECODE) attn_block_map(a, qb, head, bat);
    const int64_t q0 = (int64_t)qb * QB;

    const uint16_t* qp = a.q + bat * a.q_sb + head * a.q_sh;
    const uint16_t* kp = a.k + bat * a.k_sb + head * a.k_sh;
    const uint16_t* vp = a.v + bat * a.v_sb + head * a.v_sh;

    // ---- Q fragment: this lane's query row, 8 k-steps x 16 bytes -------------------------------------
    const int64_t qrow = q0 + wave * 32 + l31;
    const int64_t qrow_c = qrow < a.Tq ? qrow : a.Tq - 1;
    uint4 qf[8];
    {
        const uint4* qr = (const uint4*)(qp + qrow_c * a.q_st);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = qr[2 * ks + half];
    }

    // ---- key range of this workgroup / wave -------------------------------------------------------------
    int64_t q_last = q0 + QB - 1;
    if (q_last > a.Tq - 1) q_last = a.Tq - 1;
    int64_t max_key = q_last + q_pos0_;
    if (max_key > Tk_ - 1) max_key = Tk_ - 1;
    const int n_tiles = (int)(max_key / KB) + 1;
    int tile_begin = 0, tile_end = n_tiles;
    if (DECODE) {
        const int per = (n_tiles + a.n_splits - 1) / a.n_splits;
        tile_begin = (int)blockIdx.x * per;
        tile_end = tile_begin + per < n_tiles ? tile_begin + per : n_tiles;
    }
    const int64_t wq_first = q0 + wave * 32 + q_pos0_;           // position limit of the wave's first row
    const int64_t wq_last = wq_first + 31;
    const int64_t my_lim = qrow + q_pos0_;                        // this lane's query sees keys <= my_lim

    // ---- staging maps ------------------------------------------------------------------------------------
    const int kc = tid & 15;             // K: 16-byte chunk of the row
    const int kr = tid >> 4;             // K: row (key) within each group of 16
    const int vc = ((lane >> 4) << 2) | (lane & 3);        // V: 16-byte chunk (d = 8vc..8vc+7)
    const int vm = 4 * wave + ((lane >> 2) & 3);           // V: key quad (keys 4vm..4vm+3)

    // loads of one tile: wave-uniform tile base (SGPRs) + 32-bit per-lane byte offset; rows past the end
    // of a ragged last tile are clamped to the last valid key (they are masked out of the softmax)
    const uint32_t kst_b = (uint32_t)(a.k_st * 2), vst_b = (uint32_t)(a.v_st * 2);
    uint4 kreg0, kreg1, kreg2, kreg3, vreg0, vreg1, vreg2, vreg3;   // named (not arrays): must stay in VGPRs
#define ATTN_LOAD_ONE(KR, VR, I, KB_PTR, VB_PTR, RELMAX)                                   \
    {                                                                                      \
        const int kk = min(kr + 16 * (I), (RELMAX));                                       \
        const int vk = min(4 * vm + (I), (RELMAX));                                        \
        KR = *(const uint4*)((KB_PTR) + (uint32_t)kk * kst_b + kc * 16);                   \
        VR = *(const uint4*)((VB_PTR) + (uint32_t)vk * vst_b + vc * 16);                   \
    }
#define ATTN_ISSUE_LOADS(TILE)                                                             \
    {                                                                                      \
        const int64_t k0_ = (int64_t)(TILE) * KB;                                          \
        const int64_t left_ = Tk_ - 1 - k0_;                                               \
        const int rel_max_ = left_ < KB - 1 ? (int)left_ : KB - 1;                         \
        const unsigned char* kb_ = (const unsigned char*)(kp + k0_ * a.k_st);              \
        const unsigned char* vb_ = (const unsigned char*)(vp + k0_ * a.v_st);              \
        ATTN_LOAD_ONE(kreg0, vreg0, 0, kb_, vb_, rel_max_)                                 \
        ATTN_LOAD_ONE(kreg1, vreg1, 1, kb_, vb_, rel_max_)                                 \
        ATTN_LOAD_ONE(kreg2, vreg2, 2, kb_, vb_, rel_max_)                                 \
        ATTN_LOAD_ONE(kreg3, vreg3, 3, kb_, vb_, rel_max_)                                 \
    }
#define ATTN_KWRITE(KR, I)                                                                 \
    {                                                                                      \
        const int key_ = kr + 16 * (I);                                                    \
        *(uint4*)(Ks_w + key_ * 256 + ((kc ^ (key_ & 15)) << 4)) = KR;                     \
    }
    // 4 keys x 2 d-values (one dword column J of the 4 loaded rows) -> two 8-byte V^T entries
#define ATTN_VWRITE(J, C)                                                                  \
    {                                                                                      \
        const int d_even = 8 * vc + 2 * (J), d_odd = d_even + 1;                           \
        uint2 ev, od;                                                                      \
        ev.x = (vreg0.C & 0xffffu) | (vreg1.C << 16);                                      \
        ev.y = (vreg2.C & 0xffffu) | (vreg3.C << 16);                                      \
        od.x = (vreg0.C >> 16) | (vreg1.C & 0xffff0000u);                                  \
        od.y = (vreg2.C >> 16) | (vreg3.C & 0xffff0000u);                                  \
        *(uint2*)(Vt_w + d_even * 128 + ((vm ^ ((d_even >> 1) & 15)) << 3)) = ev;          \
        *(uint2*)(Vt_w + d_odd * 128 + ((vm ^ ((d_odd >> 1) & 15)) << 3)) = od;            \
    }
#define ATTN_WRITE_LDS(STAGE)                                                              \
    {                                                                                      \
        unsigned char* Ks_w = smem + (STAGE) * STAGE_B;                                    \
        unsigned char* Vt_w = Ks_w + KB * DH * 2;                                          \
        ATTN_KWRITE(kreg0, 0) ATTN_KWRITE(kreg1, 1) ATTN_KWRITE(kreg2, 2) ATTN_KWRITE(kreg3, 3) \
        ATTN_VWRITE(0, x) ATTN_VWRITE(1, y) ATTN_VWRITE(2, z) ATTN_VWRITE(3, w)            \
    }

    // V^T row offset of this lane per 32-row d-tile.  Laundered through an empty asm so hipcc keeps the
    // two 8-byte fragment reads as ds_read_b64 (conflict-free here) instead of fusing rows 4 KiB apart
    // into ds_read2st64_b64, which is serviced in 16-lane groups mod 32 banks (2-way conflict, half rate).
    uint32_t vt_row[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        vt_row[dt] = (32 * dt + l31) * 128;
        asm volatile("" : "+v"(vt_row[dt]));
    }

    f32x16_t oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY;     // running max (scaled, log2 domain)
    float l_run = 0.f;           // this lane's share of the running denominator

    if (tile_begin < tile_end) {
        ATTN_ISSUE_LOADS(tile_begin)
        ATTN_WRITE_LDS(0)
        if (tile_begin + 1 < tile_end) ATTN_ISSUE_LOADS(tile_begin + 1)
    }
    __syncthreads();
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int64_t k0 = (int64_t)tile * KB;
        const int cur = (tile - tile_begin) & 1;
        // stage cur^1 was last read before the previous barrier: refill it with tile+1 (its global loads were
        // issued one iteration ago) and put tile+2's loads in flight, all under this tile's math
        if (tile + 1 < tile_end) {
            ATTN_WRITE_LDS(cur ^ 1)
            if (tile + 2 < tile_end) ATTN_ISSUE_LOADS(tile + 2)
        }
        const unsigned char* Ks = smem + cur * STAGE_B;
        const unsigned char* Vt = Ks + KB * DH * 2;

        if (k0 <= wq_last) {                       // wave-uniform: at least one visible key
            // ---- S^T = K . Q^T -----------------------------------------------------------------------------
            f32x16_t sacc[2];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
                const int key = 32 * kt + l31;
                const unsigned char* krow = Ks + key * 256;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const uint4 kf = *(const uint4*)(krow + (((2 * ks + half) ^ (key & 15)) << 4));
                    sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf), as_frag(qf[ks]), sacc[kt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            // ---- online softmax (this lane = one query row, 32 of the tile's 64 keys) --------------------
            const bool need_mask = (k0 + KB - 1 > wq_first) || (k0 + KB > Tk_);
            float tmax = -INFINITY;
            if (need_mask) {                       // 32-bit tile-relative limits (diagonal / ragged tiles only)
                const int64_t lim64 = (my_lim < Tk_ - 1 ? my_lim : Tk_ - 1) - k0;
                const int lim = lim64 > 63 ? 63 : (lim64 < -1 ? -1 : (int)lim64);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kidx = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
                        sacc[kt][r] = kidx <= lim ? sacc[kt][r] : -INFINITY;
                    }
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[kt][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax * a.scale_log2);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);          // m_run = -inf -> 0
            // p = 2^(s*c - m): scale/subtract and the row sum run two scores per instruction (v_pk_fma_f32 /
            // v_pk_add_f32); only the exponentials are scalar
            const f32x2_t c2 = {a.scale_log2, a.scale_log2}, nm2 = {-m_use, -m_use};
            f32x2_t psum2 = {0.f, 0.f};
            uint32_t pk[2][8];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2_t s2 = {sacc[kt][r], sacc[kt][r + 1]};
                    const f32x2_t e2 = __builtin_elementwise_fma(s2, c2, nm2);
                    const f32x2_t p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
                    psum2 += p2;
                    pk[kt][r >> 1] = pack_bf2(p2[0], p2[1]);
                }
            const float psum = psum2[0] + psum2[1];
            l_run = fmaf(l_run, alpha, psum);
            const float m_run_prev = m_run;
            m_run = m_new;
            if (__any(m_new > m_run_prev)) {          // wave-uniform: late in a long row nothing moves any more
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[dt] = oacc[dt] * alpha;
            }
            // ---- O^T += V^T . P^T ----------------------------------------------------------------------------
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    uint4 pf;
                    pf.x = pk[kt][4 * u]; pf.y = pk[kt][4 * u + 1]; pf.z = pk[kt][4 * u + 2]; pf.w = pk[kt][4 * u + 3];
                    const int m1 = 8 * kt + 4 * u + half;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const int d = 32 * dt + l31;
                        const int sw = (d >> 1) & 15;
                        const unsigned char* vrow = Vt + vt_row[dt];
                        const uint2 va = *(const uint2*)(vrow + ((m1 ^ sw) << 3));
                        const uint2 vb = *(const uint2*)(vrow + (((m1 + 2) ^ sw) << 3));
                        uint4 vf;
                        vf.x = va.x; vf.y = va.y; vf.z = vb.x; vf.w = vb.y;
                        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(vf), as_frag(pf), oacc[dt], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }

    // ---- epilogue: normalise and store O[q][d] -----------------------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (DECODE) {                                   // partial result of this split for query row 0
        if (wave == 0 && l31 == 0) {
            const int64_t slot = ((int64_t)bat * a.H + head) * a.n_splits + blockIdx.x;
            float* po = a.part_o + slot * DH;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(po + 32 * dt + 8 * g + 4 * half) =
                        make_float4(oacc[dt][4 * g], oacc[dt][4 * g + 1], oacc[dt][4 * g + 2], oacc[dt][4 * g + 3]);
            if (half == 0) { a.part_ml[slot * 2] = m_run; a.part_ml[slot * 2 + 1] = l_tot; }
        }
        return;
    }
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qrow < a.Tq) {
        uint16_t* orow = a.o + ((int64_t)(bat * a.Tq + qrow) * a.H + head) * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack_bf2(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv);
                w.y = pack_bf2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
                *(uint2*)(orow + 32 * dt + 8 * g + 4 * half) = w;
            }
    }
}

// ====================================================================================================
// Pipelined prefill kernel (query ranges longer than one 128-row block).  Same math and MFMA fragment layouts as
// attn_fwd_kernel<false>; what changes is everything around them, following what the counters and the probes said about
// that kernel: per 64-key tile it spends 255 VALU instructions -- 77 staging K/V through registers,
// 48 swizzle address adds -- in phases that do not overlap its 32 MFMAs.
//   * a workgroup is 8 waves = 256 query rows (256 flop per K/V byte); K/V tiles arrive through a 4-stage LDS ring filled
//     by asynchronous global->LDS DMA three tiles ahead -- no staging registers, no transpose in registers.  The DMA is
//     inline asm (a DMA hipcc can see gets s_waitcnt vmcnt(0) in front of every later LDS read), waits are counted
//     (never a drain) and the pieces of a tile are issued ONE AT A TIME between the P.V MFMAs: all of them in a burst
//     behind the barrier block every wave on the vector-memory issue queue (1150 -> 835 TFLOP/s; spread: ~1010);
//   * both tiles are ROW-MAJOR with padded rows, so every fragment address is a per-lane base + an immediate:
//       Kp [64 keys][272 B]  -> QK^T A fragments by ds_read_b128, conflict-free (16 keys -> 16 bank quads)
//       Vp [64 keys][256 B]  -> P.V  A fragments by ds_read_b64_tr_b16 (hardware 4x4 transpose: the lane that
//                               points at row i/4, columns 4(i%4).. of a [4 keys][16 d] block receives the 4
//                               keys of column i); un-padded, 64-byte blocks XOR-swizzled by (row & 3) on the
//                               DMA's source side, so the read keeps per-d-tile bases + immediates;
//   * software pipelining INSIDE a wave, in program order: 16 x {K read, QK^T(t+1) MFMA, exp/sum/pack of two
//     scores of tile t} then 16 x {2 transpose reads, P.V(t) MFMA, one DMA piece every third}; masking (diagonal /
//     ragged tiles), the row max and the exact O rescale sit outside those blocks.
// Measured against attn_fwd_kernel<false>: 1.06-1.09 vs 0.87-0.89 PFLOP/s at T = 131,073, 0.82 vs 0.73 at 8 x 8,193;
// bit-reproducible and bit-identical across query offsets / strides like the 128-row kernel.
#define PQB 256
#define PK_ROW 272
#define PV_ROW 256                       // V rows un-padded: 64-byte block b of row r is stored at block b ^ (r & 3)
#define PK_STAGE (KB * PK_ROW)          // 17,408 B = 17 DMA pieces of 1 KiB
#define PV_STAGE (KB * PV_ROW)          // 16,384 B = 16 DMA pieces
#define P_STAGE (PK_STAGE + PV_STAGE)   // 33,792 B
#define P_NSTG 4                        // 135,168 B of LDS: one workgroup per CU
#define P_NDMA_K 17
#define P_NDMA (17 + 16)
#ifndef P_VD
#define P_VD 4                          // V^T fragment prefetch distance, in MFMAs
#endif

typedef short tr_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr_s16x4* lds_tr_ptr_t;
typedef __attribute__((address_space(3))) void* lds_void_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_void_ptr_t;

// fmaxf on MFMA outputs makes hipcc emit a canonicalising v_max per operand; the raw instruction is what is wanted
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // volatile: stays behind the nop fence
    return r;
}

// "at most K tiles' worth of this wave's DMA pieces still outstanding" (np = 4 or 5 pieces per tile per wave)
#define P_WAIT(K)                                                                                    \
    do {                                                                                             \
        if (np == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (K)) : "memory");                  \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (K)) : "memory");                          \
    } while (0)
#define P_BARRIER() asm volatile("s_barrier" ::: "memory")

__global__ __launch_bounds__(512, 1) void attn_fwd_pipe_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[P_NSTG * P_STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int np = wave + 32 < P_NDMA ? 5 : 4;                          // DMA pieces of this wave per tile

    const int64_t Tk_ = a.Tk, q_pos0_ = a.q_pos0;
    int qb, head, bat;
    attn_block_map(a, qb, head, bat);
    const int64_t q0 = (int64_t)qb * PQB;
    const uint16_t* qp = a.q + bat * a.q_sb + head * a.q_sh;
    const unsigned char* kp = (const unsigned char*)(a.k + bat * a.k_sb + head * a.k_sh);
    const unsigned char* vp = (const unsigned char*)(a.v + bat * a.v_sb + head * a.v_sh);
    const int64_t kst_b = a.k_st * 2, vst_b = a.v_st * 2;

    const int64_t qrow = q0 + wave * 32 + l31;
    const int64_t qrow_c = qrow < a.Tq ? qrow : a.Tq - 1;
    uint4 qf[8];
    {
        const uint4* qr = (const uint4*)(qp + qrow_c * a.q_st);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = qr[2 * ks + half];
    }
    int64_t q_last = q0 + PQB - 1;
    if (q_last > a.Tq - 1) q_last = a.Tq - 1;
    int64_t max_key = q_last + q_pos0_;
    if (max_key > Tk_ - 1) max_key = Tk_ - 1;
    const int n_tiles = (int)(max_key / KB) + 1;
    const int64_t wq_first = q0 + wave * 32 + q_pos0_;
    const int64_t my_lim = qrow + q_pos0_;

    // ---- DMA plan: the 37 one-KiB pieces of a (K, V) tile are dealt round-robin to the 8 waves ------------------
    // piece j < 17 -> K bytes [j KiB, (j+1) KiB) of the stage; piece j >= 17 -> V bytes.  Per lane: the (row, column)
    // its 16 bytes belong to; lanes that fall into row padding fetch the row's first granule (never read back).
    // d_off[jj]: this lane's byte offset inside the (K or V) tile for its jj-th piece (row * row stride + column; pad
    // lanes re-fetch the row's first granule).
    uint32_t d_off[5];
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
        const int j = wave + 8 * jj;
        const bool is_k = j < P_NDMA_K;
        const int pos = (is_k ? j : j - P_NDMA_K) * 1024 + 16 * lane;
        const int rowb = is_k ? PK_ROW : PV_ROW;
        const int r = pos / rowb;
        int c = pos - r * rowb;
        if (is_k) c = c < 256 ? c : 0;                             // K: pad lanes re-fetch the row's first granule
        else c = ((((c >> 6) ^ r) & 3) << 6) | (c & 63);           // V: the LDS slot's block holds source block b ^ (r & 3)
        d_off[jj] = (uint32_t)r * (uint32_t)(is_k ? kst_b : vst_b) + (uint32_t)c;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // Buffer descriptors of one tile: base = first row of the tile, num_records = bytes up to the end of the last VALID
    // key (0 past the end of the sequence).  The hardware bounds check returns zeros for everything beyond -- the ragged
    // last tile needs no clamping, and a tile that does not exist costs a DMA of zeros into a stage nobody reads, so
    // every trip of the tile loop issues the same instructions (branch-free blocks, constant vmcnt counts).
    typedef int srd_t __attribute__((ext_vector_type(4)));
    auto tile_srd = [&](const unsigned char* base, int64_t st_b, int tile) {
        const int64_t k0 = (int64_t)tile * KB;
        int64_t rows = Tk_ - k0;
        rows = rows > KB ? KB : rows;
        const uint64_t a64 = (uint64_t)(base + k0 * st_b);
        srd_t d;
        d[0] = (int)(uint32_t)a64;
        d[1] = (int)(uint32_t)(a64 >> 32);
        d[2] = rows > 0 ? (int)((rows - 1) * st_b + 256) : 0;
        d[3] = 0x00020000;
        return d;
    };
    // One DMA piece (1 KiB).  Inline asm (a DMA the compiler can see makes it drain vmcnt before every later LDS read),
    // and issued ONE AT A TIME between MFMAs: all of a tile's pieces in a burst behind the barrier block every wave on the
    // vector-memory issue queue (1150 -> 835 TFLOP/s).
#define P_DMA_PIECE(KSRD, VSRD, STAGE_LDS, JJ)                                                                \
    {                                                                                                         \
        const int j_ = wave + 8 * (JJ);                                                                       \
        if ((JJ) < 4 || j_ < P_NDMA) {                              /* (only piece 4 of waves 1..7 is absent) */ \
            const srd_t d_ = j_ < P_NDMA_K ? (KSRD) : (VSRD);                                                 \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"             \
                         ::"s"((STAGE_LDS) + j_ * 1024), "v"(d_off[JJ]), "s"(d_) : "memory", "m0");           \
        }                                                                                                     \
    }
    auto dma_tile = [&](int tile) {
        const srd_t ks_ = tile_srd(kp, kst_b, tile), vs_ = tile_srd(vp, vst_b, tile);
        const uint32_t st_ = lds0 + (tile & (P_NSTG - 1)) * P_STAGE;
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) P_DMA_PIECE(ks_, vs_, st_, jj);
    };

    // per-lane fragment bases (everything else is an immediate)
    const uint32_t k_rd = (uint32_t)(l31 * PK_ROW + half * 16);
    // V^T fragment addresses: row r0 = (lane & 15) / 4 + 4 * half (+ 16 g, + 8: immediates), 64-byte block dt ^ (r0 & 3)
    // (one base per d tile), 8 bytes at 32 * ((lane >> 4) & 1) + 8 * (lane & 3) inside the block.  The eight rows a
    // ds_read_b64_tr_b16 touches fall on four distinct 64-byte bank windows, two rows each: 512 B in two passes.
    uint32_t v_rd[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const int r0 = ((lane & 15) >> 2) + 4 * half;
        v_rd[dt] = (uint32_t)(PK_STAGE + r0 * PV_ROW + ((dt ^ (r0 & 3)) << 6) + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    }

    f32x16_t oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16_t s_cur[2], s_nxt[2];

    auto row_max = [&](const f32x16_t (&s)[2]) {
        float t = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) t = max3_raw(t, s[kt][r], s[kt][r + 1]);
        return t;
    };

    // ---- prologue: three tiles in flight, then S(0) = QK^T of tile 0 ----------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the Q fragment loads: keep the counted waits to DMA only
    dma_tile(0);
    dma_tile(1);                                           // (tiles past the end: zero-length descriptors)
    dma_tile(2);
    P_WAIT(2);
    P_BARRIER();
    {
        const unsigned char* kb = smem + k_rd;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_cur[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint4 kf = *(const uint4*)(kb + kt * (32 * PK_ROW) + ks * 32);
                s_cur[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf), as_frag(qf[ks]), s_cur[kt], 0, 0, 0);
            }
        }
    }
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // one tile; the score tiles ping-pong between the two trips of the unrolled loop below (no 32-register copy)
    auto tile_body = [&](f32x16_t (&s_cur)[2], f32x16_t (&s_nxt)[2], const int tile) {
        const int64_t k0 = (int64_t)tile * KB;
        // tile+1 must have landed (this wave's pieces: counted wait; everyone's: barrier).  The same barrier says every
        // wave is done with tile-1, whose ring slot tile+3 now takes.
        P_WAIT(1);                                         // every trip issues one tile's worth of pieces
        P_BARRIER();
        const srd_t ksrd3 = tile_srd(kp, kst_b, tile + 3), vsrd3 = tile_srd(vp, vst_b, tile + 3);
        const uint32_t st3 = lds0 + ((tile + 3) & (P_NSTG - 1)) * P_STAGE;

        // diagonal / ragged tiles: mask and redo the row max (wave-uniform, rare)
        if ((k0 + KB - 1 > wq_first) || (k0 + KB > Tk_)) {
            const int64_t lim64 = (my_lim < Tk_ - 1 ? my_lim : Tk_ - 1) - k0;
            const int lim = lim64 > 63 ? 63 : (lim64 < -1 ? -1 : (int)lim64);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kidx = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
                    s_cur[kt][r] = kidx <= lim ? s_cur[kt][r] : -INFINITY;
                }
        }
        // row max of the (masked) tile.  (Folding these 16 v_max3 under the P.V MFMAs of the previous tile made the
        // result irreproducible from run to run -- the hand-written v_max3 then sits next to MFMAs whose hazards the
        // compiler does not model for inline asm; here it costs ~1 % and is exact.)
        // XDL-write -> VALU-read hazard: the v_max3 below are inline asm, for which hipcc pads nothing, and on the first
        // trip (non-diagonal first tile) only SALU / waitcnt / barrier instructions separate them from the prologue's
        // QK^T MFMAs.  24 wait states cover the 32x32x16 result latency for every register of the tile; the fences keep
        // the compiler from moving an MFMA below them.  (Steady state: the producers are a whole P.V phase back.)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float tmax = row_max(s_cur);
        // running max and the (rare, exact) O rescale -- outside the pipelined blocks, which therefore have no branch
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax * a.scale_log2);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        if (__any(m_new > m_run)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] = oacc[dt] * alpha;
        }
        m_run = m_new;

        // ---- phase 1: 16 x { K fragment read (2 ahead) | QK^T(tile+1) MFMA | exp/sum/pack of 2 scores of tile } -------
        // program order IS the schedule (sched_barrier after every chunk): a 32x32x16 MFMA holds the matrix pipe for 32
        // cycles, which covers the chunk's 5 VALU + 1 LDS instructions
        const bool more = tile + 1 < n_tiles;              // last trip: recompute a dummy tile from a resident stage
        const unsigned char* kb = smem + ((more ? tile + 1 : tile) & (P_NSTG - 1)) * P_STAGE + k_rd;
        const float nm = -m_use;
        float psum_a = 0.f, psum_b = 0.f;
        uint32_t pk[2][8];
        // K fragments: the 8 of the first 32-key half up front; each register is refilled with the matching fragment of
        // the second half right after its MFMA (>= 8 MFMA slots of slack: LDS latency under 8 waves of traffic is
        // several hundred cycles)
        uint4 kf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[ks] = *(const uint4*)(kb + ks * 32);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kt = i >> 3, ks = i & 7;
            s_nxt[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf[ks]), as_frag(qf[ks]),
                                                                ks == 0 ? zero16 : s_nxt[kt], 0, 0, 0);     // C = inline 0
            if (kt == 0) kf[ks] = *(const uint4*)(kb + 32 * PK_ROW + ks * 32);
            {
                // scalar f32 on purpose: beside MFMAs a v_pk_fma_f32 / v_pk_add_f32 costs more than the two scalar ops
                const int r = 2 * ks;
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[kt][r], a.scale_log2, nm));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[kt][r + 1], a.scale_log2, nm));
                // (pinned with an EMPTY asm: left to itself hipcc keeps all 32 p values alive and sums them in one dependent
                //  chain after the pipelined blocks; a hand-written v_add would read the v_exp result without the wait state
                //  the compiler inserts for its own instructions.)
                psum_a += p0;
                psum_b += p1;
                asm volatile("" : "+v"(psum_a), "+v"(psum_b));
                pk[kt][ks] = pack_bf2(p0, p1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run = fmaf(l_run, alpha, psum_a + psum_b);

        // ---- phase 2: 16 x { V^T fragment transpose-reads (1 ahead) | P.V(tile) MFMA | row max of 2 scores of tile+1 } ---
        const unsigned char* vb = smem + (tile & (P_NSTG - 1)) * P_STAGE;
        // V^T fragments P_VD MFMAs ahead (ring of P_VD + 1 pairs)
        tr_s16x4 va[P_VD + 1], vc[P_VD + 1];
#pragma unroll
        for (int j = 0; j < P_VD; ++j) {
            const unsigned char* pn = vb + v_rd[j & 3] + (4 * (j >> 2)) * (4 * PV_ROW);
            va[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_t)pn);
            vc[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_t)(pn + 2 * (4 * PV_ROW)));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j + P_VD < 16) {
                const int g = (j + P_VD) >> 2, dtn = (j + P_VD) & 3;       // g = 2*kt + u  ->  key quads 4g + half (+2)
                const unsigned char* pn = vb + v_rd[dtn] + (4 * g) * (4 * PV_ROW);
                va[(j + P_VD) % (P_VD + 1)] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_t)pn);
                vc[(j + P_VD) % (P_VD + 1)] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_t)(pn + 2 * (4 * PV_ROW)));
            }
            const int g = j >> 2, dt = j & 3, kt = g >> 1, u = g & 1;
            uint4 pf;
            pf.x = pk[kt][4 * u]; pf.y = pk[kt][4 * u + 1]; pf.z = pk[kt][4 * u + 2]; pf.w = pk[kt][4 * u + 3];
            const uint2 ua = __builtin_bit_cast(uint2, va[j % (P_VD + 1)]), ub = __builtin_bit_cast(uint2, vc[j % (P_VD + 1)]);
            uint4 vf;
            vf.x = ua.x; vf.y = ua.y; vf.z = ub.x; vf.w = ub.y;
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(vf), as_frag(pf), oacc[dt], 0, 0, 0);
            if ((j % 3) == 1) P_DMA_PIECE(ksrd3, vsrd3, st3, j / 3);       // chunks 1, 4, 7, 10, 13 -> pieces 0..4
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int tile = 0; tile < n_tiles; tile += 2) {
        tile_body(s_cur, s_nxt, tile);
        if (tile + 1 < n_tiles) tile_body(s_nxt, s_cur, tile + 1);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing (empty) DMA pieces: nothing may land after exit
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qrow < a.Tq) {
        uint16_t* orow = a.o + ((int64_t)(bat * a.Tq + qrow) * a.H + head) * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack_bf2(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv);
                w.y = pack_bf2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
                *(uint2*)(orow + 32 * dt + 8 * g + 4 * half) = w;
            }
    }
}

// ---- decode attention, streaming form.  One query row per (batch, head) makes this a bandwidth problem: the KV cache is read
// once (4 * H * 128 bytes per key) and nothing is reused.  The MFMA kernel above (DECODE = true) moves every tile global ->
// VGPR -> LDS -> fragments behind a workgroup barrier for ONE useful row of its 32-row tiles and tops out at 2.1-2.9 TB/s
// (8 k ... 131 k keys).  Here a WAVE owns a run of 64-key blocks and there is no LDS and no barrier:
//   lane = (ks = lane >> 4, dc = lane & 15): in step i of a block the lane holds the dc-th 16-byte chunk of key 4 i + ks, so
//   every request covers four whole 256-byte rows; all 32 requests of a block (16 K, 16 V: 32 KiB per wave) go out together.
//   score: 8-element partial dot with the lane's q chunk (v_dot2), summed over the 16 lanes of the key; softmax statistics are
//   per block (online across blocks, log2 domain, fp32); P.V: each lane accumulates its 8 output dims over its keys, the four
//   key groups meet at the end.  The split's unnormalised (O, m, l) goes to the same partial buffers as before.
typedef __bf16 attn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float attn_dot8(const uint4& a, const uint4& b) {
    float acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(attn_bf16x2, a.x), __builtin_bit_cast(attn_bf16x2, b.x), 0.f, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(attn_bf16x2, a.y), __builtin_bit_cast(attn_bf16x2, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(attn_bf16x2, a.z), __builtin_bit_cast(attn_bf16x2, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(attn_bf16x2, a.w), __builtin_bit_cast(attn_bf16x2, b.w), acc, false);
    return acc;
}

__global__ __launch_bounds__(256) void attn_decode_stream_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int head = blockIdx.y, bat = blockIdx.z;
    const int split = blockIdx.x * 4 + wave;                 // one split of the key range per wave
    int64_t n_keys = a.Tk;
    if (a.dyn_pos) n_keys = a.dyn_pos[bat] + 1;              // the query at position p sees keys [0, p]
    const int nblk = (int)((n_keys + 63) >> 6);              // split s takes blocks s, s + n_splits, ...: the waves resident at one
                                                             // time read neighbouring blocks (contiguous runs per split put them
                                                             // megabytes apart at equal offsets: 591 us instead of 429 at 131 k keys)
    const int ks = lane >> 4, dc = lane & 15;
    const uint4 qv = ((const uint4*)(a.q + bat * a.q_sb + head * a.q_sh))[dc];
    // wave-uniform base (SGPR pair) + 32-bit lane offset: one VGPR per address instead of two (the host checks Tk * stride < 4 GiB)
    typedef const __attribute__((address_space(1))) unsigned char* gptr_t;
    const gptr_t kp = (gptr_t)(uint64_t)(a.k + bat * a.k_sb + head * a.k_sh);
    const gptr_t vp = (gptr_t)(uint64_t)(a.v + bat * a.v_sb + head * a.v_sh);
    const uint32_t kst = (uint32_t)(a.k_st * 2), vst = (uint32_t)(a.v_st * 2), dco = (uint32_t)dc * 16;
    typedef unsigned int attn_u32x4 __attribute__((ext_vector_type(4)));
    auto ld16 = [](gptr_t base, uint32_t off) {
        const attn_u32x4 t = *(const __attribute__((address_space(1))) attn_u32x4*)(base + off);
        return make_uint4(t[0], t[1], t[2], t[3]);
    };
    float m_run = -INFINITY, l_run = 0.f;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#ifndef ATTN_DECODE_PIPE
#define ATTN_DECODE_PIPE 1                   // 1 (round 6): 32-key half blocks, the next half's 16 requests in flight while this one is reduced; 0: whole 64-key blocks, load then compute
#endif
#if ATTN_DECODE_PIPE
    // A wave's blocks in HALVES of 32 keys (8 steps), double-buffered: the requests of half j + 1 go out before half j is reduced, so a
    // wave keeps 16-32 KiB in flight all the time.  (Whole blocks, loaded then reduced: at 8 k keys and 64 splits a wave has two blocks,
    // i.e. two exposed round trips with the CU's request queue draining while all its waves reduce -- 41 us for 134 MB, 3.3 TB/s.)
    const int nhalf = (int)((n_keys + 31) >> 5);
    const int nb_my = split < nblk ? (nblk - split + a.n_splits - 1) / a.n_splits : 0;
    auto half_of = [&](int jh) { return 2 * (split + (jh >> 1) * a.n_splits) + (jh & 1); };
    int n_my = 2 * nb_my;
    if (n_my > 0 && half_of(n_my - 1) >= nhalf) --n_my;     // the last block's second half may lie beyond the keys
    auto load_half = [&](int jh, uint4 (&kr)[8], uint4 (&vr)[8]) {
        const int64_t k0 = (int64_t)half_of(jh) * 32 + ks;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int64_t key = k0 + 4 * i;
            key = key < n_keys ? key : n_keys - 1;           // a ragged last half re-reads the last key (masked below)
            kr[i] = ld16(kp, (uint32_t)key * kst + dco);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int64_t key = k0 + 4 * i;
            key = key < n_keys ? key : n_keys - 1;
            vr[i] = ld16(vp, (uint32_t)key * vst + dco);
        }
    };
    auto reduce_half = [&](int jh, const uint4 (&kr)[8], const uint4 (&vr)[8]) {
        const int64_t k0 = (int64_t)half_of(jh) * 32 + ks;
        float sc[8];
        float mb = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float d = attn_dot8(kr[i], qv);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            d += __shfl_xor(d, 8, 64);
            sc[i] = k0 + 4 * i < n_keys ? d * a.scale_log2 : -INFINITY;
            mb = fmaxf(mb, sc[i]);
        }
        mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float m_new = fmaxf(m_run, mb);                // finite: the half holds at least one key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float pw = __builtin_amdgcn_exp2f(sc[i] - m_new);
            l_run += pw;
            o[0] = fmaf(pw, bf_lo(vr[i].x), o[0]); o[1] = fmaf(pw, bf_hi(vr[i].x), o[1]);
            o[2] = fmaf(pw, bf_lo(vr[i].y), o[2]); o[3] = fmaf(pw, bf_hi(vr[i].y), o[3]);
            o[4] = fmaf(pw, bf_lo(vr[i].z), o[4]); o[5] = fmaf(pw, bf_hi(vr[i].z), o[5]);
            o[6] = fmaf(pw, bf_lo(vr[i].w), o[6]); o[7] = fmaf(pw, bf_hi(vr[i].w), o[7]);
        }
        m_run = m_new;
    };
    {
        uint4 kA[8], vA[8], kB[8], vB[8];
        if (n_my > 0) load_half(0, kA, vA);
        for (int jh = 0; jh < n_my; jh += 2) {               // two halves per trip: the buffers alternate without register copies
            if (jh + 1 < n_my) load_half(jh + 1, kB, vB);
            reduce_half(jh, kA, vA);
            if (jh + 1 < n_my) {
                if (jh + 2 < n_my) load_half(jh + 2, kA, vA);
                reduce_half(jh + 1, kB, vB);
            }
        }
    }
#else
    for (int blk = split; blk < nblk; blk += a.n_splits) {
        const int64_t k0 = (int64_t)blk * 64 + ks;
        uint4 kr[16], vr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int64_t key = k0 + 4 * i;
            key = key < n_keys ? key : n_keys - 1;           // a ragged last block re-reads the last key (masked below)
            kr[i] = ld16(kp, (uint32_t)key * kst + dco);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int64_t key = k0 + 4 * i;
            key = key < n_keys ? key : n_keys - 1;
            vr[i] = ld16(vp, (uint32_t)key * vst + dco);
        }
        float sc[16];
        float mb = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float d = attn_dot8(kr[i], qv);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            d += __shfl_xor(d, 8, 64);
            sc[i] = k0 + 4 * i < n_keys ? d * a.scale_log2 : -INFINITY;
            mb = fmaxf(mb, sc[i]);
        }
        mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float m_new = fmaxf(m_run, mb);                // finite: the block holds at least one key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pw = __builtin_amdgcn_exp2f(sc[i] - m_new);
            l_run += pw;
            o[0] = fmaf(pw, bf_lo(vr[i].x), o[0]); o[1] = fmaf(pw, bf_hi(vr[i].x), o[1]);
            o[2] = fmaf(pw, bf_lo(vr[i].y), o[2]); o[3] = fmaf(pw, bf_hi(vr[i].y), o[3]);
            o[4] = fmaf(pw, bf_lo(vr[i].z), o[4]); o[5] = fmaf(pw, bf_hi(vr[i].z), o[5]);
            o[6] = fmaf(pw, bf_lo(vr[i].w), o[6]); o[7] = fmaf(pw, bf_hi(vr[i].w), o[7]);
        }
        m_run = m_new;
    }
#endif
    // the four key groups of the wave meet (every lane of a group carries the same l)
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] += __shfl_xor(o[e], 16, 64);
        o[e] += __shfl_xor(o[e], 32, 64);
    }
    if (split < a.n_splits) {
        const int64_t slot = ((int64_t)bat * a.H + head) * a.n_splits + split;
        if (ks == 0) {
            float4* po = (float4*)(a.part_o + slot * DH + dc * 8);
            po[0] = make_float4(o[0], o[1], o[2], o[3]);
            po[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (lane == 0) { a.part_ml[slot * 2] = m_run; a.part_ml[slot * 2 + 1] = l_run; }
    }
}

// merge the splits of one (batch, head): out[d] = sum_s O_s[d] 2^(m_s - M) / sum_s l_s 2^(m_s - M).
// 512 threads = 4 groups x 128 output dims; the statistics are read once into LDS (coalesced) and turned into weights, then
// group g sums splits g, g + 4, ... with eight independent row loads in flight (the first version walked the splits one
// dependent L2 round trip at a time: 7 us at 16 splits, ~30 us at the 64 the streaming kernel uses).
#define ATTN_MAX_SPLITS 1024
__global__ __launch_bounds__(512) void attn_decode_combine_kernel(const float* __restrict__ part_o,
                                                                  const float* __restrict__ part_ml,
                                                                  uint16_t* __restrict__ o, int H, int n_splits) {
    __shared__ float w_s[ATTN_MAX_SPLITS];
    __shared__ float red[16];
    __shared__ float acc_s[4][DH];
    const int tid = threadIdx.x, d = tid & (DH - 1), g = tid >> 7, head = blockIdx.x, bat = blockIdx.y;
    const int64_t base = ((int64_t)bat * H + head) * n_splits;
    // pass 1: maxima
    float M = -INFINITY;
    for (int s = tid; s < n_splits; s += 512) M = fmaxf(M, part_ml[(base + s) * 2]);
    M = wave_max(M);
    if ((tid & 63) == 0) red[tid >> 6] = M;
    __syncthreads();
    M = red[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) M = fmaxf(M, red[i]);
    // pass 2: weights and denominator
    float L = 0.f;
    for (int s = tid; s < n_splits; s += 512) {
        const float m = part_ml[(base + s) * 2];
        const float w = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - M);
        w_s[s] = w;
        L = fmaf(part_ml[(base + s) * 2 + 1], w, L);
    }
    L = wave_sum(L);
    if ((tid & 63) == 0) red[8 + (tid >> 6)] = L;
    __syncthreads();
    L = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) L += red[8 + i];
    // pass 3: weighted rows
    float acc = 0.f;
    int s = g;
    for (; s + 28 < n_splits; s += 32) {
        float r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = part_o[(base + s + 4 * u) * DH + d];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(r[u], w_s[s + 4 * u], acc);
    }
    for (; s < n_splits; s += 4) acc = fmaf(part_o[(base + s) * DH + d], w_s[s], acc);
    acc_s[g][d] = acc;
    __syncthreads();
    if (g == 0) {
        acc = acc_s[0][d] + acc_s[1][d] + acc_s[2][d] + acc_s[3][d];
        o[((int64_t)bat * H + head) * DH + d] = f_to_bf(L > 0.f ? acc / L : 0.f);
    }
}

static int attn_check_strides(int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                              int64_t v_sb, int64_t v_st, int64_t v_sh) {
    return ((q_st % 8) || (k_st % 8) || (v_st % 8) || (q_sh % 8) || (k_sh % 8) || (v_sh % 8) || (q_sb % 8) ||
            (k_sb % 8) || (v_sb % 8)) ? -1 : 0;      // 16-byte row accesses
}

extern "C" int evo_attn_fwd_causal_bf16(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H,
                                        int64_t Tq, int64_t Tk, int64_t q_pos0, int64_t q_sb, int64_t q_st,
                                        int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh, int64_t v_sb,
                                        int64_t v_st, int64_t v_sh, float softmax_scale, void* vt_ws, void* stream) {
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || q_pos0 < 0) return -1;
    if (attn_check_strides(q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh)) return -1;
    if (H > 65535 || B > 65535) return -1;
    AttnArgs a;
    a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.o = (uint16_t*)o;
    a.Tq = Tq; a.Tk = Tk; a.q_pos0 = q_pos0;
    a.q_sb = q_sb; a.q_st = q_st; a.q_sh = q_sh; a.k_sb = k_sb; a.k_st = k_st; a.k_sh = k_sh;
    a.v_sb = v_sb; a.v_st = v_st; a.v_sh = v_sh;
    a.H = (int)H;
    // softmax_scale <= 0 (ABI 10): the queries are PRE-SCALED by softmax_scale * log2(e) (evo_rope_qk_bf16's q_scale): scores are exponents
    a.prescaled = softmax_scale <= 0.f ? 1 : 0;
    a.scale_log2 = a.prescaled ? 1.0f : softmax_scale * 1.4426950408889634f;
    a.dyn_pos = nullptr; a.part_o = nullptr; a.part_ml = nullptr; a.n_splits = 1;
    // query ranges longer than one 128-row block take the 64-rows-per-wave kernel of csrc/attn_w64.hip; EVO_AMD_ATTN_FORM = 1 keeps
    // them on the 8-wave pipelined kernel of rounds 2-4 (A/B measurements), 0 on the 128-row kernel
    static const int form = [] { const char* e = getenv("EVO_AMD_ATTN_FORM"); return e ? atoi(e) : 2; }();
    a.nbh = (int)(B * H);
    a.q_pad = 0; a.vt = nullptr; a.vt_row = 0;
    if (form >= 2 && Tq > QB && vt_ws) return evo_attn_w64_launch(a, B, vt_ws, stream);
    const int use_pipe = form >= 1 && Tq > QB;
    const int qblock = use_pipe ? PQB : QB;
    a.n_qblocks = (int)((Tq + qblock - 1) / qblock);
    const int64_t n_wg = (int64_t)a.n_qblocks * a.nbh;
    if (n_wg > 0x7fffffff) return -1;
    if (use_pipe)
        hipLaunchKernelGGL(attn_fwd_pipe_kernel, dim3((unsigned)n_wg), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3((unsigned)n_wg), dim3(256), 0, (hipStream_t)stream, a);
    return evo_launch_status();
}

extern "C" int evo_attn_decode_bf16(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H,
                                    int64_t Tk, int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_st,
                                    int64_t k_sh, int64_t v_sb, int64_t v_st, int64_t v_sh, const int64_t* dyn_pos,
                                    float* part_o, float* part_ml, int64_t n_splits, float softmax_scale,
                                    void* stream) {
    if (B <= 0 || H <= 0 || Tk <= 0 || n_splits <= 0 || n_splits > 1024 || !part_o || !part_ml) return -1;
    if (attn_check_strides(q_sb, 8, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh)) return -1;
    if (H > 65535 || B > 65535) return -1;
    AttnArgs a;
    a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.o = (uint16_t*)o;
    a.Tq = 1; a.Tk = Tk; a.q_pos0 = Tk - 1;
    a.q_sb = q_sb; a.q_st = 0; a.q_sh = q_sh; a.k_sb = k_sb; a.k_st = k_st; a.k_sh = k_sh;
    a.v_sb = v_sb; a.v_st = v_st; a.v_sh = v_sh;
    a.H = (int)H;
    a.prescaled = softmax_scale <= 0.f ? 1 : 0;
    a.scale_log2 = a.prescaled ? 1.0f : softmax_scale * 1.4426950408889634f;
    a.n_qblocks = 1; a.q_pad = 0; a.vt = nullptr; a.vt_row = 0;
    a.dyn_pos = dyn_pos; a.part_o = part_o; a.part_ml = part_ml; a.n_splits = (int)n_splits; a.nbh = (int)(B * H);
    hipStream_t s = (hipStream_t)stream;
    // EVO_ATTN_DECODE_FORM=0 keeps the MFMA split kernel (measurement builds); default: the streaming kernel, one split per wave
    static const int form = [] { const char* e = getenv("EVO_ATTN_DECODE_FORM"); return e ? atoi(e) : 1; }();
    if (form != 0 && Tk * k_st * 2 < 0xffffffffll && Tk * v_st * 2 < 0xffffffffll)
        hipLaunchKernelGGL(attn_decode_stream_kernel, dim3((unsigned)((n_splits + 3) / 4), (unsigned)H, (unsigned)B), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3((unsigned)n_splits, (unsigned)H, (unsigned)B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_decode_combine_kernel, dim3((unsigned)H, (unsigned)B), dim3(512), 0, s, part_o, part_ml,
                       (uint16_t*)o, (int)H, (int)n_splits);
    return evo_launch_status();
}
