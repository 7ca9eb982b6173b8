// Causal attention forward for gfx950, head dim 128 -- the prefill kernel of round 5 (query ranges longer than one 128-row block).
// Same arithmetic contract as csrc/attn.hip (bf16 in, fp32 scores / softmax / accumulate, P rounded to bf16 for P.V, bf16 out:
// FlashAttention-2's numerics, which is what the reference runs [REF README.md:47-50; evo-1-131k-base_inference.yml:30]) and the
// same "swapped" MFMA forms (S^T = K Q^T, O^T = V^T P^T on v_mfma_f32_32x32x16_bf16: a lane owns one query column, so the softmax
// statistics are lane-local).  What changes is the shape of the work a wave does:
//
//   * a workgroup is 4 waves = ONE wave per SIMD, each with the whole 512-register file, and a wave owns 64 query rows (two 32-row
//     MFMA column blocks).  Every K fragment (ds_read_b128) and every V^T fragment pair (ds_read_b64_tr_b16) read from LDS feeds
//     TWO MFMAs -- half the LDS fragment traffic per flop of the 8 x 32-row kernel (profiles/r03_attn_notes.txt: 290 KB per 64-key
//     tile and CU there), and no second wave competing for the SIMD's matrix pipe and VALU issue slots;
//   * a 64-key tile costs a wave 64 MFMAs (2 x 32) and ~350 other instructions (64 scores per lane: scale / subtract, exp2, row sum,
//     bf16 pack, row max; 48 LDS fragment reads; 8-9 LDS-DMA pieces).  A 32x32x16 MFMA occupies the pipe for 32 cycles, which hides
//     ~5 single-issue instructions, so the softmax is cut into two streams that ride under BOTH MFMA phases of a trip:
//        phase 1:  QK^T(t+1)  ||  exp / sum / pack of the last 3/4 of tile t's scores, K fragment reads (second half of K(t+1))
//        phase 2:  P.V(t)     ||  row max of S(t+1), the running-max bookkeeping, exp / sum / pack of the FIRST quarter of S(t+1),
//                                 V^T fragment reads, the first-half K fragments of tile t+2, the DMA pieces of K(t+4) / V(t+2)
//     Program order IS the schedule (sched_barrier after every MFMA's chunk).  Nothing sits between the phases except the (rare)
//     masking of a diagonal / ragged tile and the (rare) rescale of O;
//   * K tiles live in a 4-deep LDS ring, V tiles in a 3-deep one, both filled by LDS-DMA (buffer_load ... lds, 1 KiB per
//     wave-instruction).  K runs one tile further ahead than the 8-wave kernel's so that the K fragments a trip starts with are read
//     BEFORE the trip's barrier, under the previous trip's P.V MFMAs: with one wave per SIMD nobody else covers an LDS round trip
//     at the head of a trip.  ONE barrier per tile;
//   * the running max is DEFERRED per row: a row's reference point m only moves when the tile's scaled maximum exceeds it by more
//     than W_THR (log2 units), otherwise P = 2^(s c - m) <= 2^W_THR is used as it is -- P, l and O are floating point with the
//     fp32 / bf16 exponent range (2^127), so a common factor of up to 2^W_THR costs no precision (every rounding is relative), only
//     W_THR bits of the 127 of overflow headroom: with W_THR = 32, l <= 2^32 T and |O| <= 2^32 sum |v| stay 70 binary orders below
//     the fp32 limit at T = 131,073.  The O rescale is what the threshold buys off: 136 accumulator registers per lane live in
//     AGPRs, a rescale is read - multiply - write of each (~400 instructions, about the cost of a whole trip).  The threshold was
//     8 at first (the value the FlashAttention-3/4 papers use for fp16-range P): fine on N(0, 1) scores, but the MODEL's scores at
//     block 8 have a standard deviation of 9.5 log2 units and reach +55 (tools/attn_instep_ab.py --model): a row's running maximum
//     climbs ~19 units between its first tile and its last, 64 rows share a wave, and a rescale ran on every fifth trip at
//     8 x 8,193 (5.74 ms against 4.78 ms on N(0, 1) inputs of the same shape; profiles/r05_attn_w64_model_scores.txt).  At 32 a
//     row sets its reference point on its first tile and keeps it.  The decision is PER ROW (rows that
//     do not move get alpha = 1 exactly), so a row's arithmetic never depends on which other rows share its wave: outputs stay
//     bit-identical across query offsets / launch geometries (tests/test_gpu_fullsize.py).  W_THR = 0 is the textbook update;
//   * query blocks are aligned to the END of the query range (block 0 is the short one).  T = 2^k + 1 (a BOS token in front of 2^k
//     nucleotides) then costs one extra 1-tile block instead of a whole 256-row block walking all 2^k / 64 + 1 tiles for ONE row
//     (+6 % tiles at 8 x 8,193);
//   * O^T leaves through v_permlane32_swap pairs as 16-byte stores (8 per 32-row block and lane instead of 16 x 8 bytes).
//
//   * V arrives TRANSPOSED.  One wave per SIMD gets a fifth of the LDS rate on 8-byte reads: the 32 ds_read_b64_tr_b16 of a trip cost
//     16 cycles each (16.6 of 117 ms at 1 x 131,073: profiles/r05_attn_w64_ablation_v2_lds.txt), the 16 ds_read_b128 of the K side 6.
//     A pre-pass (attn_vt_kernel: 2 bytes moved per byte of V, ~0.4 % of the attention time at 131 k) writes V^T [head][d][key], the
//     tile's DMA lands it as [128 d][64 keys] and a V^T fragment is ONE ds_read_b128 (16 per trip).  For the 8 keys of a lane's
//     fragment to be contiguous, the K fragments are read with bits 2 and 3 of the key index swapped (a different per-lane constant,
//     nothing else): accumulator register r of S^T then holds key 16 (r >> 3) + 8 (lane >> 5) + (r & 7) of its 32-key half.
//
// LDS images: K [64 keys][272 B] (padded rows: b128 fragment reads conflict-free, per-lane base + immediates); V^T [128 d][128 B], the
// 16-byte chunks of a row XOR-swizzled by ((d >> 1) & 7) on the DMA's source side (b128 fragment reads conflict-free, four per-lane
// bases + immediates).
#include <stdlib.h>
#include <utility>
#include "attn_common.h"
#include "../../include/evo_mi355x.h"

#define W_QB 256
#define W_KROW 272
#define W_KSTAGE (KB * W_KROW)              // 17,408 B = 17 DMA pieces
#define W_VSTAGE (KB * 256)                 // 16,384 B = 16 DMA pieces
#define W_NK 4
#define W_NV 3
#define W_VBASE (W_NK * W_KSTAGE)           // 69,632
#define W_LDS (W_VBASE + W_NV * W_VSTAGE)   // 118,784 B
#ifndef W_THR
#define W_THR 32.0f                         // deferred-max threshold, log2 units (0 = move the reference on every new maximum)
#endif
#ifndef W_THRP
#define W_THRP 64.0f                        // PRE (pre-scaled queries): a row's reference point leaves 0 only beyond +-W_THRP log2 units
#endif
#ifndef W_EARLY
#define W_EARLY 0                           // measurement knob (PRE only): 1 = the exp stream of a tile's FIRST 32 keys (80 of its 160 instructions) runs under the
                                            // previous trip's P.V MFMAs, speculatively.  MEASURED SLOWER (profiles/r06_attn_notes.txt: 120.1 vs 116.9 ms at 1 x 131,073): see side_early
#endif
#ifndef W_PROFILE
#define W_PROFILE 0                         // 1 (timing build, tools/attn_phase_profile.py): every wave accumulates shader clocks per trip segment and
                                            // writes them over its first output row's bytes: [resc, phase 1, between, phase 2, waits, barrier, trips]
#endif
#if W_PROFILE
#define W_STAMP(K) { const uint64_t n_ = __builtin_readcyclecounter(); tp_[K] += n_ - tl_; tl_ = n_; }
#else
#define W_STAMP(K)
#endif
#ifndef W_VD
#define W_VD 4                              // V^T fragments read ahead of their MFMAs (5+: the register file spills into AGPR copies)
#endif
// measurement builds (tools/attn_ablate.sh): W_ABL_NOEXP / NOSIDE / NOLDS / NODMA / NOBAR drop one ingredient of a trip (wrong results)
#ifndef W_LSUM_MFMA
#define W_LSUM_MFMA 0                       // 1: softmax denominators on the matrix pipe (8 more MFMAs per trip instead of 64 v_add_f32: measured 4 % SLOWER -- an MFMA costs its 32 pipe cycles, an add ~2.7)
#endif
#define W_NG_EARLY 4                        // exp groups (2 x 2 scores x ... = 14 instructions each) done under the previous trip's P.V

typedef int w_srd_t __attribute__((ext_vector_type(4)));
typedef unsigned int w_u32x4 __attribute__((ext_vector_type(4)));     // asm operands must be native vectors (a HIP uint4 is a struct)
typedef unsigned int w_u32x2 __attribute__((ext_vector_type(2)));

// value of lane ^ 32, combined with max / add (v_permlane32_swap: one VALU instruction, no LDS round trip)
__device__ __forceinline__ float w_xor32_max(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float w_xor32_add(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

#define W_WAIT(K)                                                                                    \
    do {                                                                                             \
        if (np == 9) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9 * (K)) : "memory");                  \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (K)) : "memory");                          \
    } while (0)
#define W_BARRIER() asm volatile("s_barrier" ::: "memory")

// MFMAs and LDS fragment reads are inline asm with the register FILE of every operand spelled out: O^T accumulators (128), Q fragments
// (64) and K fragments (32) live in the AGPR half of the wave's 512 registers, scores / P / V^T fragments and the softmax arithmetic
// in the VGPR half (left to itself hipcc splits the accumulators between the files and moves ~1,200 v_accvgpr_read / write per trip,
// spilling 83 registers to scratch).  The compiler neither inserts waits for asm LDS reads nor hazard padding behind asm MFMAs:
//   * LDS reads return in order; every consumer MFMA is preceded by a COUNTED s_waitcnt lgkmcnt(N), N = the LDS instructions issued
//     after its fragment's read (w_wait_* below replay the program order of a trip); a trip ends with lgkmcnt(0), 9 MFMAs behind its
//     last read;
//   * an accumulator is read by other instructions only >= 4 MFMAs (128 cycles) after the last MFMA that wrote it, or behind 24
//     explicit wait states (masking / rescale / epilogue paths).
#define W_MFMA_S0(S, KF, QF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(S) : "a"(KF), "a"(QF))
#define W_MFMA_S(S, KF, QF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S) : "a"(KF), "a"(QF))
// the LAST k-step of a score tuple: D != C -- the sum accumulated in the work tuple WK lands in the exponent tuple E (a free copy)
#define W_MFMA_SD(E, WK, KF, QF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %1" : "=v"(E) : "v"(WK), "a"(KF), "a"(QF))
#define W_MFMA_L(L, ONES, PF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(L) : "v"(ONES), "v"(PF))
#define W_DSR_K(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(DST) : "v"(ADDR), "n"(OFF))
#ifdef W_V_AGPR             /* experiment: V^T fragments in AGPRs too */
#define W_DSR_V(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(DST) : "v"(ADDR), "n"(OFF))
#define W_MFMA_O(O, VF, PF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(O) : "a"(VF), "v"(PF))
#else
#define W_DSR_V(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define W_MFMA_O(O, VF, PF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(O) : "v"(VF), "v"(PF))
#endif
// (hipcc does not capture a local that a generic lambda names ONLY in asm operands: name it once outside of them)
#define W_USE2(A, B) (void)(A), (void)(B)
#define W_USE3(A, B, C) (void)(A), (void)(B), (void)(C)
// An instruction the compiler emits for the softmax streams has no tie to the volatile asm sequence: instruction selection would sink
// it to its first real use (all 64 exp2 behind the last QK^T MFMA).  W_PIN names its result as an INPUT of an empty volatile asm right
// behind it, which puts it in its gap (an input does not make the recognizer pad anything; an asm OUTPUT read by the next instruction does).
#define W_PIN(X) asm volatile("" ::"v"(X))
#define W_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#if defined(W_ABL_NOLDS) || defined(W_ABL_NOLDSK)          /* in-trip fragment reads and their waits off (the prologue's stay) */
#define W_T_DSR_K(DST, ADDR, OFF) (void)0
#else
#define W_T_DSR_K W_DSR_K
#endif
#if defined(W_ABL_NOLDS) || defined(W_ABL_NOLDSV)
#define W_T_DSR_V(DST, ADDR, OFF) (void)0
#else
#define W_T_DSR_V W_DSR_V
#endif
#if defined(W_ABL_NOLDS) || defined(W_ABL_NOLGKM)
#define W_T_LGKM(N) (void)0
#else
#define W_T_LGKM W_LGKM
#endif
#define W_NOP24()                                                        \
    do {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                               \
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)

// LDS instructions of a trip in program order (inside a gap: [wait] MFMA, V^T fragment read, K fragment read).  The 16 K fragments of a
// tile (f = 8 kt + ks, consumed by the QK^T MFMAs of phase-1 gaps 2 f, 2 f + 1) go through a ring of FOUR registers quads: fragment f >= 4
// is read right behind the second MFMA of fragment f - 4 (gap 2 f - 7), fragments 0..3 of the NEXT tile under the P.V MFMAs.
//   phase 1, gap i:  odd i <= 23: K fragment (i + 7) / 2;                      even i >= 32 - 2 W_VD: V^T fragment (i - (32 - 2 W_VD)) / 2
//   phase 2, gap j:  even j: V^T fragment j / 2 + W_VD (while < 16);           j = 10, 14, 18, 22: K fragment (j - 10) / 4 of the next tile
constexpr int w_p1_ops(int i) { return (((i & 1) && i <= 23) ? 1 : 0) + ((i >= 32 - 2 * W_VD && !(i & 1)) ? 1 : 0); }
constexpr int w_p2_ops(int j) { return ((!(j & 1) && (j / 2 + W_VD) < 16) ? 1 : 0) + ((j >= 10 && j <= 22 && ((j - 10) & 3) == 0) ? 1 : 0); }
constexpr int w_wait_k1(int f) {            // before the MFMA of phase-1 gap 2 f (f >= 4): LDS instructions behind the read of gap 2 f - 7
    int n = 0;
    for (int i = 2 * f - 6; i < 2 * f; ++i) n += w_p1_ops(i);
    return n > 15 ? 15 : n;                 // (the counter field has 4 bits: a smaller count only waits for more)
}
constexpr int w_wait_v(int p) {             // before the MFMA of phase-2 gap 2 p: LDS instructions behind the read of V^T fragment p
    int n = 0;
    if (p < W_VD) {
        const int i0 = 32 - 2 * W_VD + 2 * p;
        n += ((i0 & 1) && i0 <= 23) ? 1 : 0;                                   // (a K read in the same gap comes after the V^T read)
        for (int i = i0 + 1; i < 32; ++i) n += w_p1_ops(i);
        for (int j = 0; j < 2 * p; ++j) n += w_p2_ops(j);
    } else {
        const int j0 = 2 * (p - W_VD);
        n += (j0 >= 10 && j0 <= 22 && ((j0 - 10) & 3) == 0) ? 1 : 0;
        for (int j = j0 + 1; j < 2 * p; ++j) n += w_p2_ops(j);
    }
    return n > 15 ? 15 : n;
}
static_assert(W_VD >= 1 && W_VD <= 8, "V^T read-ahead distance");

// compile-time loops: the asm operands that are immediates (LDS offsets, lgkmcnt counts) need integer constant expressions
template <int I> struct w_ic { static constexpr int v = I; };
template <class F, int... Is>
__device__ __forceinline__ void w_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(w_ic<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void w_static_for(F&& f) { w_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Round 6.  (1) The score tile of the next tile is accumulated in a WORK tuple per query block (reused by both 32-key halves) and the LAST
// QK^T MFMA of a tuple writes its result into the exponent tuple S[x][kt] (v_mfma with D != C): by then the exp stream has consumed that
// tuple's previous contents (tile t's exponents of half kt are read in gaps 16 kt .. 16 kt + 14, the deposits sit in gaps 16 kt + 14 / 15), so
// ONE exponent set serves both tiles -- 96 registers where S + ev took 128, and the exponents are formed IN PLACE.
// (2) PRE: the queries arrive PRE-SCALED by softmax_scale * log2(e) (the rotary kernel folds the factor into its one rounding of q), so a
// score IS an exponent: the 64 `fma(S, c, -m)` of a trip disappear (VALU per MFMA 5.7 -> 4.8).  The exponents are taken relative to a
// per-row reference point that STARTS AT 0 and only moves when a tile's largest exponent leaves +-W_THRP log2 units (a row's first visible
// tile may also move it down): P = 2^s up to 2^64, l <= 2^81, |O| <= 2^91 at T = 131,073 -- fp32 / bf16 carry the exponent, every rounding is
// relative.  While no row of the wave has moved nothing is subtracted at all; once one has (wave-uniform `any_nm`) the tile's scores get
// their row's offset in a burst between the two phases, like the masking of a diagonal tile.
template <bool PRE>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[W_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..3
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int np = wave == 0 ? 9 : 8;                                   // DMA pieces of this wave per trip (17 K + 16 V over 4 waves)

    const int64_t Tk_ = a.Tk, q_pos0_ = a.q_pos0;
    int qblk, head, bat;
    attn_block_map(a, qblk, head, bat);
    const int64_t q0 = (int64_t)qblk * W_QB - a.q_pad;                  // block 0 starts q_pad rows before the range (those rows: clamped, not stored)
    const uint16_t* qp = a.q + bat * a.q_sb + head * a.q_sh;
    const unsigned char* kp = (const unsigned char*)(a.k + bat * a.k_sb + head * a.k_sh);
    const int64_t kst_b = a.k_st * 2;

    // ---- this lane's two query rows (column l31 of the wave's two 32-row blocks) ---------------------------------------------------
    const int64_t wrow0 = q0 + wave * 64;                               // wave-uniform
    int64_t qrow[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) qrow[x] = wrow0 + 32 * x + l31;
    // key limit of a lane's row relative to the wave's first (clamped) row position; the mask path adds the wave-uniform part
    const int64_t wpos0 = (wrow0 < 0 ? 0 : wrow0) + q_pos0_;           // position of the wave's first valid row: min over the wave
    int lim_rel[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) lim_rel[x] = (int)((qrow[x] < 0 ? 0 : qrow[x]) + q_pos0_ - wpos0);      // 0..63
    const bool row_ok0 = qrow[0] >= 0, row_ok1 = qrow[1] >= 0;
    const int64_t orow_first = (qrow[0] < 0 ? 0 : qrow[0]);            // (epilogue: output rows of the lane)
    const int64_t orow_second = (qrow[1] < 0 ? 0 : qrow[1]);

    int64_t max_key = q0 + W_QB - 1 + q_pos0_;                          // (q0 + 255 <= Tq - 1 by construction)
    if (max_key > Tk_ - 1) max_key = Tk_ - 1;
    const int n_tiles = (int)(max_key / KB) + 1;

    // ---- DMA plan ---------------------------------------------------------------------------------------------------------------
    // K piece j (17 of 1 KiB: padded rows) -> waves j % 4, five per-lane offsets; V piece j (16: 4 rows each) -> one per-lane offset
    // (row (lane >> 4) of the piece, 64-byte block XOR (row & 3)) + a wave-uniform row offset in the instruction's SGPR offset.
    uint32_t dk_off[5];
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
        const int j = wave + 4 * jj;
        const int pos = j * 1024 + 16 * lane;
        const int r = pos / W_KROW;
        int c = pos - r * W_KROW;
        c = c < 256 ? c : 0;                                            // pad lanes re-fetch the row's first granule (never read back)
        dk_off[jj] = (uint32_t)r * (uint32_t)kst_b + (uint32_t)c;
    }
    // V^T piece j (16 of 1 KiB) = d rows 8 j .. 8 j + 7 of the tile, 128 B (64 keys) each: lane -> row (lane >> 3), LDS chunk lane & 7, which
    // holds the row's chunk (lane & 7) ^ ((d >> 1) & 7); d >> 1 = 4 j + (lane >> 4) and j = wave + 4 jj, so the swizzle is one per-lane
    // constant of the wave.  The piece's row offset goes into the instruction's SGPR offset.
    const int64_t vt_row_b = a.vt_row * 2;
    uint32_t dv_off;
    {
        const int r = lane >> 3;
        const int f = (4 * (wave & 1) + (lane >> 4)) & 7;
        dv_off = (uint32_t)r * (uint32_t)vt_row_b + (uint32_t)(((lane & 7) ^ f) << 4);
    }
    const uint32_t dv_soff = (uint32_t)(8 * vt_row_b);                 // per V^T piece: 8 rows
    const unsigned char* vtp = (const unsigned char*)(a.vt + ((int64_t)bat * a.H + head) * DH * a.vt_row);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // Buffer descriptors of one tile: base = its first row, num_records = bytes up to the end of the last VALID key (0 past the end):
    // the hardware bounds check returns zeros beyond -- no clamping, every trip issues the same instructions.  Branch-free scalar code:
    // tiles below n_full are whole, tile n_full holds the Tk % 64 last keys, later ones nothing.
    const int n_full = (int)(Tk_ / KB), k_rem = (int)(Tk_ % KB);
    auto tile_srd = [&](const unsigned char* base, int64_t st_b, int tile) __attribute__((always_inline)) {
        const uint32_t rec_full = (uint32_t)((KB - 1) * st_b + 256);
        const uint32_t rec_rem = k_rem ? (uint32_t)((k_rem - 1) * st_b + 256) : 0u;
        const uint64_t a64 = (uint64_t)base + (uint64_t)(uint32_t)tile * (uint64_t)(KB * st_b);
        w_srd_t d;
        d[0] = (int)(uint32_t)a64;
        d[1] = (int)(uint32_t)(a64 >> 32);
        d[2] = (int)(tile < n_full ? rec_full : (tile == n_full ? rec_rem : 0u));
        d[3] = 0x00020000;
        return d;
    };
    // V^T tile: base = column 64 tile of the head's plane, rows vt_row_b apart; the plane is padded to whole tiles
    const int n_vt = (int)(a.vt_row / KB);
    auto vt_srd = [&](int tile) __attribute__((always_inline)) {
        const uint64_t a64 = (uint64_t)vtp + (uint64_t)(uint32_t)tile * (uint64_t)(KB * 2);
        w_srd_t d;
        d[0] = (int)(uint32_t)a64;
        d[1] = (int)(uint32_t)(a64 >> 32);
        d[2] = (int)(tile < n_vt ? (uint32_t)((DH - 1) * vt_row_b + KB * 2) : 0u);
        d[3] = 0x00020000;
        return d;
    };
// One DMA piece = "m0 <- LDS address" + the load.  In the prologue both in one asm; inside a trip the m0 write heads the gap and the load
// ends it (W_M0_* / W_LD_*): the MFMA between them is the wait state the pair needs, and the piece costs two issue slots instead of four.
#define W_DMA_K(SRD, SLOT_LDS, JJ)                                                                              \
    {                                                                                                           \
        const int j_ = wave + 4 * (JJ);                                                                         \
        if ((JJ) < 4 || j_ < 17)                                                                                \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"               \
                         ::"s"((SLOT_LDS) + j_ * 1024), "v"(dk_off[JJ]), "s"(SRD) : "memory", "m0");            \
    }
#define W_DMA_V(SRD, SLOT_LDS, JJ)                                                                              \
    {                                                                                                           \
        const int j_ = wave + 4 * (JJ);                                                                         \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"                  \
                     ::"s"((SLOT_LDS) + j_ * 1024), "v"(dv_off), "s"(SRD), "s"(j_ * dv_soff) : "memory", "m0"); \
    }
#define W_M0(SLOT_LDS, PIECE) asm volatile("s_add_u32 m0, %0, %1" ::"s"(SLOT_LDS), "s"((wave + 4 * (PIECE)) * 1024) : "memory", "m0", "scc")
#define W_LD_K(SRD, JJ) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(dk_off[JJ]), "s"(SRD) : "memory")
#define W_LD_V(SRD, JJ) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(dv_off), "s"(SRD), "s"((wave + 4 * (JJ)) * dv_soff) : "memory")
#define W_KSLOT(T) (lds0 + (uint32_t)((T) & (W_NK - 1)) * W_KSTAGE)
#define W_VSLOT(T) (lds0 + W_VBASE + (uint32_t)((T) % W_NV) * W_VSTAGE)
    auto dma_k = [&](int tile) __attribute__((always_inline)) {
        const w_srd_t s_ = tile_srd(kp, kst_b, tile);
        const uint32_t st_ = W_KSLOT(tile);
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) W_DMA_K(s_, st_, jj);
    };
    auto dma_v = [&](int tile) __attribute__((always_inline)) {
        const w_srd_t s_ = vt_srd(tile);
        const uint32_t st_ = W_VSLOT(tile);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) W_DMA_V(s_, st_, jj);
    };

    // per-lane fragment addresses inside a stage (everything else is an immediate)
    //   K: this lane's row of a 32-key half = l31 with bits 2 and 3 swapped (see the header: keys of a P^T fragment contiguous)
    //   V^T: row d = l31 (+ 32 dt: immediate), chunk 2 g + half of the row (g = 16-key group), XOR (d >> 1) & 7
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const uint32_t k_rd = lds0 + (uint32_t)(krow * W_KROW + half * 16);
    uint32_t v_rd[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) v_rd[g] = lds0 + (uint32_t)(W_VBASE + l31 * 128 + (((2 * g + half) ^ ((l31 >> 1) & 7)) << 4));

    f32x16_t oacc[2][4];                          // O^T accumulators: AGPRs
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[x][dt][r] = 0.f;
            asm volatile("" : "+a"(oacc[x][dt]));
        }
    // Online softmax state of the lane's two rows.  A row's exponentials are taken relative to a reference point -nm (scaled, log2
    // domain) that moves only when a tile's largest exponent exceeds W_THR (or at the row's first visible key); `seen` = it has one.
    float nm[2] = {0.f, 0.f};                     // -(reference point); 0 until the row has seen a key
    uint32_t seenm[2] = {0u, 0u};                // per lane: all ones once the row has a reference point (non-PRE) / has seen a key (PRE)
    float alpha[2] = {1.f, 1.f};                  // factor the pending tile applies to O and l when `resc`
    // The softmax denominators ride on the matrix pipe: l^T[.][q] = ones . P^T, one more MFMA per 16-key group and query block (8 of
    // 72 per trip) with an all-ones A fragment.  Every row of the 32 x 32 result holds the column sums of the bf16 P the numerator
    // uses.  The wave is bound by instruction ISSUE, not by the pipe (profiles/r05_attn_w64_cycles_by_ablation.txt): 8 MFMA issues
    // replace 64 v_add_f32 (+ the running-sum bookkeeping).
    f32x16_t lacc[2];                             // AGPRs
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
        for (int r = 0; r < 16; ++r) lacc[x][r] = 0.f;
        asm volatile("" : "+a"(lacc[x]));
    }
    const w_u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#if !W_LSUM_MFMA
    float l_run[2] = {0.f, 0.f}, psa[2] = {0.f, 0.f}, psb[2] = {0.f, 0.f};
#endif
    bool resc = false;                            // wave-uniform: some row of the wave moved its reference point for the pending tile
    const float c_sc = a.scale_log2;

    f32x16_t WK[2];                               // QK^T accumulation of the tuple in progress [query block]: VGPRs, MFMA operands only
    f32x16_t S[2][2];                             // exponent tuples [query block][32-key half]: written by a tuple's last QK^T MFMA (raw scores of the
                                                  // NEXT tile), turned into s c - reference in place by the side stream, read by the next trip's exp stream
#define W_EL(X, I) S[X][(I) >> 4][(I) & 15]
    uint32_t pk[2][16];                           // bf16-packed P^T of the current tile [query block][4 * (16-key group) + word]
    w_u32x4 kf[4];                                // K fragments: a ring of four (see w_p1_ops): AGPRs

    // ---- phase-1 stream: P = 2^e and the bf16 pack -- 96 instructions, in the order P.V consumes the words ---------------------------
    // unit u: 16-key group g = u >> 3, query block x = (u >> 2) & 1, word w = u & 3 (two scores);  group G = units 2G, 2G+1:
    // 6 instructions: 4 x exp2, 2 x pack
    float tp[4];
#if W_LSUM_MFMA
    auto exp_op = [&](const int G, const int o) __attribute__((always_inline)) {
        const int which = o < 4 ? (o >> 1) : (o - 4);
        const int u = 2 * G + which;
        const int g = u >> 3, x = (u >> 2) & 1, w = u & 3;
        const int r = 16 * (g >> 1) + 8 * (g & 1) + 2 * w;                     // index into ev[x]: 16 kt + register of the MFMA tile
#ifdef W_ABL_EXPMUL
        if (o < 4) { tp[o] = W_EL(x, r + (o & 1)) * c_sc; W_PIN(tp[o]); }
#else
        if (o < 4) { tp[o] = __builtin_amdgcn_exp2f(W_EL(x, r + (o & 1))); W_PIN(tp[o]); }
#endif
        else { pk[x][4 * g + w] = pack_bf2(tp[2 * which], tp[2 * which + 1]); W_PIN(pk[x][4 * g + w]); }
    };

#else
    // (the form with the row sums in the VALU stream: 10 instructions per group -- 4 x exp2, 4 x add, 2 x pack; 160 per trip)
    auto exp_op = [&](const int G, const int o) __attribute__((always_inline)) {
        const int which = o < 8 ? ((o & 3) >> 1) : (o - 8);
        const int u = 2 * G + which;
        const int g = u >> 3, x = (u >> 2) & 1, w = u & 3;
        const int r = 16 * (g >> 1) + 8 * (g & 1) + 2 * w;
#ifdef W_ABL_EXPMUL
        if (o < 4) { tp[o] = W_EL(x, r + (o & 1)) * c_sc; W_PIN(tp[o]); }
#else
        if (o < 4) { tp[o] = __builtin_amdgcn_exp2f(W_EL(x, r + (o & 1))); W_PIN(tp[o]); }
#endif
        else if (o < 8) {
            const bool first_of_row = w == 0 && g == 0;
            if ((o & 1) == 0) { psa[x] = first_of_row ? tp[o - 4] : psa[x] + tp[o - 4]; W_PIN(psa[x]); }
            else { psb[x] = first_of_row ? tp[o - 4] : psb[x] + tp[o - 4]; W_PIN(psb[x]); }
        } else { pk[x][4 * g + w] = pack_bf2(tp[2 * which], tp[2 * which + 1]); W_PIN(pk[x][4 * g + w]); }
    };
#endif

    // ---- phase-2 side stream on the NEXT tile's scores: exponents, row max, the reference-point bookkeeping --------------------------
    // every instruction is the compiler's own (it pads its hazards; fmaxf of FMA results needs no canonicalising v_max; the file is
    // built with -fno-slp-vectorize: beside MFMAs a packed fp32 instruction costs more than the two it replaces)
    float tmx[2], dl[2] = {0.f, 0.f};
    bool upd_any = false;
    auto fma_op = [&](const int kt, const int k) __attribute__((always_inline)) {                               // k = 0..31: x = k >> 4, r = k & 15
        const int x = k >> 4, r = k & 15;
        S[x][kt][r] = __builtin_fmaf(S[x][kt][r], c_sc, nm[x]);              // (in place: the tuple's last MFMA retired >= 8 gaps ago)
        W_PIN(S[x][kt][r]);
    };
    auto max_op = [&](const int k) __attribute__((always_inline)) {              // k = 0..31: x = k & 1 (the two chains alternate), step k >> 1
        const int x = k & 1, st = k >> 1;
        if (st == 0) tmx[x] = fmaxf(fmaxf(W_EL(x, 0), W_EL(x, 1)), W_EL(x, 2));
        else if (st < 15) tmx[x] = fmaxf(fmaxf(tmx[x], W_EL(x, 2 * st + 1)), W_EL(x, 2 * st + 2));
        else tmx[x] = fmaxf(tmx[x], W_EL(x, 31));
        W_PIN(tmx[x]);
    };
    // the rows' reference points for tile t+1, in four steps (both query blocks side by side: two independent dependency chains per step)
    float emx[2];
    uint32_t updm[2];                             // per lane: all ones = the row's reference point moves for the pending tile
    auto book_a = [&]() __attribute__((always_inline)) {                        // the row's largest exponent in this tile (-inf: all masked)
        emx[0] = w_xor32_max(tmx[0]); emx[1] = w_xor32_max(tmx[1]);
        W_PIN(emx[0]); W_PIN(emx[1]);
    };
    bool any_nm = false;                          // PRE, wave-uniform: some row of the wave has left the reference point 0
    uint32_t gotm[2] = {0u, 0u};                  // PRE: the tile has a visible key for the row
    // The per-row decisions are written as 32-bit lane MASKS and bitwise arithmetic, not as bool expressions (round 6): from `a ? b : c` and
    // `a || b` on per-lane bools hipcc built s_and_saveexec / s_or exec regions, and every write of EXEC in the middle of the MFMA stream waits
    // for the matrix pipe to drain -- eight of them per trip cost ~660 of a trip's 3,880 cycles (tools/attn_phase_profile.py with
    // -DW_ABL_NOBOOK: phase 2 1,715 -> 1,056 cycles; profiles/r06_attn_notes.txt).  Selects on masks are v_cmp + v_cndmask: no EXEC write.
    auto book_b = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float e = emx[x];
            const uint32_t got_ = e > -INFINITY ? 0xffffffffu : 0u;
            if constexpr (PRE) {
                // the reference stays where it is (0 from the start) while the tile's largest exponent is inside +-W_THRP; a row's FIRST
                // visible tile may also pull it down (nothing is accumulated yet: alpha = 0 below, nothing is rescaled)
                const uint32_t up_ = e > W_THRP ? 0xffffffffu : 0u, lo_ = e < -W_THRP ? 0xffffffffu : 0u;
                gotm[x] = got_;
                updm[x] = up_ | (~seenm[x] & got_ & lo_);
            } else {
                const uint32_t up_ = e > W_THR ? 0xffffffffu : 0u;
                updm[x] = (seenm[x] & up_) | (~seenm[x] & got_);            // (a row's first visible key always sets its reference point)
            }
            dl[x] = __uint_as_float(__float_as_uint(e) & updm[x]);          // e where the row moves, +0.0 elsewhere
        }
        W_PIN(dl[0]); W_PIN(dl[1]);
    };
    auto book_c = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            // 1 exactly when the row stays; 0 for a row that has not seen a key yet (its l, O are still 0)
            alpha[x] = __uint_as_float(__float_as_uint(__builtin_amdgcn_exp2f(-dl[x])) & seenm[x]);
            nm[x] -= dl[x];
        }
        W_PIN(alpha[0]); W_PIN(alpha[1]); W_PIN(nm[0]); W_PIN(nm[1]);
    };
    auto book_d = [&]() __attribute__((always_inline)) {
        if constexpr (PRE) { seenm[0] |= gotm[0]; seenm[1] |= gotm[1]; }
        else { seenm[0] |= updm[0]; seenm[1] |= updm[1]; }
        upd_any = __builtin_amdgcn_ballot_w64((updm[0] | updm[1]) != 0u) != 0ull;     // one v_cmp into an SGPR pair + a scalar compare
        resc = upd_any;
        if constexpr (PRE) any_nm = any_nm | upd_any;
    };
    // side work of P.V gap j: 120 instructions spread at <= 4 per gap; the second-half score tuples (last written by the MFMAs of
    // phase-1 gaps 30 / 31) are first read in gap 8
    // PRE + W_EARLY (an experiment that LOST, kept as a build knob with its number).  Hypothesis: phase 1 is the long pole -- 64 quarter-rate
    // v_exp_f32 + ~100 other instructions of the exp stream beside 1,024 MFMA cycles, while phase 2 carries ~130 full-rate ones and, with
    // the queries pre-scaled, not even the 64 fma (removing those alone bought only 0.7 %).  So half of the NEXT tile's exp stream moves
    // here: its first 32 keys (tuples kt = 0, deposited 16+ gaps ago) are exponentiated under P.V of the current tile, in the P^T words P.V has
    // already consumed (group g's words are free behind gap 8 g + 7) -- BEFORE the tile's row maximum is known.  That is safe because
    // the reference point almost never moves (PRE: only beyond +-64 log2 units); when it does (`upd_any`, gap 28) the early words and
    // row sums are simply formed again from the re-based exponents.  Correct (every attention test green) and 2.7 % SLOWER than W_EARLY 0 in
    // the same process: the trip is not bound by where the exp stream sits -- see DESIGN 11.3.
    //   gaps 0..6, 8..16   row max (kt = 0 elements first: the kt = 1 tuple was deposited in phase-1 gaps 30 / 31)
    //   gaps 8..17         early exp of 16-key group 0  (40 instructions)        gaps 18..27  ... of group 1
    //   gaps 17..20        the reference-point bookkeeping                          gap 28       rare: re-base + redo the early groups
    auto side_early = [&](const int j) __attribute__((always_inline)) {
        if (j < 7) { max_op(2 * j); max_op(2 * j + 1); }
        else if (j >= 8 && j <= 16) { max_op(2 * (j - 1)); max_op(2 * (j - 1) + 1); }
        if (j >= 8 && j <= 27) {
            const int q = 4 * (j - 8);                                          // 80 instructions, four per gap: groups G = 0..7
            exp_op(q / 10, q % 10); exp_op((q + 1) / 10, (q + 1) % 10); exp_op((q + 2) / 10, (q + 2) % 10); exp_op((q + 3) / 10, (q + 3) % 10);
        }
        if (j == 17) book_a();
        else if (j == 18) book_b();
        else if (j == 19) book_c();
        else if (j == 20) book_d();
        else if (j == 28) {
            if (upd_any) {                          // rare: re-base the exponents of the rows that moved, then the early groups once more
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int r = 0; r < 32; ++r) W_EL(x, r) -= dl[x];
                w_static_for<80>([&](auto qc) __attribute__((always_inline)) { W_USE3(S, pk, tp); exp_op(decltype(qc)::v / 10, decltype(qc)::v % 10); });
            }
        }
    };
    auto side = [&](const int j) __attribute__((always_inline)) {
        // (no loops here: a loop that contains a pin is unrolled too late for the register promotion of S / ev)
        if constexpr (PRE && W_EARLY && !W_LSUM_MFMA) { side_early(j); return; }
        if (j < 8) {                                // exponents of the first-half tuples (written >= 16 MFMAs ago); PRE: the scores ARE the exponents
            if constexpr (!PRE) { fma_op(0, 4 * j); fma_op(0, 4 * j + 1); fma_op(0, 4 * j + 2); fma_op(0, 4 * j + 3); }
        } else if (j < 16) {                        // ... of the second-half tuples
            if constexpr (!PRE) { fma_op(1, 4 * (j - 8)); fma_op(1, 4 * (j - 8) + 1); fma_op(1, 4 * (j - 8) + 2); fma_op(1, 4 * (j - 8) + 3); }
        } else if (j < 24) {                        // row max: 16 steps per query block
#ifndef W_ABL_NOMAX
            max_op(4 * (j - 16)); max_op(4 * (j - 16) + 1); max_op(4 * (j - 16) + 2); max_op(4 * (j - 16) + 3);
#endif
#ifndef W_ABL_NOBOOK
        } else if (j == 24) { book_a();
        } else if (j == 25) { book_b();
        } else if (j == 26) { book_c();
        } else if (j == 27) { book_d();
#endif
        } else if (j == 28) {
#ifndef W_ABL_NOBOOK
            if (upd_any) {                          // rare (deferred max): re-base the exponents of the rows that moved
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int r = 0; r < 32; ++r) W_EL(x, r) -= dl[x];
            }
#endif
        }
    };
    auto mask_tile = [&](const int tile) __attribute__((always_inline)) {                                      // diagonal / ragged tiles only: S = -inf beyond the row's limit
        const int64_t k0 = (int64_t)tile * KB;
        int64_t d64 = wpos0 - k0;                                               // wave-uniform part of the limit
        d64 = d64 > 1000 ? 1000 : (d64 < -1000 ? -1000 : d64);
        const int64_t e64 = Tk_ - 1 - k0;
        const int endl = e64 > 63 ? 63 : (e64 < -1 ? -1 : (int)e64);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            int lim = lim_rel[x] + (int)d64;
            lim = (lim > endl ? endl : lim) - 8 * half;                         // key index 32 kt + 16 (r >> 3) + 8 half + (r & 7) <= limit
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[x][kt][r] = (32 * kt + 16 * (r >> 3) + (r & 7)) <= lim ? S[x][kt][r] : -INFINITY;
        }
    };
    auto add_nm = [&]() __attribute__((always_inline)) {                        // PRE, rare: rows that left the reference point 0 get their offset
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 32; ++r) W_EL(x, r) += nm[x];
    };
    // a tile needs masking from the first one that reaches past the wave's first row (k0 + 63 > wpos0) or past the last key (k0 + 64 > Tk)
    const int mask_from = (int)((wpos0 + 1) / KB) < n_full ? (int)((wpos0 + 1) / KB) : n_full;

    // ---- prologue -----------------------------------------------------------------------------------------------------------------
    // the first tiles' DMA goes out BEFORE the Q fragment loads (one memory latency per workgroup instead of two); the compiler's own
    // wait for those loads then also covers the (older) DMA pieces
    dma_k(0); dma_v(0);
    dma_k(1); dma_k(2);
    dma_k(3); dma_v(1);
    w_u32x4 qf[2][8];                                                   // Q fragments: AGPRs from here on (only ever "a" operands)
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int64_t qc = qrow[x] < 0 ? 0 : qrow[x];                   // (rows past the end cannot occur: blocks end at Tq)
        const w_u32x4* qr = (const w_u32x4*)(qp + qc * a.q_st);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[x][ks] = qr[2 * ks + half];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+a"(qf[x][ks]));
    W_BARRIER();
    {
        const uint32_t kb = k_rd + 0 * W_KSTAGE;           // tile 0 -> K slot 0: four batches of four fragments
#define W_PRO_BATCH(F0)                                                                                                             \
        w_static_for<4>([&](auto jc) __attribute__((always_inline)) { W_USE2(kf, kb); constexpr int f = (F0) + decltype(jc)::v;       \
            W_DSR_K(kf[f & 3], kb, (f >> 3) * (32 * W_KROW) + (f & 7) * 32); });                                                      \
        W_LGKM(0);                                                                                                                    \
        w_static_for<4>([&](auto jc) __attribute__((always_inline)) {                                                                 \
            W_USE3(kf, qf, S); (void)WK;                                                                                              \
            constexpr int f = (F0) + decltype(jc)::v, kt = f >> 3, ks = f & 7;                                                        \
            if (ks == 0) { W_MFMA_S0(WK[0], kf[f & 3], qf[0][ks]); W_MFMA_S0(WK[1], kf[f & 3], qf[1][ks]); }                          \
            else if (ks < 7) { W_MFMA_S(WK[0], kf[f & 3], qf[0][ks]); W_MFMA_S(WK[1], kf[f & 3], qf[1][ks]); }                        \
            else { W_MFMA_SD(S[0][kt], WK[0], kf[f & 3], qf[0][ks]); W_MFMA_SD(S[1][kt], WK[1], kf[f & 3], qf[1][ks]); }              \
        });
        W_PRO_BATCH(0) W_PRO_BATCH(4) W_PRO_BATCH(8) W_PRO_BATCH(12)
        W_NOP24();                                         // MFMA results -> the VALU below
        if (0 >= mask_from) mask_tile(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PRE && W_EARLY && !W_LSUM_MFMA) {
            // tile 0: row max and bookkeeping first, then (re-based if a row moved) the early half of its exp stream -- nothing speculative here
            w_static_for<32>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::v;
                if (j < 7) { max_op(2 * j); max_op(2 * j + 1); }
                else if (j >= 8 && j <= 16) { max_op(2 * (j - 1)); max_op(2 * (j - 1) + 1); }
            });
            book_a(); book_b(); book_c(); book_d();
            if (upd_any) {
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int r = 0; r < 32; ++r) W_EL(x, r) -= dl[x];
            }
            w_static_for<80>([&](auto qc) __attribute__((always_inline)) { W_USE3(S, pk, tp); exp_op(decltype(qc)::v / 10, decltype(qc)::v % 10); });
        } else {
            w_static_for<32>([&](auto jc) __attribute__((always_inline)) { side(decltype(jc)::v); });
        }
        resc = false;                                      // (O is still zero)
        const uint32_t kb1 = k_rd + 1 * W_KSTAGE;          // tile 1 -> K slot 1: its first-half fragments
        w_static_for<4>([&](auto jc) __attribute__((always_inline)) { W_USE2(kf, kb1); constexpr int ks = decltype(jc)::v; W_DSR_K(kf[ks], kb1, ks * 32); });
        W_LGKM(0);
        W_BARRIER();                                       // every wave is done with K slot 0 before trip 0 refills it with tile 4
    }

    // ---- one trip per key tile: `tile` = cur (its exponents in ev), tile + 1 = nxt ---------------------------------------------------
    // Everything a trip needs besides its data is computed UNDER the previous trip's P.V MFMAs and carried across the back edge (the
    // first version did it at the head of the trip: ~50 scalar instructions between the barrier and the first MFMA, with no MFMA in
    // flight -- profiles/r05_attn_w64_ablation_v1.txt: 93 ms without any softmax instruction against 67 ms for the bare MFMAs):
    //   kb          LDS address of this lane's K fragments of tile + 1          vb[4]   ... of its V^T fragments of tile (one per 16-key group)
    //   ksrd, kslot descriptor / LDS slot of the K tile this trip fetches (tile + 4)     vsrd, vslot: the V tile (tile + 2)
    //   mask_nxt    tile + 1 needs masking
    uint32_t kb = k_rd + 1 * W_KSTAGE;
    uint32_t vb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) vb[g] = v_rd[g];
    w_srd_t ksrd = tile_srd(kp, kst_b, 4), vsrd = vt_srd(2);
    uint32_t kslot = W_KSLOT(4), vslot = W_VSLOT(2);
    bool mask_nxt = 1 >= mask_from;
    int v_idx = 0;                                         // tile % 3
    uint32_t pc_off[9];                                    // LDS offset of this wave's DMA pieces inside a K / V slot
#pragma unroll
    for (int pc = 0; pc < 9; ++pc) pc_off[pc] = (uint32_t)(wave + 4 * (pc < 5 ? pc : pc - 5)) * 1024u;
#define W_M0P(SLOT_LDS, PC) asm volatile("s_add_u32 m0, %0, %1" ::"s"(SLOT_LDS), "s"(pc_off[PC]) : "memory", "m0", "scc")
#define W_PIN_S(X) asm volatile("" ::"s"(X))

#if W_PROFILE
    uint64_t tp_[6] = {0, 0, 0, 0, 0, 0}, tl_ = __builtin_readcyclecounter();
#endif
    for (int tile = 0; tile < n_tiles; ++tile) {
        if (resc) {                                        // rare (deferred max): some row moved its reference point for this tile
            W_NOP24();
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    oacc[x][dt] = oacc[x][dt] * alpha[x];
                    asm volatile("" : "+a"(oacc[x][dt]));
                }
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                lacc[x] = lacc[x] * alpha[x];
                asm volatile("" : "+a"(lacc[x]));
            }
            W_NOP24();
        }

        W_STAMP(0);
        // ---- phase 1: 32 x { QK^T(tile+1) MFMA | second-half K fragment reads | exp stream of tile | DMA pieces | first V^T fragments } ---
        w_u32x4 vfr[W_VD + 1] = {};                        // V^T fragments in flight (fragment p = (16-key group p >> 2, d tile p & 3))
        w_static_for<32>([&](auto ic) __attribute__((always_inline)) {
            W_USE3(kf, qf, S); W_USE3(kb, vfr, vb); W_USE2(WK, pk); W_USE3(dk_off, dv_off, pc_off); W_USE3(ksrd, vsrd, kslot); W_USE2(vslot, dv_soff);
            constexpr int i = decltype(ic)::v;
            constexpr int kt = i >> 4, ks = (i >> 1) & 7, x = i & 1;
            constexpr bool dma = (i % 3) == 1 && i < 27;   // gaps 1, 4, ..., 25 -> pieces 0..8 (0-4: K, 5-8: V; K piece 4 exists in wave 0 only)
            constexpr int pc = i / 3;
#ifndef W_ABL_NODMA
            if (dma) {                                     // m0 at the head of the gap, the load at its end: the MFMA between them is the wait state
                if (pc < 4) W_M0P(kslot, pc);
                else if (pc == 4) { if (wave == 0) W_M0P(kslot, pc); }
                else W_M0P(vslot, pc);
            }
#endif
            constexpr int f = i >> 1;                     // K fragment of this MFMA
            if (f >= 4 && x == 0) W_T_LGKM(w_wait_k1(f));
            if (ks == 0) W_MFMA_S0(WK[x], kf[f & 3], qf[x][ks]);
            else if (ks < 7) W_MFMA_S(WK[x], kf[f & 3], qf[x][ks]);
            else W_MFMA_SD(S[x][kt], WK[x], kf[f & 3], qf[x][ks]);   // deposit: tile t's exponents of half kt were last read in gap 16 kt + 14
            if (x == 1 && f + 4 < 16) W_T_DSR_K(kf[f & 3], kb, ((f + 4) >> 3) * (32 * W_KROW) + ((f + 4) & 7) * 32);
            if (i >= 32 - 2 * W_VD && (i & 1) == 0) {     // the first W_VD V^T fragments of P.V(tile)
                constexpr int p = (i - (32 - 2 * W_VD)) >> 1;
#ifdef W_ABL_VADDRK
                W_T_DSR_V(vfr[p], kb, (p & 3) * 32);
#else
                W_T_DSR_V(vfr[p], vb[p >> 2], (p & 3) * 4096);
#endif
            }
            // three instructions of the exp stream per gap (96 = 32 x 3)
#ifndef W_ABL_NOEXP
#if W_LSUM_MFMA
            exp_op((3 * i) / 6, (3 * i) % 6); exp_op((3 * i + 1) / 6, (3 * i + 1) % 6); exp_op((3 * i + 2) / 6, (3 * i + 2) % 6);
#else
            if constexpr (PRE && W_EARLY) {
                // the tile's LAST 32 keys only (groups G = 8..15: the first 32 went under the previous trip's P.V): 80 instructions, three
                // per gap -- done by gap 26, the tuples they read are overwritten by the deposits of gaps 30 / 31
                if constexpr (3 * i < 80) exp_op(8 + (3 * i) / 10, (3 * i) % 10);
                if constexpr (3 * i + 1 < 80) exp_op(8 + (3 * i + 1) / 10, (3 * i + 1) % 10);
                if constexpr (3 * i + 2 < 80) exp_op(8 + (3 * i + 2) / 10, (3 * i + 2) % 10);
            } else {
                exp_op((5 * i) / 10, (5 * i) % 10); exp_op((5 * i + 1) / 10, (5 * i + 1) % 10); exp_op((5 * i + 2) / 10, (5 * i + 2) % 10);
                exp_op((5 * i + 3) / 10, (5 * i + 3) % 10); exp_op((5 * i + 4) / 10, (5 * i + 4) % 10);
            }
#endif
#endif
#ifndef W_ABL_NODMA
            if (dma) {
                if (pc < 4) W_LD_K(ksrd, pc);
                else if (pc == 4) { if (wave == 0) W_LD_K(ksrd, pc); }
                else W_LD_V(vsrd, pc - 5);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
        W_STAMP(1);
#if !W_LSUM_MFMA
#pragma unroll
        for (int x = 0; x < 2; ++x) l_run[x] = fmaf(l_run[x], alpha[x], psa[x] + psb[x]);
#endif
        if (mask_nxt || (PRE && any_nm)) {                 // rare: the diagonal / ragged tile's mask; PRE: rows that left the reference point 0
            W_NOP24();
            if (mask_nxt) mask_tile(tile + 1);
            if constexpr (PRE) { if (any_nm) add_nm(); }
        }
        __builtin_amdgcn_sched_barrier(0);

        W_STAMP(2);
        // ---- phase 2: 32 x { P.V(tile) MFMA | V^T fragment reads | side stream on S | first-half K(tile+2) | the next trip's setup } ------
        uint32_t kb2 = 0, vb_n[4] = {0, 0, 0, 0}, kslot_n = 0, vslot_n = 0;
        int v_idx_n = 0;
        w_srd_t ksrd_n = ksrd, vsrd_n = vsrd;
        bool mask_nxt_n = false;
        w_static_for<32>([&](auto jc) __attribute__((always_inline)) {
            W_USE3(kf, oacc, kb2); W_USE3(vb, vfr, pk); W_USE2(S, nm); W_USE2(lacc, ones);
            constexpr int j = decltype(jc)::v;
            constexpr int p = j >> 1, x = j & 1;
            constexpr int g = p >> 2, dt = p & 3;
            if (x == 0) W_T_LGKM(w_wait_v(p));
            w_u32x4 pf;
            pf.x = pk[x][4 * g]; pf.y = pk[x][4 * g + 1]; pf.z = pk[x][4 * g + 2]; pf.w = pk[x][4 * g + 3];
            W_MFMA_O(oacc[x][dt], vfr[p % (W_VD + 1)], pf);
#if W_LSUM_MFMA && !defined(W_ABL_NOLSUM)
            if (dt == 3) W_MFMA_L(lacc[x], ones, pf);     // the group's column sums (its P^T fragment is in registers now)
#endif
            if (x == 0 && p + W_VD < 16) {
                constexpr int pp = p + W_VD;
#ifdef W_ABL_VADDRK
                W_T_DSR_V(vfr[pp % (W_VD + 1)], kb2, (pp & 3) * 32);
#else
                W_T_DSR_V(vfr[pp % (W_VD + 1)], vb[pp >> 2], (pp & 3) * 4096);
#endif
            }
            if (j >= 10 && j <= 22 && ((j - 10) & 3) == 0) W_T_DSR_K(kf[(j - 10) >> 2], kb2, ((j - 10) >> 2) * 32);
#ifndef W_ABL_NOSIDE
            side(j);
#endif
            // the next trip's setup, a few scalar instructions per gap (pinned: they would otherwise collect at the head of the loop)
            if (j == 3) { kb2 = k_rd + (uint32_t)((tile + 2) & (W_NK - 1)) * W_KSTAGE; W_PIN(kb2); }      // = the next trip's kb
            if (j == 5) { v_idx_n = v_idx == W_NV - 1 ? 0 : v_idx + 1; W_PIN_S(v_idx_n); }
            if (j == 7) { const uint32_t vo = (uint32_t)v_idx_n * W_VSTAGE; vb_n[0] = v_rd[0] + vo; vb_n[1] = v_rd[1] + vo; W_PIN(vb_n[0]); W_PIN(vb_n[1]); }
            if (j == 9) { const uint32_t vo = (uint32_t)v_idx_n * W_VSTAGE; vb_n[2] = v_rd[2] + vo; vb_n[3] = v_rd[3] + vo; W_PIN(vb_n[2]); W_PIN(vb_n[3]); }
            if (j == 11) { kslot_n = W_KSLOT(tile + 5); W_PIN_S(kslot_n); }
            if (j == 13) { vslot_n = lds0 + W_VBASE + (uint32_t)v_idx * W_VSTAGE; W_PIN_S(vslot_n); }          // V(tile + 3) -> slot (tile + 3) % 3 = tile's own slot, free after this trip
            if (j == 15) { ksrd_n = tile_srd(kp, kst_b, tile + 5); W_PIN_S(ksrd_n); }
            if (j == 19) { vsrd_n = vt_srd(tile + 3); W_PIN_S(vsrd_n); }
            if (j == 29) { mask_nxt_n = tile + 2 >= mask_from; }
            __builtin_amdgcn_sched_barrier(0);
        });
        W_STAMP(3);
        kb = kb2; v_idx = v_idx_n; ksrd = ksrd_n; vsrd = vsrd_n; kslot = kslot_n; vslot = vslot_n; mask_nxt = mask_nxt_n;
#pragma unroll
        for (int g = 0; g < 4; ++g) vb[g] = vb_n[g];
        // tile+1's V and tile+3's K must have landed before the next trip (this trip's pieces may stay in flight); every LDS read of
        // this trip has returned (the K fragments of the next trip's first MFMAs among them)
        W_T_LGKM(0);
#ifndef W_ABL_NODMA
        W_WAIT(1);
#endif
        W_STAMP(4);
#ifndef W_ABL_NOBAR
        W_BARRIER();
#endif
        W_STAMP(5);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing (empty) DMA pieces: nothing may land after exit
    W_NOP24();
    // ---- epilogue: normalise; the two halves of a row trade 8-byte pieces so that every lane stores 16 contiguous bytes -------------------
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#if W_LSUM_MFMA
        const float l_tot = lacc[x][0];                                         // (every row of l^T holds the column's sum over all keys)
#else
        const float l_tot = w_xor32_add(l_run[x]);
#endif
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        const bool ok = x == 0 ? row_ok0 : row_ok1;
        uint16_t* orow = a.o + ((int64_t)(bat * a.Tq + (x == 0 ? orow_first : orow_second)) * a.H + head) * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x16_t ov = oacc[x][dt];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ga = 2 * i, gb = 2 * i + 1;
                const uint32_t ax = pack_bf2(ov[4 * ga] * inv, ov[4 * ga + 1] * inv);
                const uint32_t ay = pack_bf2(ov[4 * ga + 2] * inv, ov[4 * ga + 3] * inv);
                const uint32_t bx = pack_bf2(ov[4 * gb] * inv, ov[4 * gb + 1] * inv);
                const uint32_t by = pack_bf2(ov[4 * gb + 2] * inv, ov[4 * gb + 3] * inv);
                // swap(A, B): A's upper lanes <-> B's lower lanes.  Afterwards lanes 0-31 hold group 2i (d 0-3 | d 4-7), lanes 32-63 group 2i+1.
                auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                uint4 w;
                w.x = sx[0]; w.y = sy[0]; w.z = sx[1]; w.w = sy[1];
                if (ok) *(uint4*)(orow + 32 * dt + 8 * (2 * i + half)) = w;
            }
        }
    }
#if W_PROFILE
    if (lane == 0) {                                        // 8 floats per wave at o + 32 B * (4 * workgroup + wave): a timing build, it overwrites outputs
        float* dbg = (float*)a.o + 8 * (4 * (int64_t)blockIdx.x + wave);
        for (int k = 0; k < 6; ++k) dbg[k] = (float)tp_[k];
        dbg[6] = (float)n_tiles;
        dbg[7] = (float)qblk;
    }
#endif
}

// V [B, Tk, H, 128] (strided rows) -> V^T [B][H][128][vt_row] (vt_row = Tk rounded up to 64, the pad keys zero): one workgroup per
// (64-key tile, head, batch row).  Whole 256-byte rows in (16 lanes each), whole 128-byte row pieces out (8 lanes each); the tile
// turns in LDS ([64 keys][130] halves: both sides conflict-free).  2 bytes moved per byte of V.
__global__ __launch_bounds__(256) void attn_vt_kernel(const uint16_t* __restrict__ v, uint16_t* __restrict__ vt, int64_t Tk, int64_t v_sb,
                                                      int64_t v_st, int64_t v_sh, int64_t vt_row, int H) {
    __shared__ uint16_t tile[KB][DH + 2];
    const int tid = threadIdx.x, head = blockIdx.y, bat = blockIdx.z;
    const int64_t k0 = (int64_t)blockIdx.x * KB;
    const uint16_t* vp = v + bat * v_sb + head * v_sh;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int key = (tid >> 4) + 16 * ps, ch = tid & 15;
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (k0 + key < Tk) w = *(const uint4*)(vp + (k0 + key) * v_st + 8 * ch);
        uint32_t* dst = (uint32_t*)&tile[key][8 * ch];           // (row pitch 260 B: 4-byte aligned)
        dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
    }
    __syncthreads();
    uint16_t* op = vt + (((int64_t)bat * H + head) * DH) * vt_row + k0;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int d = (tid >> 3) + 32 * ps, c = tid & 7;          // 8 lanes = one 128-byte piece of row d
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (uint32_t)tile[8 * c + 2 * i][d] | ((uint32_t)tile[8 * c + 2 * i + 1][d] << 16);
        *(uint4*)(op + (int64_t)d * vt_row + 8 * c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// Host side: geometry + launches (called by evo_attn_fwd_causal_bf16 in csrc/attn.hip for query ranges longer than one 128-row block).
// vt_ws: B * H * 128 * (Tk rounded up to 64) bf16 of workspace for V^T (include/evo_mi355x.h).
int evo_attn_w64_launch(AttnArgs a, int64_t B, void* vt_ws, void* stream) {
    a.nbh = (int)(B * a.H);
    a.n_qblocks = (int)((a.Tq + W_QB - 1) / W_QB);
    a.q_pad = (int)((int64_t)a.n_qblocks * W_QB - a.Tq);
    a.vt = (const uint16_t*)vt_ws;
    a.vt_row = (a.Tk + KB - 1) / KB * KB;
    const int64_t n_wg = (int64_t)a.n_qblocks * a.nbh;
    if (n_wg > 0x7fffffff || a.vt_row / KB > 0x7fffffff || DH * a.vt_row * 2 > 0xffffffffll) return -1;
    hipLaunchKernelGGL(attn_vt_kernel, dim3((unsigned)(a.vt_row / KB), (unsigned)a.H, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       a.v, (uint16_t*)vt_ws, a.Tk, a.v_sb, a.v_st, a.v_sh, a.vt_row, a.H);
    if (a.prescaled) hipLaunchKernelGGL(attn_fwd_w64_kernel<true>, dim3((unsigned)n_wg), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_fwd_w64_kernel<false>, dim3((unsigned)n_wg), dim3(256), 0, (hipStream_t)stream, a);
    return evo_launch_status();
}
