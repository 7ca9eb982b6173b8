// Shared declarations of the attention kernels (csrc/attn.hip, csrc/attn_w64.hip): launch arguments, the workgroup -> (query block,
// head, batch) map and the MFMA fragment type.
#pragma once
#include "common.h"

#define QB 128
#define KB 64
#define DH 128

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ mfma_bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(mfma_bf16x8, v); }

struct AttnArgs {
    const uint16_t* q; const uint16_t* k; const uint16_t* v; uint16_t* o;
    int64_t Tq, Tk, q_pos0;
    int64_t q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh;
    int H;
    float scale_log2;      // softmax_scale * log2(e); 1 when `prescaled`
    int prescaled;         // the queries carry softmax_scale * log2(e) already (evo_rope_qk_bf16's q_scale): scores are exponents (log2 domain)
    int n_qblocks;
    // decode (split-K) mode only
    const int64_t* dyn_pos;   // device int64 [B]: position of each row's (single) query; overrides Tk / q_pos0 when non-null
    float* part_o;            // [B, H, n_splits, 128] unnormalised partial outputs
    float* part_ml;           // [B, H, n_splits, 2]   running max (log2 domain) and denominator
    int n_splits;
    int nbh;                  // B * H (prefill: 1-D grid of n_qblocks * nbh workgroups)
    int q_pad;                // attn_fwd_w64_kernel: query blocks are aligned to the END of the query range; block 0 starts at row -q_pad
    const uint16_t* vt;       // attn_fwd_w64_kernel: V^T [B][H][128][vt_row] (keys contiguous; written by attn_vt_kernel from v)
    int64_t vt_row;           // keys per row of vt (Tk rounded up to 64)
};

// Workgroup -> (query block, head, batch) for the prefill kernel, 1-D grid.  Blocks are dispatched round-robin
// over the 8 XCDs (block b -> XCD b % 8), each with a private L2.  All query blocks of one (batch, head) re-stream
// the same K/V, so they are pinned to ONE XCD -- (batch, head) pair number p lives on XCD p % 8 -- and issued
// longest-first within it, which keeps the co-resident workgroups walking the same key tiles at the same time.
// (With the plain (qblock, head, batch) grid every XCD fetched every head: L2 hit rate ~40 %, 5.2 GB fetched for
// 268 MB of K/V at T = 16,385; pinned: 89-92 % and 1.9 GB, +5...9 % throughput.)  Placement only affects speed.
__device__ __forceinline__ void attn_block_map(const AttnArgs& a, int& qb, int& head, int& bat) {
    const int bid = blockIdx.x;
    int pair, qi;
    if (a.nbh % 8 == 0) {
        const int xcd = bid & 7, slot = bid >> 3;          // slot-th block of this XCD
        qi = slot % a.n_qblocks;
        pair = (slot / a.n_qblocks) * 8 + xcd;
    } else {
        qi = bid % a.n_qblocks;
        pair = bid / a.n_qblocks;
    }
    qb = a.n_qblocks - 1 - qi;                              // longest (latest) query blocks first
    head = pair % a.H;
    bat = pair / a.H;
}


// csrc/attn_w64.hip: the 4-wave / 64-rows-per-wave prefill kernel (geometry + launch)
int evo_attn_w64_launch(AttnArgs a, int64_t B, void* vt_ws, void* stream);
