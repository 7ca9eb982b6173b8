// Hyena operator for gfx950: short causal FIR (k=3) + bias + column split + x1*v gate + long
// convolution + (y + x1v*D)*x2 epilogue, from the projections output z [B,T,3D] straight to y [B,T,D].
//
// Long convolution: the filter is h_k = Re sum_{s<8} R_s p_s^k, so y = h (*) x1v is evaluated exactly
// through the 8 complex modes   S_t = p S_{t-1} + x1v_t ,  y_t = Re sum_s R_s S_t   (fp32).  Time is cut
// into segments of C steps; one wave owns (batch b, head h, segment k) and its 64 lanes own the head's
// 128 channels two at a time:
//   launch 1  seg_state : segment end state from a zero start (reads the x1,v thirds of z)
//   launch 2  carry_scan: exclusive scan over segments with p^C (fp64 powers) -> state entering each
//   launch 3  apply     : the full recurrence from the entering state + FIR + gates, writes y
// Algorithmic HBM bytes per token per layer: 3D*2 in + D*2 out = 32,768 B (D = 4096); this 3-launch
// form moves 49,152 B (the x1,v thirds are read twice) plus 128*D*8/C bytes of segment states.
//
// Streaming: a head's slice of one z row is ONE contiguous piece (768 B = x2|x1|v, or its 512-B x1|v tail).
// Each wave keeps a private ring of NSLOT chunks (4 rows each) in LDS, filled by asynchronous
// global->LDS DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round trip) five chunks
// ahead of the math and retired with a COUNTED s_waitcnt vmcnt(N): ~15 KiB per wave / ~120 KiB per CU in
// flight.  (The first version prefetched through registers, 3 KiB per wave, and sat at 3.6-3.8 TB/s,
// bound by latency x bytes-in-flight, not by HBM or VALU.)  No barriers: waves never share LDS.
// Entry points and reference citations: include/evo_mi355x.h.
#include "common.h"
#include "../../include/evo_mi355x.h"

#define NS 8            // state_size   [REF evo/configs/evo-1-8k-base_inference.yml:14]
#define HD 128          // channels per head (hidden_size / num_attention_heads)  [REF yml:2,9]
#define CH 4            // rows (time steps) per DMA chunk
// Ring depth in chunks (NSLOT-1 chunks are in flight while one is consumed) and the waves per SIMD the register
// allocation is bounded for, per kernel.  seg_state (164 VGPRs) runs 3 waves per SIMD with a 4-slot ring: -29 % time
// against 2 waves / 6 slots at the same bytes in flight per CU (profiles/r02_hyena_notes.txt); apply needs 236 VGPRs
// and spills at 3.
#ifndef HY_NSLOT_S
#define HY_NSLOT_S 4
#endif
#ifndef HY_OCC_S
#define HY_OCC_S 3
#endif
#ifndef HY_NSLOT_A
#define HY_NSLOT_A 5
#endif
#ifndef HY_OCC_A
#define HY_OCC_A 2
#endif
#ifndef HY_UNROLL_A
#define HY_UNROLL_A 4   // steps of a chunk scheduled together in apply
#endif
#define HY_PRAGMA_(x) _Pragma(#x)
#define HY_PRAGMA_UNROLL(n) HY_PRAGMA_(unroll n)
#ifndef HY_VARIANT
#define HY_VARIANT 1    // 0: round-1 instruction order (dependent pairs back to back); 1: mode-parallel stages
#endif

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

// All per-lane math runs on f32x2 = (channel 2j, channel 2j+1): one v_pk_fma_f32 per pair of FMAs.  The pairs
// are built by hand (hipcc's SLP pass found the same packing but paid ~45 v_mov per step to assemble them).
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t bf2_unpack(uint32_t w) { f32x2_t r = {bf_lo(w), bf_hi(w)}; return r; }

// per-lane constants of one head-slice: the (lo, hi) channel pair of each of the 3 groups
struct FirCoef {
    f32x2_t w[3][3];   // [group][tap]
    f32x2_t b[3];
};

__device__ __forceinline__ void load_fir(FirCoef& fc, const uint16_t* __restrict__ fir_w,
                                         const uint16_t* __restrict__ fir_b, int c_base, int lane, int g_first) {
#pragma unroll
    for (int g = g_first; g < 3; ++g) {
        const int c = c_base + g * HD + 2 * lane;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            f32x2_t v = {bf_to_f(fir_w[c * 3 + k]), bf_to_f(fir_w[(c + 1) * 3 + k])};
            fc.w[g][k] = v;
        }
        f32x2_t bb = {bf_to_f(fir_b[c]), bf_to_f(fir_b[c + 1])};
        fc.b[g] = bb;
    }
}

// raw z dword at (relative) time t for this lane; t < 0 reads the halo (or zero)
__device__ __forceinline__ uint32_t load_hist(const uint32_t* __restrict__ zrow0, const uint32_t* __restrict__ halo,
                                              int64_t t_abs, int64_t rowdw, int col) {
    if (t_abs >= 0) return zrow0[t_abs * rowdw + col];
    if (halo) return halo[(t_abs + 2) * rowdw + col];
    return 0u;
}

// One DMA chunk = CH rows x ROWB bytes, laid out linearly in LDS; instruction i moves LDS bytes
// [i*1024, (i+1)*1024) of the chunk, lane l the 16 bytes at i*1024 + 16*l  ->  (row, column) of the source.
template <int ROWB>
struct ChunkMap {
    static constexpr int CHUNKB = CH * ROWB;
    static constexpr int NDMA = CHUNKB / 1024;
    int row[NDMA];
    int colb[NDMA];
    __device__ __forceinline__ void init(int lane, int head_col_bytes) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int pos = i * 1024 + 16 * lane;
            row[i] = pos / ROWB;
            colb[i] = head_col_bytes + pos % ROWB;
        }
    }
    // rows past the end of the sequence are clamped to the last row (loaded, never used)
    __device__ __forceinline__ void issue(const unsigned char* zb_bytes, int64_t t_first, int64_t t_last_valid,
                                          int64_t rowbytes, unsigned char* lds_chunk) const {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            int64_t t = t_first + row[i];
            t = t < t_last_valid ? t : t_last_valid;
            const unsigned char* src = zb_bytes + t * rowbytes + colb[i];
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(lds_chunk + i * 1024), 16, 0, 0);
        }
    }
};

// Counted retire of the ring.  The VM counter retires in issue order and counts loads AND stores (gfx9 family;
// LLVM's SIInsertWaitcnts models it the same way), so "chunk c has landed" == "at most as many VMEM ops
// outstanding as were issued after chunk c's DMA".  In the steady state that is, per ring stage, NDMA DMA
// instructions plus the y stores of one chunk (apply: 4; seg_state: 0), times NSLOT-1 stages.  Counting the
// stores matters: waiting them out too (vmcnt(15)) stalled every chunk on the ~2 us store latency.
#define HY_WAIT_STATE() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (HY_NSLOT_S - 1)) : "memory")   /* 2 DMA x stages in flight            */
#define HY_WAIT_APPLY() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * (HY_NSLOT_A - 1)) : "memory")   /* (3 DMA + 4 stores) x stages in flight */
static_assert(CH == 4, "HY_WAIT_* immediates assume 4 stores per chunk");
static_assert(7 * (HY_NSLOT_A - 1) <= 63 && HY_NSLOT_S >= 2 && HY_NSLOT_A >= 2, "vmcnt is a 6-bit field");

// ------------------------------------------------------------------------------------------------ launch 1
// MASK: upstream's padding_mask [B, T] (1 = token, 0 = pad) multiplies the FIR output, i.e. x2, x1 and v of a padded
// position are zero: it injects nothing into the modes and its y is zero.  evo never passes one.
template <bool MASK>
__global__ __launch_bounds__(256, HY_OCC_S) void hyena_seg_state_kernel(
    const uint32_t* __restrict__ z, const uint32_t* __restrict__ z_halo, const uint16_t* __restrict__ fir_w,
    const uint16_t* __restrict__ fir_b, const float* __restrict__ poles, float* __restrict__ agg,
    const uint8_t* __restrict__ mask, int B, int64_t T, int D, int H, int C, int n_seg) {
    constexpr int ROWB = 2 * HD * 2;                                   // x1|v = 512 B
    typedef ChunkMap<ROWB> Map;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * HY_NSLOT_S * Map::CHUNKB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;                 // wave-uniform (SGPR)
    const int64_t total = (int64_t)B * n_seg * H;
    if (gw >= total) return;
    const int h = (int)(gw % H);
    const int seg = (int)((gw / H) % n_seg);
    const int b = (int)(gw / ((int64_t)H * n_seg));
    const int64_t rowdw = 3 * (int64_t)D / 2;
    const int64_t rowbytes = rowdw * 4;
    const int64_t t0 = (int64_t)seg * C;
    const int64_t t1 = (t0 + C < T) ? t0 + C : T;
    unsigned char* ring = smem + wave * (HY_NSLOT_S * Map::CHUNKB);

    const uint32_t* zb = z + (int64_t)b * T * rowdw;
    const unsigned char* zbb = (const unsigned char*)zb;
    Map map;
    map.init(lane, (h * 3 * HD + HD) * 2);
    const int nch = (int)((t1 - t0 + CH - 1) / CH);
#pragma unroll
    for (int c = 0; c < HY_NSLOT_S - 1; ++c)
        if (c < nch) map.issue(zbb, t0 + CH * c, T - 1, rowbytes, ring + c * Map::CHUNKB);

    FirCoef fc;
    load_fir(fc, fir_w, fir_b, h * 3 * HD, lane, 1);
    f32x2_t pr[NS], pi[NS], sr[NS], si[NS];
    {
        const float* pp = poles + ((int64_t)(h * HD + 2 * lane)) * NS * 2;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x2_t a = {pp[s * 2], pp[(NS + s) * 2]}, c = {pp[s * 2 + 1], pp[(NS + s) * 2 + 1]}, zz = {0.f, 0.f};
            pr[s] = a; pi[s] = c; sr[s] = zz; si[s] = zz;
        }
    }
    const uint32_t* hb = z_halo ? z_halo + (int64_t)b * 2 * rowdw : nullptr;
    const int col1 = (h * 3 * HD + HD) / 2 + lane;       // x1 third
    const int col2 = (h * 3 * HD + 2 * HD) / 2 + lane;   // v third
    // history z[t-2], z[t-1] for groups x1 (index 0) and v (index 1)
    f32x2_t m2[2], m1[2];
    m2[0] = bf2_unpack(load_hist(zb, hb, t0 - 2, rowdw, col1));
    m2[1] = bf2_unpack(load_hist(zb, hb, t0 - 2, rowdw, col2));
    m1[0] = bf2_unpack(load_hist(zb, hb, t0 - 1, rowdw, col1));
    m1[1] = bf2_unpack(load_hist(zb, hb, t0 - 1, rowdw, col2));

    const uint8_t* mrow = MASK ? mask + (int64_t)b * T : nullptr;
    auto step = [&](uint32_t zx1, uint32_t zv, int64_t t_abs) {
        const f32x2_t c0 = bf2_unpack(zx1), c1 = bf2_unpack(zv);
        const f32x2_t x1c = pk_fma(fc.w[1][2], c0, pk_fma(fc.w[1][1], m1[0], pk_fma(fc.w[1][0], m2[0], fc.b[1])));
        const f32x2_t vc = pk_fma(fc.w[2][2], c1, pk_fma(fc.w[2][1], m1[1], pk_fma(fc.w[2][0], m2[1], fc.b[2])));
        f32x2_t x = x1c * vc;
        if (MASK) x = x * (mrow[t_abs] ? 1.f : 0.f);
#if HY_VARIANT == 0
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f32x2_t nr = pk_fma(pr[s], sr[s], pk_fma(-pi[s], si[s], x));
            const f32x2_t ni = pk_fma(pr[s], si[s], pi[s] * sr[s]);
            sr[s] = nr;
            si[s] = ni;
        }
#else
        // the 8 modes are independent: issue each stage of the complex multiply-add for all of them before the next,
        // so that no packed FMA reads the result of the instruction just ahead of it (the round-1 order had 2/3 of
        // them at dependency distance 2)
        f32x2_t tt[NS], uu[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) tt[s] = pk_fma(-pi[s], si[s], x);
#pragma unroll
        for (int s = 0; s < NS; ++s) uu[s] = pi[s] * sr[s];
#pragma unroll
        for (int s = 0; s < NS; ++s) sr[s] = pk_fma(pr[s], sr[s], tt[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) si[s] = pk_fma(pr[s], si[s], uu[s]);
#endif
        m2[0] = m1[0]; m1[0] = c0; m2[1] = m1[1]; m1[1] = c1;
    };

    // prologue chunks + parameter loads have all landed past this point: the counted waits below only ever
    // reason about VMEM ops issued inside the loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int slot = 0, fill = HY_NSLOT_S - 1;
    for (int c = 0; c < nch; ++c) {
        if (c + HY_NSLOT_S - 1 < nch) {
            map.issue(zbb, t0 + CH * (int64_t)(c + HY_NSLOT_S - 1), T - 1, rowbytes, ring + fill * Map::CHUNKB);
            HY_WAIT_STATE();                               // chunk c has landed; 5 younger chunks stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char* sp = ring + slot * Map::CHUNKB + 4 * lane;
        const int64_t tb = t0 + CH * (int64_t)c;
        if (tb + CH <= t1) {                               // full chunk: one basic block, 4 steps scheduled together
#pragma unroll
            for (int k = 0; k < CH; ++k)
                step(*(const uint32_t*)(sp + k * ROWB), *(const uint32_t*)(sp + k * ROWB + 256), tb + k);
        } else {                                           // ragged end of the sequence
            for (int k = 0; k < CH && tb + k < t1; ++k)
                step(*(const uint32_t*)(sp + k * ROWB), *(const uint32_t*)(sp + k * ROWB + 256), tb + k);
        }
        slot = slot + 1 == HY_NSLOT_S ? 0 : slot + 1;
        fill = fill + 1 == HY_NSLOT_S ? 0 : fill + 1;
    }

    float4* out = (float4*)(agg + ((((int64_t)b * n_seg + seg) * D + h * HD + 2 * lane) * NS) * 2);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int s = 0; s < NS; s += 2)
            out[(e * NS + s) / 2] = make_float4(sr[s][e], si[s][e], sr[s + 1][e], si[s + 1][e]);
}

// ------------------------------------------------------------------------------------------------ launch 2
// one thread per (b, d, s): in-place exclusive scan over segments.  p^C in fp64 (tiny work, keeps the
// carry exact to fp32 over hundreds of segments).
__device__ __forceinline__ void cpow_int(double pr, double pi, int64_t n, double& or_, double& oi) {
    double rr = 1.0, ri = 0.0;
    while (n > 0) {
        if (n & 1) { double t = rr * pr - ri * pi; ri = rr * pi + ri * pr; rr = t; }
        double t = pr * pr - pi * pi; pi = 2.0 * pr * pi; pr = t;
        n >>= 1;
    }
    or_ = rr; oi = ri;
}

#define SCAN_UNR 8
__global__ __launch_bounds__(256) void hyena_carry_scan_kernel(float2* __restrict__ agg, const float2* __restrict__ poles,
                                                               const float2* __restrict__ s0, float2* __restrict__ s_final,
                                                               int B, int64_t T, int D, int C, int n_seg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // over B * D * NS
    const int64_t per_b = (int64_t)D * NS;
    if (i >= (int64_t)B * per_b) return;
    const int b = (int)(i / per_b);
    const int64_t ds = i - (int64_t)b * per_b;
    const float2 p = poles[ds];
    double pcr, pci;
    cpow_int((double)p.x, (double)p.y, C, pcr, pci);
    double rr = 0.0, ri = 0.0;
    if (s0) { float2 v = s0[i]; rr = v.x; ri = v.y; }
    float2* a = agg + (int64_t)b * n_seg * per_b + ds;
    const int n_lead = n_seg - 1;                                  // segments of full length C
    int k = 0;
    for (; k + SCAN_UNR <= n_lead; k += SCAN_UNR) {
        float2 v[SCAN_UNR];
#pragma unroll
        for (int u = 0; u < SCAN_UNR; ++u) v[u] = a[(int64_t)(k + u) * per_b];
#pragma unroll
        for (int u = 0; u < SCAN_UNR; ++u) {
            a[(int64_t)(k + u) * per_b] = make_float2((float)rr, (float)ri);
            double nr = pcr * rr - pci * ri + (double)v[u].x;
            ri = pcr * ri + pci * rr + (double)v[u].y;
            rr = nr;
        }
    }
    for (; k < n_lead; ++k) {
        float2 v = a[(int64_t)k * per_b];
        a[(int64_t)k * per_b] = make_float2((float)rr, (float)ri);
        double nr = pcr * rr - pci * ri + (double)v.x;
        ri = pcr * ri + pci * rr + (double)v.y;
        rr = nr;
    }
    // last (possibly ragged) segment
    {
        float2 v = a[(int64_t)n_lead * per_b];
        a[(int64_t)n_lead * per_b] = make_float2((float)rr, (float)ri);
        if (s_final) {
            int64_t cl = T - (int64_t)n_lead * C;
            double qr, qi;
            cpow_int((double)p.x, (double)p.y, cl, qr, qi);
            double nr = qr * rr - qi * ri + (double)v.x;
            ri = qr * ri + qi * rr + (double)v.y;
            rr = nr;
            s_final[i] = make_float2((float)rr, (float)ri);
        }
    }
}

// sequence-parallel fix-up: `agg` already holds the entering states for a ZERO carry-in; add the
// contribution p^(k*C) * s0 of the state s0 that enters this shard (known only after the all-gather).
__global__ __launch_bounds__(256) void hyena_carry_add_kernel(float2* __restrict__ agg, const float2* __restrict__ poles,
                                                              const float2* __restrict__ s0, int B, int D, int C,
                                                              int n_seg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_b = (int64_t)D * NS;
    if (i >= (int64_t)B * per_b) return;
    const int b = (int)(i / per_b);
    const int64_t ds = i - (int64_t)b * per_b;
    const float2 p = poles[ds];
    double pcr, pci;
    cpow_int((double)p.x, (double)p.y, C, pcr, pci);
    const float2 v0 = s0[i];
    double cr = v0.x, ci = v0.y;
    float2* a = agg + (int64_t)b * n_seg * per_b + ds;
    for (int k = 0; k < n_seg; ++k) {
        float2 v = a[(int64_t)k * per_b];
        a[(int64_t)k * per_b] = make_float2(v.x + (float)cr, v.y + (float)ci);
        double nr = pcr * cr - pci * ci;
        ci = pcr * ci + pci * cr;
        cr = nr;
    }
}

// ------------------------------------------------------------------------------------------------ launch 3
template <bool MASK>
__global__ __launch_bounds__(256, HY_OCC_A) void hyena_apply_kernel(
    const uint32_t* __restrict__ z, const uint32_t* __restrict__ z_halo, const uint16_t* __restrict__ fir_w,
    const uint16_t* __restrict__ fir_b, const float* __restrict__ poles, const float* __restrict__ residues,
    const uint16_t* __restrict__ dskip, const float* __restrict__ agg, uint32_t* __restrict__ y,
    const uint8_t* __restrict__ mask, int B, int64_t T, int D, int H, int C, int n_seg) {
    constexpr int ROWB = 3 * HD * 2;                                   // x2|x1|v = 768 B
    typedef ChunkMap<ROWB> Map;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * HY_NSLOT_A * Map::CHUNKB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;                 // wave-uniform (SGPR)
    const int64_t total = (int64_t)B * n_seg * H;
    if (gw >= total) return;
    const int h = (int)(gw % H);
    const int seg = (int)((gw / H) % n_seg);
    const int b = (int)(gw / ((int64_t)H * n_seg));
    const int64_t rowdw = 3 * (int64_t)D / 2;
    const int64_t rowbytes = rowdw * 4;
    const int64_t t0 = (int64_t)seg * C;
    const int64_t t1 = (t0 + C < T) ? t0 + C : T;
    unsigned char* ring = smem + wave * (HY_NSLOT_A * Map::CHUNKB);

    const uint32_t* zb = z + (int64_t)b * T * rowdw;
    const unsigned char* zbb = (const unsigned char*)zb;
    Map map;
    map.init(lane, (h * 3 * HD) * 2);
    const int nch = (int)((t1 - t0 + CH - 1) / CH);
#pragma unroll
    for (int c = 0; c < HY_NSLOT_A - 1; ++c)
        if (c < nch) map.issue(zbb, t0 + CH * c, T - 1, rowbytes, ring + c * Map::CHUNKB);

    FirCoef fc;
    load_fir(fc, fir_w, fir_b, h * 3 * HD, lane, 0);
    f32x2_t pr[NS], pi[NS], rr[NS], ri[NS], sr[NS], si[NS], dk;
    {
        const int64_t d0 = h * HD + 2 * lane;
        const float* pp = poles + d0 * NS * 2;
        const float* rp = residues + d0 * NS * 2;
        const float4* sp = (const float4*)(agg + ((((int64_t)b * n_seg + seg) * D + d0) * NS) * 2);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x2_t a = {pp[s * 2], pp[(NS + s) * 2]}, c = {pp[s * 2 + 1], pp[(NS + s) * 2 + 1]};
            f32x2_t e = {rp[s * 2], rp[(NS + s) * 2]}, f = {rp[s * 2 + 1], rp[(NS + s) * 2 + 1]};
            pr[s] = a; pi[s] = c; rr[s] = e; ri[s] = f;
        }
#pragma unroll
        for (int s = 0; s < NS; s += 2) {
            const float4 lo = sp[s / 2], hi = sp[(NS + s) / 2];
            f32x2_t a = {lo.x, hi.x}, c = {lo.y, hi.y}, e = {lo.z, hi.z}, f = {lo.w, hi.w};
            sr[s] = a; si[s] = c; sr[s + 1] = e; si[s + 1] = f;
        }
        f32x2_t d2 = {bf_to_f(dskip[d0]), bf_to_f(dskip[d0 + 1])};
        dk = d2;
    }
    const uint32_t* hb = z_halo ? z_halo + (int64_t)b * 2 * rowdw : nullptr;
    const int col0 = (h * 3 * HD) / 2 + lane;     // x2 third; x1 = +HD/2 dwords, v = +HD dwords
    f32x2_t m2[3], m1[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        m2[g] = bf2_unpack(load_hist(zb, hb, t0 - 2, rowdw, col0 + g * (HD / 2)));
        m1[g] = bf2_unpack(load_hist(zb, hb, t0 - 1, rowdw, col0 + g * (HD / 2)));
    }

    const uint8_t* mrow = MASK ? mask + (int64_t)b * T : nullptr;
    auto step = [&](const uint32_t (&zr)[3], int64_t t_abs) -> uint32_t {
        f32x2_t c0[3], f[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            c0[g] = bf2_unpack(zr[g]);
            f[g] = pk_fma(fc.w[g][2], c0[g], pk_fma(fc.w[g][1], m1[g], pk_fma(fc.w[g][0], m2[g], fc.b[g])));
        }
        if (MASK) {                                // padded position: x2 = x1 = v = 0
            const float mk = mrow[t_abs] ? 1.f : 0.f;
            f[0] = f[0] * mk;
            f[1] = f[1] * mk;
        }
        const f32x2_t x = f[1] * f[2];             // x1 * v
#if HY_VARIANT == 0
        f32x2_t acc = {0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f32x2_t nr = pk_fma(pr[s], sr[s], pk_fma(-pi[s], si[s], x));
            const f32x2_t ni = pk_fma(pr[s], si[s], pi[s] * sr[s]);
            sr[s] = nr;
            si[s] = ni;
            acc = pk_fma(rr[s], nr, pk_fma(-ri[s], ni, acc));
        }
#else
        // mode-parallel stages (see seg_state) and FOUR partial sums for y = Re sum_s R_s S_s instead of one
        // 16-deep dependent chain
        f32x2_t tt[NS], uu[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) tt[s] = pk_fma(-pi[s], si[s], x);
#pragma unroll
        for (int s = 0; s < NS; ++s) uu[s] = pi[s] * sr[s];
#pragma unroll
        for (int s = 0; s < NS; ++s) sr[s] = pk_fma(pr[s], sr[s], tt[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) si[s] = pk_fma(pr[s], si[s], uu[s]);
        f32x2_t a4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a4[k] = rr[k] * sr[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) a4[k] = pk_fma(rr[k + 4], sr[k + 4], a4[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) a4[k] = pk_fma(-ri[k], si[k], a4[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) a4[k] = pk_fma(-ri[k + 4], si[k + 4], a4[k]);
        const f32x2_t acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
#endif
        const f32x2_t out = pk_fma(x, dk, acc) * f[0];   // (y + x1v * D) * x2
#pragma unroll
        for (int g = 0; g < 3; ++g) { m2[g] = m1[g]; m1[g] = c0[g]; }
        return pack_bf2(out[0], out[1]);
    };

    uint32_t* yp = y + ((int64_t)b * T + t0) * (D / 2) + h * (HD / 2) + lane;
    const int ydw = D / 2;
    // prologue chunks + parameter loads have all landed past this point: the counted waits below only ever
    // reason about VMEM ops issued inside the loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int slot = 0, fill = HY_NSLOT_A - 1;
    for (int c = 0; c < nch; ++c) {
        if (c + HY_NSLOT_A - 1 < nch) {
            map.issue(zbb, t0 + CH * (int64_t)(c + HY_NSLOT_A - 1), T - 1, rowbytes, ring + fill * Map::CHUNKB);
            HY_WAIT_APPLY();                               // chunk c has landed; 5 younger chunks stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char* sp = ring + slot * Map::CHUNKB + 4 * lane;
        const int64_t tb = t0 + CH * (int64_t)c;
        if (tb + CH <= t1) {                               // full chunk: one basic block, 4 steps scheduled together
HY_PRAGMA_UNROLL(HY_UNROLL_A)
            for (int k = 0; k < CH; ++k) {
                uint32_t zr[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) zr[g] = *(const uint32_t*)(sp + k * ROWB + g * 256);
                yp[(int64_t)k * ydw] = step(zr, tb + k);
            }
        } else {                                           // ragged end of the sequence
            for (int k = 0; k < CH && tb + k < t1; ++k) {
                uint32_t zr[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) zr[g] = *(const uint32_t*)(sp + k * ROWB + g * 256);
                yp[(int64_t)k * ydw] = step(zr, tb + k);
            }
        }
        yp += (int64_t)CH * ydw;
        slot = slot + 1 == HY_NSLOT_A ? 0 : slot + 1;
        fill = fill + 1 == HY_NSLOT_A ? 0 : fill + 1;
    }
}

// ------------------------------------------------------------------------------------------------ decode step
// one thread per (b, channel pair): FIR step with the 2-sample history, roll, modal update, gate.
__global__ __launch_bounds__(256) void hyena_step_kernel(
    const uint32_t* __restrict__ z_t, uint16_t* __restrict__ fir_state, float* __restrict__ iir_state,
    const uint16_t* __restrict__ fir_w, const uint16_t* __restrict__ fir_b, const float* __restrict__ poles,
    const float* __restrict__ residues, const uint16_t* __restrict__ dskip, uint32_t* __restrict__ y, int B, int D,
    int H) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over B * D/2
    if (i >= (int64_t)B * (D / 2)) return;
    const int b = (int)(i / (D / 2));
    const int dp = (int)(i - (int64_t)b * (D / 2));                  // channel pair index within D
    const int h = dp / (HD / 2), j2 = dp - h * (HD / 2);
    float f[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int cdw = (h * 3 * HD + g * HD) / 2 + j2;              // dword column in the 3D row
        const uint32_t zr = z_t[(int64_t)b * (3 * D / 2) + cdw];
        const float zc[2] = {bf_lo(zr), bf_hi(zr)};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * cdw + e;
            uint16_t* fs = fir_state + ((int64_t)b * 3 * D + c) * 2;
            const float o0 = bf_to_f(fs[0]), o1 = bf_to_f(fs[1]);
            const uint16_t zraw = (uint16_t)(e ? (zr >> 16) : (zr & 0xffffu));
            f[g][e] = fmaf(bf_to_f(fir_w[c * 3 + 2]), zc[e],
                           fmaf(bf_to_f(fir_w[c * 3 + 1]), o1, fmaf(bf_to_f(fir_w[c * 3]), o0, bf_to_f(fir_b[c]))));
            fs[0] = fs[1];
            fs[1] = zraw;
        }
    }
    float out[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int d = 2 * dp + e;
        const float x = f[1][e] * f[2][e];
        float2* st = (float2*)iir_state + ((int64_t)b * D + d) * NS;
        const float2* pp = (const float2*)poles + (int64_t)d * NS;
        const float2* rp = (const float2*)residues + (int64_t)d * NS;
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float2 p = pp[s], r = rp[s], sv = st[s];
            const float nr = fmaf(p.x, sv.x, fmaf(-p.y, sv.y, x));
            const float ni = fmaf(p.x, sv.y, p.y * sv.x);
            st[s] = make_float2(nr, ni);
            acc = fmaf(r.x, nr, fmaf(-r.y, ni, acc));
        }
        out[e] = fmaf(x, bf_to_f(dskip[d]), acc) * f[0][e];
    }
    y[i] = pack_bf2(out[0], out[1]);
}

// ------------------------------------------------------------------------------------------------ C ABI
static int hyena_check(int64_t B, int64_t T, int64_t D, int64_t n_heads, int64_t seg_len) {
    if (B <= 0 || T <= 0 || D <= 0 || n_heads <= 0) return -1;
    if (D != n_heads * HD) return -1;                 // one wave per 128-channel head
    if (seg_len <= 0 || seg_len % CH != 0) return -1;
    return 0;
}

extern "C" int evo_hyena_seg_state(const void* z, const void* z_halo, const void* fir_w, const void* fir_b,
                                   const float* poles, float* agg, const uint8_t* mask, int64_t B, int64_t T, int64_t D,
                                   int64_t n_heads, int64_t seg_len, void* stream) {
    if (hyena_check(B, T, D, n_heads, seg_len)) return -1;
    const int n_seg = (int)((T + seg_len - 1) / seg_len);
    const int64_t waves = B * n_seg * n_heads;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (mask)
        hipLaunchKernelGGL(hyena_seg_state_kernel<true>, grid, block, 0, (hipStream_t)stream, (const uint32_t*)z,
                           (const uint32_t*)z_halo, (const uint16_t*)fir_w, (const uint16_t*)fir_b, poles, agg, mask, (int)B,
                           T, (int)D, (int)n_heads, (int)seg_len, n_seg);
    else
        hipLaunchKernelGGL(hyena_seg_state_kernel<false>, grid, block, 0, (hipStream_t)stream, (const uint32_t*)z,
                           (const uint32_t*)z_halo, (const uint16_t*)fir_w, (const uint16_t*)fir_b, poles, agg, mask, (int)B,
                           T, (int)D, (int)n_heads, (int)seg_len, n_seg);
    return evo_launch_status();
}

extern "C" int evo_hyena_carry_scan(float* agg, const float* poles, const float* s0, float* s_final, int64_t B,
                                    int64_t T, int64_t D, int64_t seg_len, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || seg_len <= 0) return -1;
    const int n_seg = (int)((T + seg_len - 1) / seg_len);
    const int64_t n = B * D * NS;
    hipLaunchKernelGGL(hyena_carry_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float2*)agg, (const float2*)poles, (const float2*)s0, (float2*)s_final, (int)B, T, (int)D,
                       (int)seg_len, n_seg);
    return evo_launch_status();
}

extern "C" int evo_hyena_carry_add(float* agg, const float* poles, const float* s0, int64_t B, int64_t T, int64_t D,
                                   int64_t seg_len, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || seg_len <= 0 || !s0) return -1;
    const int n_seg = (int)((T + seg_len - 1) / seg_len);
    const int64_t n = B * D * NS;
    hipLaunchKernelGGL(hyena_carry_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float2*)agg, (const float2*)poles, (const float2*)s0, (int)B, (int)D, (int)seg_len, n_seg);
    return evo_launch_status();
}

extern "C" int evo_hyena_apply(const void* z, const void* z_halo, const void* fir_w, const void* fir_b,
                               const float* poles, const float* residues, const void* dskip, const float* agg, void* y,
                               const uint8_t* mask, int64_t B, int64_t T, int64_t D, int64_t n_heads, int64_t seg_len,
                               void* stream) {
    if (hyena_check(B, T, D, n_heads, seg_len)) return -1;
    const int n_seg = (int)((T + seg_len - 1) / seg_len);
    const int64_t waves = B * n_seg * n_heads;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (mask)
        hipLaunchKernelGGL(hyena_apply_kernel<true>, grid, block, 0, (hipStream_t)stream, (const uint32_t*)z,
                           (const uint32_t*)z_halo, (const uint16_t*)fir_w, (const uint16_t*)fir_b, poles, residues,
                           (const uint16_t*)dskip, agg, (uint32_t*)y, mask, (int)B, T, (int)D, (int)n_heads, (int)seg_len,
                           n_seg);
    else
        hipLaunchKernelGGL(hyena_apply_kernel<false>, grid, block, 0, (hipStream_t)stream, (const uint32_t*)z,
                           (const uint32_t*)z_halo, (const uint16_t*)fir_w, (const uint16_t*)fir_b, poles, residues,
                           (const uint16_t*)dskip, agg, (uint32_t*)y, mask, (int)B, T, (int)D, (int)n_heads, (int)seg_len,
                           n_seg);
    return evo_launch_status();
}

extern "C" int evo_hyena_step(const void* z_t, void* fir_state, float* iir_state, const void* fir_w, const void* fir_b,
                              const float* poles, const float* residues, const void* dskip, void* y, int64_t B,
                              int64_t D, int64_t n_heads, void* stream) {
    if (B <= 0 || D != n_heads * HD) return -1;
    const int64_t n = B * (D / 2);
    hipLaunchKernelGGL(hyena_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)z_t, (uint16_t*)fir_state, iir_state, (const uint16_t*)fir_w,
                       (const uint16_t*)fir_b, poles, residues, (const uint16_t*)dskip, (uint32_t*)y, (int)B, (int)D,
                       (int)n_heads);
    return evo_launch_status();
}
