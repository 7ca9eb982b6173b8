// Skinny dense layer for decoding: y[M,N] = x[M,K] . W[N,K]^T (+ bias[N]) (+ residual[M,N]), M <= 16.
//
// Two kernels.  gemv_kernel (M <= 4, or any K % 8 == 0 shape): dot2 on the VALU, below.  skinny_mfma_kernel
// (5 <= M <= 16, K % 32 == 0): the batch rows become the N side of v_mfma_f32_16x16x32_bf16 -- with 5+ rows the VALU
// form spends more on re-reading x from L2 (M loads per R weight loads) than on the weights; here one 1-KiB x
// fragment meets one 1-KiB weight fragment per MFMA and the math is free.
//
// At batch 1-8 a decode step streams all 12.9 GB of weights once and is bound by HBM, not by MFMA: through
// hipBLASLt's tile GEMM it ran at ~2.5 TB/s (5.2 ms/token).  This kernel is the weight-streaming form: one wave
// owns R consecutive output rows, its lanes stride over K with 16-byte non-temporal loads straight into
// VGPRs (no LDS round trip -- the operand is streamed once and shared with nobody), accumulates with
// v_dot2c_f32_bf16 (two bf16 MACs per lane-op, fp32 accumulate, no unpack), and reduces across the wave at
// the end.  x (<= 8 x K bf16) is re-read by every wave and lives in L2.
// Entry point and reference citation: include/evo_mi355x.h.
#include "common.h"
#include "../../include/evo_mi355x.h"

typedef __bf16 dot_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot8(const uint4& a, const uint4& b, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dot_bf16x2, a.x), __builtin_bit_cast(dot_bf16x2, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dot_bf16x2, a.y), __builtin_bit_cast(dot_bf16x2, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dot_bf16x2, a.z), __builtin_bit_cast(dot_bf16x2, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dot_bf16x2, a.w), __builtin_bit_cast(dot_bf16x2, b.w), acc, false);
    return acc;
}

__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// SPLIT: the four waves of a workgroup share the SAME R rows and each takes every fourth 64-vector slice of K (the
// partial sums meet in LDS).  Used for the narrow layers (N <= 4096: out_filter_dense, out_proj, l3), where one wave
// per R rows leaves the chip with too few loads in flight: 3.3 -> see tools/bench_gemv.py.
template <int M, int R, bool SPLIT>
__global__ __launch_bounds__(256) void gemv_kernel(const uint4* __restrict__ x, const uint4* __restrict__ w,
                                                   const uint16_t* __restrict__ bias, const uint16_t* res,
                                                   uint16_t* y, int N, int nvec) {   // res may alias y
    __shared__ float part[SPLIT ? 4 * R * M : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n0 = SPLIT ? (int64_t)blockIdx.x * R : ((int64_t)blockIdx.x * 4 + wave) * R;
    if (n0 >= N) return;                                     // (workgroup-uniform when SPLIT)
    const uint4* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t n = n0 + r < N ? n0 + r : N - 1;            // rows past the end are computed on a clamp, not stored
        wrow[r] = w + n * nvec;
    }
    // the storing lane requests its bias / residual values NOW: fetched after the reduction they were a dependent ~1 us tail on
    // a launch whose whole stream is 5-17 us (out_filter_dense, out_proj, l3)
    const bool ep_lane = lane == 0 && (!SPLIT || wave == 0);
    uint16_t pf_b[R], pf_r[R][M];
    if (ep_lane) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t n = n0 + r < N ? n0 + r : N - 1;
            pf_b[r] = bias ? bias[n] : (uint16_t)0;
#pragma unroll
            for (int m = 0; m < M; ++m) pf_r[r][m] = res ? res[(int64_t)m * N + n] : (uint16_t)0;
        }
    }
    float acc[R][M];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = 0.f;

    constexpr int ST = SPLIT ? 256 : 64;                     // slice stride of this wave
    int v = SPLIT ? wave * 64 + lane : lane;
    for (; v + ST < nvec; v += 2 * ST) {                     // two k-slices per trip: 2R weight loads in flight
        uint4 w0[R], w1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { w0[r] = ld_stream(wrow[r] + v); w1[r] = ld_stream(wrow[r] + v + ST); }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint4 x0 = x[(int64_t)m * nvec + v], x1 = x[(int64_t)m * nvec + v + ST];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][m] = dot8(w1[r], x1, dot8(w0[r], x0, acc[r][m]));
        }
    }
    for (; v < nvec; v += ST) {
        uint4 w0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) w0[r] = ld_stream(wrow[r] + v);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint4 x0 = x[(int64_t)m * nvec + v];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][m] = dot8(w0[r], x0, acc[r][m]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);
    if (SPLIT) {
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int m = 0; m < M; ++m) part[(wave * R + r) * M + m] = acc[r][m];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m)
                acc[r][m] = part[r * M + m] + part[(R + r) * M + m] + part[(2 * R + r) * M + m] + part[(3 * R + r) * M + m];
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t n = n0 + r;
            if (n < N) {
                const float b = bias ? bf_to_f(pf_b[r]) : 0.f;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    float o = acc[r][m] + b;
                    if (res) o += bf_to_f(pf_r[r][m]);
                    y[(int64_t)m * N + n] = f_to_bf(o);
                }
            }
        }
    }
}

// STAGE: wave m of the workgroup normalises batch row m (and m + 4 when M > 4) ONCE into LDS and every wave's trips read it from there.
// Two things were wrong with every wave rebuilding the normalised slices itself: the sum of squares was a loop of one 16-byte
// load + s_waitcnt vmcnt(0) per 64-vector slice -- eight dependent L2 round trips per row, queued BEHIND the weight prefetch
// (vmcnt retires in order), ~6 us of every wave's life at M = 1 -- and at M >= 2 the per-trip re-normalisation made the
// launches VALU-bound (fused Hyena step 22 / 26 / 43 us at M = 1 / 2 / 4 against 20-22 us for the plain GEMV of the matrix).
// Here the staging wave requests its row's 2 x 8 vectors at once, AHEAD of its weight prefetch, reduces in rmsnorm_kernel's
// order (lane-strided sequential fp32, wave butterfly) and stores the normalised row: bit-identical values, one L2 round trip.
#define GV_WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")   // orders LDS only: weight loads stay in flight
extern __shared__ uint4 gv_xn[];                                                          // [M][nvec] normalised rows (STAGE)

__device__ __forceinline__ uint4 gv_norm8(const uint4& xv, const uint4& sv, float inv) {
    uint4 o;
    o.x = pack_bf2(bf_lo(sv.x) * (bf_lo(xv.x) * inv), bf_hi(sv.x) * (bf_hi(xv.x) * inv));
    o.y = pack_bf2(bf_lo(sv.y) * (bf_lo(xv.y) * inv), bf_hi(sv.y) * (bf_hi(xv.y) * inv));
    o.z = pack_bf2(bf_lo(sv.z) * (bf_lo(xv.z) * inv), bf_hi(sv.z) * (bf_hi(xv.z) * inv));
    o.w = pack_bf2(bf_lo(sv.w) * (bf_lo(xv.w) * inv), bf_hi(sv.w) * (bf_hi(xv.w) * inv));
    return o;
}
__device__ __forceinline__ float gv_sumsq8(const uint4& xv, float ss) {
    const float f[8] = {bf_lo(xv.x), bf_hi(xv.x), bf_lo(xv.y), bf_hi(xv.y), bf_lo(xv.z), bf_hi(xv.z), bf_lo(xv.w), bf_hi(xv.w)};
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
    return ss;
}
// rows of up to 512 vectors (the host stages K = 4096 only; other widths take the unstaged kernels): request everything (slices
// past the row end re-read the lane's first vector, unused)
__device__ __forceinline__ void gv_stage_load(const uint4* xr, const uint4* scale, int nvec, int lane, uint4 (&sx)[8], uint4 (&sc)[8]) {
    const int v0 = lane < nvec ? lane : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int v = lane + 64 * i, vc = v < nvec ? v : v0;
        sx[i] = xr[vc];
        sc[i] = scale[vc];
    }
}
__device__ __forceinline__ void gv_stage_finish(uint4* dst, int nvec, int lane, float eps, float inv_sqrt_d, const uint4 (&sx)[8],
                                                const uint4 (&sc)[8]) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < nvec) ss = gv_sumsq8(sx[i], ss);
    ss = wave_sum(ss);
    const float inv = 1.0f / (sqrtf(ss) * inv_sqrt_d + eps);
    uint4 px[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                            // (opaque copy: otherwise the 64 unpacked fp32 values of the reduction are
        px[i] = sx[i];                                       //  kept alive across it for the normalisation, +64 VGPRs on every wave)
        asm volatile("" : "+v"(px[i].x), "+v"(px[i].y), "+v"(px[i].z), "+v"(px[i].w));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < nvec) dst[lane + 64 * i] = gv_norm8(px[i], sc[i], inv);
}
// K = 4096 (nvec = 512: every decode layer behind a norm) written out as four trips with a two-deep register pipeline: the
// generic loop below ends each trip by copying the next trip's registers over the current ones, which makes the compiler wait
// for the loads it has just issued -- one trip in flight per wave.  Same operations in the same order.
#define GV_TRIP_LOAD(BUF, T)                                                                                          \
    _Pragma("unroll") for (int r = 0; r < GV_R; ++r) {                                                                \
        BUF[r] = ld_stream(wrow[r] + lane + 128 * (T));                                                               \
        BUF[GV_R + r] = ld_stream(wrow[r] + lane + 128 * (T) + 64);                                                   \
    }
#define GV_TRIP_DOT(BUF, T, XF)                                                                                       \
    _Pragma("unroll") for (int m = 0; m < M; ++m) {                                                                   \
        const uint4 x0 = XF(m, lane + 128 * (T)), x1 = XF(m, lane + 128 * (T) + 64);                                  \
        _Pragma("unroll") for (int r = 0; r < GV_R; ++r) acc[r][m] = dot8(BUF[GV_R + r], x1, dot8(BUF[r], x0, acc[r][m])); \
    }
#define GV_FOUR_TRIPS(XF)                                                                                             \
    {                                                                                                                 \
        uint4 wb[2 * GV_R];                                                                                           \
        GV_TRIP_LOAD(wb, 1)                                                                                           \
        GV_TRIP_DOT(wa, 0, XF)                                                                                        \
        __builtin_amdgcn_sched_barrier(0); /* (trips stay in program order: requests retire in issue order) */        \
        GV_TRIP_LOAD(wa, 2)                                                                                           \
        GV_TRIP_DOT(wb, 1, XF)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        GV_TRIP_LOAD(wb, 3)                                                                                           \
        GV_TRIP_DOT(wa, 2, XF)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        GV_TRIP_DOT(wb, 3, XF)                                                                                        \
    }

// ---- RMSNorm folded into the layer that consumes it (decode form): y = bf16(scale * x / (rms(x) + eps)) . W^T + bias.
// Every wave first reduces sum(x^2) with the rmsnorm kernel's own lane -> element mapping and summation order (so the
// normalised row is bit-identical to what that kernel would have stored), then streams its R rows of W while it
// rebuilds the normalised x slice by slice in registers -- one launch instead of two per block.  That is the STAGE = false
// form (any K); at K = 4096 the launches take STAGE = true (above): rows normalised once into LDS, four pipelined trips.
template <int M, int R, bool STAGE>
__global__ __launch_bounds__(256) void gemv_norm_kernel(const uint4* __restrict__ x, const uint4* __restrict__ scale,
                                                        const uint4* __restrict__ w, const uint16_t* __restrict__ bias,
                                                        uint16_t* __restrict__ y, int N, int nvec, float eps, float inv_sqrt_d) {
    constexpr int GV_R = R;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * R;
    if (!STAGE && n0 >= N) return;                           // (STAGE: every wave helps to stage x, then leaves)
    const uint4* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t n = n0 + r < N ? n0 + r : N - 1;
        wrow[r] = w + n * nvec;
    }
    // (STAGE) the staging wave's x / scale requests go out first, so they come back first
    uint4 st_x[8], st_s[8];
    const bool stager = STAGE && wave < M;                   // (M > 4: waves 0-3 take rows 4-7 in a second round)
    if (stager) gv_stage_load(x + (int64_t)wave * nvec, scale, nvec, lane, st_x, st_s);
    __builtin_amdgcn_sched_barrier(0);                       // (requests retire in issue order: keep these ahead of the weights)
    // the first trip's weights are requested BEFORE the norm pass (which only touches x, in L2): the HBM stream starts at
    // once instead of after a ~1.5 us reduction, and every later trip's loads are issued ahead of the trip that consumes
    // the previous ones (same arithmetic in the same order: results are bit-identical to the single-buffered loop)
    uint4 wa[2 * GV_R];
    const bool gv_any = lane + 64 < nvec;
    if (gv_any) {
#pragma unroll
        for (int r = 0; r < GV_R; ++r) { wa[r] = ld_stream(wrow[r] + lane); wa[GV_R + r] = ld_stream(wrow[r] + lane + 64); }
    }
    float inv[M];
#pragma unroll
    for (int m = 0; m < M; ++m) inv[m] = 0.f;
    if (!STAGE) {                                            // rows too long for LDS: every wave reduces every row itself
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float ss = 0.f;
            for (int v = lane; v < nvec; v += 64) ss = gv_sumsq8(x[(int64_t)m * nvec + v], ss);
            ss = wave_sum(ss);
            inv[m] = 1.0f / (sqrtf(ss) * inv_sqrt_d + eps);
        }
    }
    auto normed_calc = [&](int m, int v) {
        const uint4 xv = x[(int64_t)m * nvec + v], sv = scale[v];
        uint4 o;
        o.x = pack_bf2(bf_lo(sv.x) * (bf_lo(xv.x) * inv[m]), bf_hi(sv.x) * (bf_hi(xv.x) * inv[m]));
        o.y = pack_bf2(bf_lo(sv.y) * (bf_lo(xv.y) * inv[m]), bf_hi(sv.y) * (bf_hi(xv.y) * inv[m]));
        o.z = pack_bf2(bf_lo(sv.z) * (bf_lo(xv.z) * inv[m]), bf_hi(sv.z) * (bf_hi(xv.z) * inv[m]));
        o.w = pack_bf2(bf_lo(sv.w) * (bf_lo(xv.w) * inv[m]), bf_hi(sv.w) * (bf_hi(xv.w) * inv[m]));
        return o;
    };
    if (STAGE) {
        if (stager) gv_stage_finish(gv_xn + wave * nvec, nvec, lane, eps, inv_sqrt_d, st_x, st_s);
        if (M > 4) {                                         // batch rows 4-7: a second round for the same waves
            for (int r = wave + 4; r < M; r += 4) {
                gv_stage_load(x + (int64_t)r * nvec, scale, nvec, lane, st_x, st_s);
                gv_stage_finish(gv_xn + r * nvec, nvec, lane, eps, inv_sqrt_d, st_x, st_s);
            }
        }
        GV_WG_BARRIER();
        if (n0 >= N) return;
        __builtin_amdgcn_sched_barrier(0);                   // (the second trip's requests stay below the staging: its registers are free now)
    }
    auto normed = [&](int m, int v) { return STAGE ? gv_xn[m * nvec + v] : normed_calc(m, v); };
    float acc[R][M];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
    if constexpr (STAGE) {                                   // (STAGE launches have nvec == 512: the host checks)
        GV_FOUR_TRIPS(normed)
    } else {
        int v = lane;
        for (; v + 64 < nvec; v += 128) {
            uint4 wb[2 * GV_R];
            const bool more = v + 128 + 64 < nvec;
            if (more) {
#pragma unroll
                for (int r = 0; r < GV_R; ++r) { wb[r] = ld_stream(wrow[r] + v + 128); wb[GV_R + r] = ld_stream(wrow[r] + v + 192); }
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const uint4 x0 = normed(m, v), x1 = normed(m, v + 64);
#pragma unroll
                for (int r = 0; r < GV_R; ++r) acc[r][m] = dot8(wa[GV_R + r], x1, dot8(wa[r], x0, acc[r][m]));
            }
            if (more) {
#pragma unroll
                for (int r = 0; r < 2 * GV_R; ++r) wa[r] = wb[r];
            }
        }
        for (; v < nvec; v += 64) {
            uint4 w0[R];
#pragma unroll
            for (int r = 0; r < R; ++r) w0[r] = ld_stream(wrow[r] + v);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const uint4 x0 = normed(m, v);
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r][m] = dot8(w0[r], x0, acc[r][m]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t n = n0 + r;
            if (n < N) {
                const float b = bias ? bf_to_f(bias[n]) : 0.f;
#pragma unroll
                for (int m = 0; m < M; ++m) y[(int64_t)m * N + n] = f_to_bf(acc[r][m] + b);
            }
        }
    }
}

// ---- the whole Hyena mixer input of a decode step in one launch: pre-norm + projection + FIR/modal step + gate.
// A wave owns CPW (one or two) adjacent channels of one head = 3 CPW rows of the projection weight (x2, x1, v thirds), streams
// them like gemv_norm_kernel, and then its first CPW * M lanes (one per channel x batch row) run evo_hyena_step's arithmetic
// on the dot products -- same operations in the same order, so outputs and carried states are bit-identical to
// evo_norm_linear_small_m_bf16 followed by evo_hyena_step.
template <int M, int CPW, bool STAGE>
__global__ __launch_bounds__(256) void gemv_norm_hyena_kernel(
    const uint4* __restrict__ x, const uint4* __restrict__ scale, const uint4* __restrict__ w,
    const uint16_t* __restrict__ bias, uint16_t* __restrict__ fir_state, float* __restrict__ iir_state,
    const uint16_t* __restrict__ fir_w, const uint16_t* __restrict__ fir_b, const float* __restrict__ poles,
    const float* __restrict__ residues, const uint16_t* __restrict__ dskip, uint16_t* __restrict__ y, int D, int nvec,
    float eps, float inv_sqrt_d) {
    constexpr int HDc = 128, NSc = 8, GV_R = 3 * CPW;      // CPW channels per wave (1 or 2)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * 4 + wave;                     // channel (pair) index over D / CPW
    if (!STAGE && unit >= D / CPW) return;
    const int unit_c = unit < D / CPW ? unit : D / CPW - 1;      // (STAGE: a wave past the end stages x with the others, then leaves)
    const int h = unit_c / (HDc / CPW), j0 = CPW * (unit_c - h * (HDc / CPW));
    const uint4* wrow[GV_R];
#pragma unroll
    for (int r = 0; r < GV_R; ++r) wrow[r] = w + (int64_t)(h * 3 * HDc + (r / CPW) * HDc + j0 + (r % CPW)) * nvec;
    // (STAGE) the staging wave's x / scale requests go out first, so they come back first
    uint4 st_x[8], st_s[8];
    const bool stager = STAGE && wave < M;                   // (M > 4: waves 0-3 take rows 4-7 in a second round)
    if (stager) gv_stage_load(x + (int64_t)wave * nvec, scale, nvec, lane, st_x, st_s);
    __builtin_amdgcn_sched_barrier(0);                       // (requests retire in issue order: keep these ahead of the weights)
    // the first trip's weights are requested BEFORE the norm pass (which only touches x, in L2): the HBM stream starts at
    // once instead of after a ~1.5 us reduction, and every later trip's loads are issued ahead of the trip that consumes
    // the previous ones (same arithmetic in the same order: results are bit-identical to the single-buffered loop)
    uint4 wa[2 * GV_R];
    const bool gv_any = lane + 64 < nvec;
    if (gv_any) {
#pragma unroll
        for (int r = 0; r < GV_R; ++r) { wa[r] = ld_stream(wrow[r] + lane); wa[GV_R + r] = ld_stream(wrow[r] + lane + 64); }
    }
    float inv[M];
#pragma unroll
    for (int m = 0; m < M; ++m) inv[m] = 0.f;
    if (!STAGE) {                                            // rows too long for LDS: every wave reduces every row itself
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float ss = 0.f;
            for (int v = lane; v < nvec; v += 64) ss = gv_sumsq8(x[(int64_t)m * nvec + v], ss);
            ss = wave_sum(ss);
            inv[m] = 1.0f / (sqrtf(ss) * inv_sqrt_d + eps);
        }
    }
    auto normed_calc = [&](int m, int v) {
        const uint4 xv = x[(int64_t)m * nvec + v], sv = scale[v];
        uint4 o;
        o.x = pack_bf2(bf_lo(sv.x) * (bf_lo(xv.x) * inv[m]), bf_hi(sv.x) * (bf_hi(xv.x) * inv[m]));
        o.y = pack_bf2(bf_lo(sv.y) * (bf_lo(xv.y) * inv[m]), bf_hi(sv.y) * (bf_hi(xv.y) * inv[m]));
        o.z = pack_bf2(bf_lo(sv.z) * (bf_lo(xv.z) * inv[m]), bf_hi(sv.z) * (bf_hi(xv.z) * inv[m]));
        o.w = pack_bf2(bf_lo(sv.w) * (bf_lo(xv.w) * inv[m]), bf_hi(sv.w) * (bf_hi(xv.w) * inv[m]));
        return o;
    };
    if (STAGE) {
        if (stager) gv_stage_finish(gv_xn + wave * nvec, nvec, lane, eps, inv_sqrt_d, st_x, st_s);
        if (M > 4) {                                         // batch rows 4-7: a second round for the same waves
            for (int r = wave + 4; r < M; r += 4) {
                gv_stage_load(x + (int64_t)r * nvec, scale, nvec, lane, st_x, st_s);
                gv_stage_finish(gv_xn + r * nvec, nvec, lane, eps, inv_sqrt_d, st_x, st_s);
            }
        }
        GV_WG_BARRIER();
        if (unit >= D / CPW) return;
        __builtin_amdgcn_sched_barrier(0);                   // (the second trip's requests stay below the staging: its registers are free now)
    }
    // the epilogue's operands (this lane's channel x batch row: FIR taps / state / bias of x2, x1, v, the 8 poles, residues and
    // modal states, D) are requested NOW (after the staging, whose registers they reuse), behind the first weight trip: they do not depend on the dot products, and fetched
    // at the end they were a ~2 us dependent-latency tail per wave with nothing left to overlap it
    const bool ep_lane = lane < CPW * M;
    const int ep_e = lane % CPW, ep_m = ep_lane ? lane / CPW : 0;
    const int ep_dch = h * HDc + j0 + ep_e;
    uint16_t pf_bias[3], pf_fw[3][3], pf_fb[3], pf_fs[3][2], pf_dk = 0;
    float2 pf_p[NSc], pf_r[NSc], pf_s[NSc];
    if (ep_lane) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int c = h * 3 * HDc + g * HDc + j0 + ep_e;
            pf_bias[g] = bias[c];
            pf_fw[g][0] = fir_w[c * 3]; pf_fw[g][1] = fir_w[c * 3 + 1]; pf_fw[g][2] = fir_w[c * 3 + 2];
            pf_fb[g] = fir_b[c];
            const uint16_t* fs = fir_state + ((int64_t)ep_m * 3 * D + c) * 2;
            pf_fs[g][0] = fs[0]; pf_fs[g][1] = fs[1];
        }
        const float2* st = (const float2*)iir_state + ((int64_t)ep_m * D + ep_dch) * NSc;
        const float2* pp = (const float2*)poles + (int64_t)ep_dch * NSc;
        const float2* rp = (const float2*)residues + (int64_t)ep_dch * NSc;
#pragma unroll
        for (int sI = 0; sI < NSc; ++sI) { pf_p[sI] = pp[sI]; pf_r[sI] = rp[sI]; pf_s[sI] = st[sI]; }
        pf_dk = dskip[ep_dch];
    }
    auto normed = [&](int m, int v) { return STAGE ? gv_xn[m * nvec + v] : normed_calc(m, v); };
    float acc[GV_R][M];
#pragma unroll
    for (int r = 0; r < GV_R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
    if constexpr (STAGE) {                                   // (STAGE launches have nvec == 512: the host checks)
        GV_FOUR_TRIPS(normed)
    } else {
        int v = lane;
        for (; v + 64 < nvec; v += 128) {
            uint4 wb[2 * GV_R];
            const bool more = v + 128 + 64 < nvec;
            if (more) {
#pragma unroll
                for (int r = 0; r < GV_R; ++r) { wb[r] = ld_stream(wrow[r] + v + 128); wb[GV_R + r] = ld_stream(wrow[r] + v + 192); }
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const uint4 x0 = normed(m, v), x1 = normed(m, v + 64);
#pragma unroll
                for (int r = 0; r < GV_R; ++r) acc[r][m] = dot8(wa[GV_R + r], x1, dot8(wa[r], x0, acc[r][m]));
            }
            if (more) {
#pragma unroll
                for (int r = 0; r < 2 * GV_R; ++r) wa[r] = wb[r];
            }
        }
        for (; v < nvec; v += 64) {
            uint4 w0[GV_R];
#pragma unroll
            for (int r = 0; r < GV_R; ++r) w0[r] = ld_stream(wrow[r] + v);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const uint4 x0 = normed(m, v);
#pragma unroll
                for (int r = 0; r < GV_R; ++r) acc[r][m] = dot8(w0[r], x0, acc[r][m]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < GV_R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);
    if (!ep_lane) return;
    const int e = ep_e, m = ep_m;                                // this lane: channel j0 + e of batch row m
    float f[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        float d = 0.f;
#pragma unroll
        for (int mm = 0; mm < M; ++mm)
#pragma unroll
            for (int ee = 0; ee < CPW; ++ee) d = (m == mm && e == ee) ? acc[CPW * g + ee][mm] : d;
        const int c = h * 3 * HDc + g * HDc + j0 + e;            // channel in the 3D row
        const uint16_t zraw = f_to_bf(d + bf_to_f(pf_bias[g]));  // what the unfused projection stores
        uint16_t* fs = fir_state + ((int64_t)m * 3 * D + c) * 2;
        const float o0 = bf_to_f(pf_fs[g][0]), o1 = bf_to_f(pf_fs[g][1]);
        f[g] = fmaf(bf_to_f(pf_fw[g][2]), bf_to_f(zraw),
                    fmaf(bf_to_f(pf_fw[g][1]), o1, fmaf(bf_to_f(pf_fw[g][0]), o0, bf_to_f(pf_fb[g]))));
        fs[0] = pf_fs[g][1];
        fs[1] = zraw;
    }
    const int dch = ep_dch;
    const float xv = f[1] * f[2];
    float2* st = (float2*)iir_state + ((int64_t)m * D + dch) * NSc;
    float accy = 0.f;
#pragma unroll
    for (int s = 0; s < NSc; ++s) {
        const float2 p = pf_p[s], r = pf_r[s], sv = pf_s[s];
        const float nr = fmaf(p.x, sv.x, fmaf(-p.y, sv.y, xv));
        const float ni = fmaf(p.x, sv.y, p.y * sv.x);
        st[s] = make_float2(nr, ni);
        accy = fmaf(r.x, nr, fmaf(-r.y, ni, accy));
    }
    y[(int64_t)m * D + dch] = f_to_bf(fmaf(xv, bf_to_f(pf_dk), accy) * f[0]);
}

// ---- gated MLP input, decode form: a[m][n] = gelu(x_m . W1_n) * (x_m . W2_n) with W12 = [W1; W2] ([2I, K]).  Same
// streaming loop as gemv_kernel; a wave owns 2 output columns = rows (n, n+1) of W1 and (I+n, I+n+1) of W2, rounds both
// dot products to bf16 (what the unfused GEMM stores) and applies the gate -- one launch instead of two per block.
// NORM: x is the un-normalised residual row and `scale` the RMSNorm weight (same fold as gemv_norm_kernel).
template <int M, bool NORM, bool STAGE>
__global__ __launch_bounds__(256) void gemv_gate_kernel(const uint4* __restrict__ x, const uint4* __restrict__ scale,
                                                        const uint4* __restrict__ w, uint16_t* __restrict__ a, int I,
                                                        int nvec, float eps, float inv_sqrt_d, int grouped) {
    constexpr int GV_R = 4;                                  // rows per wave: (n, n+1) of W1 and of W2
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * 2;
    if (!STAGE && n0 >= I) return;
    const uint4* wrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int64_t n = n0 + (r & 1) < I ? n0 + (r & 1) : I - 1;
        // plain [W1; W2]: row (r >> 1) I + n.  grouped (HipOps.pack_gate_weights: blocks of [32 rows of W1 | the same 32 rows of W2], the
        // order the gated MFMA launch reads -- ONE copy of l1 | l2 serves prefill and decode): row 64 (n / 32) + 32 (r >> 1) + n % 32
        const int64_t row = grouped ? 64 * (n >> 5) + 32 * (r >> 1) + (n & 31) : (r >> 1) * (int64_t)I + n;
        wrow[r] = w + row * nvec;
    }
    // (STAGE) the staging wave's x / scale requests go out first, so they come back first
    uint4 st_x[8], st_s[8];
    const bool stager = STAGE && wave < M;                   // (M > 4: waves 0-3 take rows 4-7 in a second round)
    if (stager) gv_stage_load(x + (int64_t)wave * nvec, scale, nvec, lane, st_x, st_s);
    __builtin_amdgcn_sched_barrier(0);                       // (requests retire in issue order: keep these ahead of the weights)
    // first trip's weights before the norm pass, every later trip one ahead (see gemv_norm_kernel)
    uint4 wa[8];
    if (lane + 64 < nvec) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { wa[r] = ld_stream(wrow[r] + lane); wa[4 + r] = ld_stream(wrow[r] + lane + 64); }
    }
    float inv[M];
#pragma unroll
    for (int m = 0; m < M; ++m) inv[m] = 0.f;
    if (NORM && !STAGE) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float ss = 0.f;
            for (int v = lane; v < nvec; v += 64) ss = gv_sumsq8(x[(int64_t)m * nvec + v], ss);
            ss = wave_sum(ss);
            inv[m] = 1.0f / (sqrtf(ss) * inv_sqrt_d + eps);
        }
    }
    auto xin_calc = [&](int m, int v) {
        const uint4 xv = x[(int64_t)m * nvec + v];
        if (!NORM) return xv;
        const uint4 sv = scale[v];
        uint4 o;
        o.x = pack_bf2(bf_lo(sv.x) * (bf_lo(xv.x) * inv[m]), bf_hi(sv.x) * (bf_hi(xv.x) * inv[m]));
        o.y = pack_bf2(bf_lo(sv.y) * (bf_lo(xv.y) * inv[m]), bf_hi(sv.y) * (bf_hi(xv.y) * inv[m]));
        o.z = pack_bf2(bf_lo(sv.z) * (bf_lo(xv.z) * inv[m]), bf_hi(sv.z) * (bf_hi(xv.z) * inv[m]));
        o.w = pack_bf2(bf_lo(sv.w) * (bf_lo(xv.w) * inv[m]), bf_hi(sv.w) * (bf_hi(xv.w) * inv[m]));
        return o;
    };
    if (STAGE) {
        if (stager) gv_stage_finish(gv_xn + wave * nvec, nvec, lane, eps, inv_sqrt_d, st_x, st_s);
        if (M > 4) {                                         // batch rows 4-7: a second round for the same waves
            for (int r = wave + 4; r < M; r += 4) {
                gv_stage_load(x + (int64_t)r * nvec, scale, nvec, lane, st_x, st_s);
                gv_stage_finish(gv_xn + r * nvec, nvec, lane, eps, inv_sqrt_d, st_x, st_s);
            }
        }
        GV_WG_BARRIER();
        if (n0 >= I) return;
        __builtin_amdgcn_sched_barrier(0);                   // (the second trip's requests stay below the staging: its registers are free now)
    }
    auto xin = [&](int m, int v) { return STAGE ? gv_xn[m * nvec + v] : xin_calc(m, v); };
    float acc[4][M];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
    if constexpr (STAGE) {                                   // (STAGE launches have nvec == 512: the host checks)
        GV_FOUR_TRIPS(xin)
    } else {
        int v = lane;
        for (; v + 64 < nvec; v += 128) {
            uint4 wb[8];
            const bool more = v + 128 + 64 < nvec;
            if (more) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { wb[r] = ld_stream(wrow[r] + v + 128); wb[4 + r] = ld_stream(wrow[r] + v + 192); }
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const uint4 x0 = xin(m, v), x1 = xin(m, v + 64);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r][m] = dot8(wa[4 + r], x1, dot8(wa[r], x0, acc[r][m]));
            }
            if (more) {
#pragma unroll
                for (int r = 0; r < 8; ++r) wa[r] = wb[r];
            }
        }
        for (; v < nvec; v += 64) {
            uint4 w0[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w0[r] = ld_stream(wrow[r] + v);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const uint4 x0 = xin(m, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r][m] = dot8(w0[r], x0, acc[r][m]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const f32x2_t u = {round_bf(acc[0][m]), round_bf(acc[1][m])}, g = {round_bf(acc[2][m]), round_bf(acc[3][m])};
            const f32x2_t o = gelu_gate2(u, g);
            if (n0 + 1 < I) *(uint32_t*)(a + (int64_t)m * I + n0) = pack_bf2(o[0], o[1]);
            else a[(int64_t)m * I + n0] = f_to_bf(o[0]);
        }
    }
}

// ---- 5 <= M <= 16: MFMA form.  Workgroup = 8 waves = 16 output rows; wave w streams its eighth of K:
//   A operand = W rows   (lane: row n = lane & 15, k-block = lane >> 4 -> 16 B = 8 bf16, non-temporal, straight to VGPRs)
//   B operand = x rows   (lane: row m = lane & 15, same k-block; rows >= M re-read row M-1 and are never stored)
//   D[n][m]: lane holds n = 4 (lane >> 4) + r, m = lane & 15.
// The eight partial tiles meet in LDS; wave 0 adds bias / residual and stores (one rounding).
typedef float gv_f32x4 __attribute__((ext_vector_type(4)));
// MT (round 6): m tiles of 16 batch rows per weight pass -- 17 <= M <= 64 rows (the pooled decode step at 17-64 live slots; small prefill
// batches) stream the weights ONCE with MT MFMAs per weight fragment.  Until round 5 those batches ran the persistent 256 x 256 GEMM:
// N / 256 of the 256 CUs busy, 0.5-1.3 TB/s of weights (profiles/r06_pool32_kernel_stats.txt: 80 % of a 32-slot pooled step).
// NT (round 6): n tiles of 16 weight rows per workgroup sharing every x fragment -- at 17-64 batch rows a weight fragment (1 KiB per wave) pulled
// MT x fragments of the same size through L2 / L1 (2.1 TB/s of weights at 32 rows, 1.5 at 64: tools/bench_gemv.py); two n tiles halve that.
template <int MT, int NT>
__global__ __launch_bounds__(512) void skinny_mfma_kernel(const uint4* __restrict__ x, const uint4* __restrict__ w,
                                                          const uint16_t* __restrict__ bias, const uint16_t* res,
                                                          uint16_t* y, int M, int N, int nvec) {   // res may alias y
    static_assert(MT * NT <= 8, "one reducing wave per (n tile, m tile)");
    __shared__ gv_f32x4 part[NT * MT][8][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kb = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    const int nsteps = nvec >> 2;                                // 32 k per MFMA = 4 vectors of 8
    const int s0 = (int)((int64_t)nsteps * wave / 8), s1 = (int)((int64_t)nsteps * (wave + 1) / 8);
    const uint4* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nrow = n0 + 16 * j + r16 < N ? n0 + 16 * j + r16 : N - 1;
        wp[j] = w + (int64_t)nrow * nvec + kb;
    }
    const uint4* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int mrow = 16 * t + r16 < M ? 16 * t + r16 : M - 1;
        xp[t] = x + (int64_t)mrow * nvec + kb;
    }
    gv_f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[j][t] = gv_f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = NT * MT == 1 ? 8 : (NT + MT <= 4 ? 4 : 2);     // steps per trip: U * (NT weight + MT x) fragments requested together
    int s = s0;
    for (; s + U <= s1; s += U) {
        uint4 wf[NT][U], xf[MT][U];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int u = 0; u < U; ++u) wf[j][u] = ld_stream(wp[j] + 4 * (s + u));
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < U; ++u) xf[t][u] = xp[t][4 * (s + u)];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j][u]), __builtin_bit_cast(bf16x8_t, xf[t][u]),
                                                                        acc[j][t], 0, 0, 0);
    }
    for (; s < s1; ++s) {
        uint4 xv[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) xv[t] = xp[t][4 * s];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const uint4 wv = ld_stream(wp[j] + 4 * s);
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv), __builtin_bit_cast(bf16x8_t, xv[t]), acc[j][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t) part[j * MT + t][wave][lane] = acc[j][t];
    __syncthreads();
    // the eight partial tiles of (n tile j, m tile t) meet in wave j MT + t: bias / residual added in fp32, one rounding
    if (wave < NT * MT) {
        gv_f32x4 a_ = part[wave][0][lane];
#pragma unroll
        for (int q = 1; q < 8; ++q) a_ += part[wave][q][lane];
        const int jt = wave / MT, tt = wave - jt * MT;
        const int m = 16 * tt + r16;
        if (m < M) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * jt + 4 * kb + r;
                if (n < N) {
                    float o = a_[r] + (bias ? bf_to_f(bias[n]) : 0.f);
                    if (res) o += bf_to_f(res[(int64_t)m * N + n]);
                    y[(int64_t)m * N + n] = f_to_bf(o);
                }
            }
        }
    }
}

// ---- 5 <= M <= 64, N >= 8192: n-split form (round 6).  The k-split form above is bound by L2, not by HBM: every weight fragment (1 KiB per
// wave) pulls MT x fragments of the same size through L2 -- weights + x add up to 6-7 TB/s at every M (4.0 TB/s of weights at 8 rows, 3.4 at
// 16, 2.2 at 32, 1.5 at 64; tools/bench_gemv.py) where hipBLASLt streams 4.2 TB/s at 32-64 rows.  Here a workgroup is WAVES waves, each owning a
// 16-row n tile for the WHOLE K (no partial sums, no reduction), and the x rows of a 256-k chunk are staged in LDS once per workgroup and
// read by all its waves as B fragments (ds_read_b128, conflict-free at a row pitch of 528 B): x costs L2 MT / WAVES of the weight bytes
// instead of MT.  Weight fragments of chunk c + 1 and the x rows of chunk c + 1 are requested before chunk c is multiplied.  K % 256 == 0.
template <int MT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void skinny_n_kernel(const uint4* __restrict__ x, const uint4* __restrict__ w,
                                                              const uint16_t* __restrict__ bias, const uint16_t* res,
                                                              uint16_t* y, int M, int N, int nvec) {   // res may alias y
    constexpr int KC = 8, ROWB = KC * 64 + 16;                   // steps per chunk; LDS bytes per x row (512 + 16 pad)
    constexpr int XL = (MT * 16 * KC * 4 + 64 * WAVES - 1) / (64 * WAVES);   // 16-byte pieces of an x chunk per thread
    __shared__ __attribute__((aligned(16))) unsigned char xs[2][MT * 16 * ROWB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kb = lane >> 4;
    const int n0 = (blockIdx.x * WAVES + wave) * 16;
    const int nrow = n0 + r16 < N ? n0 + r16 : N - 1;
    const uint4* wp = w + (int64_t)nrow * nvec + kb;
    const int n_chunk = nvec / (KC * 4);
    gv_f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = gv_f32x4{0.f, 0.f, 0.f, 0.f};
    // x chunk pieces of this thread: piece p = tid + i * threads -> row p / 32 (32 pieces of 16 B per 512-byte row), column p % 32
    // (native vectors, not HIP's uint4 struct: as a loop-carried, conditionally written array of uint4 the pieces stayed in scratch memory)
    typedef unsigned int sn_u32x4 __attribute__((ext_vector_type(4)));
    sn_u32x4 xr[XL];
#define SN_X_LOAD(C)                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < XL; ++i_) {                                                       \
        const int pidx_ = tid + i_ * 64 * WAVES;                                                              \
        int row_ = pidx_ >> 5;                                                                                \
        row_ = row_ < M ? row_ : M - 1;              /* rows past M (and pieces past the chunk) re-read a valid row: never stored */ \
        xr[i_] = *(const sn_u32x4*)(x + (int64_t)row_ * nvec + (C) * (KC * 4) + (pidx_ & 31));               \
    }
#define SN_X_STORE(BUF)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < XL; ++i_) {                                                       \
        const int pidx_ = tid + i_ * 64 * WAVES;                                                              \
        if (pidx_ < MT * 16 * 32) *(sn_u32x4*)(xs[BUF] + (pidx_ >> 5) * ROWB + (pidx_ & 31) * 16) = xr[i_];   \
    }
#ifndef SN_PROBE_CONTIG
#define SN_PROBE_CONTIG 0                    // measurement only (WRONG results): weight fragments read as if stored fragment-major, 1 KiB contiguous per request
#endif
    auto w_load = [&](int c, uint4 (&wf)[KC]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < KC; ++u)
            wf[u] = SN_PROBE_CONTIG ? ld_stream(w + ((int64_t)(n0 >> 4) * (nvec >> 2) + c * KC + u) * 64 + lane) : ld_stream(wp + 4 * (c * KC + u));
    };
    auto mul = [&](int buf, const uint4 (&wf)[KC]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < KC; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const uint4 xv = *(const uint4*)(xs[buf] + (16 * t + r16) * ROWB + u * 64 + kb * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[u]), __builtin_bit_cast(bf16x8_t, xv), acc[t], 0, 0, 0);
            }
    };
    // weight fragments: chunk c multiplies while c + 1 is in flight (a ring of three -- 16 KiB per wave in flight -- measured 5-8 % SLOWER: the form is
    // not latency-bound; what caps it is the access pattern, 64-byte pieces of 16 rows per request: read as if the weight were stored fragment-major
    // (1 KiB contiguous per request, SN_PROBE_CONTIG) the same kernel streams 4.3-4.4 TB/s at 17-32 rows, hipBLASLt's rate, instead of 3.1-3.3)
    uint4 wA[KC], wB[KC];
    SN_X_LOAD(0);
    w_load(0, wA);
    SN_X_STORE(0);
    __syncthreads();
#define SN_STAGE(C, WCUR, WNEXT, BUF)                                                                         \
    if ((C) < n_chunk) {                                                                                      \
        if ((C) + 1 < n_chunk) { SN_X_LOAD((C) + 1); w_load((C) + 1, WNEXT); }                                \
        mul(BUF, WCUR);                                                                                       \
        if ((C) + 1 < n_chunk) { SN_X_STORE((BUF) ^ 1); }                                                     \
        __syncthreads();                                                                                      \
    }
    for (int c = 0; c < n_chunk; c += 2) {                       // two chunks per trip: the fragment sets alternate without register copies
        SN_STAGE(c, wA, wB, 0) SN_STAGE(c + 1, wB, wA, 1)
    }
#undef SN_STAGE
#undef SN_X_LOAD
#undef SN_X_STORE
    // D[n][m]: lane holds n = n0 + 4 kb + r, m = 16 t + r16 -- four consecutive output columns of its row
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = 16 * t + r16;
        if (m < M) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 4 * kb + r;
                if (n < N) {
                    float o = acc[t][r] + (bias ? bf_to_f(bias[n]) : 0.f);
                    if (res) o += bf_to_f(res[(int64_t)m * N + n]);
                    y[(int64_t)m * N + n] = f_to_bf(o);
                }
            }
        }
    }
}

// ---- the n-split form with CONTIGUOUS weight requests (round 6, second step).  skinny_n_kernel asks for a weight fragment as the MFMA wants it: 64 bytes
// of each of 16 rows that lie K * 2 bytes apart -- and streams 3.1-3.3 TB/s where the same kernel reading 1 KiB contiguous per request runs 4.3-4.4
// (profiles/r06_small_m_nsplit_ab.txt).  Here a request covers RPI = 2 (4) whole row pieces of 512 (256) contiguous bytes, the wave parks the chunk in a
// PRIVATE LDS region in row order and reads its fragments back from there (ds_write_b128 / ds_read_b128, both conflict-free at the padded row pitch);
// the requests of chunk c + 1 are in flight while chunk c multiplies.  Everything else as skinny_n_kernel.
// SPLITK (N < 8192, the two N = 4,096 layers of a block): blockIdx.y is a slice of the 256-k chunks; the slice's fp32 partial sums go to
// ws [slices][M][N] and skinny_reduce_kernel adds them in slice order (+ bias, + residual, one rounding) -- 64 n groups x 4 slices = a workgroup
// per CU with x still shared four ways, where the k-split form is L2-bound on x (2.1 TB/s at 32 rows).
// GATE (WAVES = 4): the gated MLP's first half [REF stripedhyena/layers.py ParallelGatedMLP: gelu(l1 x) * l2 x] in one launch -- a workgroup's 64 weight rows are
// 32 rows of W1 (waves 0, 1) and the MATCHING 32 rows of W2 (waves 2, 3): one block of the grouped layout (`gate_layout` 1: HipOps.pack_gate_weights), or rows
// c .. c + 31 and I + c .. of the plain [W1; W2] (2).  Waves 2, 3 hand their z2 tiles over through LDS; z1, z2 are rounded to bf16 (the dense layer's outputs),
// the gate is evaluated in fp32 and rounded once: bit for bit the dense layer + evo_gelu_gate_bf16.  N = 2 I, y = a [M, I].
template <int MT, int WAVES, bool SPLITK = false, bool GATE = false>
__global__ __launch_bounds__(64 * WAVES) void skinny_nw_kernel(const uint4* __restrict__ x, const uint4* __restrict__ w,
                                                               const uint16_t* __restrict__ bias, const uint16_t* res,
                                                               uint16_t* y, int M, int N, int nvec, float* __restrict__ ws = nullptr,
                                                               int gate_layout = 0) {   // res may alias y
    static_assert(!GATE || (WAVES == 4 && !SPLITK), "a gated workgroup is one 32 + 32 row block");
    constexpr int KC = MT <= 2 ? 8 : 4;                          // MFMA steps (32 k) per chunk
    constexpr int ROWB = KC * 64 + 16;                           // LDS bytes per row of a chunk (+ 16 pad: b128 accesses at this pitch touch all banks)
    constexpr int PPR = KC * 4;                                  // 16-byte pieces per row of a chunk
    constexpr int RPI = 64 / PPR;                                // weight rows per request
    constexpr int XL = (MT * 16 * PPR + 64 * WAVES - 1) / (64 * WAVES);   // x pieces per thread and chunk
    __shared__ __attribute__((aligned(16))) unsigned char xs[2][MT * 16 * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char wsm[WAVES][16 * ROWB];
    typedef unsigned int sn_u32x4 __attribute__((ext_vector_type(4)));   // (native vectors: loop-carried uint4 arrays end up in scratch memory)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kb = lane >> 4;
    const int n0 = GATE && gate_layout == 2 ? (wave < 2 ? blockIdx.x * 32 + wave * 16 : (N >> 1) + blockIdx.x * 32 + (wave - 2) * 16)
                                            : (blockIdx.x * WAVES + wave) * 16;
    const int n_chunk_all = nvec / PPR;
    const int c_first = SPLITK ? (int)((int64_t)n_chunk_all * blockIdx.y / gridDim.y) : 0;
    const int n_chunk = SPLITK ? (int)((int64_t)n_chunk_all * (blockIdx.y + 1) / gridDim.y) : n_chunk_all;   // chunks [c_first, n_chunk) are this workgroup's
    // request i of a chunk: weight row n0 + RPI i + lane / PPR, piece lane % PPR
    const int q_row = lane / PPR, q_piece = lane % PPR;
    const sn_u32x4* wq[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        const int row = n0 + RPI * i + q_row < N ? n0 + RPI * i + q_row : N - 1;
        wq[i] = (const sn_u32x4*)(w + (int64_t)row * nvec + q_piece);
    }
    unsigned char* wmine = wsm[wave];
    const int w_st = q_row * ROWB + q_piece * 16;               // + RPI * i * ROWB
    const int f_rd = r16 * ROWB + kb * 16;                       // + u * 64
    gv_f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = gv_f32x4{0.f, 0.f, 0.f, 0.f};
    sn_u32x4 xr[XL], wr[KC];
#define SW_X_LOAD(C)                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < XL; ++i_) {                                                       \
        const int pidx_ = tid + i_ * 64 * WAVES;                                                              \
        int row_ = pidx_ / PPR;                                                                               \
        row_ = row_ < M ? row_ : M - 1;              /* rows past M (and pieces past the chunk) re-read a valid row: never stored */ \
        xr[i_] = *(const sn_u32x4*)(x + (int64_t)row_ * nvec + (C) * PPR + (pidx_ % PPR));                    \
    }
#define SW_X_STORE(BUF)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < XL; ++i_) {                                                       \
        const int pidx_ = tid + i_ * 64 * WAVES;                                                              \
        if (pidx_ < MT * 16 * PPR) *(sn_u32x4*)(xs[BUF] + (pidx_ / PPR) * ROWB + (pidx_ % PPR) * 16) = xr[i_]; \
    }
#define SW_W_LOAD(C) _Pragma("unroll") for (int i_ = 0; i_ < KC; ++i_) wr[i_] = __builtin_nontemporal_load(wq[i_] + (C) * PPR);
#define SW_W_STORE() _Pragma("unroll") for (int i_ = 0; i_ < KC; ++i_) *(sn_u32x4*)(wmine + w_st + RPI * i_ * ROWB) = wr[i_];
    auto mul = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < KC; ++u) {
            const uint4 wv = *(const uint4*)(wmine + f_rd + u * 64);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const uint4 xv = *(const uint4*)(xs[buf] + (16 * t + r16) * ROWB + u * 64 + kb * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv), __builtin_bit_cast(bf16x8_t, xv), acc[t], 0, 0, 0);
            }
        }
    };
    SW_X_LOAD(c_first);
    SW_W_LOAD(c_first);
    SW_X_STORE(c_first & 1);
    SW_W_STORE();
    __syncthreads();
    for (int c = c_first; c < n_chunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunk) { SW_X_LOAD(c + 1); SW_W_LOAD(c + 1); }
        mul(buf);
        // no barrier here: x buffer buf ^ 1 has been free since the last barrier, and the weight region is the wave's own -- its stores follow its reads
        // in program order (same LDS array: the compiler keeps the order, the LDS unit executes a wave's accesses in order)
#ifdef SW_TWO_BARRIERS
        __syncthreads();
#endif
        if (c + 1 < n_chunk) { SW_X_STORE(buf ^ 1); SW_W_STORE(); }
        __syncthreads();
    }
#undef SW_X_LOAD
#undef SW_X_STORE
#undef SW_W_LOAD
#undef SW_W_STORE
    if constexpr (GATE) {
        // (the last barrier of the chunk loop is behind every wave's last fragment read: xs is free)
        float* ex = (float*)&xs[0][0];                           // [2 waves][MT][64 lanes] x 16 B <= 4 KiB
        if (wave >= 2) {
#pragma unroll
            for (int t = 0; t < MT; ++t) *(gv_f32x4*)(ex + (((wave - 2) * MT + t) * 64 + lane) * 4) = acc[t];
        }
        __syncthreads();
        if (wave < 2) {
            const int I_ = N >> 1;
            const int col0 = blockIdx.x * 32 + wave * 16 + 4 * kb;           // this lane's four gated columns
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const gv_f32x4 z2 = *(const gv_f32x4*)(ex + ((wave * MT + t) * 64 + lane) * 4);
                const int m = 16 * t + r16;
                if (m < M && col0 < I_) {
                    const f32x2_t ua = {round_bf(acc[t][0]), round_bf(acc[t][1])}, ga = {round_bf(z2[0]), round_bf(z2[1])};
                    const f32x2_t ub = {round_bf(acc[t][2]), round_bf(acc[t][3])}, gb = {round_bf(z2[2]), round_bf(z2[3])};
                    const f32x2_t oa = gelu_gate2(ua, ga), ob = gelu_gate2(ub, gb);
                    uint2 o;
                    o.x = pack_bf2(oa[0], oa[1]);
                    o.y = pack_bf2(ob[0], ob[1]);
                    *(uint2*)(y + (int64_t)m * I_ + col0) = o;               // (I % 32 == 0: whole groups of four)
                }
            }
        }
        return;
    }
    if constexpr (SPLITK) {
        float* wsl = ws + (int64_t)blockIdx.y * M * N;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = 16 * t + r16;
            if (m < M && n0 + 4 * kb < N) *(gv_f32x4*)(wsl + (int64_t)m * N + n0 + 4 * kb) = acc[t];     // (N % 4 == 0 on this path)
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = 16 * t + r16;
        if (m < M) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 4 * kb + r;
                if (n < N) {
                    float o = acc[t][r] + (bias ? bf_to_f(bias[n]) : 0.f);
                    if (res) o += bf_to_f(res[(int64_t)m * N + n]);
                    y[(int64_t)m * N + n] = f_to_bf(o);
                }
            }
        }
    }
}

// y [M, N] = sum over slices of ws [slices][M][N] (in slice order: deterministic) + bias + residual, one rounding; a thread owns four columns
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float* __restrict__ ws, const uint16_t* __restrict__ bias, const uint16_t* res,
                                                            uint16_t* y, int M, int N, int slices) {   // res may alias y
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)M * N / 4;
    if (q >= total) return;
    const int64_t e = q * 4;
    const int n = (int)(e % N);
    gv_f32x4 a = *(const gv_f32x4*)(ws + e);
    for (int sl = 1; sl < slices; ++sl) a += *(const gv_f32x4*)(ws + (int64_t)sl * M * N + e);
    float o[4] = {a[0], a[1], a[2], a[3]};
    if (bias) { const uint2 b = *(const uint2*)(bias + n); o[0] += bf_lo(b.x); o[1] += bf_hi(b.x); o[2] += bf_lo(b.y); o[3] += bf_hi(b.y); }
    if (res) { const uint2 r = *(const uint2*)(res + e); o[0] += bf_lo(r.x); o[1] += bf_hi(r.x); o[2] += bf_lo(r.y); o[3] += bf_hi(r.y); }
    uint2 out;
    out.x = pack_bf2(o[0], o[1]);
    out.y = pack_bf2(o[2], o[3]);
    *(uint2*)(y + e) = out;
}

template <int M>
static void gemv_launch(const void* x, const void* w, const void* bias, const void* res, void* y, int64_t N, int64_t K,
                        hipStream_t s) {
    // rows per wave.  Measured (tools/bench_gemv.py, MI355X): R=4 wins for every M (R=8 starves the chip of
    // waves at N=4096; R=2 doubles the L2 re-reads of x): 4.9-5.4 TB/s at M=1-2, 3.0-3.6 TB/s at M=8.
#ifndef GEMV_R_A
#define GEMV_R_A 4
#endif
#ifndef GEMV_R_B
#define GEMV_R_B 4
#endif
#ifndef GEMV_R_C
#define GEMV_R_C 4
#endif
    constexpr int R = M <= 2 ? GEMV_R_A : (M <= 4 ? GEMV_R_B : GEMV_R_C);
    const int64_t waves = (N + R - 1) / R;
    if (N <= 4096 && K >= 2048)
        hipLaunchKernelGGL((gemv_kernel<M, R, true>), dim3((unsigned)waves), dim3(256), 0, s, (const uint4*)x,
                           (const uint4*)w, (const uint16_t*)bias, (const uint16_t*)res, (uint16_t*)y, (int)N, (int)(K / 8));
    else
        hipLaunchKernelGGL((gemv_kernel<M, R, false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, (const uint4*)x,
                           (const uint4*)w, (const uint16_t*)bias, (const uint16_t*)res, (uint16_t*)y, (int)N, (int)(K / 8));
}

extern "C" int evo_norm_linear_small_m_bf16(const void* x, const void* scale, const void* w, const void* bias, void* y,
                                            int64_t M, int64_t N, int64_t K, float eps, void* stream) {
    if (M < 1 || M > 8 || N <= 0 || K <= 0 || K % 8 != 0 || N > 0x7fffffff - 16) return -1;
    if (M > 4 && K != 4096) return -1;                       // batches of 5-8 rows exist in the LDS-staged form only
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(((N + 3) / 4 + 3) / 4)), block(256);
    const float isd = 1.0f / sqrtf((float)K);
    const size_t lds = (size_t)M * K * 2;                   // normalised rows staged in LDS (when they fit)
    const bool stage = K == 4096;                            // the staged kernels are written out for four 1024-element trips
#define EVO_NL(MM)                                                                                            \
    if (stage)                                                                                                \
        hipLaunchKernelGGL((gemv_norm_kernel<MM, 4, true>), grid, block, lds, s, (const uint4*)x, (const uint4*)scale, \
                           (const uint4*)w, (const uint16_t*)bias, (uint16_t*)y, (int)N, (int)(K / 8), eps, isd); \
    else                                                                                                      \
        hipLaunchKernelGGL((gemv_norm_kernel<MM, 4, false>), grid, block, 0, s, (const uint4*)x, (const uint4*)scale, \
                           (const uint4*)w, (const uint16_t*)bias, (uint16_t*)y, (int)N, (int)(K / 8), eps, isd)
#define EVO_NLS(MM)                                                                                           \
    hipLaunchKernelGGL((gemv_norm_kernel<MM, 4, true>), grid, block, lds, s, (const uint4*)x, (const uint4*)scale, \
                       (const uint4*)w, (const uint16_t*)bias, (uint16_t*)y, (int)N, (int)(K / 8), eps, isd)
    switch (M) {
        case 1: EVO_NL(1); break;
        case 2: EVO_NL(2); break;
        case 3: EVO_NL(3); break;
        case 4: EVO_NL(4); break;
        case 5: EVO_NLS(5); break;
        case 6: EVO_NLS(6); break;
        case 7: EVO_NLS(7); break;
        default: EVO_NLS(8); break;
    }
#undef EVO_NLS
#undef EVO_NL
    return evo_launch_status();
}

extern "C" int evo_hyena_decode_fused_small_m(const void* x, const void* norm_scale, const void* proj_w, const void* proj_b,
                                              void* fir_state, float* iir_state, const void* fir_w, const void* fir_b,
                                              const float* poles, const float* residues, const void* dskip, void* y,
                                              int64_t M, int64_t D, int64_t n_heads, float eps, void* stream) {
    if (M < 1 || M > 8 || D <= 0 || D != n_heads * 128 || D % 8 != 0) return -1;
    if (M > 4 && D != 4096) return -1;                       // batches of 5-8 rows exist in the LDS-staged form only
    hipStream_t s = (hipStream_t)stream;
    // channels per wave, measured under a hipGraph on MI355X (tools/experiments/hyena_decode_cpw_bench.py): one channel per wave
    // (twice the waves, half the bytes in flight each) is 5 % faster at M = 1 and 6 % at M = 4, two channels win at M = 2
    // (22.8 / 26.1 / 42.6 us with two, 21.6 / 29.9 / 40.0 us with one, M = 1 / 2 / 4).  Same arithmetic either way.
#ifndef GEMV_HYENA_CPW_STAGED
#define GEMV_HYENA_CPW_STAGED 1
#endif
#ifndef GEMV_HYENA_CPW_STAGED1
#define GEMV_HYENA_CPW_STAGED1 1
#endif
    const float isd = 1.0f / sqrtf((float)D);
    const size_t lds = (size_t)M * D * 2;
#define EVO_HD(MM, CPW, ST)                                                                                   \
    hipLaunchKernelGGL((gemv_norm_hyena_kernel<MM, CPW, ST>), dim3((unsigned)((D / CPW + 3) / 4)), dim3(256), ST ? lds : 0, s, \
                       (const uint4*)x, (const uint4*)norm_scale,                                             \
                       (const uint4*)proj_w, (const uint16_t*)proj_b, (uint16_t*)fir_state, iir_state,          \
                       (const uint16_t*)fir_w, (const uint16_t*)fir_b, poles, residues, (const uint16_t*)dskip, \
                       (uint16_t*)y, (int)D, (int)(D / 8), eps, isd)
    const bool stage = D == 4096;
    switch (M) {
        case 1: if (stage) EVO_HD(1, GEMV_HYENA_CPW_STAGED1, true); else EVO_HD(1, 1, false); break;
        case 2: if (stage) EVO_HD(2, GEMV_HYENA_CPW_STAGED, true); else EVO_HD(2, 2, false); break;
        case 3: if (stage) EVO_HD(3, GEMV_HYENA_CPW_STAGED, true); else EVO_HD(3, 2, false); break;
        case 4: if (stage) EVO_HD(4, GEMV_HYENA_CPW_STAGED, true); else EVO_HD(4, 1, false); break;
        case 5: EVO_HD(5, 1, true); break;
        case 6: EVO_HD(6, 1, true); break;
        case 7: EVO_HD(7, 1, true); break;
        default: EVO_HD(8, 1, true); break;
    }
#undef EVO_HD
    return evo_launch_status();
}

static int mlp_gate_launch(const void* x, const void* scale, const void* w12, void* a, int64_t M, int64_t I, int64_t K,
                           float eps, int64_t grouped, hipStream_t s) {
    if (M < 1 || M > 8 || I <= 0 || I % 2 != 0 || K <= 0 || K % 8 != 0 || I > 0x3fffffff) return -1;
    if (grouped && I % 32 != 0) return -1;
    const int grp = grouped ? 1 : 0;
    if (M > 4 && !(scale && K == 4096)) return -1;           // batches of 5-8 rows exist in the LDS-staged form only
    const dim3 grid((unsigned)((I / 2 + 3) / 4)), block(256);
    const float isd = 1.0f / sqrtf((float)K);
    const size_t lds = (size_t)M * K * 2;
    const bool stage = scale && K == 4096;
#define EVO_MG(MM)                                                                                            \
    if (scale && stage)                                                                                       \
        hipLaunchKernelGGL((gemv_gate_kernel<MM, true, true>), grid, block, lds, s, (const uint4*)x, (const uint4*)scale, \
                           (const uint4*)w12, (uint16_t*)a, (int)I, (int)(K / 8), eps, isd, grp);             \
    else if (scale)                                                                                           \
        hipLaunchKernelGGL((gemv_gate_kernel<MM, true, false>), grid, block, 0, s, (const uint4*)x, (const uint4*)scale, \
                           (const uint4*)w12, (uint16_t*)a, (int)I, (int)(K / 8), eps, isd, grp);             \
    else                                                                                                      \
        hipLaunchKernelGGL((gemv_gate_kernel<MM, false, false>), grid, block, 0, s, (const uint4*)x, (const uint4*)nullptr, \
                           (const uint4*)w12, (uint16_t*)a, (int)I, (int)(K / 8), 0.f, 0.f, grp)
#define EVO_MGS(MM)                                                                                           \
    hipLaunchKernelGGL((gemv_gate_kernel<MM, true, true>), grid, block, lds, s, (const uint4*)x, (const uint4*)scale, \
                       (const uint4*)w12, (uint16_t*)a, (int)I, (int)(K / 8), eps, isd, grp)
    switch (M) {
        case 1: EVO_MG(1); break;
        case 2: EVO_MG(2); break;
        case 3: EVO_MG(3); break;
        case 4: EVO_MG(4); break;
        case 5: EVO_MGS(5); break;
        case 6: EVO_MGS(6); break;
        case 7: EVO_MGS(7); break;
        default: EVO_MGS(8); break;
    }
#undef EVO_MGS
#undef EVO_MG
    return evo_launch_status();
}

extern "C" int evo_mlp_gate_small_m_bf16(const void* x, const void* w12, void* a, int64_t M, int64_t I, int64_t K,
                                         int64_t grouped, void* stream) {
    if (M >= 5 && M <= 64) {
        // 5-64 rows: the MFMA weight-streaming form with the gate in its epilogue (skinny_nw_kernel GATE), both weight layouts
        if (I <= 0 || I % 32 != 0 || K <= 0 || K % 256 != 0 || I > 0x1fffffff) return -1;
        hipStream_t s = (hipStream_t)stream;
#define EVO_SNG(MT) hipLaunchKernelGGL((skinny_nw_kernel<MT, 4, false, true>), dim3((unsigned)(I / 32)), dim3(256), 0, s, (const uint4*)x, (const uint4*)w12,   \
                                       (const uint16_t*)nullptr, (const uint16_t*)nullptr, (uint16_t*)a, (int)M, (int)(2 * I), (int)(K / 8), (float*)nullptr,     \
                                       grouped ? 1 : 2)
        if (M <= 16) EVO_SNG(1); else if (M <= 32) EVO_SNG(2); else if (M <= 48) EVO_SNG(3); else EVO_SNG(4);
#undef EVO_SNG
        return evo_launch_status();
    }
    return mlp_gate_launch(x, nullptr, w12, a, M, I, K, 0.f, grouped, (hipStream_t)stream);
}

extern "C" int evo_norm_mlp_gate_small_m_bf16(const void* x, const void* scale, const void* w12, void* a, int64_t M,
                                              int64_t I, int64_t K, float eps, int64_t grouped, void* stream) {
    if (!scale) return -1;
    return mlp_gate_launch(x, scale, w12, a, M, I, K, eps, grouped, (hipStream_t)stream);
}

extern "C" int evo_linear_small_m_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y,
                                       int64_t M, int64_t N, int64_t K, void* ws, int64_t ws_bytes, void* stream) {
    if (M < 1 || M > 64 || N <= 0 || K <= 0 || K % 8 != 0 || N > 0x7fffffff - 16) return -1;
    if (M > 8 && K % 32 != 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (M >= 5 && K % 32 == 0) {
        // two n tiles per workgroup (every x fragment feeds two weight fragments) where that still leaves a workgroup or more per CU
        static const int nt_env = [] { const char* e = getenv("EVO_SK_NT"); return e ? atoi(e) : 0; }();      // measurement knob: 1 / 2 force, 0 = auto
        const int nt = nt_env ? nt_env : (N >= 8192 && M > 16 ? 2 : 1);
        const dim3 block(512);
        // n-split form: wide layers (a workgroup or more per CU with WAVES n tiles per workgroup), K in whole 256-k chunks
        static const int ns_env = [] { const char* e = getenv("EVO_SK_NSPLIT"); return e ? atoi(e) : -1; }();  // measurement knob: 0 off, 2 / 4 force WAVES
        // (measured, tools/bench_gemv.py: ahead of the k-split form from 17 rows up on the wide layers -- 2.2 -> 3.3+ TB/s at 32 rows, 1.4 -> 2.9 at 64;
        //  behind it at <= 16 rows, where x is a small share of the L2 traffic and the k-split form's 11,000 short waves hide latency better)
        const int waves_n = ns_env >= 0 ? ns_env : (K % 256 == 0 && N >= 8192 && M > 16 ? 4 : 0);
        static const int nw_env = [] { const char* e = getenv("EVO_SK_WLDS"); return e ? atoi(e) : 1; }();    // measurement knob: 0 = fragment-shaped weight requests (skinny_n_kernel)
        static const int nwm_env = [] { const char* e = getenv("EVO_SK_WLDS_MINM"); return e ? atoi(e) : 5; }();    // measurement knob: first row count on the contiguous form (measured ahead of the k-split form from 5 rows up on the wide layers)
        static const int nww_env = [] { const char* e = getenv("EVO_SK_WLDS_WAVES"); return e ? atoi(e) : 0; }();   // measurement knob: 2 / 4 waves per workgroup (0 = auto)
        // narrow layers (N < 8192: out_filter_dense / out_proj / l3 of a 7B block) with a workspace from the caller: four waves per workgroup AND a split
        // over K across workgroups, partial sums through ws (see skinny_nw_kernel SPLITK)
        static const int sk_env = [] { const char* e = getenv("EVO_SK_SPLITK"); return e ? atoi(e) : 1; }();      // measurement knob: 0 off
        static const int skm_env = [] { const char* e = getenv("EVO_SK_SPLITK_MINM"); return e ? atoi(e) : 17; }();
        // (K = 4,096 at <= 32 rows: four chunks per slice are too short a run -- 2.0 against 2.45 TB/s at 17 rows -- the k-split form keeps those)
        if (ws && sk_env && nw_env && ns_env < 0 && K % 256 == 0 && K >= 1024 && N < 8192 && N >= 1024 && N % 64 == 0 && M >= skm_env && (K >= 8192 || M > 32)) {
            int64_t slices = (256 + N / 64 - 1) / (N / 64);              // a workgroup per CU
            if (slices > K / 256) slices = K / 256;
            if (slices > 8) slices = 8;
            if (slices >= 2 && ws_bytes >= slices * M * N * 4 && ((uintptr_t)ws & 15) == 0) {
                const dim3 grid((unsigned)(N / 64), (unsigned)slices);
#define EVO_SNK(MT) hipLaunchKernelGGL((skinny_nw_kernel<MT, 4, true>), grid, dim3(256), 0, s, (const uint4*)x, (const uint4*)w, (const uint16_t*)nullptr,       \
                                       (const uint16_t*)nullptr, (uint16_t*)nullptr, (int)M, (int)N, (int)(K / 8), (float*)ws)
                if (M <= 16) EVO_SNK(1); else if (M <= 32) EVO_SNK(2); else if (M <= 48) EVO_SNK(3); else EVO_SNK(4);
#undef EVO_SNK
                hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)((M * N / 4 + 255) / 256)), dim3(256), 0, s, (const float*)ws, (const uint16_t*)bias,
                                   (const uint16_t*)residual, (uint16_t*)y, (int)M, (int)N, (int)slices);
                return evo_launch_status();
            }
        }
        if (K % 256 == 0 && N >= 8192 && nw_env && M >= nwm_env && ns_env < 0) {
            const int wv = nww_env ? nww_env : (N >= 256 * 64 ? 4 : (N >= 256 * 48 ? 3 : (N >= 256 * 32 ? 2 : 4)));   // a workgroup or more per CU where N allows
#define EVO_SNW(MT, WV) hipLaunchKernelGGL((skinny_nw_kernel<MT, WV>), dim3((unsigned)((N + 16 * WV - 1) / (16 * WV))), dim3(64 * WV), 0, s, (const uint4*)x, \
                                           (const uint4*)w, (const uint16_t*)bias, (const uint16_t*)residual, (uint16_t*)y, (int)M, (int)N, (int)(K / 8))
            if (wv == 4) { if (M <= 16) EVO_SNW(1, 4); else if (M <= 32) EVO_SNW(2, 4); else if (M <= 48) EVO_SNW(3, 4); else EVO_SNW(4, 4); }
            else if (wv == 3) { if (M <= 16) EVO_SNW(1, 3); else if (M <= 32) EVO_SNW(2, 3); else if (M <= 48) EVO_SNW(3, 3); else EVO_SNW(4, 3); }
            else { if (M <= 16) EVO_SNW(1, 2); else if (M <= 32) EVO_SNW(2, 2); else if (M <= 48) EVO_SNW(3, 2); else EVO_SNW(4, 2); }
#undef EVO_SNW
            return evo_launch_status();
        }
        if (waves_n && K % 256 == 0) {
#define EVO_SN(MT, WV)                                                                                                     \
            hipLaunchKernelGGL((skinny_n_kernel<MT, WV>), dim3((unsigned)((N + 16 * WV - 1) / (16 * WV))), dim3(64 * WV), 0, s, (const uint4*)x,  \
                               (const uint4*)w, (const uint16_t*)bias, (const uint16_t*)residual, (uint16_t*)y, (int)M, (int)N, (int)(K / 8))
            if (waves_n == 4) { if (M <= 16) EVO_SN(1, 4); else if (M <= 32) EVO_SN(2, 4); else if (M <= 48) EVO_SN(3, 4); else EVO_SN(4, 4); }
            else { if (M <= 16) EVO_SN(1, 2); else if (M <= 32) EVO_SN(2, 2); else if (M <= 48) EVO_SN(3, 2); else EVO_SN(4, 2); }
#undef EVO_SN
            return evo_launch_status();
        }
#define EVO_SK(MT, NT)                                                                                                     \
        hipLaunchKernelGGL((skinny_mfma_kernel<MT, NT>), dim3((unsigned)((N + 16 * NT - 1) / (16 * NT))), block, 0, s, (const uint4*)x, (const uint4*)w, \
                           (const uint16_t*)bias, (const uint16_t*)residual, (uint16_t*)y, (int)M, (int)N, (int)(K / 8))
        if (nt == 2) { if (M <= 16) EVO_SK(1, 2); else if (M <= 32) EVO_SK(2, 2); else if (M <= 48) EVO_SK(3, 2); else EVO_SK(4, 2); }
        else { if (M <= 16) EVO_SK(1, 1); else if (M <= 32) EVO_SK(2, 1); else if (M <= 48) EVO_SK(3, 1); else EVO_SK(4, 1); }
#undef EVO_SK
        return evo_launch_status();
    }
    switch (M) {
        case 1: gemv_launch<1>(x, w, bias, residual, y, N, K, s); break;
        case 2: gemv_launch<2>(x, w, bias, residual, y, N, K, s); break;
        case 3: gemv_launch<3>(x, w, bias, residual, y, N, K, s); break;
        case 4: gemv_launch<4>(x, w, bias, residual, y, N, K, s); break;
        case 5: gemv_launch<5>(x, w, bias, residual, y, N, K, s); break;
        case 6: gemv_launch<6>(x, w, bias, residual, y, N, K, s); break;
        case 7: gemv_launch<7>(x, w, bias, residual, y, N, K, s); break;
        default: gemv_launch<8>(x, w, bias, residual, y, N, K, s); break;
    }
    return evo_launch_status();
}
