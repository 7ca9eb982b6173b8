// Skinny dense layer for decoding: y[M,N] = x[M,K] . W[N,K]^T (+ bias[N]) (+ residual[M,N]), M <= 8.
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

template <int M, int R>
__global__ __launch_bounds__(256) void gemv_kernel(const uint4* __restrict__ x, const uint4* __restrict__ w,
                                                   const uint16_t* __restrict__ bias, const uint16_t* res,
                                                   uint16_t* y, int N, int nvec) {   // res may alias y
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * R;
    if (n0 >= N) return;
    const uint4* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t n = n0 + r < N ? n0 + r : N - 1;            // rows past the end are computed on a clamp, not stored
        wrow[r] = w + n * nvec;
    }
    float acc[R][M];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = 0.f;

    int v = lane;
    for (; v + 64 < nvec; v += 128) {                        // two k-slices per trip: 2R weight loads in flight
        uint4 w0[R], w1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { w0[r] = ld_stream(wrow[r] + v); w1[r] = ld_stream(wrow[r] + v + 64); }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint4 x0 = x[(int64_t)m * nvec + v], x1 = x[(int64_t)m * nvec + v + 64];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][m] = dot8(w1[r], x1, dot8(w0[r], x0, acc[r][m]));
        }
    }
    for (; v < nvec; v += 64) {
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
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t n = n0 + r;
            if (n < N) {
                const float b = bias ? bf_to_f(bias[n]) : 0.f;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    float o = acc[r][m] + b;
                    if (res) o += bf_to_f(res[(int64_t)m * N + n]);
                    y[(int64_t)m * N + n] = f_to_bf(o);
                }
            }
        }
    }
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
    hipLaunchKernelGGL((gemv_kernel<M, R>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, (const uint4*)x,
                       (const uint4*)w, (const uint16_t*)bias, (const uint16_t*)res, (uint16_t*)y, (int)N, (int)(K / 8));
}

extern "C" int evo_linear_small_m_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y,
                                       int64_t M, int64_t N, int64_t K, void* stream) {
    if (M < 1 || M > 8 || N <= 0 || K <= 0 || K % 8 != 0) return -1;
    hipStream_t s = (hipStream_t)stream;
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
