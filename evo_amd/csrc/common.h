// Shared device helpers for the gfx950 kernels of libevo_mi355x.so.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EVO_WAVE 64

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));   // MFMA A/B fragment: 8 bf16 in 4 VGPRs

// bf16 <-> f32.  A bf16 pair lives in one dword: low half = even element.
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf_to_f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even pack; lowers to one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ uint16_t f_to_bf(float f) { return (uint16_t)(pack_bf2(f, 0.f) & 0xffffu); }
// value of f after a round trip through bf16
__device__ __forceinline__ float round_bf(float f) { return bf_lo(pack_bf2(f, 0.f)); }

// Wave-wide butterfly reductions.  The four steps inside a 16-lane row are DPP operands of the add itself (no LDS traffic:
// a __shfl_xor is a ds_bpermute, and the 4-16 reductions that end every wave of a weight-streaming launch all hit the LDS
// pipe at the same moment); after the xor-1 and xor-2 steps the lanes of a quad agree bit for bit, so row_half_mirror /
// row_mirror fetch the same value a xor-4 / xor-8 exchange would.  Only the two cross-row steps go through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_fetch(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#define EVO_DPP_XOR1 0xB1          /* quad_perm [1,0,3,2] */
#define EVO_DPP_XOR2 0x4E          /* quad_perm [2,3,0,1] */
#define EVO_DPP_HALF_MIRROR 0x141
#define EVO_DPP_MIRROR 0x140
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_fetch<EVO_DPP_XOR1>(v);
    v += dpp_fetch<EVO_DPP_XOR2>(v);
    v += dpp_fetch<EVO_DPP_HALF_MIRROR>(v);
    v += dpp_fetch<EVO_DPP_MIRROR>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_fetch<EVO_DPP_XOR1>(v));
    v = fmaxf(v, dpp_fetch<EVO_DPP_XOR2>(v));
    v = fmaxf(v, dpp_fetch<EVO_DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_fetch<EVO_DPP_MIRROR>(v));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// erf on a pair of lanes' values: x * P(x^2) / Q(x^2) on [-4, 4] (degree 6 / 4 minimax, |error| <= 4.2e-7 -- a tenth
// of a bf16 rounding of the gated product), all in packed fp32 FMAs plus one v_rcp per value.  libdevice's erff made
// this kernel VALU-bound (0.86 ms where the HBM floor is 0.69 ms at 8 x 8,193 tokens).
__device__ __forceinline__ f32x2_t erf2(f32x2_t x) {
    const f32x2_t lim = {4.0f, 4.0f};
    x = __builtin_elementwise_min(__builtin_elementwise_max(x, -lim), lim);
    const f32x2_t x2 = x * x;
#define EVO_C2(c) f32x2_t { c, c }
    f32x2_t p = EVO_C2(-2.72614225801306e-10f);
    p = __builtin_elementwise_fma(p, x2, EVO_C2(2.77068142495902e-08f));
    p = __builtin_elementwise_fma(p, x2, EVO_C2(-2.10102402082508e-06f));
    p = __builtin_elementwise_fma(p, x2, EVO_C2(-5.69250639462346e-05f));
    p = __builtin_elementwise_fma(p, x2, EVO_C2(-7.34990630326855e-04f));
    p = __builtin_elementwise_fma(p, x2, EVO_C2(-2.95459980854025e-03f));
    p = __builtin_elementwise_fma(p, x2, EVO_C2(-1.60960333262415e-02f));
    f32x2_t q = EVO_C2(-1.45660718464996e-05f);
    q = __builtin_elementwise_fma(q, x2, EVO_C2(-2.13374055278905e-04f));
    q = __builtin_elementwise_fma(q, x2, EVO_C2(-1.68282697438203e-03f));
    q = __builtin_elementwise_fma(q, x2, EVO_C2(-7.37332916720468e-03f));
    q = __builtin_elementwise_fma(q, x2, EVO_C2(-1.42647390514189e-02f));
#undef EVO_C2
    const f32x2_t r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
    return x * p * r;
}

// exact-erf GELU gate on a pair: 0.5 u (1 + erf(u / sqrt 2)) * w
__device__ __forceinline__ f32x2_t gelu_gate2(f32x2_t u, f32x2_t w) {
    const f32x2_t hu = u * 0.5f;
    return __builtin_elementwise_fma(hu, erf2(u * 0.70710678118654752f), hu) * w;
}

static inline int evo_launch_status() { return (int)hipGetLastError(); }
