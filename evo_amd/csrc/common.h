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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int evo_launch_status() { return (int)hipGetLastError(); }
