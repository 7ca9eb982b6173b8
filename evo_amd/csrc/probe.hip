// Box-calibration probes for bench.py's `box` block (no reference counterpart: measurement infrastructure behind the same C ABI).
// Two fixed kernels whose rates depend on the BOX (HBM, clocks under the power cap) and not on anything else in this library, so that
// a driver-timed headline can be compared across boxes and rounds:
//   evo_probe_copy_f4   16 bytes per lane grid-stride copy (the guide's "float4 copy": 6.29 TB/s on the reference box)
//   evo_probe_mfma_bf16 register-resident v_mfma_f32_16x16x32_bf16 stream, one wave per SIMD, pseudo-random operands (never zeros: a
//                       zero-operand MFMA loop clocks 15-20 % higher, MI355X_MICROARCH.md), 16 independent accumulators
#include "common.h"
#include "../../include/evo_mi355x.h"

typedef uint32_t p_u32x4 __attribute__((ext_vector_type(4)));

// one workgroup = one contiguous 16 KiB piece (256 lanes x 4 x 16 B): four loads in flight per lane, then four stores; one piece per
// workgroup and a grid of n / 16 KiB workgroups (the dispatcher balances the channels; a 4,096-workgroup grid-stride loop measured 4.6 TB/s)
__global__ __launch_bounds__(256) void probe_copy_kernel(const p_u32x4* __restrict__ src, p_u32x4* __restrict__ dst, int64_t n16) {
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (base + 768 < n16) {
        const p_u32x4 a = __builtin_nontemporal_load(src + base), b = __builtin_nontemporal_load(src + base + 256),
                      c = __builtin_nontemporal_load(src + base + 512), d = __builtin_nontemporal_load(src + base + 768);
        __builtin_nontemporal_store(a, dst + base); __builtin_nontemporal_store(b, dst + base + 256);
        __builtin_nontemporal_store(c, dst + base + 512); __builtin_nontemporal_store(d, dst + base + 768);
    } else {
        for (int64_t i = base; i < n16; i += 256) dst[i] = src[i];
    }
}

extern "C" int evo_probe_copy_f4(const void* src, void* dst, int64_t nbytes, void* stream) {
    if (!src || !dst || nbytes <= 0 || nbytes % 16 != 0) return -1;
    const int64_t n16 = nbytes / 16, blocks = (n16 + 1023) / 1024;
    if (blocks > 0x7fffffff) return -1;
    hipLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const p_u32x4*)src, (p_u32x4*)dst, n16);
    return evo_launch_status();
}

__device__ __forceinline__ uint32_t probe_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// a bf16 pair in [-2, 2) with random mantissas and signs (exponent field 0x7e / 0x7f / 0x80 region kept finite and normal)
__device__ __forceinline__ uint32_t probe_bf16_pair(uint32_t h) {
    const uint32_t lo = (h & 0x807fu) | 0x3f00u | ((h >> 7) & 0x0080u);
    const uint32_t hi = ((h >> 16) & 0x807fu) | 0x3f00u | ((h >> 23) & 0x0080u);
    return lo | (hi << 16);
}

__global__ __launch_bounds__(256, 1) void probe_mfma_kernel(float* out, int iters) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    p_u32x4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            fa[i][d] = probe_bf16_pair(probe_hash(t * 32u + i * 4u + d));
            fb[i][d] = probe_bf16_pair(probe_hash(t * 32u + 16u + i * 4u + d));
        }
    f32x4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i])); }
    asm volatile("s_nop 7" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        // accumulators pinned to AGPRs by inline asm (as a builtin hipcc shuffles them through overlapping register tuples); one s_nop per
        // MFMA: back-to-back 4-pass MFMAs of one wave issue every 28 clocks, with any instruction between them every 16-18
        // (profiles/r02_gemm_notes.txt).  An accumulator is touched once per 16 MFMAs: no hazard to pad.
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 0" : "+a"(acc[i]) : "v"(fa[i & 3]), "v"(fb[i >> 2]));
        // keep the operands moving: the sign pattern of one A fragment word flips every trip (one VALU op per 16 MFMAs)
        fa[0][0] ^= 0x80008000u;
        asm volatile("" : "+v"(fa[0]));
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (out) out[t] = s;
}

// `out` [n_blocks * 256] fp32 (sink; may be NULL), `n_blocks` workgroups of 4 waves, `iters` trips of 16 MFMAs (16 x 16 x 32) per wave:
// flop = n_blocks * 4 * iters * 16 * 16384
extern "C" int evo_probe_mfma_bf16(float* out, int64_t n_blocks, int64_t iters, void* stream) {
    if (n_blocks <= 0 || n_blocks > (1 << 20) || iters <= 0 || iters > 0x7fffffff) return -1;
    hipLaunchKernelGGL(probe_mfma_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, out, (int)iters);
    return evo_launch_status();
}
