// VALU issue-rate probe for gfx950: v_pk_fma_f32 vs v_fma_f32, as a function of dependency distance (NACC independent
// accumulators in rotation) and waves per SIMD (occupancy capped through the dynamic LDS request).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o build/valu_rate_probe && build/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NACC, bool PK>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
    extern __shared__ char dyn[];
    f32x2 acc[NACC];
    f32x2 a = {a0, a0 * 1.0001f}, b = {b0, b0 * 0.999f};
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc[i][0] = threadIdx.x * 1e-3f + i; acc[i][1] = i * 0.5f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 64 / NACC; ++rep) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (PK) {
                    asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                } else {
                    asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[i][0]) : "v"(a[0]), "v"(b[0]));
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1];
    if (s == 123.456f) out[0] = s + dyn[0];
}

template <int NACC, bool PK>
static void run(int waves_per_simd, float* d_out) {
    // 256-thread blocks = 1 wave per SIMD each; LDS request caps blocks per CU
    const int lds = 160 * 1024 / waves_per_simd - 1024;
    hipFuncSetAttribute((const void*)probe<NACC, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 20000;
    const int grid = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, PK>), dim3(grid), dim3(256), lds, 0, d_out, 100, 1.0001f, 1e-6f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, PK>), dim3(grid), dim3(256), lds, 0, d_out, iters, 1.0001f, 1e-6f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * waves_per_simd;     // 64 instr per iteration per wave
    const double flop = instr_per_simd * 1024 * (PK ? 256.0 : 128.0);
    printf("%-10s NACC=%2d waves/SIMD=%d : %7.3f ms  %6.1f TFLOP/s  %.2f ns/instr/SIMD (%.2f cyc @2.4GHz)\n", PK ? "pk_fma_f32" : "fma_f32",
           NACC, waves_per_simd, ms, flop / ms / 1e9, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    for (int w = 1; w <= 4; ++w) {
        run<1, true>(w, d); run<2, true>(w, d); run<4, true>(w, d); run<8, true>(w, d); run<16, true>(w, d);
        run<1, false>(w, d); run<2, false>(w, d); run<4, false>(w, d); run<8, false>(w, d); run<16, false>(w, d);
    }
    return 0;
}
