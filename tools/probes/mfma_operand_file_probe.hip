// Which register FILE an 8-pass MFMA's operands come from, and what it costs -- one wave per SIMD (256-thread blocks, one per CU), 16 independent
// accumulator tuples, nothing else in the loop.  Question behind it (round 6, csrc/attn_w64.hip): a trip's 64 v_mfma_f32_32x32x16_bf16 take ~44 cycles
// each, not 32; QK^T reads A and B from AGPRs (K / Q fragments) and writes VGPRs, P.V reads A / B from VGPRs and accumulates in AGPRs.
//   hipcc --offload-arch=gfx950 -O3 -w tools/probes/mfma_operand_file_probe.hip -o build/mfma_operand_file_probe && build/mfma_operand_file_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: A v, B v, C/D a      1: A a, B a, C/D v      2: A a, B v, C/D a      3: A v, B v, C/D v     4: A a, B a, C/D a
// EXTRA: number of v_add_f32 (independent) between two MFMAs;  EXP: number of v_exp_f32 between two MFMAs
template <int MODE, int EXTRA, int EXP>
__global__ __launch_bounds__(256, 1) void probe(float* out, const u32x4* src, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = src[lane + 64 * i]; fb[i] = src[lane + 64 * (i + 4)]; }
    if (MODE == 1 || MODE == 2 || MODE == 4) for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(fa[i]));
    if (MODE == 1 || MODE == 4) for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(fb[i]));
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; if (MODE == 0 || MODE == 2 || MODE == 4) asm volatile("" : "+a"(acc[i])); else asm volatile("" : "+v"(acc[i])); }
    float va[8] = {1, 2, 3, 4, 5, 6, 7, 8}, ve[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    asm volatile("s_nop 7");
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(fa[i & 3]), "v"(fb[i >> 1]));
            if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(fa[i & 3]), "a"(fb[i >> 1]));
            if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "a"(fa[i & 3]), "v"(fb[i >> 1]));
            if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa[i & 3]), "v"(fb[i >> 1]));
            if (MODE == 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "a"(fa[i & 3]), "a"(fb[i >> 1]));
#pragma unroll
            for (int e = 0; e < EXTRA; ++e) asm volatile("v_add_f32 %0, %0, %0" : "+v"(va[(i + e) & 7]));
#pragma unroll
            for (int e = 0; e < EXP; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(ve[(i + e) & 7]));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 7\n\ts_nop 7");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) { f32x16 t = acc[i]; asm volatile("" : "+v"(t)); s += t[0] + va[i] + ve[i]; }
    if (s == 1.2345f) out[100] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0) / (8.f * iters);
}

template <int MODE, int EXTRA, int EXP>
static void run(const char* what, float* d_out, u32x4* d_src) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, EXTRA, EXP>), dim3(256), dim3(256), 0, 0, d_out, d_src, 200);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, EXTRA, EXP>), dim3(256), dim3(256), 0, 0, d_out, d_src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0, cyc = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&cyc, d_out, 4, hipMemcpyDeviceToHost);
    const double flop = 32768.0 * 8.0 * iters * 4 * 256;
    printf("%-44s + %d v_add + %d v_exp per MFMA : %6.1f clk per MFMA, %7.0f TFLOP/s\n", what, EXTRA, EXP, cyc, flop / ms / 1e9);
}

// ---- realistic side streams beside P.V-form MFMAs (A v, B v, C a), 64 distinct source registers like a score tile:
//   KIND 0: NCH chains of v_max3_f32 (t = max3(t, e[i], e[i+1])), PER per MFMA       KIND 1: independent v_fma_f32 d = fma(e, c, n), PER per MFMA
//   KIND 2: the exp stream's mix per 10: 4 v_exp_f32 (distinct src), 4 v_add_f32 (2 sums), 2 v_cvt_pk_bf16_f32;  PER per MFMA
//   KIND 3: v_max_f32 pairs tree (independent within a level)
template <int KIND, int NCH, int PER>
__global__ __launch_bounds__(256, 1) void probe2(float* out, const u32x4* src, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = src[lane + 64 * i]; fb[i] = src[lane + 64 * (i + 4)]; }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; asm volatile("" : "+a"(acc[i])); }
    float e[64], t[4] = {0.f, 0.f, 0.f, 0.f}, sa = 0.f, sb = 0.f, tp[4];
    uint32_t pk[2] = {0, 0};
    for (int i = 0; i < 64; ++i) { e[i] = (float)((lane * 7 + i * 13) & 31) * 0.03125f - 0.5f; asm volatile("" : "+v"(e[i])); }
    const float c = 0.1275f, n = -0.25f;
    asm volatile("s_nop 7");
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int q = 0;                                             // position in the side stream (compile-time after unrolling)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(fa[i & 3]), "v"(fb[i >> 1]));
#pragma unroll
            for (int u = 0; u < PER; ++u, ++q) {
                if (KIND == 0) { const int ch = q % NCH, st = (q / NCH) % 30; asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(t[ch]) : "v"(e[2 * st + ch]), "v"(e[2 * st + 1 + ch])); }
                if (KIND == 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[q & 63]) : "v"(c), "v"(n)); }
                if (KIND == 2) {
                    const int o = q % 10, g = (q / 10) % 16;
                    if (o < 4) asm volatile("v_exp_f32 %0, %1" : "=v"(tp[o]) : "v"(e[4 * g + o]));
                    else if (o < 8) { if (o & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sb) : "v"(tp[o - 4])); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(sa) : "v"(tp[o - 4])); }
                    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[o - 8]) : "v"(tp[2 * (o - 8)]), "v"(tp[2 * (o - 8) + 1]));
                }
                if (KIND == 3) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(e[(q & 31)]) : "v"(e[(q & 31)]), "v"(e[32 + (q & 31)])); }
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 7\n\ts_nop 7");
    float s = t[0] + t[1] + t[2] + t[3] + sa + sb + __uint_as_float(pk[0]) + __uint_as_float(pk[1]);
    for (int i = 0; i < 64; ++i) s += e[i];
    for (int i = 0; i < 8; ++i) { f32x16 tt = acc[i]; asm volatile("" : "+v"(tt)); s += tt[0]; }
    if (s == 1.2345f) out[100] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0) / (8.f * iters);
}

template <int KIND, int NCH, int PER>
static void run2(const char* what, float* d_out, u32x4* d_src) {
    const int iters = 20000;
    hipLaunchKernelGGL((probe2<KIND, NCH, PER>), dim3(256), dim3(256), 0, 0, d_out, d_src, 200);
    hipLaunchKernelGGL((probe2<KIND, NCH, PER>), dim3(256), dim3(256), 0, 0, d_out, d_src, iters);
    (void)hipDeviceSynchronize();
    float cyc = 0; (void)hipMemcpy(&cyc, d_out, 4, hipMemcpyDeviceToHost);
    printf("P.V-form MFMA + %d per MFMA of %-58s : %6.1f clk per MFMA\n", PER, what, cyc);
}

int main() {
    float* d_out; u32x4* d_src;
    (void)hipMalloc(&d_out, 4096); (void)hipMalloc(&d_src, 64 * 8 * 16);
    uint32_t h[64 * 8 * 4];
    for (int i = 0; i < 64 * 8 * 4; ++i) { uint32_t x = i * 2654435761u; x ^= x >> 15; h[i] = (x & 0x807f807fu) | 0x3f003f00u; }
    (void)hipMemcpy(d_src, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0, 0>("A v, B v, C a  (P.V form)", d_out, d_src);
    run<1, 0, 0>("A a, B a, C v  (QK^T form)", d_out, d_src);
    run<2, 0, 0>("A a, B v, C a", d_out, d_src);
    run<3, 0, 0>("A v, B v, C v", d_out, d_src);
    run<4, 0, 0>("A a, B a, C a", d_out, d_src);
    run<0, 4, 0>("A v, B v, C a  (P.V form)", d_out, d_src);
    run<1, 4, 0>("A a, B a, C v  (QK^T form)", d_out, d_src);
    run<0, 6, 0>("A v, B v, C a  (P.V form)", d_out, d_src);
    run<1, 6, 0>("A a, B a, C v  (QK^T form)", d_out, d_src);
    run<0, 8, 0>("A v, B v, C a  (P.V form)", d_out, d_src);
    run<1, 3, 2>("A a, B a, C v  (QK^T form)", d_out, d_src);
    run<1, 0, 2>("A a, B a, C v  (QK^T form)", d_out, d_src);
    run<1, 0, 1>("A a, B a, C v  (QK^T form)", d_out, d_src);
    run<0, 0, 2>("A v, B v, C a  (P.V form)", d_out, d_src);
    run<1, 0, 3>("A a, B a, C v  (QK^T form)", d_out, d_src);
    run2<0, 2, 4>("v_max3_f32, 2 chains (the row max as it is)", d_out, d_src);
    run2<0, 4, 4>("v_max3_f32, 4 chains", d_out, d_src);
    run2<0, 1, 4>("v_max3_f32, 1 chain", d_out, d_src);
    run2<0, 2, 2>("v_max3_f32, 2 chains", d_out, d_src);
    run2<3, 0, 4>("v_max_f32 independent pairs", d_out, d_src);
    run2<1, 0, 4>("v_fma_f32 independent (the exponent fma)", d_out, d_src);
    run2<1, 0, 6>("v_fma_f32 independent", d_out, d_src);
    run2<2, 0, 5>("exp stream mix (4 exp, 4 add, 2 pack per 10)", d_out, d_src);
    run2<2, 0, 3>("exp stream mix", d_out, d_src);
    run2<2, 0, 4>("exp stream mix", d_out, d_src);
    return 0;
}
