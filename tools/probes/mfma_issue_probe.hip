// MFMA issue probe for gfx950, ONE wave per SIMD (256-thread blocks, one per CU): what does an instruction of another
// class cost when it is issued between back-to-back MFMAs of the same wave?  Each mode runs 16 MFMAs per iteration with NX
// extra instructions spread between them (none of them touches a register an MFMA uses; nothing is waited for inside the
// loop); reported: cycles per iteration / 16 from the wave's own clock, and the chip-level rate.
//   hipcc --offload-arch=gfx950 -O3 -w tools/probes/mfma_issue_probe.hip -o build/mfma_issue_probe && build/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum { X_NONE = 0, X_DSREAD, X_DSWRITE, X_GLOAD, X_SALU, X_VALU, X_DMA };

template <int X>
__device__ __forceinline__ void extra(uint32_t la, u32x4& sink, const u32x4& data, const unsigned char* gp, uint32_t goff, int& sacc, float& vacc) {
    __builtin_amdgcn_sched_barrier(0);
    if (X == X_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(sink) : "v"(la));
    if (X == X_DSWRITE) asm volatile("ds_write_b128 %0, %1 offset:32768" ::"v"(la + 32768u), "v"(data) : "memory");
    if (X == X_GLOAD) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sink) : "v"(goff), "s"(gp));
    if (X == X_DMA) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(la) & 0xffffu), "v"(goff), "s"(gp) : "memory", "m0");
    if (X == X_SALU) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
    if (X == X_VALU) asm volatile("v_add_f32 %0, %0, %0" : "+v"(vacc));
    __builtin_amdgcn_sched_barrier(0);
}

// PASS4: v_mfma_f32_16x16x32_bf16 (4 passes) instead of v_mfma_f32_32x32x16_bf16 (8 passes).  EVERY: one extra per EVERY MFMAs.
template <int X, bool PASS4, int EVERY>
__global__ __launch_bounds__(256, 1) void probe(float* out, const u32x4* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    u32x4 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = src[lane + 64 * i]; fb[i] = src[lane + 64 * (i + 4)]; }
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + threadIdx.x * 16;
    u32x4 sink = {0, 0, 0, 0}, data = src[lane];
    const unsigned char* gp = (const unsigned char*)src;
    const uint32_t goff = (uint32_t)(threadIdx.x * 16 + (blockIdx.x & 63) * 4096);
    int sacc = 0; float vacc = 1.f;
    f32x16 acc8[PASS4 ? 1 : 16];
    f32x4 acc4[PASS4 ? 16 : 1];
    for (int i = 0; i < (PASS4 ? 1 : 16); ++i) for (int r = 0; r < 16; ++r) acc8[i][r] = 0.f;
    for (int i = 0; i < (PASS4 ? 16 : 1); ++i) acc4[i] = f32x4{0, 0, 0, 0};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (PASS4) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i & 3]), __builtin_bit_cast(bf16x8, fb[i >> 2]), acc4[i], 0, 0, 0);
            else acc8[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i & 3]), __builtin_bit_cast(bf16x8, fb[i >> 2]), acc8[i], 0, 0, 0);
            if (X != X_NONE && (i % EVERY) == EVERY - 1) extra<X>(la, sink, data, gp, goff, sacc, vacc);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(sink));
    float s = vacc + sacc + sink[0];
    for (int i = 0; i < (PASS4 ? 1 : 16); ++i) s += acc8[i][0];
    for (int i = 0; i < (PASS4 ? 16 : 1); ++i) s += acc4[i][0];
    if (s == 1.2345f) out[100] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0) / (16.f * iters);
}

template <int X, bool PASS4, int EVERY>
static void run(const char* what, float* d_out, u32x4* d_src) {
    const int lds = 140 * 1024;
    (void)hipFuncSetAttribute((const void*)probe<X, PASS4, EVERY>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<X, PASS4, EVERY>), dim3(256), dim3(256), lds, 0, d_out, d_src, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<X, PASS4, EVERY>), dim3(256), dim3(256), lds, 0, d_out, d_src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0, cyc = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&cyc, d_out, 4, hipMemcpyDeviceToHost);
    const double flop = (PASS4 ? 16384.0 : 32768.0) * 16.0 * iters * 4 * 256;
    printf("%-9s + %-28s : %5.1f clk per MFMA, %.3f ms, %5.0f TFLOP/s\n", PASS4 ? "16x16x32" : "32x32x16", what, cyc, ms, flop / ms / 1e9);
}


// GEMM-like mix per 16 (8-pass) or 32 (4-pass) MFMAs: 8 ds_read_b128, NW ds_write_b128, NL global loads of WD dwords per lane
template <bool PASS4, int NW, int NL, int WD>
__global__ __launch_bounds__(256, 1) void mix(float* out, const u32x4* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    u32x4 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = src[lane + 64 * i]; fb[i] = src[lane + 64 * (i + 4)]; }
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + threadIdx.x * 16;
    u32x4 sink = {0, 0, 0, 0}, sink2 = {0, 0, 0, 0}, data = src[lane];
    const unsigned char* gp = (const unsigned char*)src;
    const uint32_t goff = (uint32_t)(threadIdx.x * 16 + (blockIdx.x & 63) * 4096);
    constexpr int NM = PASS4 ? 32 : 16;
    f32x16 acc8[PASS4 ? 1 : 16];
    f32x4 acc4[PASS4 ? 32 : 1];
    for (int i = 0; i < (PASS4 ? 1 : 16); ++i) for (int r = 0; r < 16; ++r) acc8[i][r] = 0.f;
    for (int i = 0; i < (PASS4 ? 32 : 1); ++i) acc4[i] = f32x4{0, 0, 0, 0};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (PASS4) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i & 3]), __builtin_bit_cast(bf16x8, fb[(i >> 2) & 3]), acc4[i], 0, 0, 0);
            else acc8[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i & 3]), __builtin_bit_cast(bf16x8, fb[i >> 2]), acc8[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i % (NM / 8) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(sink) : "v"(la));
            if (NW > 0 && i % (NM / (NW > 0 ? NW : 1)) == 1) asm volatile("ds_write_b128 %0, %1 offset:32768" ::"v"(la + 32768u), "v"(data) : "memory");
            if (NL > 0 && i % (NM / (NL > 0 ? NL : 1)) == (NM / (NL > 0 ? NL : 1)) - 1) {
                if (WD == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sink2) : "v"(goff), "s"(gp));
                if (WD == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(*(unsigned long long*)&sink2) : "v"(goff), "s"(gp));
                if (WD == 1) asm volatile("global_load_dword %0, %1, %2" : "=v"(sink2[0]) : "v"(goff), "s"(gp));
                if (WD == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"((__builtin_amdgcn_readfirstlane(la) & 0xffffu) + 65536u), "v"(goff), "s"(gp) : "memory", "m0");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(sink), "+v"(sink2));
    float s = sink[0] + sink2[0];
    for (int i = 0; i < (PASS4 ? 1 : 16); ++i) s += acc8[i][0];
    for (int i = 0; i < (PASS4 ? 32 : 1); ++i) s += acc4[i][0];
    if (s == 1.2345f) out[100] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(NM * iters);
}

template <bool PASS4, int NW, int NL, int WD>
static void runmix(const char* what, float* d_out, u32x4* d_src) {
    const int lds = 140 * 1024;
    (void)hipFuncSetAttribute((const void*)mix<PASS4, NW, NL, WD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<PASS4, NW, NL, WD>), dim3(256), dim3(256), lds, 0, d_out, d_src, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mix<PASS4, NW, NL, WD>), dim3(256), dim3(256), lds, 0, d_out, d_src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0, cyc = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&cyc, d_out, 4, hipMemcpyDeviceToHost);
    const double flop = 32768.0 * 16.0 * iters * 4 * 256;
    printf("mix %-9s 8 ds_read + %d ds_write + %2d loads x%d dwords per 16 8-pass equivalents (%s): %5.1f clk per MFMA, %.3f ms, %5.0f TFLOP/s\n",
           PASS4 ? "16x16x32" : "32x32x16", NW, NL, WD, what, cyc, ms, flop / ms / 1e9);
}

// The hipBLASLt loop shape: 4-pass MFMAs with exactly ONE other instruction per gap -- per 32 MFMAs: 8 ds_read_b128, NL LDS-DMA
// loads (the M0 update in the gap before each), SALU filler elsewhere.  PASS8: the same gaps behind 8-pass MFMAs (16 of them).
template <int NL, bool PASS8, bool FILL>
__global__ __launch_bounds__(256, 1) void libmix(float* out, const u32x4* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    u32x4 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = src[lane + 64 * i]; fb[i] = src[lane + 64 * (i + 4)]; }
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + threadIdx.x * 16;
    const uint32_t m0v = (__builtin_amdgcn_readfirstlane(la) & 0xffffu) + 65536u;
    u32x4 sink = {0, 0, 0, 0};
    const unsigned char* gp = (const unsigned char*)src;
    const uint32_t goff = (uint32_t)(threadIdx.x * 16 + (blockIdx.x & 63) * 4096);
    int sacc = 0;
    constexpr int NM = PASS8 ? 16 : 32;
    f32x16 acc8[PASS8 ? 16 : 1];
    f32x4 acc4[PASS8 ? 1 : 32];
    for (int i = 0; i < (PASS8 ? 16 : 1); ++i) for (int r = 0; r < 16; ++r) acc8[i][r] = 0.f;
    for (int i = 0; i < (PASS8 ? 1 : 32); ++i) acc4[i] = f32x4{0, 0, 0, 0};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (PASS8) acc8[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i & 3]), __builtin_bit_cast(bf16x8, fb[i >> 2]), acc8[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i & 3]), __builtin_bit_cast(bf16x8, fb[(i >> 2) & 3]), acc4[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int RSTEP = NM / 8;
            const int slot = i % (NM / (NL > 0 ? NL : 1));
            if (i % RSTEP == 0 && RSTEP > 1) asm volatile("ds_read_b128 %0, %1" : "=v"(sink) : "v"(la));
            else if (NL > 0 && slot == 1 % RSTEP + (RSTEP > 2 ? 0 : 0) && RSTEP > 2) asm volatile("s_mov_b32 m0, %0" ::"s"(m0v) : "m0");
            else if (NL > 0 && slot == 2 && RSTEP > 2) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(goff), "s"(gp) : "memory");
            else if (FILL) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
            if (RSTEP <= 2) {   // 8-pass: two instructions per gap where needed (read + m0 / DMA)
                if (i % RSTEP == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(sink) : "v"(la));
                if (NL > 0 && slot == 0) asm volatile("s_mov_b32 m0, %0" ::"s"(m0v) : "m0");
                if (NL > 0 && slot == 1) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(goff), "s"(gp) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(sink));
    float s = sink[0] + sacc;
    for (int i = 0; i < (PASS8 ? 16 : 1); ++i) s += acc8[i][0];
    for (int i = 0; i < (PASS8 ? 1 : 32); ++i) s += acc4[i][0];
    if (s == 1.2345f) out[100] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(NM * iters);
}

template <int NL, bool PASS8, bool FILL>
static void runlib(const char* what, float* d_out, u32x4* d_src) {
    const int lds = 140 * 1024;
    (void)hipFuncSetAttribute((const void*)libmix<NL, PASS8, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((libmix<NL, PASS8, FILL>), dim3(256), dim3(256), lds, 0, d_out, d_src, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((libmix<NL, PASS8, FILL>), dim3(256), dim3(256), lds, 0, d_out, d_src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0, cyc = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&cyc, d_out, 4, hipMemcpyDeviceToHost);
    const double flop = 32768.0 * 16.0 * iters * 4 * 256;
    printf("one-per-gap %s, 8 ds_read + %d LDS-DMA per 16 8-pass equivalents%s (%s): %5.1f clk per MFMA, %.3f ms, %5.0f TFLOP/s\n",
           PASS8 ? "32x32x16" : "16x16x32", NL, FILL ? " + SALU filler" : "", what, cyc, ms, flop / ms / 1e9);
}

int main() {
    float* d; u32x4* s;
    (void)hipMalloc(&d, 4096); (void)hipMalloc(&s, 1 << 20);
    (void)hipMemset(s, 0x3c, 1 << 20);
    run<X_NONE, false, 1>("nothing", d, s);
    run<X_DSREAD, false, 2>("ds_read_b128 per 2", d, s);
    run<X_DSREAD, false, 1>("ds_read_b128 per 1", d, s);
    run<X_DSWRITE, false, 4>("ds_write_b128 per 4", d, s);
    run<X_DSWRITE, false, 1>("ds_write_b128 per 1", d, s);
    run<X_GLOAD, false, 4>("global_load_dwordx4 per 4", d, s);
    run<X_GLOAD, false, 1>("global_load_dwordx4 per 1", d, s);
    run<X_DMA, false, 4>("global_load_lds_dwordx4 per 4", d, s);
    run<X_DMA, false, 1>("global_load_lds_dwordx4 per 1", d, s);
    run<X_SALU, false, 1>("s_add per 1", d, s);
    run<X_VALU, false, 1>("v_add_f32 per 1", d, s);
    run<X_NONE, true, 1>("nothing", d, s);
    run<X_DSREAD, true, 2>("ds_read_b128 per 2", d, s);
    run<X_DSREAD, true, 1>("ds_read_b128 per 1", d, s);
    run<X_DSWRITE, true, 4>("ds_write_b128 per 4", d, s);
    run<X_GLOAD, true, 4>("global_load_dwordx4 per 4", d, s);
    run<X_SALU, true, 1>("s_add per 1", d, s);
    run<X_VALU, true, 1>("v_add_f32 per 1", d, s);
    runmix<false, 0, 0, 4>("reads only", d, s);
    runmix<false, 4, 0, 4>("+ writes", d, s);
    runmix<false, 0, 4, 4>("+ loads x4", d, s);
    runmix<false, 4, 4, 4>("GEMM mix", d, s);
    runmix<false, 4, 8, 2>("GEMM mix, dwordx2 loads", d, s);
    runmix<false, 4, 16, 1>("GEMM mix, dword loads", d, s);
    runmix<true, 0, 0, 4>("reads only", d, s);
    runmix<true, 4, 0, 4>("+ writes", d, s);
    runmix<true, 0, 4, 4>("+ loads x4", d, s);
    runmix<true, 4, 4, 4>("GEMM mix", d, s);
    runmix<true, 4, 8, 2>("GEMM mix, dwordx2 loads", d, s);
    runmix<true, 0, 4, 0>("reads + LDS-DMA x4 (WD=0)", d, s);
    runmix<false, 0, 4, 0>("reads + LDS-DMA x4 (WD=0)", d, s);
    runmix<true, 0, 8, 0>("reads + 2x the DMA", d, s);
    runlib<0, false, false>("reads only", d, s);
    runlib<0, false, true>("reads, every gap filled", d, s);
    runlib<4, false, false>("GEMM rate of DMA", d, s);
    runlib<4, false, true>("GEMM rate of DMA, every gap filled", d, s);
    runlib<8, false, true>("2x DMA, every gap filled", d, s);
    runlib<4, true, false>("8-pass", d, s);
    return 0;
}
