// Fused scoring tail for gfx950:  hidden [M, K] bf16  x  E[V = 512, K]^T  ->  per-row log-softmax statistics,
// i.e. unembed + log_softmax + gather(next token) (+ entropy) in ONE kernel: the [M, 512] logits never reach HBM
// (67 MB per 8 x 8,193 batch, 1.07 GB at 8 x 131,073).   [REF evo/scoring.py:47-57,81-84,119-121]
//
// Numerics are those of the two-kernel path it replaces (hipBLASLt unembed -> bf16 logits -> evo_logprob_entropy):
// fp32 MFMA accumulation, ONE rounding of every logit to bf16 (the reference's logits are a bf16 tensor), then
// max / sum-exp / gather / entropy in fp32 on those rounded values.
//
// Workgroup = 4 waves = 64 rows x all 512 vocabulary columns; wave w owns columns [128 w, 128 w + 128) as 4 x 2 tiles
// of v_mfma_f32_32x32x16_bf16 computed TRANSPOSED (A operand = E rows, B operand = hidden rows), so a lane holds, for
// its row m = lane & 31 of each row tile, 64 of the row's logits in registers: the row reductions are in-lane, one
// lane^32 exchange, and one 4-way exchange through LDS.  K is walked in steps of 32: a stage is 64 + 512 rows of 64 B
// (36 KiB), filled by global->LDS DMA into two alternating buffers; rows are XOR-swizzled by (row >> 2) & 3 on the
// DMA's SOURCE side (the LDS side of a DMA is lane-linear) so the ds_read_b128 fragment reads are conflict-free.
// Two workgroups per CU (72 KiB each) overlap one's DMA wait with the other's MFMAs; this layer is 0.03 % of the
// model's FLOPs, so the structure is kept simple (compiler-placed waits, two plain barriers per step).
// Entry point and reference citation: include/evo_mi355x.h.
#include "common.h"
#include "../../include/evo_mi355x.h"

#define ST_ROWS 64                         // hidden rows per workgroup
#define ST_V 512                           // vocabulary (= 4 waves x 128 columns)
#define ST_BK 32                           // k per stage
#define ST_ROWB (ST_BK * 2)                // 64 B per LDS row = 4 granules of 16 B
#define ST_XB (ST_ROWS * ST_ROWB)          // 4 KiB
#define ST_STAGE ((ST_ROWS + ST_V) * ST_ROWB)   // 36,864 B = 36 DMA pieces of 1 KiB -> 9 per wave
#define ST_NP 9
#define ST_SCRATCH (4 * ST_ROWS * 3 * 4 + ST_ROWS * 4)   // per-wave (max, sum, sum-xd) + target logit

typedef __attribute__((address_space(3))) void* st_lds_ptr_t;
typedef __attribute__((address_space(1))) const void* st_glb_ptr_t;

__global__ __launch_bounds__(256, 2) void unembed_logprob_kernel(
    const unsigned char* __restrict__ h, const unsigned char* __restrict__ emb, const int64_t* __restrict__ target,
    float* __restrict__ logprob, float* __restrict__ entropy, int64_t M, int K) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * ST_STAGE + ST_SCRATCH];   // the only LDS object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * ST_ROWS;
    const int64_t kb = (int64_t)K * 2;                         // bytes per operand row
    const int nk = K / ST_BK;

    // ---- DMA plan: piece p = wave + 4 jj (jj < 9) covers LDS bytes [p KiB, (p+1) KiB) of the stage: pieces 0..3 are
    //      the hidden rows, 4..35 the embedding rows.  The lane that fills slot s = lane & 3 of row r fetches source
    //      granule s ^ ((r >> 2) & 3) of that row (same 64-byte half line).
    const unsigned char* src[ST_NP];
#pragma unroll
    for (int jj = 0; jj < ST_NP; ++jj) {
        const int p = wave + 4 * jj;
        const int row_in_piece = lane >> 2, s = lane & 3;
        if (p < 4) {
            const int r = p * 16 + row_in_piece;
            int64_t m = m0 + r;
            if (m > M - 1) m = M - 1;                          // ragged last tile: clamp (those rows are never written)
            src[jj] = h + m * kb + ((s ^ ((r >> 2) & 3)) * 16);
        } else {
            const int r = (p - 4) * 16 + row_in_piece;
            src[jj] = emb + (int64_t)r * kb + ((s ^ ((r >> 2) & 3)) * 16);
        }
    }
    auto issue = [&](int k, int buf) {
        unsigned char* dst = smem + buf * ST_STAGE + wave * 1024;
#pragma unroll
        for (int jj = 0; jj < ST_NP; ++jj)
            __builtin_amdgcn_global_load_lds((st_glb_ptr_t)(src[jj] + (int64_t)k * ST_ROWB), (st_lds_ptr_t)(dst + jj * 4096), 16, 0, 0);
    };

    // fragment offsets inside a stage: B operand = hidden rows (m), A operand = embedding rows (n)
    int x_off[2][2], e_off[4][2];                              // [tile][k sub-step]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = j * 32 + l31;
            x_off[j][ks] = r * ST_ROWB + (((2 * ks + half) ^ (r >> 2)) & 3) * 16;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 128 + i * 32 + l31;
            e_off[i][ks] = ST_XB + r * ST_ROWB + (((2 * ks + half) ^ (r >> 2)) & 3) * 16;
        }
    }

    f32x16_t acc[4][2];                                        // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0, 0);
    for (int k = 0; k < nk; ++k) {
        __syncthreads();                                       // stage k has landed (the compiler drains the DMA here)
        if (k + 1 < nk) issue(k + 1, (k + 1) & 1);             // its buffer was last read in step k-1: free since the barrier
        const unsigned char* st = smem + (k & 1) * ST_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t xf[2], ef[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) xf[j] = *(const bf16x8_t*)(st + x_off[j][ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) ef[i] = *(const bf16x8_t*)(st + e_off[i][ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();

    // ---- epilogue.  D[n][m]: a lane holds row m = lane & 31 of row tile j and, per column tile i, the 16 columns
    //      n = (r & 3) + 8 (r >> 2) + 4 half.  Per row: max, sum exp(x - max), sum exp(x - max)(x - max), and the target's
    //      logit, reduced in-lane -> lane ^ 32 -> across the four waves through LDS.
    float* red = (float*)(smem + 2 * ST_STAGE);                // [wave][row][3]
    float* xt = red + 4 * ST_ROWS * 3;                         // [row]
    if (tid < ST_ROWS) xt[tid] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = round_bf(acc[i][j][r]);         // the logit as the reference's bf16 tensor holds it
                mx = fmaxf(mx, acc[i][j][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (half == 0) red[(wave * ST_ROWS + j * 32 + l31) * 3] = mx;
    }
    __syncthreads();
    const int64_t tg0 = (m0 + l31 < M && target) ? target[m0 + l31] : -1;
    const int64_t tg1 = (m0 + 32 + l31 < M && target) ? target[m0 + 32 + l31] : -1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = j * 32 + l31;
        const float mx = fmaxf(fmaxf(red[(0 * ST_ROWS + row) * 3], red[(1 * ST_ROWS + row) * 3]),
                               fmaxf(red[(2 * ST_ROWS + row) * 3], red[(3 * ST_ROWS + row) * 3]));
        const int64_t tg = j ? tg1 : tg0;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[i][j][r] - mx;
                const float ex = __expf(d);
                s1 += ex;
                s2 = fmaf(ex, d, s2);
                const int n = wave * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (tg == n) xt[row] = acc[i][j][r];           // exactly one lane of the workgroup owns the target column
            }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (half == 0) {
            red[(wave * ST_ROWS + row) * 3 + 1] = s1;
            red[(wave * ST_ROWS + row) * 3 + 2] = s2;
        }
    }
    __syncthreads();
    if (tid < ST_ROWS && m0 + tid < M) {
        const int row = tid;
        float mx = -INFINITY, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            mx = fmaxf(mx, red[(w * ST_ROWS + row) * 3]);
            s1 += red[(w * ST_ROWS + row) * 3 + 1];
            s2 += red[(w * ST_ROWS + row) * 3 + 2];
        }
        const float logz = logf(s1);
        if (entropy) entropy[m0 + row] = logz - s2 / s1;
        if (logprob) {
            const int64_t tg = target ? target[m0 + row] : -1;
            logprob[m0 + row] = (tg >= 0 && tg < ST_V) ? xt[row] - mx - logz : 0.f;
        }
    }
}

extern "C" int evo_unembed_logprob_bf16(const void* hidden, const void* emb, const int64_t* target, float* logprob,
                                        float* entropy, int64_t M, int64_t V, int64_t K, void* stream) {
    if (M < 0 || V != ST_V || K <= 0 || K % ST_BK != 0 || K > 0x3fffffff) return -1;
    if (M == 0) return 0;
    const int64_t tiles = (M + ST_ROWS - 1) / ST_ROWS;
    if (tiles > 0x7fffffff) return -1;
    hipLaunchKernelGGL(unembed_logprob_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)hidden, (const unsigned char*)emb, target, logprob, entropy, M, (int)K);
    return evo_launch_status();
}
