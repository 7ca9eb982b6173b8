// Memory-bound token-local kernels: embedding gather, RMSNorm(+bias/residual write), RoPE, GELU gate,
// scoring tail.  All are HBM-bound: 16-byte (8 x bf16) accesses per lane, fp32 math, one rounding
// on the way out.  Entry points and reference citations: include/evo_mi355x.h.
#include "common.h"
#include "../../include/evo_mi355x.h"

// ------------------------------------------------------------------------------------------- embed
// An id outside [0, vocab) never indexes the table: its row is written as zeros and *bad_flag (if given) is set, so
// the host can raise where the reference's F.embedding would device-assert.
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const uint4* __restrict__ w,
                                                    uint4* __restrict__ out, int64_t n_tok, int nvec, int64_t vocab,
                                                    int* __restrict__ bad_flag) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = n_tok * nvec;
    for (; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t tok = i / nvec;
        int c = (int)(i - tok * nvec);
        const int64_t id = ids[tok];
        if (id >= 0 && id < vocab) {
            out[i] = w[id * nvec + c];
        } else {
            out[i] = make_uint4(0u, 0u, 0u, 0u);
            if (bad_flag && c == 0) *bad_flag = 1;
        }
    }
}

extern "C" int evo_embed_bf16(const int64_t* ids, const void* weight, void* out, int64_t n_tok, int64_t D,
                              int64_t vocab, int* bad_flag, void* stream) {
    if (D % 8 != 0 || n_tok < 0 || vocab <= 0) return -1;
    if (n_tok == 0) return 0;
    int nvec = (int)(D / 8);
    int64_t total = n_tok * nvec;
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(embed_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, (const uint4*)weight,
                       (uint4*)out, n_tok, nvec, vocab, bad_flag);
    return evo_launch_status();
}

// ------------------------------------------------------------------------------------------- rmsnorm
// One wave per row; the row (<= 4096 elements) stays in registers between the square-sum and the
// scale pass, so HBM sees exactly one read and one write (two writes with the bias/residual form).
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]); v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    return v;
}

__device__ __forceinline__ int64_t rms_out_row(int64_t row, int64_t T, int64_t Tp, int64_t Tm, int64_t tail0) {
    if (T == 0) return row;
    const int64_t b = row / T, t = row - b * T;
    return t < Tm ? b * Tp + t : tail0 + b * (T - Tm) + (t - Tm);
}

template <bool HAS_BIAS, int NV>   // NV = register-cached 16-byte vectors per lane (row <= NV*512 elements)
__global__ __launch_bounds__(256) void rmsnorm_kernel(uint4* __restrict__ x, const uint4* __restrict__ bias,
                                                      const uint4* __restrict__ scale, uint4* __restrict__ out,
                                                      int64_t M, int nvec, float eps, float inv_sqrt_d, int64_t T, int64_t Tp, int64_t Tm, int64_t tail0) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        uint4* xr = x + row * nvec;
        uint4 v[NV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int idx = lane + 64 * i;
            if (idx < nvec) {
                v[i] = xr[idx];
                float f[8];
                unpack8(v[i], f);
                if (HAS_BIAS) {
                    float b[8];
                    unpack8(bias[idx], b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += b[e];
                    v[i] = pack8(f);            // the updated row is what is stored AND what is normed
                    xr[idx] = v[i];
                    unpack8(v[i], f);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
            }
        }
        ss = wave_sum(ss);
        const float inv = 1.0f / (sqrtf(ss) * inv_sqrt_d + eps);
        // (evo_rmsnorm_rows_bf16: token t < Tm of batch row b = row / T goes to row b * Tp + t, the row's last T - Tm tokens to the
        //  compact tail rows tail0 + b * (T - Tm) + (t - Tm))
        uint4* orow = out + rms_out_row(row, T, Tp, Tm, tail0) * nvec;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int idx = lane + 64 * i;
            if (idx < nvec) {
                float f[8], s[8];
                unpack8(v[i], f);
                unpack8(scale[idx], s);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = s[e] * (f[e] * inv);
                orow[idx] = pack8(f);
            }
        }
    }
}

// rows longer than the register cache: re-read the row for the second pass (served by L2)
template <bool HAS_BIAS>
__global__ __launch_bounds__(256) void rmsnorm_long_kernel(uint4* __restrict__ x, const uint4* __restrict__ bias,
                                                           const uint4* __restrict__ scale, uint4* __restrict__ out,
                                                           int64_t M, int nvec, float eps, float inv_sqrt_d, int64_t T, int64_t Tp, int64_t Tm, int64_t tail0) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        uint4* xr = x + row * nvec;
        float ss = 0.f;
        for (int idx = lane; idx < nvec; idx += 64) {
            uint4 v = xr[idx];
            float f[8];
            unpack8(v, f);
            if (HAS_BIAS) {
                float b[8];
                unpack8(bias[idx], b);
                for (int e = 0; e < 8; ++e) f[e] += b[e];
                v = pack8(f);
                xr[idx] = v;
                unpack8(v, f);
            }
            for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
        }
        ss = wave_sum(ss);
        const float inv = 1.0f / (sqrtf(ss) * inv_sqrt_d + eps);
        uint4* orow = out + rms_out_row(row, T, Tp, Tm, tail0) * nvec;
        for (int idx = lane; idx < nvec; idx += 64) {
            float f[8], s[8];
            unpack8(xr[idx], f);
            unpack8(scale[idx], s);
            for (int e = 0; e < 8; ++e) f[e] = s[e] * (f[e] * inv);
            orow[idx] = pack8(f);
        }
    }
}

static int rmsnorm_launch(void* x, const void* bias, const void* scale, void* out, int64_t M, int64_t D, float eps, int64_t T,
                          int64_t Tp, int64_t Tm, int64_t tail0, void* stream) {
    if (D % 8 != 0 || D <= 0 || M < 0) return -1;
    if (M == 0) return 0;
    int nvec = (int)(D / 8);
    int64_t blocks = (M + 3) / 4;
    int grid = (int)(blocks < 16384 ? blocks : 16384);
    float isd = 1.0f / sqrtf((float)D);
    hipStream_t s = (hipStream_t)stream;
#define EVO_RMS_LAUNCH(K)                                                                                          \
    hipLaunchKernelGGL(K, dim3(grid), dim3(256), 0, s, (uint4*)x, (const uint4*)bias, (const uint4*)scale,        \
                       (uint4*)out, M, nvec, eps, isd, T, Tp, Tm, tail0)
    if (nvec <= 64 * 2) {
        if (bias) EVO_RMS_LAUNCH((rmsnorm_kernel<true, 2>)); else EVO_RMS_LAUNCH((rmsnorm_kernel<false, 2>));
    } else if (nvec <= 64 * 8) {
        if (bias) EVO_RMS_LAUNCH((rmsnorm_kernel<true, 8>)); else EVO_RMS_LAUNCH((rmsnorm_kernel<false, 8>));
    } else {
        if (bias) EVO_RMS_LAUNCH((rmsnorm_long_kernel<true>)); else EVO_RMS_LAUNCH((rmsnorm_long_kernel<false>));
    }
#undef EVO_RMS_LAUNCH
    return evo_launch_status();
}

extern "C" int evo_rmsnorm_bf16(void* x, const void* bias, const void* scale, void* out, int64_t M, int64_t D,
                                float eps, void* stream) {
    return rmsnorm_launch(x, bias, scale, out, M, D, eps, 0, 0, 0, 0, stream);
}

// The same norm with the output rows in the order of a channel-major z^T (HipOps.zt_layout): token t < Tm of batch row b at row
// b * Tp + t (Tp >= Tm: the pad rows are not written), the row's last T - Tm tokens compactly at rows tail0 + b * (T - Tm) + (t - Tm) -- the
// inputs of the swapped-operand Hyena projection (main rows) and of the weight-streaming kernel (tail rows).  M = B * T rows of x; Tm = T: no tail.
extern "C" int evo_rmsnorm_rows_bf16(void* x, const void* bias, const void* scale, void* out, int64_t M, int64_t D, float eps,
                                     int64_t T, int64_t Tp, int64_t Tm, int64_t tail0, void* stream) {
    if (T <= 0 || Tm <= 0 || Tm > T || Tp < Tm || M % T != 0 || (Tm < T && tail0 < M / T * Tp)) return -1;
    return rmsnorm_launch(x, bias, scale, out, M, D, eps, T, Tp, Tm, tail0, stream);
}

// The RMSNorm factor 1 / (rms(x_row) + eps) as a vector, for the dense layers that take the norm in their epilogue (csrc/gemm.hip, NF):
// rows [0, M_main) from the partial sums of squares the stream-writing dense layers emit (ss [n_strips][ss_ld] fp32, added in strip
// order: bit-reproducible), rows [M_main, M) -- the sliver the weight-streaming kernel wrote -- from x itself, one wave per row.  Same
// arithmetic as rmsnorm_kernel's `inv` [REF stripedhyena/layers.py RMSNorm.forward].
__global__ __launch_bounds__(256) void rms_finalize_kernel(const float* __restrict__ ss, int n_strips, int64_t ss_ld, const uint4* __restrict__ x,
                                                           int64_t M_main, int64_t M, int nvec, float eps, float inv_sqrt_d,
                                                           float* __restrict__ rstd, int main_blocks) {
    if ((int)blockIdx.x < main_blocks) {
        const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (m >= M_main) return;
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= n_strips; k += 8) {                  // eight independent requests in flight, added in strip order
            float p[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) p[e] = ss[(int64_t)(k + e) * ss_ld + m];
#pragma unroll
            for (int e = 0; e < 8; ++e) s += p[e];
        }
        for (; k < n_strips; ++k) s += ss[(int64_t)k * ss_ld + m];
        rstd[m] = 1.0f / (sqrtf(s) * inv_sqrt_d + eps);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t row = M_main + ((int64_t)blockIdx.x - main_blocks) * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const uint4* xr = x + row * nvec;
    float s = 0.f;
    for (int idx = lane; idx < nvec; idx += 64) {
        float f[8];
        unpack8(xr[idx], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(f[e], f[e], s);
    }
    s = wave_sum(s);
    if (lane == 0) rstd[row] = 1.0f / (sqrtf(s) * inv_sqrt_d + eps);
}

extern "C" int evo_rms_finalize_f32(const float* sumsq, int64_t n_strips, int64_t ss_ld, const void* x, int64_t M_main, int64_t M, int64_t D,
                                    float eps, float* rstd, void* stream) {
    if (M <= 0 || M_main < 0 || M_main > M || D <= 0 || D % 8 != 0 || !rstd || (M_main > 0 && (!sumsq || n_strips <= 0 || ss_ld < M_main))
        || (M_main < M && !x)) return -1;
    const int main_blocks = (int)((M_main + 255) / 256);
    const int tail_blocks = (int)((M - M_main + 3) / 4);
    hipLaunchKernelGGL(rms_finalize_kernel, dim3((unsigned)(main_blocks + tail_blocks)), dim3(256), 0, (hipStream_t)stream, sumsq, (int)n_strips,
                       ss_ld, (const uint4*)x, M_main, M, (int)(D / 8), eps, 1.0f / sqrtf((float)D), rstd, main_blocks);
    return evo_launch_status();
}

// ------------------------------------------------------------------------------------------- rope
// qkv [B,T,3,H,hd]; one thread rotates 8 pairs (i..i+7, i+hd/2..) of one (b,t,q|k,h) row.  One workgroup per token (grid-stride):
// the token's position and row origin are wave-uniform (computed once, on the scalar unit), the per-thread index arithmetic is 32-bit
// on small numbers (the first form decomposed a 64-bit flat index with five divisions per thread and ran at 1.4 TB/s: 0.39 ms per
// launch at 8 x 8,193; same expression, same bits).
__global__ __launch_bounds__(256) void rope_kernel(uint4* __restrict__ qkv, const float4* __restrict__ cos_t,
                                                   const float4* __restrict__ sin_t, int64_t n_tok, int64_t T, int H,
                                                   int hd, float q_scale) {
    const int half_vec = hd / 16;                 // 16-byte vectors per half row
    const int per_tok = 2 * H * half_vec;         // (q | k) x heads x vectors of the first half
    const int row_vecs = hd / 8;
    for (int64_t n = blockIdx.x; n < n_tok; n += gridDim.x) {
        const int64_t t = n % T;
        uint4* tok = qkv + n * 3 * H * row_vecs;
        const float4* cp0 = cos_t + t * (hd / 2) / 4;
        const float4* sp0 = sin_t + t * (hd / 2) / 4;
        for (int j = threadIdx.x; j < per_tok; j += 256) {
            const int c = j % half_vec;
            const int hw = j / half_vec;          // which * H + h: the row's index among the token's 2 H (q | k) rows
            uint4* row = tok + hw * row_vecs;
            uint4 a = row[c];
            uint4 bb = row[half_vec + c];
            float x0[8], x1[8], co[8], si[8];
            unpack8(a, x0);
            unpack8(bb, x1);
            const float4* cp = cp0 + c * 2;
            const float4* sp = sp0 + c * 2;
            float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
            co[0] = c0.x; co[1] = c0.y; co[2] = c0.z; co[3] = c0.w; co[4] = c1.x; co[5] = c1.y; co[6] = c1.z; co[7] = c1.w;
            si[0] = s0.x; si[1] = s0.y; si[2] = s0.z; si[3] = s0.w; si[4] = s1.x; si[5] = s1.y; si[6] = s1.z; si[7] = s1.w;
            float o0[8], o1[8];
            // q rows (hw < H) may carry the attention's softmax_scale * log2(e) (q_scale: folded into the ONE rounding of the rotated
            // value, so that a score is an exponent and csrc/attn_w64.hip's PRE form needs no per-score multiply); 1 = plain rotary, exact
            const float qs = hw < H ? q_scale : 1.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o0[e] = (x0[e] * co[e] - x1[e] * si[e]) * qs;
                o1[e] = (x0[e] * si[e] + x1[e] * co[e]) * qs;
            }
            row[c] = pack8(o0);
            row[half_vec + c] = pack8(o1);
        }
    }
}

extern "C" int evo_rope_qk_bf16(void* qkv, const float* cos_t, const float* sin_t, int64_t B, int64_t T, int64_t H,
                                int64_t hd, float q_scale, void* stream) {
    if (hd % 16 != 0 || B < 0 || T < 0 || !(q_scale > 0.f)) return -1;
    const int64_t n_tok = B * T;
    if (n_tok * H == 0) return 0;
    const int grid = (int)(n_tok < 65536 ? n_tok : 65536);
    hipLaunchKernelGGL(rope_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (uint4*)qkv, (const float4*)cos_t,
                       (const float4*)sin_t, n_tok, T, (int)H, (int)hd, q_scale);
    return evo_launch_status();
}

// ---- decode step: rotary on q and k at each row's own position + append of (k, v) to the KV cache, one launch.
// The separate path built a [B, hd/2] cos / sin table with six small ATen launches per token, ran rope_kernel and then an
// indexed copy per attention layer (~45 us of launch-bound kernels per token).  Same arithmetic: angle = (p / scaling) *
// inv_freq in fp32, cosf / sinf, rounded to bf16 values like flash-attn's cached table, then rope_kernel's expression.
__global__ __launch_bounds__(256) void rope_append_decode_kernel(uint4* __restrict__ qkv, uint4* __restrict__ kv,
                                                                 const int64_t* __restrict__ pos,
                                                                 const float* __restrict__ inv_freq, float scaling, int B,
                                                                 int H, int hd, int64_t kv_sb, int64_t kv_st, int64_t kv_sw,
                                                                 int64_t kv_sh, float q_scale) {   // kv strides in 16-byte vectors
    const int half_vec = hd / 16;
    const int total = B * H * half_vec;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = i % half_vec;
    const int h = (i / half_vec) % H;
    const int b = i / (half_vec * H);
    const int64_t p = pos[b];
    float t = (float)p;
    if (scaling != 1.0f) t = t / scaling;
    float co[8], si[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float f = t * inv_freq[c * 8 + e];
        co[e] = round_bf(cosf(f));
        si[e] = round_bf(sinf(f));
    }
    const int rv = hd / 8;                                   // vectors per head row
    uint4* krow = kv + b * kv_sb + p * kv_st + h * kv_sh;
    uint4* vrow = krow + kv_sw;
#pragma unroll
    for (int which = 0; which < 2; ++which) {                // q, k
        const int64_t row_vec = (((int64_t)b * 3 + which) * H + h) * rv;
        float x0[8], x1[8], o0[8], o1[8];
        unpack8(qkv[row_vec + c], x0);
        unpack8(qkv[row_vec + half_vec + c], x1);
        const float qs = which == 0 ? q_scale : 1.0f;         // (see rope_kernel)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o0[e] = (x0[e] * co[e] - x1[e] * si[e]) * qs;
            o1[e] = (x0[e] * si[e] + x1[e] * co[e]) * qs;
        }
        const uint4 r0 = pack8(o0), r1 = pack8(o1);
        qkv[row_vec + c] = r0;
        qkv[row_vec + half_vec + c] = r1;
        if (which == 1) { krow[c] = r0; krow[half_vec + c] = r1; }
    }
    const int64_t vvec = (((int64_t)b * 3 + 2) * H + h) * rv;
    vrow[c] = qkv[vvec + c];
    vrow[half_vec + c] = qkv[vvec + half_vec + c];
}

extern "C" int evo_rope_append_decode_bf16(void* qkv, void* kv, const int64_t* pos, const float* inv_freq, float scaling,
                                           int64_t B, int64_t H, int64_t hd, int64_t kv_sb, int64_t kv_st, int64_t kv_sw,
                                           int64_t kv_sh, float q_scale, void* stream) {
    if (B <= 0 || H <= 0 || hd <= 0 || hd % 16 != 0 || !qkv || !kv || !pos || !inv_freq || scaling <= 0.f || !(q_scale > 0.f)) return -1;
    if ((kv_sb % 8) || (kv_st % 8) || (kv_sw % 8) || (kv_sh % 8) || B * H * (hd / 16) > 0x7fffffff) return -1;
    const int total = (int)(B * H * (hd / 16));
    hipLaunchKernelGGL(rope_append_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (uint4*)qkv, (uint4*)kv, pos, inv_freq, scaling, (int)B, (int)H, (int)hd, kv_sb / 8, kv_st / 8, kv_sw / 8,
                       kv_sh / 8, q_scale);
    return evo_launch_status();
}

// ------------------------------------------------------------------------------------------- gelu gate
__global__ __launch_bounds__(128) void gelu_gate_kernel(const uint4* __restrict__ g, uint4* __restrict__ a, int64_t M,
                                                        int ivec) {
    // block = 128 consecutive 16-byte columns of one row; rows are strided over gridDim.y (no per-element division)
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= ivec) return;
    for (int64_t row = blockIdx.y; row < M; row += gridDim.y) {
        const uint4* gr = g + row * 2 * ivec;
        float u[8], w[8], o[8];
        unpack8(gr[c], u);
        unpack8(gr[ivec + c], w);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2_t uu = {u[e], u[e + 1]}, ww = {w[e], w[e + 1]};
            const f32x2_t oo = gelu_gate2(uu, ww);
            o[e] = oo[0]; o[e + 1] = oo[1];
        }
        a[row * ivec + c] = pack8(o);
    }
}

extern "C" int evo_gelu_gate_bf16(const void* g, void* a, int64_t M, int64_t I, void* stream) {
    if (I % 8 != 0 || M < 0) return -1;
    if (M * I == 0) return 0;
    const int ivec = (int)(I / 8);
    const unsigned gy = (unsigned)(M < 16384 ? M : 16384);
    hipLaunchKernelGGL(gelu_gate_kernel, dim3((unsigned)((ivec + 127) / 128), gy), dim3(128), 0, (hipStream_t)stream,
                       (const uint4*)g, (uint4*)a, M, ivec);
    return evo_launch_status();
}

// ------------------------------------------------------------------------------------------- scoring tail
// one wave per row: fp32 log-softmax, gather of the target's log-prob, entropy -sum p log p.
// Logits are bf16 (model output) or f32 (the generation loop's score buffer).
template <bool F32>
__device__ __forceinline__ float logit_at(const void* row, int64_t j) {
    return F32 ? ((const float*)row)[j] : bf_to_f(((const uint16_t*)row)[j]);
}

template <bool F32>
__global__ __launch_bounds__(256) void logprob_entropy_kernel(const void* __restrict__ logits,
                                                              const int64_t* __restrict__ target,
                                                              float* __restrict__ logprob, float* __restrict__ entropy,
                                                              int64_t M, int V) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nvec = V / 8;                       // 8 logits per lane per step
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        const char* lr = (const char*)logits + row * (int64_t)V * (F32 ? 4 : 2);
        float m = -INFINITY;
        for (int idx = lane; idx < nvec; idx += 64) {
            float f[8];
            if (F32) {
                const float4 a = ((const float4*)lr)[2 * idx], b = ((const float4*)lr)[2 * idx + 1];
                f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
            } else {
                unpack8(((const uint4*)lr)[idx], f);
            }
            for (int e = 0; e < 8; ++e) m = fmaxf(m, f[e]);
        }
        m = wave_max(m);
        float s1 = 0.f, s2 = 0.f;
        for (int idx = lane; idx < nvec; idx += 64) {
            float f[8];
            if (F32) {
                const float4 a = ((const float4*)lr)[2 * idx], b = ((const float4*)lr)[2 * idx + 1];
                f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
            } else {
                unpack8(((const uint4*)lr)[idx], f);
            }
            for (int e = 0; e < 8; ++e) {
                float d = f[e] - m;
                float ex = __expf(d);
                s1 += ex;
                s2 = fmaf(ex, d, s2);
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const float logz = logf(s1);
        if (lane == 0) {
            if (entropy) entropy[row] = logz - s2 / s1;
            if (logprob) {
                int64_t tg = target ? target[row] : -1;
                logprob[row] = (tg >= 0 && tg < V) ? logit_at<F32>(lr, tg) - m - logz : 0.f;
            }
        }
    }
}

extern "C" int evo_logprob_entropy(const void* logits, int64_t logits_f32, const int64_t* target, float* logprob,
                                   float* entropy, int64_t M, int64_t V, void* stream) {
    if (V % 8 != 0 || M < 0) return -1;
    if (M == 0) return 0;
    int64_t blocks = (M + 3) / 4;
    int grid = (int)(blocks < 16384 ? blocks : 16384);
    if (logits_f32)
        hipLaunchKernelGGL(logprob_entropy_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target,
                           logprob, entropy, M, (int)V);
    else
        hipLaunchKernelGGL(logprob_entropy_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits,
                           target, logprob, entropy, M, (int)V);
    return evo_launch_status();
}

extern "C" int evo_abi_version(void) { return EVO_ABI_VERSION; }
