// Hyena operator for gfx950: short causal FIR (k=3) + bias + column split + x1*v gate + long
// convolution + (y + x1v*D)*x2 epilogue, from the projections output z [B,T,3D] straight to y [B,T,D].
//
// Long convolution: the filter is h_k = Re sum_{s<8} R_s p_s^k, so y = h (*) x1v is evaluated exactly
// through the 8 complex modes   S_t = p S_{t-1} + x1v_t ,  y_t = Re sum_s R_s S_t   (fp32).  Time is cut
// into segments of C steps; one wave owns (batch b, head h, segment k) and its 64 lanes own the head's
// 128 channels two at a time, so every global access of a wave is one contiguous 256-byte row piece:
//   launch 1  seg_state : segment end state from a zero start (reads the x1,v thirds of z)
//   launch 2  carry_scan: exclusive scan over segments with p^C (fp64 powers) -> state entering each
//   launch 3  apply     : the full recurrence from the entering state + FIR + gates, writes y
// Algorithmic HBM bytes per token per layer: 3D*2 in + D*2 out = 32,768 B (D = 4096); this 3-launch
// form moves 49,152 B (the x1,v thirds are read twice) plus 128*D*8/C bytes of segment states.
// Entry points and reference citations: include/evo_mi355x.h.
#include "common.h"
#include "../../include/evo_mi355x.h"

#define NS 8            // state_size   [REF evo/configs/evo-1-8k-base_inference.yml:14]
#define HD 128          // channels per head (hidden_size / num_attention_heads)  [REF yml:2,9]
#define UNR 4           // time steps per software-pipelined group

struct RowPtr {         // the three 256-byte pieces (x2 | x1 | v) of one head in one z row, as dwords
    const uint32_t* p;
};

// per-lane constants of one head-slice: 2 channels (lo, hi) of each of the 3 groups
struct FirCoef {
    float w[3][2][3];   // [group][lo/hi][tap]
    float b[3][2];
};

__device__ __forceinline__ void load_fir(FirCoef& fc, const uint16_t* __restrict__ fir_w,
                                         const uint16_t* __restrict__ fir_b, int c_base, int lane, int g_first) {
#pragma unroll
    for (int g = g_first; g < 3; ++g) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int c = c_base + g * HD + 2 * lane + e;
#pragma unroll
            for (int k = 0; k < 3; ++k) fc.w[g][e][k] = bf_to_f(fir_w[c * 3 + k]);
            fc.b[g][e] = bf_to_f(fir_b[c]);
        }
    }
}

// raw z dword of group g at (relative) time t for this lane; t < 0 reads the halo (or zero)
__device__ __forceinline__ uint32_t load_hist(const uint32_t* __restrict__ zrow0, const uint32_t* __restrict__ halo,
                                              int64_t t_abs, int64_t rowdw, int col) {
    if (t_abs >= 0) return zrow0[t_abs * rowdw + col];
    if (halo) return halo[(t_abs + 2) * rowdw + col];
    return 0u;
}

// ------------------------------------------------------------------------------------------------ launch 1
__global__ __launch_bounds__(256, 2) void hyena_seg_state_kernel(
    const uint32_t* __restrict__ z, const uint32_t* __restrict__ z_halo, const uint16_t* __restrict__ fir_w,
    const uint16_t* __restrict__ fir_b, const float* __restrict__ poles, float* __restrict__ agg, int B, int64_t T,
    int D, int H, int C, int n_seg) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform (SGPR)
    const int64_t total = (int64_t)B * n_seg * H;
    if (gw >= total) return;
    const int h = (int)(gw % H);
    const int seg = (int)((gw / H) % n_seg);
    const int b = (int)(gw / ((int64_t)H * n_seg));
    const int64_t rowdw = 3 * (int64_t)D / 2;
    const int64_t t0 = (int64_t)seg * C;
    const int64_t t1 = (t0 + C < T) ? t0 + C : T;

    FirCoef fc;
    load_fir(fc, fir_w, fir_b, h * 3 * HD, lane, 1);

    float pr[2][NS], pi[2][NS], sr[2][NS], si[2][NS];
    {
        const float* pp = poles + ((int64_t)(h * HD + 2 * lane)) * NS * 2;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                pr[e][s] = pp[(e * NS + s) * 2];
                pi[e][s] = pp[(e * NS + s) * 2 + 1];
                sr[e][s] = 0.f;
                si[e][s] = 0.f;
            }
    }

    const uint32_t* zb = z + (int64_t)b * T * rowdw;
    const uint32_t* hb = z_halo ? z_halo + (int64_t)b * 2 * rowdw : nullptr;
    const int col1 = (h * 3 * HD + HD) / 2 + lane;       // x1 third
    const int col2 = (h * 3 * HD + 2 * HD) / 2 + lane;   // v third

    // history z[t-2], z[t-1] (unpacked) for groups x1 (index 0) and v (index 1)
    float m2[2][2], m1[2][2];
    {
        uint32_t a = load_hist(zb, hb, t0 - 2, rowdw, col1), c = load_hist(zb, hb, t0 - 2, rowdw, col2);
        m2[0][0] = bf_lo(a); m2[0][1] = bf_hi(a); m2[1][0] = bf_lo(c); m2[1][1] = bf_hi(c);
        a = load_hist(zb, hb, t0 - 1, rowdw, col1); c = load_hist(zb, hb, t0 - 1, rowdw, col2);
        m1[0][0] = bf_lo(a); m1[0][1] = bf_hi(a); m1[1][0] = bf_lo(c); m1[1][1] = bf_hi(c);
    }

    auto step = [&](uint32_t zx1, uint32_t zv) {
        float c0[2][2];
        c0[0][0] = bf_lo(zx1); c0[0][1] = bf_hi(zx1); c0[1][0] = bf_lo(zv); c0[1][1] = bf_hi(zv);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float x1c = fmaf(fc.w[1][e][2], c0[0][e], fmaf(fc.w[1][e][1], m1[0][e], fmaf(fc.w[1][e][0], m2[0][e], fc.b[1][e])));
            float vc = fmaf(fc.w[2][e][2], c0[1][e], fmaf(fc.w[2][e][1], m1[1][e], fmaf(fc.w[2][e][0], m2[1][e], fc.b[2][e])));
            float x = x1c * vc;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float nr = fmaf(pr[e][s], sr[e][s], fmaf(-pi[e][s], si[e][s], x));
                float ni = fmaf(pr[e][s], si[e][s], pi[e][s] * sr[e][s]);
                sr[e][s] = nr;
                si[e][s] = ni;
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 2; ++e) { m2[g][e] = m1[g][e]; m1[g][e] = c0[g][e]; }
    };

    // ring prefetch: the raw rows of steps t..t+UNR-1 are in flight while step t-1.. computes
    const uint32_t* zp = zb + t0 * rowdw;
    const int n_full = (int)((t1 - t0) / UNR);
    uint32_t ring[UNR][2];
    if (n_full > 0) {
#pragma unroll
        for (int k = 0; k < UNR; ++k) { ring[k][0] = zp[k * rowdw + col1]; ring[k][1] = zp[k * rowdw + col2]; }
    }
    for (int g = 0; g < n_full; ++g) {
        const uint32_t* zn = zp + UNR * rowdw;
        const bool more = g + 1 < n_full;
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const uint32_t a = ring[k][0], c = ring[k][1];
            if (more) { ring[k][0] = zn[k * rowdw + col1]; ring[k][1] = zn[k * rowdw + col2]; }
            step(a, c);
        }
        zp = zn;
    }
    for (int64_t t = t0 + (int64_t)n_full * UNR; t < t1; ++t) {
        step(zp[col1], zp[col2]);
        zp += rowdw;
    }

    float4* out = (float4*)(agg + ((((int64_t)b * n_seg + seg) * D + h * HD + 2 * lane) * NS) * 2);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int s = 0; s < NS; s += 2)
            out[(e * NS + s) / 2] = make_float4(sr[e][s], si[e][s], sr[e][s + 1], si[e][s + 1]);
}

// ------------------------------------------------------------------------------------------------ launch 2
// one thread per (b, d, s): in-place exclusive scan over segments.  p^C in fp64 (tiny work, keeps the
// carry exact to fp32 over hundreds of segments).
__device__ __forceinline__ void cpow_int(double pr, double pi, int64_t n, double& or_, double& oi) {
    double rr = 1.0, ri = 0.0;
    while (n > 0) {
        if (n & 1) { double t = rr * pr - ri * pi; ri = rr * pi + ri * pr; rr = t; }
        double t = pr * pr - pi * pi; pi = 2.0 * pr * pi; pr = t;
        n >>= 1;
    }
    or_ = rr; oi = ri;
}

#define SCAN_UNR 8
__global__ __launch_bounds__(256) void hyena_carry_scan_kernel(float2* __restrict__ agg, const float2* __restrict__ poles,
                                                               const float2* __restrict__ s0, float2* __restrict__ s_final,
                                                               int B, int64_t T, int D, int C, int n_seg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // over B * D * NS
    const int64_t per_b = (int64_t)D * NS;
    if (i >= (int64_t)B * per_b) return;
    const int b = (int)(i / per_b);
    const int64_t ds = i - (int64_t)b * per_b;
    const float2 p = poles[ds];
    double pcr, pci;
    cpow_int((double)p.x, (double)p.y, C, pcr, pci);
    double rr = 0.0, ri = 0.0;
    if (s0) { float2 v = s0[i]; rr = v.x; ri = v.y; }
    float2* a = agg + (int64_t)b * n_seg * per_b + ds;
    const int n_lead = n_seg - 1;                                  // segments of full length C
    int k = 0;
    for (; k + SCAN_UNR <= n_lead; k += SCAN_UNR) {
        float2 v[SCAN_UNR];
#pragma unroll
        for (int u = 0; u < SCAN_UNR; ++u) v[u] = a[(int64_t)(k + u) * per_b];
#pragma unroll
        for (int u = 0; u < SCAN_UNR; ++u) {
            a[(int64_t)(k + u) * per_b] = make_float2((float)rr, (float)ri);
            double nr = pcr * rr - pci * ri + (double)v[u].x;
            ri = pcr * ri + pci * rr + (double)v[u].y;
            rr = nr;
        }
    }
    for (; k < n_lead; ++k) {
        float2 v = a[(int64_t)k * per_b];
        a[(int64_t)k * per_b] = make_float2((float)rr, (float)ri);
        double nr = pcr * rr - pci * ri + (double)v.x;
        ri = pcr * ri + pci * rr + (double)v.y;
        rr = nr;
    }
    // last (possibly ragged) segment
    {
        float2 v = a[(int64_t)n_lead * per_b];
        a[(int64_t)n_lead * per_b] = make_float2((float)rr, (float)ri);
        if (s_final) {
            int64_t cl = T - (int64_t)n_lead * C;
            double qr, qi;
            cpow_int((double)p.x, (double)p.y, cl, qr, qi);
            double nr = qr * rr - qi * ri + (double)v.x;
            ri = qr * ri + qi * rr + (double)v.y;
            rr = nr;
            s_final[i] = make_float2((float)rr, (float)ri);
        }
    }
}

// sequence-parallel fix-up: `agg` already holds the entering states for a ZERO carry-in; add the
// contribution p^(k*C) * s0 of the state s0 that enters this shard (known only after the all-gather).
__global__ __launch_bounds__(256) void hyena_carry_add_kernel(float2* __restrict__ agg, const float2* __restrict__ poles,
                                                              const float2* __restrict__ s0, int B, int D, int C,
                                                              int n_seg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_b = (int64_t)D * NS;
    if (i >= (int64_t)B * per_b) return;
    const int b = (int)(i / per_b);
    const int64_t ds = i - (int64_t)b * per_b;
    const float2 p = poles[ds];
    double pcr, pci;
    cpow_int((double)p.x, (double)p.y, C, pcr, pci);
    const float2 v0 = s0[i];
    double cr = v0.x, ci = v0.y;
    float2* a = agg + (int64_t)b * n_seg * per_b + ds;
    for (int k = 0; k < n_seg; ++k) {
        float2 v = a[(int64_t)k * per_b];
        a[(int64_t)k * per_b] = make_float2(v.x + (float)cr, v.y + (float)ci);
        double nr = pcr * cr - pci * ci;
        ci = pcr * ci + pci * cr;
        cr = nr;
    }
}

// ------------------------------------------------------------------------------------------------ launch 3
__global__ __launch_bounds__(256, 2) void hyena_apply_kernel(
    const uint32_t* __restrict__ z, const uint32_t* __restrict__ z_halo, const uint16_t* __restrict__ fir_w,
    const uint16_t* __restrict__ fir_b, const float* __restrict__ poles, const float* __restrict__ residues,
    const uint16_t* __restrict__ dskip, const float* __restrict__ agg, uint32_t* __restrict__ y, int B, int64_t T, int D,
    int H, int C, int n_seg) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform (SGPR)
    const int64_t total = (int64_t)B * n_seg * H;
    if (gw >= total) return;
    const int h = (int)(gw % H);
    const int seg = (int)((gw / H) % n_seg);
    const int b = (int)(gw / ((int64_t)H * n_seg));
    const int64_t rowdw = 3 * (int64_t)D / 2;
    const int64_t t0 = (int64_t)seg * C;
    const int64_t t1 = (t0 + C < T) ? t0 + C : T;

    FirCoef fc;
    load_fir(fc, fir_w, fir_b, h * 3 * HD, lane, 0);

    float pr[2][NS], pi[2][NS], rr[2][NS], ri[2][NS], sr[2][NS], si[2][NS], dk[2];
    {
        const int64_t d0 = h * HD + 2 * lane;
        const float* pp = poles + d0 * NS * 2;
        const float* rp = residues + d0 * NS * 2;
        const float4* sp = (const float4*)(agg + ((((int64_t)b * n_seg + seg) * D + d0) * NS) * 2);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                pr[e][s] = pp[(e * NS + s) * 2];
                pi[e][s] = pp[(e * NS + s) * 2 + 1];
                rr[e][s] = rp[(e * NS + s) * 2];
                ri[e][s] = rp[(e * NS + s) * 2 + 1];
            }
#pragma unroll
            for (int s = 0; s < NS; s += 2) {
                float4 v = sp[(e * NS + s) / 2];
                sr[e][s] = v.x; si[e][s] = v.y; sr[e][s + 1] = v.z; si[e][s + 1] = v.w;
            }
            dk[e] = bf_to_f(dskip[d0 + e]);
        }
    }

    const uint32_t* zb = z + (int64_t)b * T * rowdw;
    const uint32_t* hb = z_halo ? z_halo + (int64_t)b * 2 * rowdw : nullptr;
    const int col0 = (h * 3 * HD) / 2 + lane;     // x2 third; x1 = +HD/2 dwords, v = +HD dwords

    float m2[3][2], m1[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        uint32_t a = load_hist(zb, hb, t0 - 2, rowdw, col0 + g * (HD / 2));
        m2[g][0] = bf_lo(a); m2[g][1] = bf_hi(a);
        a = load_hist(zb, hb, t0 - 1, rowdw, col0 + g * (HD / 2));
        m1[g][0] = bf_lo(a); m1[g][1] = bf_hi(a);
    }

    auto step = [&](const uint32_t (&zr)[3]) -> uint32_t {
        float c0[3][2], out[2];
#pragma unroll
        for (int g = 0; g < 3; ++g) { c0[g][0] = bf_lo(zr[g]); c0[g][1] = bf_hi(zr[g]); }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float f[3];
#pragma unroll
            for (int g = 0; g < 3; ++g)
                f[g] = fmaf(fc.w[g][e][2], c0[g][e], fmaf(fc.w[g][e][1], m1[g][e], fmaf(fc.w[g][e][0], m2[g][e], fc.b[g][e])));
            const float x = f[1] * f[2];           // x1 * v
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float nr = fmaf(pr[e][s], sr[e][s], fmaf(-pi[e][s], si[e][s], x));
                float ni = fmaf(pr[e][s], si[e][s], pi[e][s] * sr[e][s]);
                sr[e][s] = nr;
                si[e][s] = ni;
                acc = fmaf(rr[e][s], nr, fmaf(-ri[e][s], ni, acc));
            }
            out[e] = fmaf(x, dk[e], acc) * f[0];   // (y + x1v * D) * x2
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int e = 0; e < 2; ++e) { m2[g][e] = m1[g][e]; m1[g][e] = c0[g][e]; }
        return pack_bf2(out[0], out[1]);
    };

    const uint32_t* zp = zb + t0 * rowdw + col0;
    uint32_t* yp = y + ((int64_t)b * T + t0) * (D / 2) + h * (HD / 2) + lane;
    const int ydw = D / 2;
    const int n_full = (int)((t1 - t0) / UNR);
    uint32_t ring[UNR][3];
    if (n_full > 0) {
#pragma unroll
        for (int k = 0; k < UNR; ++k)
#pragma unroll
            for (int g = 0; g < 3; ++g) ring[k][g] = zp[k * rowdw + g * (HD / 2)];
    }
    for (int gi = 0; gi < n_full; ++gi) {
        const uint32_t* zn = zp + UNR * rowdw;
        const bool more = gi + 1 < n_full;
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            uint32_t zr[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) zr[g] = ring[k][g];
            if (more) {
#pragma unroll
                for (int g = 0; g < 3; ++g) ring[k][g] = zn[k * rowdw + g * (HD / 2)];
            }
            yp[(int64_t)k * ydw] = step(zr);
        }
        zp = zn;
        yp += (int64_t)UNR * ydw;
    }
    for (int64_t t = t0 + (int64_t)n_full * UNR; t < t1; ++t) {
        uint32_t zr[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) zr[g] = zp[g * (HD / 2)];
        *yp = step(zr);
        zp += rowdw;
        yp += ydw;
    }
}

// ------------------------------------------------------------------------------------------------ decode step
// one thread per (b, channel pair): FIR step with the 2-sample history, roll, modal update, gate.
__global__ __launch_bounds__(256) void hyena_step_kernel(
    const uint32_t* __restrict__ z_t, uint16_t* __restrict__ fir_state, float* __restrict__ iir_state,
    const uint16_t* __restrict__ fir_w, const uint16_t* __restrict__ fir_b, const float* __restrict__ poles,
    const float* __restrict__ residues, const uint16_t* __restrict__ dskip, uint32_t* __restrict__ y, int B, int D,
    int H) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over B * D/2
    if (i >= (int64_t)B * (D / 2)) return;
    const int b = (int)(i / (D / 2));
    const int dp = (int)(i - (int64_t)b * (D / 2));                  // channel pair index within D
    const int h = dp / (HD / 2), j2 = dp - h * (HD / 2);
    float f[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int cdw = (h * 3 * HD + g * HD) / 2 + j2;              // dword column in the 3D row
        const uint32_t zr = z_t[(int64_t)b * (3 * D / 2) + cdw];
        const float zc[2] = {bf_lo(zr), bf_hi(zr)};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * cdw + e;
            uint16_t* fs = fir_state + ((int64_t)b * 3 * D + c) * 2;
            const float o0 = bf_to_f(fs[0]), o1 = bf_to_f(fs[1]);
            const uint16_t zraw = (uint16_t)(e ? (zr >> 16) : (zr & 0xffffu));
            f[g][e] = fmaf(bf_to_f(fir_w[c * 3 + 2]), zc[e],
                           fmaf(bf_to_f(fir_w[c * 3 + 1]), o1, fmaf(bf_to_f(fir_w[c * 3]), o0, bf_to_f(fir_b[c]))));
            fs[0] = fs[1];
            fs[1] = zraw;
        }
    }
    float out[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int d = 2 * dp + e;
        const float x = f[1][e] * f[2][e];
        float2* st = (float2*)iir_state + ((int64_t)b * D + d) * NS;
        const float2* pp = (const float2*)poles + (int64_t)d * NS;
        const float2* rp = (const float2*)residues + (int64_t)d * NS;
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float2 p = pp[s], r = rp[s], sv = st[s];
            const float nr = fmaf(p.x, sv.x, fmaf(-p.y, sv.y, x));
            const float ni = fmaf(p.x, sv.y, p.y * sv.x);
            st[s] = make_float2(nr, ni);
            acc = fmaf(r.x, nr, fmaf(-r.y, ni, acc));
        }
        out[e] = fmaf(x, bf_to_f(dskip[d]), acc) * f[0][e];
    }
    y[i] = pack_bf2(out[0], out[1]);
}

// ------------------------------------------------------------------------------------------------ C ABI
static int hyena_check(int64_t B, int64_t T, int64_t D, int64_t n_heads, int64_t seg_len) {
    if (B <= 0 || T <= 0 || D <= 0 || n_heads <= 0) return -1;
    if (D != n_heads * HD) return -1;                 // one wave per 128-channel head
    if (seg_len <= 0 || seg_len % UNR != 0) return -1;
    return 0;
}

extern "C" int evo_hyena_seg_state(const void* z, const void* z_halo, const void* fir_w, const void* fir_b,
                                   const float* poles, float* agg, int64_t B, int64_t T, int64_t D, int64_t n_heads,
                                   int64_t seg_len, void* stream) {
    if (hyena_check(B, T, D, n_heads, seg_len)) return -1;
    const int n_seg = (int)((T + seg_len - 1) / seg_len);
    const int64_t waves = B * n_seg * n_heads;
    hipLaunchKernelGGL(hyena_seg_state_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)z, (const uint32_t*)z_halo, (const uint16_t*)fir_w, (const uint16_t*)fir_b, poles,
                       agg, (int)B, T, (int)D, (int)n_heads, (int)seg_len, n_seg);
    return evo_launch_status();
}

extern "C" int evo_hyena_carry_scan(float* agg, const float* poles, const float* s0, float* s_final, int64_t B,
                                    int64_t T, int64_t D, int64_t seg_len, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || seg_len <= 0) return -1;
    const int n_seg = (int)((T + seg_len - 1) / seg_len);
    const int64_t n = B * D * NS;
    hipLaunchKernelGGL(hyena_carry_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float2*)agg, (const float2*)poles, (const float2*)s0, (float2*)s_final, (int)B, T, (int)D,
                       (int)seg_len, n_seg);
    return evo_launch_status();
}

extern "C" int evo_hyena_carry_add(float* agg, const float* poles, const float* s0, int64_t B, int64_t T, int64_t D,
                                   int64_t seg_len, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || seg_len <= 0 || !s0) return -1;
    const int n_seg = (int)((T + seg_len - 1) / seg_len);
    const int64_t n = B * D * NS;
    hipLaunchKernelGGL(hyena_carry_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float2*)agg, (const float2*)poles, (const float2*)s0, (int)B, (int)D, (int)seg_len, n_seg);
    return evo_launch_status();
}

extern "C" int evo_hyena_apply(const void* z, const void* z_halo, const void* fir_w, const void* fir_b,
                               const float* poles, const float* residues, const void* dskip, const float* agg, void* y,
                               int64_t B, int64_t T, int64_t D, int64_t n_heads, int64_t seg_len, void* stream) {
    if (hyena_check(B, T, D, n_heads, seg_len)) return -1;
    const int n_seg = (int)((T + seg_len - 1) / seg_len);
    const int64_t waves = B * n_seg * n_heads;
    hipLaunchKernelGGL(hyena_apply_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)z, (const uint32_t*)z_halo, (const uint16_t*)fir_w, (const uint16_t*)fir_b, poles,
                       residues, (const uint16_t*)dskip, agg, (uint32_t*)y, (int)B, T, (int)D, (int)n_heads, (int)seg_len,
                       n_seg);
    return evo_launch_status();
}

extern "C" int evo_hyena_step(const void* z_t, void* fir_state, float* iir_state, const void* fir_w, const void* fir_b,
                              const float* poles, const float* residues, const void* dskip, void* y, int64_t B,
                              int64_t D, int64_t n_heads, void* stream) {
    if (B <= 0 || D != n_heads * HD) return -1;
    const int64_t n = B * (D / 2);
    hipLaunchKernelGGL(hyena_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)z_t, (uint16_t*)fir_state, iir_state, (const uint16_t*)fir_w,
                       (const uint16_t*)fir_b, poles, residues, (const uint16_t*)dskip, (uint32_t*)y, (int)B, (int)D,
                       (int)n_heads);
    return evo_launch_status();
}
