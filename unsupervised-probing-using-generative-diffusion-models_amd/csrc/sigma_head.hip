// sigma head g(x) for gfx950: sliding-window gather + scaler + trailing variance + 3-layer MLP.
//
// Follows wv_sigma_trailing (reference NsDiff/sigma.py:34-72) and SigmaEstimation.forward
// (NsDiff/g_backbone.py:49-72).  Only the last T-k trailing variances feed the MLP, and for
// those the replicate padding never enters the window, so v[i] = Var_pop(x[i+1 .. i+k]),
// i = 0..T-k-1.  torch's CPU variance accumulates in double (Welford); so do we (two-pass).
//
// This kernel runs once per window (<1% of the path's time); it is HBM/L2-bound on the
// 1.6 MB of MLP weights, which every workgroup streams once for GROUPS (window,node) rows
// from the transposed [in][out] copy (coalesced across the threads of a workgroup).
#include "mpv_common.h"

namespace {

constexpr int SG_THREADS = 256;
constexpr int GROUPS = 4;   // (window,node) rows per workgroup

struct SigmaArgs {
    const float *w0t, *b0, *ln0w, *ln0b, *w1t, *b1, *ln1w, *ln1b, *w2t, *b2;
    const float* series;   // [V][L][F]
    const float* mean;     // [F] or null
    const float* std;      // [F] or null
    float* gx;             // [n_windows][V][O][F]
    long long L, first, step;
    long long n_groups;    // n_windows * V
    int V, Tw, k, K0, O, F, H;
    float add_eps;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// sum over the whole workgroup; every thread gets the result.  red: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < SG_THREADS / 64; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float softplus_acc(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// out[g][f][j] = act(bias[j] + sum_k in[g][f][k] * Wt[k][j])   for j < N
template <bool RELU>
__device__ __forceinline__ void dense(const float* __restrict__ Wt, const float* __restrict__ bias, const float* in,
                                      int in_stride_f, int in_stride_g, float* out, int out_stride_f, int out_stride_g,
                                      int K, int N, int F, int ng) {
    for (int j = threadIdx.x; j < N; j += SG_THREADS) {
        float acc[GROUPS][MPV_MAX_NF];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int f = 0; f < MPV_MAX_NF; ++f) acc[g][f] = bias[j];
        for (int k = 0; k < K; ++k) {
            const float w = Wt[(long long)k * N + j];
#pragma unroll
            for (int g = 0; g < GROUPS; ++g)
#pragma unroll
                for (int f = 0; f < MPV_MAX_NF; ++f)
                    if (g < ng && f < F) acc[g][f] = __builtin_fmaf(in[g * in_stride_g + f * in_stride_f + k], w, acc[g][f]);
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int f = 0; f < MPV_MAX_NF; ++f)
                if (g < ng && f < F) {
                    const float v = acc[g][f];
                    out[g * out_stride_g + f * out_stride_f + j] = RELU ? fmaxf(v, 0.0f) : v;
                }
    }
}

// LayerNorm over the [F][H] slab of each group, in place (eps = 1e-5, biased variance)
__device__ __forceinline__ void layer_norm(float* h, int stride_g, const float* __restrict__ w, const float* __restrict__ b,
                                           int F, int H, int ng, double* red) {
    const int n = F * H;
    for (int g = 0; g < ng; ++g) {
        float* x = h + g * stride_g;
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += SG_THREADS) s += (double)x[i];
        const double mean = block_sum(s, red) / n;
        double q = 0.0;
        for (int i = threadIdx.x; i < n; i += SG_THREADS) {
            const double d = (double)x[i] - mean;
            q += d * d;
        }
        const double var = block_sum(q, red) / n;
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        const float fmean = (float)mean;
        for (int i = threadIdx.x; i < n; i += SG_THREADS) x[i] = (x[i] - fmean) * rstd * w[i] + b[i];
    }
    __syncthreads();
}

__global__ __launch_bounds__(SG_THREADS) void sigma_head_kernel(const SigmaArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* red = reinterpret_cast<double*>(smem);                 // 8 doubles
    float* xw = reinterpret_cast<float*>(smem + 64);              // [GROUPS][F][Tw]
    float* v = xw + GROUPS * p.F * p.Tw;                           // [GROUPS][F][K0]
    float* h1 = v + GROUPS * p.F * p.K0;                           // [GROUPS][F][H]
    float* h2 = h1 + GROUPS * p.F * p.H;                           // [GROUPS][F][H]

    const long long g0 = (long long)blockIdx.x * GROUPS;
    const long long left = p.n_groups - g0;
    const int ng = (int)(left < GROUPS ? left : GROUPS);

    // ---- gather + scale the windows: xw[g][f][t] --------------------------------------
    for (int i = threadIdx.x; i < ng * p.Tw * p.F; i += SG_THREADS) {
        const int f = i % p.F;
        const int t = (i / p.F) % p.Tw;
        const int g = i / (p.F * p.Tw);
        const long long gi = g0 + g;
        const long long w = gi / p.V;
        const int node = (int)(gi % p.V);
        float x = p.series[((long long)node * p.L + p.first + w * p.step + t) * p.F + f];
        if (p.mean) x = (x - p.mean[f]) / p.std[f];
        xw[(g * p.F + f) * p.Tw + t] = x;
    }
    __syncthreads();
    // ---- trailing population variance of the last K0 positions, + 10e-8 ----------------
    for (int i = threadIdx.x; i < ng * p.F * p.K0; i += SG_THREADS) {
        const int pos = i % p.K0;
        const float* x = xw + (i / p.K0) * p.Tw + pos + 1;
        double s = 0.0;
        for (int j = 0; j < p.k; ++j) s += (double)x[j];
        const double m = s / p.k;
        double q = 0.0;
        for (int j = 0; j < p.k; ++j) {
            const double d = (double)x[j] - m;
            q += d * d;
        }
        v[i] = (float)(q / p.k) + 10e-8f;
    }
    __syncthreads();
    // ---- MLP --------------------------------------------------------------------------
    dense<true>(p.w0t, p.b0, v, p.K0, p.F * p.K0, h1, p.H, p.F * p.H, p.K0, p.H, p.F, ng);
    __syncthreads();
    layer_norm(h1, p.F * p.H, p.ln0w, p.ln0b, p.F, p.H, ng, red);
    dense<true>(p.w1t, p.b1, h1, p.H, p.F * p.H, h2, p.H, p.F * p.H, p.H, p.H, p.F, ng);
    __syncthreads();
    layer_norm(h2, p.F * p.H, p.ln1w, p.ln1b, p.F, p.H, ng, red);
    // last layer straight to HBM: gx[gi][o][f] = softplus(.) + add_eps
    for (int j = threadIdx.x; j < p.O; j += SG_THREADS) {
        float acc[GROUPS][MPV_MAX_NF];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int f = 0; f < MPV_MAX_NF; ++f) acc[g][f] = p.b2[j];
        for (int k = 0; k < p.H; ++k) {
            const float w = p.w2t[(long long)k * p.O + j];
#pragma unroll
            for (int g = 0; g < GROUPS; ++g)
#pragma unroll
                for (int f = 0; f < MPV_MAX_NF; ++f)
                    if (g < ng && f < p.F) acc[g][f] = __builtin_fmaf(h2[(g * p.F + f) * p.H + k], w, acc[g][f]);
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int f = 0; f < MPV_MAX_NF; ++f)
                if (g < ng && f < p.F) p.gx[((g0 + g) * p.O + j) * p.F + f] = softplus_acc(acc[g][f]) + p.add_eps;
    }
}

}  // namespace

int launch_sigma_head(mpv_ctx* ctx, const float* series, int V, int64_t L, int n_windows, int64_t first, int64_t step,
                      const float* mean, const float* std, float add_eps, float* gx) {
    const SigmaModel& m = ctx->sg;
    SigmaArgs a{};
    a.w0t = m.w0t; a.b0 = m.b0; a.ln0w = m.ln0w; a.ln0b = m.ln0b; a.w1t = m.w1t; a.b1 = m.b1;
    a.ln1w = m.ln1w; a.ln1b = m.ln1b; a.w2t = m.w2t; a.b2 = m.b2;
    a.series = series; a.mean = mean; a.std = std; a.gx = gx;
    a.L = L; a.first = first; a.step = step; a.n_groups = (long long)n_windows * V;
    a.V = V; a.Tw = m.d.seq_len; a.k = m.d.rolling; a.K0 = m.d.seq_len - m.d.rolling; a.O = m.d.pred_len;
    a.F = m.d.nf; a.H = m.d.hidden; a.add_eps = add_eps;
    if (a.n_groups == 0) return MPV_OK;
    const size_t lds = 64 + sizeof(float) * (size_t)GROUPS * a.F * (a.Tw + a.K0 + 2 * a.H);
    if (lds > ctx->max_lds) return mpv_fail(ctx, MPV_EINVAL, "sigma head: window too long for LDS staging");
    MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sigma_head_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long blocks = (a.n_groups + GROUPS - 1) / GROUPS;
    hipLaunchKernelGGL(sigma_head_kernel, dim3((unsigned)blocks), dim3(SG_THREADS), lds, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}
