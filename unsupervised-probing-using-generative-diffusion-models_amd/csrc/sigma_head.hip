// sigma head g(x) for gfx950: sliding-window gather + scaler + trailing variance + 3-layer MLP.
//
// Follows wv_sigma_trailing (reference NsDiff/sigma.py:34-72) and SigmaEstimation.forward
// (NsDiff/g_backbone.py:49-72).  Only the last T-k trailing variances feed the MLP, and for
// those the replicate padding never enters the window, so v[i] = Var_pop(x[i+1 .. i+k]),
// i = 0..T-k-1.  torch's CPU variance accumulates in double (Welford); so do we (two-pass).
//
// This kernel runs once per window (<1% of the sampling path's time, all of the gx-only indicator's); it is L2-bound
// on the 1.6 MB of MLP weights, which every workgroup streams once for G (window,node) rows from the transposed
// [in][out] copy (coalesced across the threads of a workgroup).  G is as large as the accumulator registers allow
// (G*F = 16 dot products per thread): 16 rows per workgroup at F = 1 cuts the weight traffic 4x against the first
// version (4 rows) -- 18 100 rows: 2.55 ms -> see DESIGN.md 3.2.
#include "mpv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SG_THREADS = 256;

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

// out[r][j] = act(bias[j] + sum_k in[r][k] * Wt[k][j])   for j < N, r = g*F + f over the R = G*F rows of the workgroup
// (rows beyond the live ones hold zeros and are computed but never stored: no divergence in the inner loop).
// `in` rows are 16-byte aligned (stride a multiple of 4 floats): one broadcast ds_read_b128 feeds four FMAs.
template <int R>
__device__ __forceinline__ void dot_rows(const float* __restrict__ Wt, const float* in, int in_stride, int K, int N, int j, float (&acc)[R]) {
    int k = 0;
    for (; k + 4 <= K; k += 4) {
        const float w0 = Wt[(long long)k * N + j], w1 = Wt[(long long)(k + 1) * N + j];
        const float w2 = Wt[(long long)(k + 2) * N + j], w3 = Wt[(long long)(k + 3) * N + j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(in + r * in_stride + k);
            acc[r] = __builtin_fmaf(x[0], w0, acc[r]);
            acc[r] = __builtin_fmaf(x[1], w1, acc[r]);
            acc[r] = __builtin_fmaf(x[2], w2, acc[r]);
            acc[r] = __builtin_fmaf(x[3], w3, acc[r]);
        }
    }
    for (; k < K; ++k) {
        const float w = Wt[(long long)k * N + j];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_fmaf(in[r * in_stride + k], w, acc[r]);
    }
}

template <bool RELU, int R>
__device__ __forceinline__ void dense(const float* __restrict__ Wt, const float* __restrict__ bias, const float* in, int in_stride,
                                      float* out, int out_stride, int K, int N, int live_rows) {
    for (int j = threadIdx.x; j < N; j += SG_THREADS) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = bias[j];
        dot_rows<R>(Wt, in, in_stride, K, N, j, acc);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < live_rows) out[r * out_stride + j] = RELU ? fmaxf(acc[r], 0.0f) : acc[r];
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

template <int F, int GROUPS>
__global__ __launch_bounds__(SG_THREADS) void sigma_head_kernel(const SigmaArgs p) {
    constexpr int R = GROUPS * F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* red = reinterpret_cast<double*>(smem);                 // 8 doubles
    const int K0p = (p.K0 + 3) & ~3;                               // row stride of v: 16-byte aligned rows
    const int Twp = (p.Tw + 3) & ~3;
    float* v = reinterpret_cast<float*>(smem + 64);               // [GROUPS][F][K0p]
    float* xw = v + GROUPS * F * K0p;                              // [GROUPS][F][Twp]: dead once v is formed, so the
    float* h1 = xw;                                                // hidden layers reuse its space: [GROUPS][F][H]
    float* h2 = h1 + GROUPS * F * p.H;                             // [GROUPS][F][H]

    const long long g0 = (long long)blockIdx.x * GROUPS;
    const long long left = p.n_groups - g0;
    const int ng = (int)(left < GROUPS ? left : GROUPS);
    if (ng < GROUPS) {                                             // rows of a ragged last workgroup: zeros, never stored
        for (int i = threadIdx.x; i < R * K0p; i += SG_THREADS) v[i] = 0.0f;
    }

    // ---- gather + scale the windows: xw[g][f][t] --------------------------------------
    for (int i = threadIdx.x; i < ng * p.Tw * F; i += SG_THREADS) {
        const int f = i % F;
        const int t = (i / F) % p.Tw;
        const int g = i / (F * p.Tw);
        const long long gi = g0 + g;
        const long long w = gi / p.V;
        const int node = (int)(gi % p.V);
        float x = p.series[((long long)node * p.L + p.first + w * p.step + t) * F + f];
        if (p.mean) x = (x - p.mean[f]) / p.std[f];
        xw[(g * F + f) * Twp + t] = x;
    }
    __syncthreads();
    // ---- trailing population variance of the last K0 positions, + 10e-8 ----------------
    for (int i = threadIdx.x; i < ng * F * p.K0; i += SG_THREADS) {
        const int pos = i % p.K0;
        const float* x = xw + (i / p.K0) * Twp + pos + 1;
        double s = 0.0;
        for (int j = 0; j < p.k; ++j) s += (double)x[j];
        const double m = s / p.k;
        double q = 0.0;
        for (int j = 0; j < p.k; ++j) {
            const double d = (double)x[j] - m;
            q += d * d;
        }
        v[(i / p.K0) * K0p + pos] = (float)(q / p.k) + 10e-8f;
    }
    __syncthreads();
    if (ng < GROUPS) {                                             // (xw is dead now: its space becomes h1 / h2)
        for (int i = threadIdx.x; i < R * p.H; i += SG_THREADS) h1[i] = h2[i] = 0.0f;
        __syncthreads();
    }
    // ---- MLP --------------------------------------------------------------------------
    dense<true, R>(p.w0t, p.b0, v, K0p, h1, p.H, p.K0, p.H, ng * F);
    __syncthreads();
    layer_norm(h1, F * p.H, p.ln0w, p.ln0b, F, p.H, ng, red);
    dense<true, R>(p.w1t, p.b1, h1, p.H, h2, p.H, p.H, p.H, ng * F);
    __syncthreads();
    layer_norm(h2, F * p.H, p.ln1w, p.ln1b, F, p.H, ng, red);
    // last layer straight to HBM: gx[gi][o][f] = softplus(.) + add_eps
    for (int j = threadIdx.x; j < p.O; j += SG_THREADS) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = p.b2[j];
        dot_rows<R>(p.w2t, h2, p.H, p.H, p.O, j, acc);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < ng * F) p.gx[((g0 + r / F) * p.O + j) * F + (r % F)] = softplus_acc(acc[r]) + p.add_eps;
    }
}

// LDS of one workgroup: v rows + max(window rows, two hidden layers)
static size_t sigma_lds_bytes(const SigmaArgs& a, int groups) {
    const size_t twp = (a.Tw + 3) & ~3, k0p = (a.K0 + 3) & ~3, hid = 2 * (size_t)a.H;
    return 64 + sizeof(float) * (size_t)groups * a.F * (k0p + (twp > hid ? twp : hid));
}

// G rows per workgroup: as many as the accumulator registers allow (G*F = 16), halved until the staging fits the LDS
template <int F, int GROUPS>
int launch_sigma_fg(mpv_ctx* ctx, const SigmaArgs& a) {
    const size_t lds = sigma_lds_bytes(a, GROUPS);
    if (lds > ctx->max_lds) {
        if constexpr (GROUPS > 1) return launch_sigma_fg<F, GROUPS / 2>(ctx, a);
        else return mpv_fail(ctx, MPV_EINVAL, "sigma head: window too long for LDS staging even with one row per workgroup");
    }
    MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sigma_head_kernel<F, GROUPS>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long blocks = (a.n_groups + GROUPS - 1) / GROUPS;
    hipLaunchKernelGGL((sigma_head_kernel<F, GROUPS>), dim3((unsigned)blocks), dim3(SG_THREADS), lds, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
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
    switch (a.F) {                                  // G*F = 16 accumulators per thread (8 for F = 3)
        case 1: return launch_sigma_fg<1, 16>(ctx, a);
        case 2: return launch_sigma_fg<2, 8>(ctx, a);
        case 3: return launch_sigma_fg<3, 4>(ctx, a);
        case 4: return launch_sigma_fg<4, 4>(ctx, a);
        default: return mpv_fail(ctx, MPV_EINVAL, "sigma head: nf must be 1..4");
    }
}
