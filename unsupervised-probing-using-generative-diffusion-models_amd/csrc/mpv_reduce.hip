// Mean-predictive-variance reduction for gfx950.
//
// samples [W][R][F][S] -> per window: mean over (r,f) of Var_pop over s of (x*std+mean), the mean
// of everything, and the per-feature variance means.  Follows summarize_pred_future_list
// (reference diffusion_model_uncertainy.py:286-303, inverse scaling :267-283) and the variance
// part of summarize_slbp_sensitivity (:529-538).
//
// HBM-bound streaming read (4*R*F*S bytes per window, read once).  A (r,f) row of S samples sits
// in one contiguous segment; SEG = next power of two >= min(S,64) lanes share a row and reduce it
// with wavefront shuffles (xor butterflies inside the segment); rows longer than 64 loop.  Sums
// across rows are kept in fp64 and combined in a fixed order (partials + finalize kernel), so
// the result is bitwise reproducible.
#include "mpv_common.h"

namespace {

constexpr int RD_THREADS = 256;
constexpr int RD_BLOCKS_PER_WINDOW = 32;

template <int SEG>
__device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
    for (int m = SEG / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

struct ReduceArgs {
    const float* samples;
    const float* std;
    const float* mean;
    double* partial;     // [W][nblk][F][2]
    long long R;
    int W, F, S, nblk;
};

// SEG lanes share a row; each lane reads VEC consecutive samples per slot (VEC = 4: one 16-byte load, S % 4 == 0)
template <int SEG, int VEC>
__global__ __launch_bounds__(RD_THREADS) void mpv_partial_kernel(const ReduceArgs p) {
    __shared__ double red[RD_THREADS / 64][MPV_MAX_NF][2];
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = lane / SEG, ls = lane % SEG;
    constexpr int ROWS_PER_WAVE = 64 / SEG;
    const long long rows = p.R * p.F;                       // rows of S samples, row q = r*F + f
    const float* base = p.samples + (long long)w * rows * p.S;
    double vsum[MPV_MAX_NF] = {0, 0, 0, 0}, msum[MPV_MAX_NF] = {0, 0, 0, 0};
    const long long stride = (long long)p.nblk * (RD_THREADS / 64) * ROWS_PER_WAVE;
    const float inv_s = 1.0f / (float)p.S;
    // every lane iterates the same number of times so the shuffles stay converged
    const long long iters = (rows + stride - 1) / stride;
    for (long long it = 0; it < iters; ++it) {
        const long long q = it * stride + ((long long)blockIdx.x * (RD_THREADS / 64) + wave) * ROWS_PER_WAVE + seg;
        const bool live = q < rows;
        const int f = live ? (int)(q % p.F) : 0;
        const float sc = p.std ? p.std[f] : 1.0f, sh = p.mean ? p.mean[f] : 0.0f;
        const float* x = base + (live ? q : 0) * p.S;
        float s1 = 0.0f, s2 = 0.0f;
        if (VEC == 4) {
            for (int s = ls * 4; s < p.S; s += SEG * 4) {
                const float4 t = live ? *reinterpret_cast<const float4*>(x + s) : make_float4(0.f, 0.f, 0.f, 0.f);
                s1 += live ? ((t.x * sc + sh) + (t.y * sc + sh)) + ((t.z * sc + sh) + (t.w * sc + sh)) : 0.0f;
            }
        } else {
            for (int s = ls; s < p.S; s += SEG) s1 += live ? (x[s] * sc + sh) : 0.0f;
        }
        const float m = seg_sum<SEG>(s1) * inv_s;
        if (VEC == 4) {
            for (int s = ls * 4; s < p.S; s += SEG * 4) {
                if (live) {
                    const float4 t = *reinterpret_cast<const float4*>(x + s);          // second pass hits L1/L2
                    const float d0 = (t.x * sc + sh) - m, d1 = (t.y * sc + sh) - m, d2 = (t.z * sc + sh) - m, d3 = (t.w * sc + sh) - m;
                    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
        } else {
            for (int s = ls; s < p.S; s += SEG) {
                const float d = live ? (x[s] * sc + sh) - m : 0.0f;
                s2 += d * d;
            }
        }
        const float var = seg_sum<SEG>(s2) * inv_s;
        if (live && ls == 0) {
#pragma unroll
            for (int k = 0; k < MPV_MAX_NF; ++k)
                if (k == f) {
                    vsum[k] += (double)var;
                    msum[k] += (double)m;
                }
        }
    }
#pragma unroll
    for (int k = 0; k < MPV_MAX_NF; ++k) {
        const double a = wave_sum_d(vsum[k]), b = wave_sum_d(msum[k]);
        if (lane == 0) {
            red[wave][k][0] = a;
            red[wave][k][1] = b;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < p.F * 2) {
        const int k = threadIdx.x >> 1, c = threadIdx.x & 1;
        double t = 0.0;
        for (int i = 0; i < RD_THREADS / 64; ++i) t += red[i][k][c];
        p.partial[(((long long)w * p.nblk + blockIdx.x) * p.F + k) * 2 + c] = t;
    }
}

__global__ void mpv_finalize_kernel(const double* partial, int W, int nblk, int F, long long R, float* mpv,
                                    float* pred_mean, float* mpv_feat) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    double vt = 0.0, mt = 0.0;
    for (int f = 0; f < F; ++f) {
        double v = 0.0, m = 0.0;
        for (int b = 0; b < nblk; ++b) {
            v += partial[(((long long)w * nblk + b) * F + f) * 2 + 0];
            m += partial[(((long long)w * nblk + b) * F + f) * 2 + 1];
        }
        if (mpv_feat) mpv_feat[w * F + f] = (float)(v / (double)R);
        vt += v;
        mt += m;
    }
    mpv[w] = (float)(vt / ((double)R * F));
    pred_mean[w] = (float)(mt / ((double)R * F));   // mean of per-row means = mean of everything
}

}  // namespace

int launch_mpv_reduce(mpv_ctx* ctx, const float* samples, int W, int64_t R, int F, int S, const float* std,
                      const float* mean, float* mpv, float* pred_mean, float* mpv_feat) {
    if (W == 0) return MPV_OK;
    const int nblk = RD_BLOCKS_PER_WINDOW;
    const size_t need = sizeof(double) * (size_t)W * nblk * F * 2;
    if (need > ctx->reduce_ws_bytes) {
        if (ctx->reduce_ws) (void)hipFree(ctx->reduce_ws);
        ctx->reduce_ws = nullptr;
        ctx->reduce_ws_bytes = 0;
        if (hipMalloc(&ctx->reduce_ws, need) != hipSuccess) return mpv_fail(ctx, MPV_ENOMEM, "reduce workspace");
        ctx->reduce_ws_bytes = need;
    }
    ReduceArgs a{samples, std, mean, ctx->reduce_ws, (long long)R, W, F, S, nblk};
    const dim3 grid(nblk, W), block(RD_THREADS);
    const bool vec = (S % 4 == 0) && (reinterpret_cast<uintptr_t>(samples) % 16 == 0);
    const int slots = vec ? S / 4 : S;                        // lanes a row can keep busy
    int seg = 1;
    while (seg < slots && seg < 64) seg <<= 1;
#define MPV_RD_LAUNCH(SEG_)                                                                                   \
    if (vec) hipLaunchKernelGGL((mpv_partial_kernel<SEG_, 4>), grid, block, 0, ctx->stream, a);              \
    else hipLaunchKernelGGL((mpv_partial_kernel<SEG_, 1>), grid, block, 0, ctx->stream, a)
    switch (seg) {
        case 1: MPV_RD_LAUNCH(1); break;
        case 2: MPV_RD_LAUNCH(2); break;
        case 4: MPV_RD_LAUNCH(4); break;
        case 8: MPV_RD_LAUNCH(8); break;
        case 16: MPV_RD_LAUNCH(16); break;
        case 32: MPV_RD_LAUNCH(32); break;
        default: MPV_RD_LAUNCH(64); break;
    }
#undef MPV_RD_LAUNCH
    MPV_HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(mpv_finalize_kernel, dim3((W + 63) / 64), dim3(64), 0, ctx->stream, ctx->reduce_ws, W, nblk, F,
                       (long long)R, mpv, pred_mean, mpv_feat);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}
