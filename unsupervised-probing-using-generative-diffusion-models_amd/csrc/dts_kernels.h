// DiffusionTS transformer for gfx950: device kernels (forward and input-gradient).
//
// Follows reference models/Diffusion_model/DiffusionTS/diffusionts_transformer.py:12-438 and
// diffusionts_model_utils.py:158-202 (Conv_MLP, AdaLayerNorm).  Activations are fp32 [B][L][D] ("token major", D fastest).
// Every dense contraction runs on the exact-fp32 matrix pipe, v_mfma_f32_16x16x4_f32 (the Langevin infill update is
// g / (|g| + 1e-10), i.e. sign descent: reduced-precision products would flip signs and leave the reference's trajectory):
//   * dts_lin_kernel      token-wise Linear  Y^T[n][tok] = W[n][k] X^T[k][tok]  with LayerNorm / AdaLayerNorm / GELU / GELU'
//                         prologues and bias / residual / accumulate epilogues  (QKV, proj, MLP, and their transposes)
//   * dts_seqmix_kernel   per-sequence  Y_b = W X_b  (Conv1d(L -> 2L, k=1) over the TOKEN axis, its transpose, the real DFT)
//   * dts_attn_*          softmax attention per (sequence, head): forward (online softmax), dQ pass, dK/dV pass.  A 16x16
//                         score tile in accumulator layout (column = lane&15, rows = 4*(lane>>4)+reg) is directly the B
//                         operand of the next product, so P never leaves registers and nothing is transposed.
// Operand convention of the builtin: lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D[i = 4*(l>>4)+r][j = l&15].
// A lane reads 4 consecutive k (one ds_read_b128) and issues 4 MFMAs; MFMA e then contracts k = {16s + 4g + e : g = 0..3}.
#pragma once
#include "mpv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace dts {

#define DTS_MMA(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (acc), 0, 0, 0)

constexpr float LOG2E = 1.44269504088896341f;

// erf on one path: erf(|x|) = 1 - 2^p(|x|), p = log2 erfc as a degree-9 polynomial without constant term on [0, 4] (weighted minimax fit; erf = 1 in
// fp32 beyond 3.92).  Absolute error <= 1e-7 (2e-9 for the polynomial, the rest is the rounding of 1 - 2^p), the same as the library's near |x| ~ 1;
// unlike the library's it is not RELATIVELY accurate for |x| << 1, which GELU does not need: erf enters as 1 + erf.  13 instructions against the ~28
// of the two-branch library routine, which the fused MLP kernels (256 hidden units per token) spent a third of their time in.
__device__ __forceinline__ float erf_abs7(float x) {
    float t = fabsf(x);
    t = t > 4.0f ? 4.0f : t;                                           // (a NaN stays a NaN)
    float p = 1.162548672e-05f;
    p = __builtin_fmaf(p, t, -1.531847762e-04f);
    p = __builtin_fmaf(p, t, 8.491812494e-04f);
    p = __builtin_fmaf(p, t, -2.276909126e-03f);
    p = __builtin_fmaf(p, t, 8.749411102e-05f);
    p = __builtin_fmaf(p, t, 2.772253888e-02f);
    p = __builtin_fmaf(p, t, -1.483071528e-01f);
    p = __builtin_fmaf(p, t, -9.184430430e-01f);
    p = __builtin_fmaf(p, t, -1.627907258e+00f);
    return copysignf(1.0f - __builtin_amdgcn_exp2f(p * t), x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_abs7(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erf_abs7(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

// ---------------------------------------------------------------------------------------------------------------------
// token-wise Linear
// ---------------------------------------------------------------------------------------------------------------------
enum { PRO_NONE = 0, PRO_LN = 1, PRO_ADALN = 2, PRO_GELUGRAD = 3, PRO_GELU = 4 };

struct LinArgs {
    const float* X; int ldx;        // [rows][ldx], first K columns
    const float* W;                 // [N][K]
    const float* bias;              // [N] | null
    const float* R; int ldr;        // residual [rows][ldr] | null
    float* Y; int ldy;              // [rows][ldy]
    const float* P1; const float* P2;   // LN: weight, bias [K]; ADALN: scale, shift [K]; GELUGRAD: P1 = u [rows][ldp]
    int ldp;
    long long rows;
    int K, N;
    int pro;
    int accumulate;                 // Y += result
    // 16-column BLOCKED operands (dts_lin_wave_kernel only): element (row, c) of X at X + (c / 16) xblk + row * ldx + c % 16 (ldx = 16), likewise Y with
    // yblk; 0 = plain row-major.  The attention operands (Q | K | V, the attention output and their gradients) live in this layout, so that one
    // (sequence, head) tile is contiguous: [block = which * H + head][rows][16].
    long long xblk, yblk;
    // dts_lin_wave_h_kernel only, N <= 64: the product is the gradient w.r.t. a LayerNorm OUTPUT and the kernel finishes the LayerNorm's input gradient
    // (dts_ln_bwd_kernel fused): lnx = the LayerNorm's input rows [rows][N], lng = its weight (ln_ada: 1 + scale of an AdaLayerNorm); Y then receives
    // (accumulate: +=) rstd (g' - mean(g') - xhat mean(g' xhat)), g' = product x weight.  null: plain epilogue.
    const float* lnx = nullptr; const float* lng = nullptr; int ln_ada = 0;
};

constexpr int LIN_TM = 64;          // tokens per workgroup pass
constexpr int LIN_THREADS = 256;

__global__ __launch_bounds__(LIN_THREADS) void dts_lin_kernel(const LinArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int KS = p.K + 4;                         // padded row stride
    float* Wl = lds;                                // [N][KS]
    float* Xl = lds + (size_t)p.N * KS;             // [64][KS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int K4 = p.K >> 2;
    for (int i = tid; i < p.N * K4; i += LIN_THREADS) {
        const int n = i / K4, c = i - n * K4;
        *reinterpret_cast<f32x4*>(Wl + n * KS + 4 * c) = *reinterpret_cast<const f32x4*>(p.W + (size_t)n * p.K + 4 * c);
    }
    const long long n_tiles = (p.rows + LIN_TM - 1) / LIN_TM;
    const int srow = tid >> 2, spart = tid & 3;     // staging: 4 threads per token row
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                            // previous pass done with Xl (and Wl staged)
        // ---- stage 64 token rows with the prologue applied ----
        {
            const long long tok = tile * LIN_TM + srow;
            const bool live = tok < p.rows;
            const float* xr = p.X + (live ? tok : 0) * (long long)p.ldx;
            float* dst = Xl + srow * KS;
            if (p.pro == PRO_LN || p.pro == PRO_ADALN) {
                float s = 0.0f;
                for (int c = spart; c < K4; c += 4) {
                    const f32x4 v = live ? *reinterpret_cast<const f32x4*>(xr + 4 * c) : f32x4{0, 0, 0, 0};
                    *reinterpret_cast<f32x4*>(dst + 4 * c) = v;
                    s += (v[0] + v[1]) + (v[2] + v[3]);
                }
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                const float mean = s / (float)p.K;
                float q = 0.0f;
                for (int c = spart; c < K4; c += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(dst + 4 * c);
                    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
                q += __shfl_xor(q, 1);
                q += __shfl_xor(q, 2);
                const float rstd = 1.0f / sqrtf(q / (float)p.K + 1e-5f);
                for (int c = spart; c < K4; c += 4) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(dst + 4 * c);
                    const f32x4 a = *reinterpret_cast<const f32x4*>(p.P1 + 4 * c), b = *reinterpret_cast<const f32x4*>(p.P2 + 4 * c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (v[e] - mean) * rstd;
                        v[e] = (p.pro == PRO_ADALN) ? xh * (1.0f + a[e]) + b[e] : xh * a[e] + b[e];
                    }
                    *reinterpret_cast<f32x4*>(dst + 4 * c) = v;
                }
            } else {
                const float* ur = p.P1 ? p.P1 + (live ? tok : 0) * (long long)p.ldp : nullptr;
                for (int c = spart; c < K4; c += 4) {
                    f32x4 v = live ? *reinterpret_cast<const f32x4*>(xr + 4 * c) : f32x4{0, 0, 0, 0};
                    if (p.pro == PRO_GELUGRAD) {
                        const f32x4 u = live ? *reinterpret_cast<const f32x4*>(ur + 4 * c) : f32x4{0, 0, 0, 0};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(u[e]);
                    } else if (p.pro == PRO_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
                    }
                    *reinterpret_cast<f32x4*>(dst + 4 * c) = v;
                }
            }
        }
        __syncthreads();
        // ---- wave `wave` multiplies its 16 tokens with every 16-row block of W ----
        const long long tok = tile * LIN_TM + 16 * wave + r16;
        const float* xb = Xl + (16 * wave + r16) * KS + 4 * g;
        for (int nt = 0; nt < (p.N >> 4); ++nt) {
            f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            const float* wa = Wl + (16 * nt + r16) * KS + 4 * g;
            int s = 0;
            for (; s + 2 <= (p.K >> 4); s += 2) {                 // two independent accumulator chains
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(wa + 16 * s), b0 = *reinterpret_cast<const f32x4*>(xb + 16 * s);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(wa + 16 * s + 16), b1 = *reinterpret_cast<const f32x4*>(xb + 16 * s + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    DTS_MMA(acc0, a0[e], b0[e]);
                    DTS_MMA(acc1, a1[e], b1[e]);
                }
            }
            for (; s < (p.K >> 4); ++s) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(wa + 16 * s), b0 = *reinterpret_cast<const f32x4*>(xb + 16 * s);
#pragma unroll
                for (int e = 0; e < 4; ++e) DTS_MMA(acc0, a0[e], b0[e]);
            }
            if (tok < p.rows) {
                const int n0 = 16 * nt + 4 * g;
                f32x4 v = acc0 + acc1;
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n0);
                if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + tok * (long long)p.ldr + n0);
                float* y = p.Y + tok * (long long)p.ldy + n0;
                if (p.accumulate) v += *reinterpret_cast<const f32x4*>(y);
                *reinterpret_cast<f32x4*>(y) = v;
            }
        }
    }
}

// The same Linear with one 16-token tile per WAVE and nothing but W in LDS: a tile's rows are loaded from global memory straight into the
// B-operand layout (lane = token r16, k-slot g: X[tok][16 s + 4 g ..]), the prologue (LayerNorm statistics over a token's four lanes with two
// shuffles, AdaLN, GELU, GELU') runs in registers, and no barrier separates the waves.  8 waves per workgroup, as many workgroups per CU as W's
// footprint allows (W 192 x 64: three).  KT = K / 16.
constexpr int LINW_WAVES = 8;

template <int KT>
__global__ __launch_bounds__(64 * LINW_WAVES) void dts_lin_wave_kernel(const LinArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int K = 16 * KT, KS = K + 4;
    float* Wl = lds;                                // [N][KS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < p.N * (K / 4); i += 64 * LINW_WAVES) {
        const int n = i / (K / 4), c = i - n * (K / 4);
        *reinterpret_cast<f32x4*>(Wl + n * KS + 4 * c) = *reinterpret_cast<const f32x4*>(p.W + (size_t)n * K + 4 * c);
    }
    __syncthreads();
    const long long n_tiles = (p.rows + 15) / 16;
    const long long xs = p.xblk ? p.xblk : 16, ys = p.yblk ? p.yblk : 16;
    // the X tile of the wave's NEXT iteration is requested before this one's products (narrow K only: it costs KT more float4 registers)
#ifndef DTS_LINW_PREFETCH
#define DTS_LINW_PREFETCH 1
#endif
    constexpr bool AHEAD = (DTS_LINW_PREFETCH != 0) && KT <= 4;
    const long long tile0 = (long long)blockIdx.x * LINW_WAVES + wave, tstep = (long long)gridDim.x * LINW_WAVES;
    auto load_tile = [&](long long tile, f32x4 (&dst)[KT]) {
        const long long tok = tile * 16 + r16;
        const bool live = tile < n_tiles && tok < p.rows;
        const float* xr = p.X + (live ? tok : 0) * (long long)p.ldx + 4 * g;
#pragma unroll
        for (int s = 0; s < KT; ++s) dst[s] = live ? *reinterpret_cast<const f32x4*>(xr + xs * s) : f32x4{0, 0, 0, 0};
    };
    f32x4 xn[AHEAD ? KT : 1];
    if constexpr (AHEAD) load_tile(tile0, xn);
    for (long long tile = tile0; tile < n_tiles; tile += tstep) {
        asm volatile("" ::: "memory");              // weight fragments stay in LDS (hoisted out of the loop they would spill)
        const long long tok = tile * 16 + r16;
        const bool live = tok < p.rows;
        f32x4 xb[KT];
        if constexpr (AHEAD) {
#pragma unroll
            for (int s = 0; s < KT; ++s) xb[s] = xn[s];
        } else {
            load_tile(tile, xb);
        }
        if (p.pro == PRO_LN || p.pro == PRO_ADALN) {
            float sum = 0.0f;
#pragma unroll
            for (int s = 0; s < KT; ++s) sum += (xb[s][0] + xb[s][1]) + (xb[s][2] + xb[s][3]);
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float mean = sum / (float)K;
            float q = 0.0f;
#pragma unroll
            for (int s = 0; s < KT; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xb[s][e] -= mean;
                    q += xb[s][e] * xb[s][e];
                }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            const float rstd = 1.0f / sqrtf(q / (float)K + 1e-5f);
#pragma unroll
            for (int s = 0; s < KT; ++s) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.P1 + 16 * s + 4 * g), b = *reinterpret_cast<const f32x4*>(p.P2 + 16 * s + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = xb[s][e] * rstd;
                    xb[s][e] = (p.pro == PRO_ADALN) ? xh * (1.0f + a[e]) + b[e] : xh * a[e] + b[e];
                }
            }
        } else if (p.pro == PRO_GELUGRAD) {
            const float* ur = p.P1 + (live ? tok : 0) * (long long)p.ldp + 4 * g;
#pragma unroll
            for (int s = 0; s < KT; ++s) {
                const f32x4 u = live ? *reinterpret_cast<const f32x4*>(ur + 16 * s) : f32x4{0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 4; ++e) xb[s][e] *= gelu_grad_f(u[e]);
            }
        } else if (p.pro == PRO_GELU) {
#pragma unroll
            for (int s = 0; s < KT; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) xb[s][e] = gelu_f(xb[s][e]);
        }
        // output blocks four at a time: everything the four epilogues add (residual, accumulation target) is requested up front, ahead of the
        // next tile's X, so that the first wait of the products covers these older loads only and the X tile stays in flight
        const int NT = p.N >> 4;
        for (int nt0 = 0; nt0 < NT; nt0 += 4) {
            f32x4 addend[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nt = nt0 + i;
                addend[i] = f32x4{0, 0, 0, 0};
                if (live && nt < NT) {
                    if (p.R) addend[i] = *reinterpret_cast<const f32x4*>(p.R + tok * (long long)p.ldr + 16 * nt + 4 * g);
                    if (p.accumulate) addend[i] += *reinterpret_cast<const f32x4*>(p.Y + tok * (long long)p.ldy + ys * nt + 4 * g);
                }
            }
            if constexpr (AHEAD) {
                if (nt0 == 0) load_tile(tile + tstep, xn);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nt = nt0 + i;
                if (nt >= NT) break;
                const int n0 = 16 * nt + 4 * g;
                f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                const float* wa = Wl + (16 * nt + r16) * KS + 4 * g;
#pragma unroll
                for (int s = 0; s < KT; ++s) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(wa + 16 * s);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (s & 1) DTS_MMA(acc1, a[e], xb[s][e]);
                        else DTS_MMA(acc0, a[e], xb[s][e]);
                    }
                }
                if (live) {
                    f32x4 v = acc0 + acc1;
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n0);
                    v += addend[i];
                    *reinterpret_cast<f32x4*>(p.Y + tok * (long long)p.ldy + ys * nt + 4 * g) = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm / AdaLayerNorm backward w.r.t. the input:  dx (+)= rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))
// 16 lanes per token, NPL = D/16 channels per lane
// ---------------------------------------------------------------------------------------------------------------------
template <int NPL>
__global__ __launch_bounds__(256) void dts_ln_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ gam,
                                                         int ada, float* dx, int accumulate, long long rows) {
    constexpr int D = 16 * NPL;
    const long long tok = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int li = threadIdx.x & 15;
    const bool live = tok < rows;
    // all three streams (x, g and, when accumulating, the old dx) are requested before the first reduction
    float xv[NPL], gv[NPL], ov[NPL];
#pragma unroll
    for (int e = 0; e < NPL; ++e) {
        xv[e] = live ? x[tok * D + NPL * li + e] : 0.0f;
        gv[e] = live ? g[tok * D + NPL * li + e] : 0.0f;
        ov[e] = (live && accumulate) ? dx[tok * D + NPL * li + e] : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < NPL; ++e) s += xv[e];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m);
    const float mean = s / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < NPL; ++e) {
        xv[e] -= mean;
        q += xv[e] * xv[e];
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) q += __shfl_xor(q, m);
    const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
    float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
    for (int e = 0; e < NPL; ++e) {
        const float gm = ada ? 1.0f + gam[NPL * li + e] : gam[NPL * li + e];
        xv[e] *= rstd;                                             // xhat
        gv[e] = gv[e] * gm;
        m1 += gv[e];
        m2 += gv[e] * xv[e];
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        m1 += __shfl_xor(m1, m);
        m2 += __shfl_xor(m2, m);
    }
    m1 /= (float)D;
    m2 /= (float)D;
    if (live) {
#pragma unroll
        for (int e = 0; e < NPL; ++e) {
            const float v = rstd * (gv[e] - m1 - xv[e] * m2);
            dx[tok * D + NPL * li + e] = accumulate ? ov[e] + v : v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// per-sequence mixing along the token axis:  Y[b][m][d] (+)= bias[m] + sum_k W[m][k] X[b][k][d]
// W is stored padded [Mp][Kp] (multiples of 16, zero filled); one workgroup per sequence, X_b staged in LDS.
// ---------------------------------------------------------------------------------------------------------------------
struct SeqmixArgs {
    const float* W; int Mp, Kp;
    const float* bias;              // [M] | null
    const float* X; long long xs; int ldx; int K;     // X_b = X + b*xs, rows k < K, row stride ldx
    float* Y; long long ys; int ldy; int M;           // Y_b = Y + b*ys, rows m < M
    int accumulate;
};

template <int DT>                   // D = 16*DT
__global__ __launch_bounds__(256) void dts_seqmix_kernel(const SeqmixArgs p) {
    constexpr int D = 16 * DT, DS = D + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [Kp][DS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const float* xb = p.X + (long long)blockIdx.x * p.xs;
    for (int i = tid; i < p.Kp * (D / 4); i += 256) {
        const int k = i / (D / 4), c = i - k * (D / 4);
        const f32x4 v = k < p.K ? *reinterpret_cast<const f32x4*>(xb + (long long)k * p.ldx + 4 * c) : f32x4{0, 0, 0, 0};
        *reinterpret_cast<f32x4*>(lds + k * DS + 4 * c) = v;
    }
    __syncthreads();
    float* yb = p.Y + (long long)blockIdx.x * p.ys;
    for (int mt = wave; mt < (p.Mp >> 4); mt += 4) {
        f32x4 acc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0, 0, 0, 0};
        const float* wa = p.W + (long long)(16 * mt + r16) * p.Kp + 4 * g;
        for (int s = 0; s < (p.Kp >> 4); ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(wa + 16 * s);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* xr = lds + (16 * s + 4 * g + e) * DS + r16;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) DTS_MMA(acc[dt], a[e], xr[16 * dt]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * mt + 4 * g + r;
            if (m < p.M) {
                const float bm = p.bias ? p.bias[m] : 0.0f;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    float* o = yb + (long long)m * p.ldy + 16 * dt + r16;
                    const float v = acc[dt][r] + bm;
                    *o = p.accumulate ? *o + v : v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// attention, one workgroup per (sequence, head); E = head_dim / 4
// ---------------------------------------------------------------------------------------------------------------------
struct AttnArgs {
    const float *Q, *K, *V; int ldq, ldk, ldv;      // token rows; column 0 of head 0
    const float *dO, *O; int ldo;
    float* out; int ldout;                          // fwd: O ; dq: dQ ; dkv: dK
    float* out2; int ldout2;                        // dkv: dV
    float* lse;                                     // [B][H][Lq], base-2 log-sum-exp of the scaled scores
    const float* delta;                             // [B][H][Lq]
    int H, Lq, Lk, hd;
    float scale;
    long long hs;                                   // offset of head h's columns: h * hs floats.  Row-major token rows [rows][H hd]: hs = hd, ld = the row
                                                    // width; 16-column BLOCKED buffers [block][rows][16] (dts.hip): hs = rows * 16, every ld = 16
    float* out3 = nullptr; int ldout3 = 0;          // merged backward (dts_attn_bwd_h_kernel): dQ (out = dK, out2 = dV)
};

template <int E>
__device__ __forceinline__ void load_frag(const float* p, float (&f)[E]) {
    if constexpr (E == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) f[e] = p[e];
    }
}

// rows [n][hd] of one head (token stride ld) -> LDS row-major [np][hd+4] (rows >= n zero) and/or transposed [16][np+4]
template <int E>
__device__ __forceinline__ void stage_head(const float* src, int ld, int n, int np, float* row_major, float* transposed) {
    constexpr int HD = 4 * E;
    for (int i = threadIdx.x; i < np * E; i += 256) {                    // one float4 (four head columns of a token) per thread and pass
        const int r = i / E, d = 4 * (i - r * E);
        const f32x4 v = r < n ? *reinterpret_cast<const f32x4*>(src + (long long)r * ld + d) : f32x4{0, 0, 0, 0};
        if (row_major) *reinterpret_cast<f32x4*>(row_major + r * (HD + 4) + d) = v;
        if (transposed) {
#pragma unroll
            for (int e = 0; e < 4; ++e) transposed[(d + e) * (np + 4) + r] = v[e];
        }
    }
    if (transposed && HD < 16)
        for (int i = threadIdx.x; i < (16 - HD) * (np + 4); i += 256) transposed[HD * (np + 4) + i] = 0.0f;
}

template <int E>
__global__ __launch_bounds__(256) void dts_attn_fwd_kernel(const AttnArgs p) {
    constexpr int HD = 4 * E, RS = HD + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int Lkp = (p.Lk + 15) & ~15, TS = Lkp + 4;
    float* Kl = lds;                       // [Lkp][RS]
    float* VT = Kl + Lkp * RS;             // [16][TS]
    stage_head<E>(p.K + (long long)b * p.Lk * p.ldk + h * p.hs, p.ldk, p.Lk, Lkp, Kl, nullptr);
    stage_head<E>(p.V + (long long)b * p.Lk * p.ldv + h * p.hs, p.ldv, p.Lk, Lkp, nullptr, VT);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const float c = p.scale * LOG2E;
    for (int jt = wave; jt < ((p.Lq + 15) >> 4); jt += 4) {
        const int j = 16 * jt + r16;
        const bool jl = j < p.Lq;
        float qf[E];
        load_frag<E>(p.Q + ((long long)b * p.Lq + (jl ? j : 0)) * p.ldq + h * p.hs + E * g, qf);
        float m = -INFINITY, l = 0.0f;
        f32x4 o = {0, 0, 0, 0};
        for (int it = 0; it < (Lkp >> 4); ++it) {
            float kf[E];
            load_frag<E>(Kl + (16 * it + r16) * RS + E * g, kf);
            f32x4 s = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < E; ++e) DTS_MMA(s, kf[e], qf[e]);
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[r] = (16 * it + 4 * g + r < p.Lk) ? s[r] * c : -INFINITY;
                tmax = fmaxf(tmax, s[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float mn = fmaxf(m, tmax);                       // finite from the first tile on (Lk >= 1)
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - mn);
                ps += s[r];
            }
            l = l * alpha + ps;
            o *= alpha;
            m = mn;
            const f32x4 vt = *reinterpret_cast<const f32x4*>(VT + r16 * TS + 16 * it + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) DTS_MMA(o, vt[r], s[r]);
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        if (jl) {
            if (4 * g < HD) {
                const float inv = 1.0f / l;
                float* dst = p.out + ((long long)b * p.Lq + j) * p.ldout + h * p.hs + 4 * g;
                *reinterpret_cast<f32x4*>(dst) = o * inv;
            }
            if (g == 0 && p.lse) p.lse[((long long)b * p.H + h) * p.Lq + j] = m + __builtin_amdgcn_logf(l);
        }
    }
}

// delta[b][h][q] = sum_d dO[q][d] O[q][d]; thread per (token, head), heads fastest: a wave reads 64 consecutive head slices
__global__ void dts_attn_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O, int ld, long long hs, int H, int L, int hd, long long n, float* delta) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // tok*H + h
    if (i >= n) return;
    const int h = (int)(i % H);
    const long long tok = i / H;
    const long long b = tok / L;
    const int q = (int)(tok - b * L);
    const float* a = dO + tok * ld + h * hs;
    const float* o = O + tok * ld + h * hs;
    float s = 0.0f;
    for (int d = 0; d < hd; d += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + d), y = *reinterpret_cast<const f32x4*>(o + d);
        s += (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]);
    }
    delta[(b * H + h) * L + q] = s;
}

template <int E>
__global__ __launch_bounds__(256) void dts_attn_dq_kernel(const AttnArgs p) {
    constexpr int HD = 4 * E, RS = HD + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int Lkp = (p.Lk + 15) & ~15, TS = Lkp + 4;
    float* Kl = lds;                       // [Lkp][RS]
    float* Vl = Kl + Lkp * RS;             // [Lkp][RS]
    // (no transposed copy of K: the dS.K operand is read column-wise out of Kl with four ds_read_b32 -- 27 % less LDS, one more workgroup per CU)
    stage_head<E>(p.K + (long long)b * p.Lk * p.ldk + h * p.hs, p.ldk, p.Lk, Lkp, Kl, nullptr);
    stage_head<E>(p.V + (long long)b * p.Lk * p.ldv + h * p.hs, p.ldv, p.Lk, Lkp, Vl, nullptr);
    if (HD < 16) {                         // columns HD..15 of the padded rows are read by lanes r16 >= HD: keep them finite
        for (int i = threadIdx.x; i < Lkp * 4; i += 256) Kl[(i >> 2) * RS + HD + (i & 3)] = 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const float c = p.scale * LOG2E;
    for (int jt = wave; jt < ((p.Lq + 15) >> 4); jt += 4) {
        const int j = 16 * jt + r16;
        const bool jl = j < p.Lq;
        const long long tok = (long long)b * p.Lq + (jl ? j : 0);
        float qf[E], df[E];
        load_frag<E>(p.Q + tok * p.ldq + h * p.hs + E * g, qf);
        load_frag<E>(p.dO + tok * p.ldo + h * p.hs + E * g, df);
        const float lse = p.lse[((long long)b * p.H + h) * p.Lq + (jl ? j : 0)];
        const float del = p.delta[((long long)b * p.H + h) * p.Lq + (jl ? j : 0)];
        f32x4 dq = {0, 0, 0, 0};
        for (int it = 0; it < (Lkp >> 4); ++it) {
            float kf[E], vf[E];
            load_frag<E>(Kl + (16 * it + r16) * RS + E * g, kf);
            load_frag<E>(Vl + (16 * it + r16) * RS + E * g, vf);
            f32x4 s = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < E; ++e) {
                DTS_MMA(s, kf[e], qf[e]);
                DTS_MMA(t2, vf[e], df[e]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = (16 * it + 4 * g + r < p.Lk) ? __builtin_amdgcn_exp2f(s[r] * c - lse) : 0.0f;
                s[r] = pr * (t2[r] - del) * p.scale;
            }
            const float* kc = Kl + (16 * it + 4 * g) * RS + r16;                   // K^T[d = r16][key 4 g + r] = Kl[key][d]
#pragma unroll
            for (int r = 0; r < 4; ++r) DTS_MMA(dq, (r16 < HD ? kc[r * RS] : 0.0f), s[r]);
        }
        if (jl && 4 * g < HD)
            *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Lq + j) * p.ldout + h * p.hs + 4 * g) = dq;
    }
}

template <int E>
__global__ __launch_bounds__(256) void dts_attn_dkv_kernel(const AttnArgs p) {
    constexpr int HD = 4 * E, RS = HD + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int Lqp = (p.Lq + 15) & ~15, TS = Lqp + 4;
    float* Ql = lds;                       // [Lqp][RS]
    float* Dl = Ql + Lqp * RS;             // [Lqp][RS]   dO
    float* lsel = Dl + Lqp * RS;           // [Lqp]
    float* dell = lsel + Lqp;              // [Lqp]
    // (no transposed copies: the P^T.dO and dS^T.Q operands are read column-wise out of Dl / Ql -- 44 % less LDS, twice the workgroups per CU)
    stage_head<E>(p.Q + (long long)b * p.Lq * p.ldq + h * p.hs, p.ldq, p.Lq, Lqp, Ql, nullptr);
    stage_head<E>(p.dO + (long long)b * p.Lq * p.ldo + h * p.hs, p.ldo, p.Lq, Lqp, Dl, nullptr);
    for (int i = threadIdx.x; i < Lqp; i += 256) {
        lsel[i] = i < p.Lq ? p.lse[((long long)b * p.H + h) * p.Lq + i] : INFINITY;     // padded queries: P = 0
        dell[i] = i < p.Lq ? p.delta[((long long)b * p.H + h) * p.Lq + i] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const float c = p.scale * LOG2E;
    for (int jt = wave; jt < ((p.Lk + 15) >> 4); jt += 4) {
        const int j = 16 * jt + r16;                                  // key
        const bool jl = j < p.Lk;
        const long long tok = (long long)b * p.Lk + (jl ? j : 0);
        float kf[E], vf[E];
        load_frag<E>(p.K + tok * p.ldk + h * p.hs + E * g, kf);
        load_frag<E>(p.V + tok * p.ldv + h * p.hs + E * g, vf);
        f32x4 dk = {0, 0, 0, 0}, dv = {0, 0, 0, 0};
        for (int it = 0; it < (Lqp >> 4); ++it) {
            float qa[E], da[E];
            load_frag<E>(Ql + (16 * it + r16) * RS + E * g, qa);
            load_frag<E>(Dl + (16 * it + r16) * RS + E * g, da);
            f32x4 s = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < E; ++e) {
                DTS_MMA(s, qa[e], kf[e]);
                DTS_MMA(t2, da[e], vf[e]);
            }
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsel + 16 * it + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dell + 16 * it + 4 * g);
            f32x4 pr, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[r] = __builtin_amdgcn_exp2f(s[r] * c - l4[r]);
                ds[r] = pr[r] * (t2[r] - d4[r]) * p.scale;
            }
            const float* dc = Dl + (16 * it + 4 * g) * RS + r16;                   // dO^T[d = r16][query 4 g + r]
            const float* qc = Ql + (16 * it + 4 * g) * RS + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                DTS_MMA(dv, (r16 < HD ? dc[r * RS] : 0.0f), pr[r]);
                DTS_MMA(dk, (r16 < HD ? qc[r * RS] : 0.0f), ds[r]);
            }
        }
        if (jl && 4 * g < HD) {
            *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Lk + j) * p.ldout + h * p.hs + 4 * g) = dk;
            *reinterpret_cast<f32x4*>(p.out2 + ((long long)b * p.Lk + j) * p.ldout2 + h * p.hs + 4 * g) = dv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fused MLP of a block:  Y = X + W2 gelu(W0 LayerNorm(X) + b0) + b2   (transformer.py:227-232, 238).  The hidden activation
// of a wave's 16 tokens stays in accumulator registers (M/16 tiles x 4) and is the B operand of the second product as it
// stands; it goes to HBM only when the backward pass will need it (U != null).  D = 16*DT, M = 16*MT.
// ---------------------------------------------------------------------------------------------------------------------
struct MlpArgs {
    const float *X, *W0, *b0, *W2, *b2, *lnw, *lnb;
    float* Y;
    float* U;                       // [rows][M] pre-GELU hidden, or null
    long long rows;
};

constexpr int MLP_WAVES = 8;             // both weight matrices fill the LDS of a CU: one workgroup per CU, so it carries two waves per SIMD

template <int DT, int MT>
__global__ __launch_bounds__(64 * MLP_WAVES) void dts_mlp_fwd_kernel(const MlpArgs p) {
    constexpr int D = 16 * DT, M = 16 * MT, KS = D + 4, MS = M + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W0l = lds;                     // [M][KS]
    float* W2l = W0l + M * KS;            // [D][MS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < M * (D / 4); i += 64 * MLP_WAVES) {
        const int n = i / (D / 4), c = i - n * (D / 4);
        *reinterpret_cast<f32x4*>(W0l + n * KS + 4 * c) = *reinterpret_cast<const f32x4*>(p.W0 + (size_t)n * D + 4 * c);
    }
    for (int i = tid; i < D * (M / 4); i += 64 * MLP_WAVES) {
        const int n = i / (M / 4), c = i - n * (M / 4);
        *reinterpret_cast<f32x4*>(W2l + n * MS + 4 * c) = *reinterpret_cast<const f32x4*>(p.W2 + (size_t)n * M + 4 * c);
    }
    __syncthreads();
    // Every wave walks over its own 16-token tiles, nothing else goes through LDS: the tile's rows are loaded straight into the B-operand
    // layout (lane = token r16, k-slot g: X[tok][16 s + 4 g ..]), the LayerNorm statistics of a token are combined over its four lanes with
    // two shuffles, and no barrier separates the waves -- one wave's GELU (vector pipe) runs under another wave's MFMAs.
    f32x4 lw[DT], lb[DT];
#pragma unroll
    for (int s = 0; s < DT; ++s) {
        lw[s] = *reinterpret_cast<const f32x4*>(p.lnw + 16 * s + 4 * g);
        lb[s] = *reinterpret_cast<const f32x4*>(p.lnb + 16 * s + 4 * g);
    }
    const long long n_tiles = (p.rows + 15) / 16;
    for (long long tile = (long long)blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * MLP_WAVES) {
        asm volatile("" ::: "memory");        // the weight fragments are re-read from LDS every tile (hoisted out of the loop they need 320 registers)
        const long long tok = tile * 16 + r16;
        const bool live = tok < p.rows;
        const float* xr = p.X + (live ? tok : 0) * (long long)D + 4 * g;
        f32x4 xb[DT];
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s < DT; ++s) {
            xb[s] = live ? *reinterpret_cast<const f32x4*>(xr + 16 * s) : f32x4{0, 0, 0, 0};
            sum += (xb[s][0] + xb[s][1]) + (xb[s][2] + xb[s][3]);
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum / (float)D;
        float q = 0.0f;
#pragma unroll
        for (int s = 0; s < DT; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xb[s][e] -= mean;
                q += xb[s][e] * xb[s][e];
            }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
        for (int s = 0; s < DT; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) xb[s][e] = xb[s][e] * rstd * lw[s][e] + lb[s][e];
        // hidden units in blocks of HB tiles, each consumed by the second product before the next is formed (see dts_mlp_bwd_kernel)
        constexpr int HB = MT < 8 ? MT : 8;
        f32x4 out[DT];
#pragma unroll
        for (int mt = 0; mt < DT; ++mt) out[mt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int hb = 0; hb < MT; hb += HB) {
            f32x4 u[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int nt = hb + j;
                f32x4 acc = *reinterpret_cast<const f32x4*>(p.b0 + 16 * nt + 4 * g);
                const float* wa = W0l + (16 * nt + r16) * KS + 4 * g;
#pragma unroll
                for (int s = 0; s < DT; ++s) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(wa + 16 * s);
#pragma unroll
                    for (int e = 0; e < 4; ++e) DTS_MMA(acc, a[e], xb[s][e]);
                }
                if (p.U && live) *reinterpret_cast<f32x4*>(p.U + tok * (long long)M + 16 * nt + 4 * g) = acc;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = gelu_f(acc[e]);
                u[j] = acc;
            }
#pragma unroll
            for (int j = 0; j < HB; ++j)                              // DT independent accumulator chains side by side
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mt = 0; mt < DT; ++mt)
                        DTS_MMA(out[mt], W2l[(16 * mt + r16) * MS + 16 * (hb + j) + 4 * g + r], u[j][r]);
        }
        if (live) {
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) {
                const int n0 = 16 * mt + 4 * g;
                f32x4 v = out[mt] + *reinterpret_cast<const f32x4*>(p.b2 + n0);
                v += *reinterpret_cast<const f32x4*>(p.X + tok * (long long)D + n0);
                *reinterpret_cast<f32x4*>(p.Y + tok * (long long)D + n0) = v;
            }
        }
    }
}

// The backward of the same MLP w.r.t. its LayerNorm output, fused the same way:  d_ln = W0^T (gelu'(u) * (W2^T g)).  A wave's 16 tokens: the
// gradient rows are loaded in B-operand layout, W2^T g lands in accumulator layout [hidden unit = 16 nt + 4 (lane>>4) + reg][token], is
// multiplied by gelu'(u) (u read in that same layout) and is the B operand of the second product as it stands.  The 256-wide hidden gradient never
// reaches HBM (the two-GEMM path writes and re-reads 656 MB per call).  W2T = W2 transposed [M][D], W0T = W0 transposed [D][M].
struct MlpBwdArgs {
    const float *G, *U, *W2T, *W0T;
    float* Y;                       // [rows][D]
    long long rows;
    // dts_mlp_bwd_h_kernel only: finish the input gradient of the block's LayerNorm (dts_ln_bwd_kernel fused): lnx = the LayerNorm's input rows,
    // lng = its weight; Y then ACCUMULATES rstd (g' - mean(g') - xhat mean(g' xhat)), g' = d_ln x weight (Y may be G itself).  null: Y = d_ln.
    const float* lnx = nullptr; const float* lng = nullptr;
};

template <int DT, int MT>
__global__ __launch_bounds__(64 * MLP_WAVES) void dts_mlp_bwd_kernel(const MlpBwdArgs p) {
    constexpr int D = 16 * DT, M = 16 * MT, KS = D + 4, MS = M + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W2l = lds;                     // [M][KS]   W2^T
    float* W0l = W2l + M * KS;            // [D][MS]   W0^T
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < M * (D / 4); i += 64 * MLP_WAVES) {
        const int n = i / (D / 4), c = i - n * (D / 4);
        *reinterpret_cast<f32x4*>(W2l + n * KS + 4 * c) = *reinterpret_cast<const f32x4*>(p.W2T + (size_t)n * D + 4 * c);
    }
    for (int i = tid; i < D * (M / 4); i += 64 * MLP_WAVES) {
        const int n = i / (M / 4), c = i - n * (M / 4);
        *reinterpret_cast<f32x4*>(W0l + n * MS + 4 * c) = *reinterpret_cast<const f32x4*>(p.W0T + (size_t)n * M + 4 * c);
    }
    __syncthreads();
    const long long n_tiles = (p.rows + 15) / 16;
    for (long long tile = (long long)blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * MLP_WAVES) {
        asm volatile("" ::: "memory");        // keep the weight fragments in LDS, not in 320 hoisted registers
        const long long tok = tile * 16 + r16;
        const bool live = tok < p.rows;
        const float* gr = p.G + (live ? tok : 0) * (long long)D + 4 * g;
        f32x4 gb[DT];
#pragma unroll
        for (int s = 0; s < DT; ++s) gb[s] = live ? *reinterpret_cast<const f32x4*>(gr + 16 * s) : f32x4{0, 0, 0, 0};
        // the hidden units in blocks of HB tiles: a block's gradient (HB x 4 registers) is formed, multiplied by gelu'(u) and consumed by the
        // second product before the next block is started -- 64 + 16 live accumulator registers would spill at M = 256
        constexpr int HB = MT < 8 ? MT : 8;
        f32x4 out[DT];
#pragma unroll
        for (int mt = 0; mt < DT; ++mt) out[mt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int hb = 0; hb < MT; hb += HB) {
            f32x4 e[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int nt = hb + j;
                f32x4 acc = {0, 0, 0, 0};
                const float* wa = W2l + (16 * nt + r16) * KS + 4 * g;
#pragma unroll
                for (int s = 0; s < DT; ++s) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(wa + 16 * s);
#pragma unroll
                    for (int q = 0; q < 4; ++q) DTS_MMA(acc, a[q], gb[s][q]);
                }
                const f32x4 u = live ? *reinterpret_cast<const f32x4*>(p.U + tok * (long long)M + 16 * nt + 4 * g) : f32x4{0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] *= gelu_grad_f(u[q]);
                e[j] = acc;
            }
#pragma unroll
            for (int j = 0; j < HB; ++j)                              // DT independent accumulator chains side by side
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mt = 0; mt < DT; ++mt)
                        DTS_MMA(out[mt], W0l[(16 * mt + r16) * MS + 16 * (hb + j) + 4 * g + r], e[j][r]);
        }
        if (live) {
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) *reinterpret_cast<f32x4*>(p.Y + tok * (long long)D + 16 * mt + 4 * g) = out[mt];
        }
    }
}

}  // namespace dts
