// DiffSTG on gfx950: UGnet epsilon-network + DDIM / DDPM posterior sampler (reference models/Diffusion_model/DiffSTG/ugnet.py:55-295,
// graph_diffusion_model.py:46-101, 193-282), behind the mpv_stg_* entry points of include/mpv_hip.h.
//
// Activations are fp32 [N][c][T'] (N = windows x parallel samples x graph nodes; channels x time of one node contiguous).
// Per ResidualBlock (ugnet.py:115-135):
//   stg_tcn_ln_kernel   one workgroup per node row, everything in LDS: causal k=3 TCN + 1x1 skip, time-embedding add, second TCN,
//                       LayerNorm over channels.  (The Conv2d kernels are 3x3 over a height-1 axis: only their middle row acts.)
//   stg_gemm_kernel     fp32 MFMA (v_mfma_f32_16x16x4_f32) GEMM over all rows: the k = T+1 "downsampling" convolution as
//                       [Td_h*c] x [c*T] (its 10 output taps are 10 shifted copies of the kernel, built at load time), the four
//                       Linear maps of the gated graph convolution as one [4C] x [C] product, the transposed k = T+1 convolution
//                       as [c*T] x [Td_h*c], and the final Linear(2T -> T).
//   stg_gate_kernel     ResGatedGraphConv aggregation: CSR gather over the incoming edges of each node, sigmoid gate, skip, bias,
//                       ReLU.  The graph is stored once; sample copy p of window w uses node offset (w*P + p)*V (the reference
//                       duplicates the edge index instead, graph_diffusion_model.py:77-84).
//   stg_res_out_kernel  + shortcut (1x1 conv or identity) of the block input.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>

#include "chain_common.h"
#include "nst_gemm.h"
#include "stg_tcn_mfma.h"
#include "stg_down_mfma.h"
#include "graph_aggr.h"

namespace {

#define STG_MMA(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (acc), 0, 0, 0)

// ---- general fp32 MFMA GEMM: Y[rows][N] (+)= X[rows][K] W[N][K]^T + bias[N]; K, N multiples of 4 -----------------------------
struct GemmArgs {
    const float* X; long long ldx;
    const float* W;                 // [N][K]
    const float* bias;              // [N] | null
    float* Y; long long ldy;
    long long rows;
    int K, N, KC;                   // KC: K chunk staged per iteration (multiple of 16)
    int relu;
    const float* r1;                // optional residual [rows][N] added in the epilogue (a ResidualBlock's shortcut) | null
};
constexpr int GEMM_NT = 10;         // 16-wide column tiles per workgroup (N tile = 160)

__global__ __launch_bounds__(256) void stg_gemm_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int KS = p.KC + 4;
    float* Wl = lds;                          // [160][KS]
    float* Xl = lds + 16 * GEMM_NT * KS;      // [64][KS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const long long tok0 = (long long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 16 * GEMM_NT;
    f32x4 acc[GEMM_NT];
#pragma unroll
    for (int i = 0; i < GEMM_NT; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const int C4 = p.KC >> 2;
    for (int k0 = 0; k0 < p.K; k0 += p.KC) {
        __syncthreads();
        for (int i = tid; i < 16 * GEMM_NT * C4; i += 256) {
            const int n = i / C4, c = i - n * C4;
            const bool ok = (n0 + n) < p.N && (k0 + 4 * c) < p.K;
            *reinterpret_cast<f32x4*>(Wl + n * KS + 4 * c) = ok ? *reinterpret_cast<const f32x4*>(p.W + (long long)(n0 + n) * p.K + k0 + 4 * c) : f32x4{0, 0, 0, 0};
        }
        for (int i = tid; i < 64 * C4; i += 256) {
            const int r = i / C4, c = i - r * C4;
            const bool ok = (tok0 + r) < p.rows && (k0 + 4 * c) < p.K;
            *reinterpret_cast<f32x4*>(Xl + r * KS + 4 * c) = ok ? *reinterpret_cast<const f32x4*>(p.X + (tok0 + r) * p.ldx + k0 + 4 * c) : f32x4{0, 0, 0, 0};
        }
        __syncthreads();
        const float* xb = Xl + (16 * wave + r16) * KS + 4 * g;
        for (int s = 0; s < (p.KC >> 4); ++s) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(xb + 16 * s);
#pragma unroll
            for (int nt = 0; nt < GEMM_NT; ++nt) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(Wl + (16 * nt + r16) * KS + 16 * s + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) STG_MMA(acc[nt], a[e], b[e]);
            }
        }
    }
    const long long tok = tok0 + 16 * wave + r16;
    if (tok < p.rows) {
#pragma unroll
        for (int nt = 0; nt < GEMM_NT; ++nt) {
            const int n = n0 + 16 * nt + 4 * g;
            if (n < p.N) {
                f32x4 v = acc[nt];
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                }
                if (p.r1) v += *reinterpret_cast<const f32x4*>(p.r1 + tok * p.ldy + n);       // + shortcut(x) of the block input, [rows][N]
                *reinterpret_cast<f32x4*>(p.Y + tok * p.ldy + n) = v;
            }
        }
    }
}

// ---- per-row TCN + time embedding + TCN + LayerNorm -----------------------------------------------------------------------------
struct TcnArgs {
    const float* src1; int c1;      // [N][c1][T]
    const float* src2; int c2;      // concatenated skip [N][c2][T] | null
    const float *w1, *b1;           // [co][ci][3] (middle row of the 3x3 kernel), [co]
    const float *ws1, *bs1;         // tcn1.shortcut [co][ci], [co] | null (identity)
    const float *wt, *bt;           // t_conv [co][d_h], [co]
    const float* temb;              // [d_h] sinusoidal embedding of this step's t
    const float *w2, *b2;           // tcn2 [co][co][3], [co]
    const float *lnw, *lnb;         // [co]
    float* out;                     // [N][co][T]
    int co, T, d_h;
    const float *wsb, *bsb;         // the BLOCK's shortcut 1x1 conv [co][ci], [co] | null (identity)
    float* sc;                      // [N][co][T] shortcut(x), written when wsb
    long long rows;                 // N
    int tseg, nseg;                 // time steps per workgroup item, items per row (nseg * tseg >= T)
};

// (A matrix-pipe version of this block -- each causal convolution as a 16-channel x 16-step MFMA tile GEMM over (channel, tap) pairs, B
// operand read straight out of the padded row in LDS -- was built twice and measured: 3.6 and 3.98 windows/s at C5 against 4.65 and 5.33 for this kernel.)
// A thread owns one time step and ALL output channels (CO accumulators): an input sample is read from LDS once per tap instead of
// once per output channel, the weights are staged in LDS as [k][c][4] so that one broadcast ds_read_b128 feeds four FMAs, and the
// LayerNorm over the channels of a time step is register-local.  When the block's shortcut is a 1x1 convolution it is formed here
// too (the block input is in LDS anyway) and written to `sc` for the epilogue of the up-sampling product.
template <int CO>
__global__ __launch_bounds__(256) void stg_tcn_ln_kernel(const TcnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // A workgroup handles one SEGMENT of a row's time axis (p.tseg steps, p.nseg segments per row) with a causal halo: tcn2 at step t reads
    // h1[t-2..t] and h1 at step t reads x[t-2..t], so a segment [t0, t1) stages x[t0-4, t1) and forms h1 on [t0-2, t1).  Halving the
    // segment halves the LDS footprint: the concatenating blocks (ci = 2 co) go from one to two or three workgroups per CU.
    const int ci = p.c1 + p.c2, T = p.T, XP = p.tseg + 4, HP = p.tseg + 2;
    float* W1 = sm;                           // [ci][CO][4] = (w1 tap0, tap1, tap2, tcn1.shortcut)
    float* W2 = W1 + ci * CO * 4;             // [CO][CO][4] = (w2 tap0, tap1, tap2, 0)
    float* WB = W2 + CO * CO * 4;             // [ci][CO]     block shortcut (1x1 conv) | unused
    float* xs = WB + ci * CO;                 // [ci][tseg+4]: x at steps t0-4 .. t1-1 (zeros before the row starts)
    float* h1 = xs + ci * XP;                 // [CO][tseg+2]: h1 at steps t0-2 .. t1-1 (zeros before the row starts: tcn2's causal padding)
    float* tvs = h1 + CO * HP;                // [CO] time-embedding projection of this step (the same for every row)
    for (int i = threadIdx.x; i < ci * CO; i += 256) {
        const int k = i / CO, c = i - k * CO;
        const float* w = p.w1 + (c * ci + k) * 3;
        *reinterpret_cast<f32x4*>(W1 + i * 4) = f32x4{w[0], w[1], w[2], p.ws1 ? p.ws1[c * ci + k] : 0.0f};
        WB[i] = p.wsb ? p.wsb[c * ci + k] : 0.0f;
    }
    for (int i = threadIdx.x; i < CO * CO; i += 256) {
        const int k = i / CO, c = i - k * CO;
        const float* w = p.w2 + (c * CO + k) * 3;
        *reinterpret_cast<f32x4*>(W2 + i * 4) = f32x4{w[0], w[1], w[2], 0.0f};
    }
    for (int c = threadIdx.x; c < CO; c += 256) {
        float a = p.bt[c];
        for (int d = 0; d < p.d_h; ++d) a += p.wt[c * p.d_h + d] * p.temb[d];
        tvs[c] = a;
    }
    // weights and the time embedding are staged once; the workgroup then walks over its (row, segment) items
    for (long long item = blockIdx.x; item < p.rows * p.nseg; item += gridDim.x) {
        const long long n = item / p.nseg;
        const int t0 = (int)(item - n * p.nseg) * p.tseg, t1 = min(T, t0 + p.tseg), tl = t1 - t0;
        __syncthreads();                      // (first item: the staging above; later items: the previous one is done with xs / h1)
        for (int i = threadIdx.x; i < ci * XP; i += 256) {
            const int c = i / XP, t = t0 - 4 + (i - c * XP);
            float v = 0.0f;
            if (t >= 0 && t < t1) v = c < p.c1 ? p.src1[(n * p.c1 + c) * T + t] : p.src2[(n * p.c2 + (c - p.c1)) * T + t];
            xs[i] = v;
        }
        __syncthreads();
        float tv[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) tv[c] = tvs[c];
        for (int u = threadIdx.x; u < tl + 2; u += 256) {            // h1 at step t = t0 - 2 + u from x at local columns u .. u+2
            const int t = t0 - 2 + u;
            if (t < 0) {
#pragma unroll
                for (int c = 0; c < CO; ++c) h1[c * HP + u] = 0.0f;
                continue;
            }
            float acc[CO], sk[CO], sb[CO];
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                acc[c] = p.b1[c];
                sk[c] = p.ws1 ? p.bs1[c] : xs[c * XP + u + 2];
                sb[c] = p.wsb ? p.bsb[c] : 0.0f;
            }
            if (!p.ws1 && !p.wsb) {              // identity shortcuts (most blocks): the convolution alone
#pragma nounroll
                for (int k = 0; k < ci; ++k) {
                    const float x0 = xs[k * XP + u], x1 = xs[k * XP + u + 1], x2 = xs[k * XP + u + 2];
#pragma unroll
                    for (int c = 0; c < CO; ++c) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(W1 + (k * CO + c) * 4);
                        acc[c] += w[0] * x0 + w[1] * x1 + w[2] * x2;
                    }
                }
            } else {
#pragma nounroll
                for (int k = 0; k < ci; ++k) {
                    const float x0 = xs[k * XP + u], x1 = xs[k * XP + u + 1], x2 = xs[k * XP + u + 2];
#pragma unroll
                    for (int c = 0; c < CO; ++c) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(W1 + (k * CO + c) * 4);
                        acc[c] += w[0] * x0 + w[1] * x1 + w[2] * x2;
                        sk[c] += w[3] * x2;
                        sb[c] += WB[k * CO + c] * x2;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                h1[c * HP + u] = (acc[c] + sk[c]) + tv[c];
                if (p.wsb && t >= t0) p.sc[(n * CO + c) * T + t] = sb[c];          // (halo steps belong to the previous segment)
            }
        }
        __syncthreads();
        for (int v = threadIdx.x; v < tl; v += 256) {                 // output step t = t0 + v from h1 at local columns v .. v+2
            const int t = t0 + v;
            float acc[CO];
#pragma unroll
            for (int c = 0; c < CO; ++c) acc[c] = p.b2[c];
#pragma nounroll
            for (int k = 0; k < CO; ++k) {
                const float x0 = h1[k * HP + v], x1 = h1[k * HP + v + 1], x2 = h1[k * HP + v + 2];
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(W2 + (k * CO + c) * 4);
                    acc[c] += w[0] * x0 + w[1] * x1 + w[2] * x2;
                }
            }
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                acc[c] += h1[c * HP + v + 2];
                s += acc[c];
            }
            const float mean = s / (float)CO;                        // LayerNorm([1, co]) over the channels of this time step
            float q = 0.0f;
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                acc[c] -= mean;
                q += acc[c] * acc[c];
            }
            const float rstd = 1.0f / sqrtf(q / (float)CO + 1e-5f);
#pragma unroll
            for (int c = 0; c < CO; ++c) p.out[(n * CO + c) * T + t] = acc[c] * rstd * p.lnw[c] + p.lnb[c];
        }
    }                                         // items
}

// ---- ResGatedGraphConv aggregation + ReLU: kqvs [N][4C] = (key | query | value | skip) -> out [N][C] ------------------------------
__global__ void stg_gate_kernel(const float* __restrict__ kqvs, const float* __restrict__ bias, const int* __restrict__ row_ptr,
                                const int* __restrict__ col_idx, int V, int C, long long n, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // node*C + ch
    if (i >= n) return;
    const int ch = (int)(i % C);
    const long long node = i / C;
    const int v = (int)(node % V);
    const long long base = node - v;                                            // first node of this graph copy
    const float k = kqvs[node * 4 * C + ch];
    float a = 0.0f;
    for (int e = row_ptr[v]; e < row_ptr[v + 1]; ++e) {
        const float* src = kqvs + (base + col_idx[e]) * 4 * C;
        const float gate = 1.0f / (1.0f + __expf(-(k + src[C + ch])));
        a += gate * src[2 * C + ch];
    }
    a = (a + kqvs[node * 4 * C + 3 * C + ch]) + bias[ch];
    out[i] = fmaxf(a, 0.0f);
}

// ---- block output: up [N][co*T] + shortcut(x) ---------------------------------------------------------------------------------------
__global__ void stg_res_out_kernel(const float* __restrict__ up, const float* __restrict__ src1, int c1, const float* __restrict__ src2, int c2,
                                   const float* __restrict__ ws, const float* __restrict__ bs, int co, int T, long long n, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // (node*co + c)*T + t
    if (i >= n) return;
    const int t = (int)(i % T);
    const int c = (int)((i / T) % co);
    const long long node = i / ((long long)T * co);
    float s;
    if (ws) {
        s = bs[c];
        for (int k = 0; k < c1; ++k) s += ws[c * (c1 + c2) + k] * src1[(node * c1 + k) * T + t];
        for (int k = 0; k < c2; ++k) s += ws[c * (c1 + c2) + c1 + k] * src2[(node * c2 + k) * T + t];
    } else {
        s = src1[(node * c1 + c) * T + t];
    }
    out[i] = up[i] + s;
}

// ---- Downsample: Conv2d(c, c, (1,3), stride 2, pad 1); Upsample: ConvTranspose2d(c, c, (1,4), stride 2, pad 1) ---------------
// Resampling convolutions of the U-Net (Downsample: Conv1d(c, c, 3, stride 2, pad 1); Upsample: ConvTranspose1d(c, c, 4, stride 2, pad 1)), one
// node row per pass of a persistent workgroup: the row is staged in LDS once, a thread owns one output step and ALL output channels (an input
// sample is read once per tap, not once per output channel), weights sit in LDS as [ci][tap][co] so one broadcast ds_read_b128 feeds 4 FMAs.
template <int C, bool UP>
__global__ __launch_bounds__(256) void stg_resample_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, int T, int To,
                                                           long long rows, float* __restrict__ out) {
    constexpr int KT = UP ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* W = sm;                              // [C][KT][C]
    float* xs = W + C * KT * C;                 // [C][T + 2], one zero on each side
    const int TP = T + 2, tid = threadIdx.x;
    for (int i = tid; i < C * KT * C; i += 256) {
        const int ci = i / (KT * C), k = (i / C) % KT, co = i % C;
        W[i] = UP ? w[(ci * C + co) * 4 + k] : w[(co * C + ci) * 3 + k];       // ConvTranspose weights are [in][out][k]
    }
    for (int c = tid; c < C; c += 256) xs[c * TP] = xs[c * TP + T + 1] = 0.0f;
    for (long long n = blockIdx.x; n < rows; n += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < C * T; i += 256) {
            const int c = i / T, t = i - c * T;
            xs[c * TP + 1 + t] = x[n * C * T + i];
        }
        __syncthreads();
        for (int t = tid; t < To; t += 256) {
            float acc[C];
#pragma unroll
            for (int co = 0; co < C; ++co) acc[co] = b[co];
            if (UP) {                            // t = 2 j + k - 1: even t takes taps (1, j = t/2) and (3, j = t/2 - 1); odd t taps (0, (t+1)/2) and (2, (t-1)/2)
                const int ka = (t & 1) ? 0 : 1, ja = (t & 1) ? (t + 1) >> 1 : t >> 1, kb = ka + 2, jb = ja - 1;
#pragma nounroll
                for (int ci = 0; ci < C; ++ci) {
                    const float xa = xs[ci * TP + 1 + ja], xb = xs[ci * TP + 1 + jb];     // ja == T and jb == -1 read the zero pads
                    const float* wa = W + (ci * KT + ka) * C;
                    const float* wb = W + (ci * KT + kb) * C;
#pragma unroll
                    for (int co = 0; co < C; co += 4) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(wa + co), b4 = *reinterpret_cast<const f32x4*>(wb + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[co + e] += a4[e] * xa + b4[e] * xb;
                    }
                }
            } else {
#pragma nounroll
                for (int ci = 0; ci < C; ++ci) {
                    const float x0 = xs[ci * TP + 2 * t], x1 = xs[ci * TP + 2 * t + 1], x2 = xs[ci * TP + 2 * t + 2];      // input steps 2t-1, 2t, 2t+1 (pads are zero)
                    const float* w0 = W + ci * KT * C;
#pragma unroll
                    for (int co = 0; co < C; co += 4) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(w0 + co), b4 = *reinterpret_cast<const f32x4*>(w0 + C + co),
                                    c4 = *reinterpret_cast<const f32x4*>(w0 + 2 * C + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[co + e] += (a4[e] * x0 + b4[e] * x1) + c4[e] * x2;
                    }
                }
            }
#pragma unroll
            for (int co = 0; co < C; ++co) out[(n * C + co) * To + t] = acc[co];
        }
    }
}

__global__ void stg_down_kernel(const float* __restrict__ x, const float* __restrict__ w /*[c][c][3]*/, const float* __restrict__ b, int c, int T, int To,
                                long long n, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // (node*c + co)*To + t
    if (i >= n) return;
    const int t = (int)(i % To);
    const int co = (int)((i / To) % c);
    const long long node = i / ((long long)To * c);
    float a = b[co];
    for (int ci = 0; ci < c; ++ci)
        for (int k = 0; k < 3; ++k) {
            const int ti = 2 * t + k - 1;
            if (ti >= 0 && ti < T) a += w[(co * c + ci) * 3 + k] * x[(node * c + ci) * T + ti];
        }
    out[i] = a;
}

__global__ void stg_up_kernel(const float* __restrict__ x, const float* __restrict__ w /*[ci][co][4]*/, const float* __restrict__ b, int c, int T, int To,
                              long long n, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // (node*c + co)*To + t
    if (i >= n) return;
    const int t = (int)(i % To);
    const int co = (int)((i / To) % c);
    const long long node = i / ((long long)To * c);
    float a = b[co];
    for (int k = 0; k < 4; ++k) {
        const int num = t - k + 1;                                              // t = 2 j + k - 1
        if (num < 0 || (num & 1)) continue;
        const int j = num >> 1;
        if (j >= T) continue;
        for (int ci = 0; ci < c; ++ci) a += w[(ci * c + co) * 4 + k] * x[(node * c + ci) * T + j];
    }
    out[i] = a;
}

// ---- input projection (ugnet.py:271-277) and output 1x1 conv (:292) -----------------------------------------------------------------
__global__ void stg_xproj_kernel(const float* __restrict__ xt, const float* __restrict__ xm, const float* __restrict__ w /*[d_h][F]*/,
                                 const float* __restrict__ b, int T, int F, int d_h, long long n, float* out /*[N][d_h][2T]*/) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = (int)(i % (2 * T));
    const int c = (int)((i / (2 * T)) % d_h);
    const long long node = i / ((long long)2 * T * d_h);
    const float* src = t < T ? xt + (node * T + t) * F : xm + (node * T + (t - T)) * F;
    float a = b[c];
    for (int f = 0; f < F; ++f) a += w[c * F + f] * src[f];
    out[i] = a;
}

__global__ void stg_outconv_kernel(const float* __restrict__ x /*[N][d_h][T2]*/, const float* __restrict__ w /*[F][d_h]*/, const float* __restrict__ b,
                                   int T2, int F, int d_h, long long n, float* out /*[N][F][T2]*/) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = (int)(i % T2);
    const int f = (int)((i / T2) % F);
    const long long node = i / ((long long)T2 * F);
    float a = b[f];
    for (int c = 0; c < d_h; ++c) a += w[f * d_h + c] * x[(node * d_h + c) * T2 + t];
    out[i] = a;
}

// eps [N][F][T] -> posterior update of xt [N][T][F] (graph_diffusion_model.py:46-73); the coefficients are the reference's Python
// floats rounded to fp32 by the scalar-tensor product
__global__ void stg_posterior_kernel(const float* __restrict__ eps, const float* __restrict__ z, float c1, float c2, float c3, int ddpm, int T, int F,
                                     long long n, float* xt) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // (node*T + t)*F + f
    if (i >= n) return;
    const int f = (int)(i % F);
    const int t = (int)((i / F) % T);
    const long long node = i / ((long long)F * T);
    const float pred = eps[(node * F + f) * T + t];
    const float a = c1 * (xt[i] - c2 * pred);
    xt[i] = ddpm ? a + c3 * z[i] : a + c3 * pred;
}

// eps [N][F][T] (the network's channel-major output) -> [N][T][F] (the layout of x_t and of the reference's return value)
__global__ void stg_eps_transpose_kernel(const float* __restrict__ eps, int T, int F, long long n, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // (node*T + t)*F + f
    if (i >= n) return;
    const int f = (int)(i % F);
    const int t = (int)((i / F) % T);
    const long long node = i / ((long long)F * T);
    out[i] = eps[(node * F + f) * T + t];
}

// x_masked rows: r = (w*P + p)*V + v  <-  history[w][v][t][f] for t < T_h, 0 after
__global__ void stg_build_masked_kernel(const float* __restrict__ hist, int V, int P, int T, int T_h, int F, long long n, float* xm) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = (int)(i % F);
    const int t = (int)((i / F) % T);
    const long long r = i / ((long long)F * T);
    const int v = (int)(r % V);
    const long long w = r / ((long long)V * P);
    xm[i] = t < T_h ? hist[((w * V + v) * T_h + t) * F + f] : 0.0f;
}

// xt [W*P*V][T][F] of sequential pass s -> samples[w][v][t][f][s*P + p]
__global__ void stg_gather_kernel(const float* __restrict__ xt, int V, int P, int T, int F, int S_total, int s_pass, long long n, float* samples) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long r = i / ((long long)F * T);
    const long long tf = i - r * (long long)F * T;
    const int v = (int)(r % V);
    const int pcopy = (int)((r / V) % P);
    const long long w = r / ((long long)V * P);
    samples[((w * V + v) * (long long)T * F + tf) * S_total + s_pass * P + pcopy] = xt[i];
}

__global__ void stg_philox_kernel(unsigned seed_lo, unsigned seed_hi, unsigned stream, long long n, float* out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * q >= n) return;
    float z[4];
    chain::philox_normals((unsigned long long)q, stream, seed_lo, seed_hi, 0x53544700u, z);
    for (int e = 0; e < 4 && 4 * q + e < n; ++e) out[4 * q + e] = z[e];
}

// ---- NsDiff_spatial (NsDiff/ugnet.py:257-292, NsDiff/nsdiff_utils.py:111-158, 209-239, 271-284) ----------------------------------------
// x_proj over cat(y_t, y_0_hat, g) on the feature axis: out [N][d_h][T] from three [N][T][F] tensors, w [d_h][3F]
__global__ void nss_xproj_kernel(const float* __restrict__ y, const float* __restrict__ mu, const float* __restrict__ g, const float* __restrict__ w,
                                 const float* __restrict__ b, int T, int F, int d_h, long long n, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // (node*d_h + c)*T + t
    if (i >= n) return;
    const int t = (int)(i % T);
    const int c = (int)((i / T) % d_h);
    const long long at = ((i / ((long long)T * d_h)) * T + t) * F;
    float a = b[c];
    for (int f = 0; f < F; ++f) a += w[c * 3 * F + f] * y[at + f];
    for (int f = 0; f < F; ++f) a += w[c * 3 * F + F + f] * mu[at + f];
    for (int f = 0; f < F; ++f) a += w[c * 3 * F + 2 * F + f] * g[at + f];
    out[i] = a;
}

// the two heads over the d_h channels of a (row, time) position: eps = lin4(h), sigma = softplus(sigma_lin(softplus(h)))   (:291)
__global__ void nss_heads_kernel(const float* __restrict__ h /*[N][d_h][T]*/, const float* __restrict__ w4, const float* __restrict__ b4,
                                 const float* __restrict__ wsg, const float* __restrict__ bsg, int T, int F, int d_h, long long n /*N*T*/,
                                 float* eps /*[N][T][F]*/, float* sig) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // node*T + t
    if (i >= n) return;
    const int t = (int)(i % T);
    const long long node = i / T;
    for (int f = 0; f < F; ++f) {
        float a = b4[f], q = bsg[f];
        for (int c = 0; c < d_h; ++c) {
            const float v = h[(node * d_h + c) * T + t];
            a += w4[f * d_h + c] * v;
            q += wsg[f * d_h + c] * chain::softplus_dev(v);
        }
        eps[i * F + f] = a;
        sig[i * F + f] = chain::softplus_dev(q);
    }
}

// rows of one pass: R = w*(P*V) + r, r = v*P + p  (node-major tiling of NsDiff_model.py:756-763)  <-  src [W][V][O][F]
__global__ void nss_tile_kernel(const float* __restrict__ src, int V, int P, long long OF, long long n, float* dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long R = i / OF, rem = i - R * OF;
    const long long w = R / ((long long)P * V);
    const int v = (int)((R - w * P * V) / P);
    dst[i] = src[(w * V + v) * OF + rem];
}

// draw k of chunk c for the rows of one pass: injected (the reference's stream, [W][C][T][P*V][O][F]) or Philox keyed by the element's
// position in [W][C][P*V][O][F] with the draw index as the stream
// (`inj` points at the block's first window, `w0` is that window's index in the call: Philox keys do not depend on the window blocking)
__global__ void nss_noise_kernel(const float* __restrict__ inj, int C, int c, int Tn, int k, long long PV, long long OF, unsigned seed_lo, unsigned seed_hi,
                                 long long w0, long long n, float* z) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // (w*PV + r)*OF + rem
    if (i >= n) return;
    const long long per = PV * OF, w = i / per, rest = i - w * per;
    if (inj) {
        z[i] = inj[((w * C + c) * Tn + k) * per + rest];
        return;
    }
    const unsigned long long id = (unsigned long long)(((w0 + w) * C + c) * per + rest);
    float zz[4];
    chain::philox_normals(id >> 2, (unsigned)k, seed_lo, seed_hi, 0x4E535300u, zz);
    const unsigned e = (unsigned)(id & 3ull);
    z[i] = e == 0u ? zz[0] : (e == 1u ? zz[1] : (e == 2u ? zz[2] : zz[3]));
}

// y_T = sqrt(g) z + mu   (p_sample_loop, nsdiff_utils.py:273-274)
__global__ void nss_init_kernel(const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ mu, long long n, float* y) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = sqrtf(g[i]) * z[i] + mu[i];
}

// one reverse step on every element (p_sample / p_sample_t_1to0): the arithmetic of the pointwise sampler (chain_common.h::posterior)
__global__ void nss_posterior_kernel(const float* __restrict__ coef_row, const float* __restrict__ mu, const float* __restrict__ g,
                                     const float* __restrict__ eps, const float* __restrict__ sig, const float* __restrict__ z, int last, long long n,
                                     float* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    chain::StepCoef c;
    float* cf = reinterpret_cast<float*>(&c);
#pragma unroll
    for (int j = 0; j < 18; ++j) cf[j] = coef_row[j];
    y[i] = chain::posterior(c, y[i], mu[i], g[i], eps[i], sig[i], last ? 0.0f : z[i], last != 0);
}

// rows of one pass -> samples [W][V][O][F][S] at sample index c*P + p
__global__ void nss_gather_kernel(const float* __restrict__ y, int V, int P, long long OF, int S, int c, long long n, float* samples) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long R = i / OF, rem = i - R * OF;
    const long long w = R / ((long long)P * V);
    const long long r = R - w * P * V;
    const int v = (int)(r / P), pp = (int)(r - (long long)v * P);
    samples[((w * V + v) * OF + rem) * S + (long long)c * P + pp] = y[i];
}

struct ResW {
    const float *w1, *b1, *ws1, *bs1, *wt, *bt, *w2, *b2, *lnw, *lnb, *wdown, *bdown, *wg, *bg4, *gbias, *wup, *bup, *ws, *bs;
};
struct TcnFrag {                                     // matrix-pipe TCN (stg_tcn_mfma.h): A fragments and scales of one residual block
    const th8 *a1 = nullptr, *a2 = nullptr, *asb = nullptr;
    float inv1 = 1.0f, inv2 = 1.0f, invsb = 1.0f;
    const float* b1s = nullptr;                      // tcn1.conv.bias + tcn1.shortcut.bias
    int cip1 = 0, cip2 = 0, nc1 = 0, nc2 = 0;
    bool ok = false;
};
struct Step {
    int kind = 0, c_in = 0, c_out = 0, T = 0, pop = 0, push = 0;
    int gnn = 0, ng = 0, heads = 1;                  // graph convolution of a residual block: 0 gated, 1 SAGE mean, 2 sum, 3 GCN, 4 GAT; columns of its product
    float slope = 0.2f;
    ResW r{};
    const float *cw = nullptr, *cb = nullptr;        // down / up sample conv
    TcnFrag f{};
};

// A fragments of D[co][t] = sum_k A[co][k] B[k][t] for v_mfma_f32_16x16x32_f16: lane l holds A[row l & 15][k = 32 chunk + 8 (l >> 4) + j], j = 0..7,
// with k = tap * cip + channel (taps = 3: the causal convolution; taps = 1: a 1x1 convolution).  w: [co][ci][taps] fp32.  Appends
// [chunks][hi, lo][64 lanes][8] halves of w * 2^s to `dst` and returns 2^-s.
inline float tcn_fragments(const std::vector<float>& w, int co, int ci, int cip, int taps, int chunks, std::vector<_Float16>& dst) {
    float mx = 0.0f;
    for (float v : w) mx = std::max(mx, std::fabs(v));
    int ex = 0;
    if (mx > 0.0f && std::isfinite(mx)) { (void)std::frexp(mx, &ex); ex = 14 - ex; }
    const float sc = std::ldexp(1.0f, ex);
    const size_t at = dst.size();
    dst.resize(at + (size_t)chunks * 2 * 64 * 8, (_Float16)0.0f);
    for (int c = 0; c < chunks; ++c)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int row = lane & 15, k = 32 * c + 8 * (lane >> 4) + j, tap = k / cip, ch = k - tap * cip;
                if (row >= co || tap >= taps || ch >= ci) continue;
                const float v = w[((size_t)row * ci + ch) * taps + tap] * sc;
                uint32_t bits;
                std::memcpy(&bits, &v, 4);
                bits &= 0xFFFFE000u;
                float hi;
                std::memcpy(&hi, &bits, 4);
                dst[at + ((size_t)(c * 2 + 0) * 64 + lane) * 8 + j] = (_Float16)hi;
                dst[at + ((size_t)(c * 2 + 1) * 64 + lane) * 8 + j] = (_Float16)(v - hi);
            }
    return std::ldexp(1.0f, -ex);
}

inline dim3 g1d(long long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

struct SplitW { const _Float16* h; const _Float16* l; float inv_scale; int ldw; };    // two-term f16 split of one GEMM weight (nst_gemm.h)

struct StgEngine {
    mpv_stg_desc d{};
    _Float16* blob16 = nullptr;
    _Float16* frag16 = nullptr;                         // A fragments of the matrix-pipe TCN blocks
    std::unordered_map<const float*, SplitW> split;     // fp32 weight pointer -> its split copy
    std::vector<Step> steps;
    float* blob = nullptr;
    const float *wx = nullptr, *bx = nullptr, *wo = nullptr, *bo = nullptr, *wl = nullptr, *bl = nullptr, *temb = nullptr;
    int* graph = nullptr;            // row_ptr [V+1] then col_idx [E] (incoming edges by target node)
    int V = 0, E = 0;
    float* ws = nullptr;
    size_t ws_floats = 0;
    long long cap_N = 0;
    std::vector<float*> act;         // one output buffer per step (+ input projection at [0])
    float *hn = nullptr, *z = nullptr, *kqvs = nullptr, *sp = nullptr, *up = nullptr, *e1 = nullptr, *eps = nullptr, *xt = nullptr, *xm = nullptr,
          *noise = nullptr;
    // mode 1 = NsDiff_spatial (NsDiff/ugnet.py:194-292): the same U-Net over cat(y_t, y0hat, g) at time length pred_len, a d_h-wide output head
    // with an eps and a softplus sigma projection, stepped by the NsDiff posterior instead of DDIM
    int mode = 0;
    int T0 = 0;                      // time length at the top resolution: 2 (T_h + T_p) for DiffSTG, pred_len for NsDiff_spatial
    int steps_T = 0;                 // NsDiff_spatial: diffusion steps (rows of coef)
    const float *w4 = nullptr, *b4 = nullptr, *wsg = nullptr, *bsg = nullptr, *coef = nullptr;
    float *sig = nullptr, *e2 = nullptr, *gg = nullptr;
};

void stg_engine_destroy(StgEngine* e) {
    if (!e) return;
    if (e->blob) (void)hipFree(e->blob);
    if (e->blob16) (void)hipFree(e->blob16);
    if (e->frag16) (void)hipFree(e->frag16);
    if (e->ws) (void)hipFree(e->ws);
    if (e->graph) (void)hipFree(e->graph);
    delete e;
}

namespace {

#define STG_TRY(expr)            \
    do {                         \
        if (int rc_ = (expr)) return rc_; \
    } while (0)

constexpr int STG_KSPLIT_MAX = 4;                       // partial buffers carved per row of the down-sampled features (e.z)

// How many ways the k loop of a "downsampling" product [rows][K] x [N][K]^T is split (1 = not at all).  Round 4 measured negative: on the c5 step the
// split products (MPV_STG_KSPLIT=0: as many ways as it takes to give the chip 3 workgroups per CU, at most 4; the graph-conv product sums the partial
// tensors as it loads them, fixed order) average 84.9 us per launch against 75.8 us unsplit on the same box (8.64 against 8.83 windows/s) -- alone on the
// device the long-K products are concurrency-bound (1.6-2.4 TB/s from 200-400 workgroups), inside the step the neighbouring launches' tails fill the
// chip anyway.  Off unless MPV_STG_KSPLIT is set (n >= 2: that many ways; 0: the automatic rule).  (The automatic rule also rounds badly: 200 tiles x 4
// ways = 800 workgroups on 2 x 256 slots is two rounds.  Superseded by the products' own kernel, stg_down_mfma.h, which gemm() picks when unsplit.)
int down_ksplit(mpv_ctx* ctx, const StgEngine& e, const float* W, int K, int N, long long rows) {
    static const int forced = getenv("MPV_STG_KSPLIT") ? atoi(getenv("MPV_STG_KSPLIT")) : 1;
    if (e.d.gemm == MPV_GEMM_F32 || e.split.find(W) == e.split.end() || K % 4 != 0 || N % 4 != 0) return 1;
    if (forced >= 1) return std::min(forced, STG_KSPLIT_MAX);
    const long long tiles = ((rows + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
    int ks = 1;
    while (ks < STG_KSPLIT_MAX && tiles * ks < 3LL * ctx->cu_count && K / (2 * ks) >= 256) ks *= 2;
    return ks;
}

int gemm(mpv_ctx* ctx, const StgEngine& e, const float* X, long long ldx, const float* W, const float* bias, int K, int N, long long rows, float* Y, long long ldy,
         int relu = 0, const float* r1 = nullptr, int ksplit = 1, long long ypart = 0, int xparts = 1, long long xpart = 0, const float* xbias = nullptr) {
    const auto sp = e.split.find(W);
    if (e.d.gemm != MPV_GEMM_F32 && sp != e.split.end() && K % 4 == 0 && N % 4 == 0) {
        // two-term f16 split on the f16 matrix pipe, fp32-equivalent: the k = T+1 convolutions and the graph-conv Linear maps
        NstGemmHArgs a{X, ldx, sp->second.h, sp->second.l, sp->second.inv_scale, bias, r1, ldy, Y, ldy, rows, N, K, relu, sp->second.ldw};
        a.ksplit = ksplit; a.ypart = ypart; a.xparts = xparts; a.xpart = xpart; a.xbias = xbias;
        if (stg_down_fits(a)) stg_down_h3_launch(a, ctx->stream);            // long K, N <= 160: X streamed straight into the MFMA operands (stg_down_mfma.h)
        else nst_gemm_h3_launch(a, ctx->stream);
        MPV_HIP_TRY(ctx, hipGetLastError());
        return MPV_OK;
    }
    if (ksplit > 1 || xparts > 1) return mpv_fail(ctx, MPV_ESTATE, "DiffSTG: split-K products need the split-f16 GEMM");
    int KC = 64;
    if (K < 64) KC = (K + 15) & ~15;
    GemmArgs a{X, ldx, W, bias, Y, ldy, rows, K, N, KC, relu, r1};
    const size_t lds = sizeof(float) * (size_t)(16 * GEMM_NT + 64) * (KC + 4);
    MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(stg_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((N + 16 * GEMM_NT - 1) / (16 * GEMM_NT)));
    hipLaunchKernelGGL(stg_gemm_kernel, grid, dim3(256), lds, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

int carve(mpv_ctx* ctx, StgEngine& e, long long N) {
    if (N <= e.cap_N) return MPV_OK;
    const int T2 = e.T0, F = e.d.nf, Ts = e.mode ? T2 : T2 / 2, Fo = e.mode ? e.d.d_h : F;      // Ts: length of the sampled sequence
    size_t total = 0;
    auto take = [&](size_t n) { const size_t at = total; total += (n + 3) & ~size_t(3); return at; };
    std::vector<size_t> o_act(e.steps.size() + 1);
    size_t max_ct = (size_t)e.d.d_h * T2, max_c = 0;
    o_act[0] = take((size_t)N * e.d.d_h * T2);
    for (size_t i = 0; i < e.steps.size(); ++i) {
        const Step& s = e.steps[i];
        const int To = s.kind == 1 ? (s.T - 1) / 2 + 1 : (s.kind == 2 ? 2 * s.T : s.T);
        o_act[i + 1] = take((size_t)N * s.c_out * To);
        max_ct = std::max(max_ct, (size_t)s.c_out * To);
        max_c = std::max(max_c, (size_t)s.c_out);
    }
    const size_t C = (size_t)e.d.Td_h * max_c;
    const size_t o_hn = take((size_t)N * max_ct), o_z = take((size_t)N * C * STG_KSPLIT_MAX), o_kqvs = take((size_t)N * 4 * C), o_sp = take((size_t)N * C),
                 o_up = take((size_t)N * max_ct), o_e1 = take((size_t)N * Fo * T2), o_eps = take((size_t)N * F * Ts),
                 o_xt = take((size_t)N * Ts * F), o_xm = take((size_t)N * Ts * F), o_nz = take((size_t)N * Ts * F),
                 o_sig = take(e.mode ? (size_t)N * Ts * F : 0), o_e2 = take(e.mode ? (size_t)N * Fo * T2 : 0), o_gg = take(e.mode ? (size_t)N * Ts * F : 0);
    if (e.ws) {
        MPV_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(e.ws);
        e.ws = nullptr;
        e.cap_N = 0;
    }
    if (hipMalloc(&e.ws, total * sizeof(float)) != hipSuccess) return mpv_fail(ctx, MPV_ENOMEM, "DiffSTG workspace");
    e.ws_floats = total;
    e.cap_N = N;
    e.act.resize(o_act.size());
    for (size_t i = 0; i < o_act.size(); ++i) e.act[i] = e.ws + o_act[i];
    e.hn = e.ws + o_hn; e.z = e.ws + o_z; e.kqvs = e.ws + o_kqvs; e.sp = e.ws + o_sp; e.up = e.ws + o_up; e.e1 = e.ws + o_e1; e.eps = e.ws + o_eps;
    e.xt = e.ws + o_xt; e.xm = e.ws + o_xm; e.noise = e.ws + o_nz;
    e.sig = e.ws + o_sig; e.e2 = e.ws + o_e2; e.gg = e.ws + o_gg;
    return MPV_OK;
}

// The U-Net body shared by both modes (down path, middle, up path of UGnet.forward): e.act[0] [N][d_h][T0] -> *out_cur [N][d_h][T0]
int unet(mpv_ctx* ctx, StgEngine& e, long long N, int t, const float** out_cur) {
    const int d_h = e.d.d_h, Td = e.d.Td_h;
    hipStream_t st = ctx->stream;
    const float* temb = e.temb + (size_t)t * d_h;
    std::vector<std::pair<const float*, int>> skips;                    // (buffer, channels)
    skips.push_back({e.act[0], d_h});
    const float* cur = e.act[0];
    int cur_c = d_h;
    for (size_t i = 0; i < e.steps.size(); ++i) {
        const Step& s = e.steps[i];
        float* out = e.act[i + 1];
        if (s.kind == 1 || s.kind == 2) {
            const bool up = s.kind == 2;
            const int c = s.c_out, To = up ? 2 * s.T : (s.T - 1) / 2 + 1;
            const size_t lds = sizeof(float) * ((size_t)c * (up ? 4 : 3) * c + (size_t)c * (s.T + 2));
            const long long per_cu = std::max<long long>(1, std::min<long long>(8, (long long)ctx->max_lds / (long long)std::max<size_t>(lds, 1)));
            const unsigned grid = (unsigned)std::min<long long>(N, per_cu * ctx->cu_count);
#define STG_RESAMPLE(C_)                                                                                                                       \
    do {                                                                                                                                      \
        if (up) {                                                                                                                             \
            MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(stg_resample_kernel<C_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((stg_resample_kernel<C_, true>), dim3(grid), dim3(256), lds, st, cur, s.cw, s.cb, s.T, To, N, out);             \
        } else {                                                                                                                              \
            MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(stg_resample_kernel<C_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((stg_resample_kernel<C_, false>), dim3(grid), dim3(256), lds, st, cur, s.cw, s.cb, s.T, To, N, out);            \
        }                                                                                                                                     \
    } while (0)
            if (lds <= ctx->max_lds && c == 4) STG_RESAMPLE(4);
            else if (lds <= ctx->max_lds && c == 8) STG_RESAMPLE(8);
            else if (lds <= ctx->max_lds && c == 16) STG_RESAMPLE(16);
            else if (lds <= ctx->max_lds && c == 32) STG_RESAMPLE(32);
            else if (up) hipLaunchKernelGGL(stg_up_kernel, g1d(N * c * To), dim3(256), 0, st, cur, s.cw, s.cb, c, s.T, To, N * c * To, out);
            else hipLaunchKernelGGL(stg_down_kernel, g1d(N * c * To), dim3(256), 0, st, cur, s.cw, s.cb, c, s.T, To, N * c * To, out);
#undef STG_RESAMPLE
        } else {
            const float* src2 = nullptr;
            int c2 = 0;
            if (s.pop) {
                src2 = skips.back().first;
                c2 = skips.back().second;
                skips.pop_back();
            }
            if (cur_c + c2 != s.c_in) return mpv_fail(ctx, MPV_ESTATE, "DiffSTG: block input channels do not match the layout");
            const int co = s.c_out, C = Td * co;
            static const bool tcn_valu = getenv("MPV_STG_TCN_VALU") != nullptr;      // the round-2 vector kernel, kept for A/B measurements
            if (s.f.ok && !tcn_valu) {
                // front half of the block on the f16 matrix pipe (stg_tcn_mfma.h): time segments sized so that four workgroups share a CU
                const TcnFrag& f = s.f;
                auto lds_of = [&](int tseg) {
                    const size_t nt1 = (size_t)(tseg + 2 + 15) / 16;
                    return 2 * sizeof(_Float16) * ((16 * nt1 + 2) * (size_t)(f.cip1 + 8) + (16 * nt1 + 16) * (size_t)(f.cip2 + 8)) + 64;
                };
                static const int seg_env = getenv("MPV_STG_TCN_SEG") ? atoi(getenv("MPV_STG_TCN_SEG")) : 0;
                // as few, equally long segments (multiples of 16 steps) as keep `wg_target` workgroups on a CU: the kernel has no software
                // pipeline of its own -- other workgroups' MFMA phases cover a workgroup's loads (measured: register prefetch of the next
                // segment, +60 VGPRs and two workgroups per CU, and 16-byte loads with a register transpose were both slower)
                static const int wg_target = getenv("MPV_STG_TCN_WGS") ? std::max(1, atoi(getenv("MPV_STG_TCN_WGS"))) : 4;
                int nseg = seg_env > 0 ? seg_env : 1;
                int tseg = ((s.T + nseg - 1) / nseg + 15) & ~15;
                while (seg_env <= 0 && tseg > 16 && lds_of(tseg) > ctx->max_lds / wg_target) {
                    ++nseg;
                    tseg = ((s.T + nseg - 1) / nseg + 15) & ~15;
                }
                nseg = (s.T + tseg - 1) / tseg;
                const size_t lds = lds_of(tseg);
                if (lds > ctx->max_lds) return mpv_fail(ctx, MPV_EINVAL, "DiffSTG: a time segment does not fit the LDS");
                TcnMArgs a{cur, cur_c, src2, c2, f.a1, f.a2, f.asb, f.inv1, f.inv2, f.invsb, f.b1s, s.r.wt, s.r.bt, temb, s.r.b2, s.r.lnw, s.r.lnb, s.r.bs,
                           e.hn, e.up, co, s.T, d_h, f.cip1, f.cip2, N, tseg, nseg};
                const long long per_cu = std::max<long long>(1, std::min<long long>(8, (long long)ctx->max_lds / (long long)lds));
                const unsigned grid = (unsigned)std::min<long long>(N * nseg, per_cu * ctx->cu_count);
#define STG_TCNM(N1_, N2_)                                                                                                                      \
    do {                                                                                                                                      \
        if (f.asb) {                                                                                                                          \
            MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(stg_tcn_mfma_kernel<N1_, N2_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((stg_tcn_mfma_kernel<N1_, N2_, true>), dim3(grid), dim3(256), lds, st, a);                                     \
        } else {                                                                                                                              \
            MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(stg_tcn_mfma_kernel<N1_, N2_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((stg_tcn_mfma_kernel<N1_, N2_, false>), dim3(grid), dim3(256), lds, st, a);                                    \
        }                                                                                                                                     \
    } while (0)
                switch (f.nc1 * 10 + f.nc2) {
                    case 11: STG_TCNM(1, 1); break;
                    case 12: STG_TCNM(1, 2); break;
                    case 21: STG_TCNM(2, 1); break;
                    case 22: STG_TCNM(2, 2); break;
                    case 31: STG_TCNM(3, 1); break;
                    case 32: STG_TCNM(3, 2); break;
                    default: return mpv_fail(ctx, MPV_ESTATE, "DiffSTG: unexpected TCN chunk counts");
                }
#undef STG_TCNM
                MPV_HIP_TRY(ctx, hipGetLastError());
            } else {
                TcnArgs a{cur, cur_c, src2, c2, s.r.w1, s.r.b1, s.r.ws1, s.r.bs1, s.r.wt, s.r.bt, temb, s.r.w2, s.r.b2, s.r.lnw, s.r.lnb, e.hn, co, s.T, d_h,
                          s.r.ws, s.r.bs, e.up, N, 0, 1};
                // segments of the time axis (256 steps keep every thread busy in one pass) until at least two workgroups fit a CU
                static const int seg_env = getenv("MPV_STG_TCN_SEG") ? atoi(getenv("MPV_STG_TCN_SEG")) : 0;
                auto lds_of = [&](int tseg) { return sizeof(float) * ((size_t)s.c_in * (tseg + 4) + (size_t)co * (tseg + 2) + (size_t)s.c_in * co * 5 + (size_t)co * co * 4 + co); };
                a.nseg = seg_env > 0 ? seg_env : 1;
                a.tseg = (s.T + a.nseg - 1) / a.nseg;
                while (seg_env == 0 && a.tseg > 64 && (lds_of(a.tseg) > ctx->max_lds / 2 || a.tseg > 256)) {
                    ++a.nseg;
                    a.tseg = (s.T + a.nseg - 1) / a.nseg;
                }
                const size_t lds = lds_of(a.tseg);
                // a workgroup stages the weights once and walks over its items: as many workgroups as fit the chip at this LDS footprint
                const long long per_cu = std::max<long long>(1, std::min<long long>(4, (long long)ctx->max_lds / (long long)lds));
                const unsigned tcn_grid = (unsigned)std::min<long long>(N * a.nseg, per_cu * ctx->cu_count);
                if (lds > ctx->max_lds) return mpv_fail(ctx, MPV_EINVAL, "DiffSTG: a node row does not fit the LDS (sequence too long)");
#define STG_TCN(CO_)                                                                                                                          \
    do {                                                                                                                                      \
        MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(stg_tcn_ln_kernel<CO_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(stg_tcn_ln_kernel<CO_>, dim3(tcn_grid), dim3(256), lds, st, a);                                                 \
    } while (0)
                switch (co) {
                    case 2: STG_TCN(2); break;
                    case 4: STG_TCN(4); break;
                    case 8: STG_TCN(8); break;
                    case 16: STG_TCN(16); break;
                    case 32: STG_TCN(32); break;
                    default: return mpv_fail(ctx, MPV_EINVAL, "DiffSTG: block channel counts must be 2, 4, 8, 16 or 32 (d_h times the channel multipliers)");
                }
#undef STG_TCN
                MPV_HIP_TRY(ctx, hipGetLastError());
            }
            // long-K, narrow-N: split the k loop over up to four workgroups per tile; the graph-conv product sums the partials (+ the bias) as it loads them
            const int ks = down_ksplit(ctx, e, s.r.wdown, co * s.T, C, N);
            if (ks > 1) {
                STG_TRY(gemm(ctx, e, e.hn, (long long)co * s.T, s.r.wdown, nullptr, co * s.T, C, N, e.z, C, 0, nullptr, ks, (long long)N * C));
                STG_TRY(gemm(ctx, e, e.z, C, s.r.wg, s.r.bg4, C, s.ng, N, e.kqvs, s.ng, 0, nullptr, 1, 0, ks, (long long)N * C, s.r.bdown));
            } else {
                STG_TRY(gemm(ctx, e, e.hn, (long long)co * s.T, s.r.wdown, s.r.bdown, co * s.T, C, N, e.z, C));
                STG_TRY(gemm(ctx, e, e.z, C, s.r.wg, s.r.bg4, C, s.ng, N, e.kqvs, s.ng));
            }
            const int* rp = e.graph;
            const int* ci_ = e.graph + e.V + 1;
            const int* dg_ = ci_ + std::max(e.E, 1);
            if (s.gnn == 0) hipLaunchKernelGGL(stg_gate_kernel, g1d(N * C), dim3(256), 0, st, e.kqvs, s.r.gbias, rp, ci_, e.V, C, N * C, e.sp);
            else graph_aggr_launch(s.gnn, st, e.kqvs, s.ng, s.r.gbias, rp, ci_, dg_, e.V, C, s.heads, s.slope, N, e.sp);
            MPV_HIP_TRY(ctx, hipGetLastError());
            // up-sampling product with the block's shortcut added in its epilogue: the block input itself (identity) or the 1x1
            // convolution the TCN kernel left in e.up
            if (!s.r.ws && cur_c != co) return mpv_fail(ctx, MPV_ESTATE, "DiffSTG: identity shortcut with differing channel counts");
            STG_TRY(gemm(ctx, e, e.sp, C, s.r.wup, s.r.bup, C, co * s.T, N, out, (long long)co * s.T, 0, s.r.ws ? e.up : cur));
        }
        MPV_HIP_TRY(ctx, hipGetLastError());
        cur = out;
        cur_c = s.c_out;
        if (s.push) skips.push_back({out, s.c_out});
    }
    *out_cur = cur;
    return MPV_OK;
}

// UGnet.forward (DiffSTG/ugnet.py:252-295): xt, xm [N][T][F], timestep t -> e.eps [N][F][T]
int forward(mpv_ctx* ctx, StgEngine& e, const float* xt, const float* xm, long long N, int t) {
    const int T = e.d.T_h + e.d.T_p, T2 = 2 * T, F = e.d.nf, d_h = e.d.d_h;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(stg_xproj_kernel, g1d(N * d_h * T2), dim3(256), 0, st, xt, xm, e.wx, e.bx, T, F, d_h, N * d_h * T2, e.act[0]);
    MPV_HIP_TRY(ctx, hipGetLastError());
    const float* cur = nullptr;
    STG_TRY(unet(ctx, e, N, t, &cur));
    hipLaunchKernelGGL(stg_outconv_kernel, g1d(N * F * T2), dim3(256), 0, st, cur, e.wo, e.bo, T2, F, d_h, N * F * T2, e.e1);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return gemm(ctx, e, e.e1, T2, e.wl, e.bl, T2, T, N * F, e.eps, T);
}

// NsDiff's UGnet.forward (NsDiff/ugnet.py:257-292): y, mu (= y_0_hat), g [N][T][F], timestep t -> e.eps, e.sig [N][T][F]
int forward_nss(mpv_ctx* ctx, StgEngine& e, const float* y, const float* mu, const float* g, long long N, int t) {
    const int T = e.T0, F = e.d.nf, d_h = e.d.d_h;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(nss_xproj_kernel, g1d(N * d_h * T), dim3(256), 0, st, y, mu, g, e.wx, e.bx, T, F, d_h, N * d_h * T, e.act[0]);
    MPV_HIP_TRY(ctx, hipGetLastError());
    const float* cur = nullptr;
    STG_TRY(unet(ctx, e, N, t, &cur));
    hipLaunchKernelGGL(stg_outconv_kernel, g1d(N * d_h * T), dim3(256), 0, st, cur, e.wo, e.bo, T, d_h, d_h, N * d_h * T, e.e1);    // out.0: Conv2d(d_h, d_h, 1x1)
    MPV_HIP_TRY(ctx, hipGetLastError());
    STG_TRY(gemm(ctx, e, e.e1, T, e.wl, e.bl, T, T, N * d_h, e.e2, T));                                                             // out.1: Linear(T, T) over time
    hipLaunchKernelGGL(nss_heads_kernel, g1d(N * T), dim3(256), 0, st, e.e2, e.w4, e.b4, e.wsg, e.bsg, T, F, d_h, N * T, e.eps, e.sig);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// Shared loader of both modes.  head[] (host): mode 0 { x_proj.weight [d_h][F], x_proj.bias, out.0.weight [F][d_h], out.0.bias, out.1.weight
// [T][2T], out.1.bias }; mode 1 { x_proj.weight [d_h][3F], x_proj.bias, out.0.weight [d_h][d_h], out.0.bias, out.1.weight [T][T], out.1.bias,
// lin4.weight [F][d_h], lin4.bias, sigma_lin.weight [F][d_h], sigma_lin.bias, coef [steps][MPV_COEF_COLS] }.
int load_engine(mpv_ctx* ctx, const std::string& who, int mode, const mpv_stg_desc* d, int T0, int steps_T, const mpv_stg_step* steps, int32_t n_steps,
                const float* const* head, const float* temb, StgEngine** slot) {
    const int T = d->T_h + d->T_p, F = d->nf, d_h = d->d_h, Td = d->Td_h, n_head = mode ? 11 : 6;
    for (int i = 0; i < n_head; ++i)
        if (!head[i]) return mpv_fail(ctx, MPV_EINVAL, who + ": NULL head tensor");
    DeviceGuard dg(ctx->device);
    MPV_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    stg_engine_destroy(*slot);
    *slot = nullptr;
    StgEngine* e = new (std::nothrow) StgEngine();
    if (!e) return mpv_fail(ctx, MPV_ENOMEM, who + ": out of host memory");
    e->d = *d;
    e->mode = mode;
    e->T0 = T0;
    e->steps_T = steps_T;
    std::vector<float> host;
    auto put = [&](const float* p, size_t n) { const size_t at = host.size(); host.insert(host.end(), p, p + n); host.resize((host.size() + 3) & ~size_t(3)); return at; };
    auto putv = [&](const std::vector<float>& v) { return put(v.data(), v.size()); };
    std::vector<std::vector<size_t>> offs(n_steps);
    std::vector<_Float16> frag_host;                                  // matrix-pipe TCN fragments of every residual block
    struct FragOff { size_t a1 = 0, a2 = 0, asb = 0, b1s = 0; bool sb = false; };
    std::vector<FragOff> frag_off(n_steps);
    const int pad = Td / 2;
    for (int i = 0; i < n_steps; ++i) {
        const mpv_stg_step& s = steps[i];
        Step st;
        st.kind = s.kind; st.c_in = s.c_in; st.c_out = s.c_out; st.T = s.T; st.pop = s.pop; st.push = s.push;
        if (s.kind >= MPV_STG_RES_SAGE_MEAN && s.kind <= MPV_STG_RES_GAT) { st.gnn = s.kind - MPV_STG_RES_SAGE_MEAN + 1; st.kind = 0; }
        else if (s.kind < 0 || s.kind > 2) { delete e; return mpv_fail(ctx, MPV_EINVAL, who + ": unknown step kind"); }
        if (s.c_in < 1 || s.c_out < 1 || s.T < 2 || ((s.c_out * s.T) & 3) || ((Td * s.c_out) & 3)) { delete e; return mpv_fail(ctx, MPV_EINVAL, who + ": bad step shape"); }
        std::vector<size_t>& o = offs[i];
        if (st.kind == 0) {
            const int ci = s.c_in, co = s.c_out, Ts = s.T, C = Td * co;
            for (int k : {0, 1, 4, 5, 6, 7, 8, 9, 10, 11, 12, 20, 21})
                if (!s.w[k]) { delete e; return mpv_fail(ctx, MPV_EINVAL, who + ": NULL tensor in a residual block"); }
            if (st.gnn == 0)
                for (int k : {13, 14, 15, 16, 17, 18, 19})
                    if (!s.w[k]) { delete e; return mpv_fail(ctx, MPV_EINVAL, who + ": NULL tensor in a residual block"); }
            if (st.gnn == 4) {
                if (!s.w[14] || !s.w[15] || !s.w[16]) { delete e; return mpv_fail(ctx, MPV_EINVAL, who + ": GATConv needs att_src, att_dst and {heads, negative_slope}"); }
                st.heads = (int)s.w[16][0];
                st.slope = s.w[16][1];
                if (st.heads < 1 || C % st.heads != 0 || (float)st.heads != s.w[16][0]) { delete e; return mpv_fail(ctx, MPV_EINVAL, who + ": GATConv heads must divide the channels"); }
            }
            auto mid = [&](const float* w, int o_, int i_) {            // [o][i][3][3] -> middle row [o][i][3]
                std::vector<float> m((size_t)o_ * i_ * 3);
                for (int a = 0; a < o_ * i_; ++a)
                    for (int k = 0; k < 3; ++k) m[(size_t)a * 3 + k] = w[(size_t)a * 9 + 3 + k];
                return m;
            };
            o.push_back(putv(mid(s.w[0], co, ci))); o.push_back(put(s.w[1], co));
            if (ci <= 32 && co <= 16) {                              // the block's front half on the f16 matrix pipe (stg_tcn_mfma.h)
                TcnFrag& f = st.f;
                f.cip1 = (ci + 7) & ~7; f.cip2 = (co + 7) & ~7;
                f.nc1 = (3 * f.cip1 + 31) / 32; f.nc2 = (3 * f.cip2 + 31) / 32;
                std::vector<float> w1 = mid(s.w[0], co, ci), w2 = mid(s.w[6], co, co), b1s(s.w[1], s.w[1] + co);
                for (int oc = 0; oc < co; ++oc) {
                    for (int ic = 0; ic < ci; ++ic)                  // tcn1's 1x1 skip (or the identity) rides in the current-time tap
                        w1[((size_t)oc * ci + ic) * 3 + 2] += s.w[2] ? s.w[2][(size_t)oc * ci + ic] : (oc == ic ? 1.0f : 0.0f);
                    w2[((size_t)oc * co + oc) * 3 + 2] += 1.0f;      // tcn2's skip is the identity (c_in == c_out)
                    if (s.w[3]) b1s[oc] += s.w[3][oc];
                }
                FragOff& fo = frag_off[i];
                fo.a1 = frag_host.size();
                f.inv1 = tcn_fragments(w1, co, ci, f.cip1, 3, f.nc1, frag_host);
                fo.a2 = frag_host.size();
                f.inv2 = tcn_fragments(w2, co, co, f.cip2, 3, f.nc2, frag_host);
                if (s.w[22]) {
                    fo.sb = true;
                    fo.asb = frag_host.size();
                    f.invsb = tcn_fragments(std::vector<float>(s.w[22], s.w[22] + (size_t)co * ci), co, ci, f.cip1, 1, 1, frag_host);
                }
                fo.b1s = putv(b1s);
                f.ok = (s.w[2] != nullptr) || ci == co;             // (an identity skip needs c_in == c_out, as in the reference)
            }
            o.push_back(s.w[2] ? put(s.w[2], (size_t)co * ci) : (size_t)-1); o.push_back(s.w[3] ? put(s.w[3], co) : (size_t)-1);
            o.push_back(put(s.w[4], (size_t)co * d_h)); o.push_back(put(s.w[5], co));
            o.push_back(putv(mid(s.w[6], co, co))); o.push_back(put(s.w[7], co));
            o.push_back(put(s.w[8], co)); o.push_back(put(s.w[9], co));
            // downsampling Conv2d(co, co, (1, T+1), pad Td/2): row j*co + o, column i*T + t  <-  w[o][i][t - j + pad]
            std::vector<float> wd((size_t)C * co * Ts, 0.0f), bd(C);
            for (int j = 0; j < Td; ++j)
                for (int oc = 0; oc < co; ++oc) {
                    bd[(size_t)j * co + oc] = s.w[11][oc];
                    for (int ic = 0; ic < co; ++ic)
                        for (int t = 0; t < Ts; ++t) {
                            const int k = t - j + pad;
                            if (k >= 0 && k <= Ts) wd[((size_t)j * co + oc) * co * Ts + (size_t)ic * Ts + t] = s.w[10][((size_t)oc * co + ic) * (Ts + 1) + k];
                        }
                }
            o.push_back(putv(wd)); o.push_back(putv(bd));
            // the graph convolution's Linear maps as ONE weight matrix [ng][C] (+ a bias per product column, + the bias added behind the aggregation)
            st.ng = graph_conv_ng(st.gnn, C, st.heads);
            std::vector<float> wg, bg, gb;
            graph_conv_pack(st.gnn, C, st.heads, &s.w[12], wg, bg, gb);
            o.push_back(putv(wg)); o.push_back(putv(bg)); o.push_back(putv(gb));
            // upsampling ConvTranspose2d(co, co, (1, T+1), pad Td/2): row o*T + t, column j*co + i  <-  w[i][o][t - j + pad]
            std::vector<float> wu((size_t)co * Ts * C, 0.0f), bu((size_t)co * Ts);
            for (int oc = 0; oc < co; ++oc)
                for (int t = 0; t < Ts; ++t) {
                    bu[(size_t)oc * Ts + t] = s.w[21][oc];
                    for (int j = 0; j < Td; ++j) {
                        const int k = t - j + pad;
                        if (k < 0 || k > Ts) continue;
                        for (int ic = 0; ic < co; ++ic) wu[((size_t)oc * Ts + t) * C + (size_t)j * co + ic] = s.w[20][((size_t)ic * co + oc) * (Ts + 1) + k];
                    }
                }
            o.push_back(putv(wu)); o.push_back(putv(bu));
            o.push_back(s.w[22] ? put(s.w[22], (size_t)co * ci) : (size_t)-1); o.push_back(s.w[23] ? put(s.w[23], co) : (size_t)-1);
        } else {
            if (!s.w[0] || !s.w[1] || s.c_in != s.c_out) { delete e; return mpv_fail(ctx, MPV_EINVAL, who + ": bad resampling step"); }
            o.push_back(put(s.w[0], (size_t)s.c_out * s.c_out * (s.kind == 1 ? 3 : 4)));
            o.push_back(put(s.w[1], s.c_out));
        }
        e->steps.push_back(st);
    }
    const int Fi = mode ? 3 * F : F, Fo = mode ? d_h : F, Tk = mode ? T : 2 * T;
    const size_t o_wx = put(head[0], (size_t)d_h * Fi), o_bx = put(head[1], d_h), o_wo = put(head[2], (size_t)Fo * d_h), o_bo = put(head[3], Fo),
                 o_wl = put(head[4], (size_t)T * Tk), o_bl = put(head[5], T), o_temb = put(temb, (size_t)(d->timesteps + 1) * d_h);
    size_t o_w4 = 0, o_b4 = 0, o_wsg = 0, o_bsg = 0, o_coef = 0;
    if (mode) {
        o_w4 = put(head[6], (size_t)F * d_h); o_b4 = put(head[7], F); o_wsg = put(head[8], (size_t)F * d_h); o_bsg = put(head[9], F);
        o_coef = put(head[10], (size_t)steps_T * MPV_COEF_COLS);
    }
    if (hipMalloc(&e->blob, host.size() * sizeof(float)) != hipSuccess) { delete e; return mpv_fail(ctx, MPV_ENOMEM, who + ": weight blob"); }
    if (hipMemcpy(e->blob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        stg_engine_destroy(e);
        return mpv_fail(ctx, MPV_EHIP, who + ": upload failed");
    }
    auto P = [&](size_t off) { return off == (size_t)-1 ? (const float*)nullptr : (const float*)(e->blob + off); };
    for (int i = 0; i < n_steps; ++i) {
        Step& st = e->steps[i];
        const std::vector<size_t>& o = offs[i];
        if (st.kind == 0) {
            st.r = ResW{P(o[0]), P(o[1]), P(o[2]), P(o[3]), P(o[4]), P(o[5]), P(o[6]), P(o[7]), P(o[8]), P(o[9]), P(o[10]), P(o[11]), P(o[12]), P(o[13]),
                        P(o[14]), P(o[15]), P(o[16]), P(o[17]), P(o[18])};
        } else {
            st.cw = P(o[0]);
            st.cb = P(o[1]);
        }
    }
    if (!frag_host.empty()) {
        if (hipMalloc(&e->frag16, frag_host.size() * sizeof(_Float16)) != hipSuccess ||
            hipMemcpy(e->frag16, frag_host.data(), frag_host.size() * sizeof(_Float16), hipMemcpyHostToDevice) != hipSuccess) {
            stg_engine_destroy(e);
            return mpv_fail(ctx, MPV_ENOMEM, who + ": TCN fragment blob");
        }
        for (int i = 0; i < n_steps; ++i) {
            TcnFrag& f = e->steps[i].f;
            if (!f.ok) continue;
            f.a1 = reinterpret_cast<const th8*>(e->frag16 + frag_off[i].a1);
            f.a2 = reinterpret_cast<const th8*>(e->frag16 + frag_off[i].a2);
            f.asb = frag_off[i].sb ? reinterpret_cast<const th8*>(e->frag16 + frag_off[i].asb) : nullptr;
            f.b1s = P(frag_off[i].b1s);
        }
    }
    e->wx = P(o_wx); e->bx = P(o_bx); e->wo = P(o_wo); e->bo = P(o_bo); e->wl = P(o_wl); e->bl = P(o_bl); e->temb = P(o_temb);
    if (mode) { e->w4 = P(o_w4); e->b4 = P(o_b4); e->wsg = P(o_wsg); e->bsg = P(o_bsg); e->coef = P(o_coef); }
    // two-term f16 split (power-of-two scaled so the low term stays a normal f16) of the three big GEMM weights of every block
    {
        std::vector<_Float16> host16;
        struct Pending { const float* key; size_t at, n; float inv; int ldw; };
        std::vector<Pending> pend;
        auto add = [&](const float* dev_w, size_t off, size_t rows_, size_t K_) {      // rows padded with zeros to a multiple of the GEMM's k-tile
            const float* w = host.data() + off;
            const size_t Kp = (K_ + G_BK - 1) / G_BK * G_BK, n = rows_ * Kp;
            float mx = 0.0f;
            for (size_t i = 0; i < rows_ * K_; ++i) mx = std::max(mx, std::fabs(w[i]));
            int ex = 0;
            if (mx > 0.0f && std::isfinite(mx)) { (void)std::frexp(mx, &ex); ex = 14 - ex; }
            const float sc = std::ldexp(1.0f, ex);
            const size_t at = host16.size();
            host16.resize(at + 2 * n, (_Float16)0.0f);
            for (size_t r = 0; r < rows_; ++r)
                for (size_t c = 0; c < K_; ++c) {
                    const float v = w[r * K_ + c] * sc;
                    uint32_t bits;
                    std::memcpy(&bits, &v, 4);
                    bits &= 0xFFFFE000u;
                    float hi;
                    std::memcpy(&hi, &bits, 4);
                    host16[at + r * Kp + c] = (_Float16)hi;
                    host16[at + n + r * Kp + c] = (_Float16)(v - hi);
                }
            pend.push_back(Pending{dev_w, at, n, std::ldexp(1.0f, -ex), (int)Kp});
        };
        for (int i = 0; i < n_steps; ++i) {
            const Step& st = e->steps[i];
            if (st.kind != 0) continue;
            const size_t C = (size_t)Td * st.c_out, CT = (size_t)st.c_out * st.T;
            add(st.r.wdown, offs[i][10], C, CT);
            add(st.r.wg, offs[i][12], (size_t)st.ng, C);
            add(st.r.wup, offs[i][15], CT, C);
        }
        if (!host16.empty()) {
            if (hipMalloc(&e->blob16, host16.size() * sizeof(_Float16)) != hipSuccess ||
                hipMemcpy(e->blob16, host16.data(), host16.size() * sizeof(_Float16), hipMemcpyHostToDevice) != hipSuccess) {
                stg_engine_destroy(e);
                return mpv_fail(ctx, MPV_ENOMEM, who + ": f16 weight blob");
            }
            for (const Pending& pd : pend) e->split[pd.key] = SplitW{e->blob16 + pd.at, e->blob16 + pd.at + pd.n, pd.inv, pd.ldw};
        }
    }
    *slot = e;
    return MPV_OK;
}

int set_graph(mpv_ctx* ctx, StgEngine& e, const char* who, int32_t n_nodes, int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst) {
    if (n_nodes < 1 || n_edges < 0 || (n_edges > 0 && (!edge_src || !edge_dst))) return mpv_fail(ctx, MPV_EINVAL, std::string(who) + ": bad argument");
    std::vector<int> row_ptr(n_nodes + 1, 0), col(n_edges);
    for (int64_t k = 0; k < n_edges; ++k) {
        if (edge_src[k] < 0 || edge_src[k] >= n_nodes || edge_dst[k] < 0 || edge_dst[k] >= n_nodes) return mpv_fail(ctx, MPV_EINVAL, std::string(who) + ": node index out of range");
        ++row_ptr[edge_dst[k] + 1];
    }
    for (int v = 0; v < n_nodes; ++v) row_ptr[v + 1] += row_ptr[v];
    std::vector<int> fill(row_ptr.begin(), row_ptr.end() - 1);
    for (int64_t k = 0; k < n_edges; ++k) col[fill[edge_dst[k]]++] = (int)edge_src[k];        // edge order kept within a target: the sum order of index_add_
    DeviceGuard dg(ctx->device);
    MPV_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (e.graph) (void)hipFree(e.graph);
    e.graph = nullptr;
    std::vector<int> deg(n_nodes, 0);                                  // in-degree without self loops (GCNConv's normalisation: + 1 for the loop it adds)
    for (int64_t k = 0; k < n_edges; ++k)
        if (edge_src[k] != edge_dst[k]) ++deg[edge_dst[k]];
    const size_t n_col = (size_t)std::max<int64_t>(n_edges, 1);
    if (hipMalloc(&e.graph, sizeof(int) * ((size_t)n_nodes + 1 + n_col + (size_t)n_nodes)) != hipSuccess) return mpv_fail(ctx, MPV_ENOMEM, who);
    MPV_HIP_TRY(ctx, hipMemcpy(e.graph, row_ptr.data(), sizeof(int) * (n_nodes + 1), hipMemcpyHostToDevice));
    if (n_edges) MPV_HIP_TRY(ctx, hipMemcpy(e.graph + n_nodes + 1, col.data(), sizeof(int) * n_edges, hipMemcpyHostToDevice));
    MPV_HIP_TRY(ctx, hipMemcpy(e.graph + n_nodes + 1 + n_col, deg.data(), sizeof(int) * n_nodes, hipMemcpyHostToDevice));
    e.V = n_nodes;
    e.E = (int)n_edges;
    return MPV_OK;
}

}  // namespace

extern "C" {

int mpv_stg_load(mpv_ctx* ctx, const mpv_stg_desc* d, const mpv_stg_step* steps, int32_t n_steps, const float* const* head /*6*/, const float* temb) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_stg_load: ctx is NULL");
    if (!d || !steps || !head || !temb || n_steps < 1) return mpv_fail(ctx, MPV_EINVAL, "mpv_stg_load: NULL argument");
    const int T = d->T_h + d->T_p;
    if (T < 2 || (T & 1) || d->nf < 1 || d->d_h < 2 || d->Td_h < 2 || (d->Td_h & 1) || d->timesteps < 1)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_stg_load: need an even T_h + T_p, F >= 1, d_h >= 2, an even Td_h >= 2");
    return load_engine(ctx, "mpv_stg_load", 0, d, 2 * T, 0, steps, n_steps, head, temb, &ctx->stg);
}

int mpv_nss_load(mpv_ctx* ctx, const mpv_nss_desc* d, const mpv_stg_step* steps, int32_t n_steps, const float* const* head /*11*/, const float* temb) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_nss_load: ctx is NULL");
    if (!d || !steps || !head || !temb || n_steps < 1) return mpv_fail(ctx, MPV_EINVAL, "mpv_nss_load: NULL argument");
    if (d->pred_len < 4 || (d->pred_len & 3) || d->nf < 1 || d->d_h < 2 || d->Td_h < 2 || (d->Td_h & 1) || d->steps < 1)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nss_load: need pred_len a multiple of 4, F >= 1, d_h >= 2, an even Td_h >= 2, steps >= 1");
    mpv_stg_desc sd{};
    sd.T_h = 0; sd.T_p = d->pred_len; sd.nf = d->nf; sd.d_h = d->d_h; sd.Td_h = d->Td_h; sd.timesteps = d->steps - 1; sd.gemm = d->gemm;
    return load_engine(ctx, "mpv_nss_load", 1, &sd, d->pred_len, d->steps, steps, n_steps, head, temb, &ctx->nss);
}

int mpv_stg_set_graph(mpv_ctx* ctx, int32_t n_nodes, int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_stg_set_graph: ctx is NULL");
    if (!ctx->stg) return mpv_fail(ctx, MPV_ESTATE, "mpv_stg_set_graph: call mpv_stg_load first");
    return set_graph(ctx, *ctx->stg, "mpv_stg_set_graph", n_nodes, n_edges, edge_src, edge_dst);
}

int mpv_nss_set_graph(mpv_ctx* ctx, int32_t n_nodes, int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_nss_set_graph: ctx is NULL");
    if (!ctx->nss) return mpv_fail(ctx, MPV_ESTATE, "mpv_nss_set_graph: call mpv_nss_load first");
    return set_graph(ctx, *ctx->nss, "mpv_nss_set_graph", n_nodes, n_edges, edge_src, edge_dst);
}

int mpv_stg_forward(mpv_ctx* ctx, const float* xt, const float* x_masked, int64_t n_rows, int32_t t, float* eps_out) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_stg_forward: ctx is NULL");
    if (!ctx->stg || !ctx->stg->graph) return mpv_fail(ctx, MPV_ESTATE, "mpv_stg_forward: call mpv_stg_load and mpv_stg_set_graph first");
    StgEngine& e = *ctx->stg;
    if (!xt || !x_masked || !eps_out || n_rows < 1 || n_rows % e.V != 0 || t < 0 || t > e.d.timesteps)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_stg_forward: need rows = a multiple of the graph's node count and 0 <= t <= diffusion_steps");
    DeviceGuard dg(ctx->device);
    STG_TRY(carve(ctx, e, n_rows));
    STG_TRY(forward(ctx, e, xt, x_masked, n_rows, t));
    // eps [N][F][T] -> [N][T][F]
    const int T = e.d.T_h + e.d.T_p, F = e.d.nf;
    if (F == 1) {
        MPV_HIP_TRY(ctx, hipMemcpyAsync(eps_out, e.eps, sizeof(float) * n_rows * T, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        const long long n = (long long)n_rows * T * F;
        hipLaunchKernelGGL(stg_eps_transpose_kernel, g1d(n), dim3(256), 0, ctx->stream, e.eps, T, F, n, eps_out);
        MPV_HIP_TRY(ctx, hipGetLastError());
    }
    return MPV_OK;
}

int mpv_stg_sample(mpv_ctx* ctx, const mpv_stg_sample_shape* s, const float* history, const double* plan, const float* noise, float* samples) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_stg_sample: ctx is NULL");
    if (!ctx->stg || !ctx->stg->graph) return mpv_fail(ctx, MPV_ESTATE, "mpv_stg_sample: call mpv_stg_load and mpv_stg_set_graph first");
    if (!s || !history || !plan || !samples) return mpv_fail(ctx, MPV_EINVAL, "mpv_stg_sample: NULL argument");
    StgEngine& e = *ctx->stg;
    if (s->n_windows < 1 || s->parallel < 1 || s->sequential < 1 || s->n_steps < 1) return mpv_fail(ctx, MPV_EINVAL, "mpv_stg_sample: need n_windows, parallel, sequential, n_steps >= 1");
    DeviceGuard dg(ctx->device);
    const int T = e.d.T_h + e.d.T_p, F = e.d.nf, P = s->parallel, V = e.V;
    const long long N = (long long)s->n_windows * P * V;
    STG_TRY(carve(ctx, e, N));
    hipStream_t st = ctx->stream;
    const long long n = N * T * F;
    const unsigned seed_lo = (unsigned)s->seed, seed_hi = (unsigned)(s->seed >> 32);
    int draws = 1;
    for (int i = 0; i < s->n_steps; ++i) draws += plan[i * MPV_STG_PLAN_COLS + 1] > 0.0 ? 1 : 0;
    hipLaunchKernelGGL(stg_build_masked_kernel, g1d(n), dim3(256), 0, st, history, V, P, T, e.d.T_h, F, n, e.xm);
    MPV_HIP_TRY(ctx, hipGetLastError());
    for (int pass = 0; pass < s->sequential; ++pass) {
        long long draw = (long long)pass * draws;
        unsigned stream_id = (unsigned)pass * (unsigned)draws;            // passes never share a Philox stream, whatever the number of DDPM steps
        auto next_noise = [&](float* dst) -> int {
            if (noise) MPV_HIP_TRY(ctx, hipMemcpyAsync(dst, noise + draw * n, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
            else {
                hipLaunchKernelGGL(stg_philox_kernel, g1d((n + 3) / 4), dim3(256), 0, st, seed_lo, seed_hi, stream_id, n, dst);
                MPV_HIP_TRY(ctx, hipGetLastError());
            }
            ++draw;
            ++stream_id;
            return MPV_OK;
        };
        STG_TRY(next_noise(e.xt));                                       // x_T = randn_like(x_masked), :253
        for (int i = 0; i < s->n_steps; ++i) {
            const double* pr = plan + i * MPV_STG_PLAN_COLS;
            const int t1 = (int)pr[0];
            if (t1 < 0 || t1 > e.d.timesteps) return mpv_fail(ctx, MPV_EINVAL, "mpv_stg_sample: plan row with t outside [0, diffusion_steps]");
            STG_TRY(forward(ctx, e, e.xt, e.xm, N, t1));
            const int ddpm = pr[1] > 0.0;
            if (ddpm) STG_TRY(next_noise(e.noise));
            hipLaunchKernelGGL(stg_posterior_kernel, g1d(n), dim3(256), 0, st, e.eps, e.noise, (float)pr[2], (float)pr[3], (float)pr[4], ddpm, T, F, n, e.xt);
            MPV_HIP_TRY(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(stg_gather_kernel, g1d(n), dim3(256), 0, st, e.xt, V, P, T, F, P * s->sequential, pass, n, samples);
        MPV_HIP_TRY(ctx, hipGetLastError());
    }
    return MPV_OK;
}

int mpv_nss_forward(mpv_ctx* ctx, const float* y, const float* y0hat, const float* gx, int64_t n_rows, int32_t t, float* eps_out, float* sigma_out) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_nss_forward: ctx is NULL");
    if (!ctx->nss || !ctx->nss->graph) return mpv_fail(ctx, MPV_ESTATE, "mpv_nss_forward: call mpv_nss_load and mpv_nss_set_graph first");
    StgEngine& e = *ctx->nss;
    if (!y || !y0hat || !gx || !eps_out || !sigma_out || n_rows < 1 || n_rows % e.V != 0 || t < 0 || t >= e.steps_T)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nss_forward: need rows = a multiple of the graph's node count and 0 <= t < diffusion_steps");
    DeviceGuard dg(ctx->device);
    STG_TRY(carve(ctx, e, n_rows));
    STG_TRY(forward_nss(ctx, e, y, y0hat, gx, n_rows, t));
    const size_t bytes = sizeof(float) * (size_t)n_rows * e.T0 * e.d.nf;
    MPV_HIP_TRY(ctx, hipMemcpyAsync(eps_out, e.eps, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    MPV_HIP_TRY(ctx, hipMemcpyAsync(sigma_out, e.sig, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MPV_OK;
}

int mpv_nss_sample(mpv_ctx* ctx, const mpv_nss_sample_shape* s, const float* y0hat, const float* gx, const float* noise, float* samples) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_nss_sample: ctx is NULL");
    if (!ctx->nss || !ctx->nss->graph) return mpv_fail(ctx, MPV_ESTATE, "mpv_nss_sample: call mpv_nss_load and mpv_nss_set_graph first");
    if (!s || !y0hat || !gx || !samples) return mpv_fail(ctx, MPV_EINVAL, "mpv_nss_sample: NULL argument");
    StgEngine& e = *ctx->nss;
    if (s->n_windows < 1 || s->nodes != e.V || s->chunk < 1 || s->n_samples < 1 || s->n_samples % s->chunk != 0)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nss_sample: need n_windows >= 1, nodes = the graph's node count, n_samples a positive multiple of chunk");
    DeviceGuard dg(ctx->device);
    const int O = e.T0, F = e.d.nf, P = s->chunk, V = e.V, C = s->n_samples / P, Tn = e.steps_T;
    if (s->rows_per_pass < 0) return mpv_fail(ctx, MPV_EINVAL, "mpv_nss_sample: rows_per_pass must be >= 0");
    const long long PV = (long long)P * V, OF = (long long)O * F;
    // windows in blocks of at most rows_per_pass node rows (at least one window): the workspace is carved for ONE block, so a long window list
    // costs launches, not memory (the whole list used to be carved at once)
    const long long limit = s->rows_per_pass > 0 ? s->rows_per_pass : 32768;
    const long long wb_max = std::max<long long>(1, std::min<long long>(s->n_windows, limit / PV));
    if (wb_max * PV * OF > (long long)INT32_MAX * 256) return mpv_fail(ctx, MPV_EINVAL, "mpv_nss_sample: a window block exceeds the grid limit; lower rows_per_pass");
    STG_TRY(carve(ctx, e, wb_max * PV));
    hipStream_t st = ctx->stream;
    const unsigned seed_lo = (unsigned)s->seed, seed_hi = (unsigned)(s->seed >> 32);
    float *mu = e.xm, *g = e.gg, *y = e.xt, *z = e.noise;
    for (long long w0 = 0; w0 < s->n_windows; w0 += wb_max) {
        const long long wb = std::min<long long>(wb_max, s->n_windows - w0), N = wb * PV, n = N * OF;
        const float* inj = noise ? noise + (size_t)w0 * C * Tn * PV * OF : nullptr;
        float* out = samples + (size_t)w0 * V * OF * s->n_samples;
        hipLaunchKernelGGL(nss_tile_kernel, g1d(n), dim3(256), 0, st, y0hat + (size_t)w0 * V * OF, V, P, OF, n, mu);
        hipLaunchKernelGGL(nss_tile_kernel, g1d(n), dim3(256), 0, st, gx + (size_t)w0 * V * OF, V, P, OF, n, g);
        MPV_HIP_TRY(ctx, hipGetLastError());
        for (int c = 0; c < C; ++c) {                                                  // chunks of parallel_sample graph copies (NsDiff_model.py:755)
            hipLaunchKernelGGL(nss_noise_kernel, g1d(n), dim3(256), 0, st, inj, C, c, Tn, 0, PV, OF, seed_lo, seed_hi, w0, n, z);
            hipLaunchKernelGGL(nss_init_kernel, g1d(n), dim3(256), 0, st, g, z, mu, n, y);
            MPV_HIP_TRY(ctx, hipGetLastError());
            for (int t = Tn - 1; t >= 0; --t) {                                        // p_sample for t = T-1 .. 1, p_sample_t_1to0 at t = 0
                STG_TRY(forward_nss(ctx, e, y, mu, g, N, t));
                if (t > 0) hipLaunchKernelGGL(nss_noise_kernel, g1d(n), dim3(256), 0, st, inj, C, c, Tn, Tn - t, PV, OF, seed_lo, seed_hi, w0, n, z);   // drawn after the model call (:130)
                hipLaunchKernelGGL(nss_posterior_kernel, g1d(n), dim3(256), 0, st, e.coef + (size_t)t * MPV_COEF_COLS, mu, g, e.eps, e.sig, z, t == 0 ? 1 : 0, n, y);
                MPV_HIP_TRY(ctx, hipGetLastError());
            }
            hipLaunchKernelGGL(nss_gather_kernel, g1d(n), dim3(256), 0, st, y, V, P, OF, s->n_samples, c, n, out);
            MPV_HIP_TRY(ctx, hipGetLastError());
        }
    }
    return MPV_OK;
}

}  // extern "C"
