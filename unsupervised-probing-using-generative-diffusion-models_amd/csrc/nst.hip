// f(x): the non-stationary-transformer mean backbone of NsDiff, and TMDM's prior (the same network with a latent bottleneck),
// for gfx950.  One forward per window row, no gradient.
//
// Wiring follows reference models/Diffusion_model/NsDiff/mu_backbone.py:12-183 (Projector :12-41, Model.forward :150-183) and
// TMDM/tmdm_ns_transformer.py:136-172; the encoder / decoder / attention / embedding layers are torch-timeseries==0.1.10
// classes that the reference only imports (source absent): they are restated from the published Non-stationary Transformer
// definition, see oracle/ns_transformer_oracle.py -- PARITY UNPINNED for those layers.
//
// Activations are fp32 [rows][d_model], row = (window row b, position l).  Kernels:
//   nst_prepare_kernel   per b: gather the window from the resident series (scaler fused), series statistics, normalised window,
//                        both Projector MLPs (tau = exp(.), delta[seq_len])                                    HBM, tiny
//   nst_embed_kernel     circular k=3 token convolution + position table; the decoder input [x_norm[-label:], 0] is formed on the fly
//   nst_gemm_h3_kernel   Y = act(X W^T + b) (+ R): 128x128x32 LDS tiles; two-term f16 split of both operands, three
//                        v_mfma_f32_32x32x16_f16 per k-step, fp32 accumulate (fp32-equivalent, the default)       MFMA
//   nst_gemm_kernel      the same contraction on the exact-fp32 pipe (v_mfma_f32_32x32x2_f32), gemm = MPV_GEMM_F32  MFMA
//   nst_attn_kernel      de-stationary attention softmax(scale (q.k tau + delta)) v, causal option; online softmax (base 2) over
//                        128-key chunks, S^T accumulators reused as the B operand of the P.V product; v_mfma_f32_16x16x4_f32   MFMA
//   nst_ln_kernel        LayerNorm of a row (the residual is added in the producing GEMM's epilogue)           HBM
//   nst_head_kernel      decoder.norm + Linear(d_model, c_out) + de-normalisation, only the rows that are returned
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "mpv_common.h"

#include "nst_gemm.h"
#include "graph_aggr.h"

namespace {

constexpr float LOG2E = 1.44269504088896341f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// window gather + statistics + the two Projector MLPs                       (mu_backbone.py:12-41, :154-163)
// ---------------------------------------------------------------------------------------------------------------------
struct ProjW {
    const float* conv;        // [S][3]
    const float* w[5];        // hidden layers then the bias-free output layer
    const float* b[4];
};

struct PrepArgs {
    const float* series;      // [V][L][F]
    const float* mean;        // scaler [F] | null
    const float* std;
    long long L, first, step;
    int V, S, F, n_out_delta;
    int p_layers, p_dim[4];
    ProjW tau, delta;
    float* xn;                // [B][S][F]
    float* stats;             // [B][2][F]  (mean, std)
    float* tau_out;           // [B]
    float* delta_out;         // [B][S]
    long long b0;             // first window row of this pass
};

__device__ void projector_mlp(const PrepArgs& p, const ProjW& w, const float* conv_out, const float* stat, float* h0, float* h1, int n_out, float* out,
                              bool exp_out) {
    const int tid = threadIdx.x, nt = blockDim.x, F = p.F;
    // input [conv_out (F) | stat (F)]
    for (int i = tid; i < 2 * F; i += nt) h0[i] = i < F ? conv_out[i] : stat[i - F];
    __syncthreads();
    int n_in = 2 * F;
    float* src = h0;
    float* dst = h1;
    for (int l = 0; l < p.p_layers; ++l) {
        const int n = p.p_dim[l];
        for (int o = tid; o < n; o += nt) {
            float acc = w.b[l][o];
            const float* row = w.w[l] + (size_t)o * n_in;
            for (int i = 0; i < n_in; ++i) acc += row[i] * src[i];
            dst[o] = fmaxf(acc, 0.0f);
        }
        __syncthreads();
        float* t = src; src = dst; dst = t;
        n_in = n;
    }
    for (int o = tid; o < n_out; o += nt) {
        float acc = 0.0f;
        const float* row = w.w[p.p_layers] + (size_t)o * n_in;
        for (int i = 0; i < n_in; ++i) acc += row[i] * src[i];
        out[o] = exp_out ? expf(acc) : acc;
    }
    __syncthreads();
}

__global__ __launch_bounds__(128) void nst_prepare_kernel(const PrepArgs p) {
    extern __shared__ float sm[];
    const int S = p.S, F = p.F, tid = threadIdx.x, nt = blockDim.x;
    float* win = sm;                    // [S][F] scaled window (x_raw)
    float* red = win + S * F;           // [2][F] mean, std
    float* conv = red + 2 * F;          // [F]
    float* h0 = conv + F;               // [256]
    float* h1 = h0 + 256;               // [256]
    const long long b = p.b0 + blockIdx.x;
    const long long w = b / p.V;
    const int v = (int)(b % p.V);
    const float* src = p.series + ((size_t)v * p.L + p.first + w * p.step) * F;
    for (int i = tid; i < S * F; i += nt) {
        float x = src[i];
        if (p.mean) x = (x - p.mean[i % F]) / p.std[i % F];
        win[i] = x;
    }
    __syncthreads();
    // per-feature mean and population std over the window: one wave per feature, round-robin
    const int lane = tid & 63, wave = tid >> 6, n_waves = nt >> 6;
    for (int f = wave; f < F; f += n_waves) {
        float s = 0.0f;
        for (int t = lane; t < S; t += 64) s += win[t * F + f];
        const float m = wave_sum(s) / (float)S;
        float q = 0.0f;
        for (int t = lane; t < S; t += 64) { const float d = win[t * F + f] - m; q += d * d; }
        const float sd = sqrtf(wave_sum(q) / (float)S + 1e-5f);
        if (lane == 0) { red[f] = m; red[F + f] = sd; }
    }
    __syncthreads();
    float* xn = p.xn + (size_t)blockIdx.x * S * F;
    for (int i = tid; i < S * F; i += nt) xn[i] = (win[i] - red[i % F]) / red[F + i % F];
    if (tid < 2 * F) p.stats[(size_t)blockIdx.x * 2 * F + tid] = red[tid];
    // tau: conv over features (circular, k=3) with the time axis as input channels; stats row = std
    for (int pass = 0; pass < 2; ++pass) {
        const ProjW& pw = pass == 0 ? p.tau : p.delta;
        for (int e = wave; e < F; e += n_waves) {
            float s = 0.0f;
            const int em = (e + F - 1) % F, ep = (e + 1) % F;
            for (int t = lane; t < S; t += 64)
                s += pw.conv[t * 3 + 0] * win[t * F + em] + pw.conv[t * 3 + 1] * win[t * F + e] + pw.conv[t * 3 + 2] * win[t * F + ep];
            s = wave_sum(s);
            if (lane == 0) conv[e] = s;
        }
        __syncthreads();
        if (pass == 0) projector_mlp(p, pw, conv, red + F, h0, h1, 1, p.tau_out + blockIdx.x, true);
        else projector_mlp(p, pw, conv, red, h0, h1, p.n_out_delta, p.delta_out + (size_t)blockIdx.x * p.n_out_delta, false);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// DataEmbedding: circular k=3 token convolution + position table
// ---------------------------------------------------------------------------------------------------------------------
struct EmbedArgs {
    const float* xn;          // [B][S][F]
    const float* wt;          // [F][3][D]  (transposed at load)
    const float* pe;          // [rows][D]
    float* out;               // [B][Lx][D]
    int S, F, D, Lx, label;   // label < 0: encoder (Lx == S); else decoder input [xn[S-label:], zeros]
    long long n_rows;         // B * Lx
};

__global__ __launch_bounds__(256) void nst_embed_kernel(const EmbedArgs p) {
    const int dq = p.D >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.n_rows * dq) return;
    const long long row = idx / dq;
    const int c = (int)(idx % dq) * 4;
    const long long b = row / p.Lx;
    const int l = (int)(row % p.Lx);
    v4f acc = *reinterpret_cast<const v4f*>(p.pe + (size_t)l * p.D + c);
    v4f val = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        int pos = l + tap - 1;
        pos = pos < 0 ? pos + p.Lx : (pos >= p.Lx ? pos - p.Lx : pos);
        int src = pos;
        if (p.label >= 0) { if (pos >= p.label) continue; src = p.S - p.label + pos; }
        for (int f = 0; f < p.F; ++f) {
            const float x = p.xn[((size_t)b * p.S + src) * p.F + f];
            const v4f w = *reinterpret_cast<const v4f*>(p.wt + ((size_t)f * 3 + tap) * p.D + c);
            val += w * x;
        }
    }
    *reinterpret_cast<v4f*>(p.out + (size_t)row * p.D + c) = val + acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// de-stationary attention
// ---------------------------------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* Q; long long ldq;      // [B][L][ldq], head h at column h*E
    const float* K; long long ldk;      // [B][S][ldk]
    const float* V; long long ldv;
    const float* tau;                   // [B] | null
    const float* delta;                 // [B][S] | null
    float* O; long long ldo;
    int L, S, causal;
    float scale;
};

constexpr int A_KC = 64;                // keys per chunk (35 KB of LDS per workgroup: four per CU)
constexpr int A_QW = 1;                 // 16-query tiles per wave (2 measured no faster: the kernel is not LDS-read bound)
constexpr int A_QT = 64 * A_QW;         // queries per workgroup

template <int E>
__global__ __launch_bounds__(256, 4) void nst_attn_kernel(const AttnArgs p) {
    constexpr int KLD = E + 4, VLD = A_KC + 4, ES = E / 16;
    __shared__ float Ks[A_KC * KLD];
    __shared__ float Vt[E * VLD];
    __shared__ float Dl[A_KC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4;
    const int h = blockIdx.y;
    const long long b = blockIdx.z;
    const int q0 = blockIdx.x * A_QT;
    const float qs = p.scale * LOG2E * (p.tau ? p.tau[b] : 1.0f);
    const float ds = p.scale * LOG2E;
    // Q fragments of the wave's tiles: lane (query lq, g) holds Q[query][16 s + 4 g + i], pre-scaled
    int qrow[A_QW];
    bool live[A_QW];                    // the tile holds at least one real query (wave-uniform)
    v4f qf[A_QW][ES], o[A_QW][ES];      // O^T tiles: d = 16 dt + 4 g + r, column = query
    float m_run[A_QW], l_run[A_QW];
#pragma unroll
    for (int t = 0; t < A_QW; ++t) {
        const int qt0 = q0 + (wave * A_QW + t) * 16;
        qrow[t] = qt0 + lq;
        live[t] = qt0 < p.L;
        const int qclamp = qrow[t] < p.L ? qrow[t] : p.L - 1;
        const float* qp = p.Q + ((size_t)b * p.L + qclamp) * p.ldq + (size_t)h * E + 4 * g;
#pragma unroll
        for (int s = 0; s < ES; ++s) {
            qf[t][s] = *reinterpret_cast<const v4f*>(qp + 16 * s) * qs;
            o[t][s] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        }
        m_run[t] = -INFINITY;
        l_run[t] = 0.0f;
    }
    const int s_end = p.causal ? min(p.S, q0 + A_QT) : p.S;
    for (int k0 = 0; k0 < s_end; k0 += A_KC) {
        // only the 16-key tiles that hold a key this workgroup (n_kt) / this query tile (w_kt, causal) can see are staged and multiplied
        const int n_kt = min(A_KC / 16, (s_end - k0 + 15) >> 4);
        int w_kt[A_QW];
#pragma unroll
        for (int t = 0; t < A_QW; ++t) {
            const int qt0 = q0 + (wave * A_QW + t) * 16;
            w_kt[t] = !live[t] ? 0 : (p.causal ? max(0, min(n_kt, ((qt0 + 15 - k0) >> 4) + 1)) : n_kt);
        }
        int w_max = 0;
#pragma unroll
        for (int t = 0; t < A_QW; ++t) w_max = max(w_max, w_kt[t]);
        __syncthreads();                // previous chunk fully consumed
        // stage K chunk [key][E] and V chunk transposed [e][key]: every load of the chunk is issued before the first LDS write, so the
        // global latency is paid once per chunk and not once per 256 float4
        {
            constexpr int NST = A_KC * (E / 4) / 256;
            const int n_st = n_kt * 16 * (E / 4);
            v4f kr[NST], vr[NST];
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const int i = tid + 256 * j;
                if (i < n_st) {
                    const int key = i / (E / 4), e4 = (i % (E / 4)) * 4;
                    const int src = k0 + key < p.S ? k0 + key : p.S - 1;
                    kr[j] = *reinterpret_cast<const v4f*>(p.K + ((size_t)b * p.S + src) * p.ldk + (size_t)h * E + e4);
                    vr[j] = *reinterpret_cast<const v4f*>(p.V + ((size_t)b * p.S + src) * p.ldv + (size_t)h * E + e4);
                }
            }
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const int i = tid + 256 * j;
                if (i < n_st) {
                    const int key = i / (E / 4), e4 = (i % (E / 4)) * 4;
                    *reinterpret_cast<v4f*>(Ks + key * KLD + e4) = kr[j];
                    Vt[(e4 + 0) * VLD + key] = vr[j][0];
                    Vt[(e4 + 1) * VLD + key] = vr[j][1];
                    Vt[(e4 + 2) * VLD + key] = vr[j][2];
                    Vt[(e4 + 3) * VLD + key] = vr[j][3];
                }
            }
        }
        if (tid < A_KC) Dl[tid] = (p.delta && k0 + tid < p.S) ? p.delta[(size_t)b * p.S + k0 + tid] * ds : 0.0f;
        __syncthreads();
        // S^T tiles: rows = keys (16 kt + 4 g + r), column = query
        v4f sc[A_QW][A_KC / 16];
        float cmax[A_QW];
#pragma unroll
        for (int t = 0; t < A_QW; ++t) cmax[t] = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < A_KC / 16; ++kt) {
            if (kt >= w_max) continue;
            v4f acc[A_QW];
#pragma unroll
            for (int t = 0; t < A_QW; ++t) acc[t] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
            const float* kp = Ks + (kt * 16 + lq) * KLD + 4 * g;
#pragma unroll
            for (int s = 0; s < ES; ++s) {
                const v4f kf = *reinterpret_cast<const v4f*>(kp + 16 * s);
#pragma unroll
                for (int t = 0; t < A_QW; ++t) {
                    if (kt >= w_kt[t]) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i], qf[t][s][i], acc[t], 0, 0, 0);
                }
            }
            const v4f dl = *reinterpret_cast<const v4f*>(Dl + kt * 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < A_QW; ++t) {
                if (kt >= w_kt[t]) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kt * 16 + 4 * g + r;
                    float v = acc[t][r] + dl[r];
                    if (key >= p.S || (p.causal && key > qrow[t])) v = -INFINITY;
                    acc[t][r] = v;
                    cmax[t] = fmaxf(cmax[t], v);
                }
                sc[t][kt] = acc[t];
            }
        }
#pragma unroll
        for (int t = 0; t < A_QW; ++t) {
            if (w_kt[t] == 0) continue;
            float cm = cmax[t];
            cm = fmaxf(cm, __shfl_xor(cm, 16));
            cm = fmaxf(cm, __shfl_xor(cm, 32));
            const float m_new = fmaxf(m_run[t], cm);
            const float alpha = exp2f(m_run[t] - m_new);            // (every live query sees key 0 in chunk 0, so m_new is finite)
            m_run[t] = m_new;
            float psum = 0.0f;
#pragma unroll
            for (int kt = 0; kt < A_KC / 16; ++kt) {
                if (kt >= w_kt[t]) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(sc[t][kt][r] - m_new);
                    sc[t][kt][r] = e;
                    psum += e;
                }
            }
            psum += __shfl_xor(psum, 16);
            psum += __shfl_xor(psum, 32);
            l_run[t] = l_run[t] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < ES; ++dt) o[t][dt] *= alpha;
        }
        // O^T[d][query] += V^T[d][key] P^T[key][query]
#pragma unroll
        for (int kt = 0; kt < A_KC / 16; ++kt) {
            if (kt >= w_max) continue;
#pragma unroll
            for (int dt = 0; dt < ES; ++dt) {
                const v4f vf = *reinterpret_cast<const v4f*>(Vt + (dt * 16 + lq) * VLD + kt * 16 + 4 * g);
#pragma unroll
                for (int t = 0; t < A_QW; ++t) {
                    if (kt >= w_kt[t]) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r], sc[t][kt][r], o[t][dt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < A_QW; ++t) {
        if (qrow[t] < p.L) {
            const float inv = 1.0f / l_run[t];
            float* op = p.O + ((size_t)b * p.L + qrow[t]) * p.ldo + (size_t)h * E + 4 * g;
#pragma unroll
            for (int dt = 0; dt < ES; ++dt) *reinterpret_cast<v4f*>(op + 16 * dt) = o[t][dt] * inv;
        }
    }
}

// Round 4: the same kernel on the f16 matrix pipe (gemm mode f16x3): K and V^T are staged as two-term f16 planes (hi = 11 significant bits, lo = x - hi,
// split ONCE per chunk for the four waves that share it), the pre-scaled Q fragments are split once per kernel, P is split where it leaves the
// softmax -- its S^T accumulator layout (rows = keys 4 g + r, column = query) IS the B operand of v_mfma_f32_16x16x16_f16, no LDS round trip.  Per
// (query tile, key tile) and 16 head columns: 3 MFMAs of 16x16x16 for q.k and 3 for p.v where the fp32 pipe needs 4 + 4 of 16x16x4 at four times the
// cycles each; softmax, running maximum, de-stationary factors and the output are fp32 as before.
typedef _Float16 ah4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void attn_split4(const v4f& v, ah4& hi, ah4& lo) {
    float h[4], l[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        h[c] = __uint_as_float(__float_as_uint(v[c]) & 0xFFFFE000u);
        l[c] = v[c] - h[c];
    }
    typedef __fp16 pk2 __attribute__((ext_vector_type(2)));
    const pk2 h01 = __builtin_amdgcn_cvt_pkrtz(h[0], h[1]), h23 = __builtin_amdgcn_cvt_pkrtz(h[2], h[3]);
    const pk2 l01 = __builtin_amdgcn_cvt_pkrtz(l[0], l[1]), l23 = __builtin_amdgcn_cvt_pkrtz(l[2], l[3]);
    hi = ah4{(_Float16)h01[0], (_Float16)h01[1], (_Float16)h23[0], (_Float16)h23[1]};
    lo = ah4{(_Float16)l01[0], (_Float16)l01[1], (_Float16)l23[0], (_Float16)l23[1]};
}
#define NST_MMH(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x16f16((a), (b), (acc), 0, 0, 0)
#ifndef AH_KC
#define AH_KC 64                        // keys per chunk of the split-f16 kernel
#endif
#ifndef AH_WGS
#define AH_WGS 4
#endif

template <int E>
__global__ __launch_bounds__(256, AH_WGS) void nst_attn_h_kernel(const AttnArgs p) {
    constexpr int KLD = E + 8, VLD = AH_KC + 8, ES = E / 16;              // halves per row: 16 rows x two 8-byte pieces of a fragment read cover all 64 banks
    extern __shared__ __attribute__((aligned(16))) unsigned char ah_smem[];
    _Float16* Kh = reinterpret_cast<_Float16*>(ah_smem);                  // [AH_KC][KLD]
    _Float16* Kl = Kh + AH_KC * KLD;
    _Float16* Vh = Kl + AH_KC * KLD;                                      // transposed: [E][VLD]
    _Float16* Vl = Vh + E * VLD;
    float* Dl = reinterpret_cast<float*>(Vl + E * VLD);                   // [AH_KC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4;
    const int h = blockIdx.y;
    const long long b = blockIdx.z;
    const int q0 = blockIdx.x * A_QT;
    const float qs = p.scale * LOG2E * (p.tau ? p.tau[b] : 1.0f);
    const float ds = p.scale * LOG2E;
    const int qt0 = q0 + wave * 16, qrow = qt0 + lq;
    const bool live = qt0 < p.L;                                          // (wave-uniform)
    ah4 qh[ES], ql[ES];
    v4f o[ES];                                                            // O^T tiles: d = 16 dt + 4 g + r, column = query
    {
        const int qclamp = qrow < p.L ? qrow : p.L - 1;
        const float* qp = p.Q + ((size_t)b * p.L + qclamp) * p.ldq + (size_t)h * E + 4 * g;
#pragma unroll
        for (int s = 0; s < ES; ++s) {
            attn_split4(*reinterpret_cast<const v4f*>(qp + 16 * s) * qs, qh[s], ql[s]);
            o[s] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    float m_run = -INFINITY, l_run = 0.0f;
    const int s_end = p.causal ? min(p.S, q0 + A_QT) : p.S;
    for (int k0 = 0; k0 < s_end; k0 += AH_KC) {
        const int n_kt = min(AH_KC / 16, (s_end - k0 + 15) >> 4);
        const int w_kt = !live ? 0 : (p.causal ? max(0, min(n_kt, ((qt0 + 15 - k0) >> 4) + 1)) : n_kt);
        __syncthreads();                // previous chunk fully consumed
        {
            constexpr int NST = AH_KC * (E / 4) / 256;
            const int n_st = n_kt * 16 * (E / 4);
            v4f kr[NST], vr[NST];
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const int i = min(tid + 256 * j, n_st - 1);              // (unconditional loads: a spare thread repeats the last piece)
                const int key = i / (E / 4), e4 = (i % (E / 4)) * 4;
                const int src = k0 + key < p.S ? k0 + key : p.S - 1;
                kr[j] = *reinterpret_cast<const v4f*>(p.K + ((size_t)b * p.S + src) * p.ldk + (size_t)h * E + e4);
                vr[j] = *reinterpret_cast<const v4f*>(p.V + ((size_t)b * p.S + src) * p.ldv + (size_t)h * E + e4);
            }
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const int i = tid + 256 * j;
                if (i < n_st) {
                    const int key = i / (E / 4), e4 = (i % (E / 4)) * 4;
                    ah4 hi, lo;
                    attn_split4(kr[j], hi, lo);
                    *reinterpret_cast<ah4*>(Kh + key * KLD + e4) = hi;
                    *reinterpret_cast<ah4*>(Kl + key * KLD + e4) = lo;
                    attn_split4(vr[j], hi, lo);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        Vh[(e4 + c) * VLD + key] = hi[c];
                        Vl[(e4 + c) * VLD + key] = lo[c];
                    }
                }
            }
        }
        if (tid < AH_KC) Dl[tid] = (p.delta && k0 + tid < p.S) ? p.delta[(size_t)b * p.S + k0 + tid] * ds : 0.0f;
        __syncthreads();
        // S^T tiles: rows = keys (16 kt + 4 g + r), column = query
        v4f sc[AH_KC / 16];
        float cmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < AH_KC / 16; ++kt) {
            if (kt >= w_kt) continue;
            v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
            const int ko = (kt * 16 + lq) * KLD + 4 * g;
#pragma unroll
            for (int s = 0; s < ES; ++s) {
                const ah4 kh = *reinterpret_cast<const ah4*>(Kh + ko + 16 * s), kl = *reinterpret_cast<const ah4*>(Kl + ko + 16 * s);
                NST_MMH(acc, kl, qh[s]);
                NST_MMH(acc, kh, ql[s]);
                NST_MMH(acc, kh, qh[s]);
            }
            const v4f dl = *reinterpret_cast<const v4f*>(Dl + kt * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kt * 16 + 4 * g + r;
                float v = acc[r] + dl[r];
                if (key >= p.S || (p.causal && key > qrow)) v = -INFINITY;
                acc[r] = v;
                cmax = fmaxf(cmax, v);
            }
            sc[kt] = acc;
        }
        if (w_kt > 0) {
            float cm = cmax;
            cm = fmaxf(cm, __shfl_xor(cm, 16));
            cm = fmaxf(cm, __shfl_xor(cm, 32));
            const float m_new = fmaxf(m_run, cm);
            const float alpha = exp2f(m_run - m_new);                   // (every live query sees key 0 in chunk 0, so m_new is finite)
            m_run = m_new;
            float psum = 0.0f;
#pragma unroll
            for (int kt = 0; kt < AH_KC / 16; ++kt) {
                if (kt >= w_kt) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(sc[kt][r] - m_new);
                    sc[kt][r] = e;
                    psum += e;
                }
            }
            psum += __shfl_xor(psum, 16);
            psum += __shfl_xor(psum, 32);
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < ES; ++dt) o[dt] *= alpha;
        }
        // O^T[d][query] += V^T[d][key] P^T[key][query]
#pragma unroll
        for (int kt = 0; kt < AH_KC / 16; ++kt) {
            if (kt >= w_kt) continue;
            ah4 ph, pl;
            attn_split4(sc[kt], ph, pl);
#pragma unroll
            for (int dt = 0; dt < ES; ++dt) {
                const int vo = (dt * 16 + lq) * VLD + kt * 16 + 4 * g;
                const ah4 vh = *reinterpret_cast<const ah4*>(Vh + vo), vl = *reinterpret_cast<const ah4*>(Vl + vo);
                NST_MMH(o[dt], vl, ph);
                NST_MMH(o[dt], vh, pl);
                NST_MMH(o[dt], vh, ph);
            }
        }
    }
    if (qrow < p.L) {
        const float inv = 1.0f / l_run;
        float* op = p.O + ((size_t)b * p.L + qrow) * p.ldo + (size_t)h * E + 4 * g;
#pragma unroll
        for (int dt = 0; dt < ES; ++dt) *reinterpret_cast<v4f*>(op + 16 * dt) = o[dt] * inv;
    }
}
template <int E>
constexpr size_t nst_attn_h_lds() { return 2 * sizeof(_Float16) * ((size_t)AH_KC * (E + 8) + (size_t)E * (AH_KC + 8)) + sizeof(float) * AH_KC; }

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over d_model (one wave per row), and the output head
// ---------------------------------------------------------------------------------------------------------------------
struct LnArgs {
    const float* X; float* Y;
    const float* w; const float* b;
    long long rows;
    int D;
};

constexpr int LN_MAXQ = 4;               // float4 per lane: d_model <= 1024

__device__ __forceinline__ void ln_row(const float* x, int D, const float* w, const float* bia, int lane, v4f (&v)[LN_MAXQ]) {
    const int nq = D >> 2;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        const int q = lane + 64 * j;
        v[j] = q < nq ? *reinterpret_cast<const v4f*>(x + 4 * q) : v4f{0.0f, 0.0f, 0.0f, 0.0f};
        s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
    const float mean = wave_sum(s) / (float)D;
    float qd = 0.0f;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        const int q = lane + 64 * j;
        if (q < nq) {
            v[j] -= mean;
            qd += v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3];
        }
    }
    const float rstd = rsqrtf(wave_sum(qd) / (float)D + 1e-5f);
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        const int q = lane + 64 * j;
        if (q < nq) v[j] = v[j] * rstd * *reinterpret_cast<const v4f*>(w + 4 * q) + *reinterpret_cast<const v4f*>(bia + 4 * q);
    }
}

__global__ __launch_bounds__(256) void nst_ln_kernel(const LnArgs p) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int lane = threadIdx.x & 63;
    v4f v[LN_MAXQ];
    ln_row(p.X + (size_t)row * p.D, p.D, p.w, p.b, lane, v);
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        const int q = lane + 64 * j;
        if (q < (p.D >> 2)) *reinterpret_cast<v4f*>(p.Y + (size_t)row * p.D + 4 * q) = v[j];
    }
}

struct HeadArgs {
    const float* X;           // decoder output [B][Ld][D]
    const float* nw; const float* nb;     // decoder.norm
    const float* pw; const float* pb;     // projection [C][D], [C]
    const float* stats;       // [B][2][F]
    float* out;               // [B][n_keep][C]
    int D, C, Ld, n_keep, F;
    long long rows;           // B * n_keep
};

__global__ __launch_bounds__(256) void nst_head_kernel(const HeadArgs p) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int lane = threadIdx.x & 63;
    const long long b = row / p.n_keep;
    const int l = p.Ld - p.n_keep + (int)(row % p.n_keep);
    v4f v[LN_MAXQ];
    ln_row(p.X + ((size_t)b * p.Ld + l) * p.D, p.D, p.nw, p.nb, lane, v);
    for (int c = 0; c < p.C; ++c) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < LN_MAXQ; ++j) {
            const int q = lane + 64 * j;
            if (q < (p.D >> 2)) {
                const v4f w = *reinterpret_cast<const v4f*>(p.pw + (size_t)c * p.D + 4 * q);
                s += v[j][0] * w[0] + v[j][1] * w[1] + v[j][2] * w[2] + v[j][3] * w[3];
            }
        }
        s = wave_sum(s) + p.pb[c];
        if (lane == 0) p.out[(size_t)row * p.C + c] = s * p.stats[(size_t)b * 2 * p.F + p.F + c] + p.stats[(size_t)b * 2 * p.F + c];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------------------------------------
// ResGatedGraphConv aggregation + ReLU (SpatialBlock.forward, mu_backbone.py:43-51; torch_geometric's published definition, restated:
// out_i = W_skip x_i + sum_{j -> i} sigmoid(W_k x_i + W_q x_j) * W_v x_j + bias).  kqvs [rows][4C] = key | query | value | skip; rows = consecutive
// blocks of V nodes (one block per window); incoming edges of a node in edge order = the index_add_ order of the reference.
__global__ __launch_bounds__(256) void nst_gate_kernel(const float* __restrict__ kqvs, const float* __restrict__ bias, const int* __restrict__ row_ptr,
                                                        const int* __restrict__ col_idx, int V, int C4, long long n4, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // row * C/4 + channel quad
    if (i >= n4) return;
    const int cq = (int)(i % C4), C = 4 * C4;
    const long long row = i / C4;
    const int v = (int)(row % V);
    const long long base = row - v;                                             // first row of this window's graph
    const v4f k = *reinterpret_cast<const v4f*>(kqvs + row * 4 * C + 4 * cq);
    v4f a = {0, 0, 0, 0};
    for (int e = row_ptr[v]; e < row_ptr[v + 1]; ++e) {
        const float* src = kqvs + (base + col_idx[e]) * 4 * C + 4 * cq;
        const v4f q = *reinterpret_cast<const v4f*>(src + C), val = *reinterpret_cast<const v4f*>(src + 2 * C);
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] += val[c] / (1.0f + __expf(-(k[c] + q[c])));
    }
    const v4f skip = *reinterpret_cast<const v4f*>(kqvs + row * 4 * C + 3 * C + 4 * cq), b = *reinterpret_cast<const v4f*>(bias + 4 * cq);
    v4f o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = fmaxf((a[c] + skip[c]) + b[c], 0.0f);
    *reinterpret_cast<v4f*>(out + row * C + 4 * cq) = o;
}

struct Mat {                                                 // one Linear weight [N][K]: fp32, and its scaled two-term f16 split
    const float* w = nullptr;
    const _Float16* h = nullptr;
    const _Float16* l = nullptr;
    float inv_scale = 1.0f;
    int K = 0;
};
struct AttnW { Mat wqkv; const float* bqkv; Mat wo; const float* bo; };       // wqkv rows: q (D), k (D), v (D)
struct FfnW { Mat w1; const float* b1; Mat w2; const float* b2; };
struct NormW { const float *w, *b; };
struct EncLayer { AttnW att; FfnW ffn; NormW n1, n2; };
struct DecLayer { AttnW self, cross; FfnW ffn; NormW n1, n2, n3; };
// Model_spatial (NsDiff/mu_backbone.py:186-340): one gated graph convolution over the fT_h x d_model features of a (window, node) row --
// the four Linear maps as ONE [4C][C] product (key | query | value | skip; the skip map has no bias), then the gather over incoming edges
struct SpatialLayer { Mat w4; const float* b4; const float* bias; };       // (w4: the layer's [ng][C] product -- 4 C columns for the gated layer)

struct NstEngine {
    mpv_nst_desc d{};
    float* blob = nullptr;
    const float *enc_tok = nullptr, *enc_pe = nullptr, *dec_tok = nullptr, *dec_pe = nullptr;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
    NormW enc_norm{}, dec_norm{};
    const float *proj_w = nullptr, *proj_b = nullptr;
    ProjW tau{}, delta{};
    Mat zw[4];
    const float* zb[4] = {};
    // spatial stage between encoder and decoder (spatial mode: d.fT_h > 0): k = S + 1 convolution along time as a dense [fT_h D] x [S D]
    // product (its fT_h output steps are shifted copies of the kernel, laid out at load time), graph convolutions, transposed convolution back
    Mat wdown, wup;
    const float *bdown = nullptr, *bup = nullptr;             // biases repeated per output step: [fT_h][D], [S][D]
    std::vector<SpatialLayer> sp;
    int gV = 0;                                               // the graph (CSR by target node, incoming edges in edge order), mpv_nst_set_graph
    long long gE = 0;
    int *row_ptr = nullptr, *col_idx = nullptr, *deg = nullptr;
    float *sp_a = nullptr, *sp_b = nullptr, *sp_kqvs = nullptr;
    _Float16* blob16 = nullptr;
    int gemm_mode = MPV_GEMM_F16X3;
    float* ws = nullptr;
    size_t ws_floats = 0;
    long long cap_b = 0;
    // workspace (per pass of cap_b window rows)
    float *xn = nullptr, *stats = nullptr, *tau_v = nullptr, *delta_v = nullptr, *ex = nullptr, *ey = nullptr, *dx = nullptr, *dy = nullptr, *qkv = nullptr,
          *att = nullptr, *hid = nullptr;
    long long last_b = 0;
};

void nst_engine_destroy(NstEngine* e) {
    if (!e) return;
    if (e->blob) (void)hipFree(e->blob);
    if (e->blob16) (void)hipFree(e->blob16);
    if (e->ws) (void)hipFree(e->ws);
    if (e->row_ptr) (void)hipFree(e->row_ptr);
    if (e->col_idx) (void)hipFree(e->col_idx);
    if (e->deg) (void)hipFree(e->deg);
    delete e;
}

namespace {

#define NST_TRY(expr)                     \
    do {                                  \
        if (int rc_ = (expr)) return rc_; \
    } while (0)

size_t per_row_floats(const mpv_nst_desc& d) {
    const size_t S = d.seq_len, Ld = d.label_len + d.pred_len, D = d.d_model, F = d.enc_in;
    const size_t Lm = std::max(S, Ld);
    return S * F + 2 * F + 1 + S          // xn, stats, tau, delta
           + 2 * S * D + 2 * Ld * D       // ex, ey, dx, dy
           + Lm * 3 * D + S * 2 * D       // qkv (self) ; cross kv shares the tail of the same buffer
           + Lm * D + Lm * std::max<size_t>(d.d_ff, D) + 64
           + 6 * (size_t)d.fT_h * D;     // spatial stage: two feature buffers + key | query | value | skip
}

int carve(mpv_ctx* ctx, NstEngine& e, long long B) {
    if (B <= e.cap_b) return MPV_OK;
    const mpv_nst_desc& d = e.d;
    const size_t S = d.seq_len, Ld = d.label_len + d.pred_len, D = d.d_model, F = d.enc_in, Lm = std::max(S, Ld);
    size_t total = 0;
    auto take = [&](size_t n) { const size_t at = total; total += (n + 3) & ~size_t(3); return at; };
    const size_t o_xn = take(B * S * F), o_st = take(B * 2 * F), o_tau = take(B), o_del = take(B * S), o_ex = take(B * S * D), o_ey = take(B * S * D),
                 o_dx = take(B * Ld * D), o_dy = take(B * Ld * D), o_qkv = take(B * (Lm * 3 * D + S * 2 * D)), o_att = take(B * Lm * D),
                 o_hid = take(B * Lm * std::max<size_t>(d.d_ff, D));
    const size_t C = (size_t)d.fT_h * D, o_spa = take(B * C), o_spb = take(B * C), o_spk = take(B * 4 * C);
    if (total > e.ws_floats) {
        MPV_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (e.ws) (void)hipFree(e.ws);
        e.ws = nullptr;
        e.ws_floats = 0;
        e.cap_b = 0;
        if (hipMalloc(&e.ws, total * sizeof(float)) != hipSuccess) return mpv_fail(ctx, MPV_ENOMEM, "mpv_nst_forward: workspace allocation failed");
        e.ws_floats = total;
    }
    e.xn = e.ws + o_xn; e.stats = e.ws + o_st; e.tau_v = e.ws + o_tau; e.delta_v = e.ws + o_del; e.ex = e.ws + o_ex; e.ey = e.ws + o_ey;
    e.dx = e.ws + o_dx; e.dy = e.ws + o_dy; e.qkv = e.ws + o_qkv; e.att = e.ws + o_att; e.hid = e.ws + o_hid;
    e.sp_a = e.ws + o_spa; e.sp_b = e.ws + o_spb; e.sp_kqvs = e.ws + o_spk;
    e.cap_b = B;
    return MPV_OK;
}

// Y = act(X W[row0 : row0 + N]^T + bias) + R
int gemm(mpv_ctx* ctx, const NstEngine& e, const float* X, long long ldx, const Mat& W, int row0, const float* bias, const float* R, long long ldr, float* Y,
         long long ldy, long long M, int N, int act) {
    const int K = W.K;
    const long long tiles = ((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
    const dim3 grid((unsigned)(8 * ((tiles + 7) / 8)));
    if (e.gemm_mode == MPV_GEMM_F32) {
        NstGemmArgs a{X, ldx, W.w + (size_t)row0 * K, bias, R, ldr, Y, ldy, M, N, K, act};
        const size_t lds = sizeof(float) * 2 * (G_BM + G_BN) * G_LD;
        static bool attr_set = false;
        if (!attr_set) {
            MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(nst_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(nst_gemm_kernel, grid, dim3(256), lds, ctx->stream, a);
    } else {
        NstGemmHArgs a{X, ldx, W.h + (size_t)row0 * K, W.l + (size_t)row0 * K, W.inv_scale, bias, R, ldr, Y, ldy, M, N, K, act, K};
        nst_gemm_h3_launch(a, ctx->stream);
    }
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

int layer_norm(mpv_ctx* ctx, const float* X, float* Y, const NormW& n, long long rows, int D) {
    LnArgs a{X, Y, n.w, n.b, rows, D};
    hipLaunchKernelGGL(nst_ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

int attention(mpv_ctx* ctx, const NstEngine& e, const float* Q, long long ldq, const float* K, long long ldk, const float* V, long long ldv, const float* delta,
              float* O, long long B, int L, int S, int causal) {
    const int H = e.d.n_heads, E = e.d.d_model / H;
    AttnArgs a{Q, ldq, K, ldk, V, ldv, e.tau_v, delta, O, (long long)e.d.d_model, L, S, causal, 1.0f / sqrtf((float)E)};
    const dim3 grid((unsigned)((L + A_QT - 1) / A_QT), (unsigned)H, (unsigned)B);
    static const bool attn_f32 = getenv("MPV_NST_ATTN_F32") && atoi(getenv("MPV_NST_ATTN_F32")) != 0;      // (A/B aid: the fp32-pipe kernel in f16x3 mode too)
    if (e.gemm_mode != MPV_GEMM_F32 && !attn_f32) {                  // two-term f16 products, fp32-equivalent (like the Linear layers of this mode)
#define NST_ATTN_H(E_)                                                                                                                        \
    do {                                                                                                                                      \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(nst_attn_h_kernel<E_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                           (int)nst_attn_h_lds<E_>());                                                       \
        (void)attr;                                                                                                                           \
        hipLaunchKernelGGL(nst_attn_h_kernel<E_>, grid, dim3(256), nst_attn_h_lds<E_>(), ctx->stream, a);                                     \
    } while (0)
        if (E == 64) NST_ATTN_H(64);
        else if (E == 32) NST_ATTN_H(32);
        else NST_ATTN_H(16);
#undef NST_ATTN_H
        MPV_HIP_TRY(ctx, hipGetLastError());
        return MPV_OK;
    }
    if (E == 64) hipLaunchKernelGGL(nst_attn_kernel<64>, grid, dim3(256), 0, ctx->stream, a);
    else if (E == 32) hipLaunchKernelGGL(nst_attn_kernel<32>, grid, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(nst_attn_kernel<16>, grid, dim3(256), 0, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

int ffn(mpv_ctx* ctx, NstEngine& e, const FfnW& f, const float* x, float* y, long long rows) {
    const int D = e.d.d_model, dff = e.d.d_ff, act = e.d.activation ? 2 : 1;
    NST_TRY(gemm(ctx, e, x, D, f.w1, 0, f.b1, nullptr, 0, e.hid, dff, rows, dff, act));
    return gemm(ctx, e, e.hid, dff, f.w2, 0, f.b2, x, D, y, D, rows, D, 0);           // + residual
}

// one pass over B window rows starting at global row b0
int forward_pass(mpv_ctx* ctx, NstEngine& e, const float* series, int V, long long L, long long first, long long step, const float* mean, const float* std,
                 long long b0, long long B, int full, float* out) {
    const mpv_nst_desc& d = e.d;
    const int S = d.seq_len, Ld = d.label_len + d.pred_len, D = d.d_model, F = d.enc_in;
    PrepArgs pa{};
    pa.series = series; pa.mean = mean; pa.std = std; pa.L = L; pa.first = first; pa.step = step; pa.V = V; pa.S = S; pa.F = F; pa.n_out_delta = S;
    pa.p_layers = d.p_layers;
    for (int i = 0; i < 4; ++i) pa.p_dim[i] = d.p_hidden[i];
    pa.tau = e.tau; pa.delta = e.delta; pa.xn = e.xn; pa.stats = e.stats; pa.tau_out = e.tau_v; pa.delta_out = e.delta_v; pa.b0 = b0;
    const size_t prep_lds = sizeof(float) * ((size_t)S * F + 3 * F + 512);
    hipLaunchKernelGGL(nst_prepare_kernel, dim3((unsigned)B), dim3(128), prep_lds, ctx->stream, pa);
    MPV_HIP_TRY(ctx, hipGetLastError());
    // encoder
    {
        EmbedArgs ea{e.xn, e.enc_tok, e.enc_pe, e.ex, S, F, D, S, -1, B * S};
        hipLaunchKernelGGL(nst_embed_kernel, dim3((unsigned)((B * S * (D / 4) + 255) / 256)), dim3(256), 0, ctx->stream, ea);
        MPV_HIP_TRY(ctx, hipGetLastError());
    }
    const long long Me = B * S, Md = B * Ld;
    for (const EncLayer& l : e.enc) {
        NST_TRY(gemm(ctx, e, e.ex, D, l.att.wqkv, 0, l.att.bqkv, nullptr, 0, e.qkv, 3 * D, Me, 3 * D, 0));
        NST_TRY(attention(ctx, e, e.qkv, 3 * D, e.qkv + D, 3 * D, e.qkv + 2 * D, 3 * D, e.delta_v, e.att, B, S, S, 0));
        NST_TRY(gemm(ctx, e, e.att, D, l.att.wo, 0, l.att.bo, e.ex, D, e.ey, D, Me, D, 0));
        NST_TRY(layer_norm(ctx, e.ey, e.ex, l.n1, Me, D));
        NST_TRY(ffn(ctx, e, l.ffn, e.ex, e.ey, Me));
        NST_TRY(layer_norm(ctx, e.ey, e.ex, l.n2, Me, D));
    }
    NST_TRY(layer_norm(ctx, e.ex, e.ey, e.enc_norm, Me, D));
    float* enc_out = e.ey;
    if (d.latent) {                       // tmdm_ns_transformer.py:155-162, eval mode: z = z_mean(enc); enc = z_out(z)
        NST_TRY(gemm(ctx, e, e.ey, D, e.zw[0], 0, e.zb[0], nullptr, 0, e.ex, D, Me, D, 1));
        NST_TRY(gemm(ctx, e, e.ex, D, e.zw[1], 0, e.zb[1], nullptr, 0, e.ey, D, Me, D, 0));
        NST_TRY(gemm(ctx, e, e.ey, D, e.zw[2], 0, e.zb[2], nullptr, 0, e.ex, D, Me, D, 1));
        NST_TRY(gemm(ctx, e, e.ex, D, e.zw[3], 0, e.zb[3], nullptr, 0, e.ey, D, Me, D, 0));
    }
    if (d.fT_h > 0) {                     // Model_spatial.forward, mu_backbone.py:312-325 (B is a whole number of V-node windows here)
        const int C = d.fT_h * D;
        NST_TRY(gemm(ctx, e, enc_out, (long long)S * D, e.wdown, 0, e.bdown, nullptr, 0, e.sp_a, C, B, C, 0));         // downsampling: [B][S D] -> [B][fT_h D]
        float *cur = e.sp_a, *nxt = e.sp_b;
        const int gnn = d.spatial_gnn ? d.spatial_gnn - MPV_STG_RES_SAGE_MEAN + 1 : 0, H = std::max(d.spatial_heads, 1), ng = graph_conv_ng(gnn, C, H);
        for (const SpatialLayer& l : e.sp) {
            NST_TRY(gemm(ctx, e, cur, C, l.w4, 0, l.b4, nullptr, 0, e.sp_kqvs, ng, B, ng, 0));
            const long long n4 = B * (C / 4);
            if (gnn == 0) hipLaunchKernelGGL(nst_gate_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx->stream, e.sp_kqvs, l.bias, e.row_ptr, e.col_idx, e.gV, C / 4, n4, nxt);
            else graph_aggr_launch(gnn, ctx->stream, e.sp_kqvs, ng, l.bias, e.row_ptr, e.col_idx, e.deg, e.gV, C, H, d.spatial_slope, B, nxt);
            MPV_HIP_TRY(ctx, hipGetLastError());
            std::swap(cur, nxt);
        }
        NST_TRY(gemm(ctx, e, cur, C, e.wup, 0, e.bup, nullptr, 0, e.ex, (long long)S * D, B, S * D, 0));               // upsampling: -> [B][S D]
        enc_out = e.ex;
    }
    // decoder
    {
        EmbedArgs ea{e.xn, e.dec_tok, e.dec_pe, e.dx, S, F, D, Ld, d.label_len, B * Ld};
        hipLaunchKernelGGL(nst_embed_kernel, dim3((unsigned)((B * Ld * (D / 4) + 255) / 256)), dim3(256), 0, ctx->stream, ea);
        MPV_HIP_TRY(ctx, hipGetLastError());
    }
    float* ckv = e.qkv + (size_t)B * std::max(S, Ld) * 3 * D;       // cross K | V  [B*S][2D]
    for (const DecLayer& l : e.dec) {
        NST_TRY(gemm(ctx, e, e.dx, D, l.self.wqkv, 0, l.self.bqkv, nullptr, 0, e.qkv, 3 * D, Md, 3 * D, 0));
        NST_TRY(attention(ctx, e, e.qkv, 3 * D, e.qkv + D, 3 * D, e.qkv + 2 * D, 3 * D, nullptr, e.att, B, Ld, Ld, 1));
        NST_TRY(gemm(ctx, e, e.att, D, l.self.wo, 0, l.self.bo, e.dx, D, e.dy, D, Md, D, 0));
        NST_TRY(layer_norm(ctx, e.dy, e.dx, l.n1, Md, D));
        NST_TRY(gemm(ctx, e, e.dx, D, l.cross.wqkv, 0, l.cross.bqkv, nullptr, 0, e.qkv, D, Md, D, 0));                      // q
        NST_TRY(gemm(ctx, e, enc_out, D, l.cross.wqkv, D, l.cross.bqkv + D, nullptr, 0, ckv, 2 * D, Me, 2 * D, 0));       // k | v
        NST_TRY(attention(ctx, e, e.qkv, D, ckv, 2 * D, ckv + D, 2 * D, e.delta_v, e.att, B, Ld, S, 0));
        NST_TRY(gemm(ctx, e, e.att, D, l.cross.wo, 0, l.cross.bo, e.dx, D, e.dy, D, Md, D, 0));
        NST_TRY(layer_norm(ctx, e.dy, e.dx, l.n2, Md, D));
        NST_TRY(ffn(ctx, e, l.ffn, e.dx, e.dy, Md));
        NST_TRY(layer_norm(ctx, e.dy, e.dx, l.n3, Md, D));
    }
    const int n_keep = full ? Ld : d.pred_len;
    HeadArgs ha{e.dx, e.dec_norm.w, e.dec_norm.b, e.proj_w, e.proj_b, e.stats, out, D, d.c_out, Ld, n_keep, F, B * n_keep};
    hipLaunchKernelGGL(nst_head_kernel, dim3((unsigned)((B * n_keep + 3) / 4)), dim3(256), 0, ctx->stream, ha);
    MPV_HIP_TRY(ctx, hipGetLastError());
    e.last_b = B;
    return MPV_OK;
}

}  // namespace

extern "C" {

int mpv_nst_load(mpv_ctx* ctx, const mpv_nst_desc* d, const float* const* t, int32_t n_tensors) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_nst_load: ctx is NULL");
    if (!d || !t) return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: NULL argument");
    const int D = d->d_model, F = d->enc_in, S = d->seq_len, H = d->n_heads, dff = d->d_ff, Ld = d->label_len + d->pred_len;
    if (F < 1 || F > 64 || d->c_out != F || S < 2 || d->label_len < 0 || d->label_len > S || d->pred_len < 1 || H < 1 || D % H != 0 || d->e_layers < 0 ||
        d->d_layers < 1 || d->p_layers < 1 || d->p_layers > 4)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: bad shape (need c_out == enc_in <= 64, label_len <= seq_len, 1 <= p_hidden_layers <= 4)");
    const int E = D / H;
    if ((E != 16 && E != 32 && E != 64) || D % 32 != 0 || D > 1024 || dff % 32 != 0)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: head width d_model / n_heads must be 16, 32 or 64, d_model and d_ff multiples of 32, d_model <= 1024");
    for (int i = 0; i < d->p_layers; ++i)
        if (d->p_hidden[i] < 1 || d->p_hidden[i] > 256) return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: projector hidden widths must be in [1, 256]");
    if (2 * F > 256) return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: too many features");
    if (d->pe_rows < std::max(S, Ld)) return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: position table shorter than the sequences");
    if (d->fT_h < 0 || d->spatial_layers < 0 || (d->fT_h == 0 && d->spatial_layers > 0) || (d->fT_h % 2) != 0 || d->fT_h > S || (d->fT_h > 0 && d->latent))
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: spatial mode needs an even 0 < fT_h <= seq_len, spatial_layers >= 0 and no latent bottleneck");
    if (d->fT_h > 0 && (double)d->fT_h * D * (double)S * D > 1.0e9)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: the dense [fT_h d_model] x [seq_len d_model] image of the k = seq_len + 1 convolution exceeds 1e9 entries");
    const int n_proj = 2 + 2 * d->p_layers;
    const int expect = 4 + 16 * d->e_layers + 2 + 26 * d->d_layers + 4 + 2 * n_proj + (d->latent ? 8 : 0) + (d->fT_h > 0 ? 4 + 8 * d->spatial_layers : 0);
    if (n_tensors != expect) return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: expected " + std::to_string(expect) + " tensors, got " + std::to_string(n_tensors));
    const int sp_first = expect - (d->fT_h > 0 ? 2 + 8 * d->spatial_layers : 0), sp_end = expect - 2;            // the graph layers' slots
    if (d->spatial_gnn != 0 && (d->fT_h <= 0 || d->spatial_gnn < MPV_STG_RES_SAGE_MEAN || d->spatial_gnn > MPV_STG_RES_GAT))
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: spatial_gnn must be 0 or one of MPV_STG_RES_SAGE_MEAN .. MPV_STG_RES_GAT (with fT_h > 0)");
    if (d->spatial_gnn == MPV_STG_RES_GAT && (d->spatial_heads < 1 || (d->fT_h * D) % d->spatial_heads != 0))
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: GATConv heads must divide fT_h d_model");
    for (int i = 0; i < n_tensors; ++i) {
        bool optional = false;
        if (d->spatial_gnn != 0 && i >= sp_first && i < sp_end) {                  // kinds other than the gated layer use a prefix of the 8 slots
            const int slot = (i - sp_first) % 8;
            optional = slot >= 1 && !(d->spatial_gnn == MPV_STG_RES_GAT && (slot == 2 || slot == 3));
        }
        if (!t[i] && !optional) return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_load: tensor " + std::to_string(i) + " is NULL");
    }
    DeviceGuard dg(ctx->device);
    MPV_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    nst_engine_destroy(ctx->nst);
    ctx->nst = nullptr;
    NstEngine* e = new (std::nothrow) NstEngine();
    if (!e) return mpv_fail(ctx, MPV_ENOMEM, "mpv_nst_load: out of host memory");
    e->d = *d;
    std::vector<float> host;
    auto put = [&](const float* p, size_t n) { const size_t at = host.size(); host.insert(host.end(), p, p + n); host.resize((host.size() + 3) & ~size_t(3)); return at; };
    // a Linear weight [N][K]: fp32 copy + two-term f16 split of w * 2^s, s chosen so the largest entry lands in [2^13, 2^14)
    std::vector<_Float16> host16;
    struct MatO { size_t w, h, l; float inv; int K; };
    auto put_mat = [&](const float* w, size_t N, int K) {
        MatO o{};
        o.w = put(w, N * K);
        o.K = K;
        float mx = 0.0f;
        for (size_t i = 0; i < N * K; ++i) mx = std::max(mx, std::fabs(w[i]));
        int ex = 0;
        if (mx > 0.0f && std::isfinite(mx)) { (void)std::frexp(mx, &ex); ex = 14 - ex; }      // mx * 2^ex in [2^13, 2^14)
        const float sc = std::ldexp(1.0f, ex);
        o.inv = std::ldexp(1.0f, -ex);
        o.h = host16.size();
        host16.resize(host16.size() + 2 * N * K);
        o.l = o.h + N * K;
        for (size_t i = 0; i < N * K; ++i) {
            const float v = w[i] * sc;
            uint32_t bits;
            std::memcpy(&bits, &v, 4);
            bits &= 0xFFFFE000u;
            float hi;
            std::memcpy(&hi, &bits, 4);
            host16[o.h + i] = (_Float16)hi;
            host16[o.l + i] = (_Float16)(v - hi);
        }
        return o;
    };
    int k = 0;
    auto tok = [&](const float* w) {       // [D][F][3] -> [F][3][D]
        std::vector<float> o((size_t)F * 3 * D);
        for (int c = 0; c < D; ++c)
            for (int f = 0; f < F; ++f)
                for (int tap = 0; tap < 3; ++tap) o[((size_t)f * 3 + tap) * D + c] = w[((size_t)c * F + f) * 3 + tap];
        return put(o.data(), o.size());
    };
    struct AttnO { MatO wqkv; size_t bqkv; MatO wo; size_t bo; };
    struct FfnO { MatO w1; size_t b1; MatO w2; size_t b2; };
    struct NormO { size_t w, b; };
    auto attn = [&]() {
        AttnO o;
        std::vector<float> w((size_t)3 * D * D), b((size_t)3 * D);
        for (int part = 0; part < 3; ++part) {
            std::memcpy(&w[(size_t)part * D * D], t[k + 2 * part], sizeof(float) * D * D);
            std::memcpy(&b[(size_t)part * D], t[k + 2 * part + 1], sizeof(float) * D);
        }
        o.wqkv = put_mat(w.data(), (size_t)3 * D, D);
        o.bqkv = put(b.data(), b.size());
        o.wo = put_mat(t[k + 6], D, D);
        o.bo = put(t[k + 7], D);
        k += 8;
        return o;
    };
    auto ffn_o = [&]() {
        FfnO o{put_mat(t[k], dff, D), put(t[k + 1], dff), put_mat(t[k + 2], D, dff), put(t[k + 3], D)};
        k += 4;
        return o;
    };
    auto norm = [&]() {
        NormO o{put(t[k], D), put(t[k + 1], D)};
        k += 2;
        return o;
    };
    const size_t o_etok = tok(t[0]), o_epe = put(t[1], (size_t)d->pe_rows * D), o_dtok = tok(t[2]), o_dpe = put(t[3], (size_t)d->pe_rows * D);
    k = 4;
    struct EncO { AttnO a; FfnO f; NormO n1, n2; };
    struct DecO { AttnO s, c; FfnO f; NormO n1, n2, n3; };
    std::vector<EncO> eo(d->e_layers);
    for (EncO& o : eo) { o.a = attn(); o.f = ffn_o(); o.n1 = norm(); o.n2 = norm(); }
    const NormO o_en = norm();
    std::vector<DecO> dofs(d->d_layers);
    for (DecO& o : dofs) { o.s = attn(); o.c = attn(); o.f = ffn_o(); o.n1 = norm(); o.n2 = norm(); o.n3 = norm(); }
    const NormO o_dn = norm();
    const size_t o_pw = put(t[k], (size_t)d->c_out * D), o_pb = put(t[k + 1], d->c_out);
    k += 2;
    struct ProjO { size_t conv, w[5], b[4]; };
    auto proj = [&](int n_out) {
        ProjO o{};
        o.conv = put(t[k++], (size_t)S * 3);
        int n_in = 2 * F;
        for (int i = 0; i < d->p_layers; ++i) {
            o.w[i] = put(t[k++], (size_t)d->p_hidden[i] * n_in);
            o.b[i] = put(t[k++], d->p_hidden[i]);
            n_in = d->p_hidden[i];
        }
        o.w[d->p_layers] = put(t[k++], (size_t)n_out * n_in);
        return o;
    };
    const ProjO o_tau = proj(1), o_delta = proj(S);
    MatO o_zw[4] = {};
    size_t o_zb[4] = {};
    if (d->latent)
        for (int i = 0; i < 4; ++i) { o_zw[i] = put_mat(t[k + 2 * i], D, D); o_zb[i] = put(t[k + 2 * i + 1], D); }
    if (d->latent) k += 8;
    // spatial stage: downsampling.{weight [D][D][1][S+1], bias}, per layer gnn.{lin_key, lin_query, lin_value}.{weight [C][C], bias}, gnn.lin_skip.weight,
    // gnn.bias; upsampling.{weight [D][D][1][S+1] (ConvTranspose2d: [in][out]), bias}
    MatO o_wdown{}, o_wup{};
    size_t o_bdown = 0, o_bup = 0;
    struct SpO { MatO w4; size_t b4, bias; };
    std::vector<SpO> spo(d->fT_h > 0 ? d->spatial_layers : 0);
    if (d->fT_h > 0) {
        const int Tn = d->fT_h, pd = Tn / 2, KW = S + 1, C = Tn * D;
        {
            const float* w = t[k];                            // Conv2d cross-correlation: out[to] = sum_k w[k] x[to + k - pd]  =>  k = ti - to + pd
            std::vector<float> m((size_t)C * S * D, 0.0f), b((size_t)C);
            for (int to = 0; to < Tn; ++to)
                for (int co = 0; co < D; ++co) {
                    float* row = &m[((size_t)to * D + co) * S * D];
                    for (int ti = 0; ti < S; ++ti) {
                        const int kk = ti - to + pd;
                        if (kk < 0 || kk >= KW) continue;
                        for (int ci = 0; ci < D; ++ci) row[(size_t)ti * D + ci] = w[((size_t)co * D + ci) * KW + kk];
                    }
                    b[(size_t)to * D + co] = t[k + 1][co];
                }
            o_wdown = put_mat(m.data(), (size_t)C, S * D);
            o_bdown = put(b.data(), b.size());
            k += 2;
        }
        const int gnn = d->spatial_gnn ? d->spatial_gnn - MPV_STG_RES_SAGE_MEAN + 1 : 0, H = std::max(d->spatial_heads, 1), ng = graph_conv_ng(gnn, C, H);
        for (SpO& o : spo) {
            std::vector<float> w, b, gb;
            graph_conv_pack(gnn, C, H, &t[k], w, b, gb);      // the 8 slots of the layer (include/mpv_hip.h): one [ng][C] product, its bias, the bias behind the gather
            o.w4 = put_mat(w.data(), (size_t)ng, C);
            o.b4 = put(b.data(), b.size());
            o.bias = put(gb.data(), gb.size());
            k += 8;
        }
        {
            const float* w = t[k];                            // ConvTranspose2d (stride 1): out[t] = sum_{to, k: t = to + k - pd} x[to] w[k]  =>  k = t - to + pd
            std::vector<float> m((size_t)S * D * C, 0.0f), b((size_t)S * D);
            for (int tt = 0; tt < S; ++tt)
                for (int co = 0; co < D; ++co) {
                    float* row = &m[((size_t)tt * D + co) * C];
                    for (int to = 0; to < Tn; ++to) {
                        const int kk = tt - to + pd;
                        if (kk < 0 || kk >= KW) continue;
                        for (int ci = 0; ci < D; ++ci) row[(size_t)to * D + ci] = w[((size_t)ci * D + co) * KW + kk];
                    }
                    b[(size_t)tt * D + co] = t[k + 1][co];
                }
            o_wup = put_mat(m.data(), (size_t)S * D, C);
            o_bup = put(b.data(), b.size());
            k += 2;
        }
    }
    if (hipMalloc(&e->blob, host.size() * sizeof(float)) != hipSuccess) { delete e; return mpv_fail(ctx, MPV_ENOMEM, "mpv_nst_load: weight blob"); }
    if (hipMalloc(&e->blob16, host16.size() * sizeof(_Float16)) != hipSuccess) { nst_engine_destroy(e); return mpv_fail(ctx, MPV_ENOMEM, "mpv_nst_load: f16 weight blob"); }
    if (hipMemcpy(e->blob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->blob16, host16.data(), host16.size() * sizeof(_Float16), hipMemcpyHostToDevice) != hipSuccess) {
        nst_engine_destroy(e);
        return mpv_fail(ctx, MPV_EHIP, "mpv_nst_load: upload failed");
    }
    auto P = [&](size_t off) { return (const float*)(e->blob + off); };
    auto M_ = [&](const MatO& o) { Mat m; m.w = P(o.w); m.h = e->blob16 + o.h; m.l = e->blob16 + o.l; m.inv_scale = o.inv; m.K = o.K; return m; };
    auto A = [&](const AttnO& o) { return AttnW{M_(o.wqkv), P(o.bqkv), M_(o.wo), P(o.bo)}; };
    auto Fw = [&](const FfnO& o) { return FfnW{M_(o.w1), P(o.b1), M_(o.w2), P(o.b2)}; };
    auto N = [&](const NormO& o) { return NormW{P(o.w), P(o.b)}; };
    e->enc_tok = P(o_etok); e->enc_pe = P(o_epe); e->dec_tok = P(o_dtok); e->dec_pe = P(o_dpe);
    for (const EncO& o : eo) e->enc.push_back(EncLayer{A(o.a), Fw(o.f), N(o.n1), N(o.n2)});
    for (const DecO& o : dofs) e->dec.push_back(DecLayer{A(o.s), A(o.c), Fw(o.f), N(o.n1), N(o.n2), N(o.n3)});
    e->enc_norm = N(o_en); e->dec_norm = N(o_dn);
    e->proj_w = P(o_pw); e->proj_b = P(o_pb);
    auto PW = [&](const ProjO& o) {
        ProjW w{};
        w.conv = P(o.conv);
        for (int i = 0; i <= d->p_layers; ++i) w.w[i] = P(o.w[i]);
        for (int i = 0; i < d->p_layers; ++i) w.b[i] = P(o.b[i]);
        return w;
    };
    e->tau = PW(o_tau); e->delta = PW(o_delta);
    if (d->latent)
        for (int i = 0; i < 4; ++i) { e->zw[i] = M_(o_zw[i]); e->zb[i] = P(o_zb[i]); }
    if (d->fT_h > 0) {
        e->wdown = M_(o_wdown); e->bdown = P(o_bdown); e->wup = M_(o_wup); e->bup = P(o_bup);
        for (const SpO& o : spo) e->sp.push_back(SpatialLayer{M_(o.w4), P(o.b4), P(o.bias)});
    }
    ctx->nst = e;
    return MPV_OK;
}

/* The ONE graph of Model_spatial's graph convolutions (reference: the `edge_index` argument of Model_spatial.forward, mu_backbone.py:296): stored as
 * CSR by target node with every node's incoming edges in edge order; every window of a call is one copy of it. */
int mpv_nst_set_graph(mpv_ctx* ctx, int32_t n_nodes, int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_nst_set_graph: ctx is NULL");
    if (!ctx->nst) return mpv_fail(ctx, MPV_ESTATE, "mpv_nst_set_graph: call mpv_nst_load first");
    NstEngine& e = *ctx->nst;
    if (e.d.fT_h <= 0) return mpv_fail(ctx, MPV_ESTATE, "mpv_nst_set_graph: the loaded backbone has no spatial stage (fT_h = 0)");
    if (n_nodes < 1 || n_edges < 0 || (n_edges > 0 && (!edge_src || !edge_dst))) return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_set_graph: bad argument");
    std::vector<int> rp((size_t)n_nodes + 1, 0), ci((size_t)n_edges);
    for (int64_t i = 0; i < n_edges; ++i) {
        if (edge_src[i] < 0 || edge_src[i] >= n_nodes || edge_dst[i] < 0 || edge_dst[i] >= n_nodes)
            return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_set_graph: edge " + std::to_string(i) + " names a node outside [0, n_nodes)");
        ++rp[(size_t)edge_dst[i] + 1];
    }
    for (int v = 0; v < n_nodes; ++v) rp[v + 1] += rp[v];
    std::vector<int> fill(rp.begin(), rp.end() - 1);
    for (int64_t i = 0; i < n_edges; ++i) ci[(size_t)fill[(size_t)edge_dst[i]]++] = (int)edge_src[i];       // stable: edge order within a target
    DeviceGuard dg(ctx->device);
    MPV_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> indeg((size_t)n_nodes, 0);                       // in-degree without self loops (GCNConv's normalisation)
    for (int64_t i = 0; i < n_edges; ++i)
        if (edge_src[i] != edge_dst[i]) ++indeg[(size_t)edge_dst[i]];
    if (e.row_ptr) (void)hipFree(e.row_ptr);
    if (e.col_idx) (void)hipFree(e.col_idx);
    if (e.deg) (void)hipFree(e.deg);
    e.row_ptr = e.col_idx = e.deg = nullptr;
    e.gV = 0;
    if (hipMalloc(&e.row_ptr, sizeof(int) * rp.size()) != hipSuccess || hipMalloc(&e.col_idx, sizeof(int) * std::max<size_t>(1, ci.size())) != hipSuccess ||
        hipMalloc(&e.deg, sizeof(int) * indeg.size()) != hipSuccess)
        return mpv_fail(ctx, MPV_ENOMEM, "mpv_nst_set_graph: graph buffers");
    MPV_HIP_TRY(ctx, hipMemcpy(e.row_ptr, rp.data(), sizeof(int) * rp.size(), hipMemcpyHostToDevice));
    MPV_HIP_TRY(ctx, hipMemcpy(e.deg, indeg.data(), sizeof(int) * indeg.size(), hipMemcpyHostToDevice));
    if (!ci.empty()) MPV_HIP_TRY(ctx, hipMemcpy(e.col_idx, ci.data(), sizeof(int) * ci.size(), hipMemcpyHostToDevice));
    e.gV = n_nodes;
    e.gE = n_edges;
    return MPV_OK;
}

int mpv_nst_forward(mpv_ctx* ctx, const float* series, int32_t nodes, int64_t series_len, int32_t n_windows, int64_t first, int64_t step, const float* mean,
                    const float* std, int32_t full, int32_t gemm, int64_t rows_per_pass, float* y0hat) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_nst_forward: ctx is NULL");
    if (!ctx->nst) return mpv_fail(ctx, MPV_ESTATE, "mpv_nst_forward: call mpv_nst_load first");
    NstEngine& e = *ctx->nst;
    if (!series || !y0hat || nodes < 1 || n_windows < 1 || first < 0 || step < 0 || (mean == nullptr) != (std == nullptr))
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_forward: bad argument");
    if (gemm != MPV_GEMM_F16X3 && gemm != MPV_GEMM_F32) return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_forward: gemm must be MPV_GEMM_F16X3 or MPV_GEMM_F32");
    e.gemm_mode = gemm;
    if (first + (int64_t)(n_windows - 1) * step + e.d.seq_len > series_len)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_forward: the last window runs past the end of the series");
    if (e.d.fT_h > 0 && (e.gV < 1 || e.gV != nodes))
        return mpv_fail(ctx, MPV_ESTATE, "mpv_nst_forward: spatial backbone: call mpv_nst_set_graph with a graph of `nodes` nodes first");
    DeviceGuard dg(ctx->device);
    const long long B = (long long)n_windows * nodes;
    long long pass = rows_per_pass > 0 ? rows_per_pass : std::max<long long>(1, (long long)((6.0 * (1ull << 30) / sizeof(float)) / (double)per_row_floats(e.d)));
    if (e.d.fT_h > 0) pass = std::max<long long>(nodes, pass / nodes * nodes);       // the graph convolution couples the nodes of a window: whole windows per pass
    pass = std::min(pass, B);
    NST_TRY(carve(ctx, e, pass));
    const int n_keep = full ? e.d.label_len + e.d.pred_len : e.d.pred_len;
    for (long long b0 = 0; b0 < B; b0 += pass) {
        const long long nb = std::min(pass, B - b0);
        NST_TRY(forward_pass(ctx, e, series, nodes, series_len, first, step, mean, std, b0, nb, full, y0hat + (size_t)b0 * n_keep * e.d.c_out));
    }
    return MPV_OK;
}

int mpv_nst_debug_tensor(mpv_ctx* ctx, int32_t which, float* out, int64_t capacity) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_nst_debug_tensor: ctx is NULL");
    if (!ctx->nst || ctx->nst->last_b < 1) return mpv_fail(ctx, MPV_ESTATE, "mpv_nst_debug_tensor: run mpv_nst_forward first");
    NstEngine& e = *ctx->nst;
    const long long B = e.last_b, S = e.d.seq_len, F = e.d.enc_in, D = e.d.d_model;
    const float* src = nullptr;
    long long n = 0;
    switch (which) {
        case 0: src = e.xn; n = B * S * F; break;
        case 1: src = e.stats; n = B * 2 * F; break;
        case 2: src = e.tau_v; n = B; break;
        case 3: src = e.delta_v; n = B * S; break;
        case 4: src = e.d.fT_h > 0 ? e.ex : e.ey; n = B * S * D; break;     // what the decoder cross-attends to: the encoder output (after the latent
                                                                            // bottleneck / the spatial stage when present)
        case 5: src = e.sp.size() % 2 ? e.sp_b : e.sp_a; n = B * e.d.fT_h * D; break;      // spatial features after the last graph convolution
        default: return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_debug_tensor: unknown tensor id");
    }
    if (!out || capacity < n) return mpv_fail(ctx, MPV_EINVAL, "mpv_nst_debug_tensor: output too small (need " + std::to_string(n) + " floats)");
    DeviceGuard dg(ctx->device);
    MPV_HIP_TRY(ctx, hipMemcpyAsync(out, src, sizeof(float) * n, hipMemcpyDeviceToDevice, ctx->stream));
    return (int)0;
}

}  // extern "C"
