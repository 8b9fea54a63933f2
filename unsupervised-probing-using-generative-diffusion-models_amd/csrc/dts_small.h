// DiffusionTS: the small per-sequence / per-token kernels around the MFMA ones (embedding and inverse k=3 convolutions,
// trend polynomial block, Fourier season layer, mean removal, the output tail) and the elementwise steps of the DDIM +
// Langevin-infill sampler.  All VALU work; each is a few hundred FLOP per token.  Reference line numbers: see each kernel.
#pragma once
#include "chain_common.h"
#include "dts_kernels.h"
#include "dts_attn_h.h"

namespace dts {

// ---- Conv_MLP embedding (model_utils.py:158-169): emb = Conv1d(F -> D, k=3, pad=1) over tokens; + the two learnable
// positional encodings (transformer.py:423-427).  thread per (token, d).
__global__ void dts_embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w /*[D][F][3]*/, const float* __restrict__ bias,
                                     const float* __restrict__ pe_enc, const float* __restrict__ pe_dec, int L, int F, int D,
                                     long long n, float* h_enc, float* h_dec) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % D);
    const long long tok = i / D;
    const int l = (int)(tok % L);
    float a = bias[d];
    for (int f = 0; f < F; ++f)
        for (int tap = 0; tap < 3; ++tap) {
            const int ll = l + tap - 1;
            if (ll >= 0 && ll < L) a += w[(d * F + f) * 3 + tap] * x[(tok + tap - 1) * F + f];
        }
    h_enc[i] = a + pe_enc[l * D + d];
    h_dec[i] = a + pe_dec[l * D + d];
}

// d_x[b][l][f] = sum_d sum_tap w[d][f][tap] (g1 + g2)[b][l - tap + 1][d]; 16 lanes per output, channels strided over the lanes
__global__ void dts_embed_bwd_kernel(const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ w, int L, int F, int D,
                                     long long n, float* dx) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;       // (tok, f)
    const int li = threadIdx.x & 15;
    const bool live = i < n;
    const int f = live ? (int)(i % F) : 0;
    const long long tok = live ? i / F : 0;
    const int l = (int)(tok % L);
    float a = 0.0f;
    for (int tap = 0; tap < 3; ++tap) {
        const int ll = l - tap + 1;
        if (!live || ll < 0 || ll >= L) continue;
        const float* ga = g1 + (tok - tap + 1) * D;
        const float* gb = g2 + (tok - tap + 1) * D;
        for (int d = li; d < D; d += 16) a += w[(d * F + f) * 3 + tap] * (ga[d] + gb[d]);
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m);
    if (live && li == 0) dx[i] = a;
}

// ---- decoder block tail (transformer.py:330-332): m = mean over tokens, out = x - m, mean_i = Linear(m).  block per sequence.
__global__ __launch_bounds__(256) void dts_meansub_fwd_kernel(const float* __restrict__ x3, const float* __restrict__ wlin /*[F][D]*/,
                                                              const float* __restrict__ blin, int L, int F, int D, float* xout,
                                                              float* mean_i /*[B][n_dec][F] slice*/, int mean_stride) {
    extern __shared__ float sm[];                 // part[256], mean[D]
    float* part = sm;
    float* mean = sm + 256;
    const int b = blockIdx.x;
    const float* xb = x3 + (long long)b * L * D;
    const int d = threadIdx.x % D, rg = threadIdx.x / D, ngroups = 256 / D;
    constexpr int KEEP = 64;                        // a thread's values stay in registers between the sum and the subtraction (L <= 64 * 256 / D)
    const bool keep = 256 % D == 0 && (L + ngroups - 1) / ngroups <= KEEP;
    float v[KEEP];
    float s = 0.0f;
    if (keep) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u) v[u] = rg + u * ngroups < L ? xb[(rg + u * ngroups) * D + d] : 0.0f;
#pragma unroll
        for (int u = 0; u < KEEP; ++u) s += v[u];
    } else {
        for (int l = rg; l < L; l += ngroups) s += xb[l * D + d];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < D) {
        float t = 0.0f;
        for (int k = 0; k < ngroups; ++k) t += part[k * D + threadIdx.x];
        mean[threadIdx.x] = t / (float)L;
    }
    __syncthreads();
    if (keep) {
        const float m = mean[d];
#pragma unroll
        for (int u = 0; u < KEEP; ++u)
            if (rg + u * ngroups < L) xout[(long long)b * L * D + (rg + u * ngroups) * D + d] = v[u] - m;
    } else {
        for (int i = threadIdx.x; i < L * D; i += 256) xout[(long long)b * L * D + i] = xb[i] - mean[i % D];
    }
    for (int f = threadIdx.x; f < F; f += 256) {
        float a = blin[f];
        for (int k = 0; k < D; ++k) a += wlin[f * D + k] * mean[k];
        mean_i[(long long)b * mean_stride + f] = a;
    }
}

// d_x3 = d_out - mean_l(d_out) + (1/L) W^T d_mean
__global__ __launch_bounds__(256) void dts_meansub_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ wlin,
                                                              const float* __restrict__ dmean /*[B][stride] slice*/, int mean_stride, int L, int F,
                                                              int D, float* dx3) {
    extern __shared__ float sm[];                 // part[256], corr[D]
    float* part = sm;
    float* corr = sm + 256;
    const int b = blockIdx.x;
    const float* gb = dout + (long long)b * L * D;
    const int d = threadIdx.x % D, rg = threadIdx.x / D, ngroups = 256 / D;
    constexpr int KEEP = 64;
    const bool keep = 256 % D == 0 && (L + ngroups - 1) / ngroups <= KEEP;
    float v[KEEP];
    float s = 0.0f;
    if (keep) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u) v[u] = rg + u * ngroups < L ? gb[(rg + u * ngroups) * D + d] : 0.0f;
#pragma unroll
        for (int u = 0; u < KEEP; ++u) s += v[u];
    } else {
        for (int l = rg; l < L; l += ngroups) s += gb[l * D + d];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < D) {
        float t = 0.0f, sum = 0.0f;
        for (int k = 0; k < ngroups; ++k) sum += part[k * D + threadIdx.x];
        for (int f = 0; f < F; ++f) t += wlin[f * D + threadIdx.x] * dmean[(long long)b * mean_stride + f];
        corr[threadIdx.x] = (t - sum) / (float)L;
    }
    __syncthreads();
    if (keep) {
        const float c = corr[d];
#pragma unroll
        for (int u = 0; u < KEEP; ++u)
            if (rg + u * ngroups < L) dx3[(long long)b * L * D + (rg + u * ngroups) * D + d] = v[u] + c;
    } else {
        for (int i = threadIdx.x; i < L * D; i += 256) dx3[(long long)b * L * D + i] = gb[i] + corr[i % D];
    }
}

// ---- TrendBlock (transformer.py:12-34).  x1 = rows [0, L) of the projected [2L][D] block (token axis = conv channels).
//   t3[p][d] = b0[p] + sum_l sum_tap W0[p][l][tap] x1[l][d+tap-1]          Conv1d(L -> 3, k=3, pad=1) along D
//   tv[f][p] = b3[f] + sum_c sum_tap W3[f][c][tap] gelu(t3)[p+tap-1][c]     Conv1d(D -> F, k=3, pad=1) along the 3 poly terms
//   trend[l][f] += sum_p tv[f][p] poly[p][l]
__global__ __launch_bounds__(256) void dts_trend_fwd_kernel(const float* __restrict__ pr, long long pr_stride, const float* __restrict__ w0,
                                                            const float* __restrict__ b0, const float* __restrict__ w3, const float* __restrict__ b3,
                                                            const float* __restrict__ poly /*[3][L]*/, int L, int F, int D, float* t3_save /*[B][3][D]*/,
                                                            float* trend_sum /*[B][L][F]*/) {
    extern __shared__ float sm[];                 // a[3][D], tv[F][3], part[4][3][D]
    float* a = sm;
    float* tv = sm + 3 * D;
    float* part = tv + 3 * F;
    const int b = blockIdx.x;
    const float* x1 = pr + (long long)b * pr_stride;
    // the 3 D outputs are 600-term sums: four token ranges per output (all 256 threads busy), loads issued 8 tokens ahead
    // (one thread per (token range, channel) forms all three poly terms from one read of its x values)
    const int Lq = (L + 3) / 4;
    for (int i = threadIdx.x; i < 4 * D; i += 256) {
        const int q = i / D, d = i - q * D;
        const int l1 = min(L, (q + 1) * Lq);
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll 8
        for (int l = q * Lq; l < l1; ++l) {
            const float* xr = x1 + (long long)l * D;
            const float xm = d > 0 ? xr[d - 1] : 0.0f, xc = xr[d], xp = d + 1 < D ? xr[d + 1] : 0.0f;
            const float* wr = w0 + l * 3;
            s0 += wr[0] * xm;
            s0 += wr[1] * xc;
            s0 += wr[2] * xp;
            wr += L * 3;
            s1 += wr[0] * xm;
            s1 += wr[1] * xc;
            s1 += wr[2] * xp;
            wr += L * 3;
            s2 += wr[0] * xm;
            s2 += wr[1] * xc;
            s2 += wr[2] * xp;
        }
        part[q * 3 * D + d] = s0;
        part[q * 3 * D + D + d] = s1;
        part[q * 3 * D + 2 * D + d] = s2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * D; i += 256) {
        const float s = b0[i] + ((part[i] + part[3 * D + i]) + (part[6 * D + i] + part[9 * D + i]));     // bias per (p, d): the composite built at load
        t3_save[(long long)b * 3 * D + i] = s;
        a[i] = gelu_f(s);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F * 3; i += 256) {
        const int f = i / 3, p = i - f * 3;
        float s = b3[f];
        for (int c = 0; c < D; ++c)
            for (int tap = 0; tap < 3; ++tap) {
                const int pp = p + tap - 1;
                if (pp >= 0 && pp < 3) s += w3[(f * D + c) * 3 + tap] * a[pp * D + c];
            }
        tv[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * F; i += 256) {
        const int l = i / F, f = i - l * F;
        const float v = (tv[f * 3] * poly[l] + tv[f * 3 + 1] * poly[L + l]) + tv[f * 3 + 2] * poly[2 * L + l];
        trend_sum[(long long)b * L * F + i] += v;
    }
}

// g [B][L][F] -> d_x2 (through the composite taps) ADDED to dpr [B][L][D]
__global__ __launch_bounds__(256) void dts_trend_bwd_kernel(const float* __restrict__ g, const float* __restrict__ t3_save, const float* __restrict__ w0,
                                                            const float* __restrict__ w3, const float* __restrict__ poly, int L, int F, int D,
                                                            float* dpr, long long pr_stride) {
    extern __shared__ float sm[];                 // dtv[F][3], dt3[3][D]
    float* dtv = sm;
    float* dt3 = sm + F * 3;
    const int b = blockIdx.x;
    __shared__ float red[4];
    for (int i = 0; i < F * 3; ++i) {            // 3 F sums over the tokens, each reduced across the workgroup
        const int f = i / 3, p = i - f * 3;
        float s = 0.0f;
        for (int l = threadIdx.x; l < L; l += 256) s += g[((long long)b * L + l) * F + f] * poly[p * L + l];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) dtv[i] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 3 * D; i += 256) {
        const int p = i / D, c = i - p * D;
        float s = 0.0f;
        for (int f = 0; f < F; ++f)
            for (int tap = 0; tap < 3; ++tap) {
                const int po = p - tap + 1;                          // output position whose tap reads a[p]
                if (po >= 0 && po < 3) s += w3[(f * D + c) * 3 + tap] * dtv[f * 3 + po];
            }
        dt3[i] = s * gelu_grad_f(t3_save[(long long)b * 3 * D + i]);
    }
    __syncthreads();
    float* out = dpr + (long long)b * pr_stride;
    constexpr int NB = 5;                                             // outputs per thread whose old values are requested together
    for (int i0 = threadIdx.x; i0 < L * D; i0 += 256 * NB) {
        float old[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) old[u] = i0 + 256 * u < L * D ? out[i0 + 256 * u] : 0.0f;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = i0 + 256 * u;
            if (i >= L * D) break;
            const int l = i / D, d = i - l * D;
            float s = 0.0f;
            for (int p = 0; p < 3; ++p) {
                const float* wr = w0 + (p * L + l) * 3;
                if (d + 1 < D) s += wr[0] * dt3[p * D + d + 1];      // tap 0 reads x1[d'-1] for output d' = d+1
                s += wr[1] * dt3[p * D + d];
                if (d > 0) s += wr[2] * dt3[p * D + d - 1];
            }
            out[i] = old[u] + s;
        }
    }
}

// ---- FourierLayer (transformer.py:52-97).  spec[b][k][d], spec[b][NFp + k][d] = Re, Im of rfft bin k+1 (from dts_seqmix with
// the DFT basis).  Per (b, d): top-K amplitudes (torch.topk of |X|, largest, sorted), their amplitude and phase.
constexpr int DTS_MAX_TOPK = 8;
__global__ void dts_fourier_select_kernel(const float* __restrict__ spec, int NF, int NFp, int D, int topk, long long n, int* idx /*[B][topk][D]*/,
                                          float* amp, float* phase) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // b*D + d
    if (i >= n) return;
    const int d = (int)(i % D);
    const long long b = i / D;
    const float* re = spec + b * 2 * NFp * D + d;
    const float* im = re + (long long)NFp * D;
    float best[DTS_MAX_TOPK];
    int bi[DTS_MAX_TOPK];
#pragma unroll
    for (int j = 0; j < DTS_MAX_TOPK; ++j) { best[j] = -1.0f; bi[j] = 0; }
    constexpr int CH = 12;                                             // bins fetched per batch: the loads go out together, then the serial insertion runs
    for (int k0 = 0; k0 < NF; k0 += CH) {
        float rr[CH], ii[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int k = min(k0 + u, NF - 1);
            rr[u] = re[(long long)k * D];
            ii[u] = im[(long long)k * D];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int k = k0 + u;
            if (k >= NF) break;
            const float a = hypotf(rr[u], ii[u]);
            if (!(a > best[topk - 1])) continue;                       // strict: an earlier bin wins a tie
            // insertion with compile-time register indices only (a run-time index would put best[] / bi[] in scratch memory)
            int j = 0;
#pragma unroll
            for (int q = 0; q < DTS_MAX_TOPK; ++q) j += (q < topk && best[q] >= a) ? 1 : 0;
#pragma unroll
            for (int q = DTS_MAX_TOPK - 1; q > 0; --q)
                if (q < topk && q > j) { best[q] = best[q - 1]; bi[q] = bi[q - 1]; }
#pragma unroll
            for (int q = 0; q < DTS_MAX_TOPK; ++q)
                if (q == j) { best[q] = a; bi[q] = k; }
        }
    }
#pragma unroll
    for (int j = 0; j < DTS_MAX_TOPK; ++j) {
        if (j >= topk) break;
        const long long o = (b * topk + j) * D + d;
        idx[o] = bi[j];
        amp[o] = best[j];
        phase[o] = atan2f(im[(long long)bi[j] * D], re[(long long)bi[j] * D]);
    }
}

// season[b][t][d] += sum_j amp cos(2 pi f_j t + phase_j) over the selected bins AND their conjugates, with the reference's
// fp32 operation order for the argument (transformer.py:84-86): ((2 pi) * f) * t + phase, then cos, then amp * cos.
// cos(x) for |x| < 2^13 (here: 2 pi f t + phase, at most pi (L + 1)): three-term Cody-Waite reduction by pi/2 in fused multiply-adds, then the
// minimax polynomials of sin and cos on [-pi/4, pi/4] (Cephes sinf / cosf coefficients, < 2 ulp), selected by the quadrant -- a third of
// the instructions of the library cosf, whose generic argument reduction the synthesis kernel spent its time in (336 us per launch at
// 3200 x 200 x 64 for 1.6e8 cosines against 164 MB read and written).
__device__ __forceinline__ float cos_medium(float x) {
    const float k = rintf(x * 0.636619772367581343f);                    // 2 / pi
    float r = __builtin_fmaf(-k, 1.57079601287841796875f, x);
    r = __builtin_fmaf(-k, 3.1391647326017846353352069854736328125e-7f, r);
    r = __builtin_fmaf(-k, 5.390302529957764765544681040410068817436695098876953125e-15f, r);
    const int q = (int)k;
    const float z = r * r;
    const float c = __builtin_fmaf(z * z, __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                                   __builtin_fmaf(-0.5f, z, 1.0f));
    const float sn = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float v = (q & 1) ? sn : c;                                     // quadrant 0: cos r, 1: -sin r, 2: -cos r, 3: sin r
    return ((q + 1) & 2) ? -v : v;
}

// block per sequence: thread (channel d, quarter) keeps the selected bins of its channel -- angular frequency, phase, amplitude -- in registers
// and runs over the time steps t = quarter, quarter + 4, ...; the argument is formed exactly as the reference forms it (fp32, (2 pi f) t + phase).
__global__ __launch_bounds__(256) void dts_fourier_synth_kernel(const int* __restrict__ idx, const float* __restrict__ amp, const float* __restrict__ phase,
                                                                const float* __restrict__ freq /*rfftfreq[1 + k]*/, int L, int D, int topk, float* season) {
#pragma clang fp contract(off)
    constexpr int MAXK = 8;
    const int b = blockIdx.x, dl = threadIdx.x & 63, qu = threadIdx.x >> 6;
    const float two_pi = 6.283185307179586f;
    // the conjugate half has f -> -f, phase -> -phase: the same cosines again (rounding is symmetric), so the sum over the 2*topk
    // terms is twice the sum over the selected bins (to within the order of the additions)
    for (int d = dl; d < D; d += 64) {
        for (int j0 = 0; j0 < topk; j0 += MAXK) {               // (one pass for topk <= 8; more bins: one more pass over the sequence per eight)
            float w[MAXK], ph[MAXK], am[MAXK];
#pragma unroll
            for (int j = 0; j < MAXK; ++j) {
                const long long o = ((long long)b * topk + min(j0 + j, topk - 1)) * D + d;
                w[j] = two_pi * freq[idx[o]];
                ph[j] = phase[o];
                am[j] = j0 + j < topk ? amp[o] : 0.0f;
            }
            for (int t = qu; t < L; t += 4) {
                const float tf = (float)t;
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < MAXK; ++j) {
                    if (j0 + j >= topk) break;
                    s += am[j] * cos_medium(w[j] * tf + ph[j]);
                }
                season[((long long)b * L + t) * D + d] += 2.0f * s;
            }
        }
    }
}

// exact-arithmetic transpose of the layer: out[t] = 2 sum_j (Re_j cos th_j t - Im_j sin th_j t) is linear in x through the
// selected bins, so d_x[t'] = 2 sum_j (C_j cos th_j t' + S_j sin th_j t'),  C_j = sum_t g[t] cos th_j t,  S_j = sum_t g[t] sin th_j t.
// block per sequence; basis tables cosT/sinT[m] = cos/sin(2 pi m / L), m = (k t) mod L.
__global__ __launch_bounds__(256) void dts_fourier_bwd_kernel(const float* __restrict__ g /*[B][L][D]*/, const int* __restrict__ idx,
                                                              const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                              const float* __restrict__ wdft_t /*[L][2 NFp]*/, int L, int D, int topk, int NFp, float* dx,
                                                              long long dx_stride) {
    extern __shared__ float sm[];                 // C[topk][D], S[topk][D], then the basis tables cos / sin [L] and the partial sums [4][2][topk][D]
    float* Cs = sm;
    float* Ss = sm + topk * D;
    float* ct = Ss + topk * D;
    float* st = ct + L;
    float* part = st + L;
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < L; i += 256) {
        ct[i] = cosT[i];
        st[i] = sinT[i];
    }
    __syncthreads();
    // C_j[d], S_j[d]: thread (d, quarter) runs over the time steps t = quarter, quarter + 4, ... for every selected bin of channel d (g is read
    // once for all bins), the four partial sums of a (j, d) are added in quarter order.  D <= 64 channels per pass of 256 threads.
    const int dl = threadIdx.x & 63, qu = threadIdx.x >> 6;
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + dl;
        if (d < D) {
            for (int j0 = 0; j0 < topk; j0 += 8) {
                const int nj = min(8, topk - j0);
                int kk[8], m[8];
                float c[8], sa[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    kk[j] = j < nj ? idx[((long long)b * topk + j0 + j) * D + d] + 1 : 0;      // rfft bin number
                    m[j] = (int)(((long long)kk[j] * qu) % L);
                    kk[j] = (int)(((long long)kk[j] * 4) % L);
                    c[j] = sa[j] = 0.0f;
                }
                for (int t = qu; t < L; t += 4) {
                    const float gv = g[((long long)b * L + t) * D + d];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        c[j] += gv * ct[m[j]];
                        sa[j] += gv * st[m[j]];
                        m[j] += kk[j];
                        if (m[j] >= L) m[j] -= L;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < nj) {
                        part[((qu * 2 + 0) * topk + j0 + j) * D + d] = c[j];
                        part[((qu * 2 + 1) * topk + j0 + j) * D + d] = sa[j];
                    }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < topk * D; i += 256) {
        Cs[i] = ((part[(0 * 2 + 0) * topk * D + i] + part[(1 * 2 + 0) * topk * D + i]) + part[(2 * 2 + 0) * topk * D + i]) + part[(3 * 2 + 0) * topk * D + i];
        Ss[i] = ((part[(0 * 2 + 1) * topk * D + i] + part[(1 * 2 + 1) * topk * D + i]) + part[(2 * 2 + 1) * topk * D + i]) + part[(3 * 2 + 1) * topk * D + i];
    }
    __syncthreads();
    // d_x2part[t'] = 2 sum_j (C_j cos th_j t' + S_j sin th_j t') pulled back through proj[L:]: with Mre = cos-basis x proj, Mim = -sin-basis x proj
    // (rows kj and NFp + kj of the composite), d_x2[l'] += 2 sum_j (C_j Mre[kj][l'] - S_j Mim[kj][l'])
    float* out = dx + (long long)b * dx_stride;
    for (int i = threadIdx.x; i < L * D; i += 256) {
        const int l = i / D, d = i - l * D;
        const float* wr = wdft_t + (long long)l * 2 * NFp;
        float a = 0.0f;
        for (int j = 0; j < topk; ++j) {
            const int kj = idx[((long long)b * topk + j) * D + d];
            a += Cs[j * D + d] * wr[kj] - Ss[j * D + d] * wr[NFp + kj];
        }
        out[i] += 2.0f * a;
    }
}

// ---- output tail (transformer.py:430-438): res = Conv1d(D -> F, k=3, pad=1)(h); res_m = mean_l res;
//   season_error = combine_s(season) [circular, `taps` wide, no bias] + res - res_m;  trend = combine_m(mean) + res_m + trend_sum
// block per sequence.
__global__ __launch_bounds__(256) void dts_tail_fwd_kernel(const float* __restrict__ h, const float* __restrict__ season, const float* __restrict__ trend_sum,
                                                           const float* __restrict__ means /*[B][n_dec][F]*/, const float* __restrict__ winv /*[F][D][3]*/,
                                                           const float* __restrict__ binv, const float* __restrict__ wcs /*[F][D][taps]*/,
                                                           const float* __restrict__ wcm /*[n_dec]*/, int L, int F, int D, int taps, int n_dec,
                                                           float* trend_out, float* season_out) {
    extern __shared__ float sm[];                 // res[L][F], cs[L][F], resm[F], cm[F]
    float* res = sm;
    float* cs = sm + L * F;
    float* resm = cs + L * F;
    float* cm = resm + F;
    const int b = blockIdx.x, pad = (taps - 1) / 2;
    const float* hb = h + (long long)b * L * D;
    const float* sb = season + (long long)b * L * D;
    const int li = threadIdx.x & 15, grp = threadIdx.x >> 4;       // 16 lanes per output (l, f), channels strided over the lanes
    for (int i = grp; i < L * F; i += 16) {
        const int l = i / F, f = i - l * F;
        float a = 0.0f, c = 0.0f;
        for (int tap = 0; tap < 3; ++tap) {
            const int ll = l + tap - 1;
            if (ll < 0 || ll >= L) continue;
            for (int d = li; d < D; d += 16) a += winv[(f * D + d) * 3 + tap] * hb[ll * D + d];
        }
        for (int tap = 0; tap < taps; ++tap) {
            int ll = l + tap - pad;
            ll = ll < 0 ? ll + L : (ll >= L ? ll - L : ll);
            for (int d = li; d < D; d += 16) c += wcs[(f * D + d) * taps + tap] * sb[ll * D + d];
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            a += __shfl_xor(a, m);
            c += __shfl_xor(c, m);
        }
        if (li == 0) {
            res[i] = a + binv[f];
            cs[i] = c;
        }
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256) {
        float s = 0.0f;
        for (int l = 0; l < L; ++l) s += res[l * F + f];
        resm[f] = s / (float)L;
        float m = 0.0f;
        for (int i = 0; i < n_dec; ++i) m += wcm[i] * means[((long long)b * n_dec + i) * F + f];
        cm[f] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * F; i += 256) {
        const int f = i % F;
        season_out[(long long)b * L * F + i] = (cs[i] + res[i]) - resm[f];
        trend_out[(long long)b * L * F + i] = (cm[f] + resm[f]) + trend_sum[(long long)b * L * F + i];
    }
}

// g = d/d(trend + season_error) [B][L][F] -> d_h [B][L][D], d_season [B][L][D], d_means [B][n_dec][F]   (res_m cancels)
__global__ __launch_bounds__(256) void dts_tail_bwd_kernel(const float* __restrict__ g, const float* __restrict__ winv, const float* __restrict__ wcs,
                                                           const float* __restrict__ wcm, int L, int F, int D, int taps, int n_dec, float* dh,
                                                           float* dseason, float* dmeans) {
    const int b = blockIdx.x, pad = (taps - 1) / 2;
    const float* gb = g + (long long)b * L * F;
    for (int i = threadIdx.x; i < L * D; i += 256) {
        const int l = i / D, d = i - l * D;
        float a = 0.0f, c = 0.0f;
        for (int tap = 0; tap < 3; ++tap) {
            const int lo = l - tap + 1;                               // output token whose tap reads h[l]
            if (lo < 0 || lo >= L) continue;
            for (int f = 0; f < F; ++f) a += winv[(f * D + d) * 3 + tap] * gb[lo * F + f];
        }
        for (int tap = 0; tap < taps; ++tap) {
            int lo = l - tap + pad;
            lo = lo < 0 ? lo + L : (lo >= L ? lo - L : lo);
            for (int f = 0; f < F; ++f) c += wcs[(f * D + d) * taps + tap] * gb[lo * F + f];
        }
        dh[(long long)b * L * D + i] = a;
        dseason[(long long)b * L * D + i] = c;
    }
    for (int i = threadIdx.x; i < n_dec * F; i += 256) {
        const int f = i % F, k = i / F;
        float s = 0.0f;
        for (int l = 0; l < L; ++l) s += gb[l * F + f];
        dmeans[(long long)b * n_dec * F + i] = wcm[k] * s;
    }
}

// ---- sampler steps (DiffusionTS.py:277-310, 359-407), elementwise over [B][L][F]; mask = token < windows -------------------
struct PairCoef { float a, bq, sa, c, sqrt_ac, sqrt_1mac; };

// x_start = clamp(trend + season, -1, 1); pred_noise = (a x - x_start) / b; pred_mean = x_start sqrt(alpha_next) + c pred_noise
__global__ void dts_predict_kernel(const float* __restrict__ img, const float* __restrict__ trend, const float* __restrict__ season, PairCoef k,
                                   int last, long long n, float* x_start, float* pred_mean) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float xs = trend[i] + season[i];
    xs = fminf(fmaxf(xs, -1.0f), 1.0f);
    x_start[i] = xs;
    if (last) return;
    const float pn = (k.a * img[i] - xs) / k.bq;
    pred_mean[i] = xs * k.sa + k.c * pn;
}

// img = pred_mean + sigma * noise   (the stochastic DDIM step, eta != 0: DiffusionTS.py:300)
__global__ void dts_add_noise_kernel(const float* __restrict__ pmean, const float* __restrict__ noise, float sigma, long long n, float* img) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    img[i] = pmean[i] + sigma * noise[i];
}

// g_out = d(infill)/d(x_start): 2 (x_start - target) / N_masked on the observed part (x_start NOT clamped here, :389)
__global__ void dts_infill_grad_kernel(const float* __restrict__ trend, const float* __restrict__ season, const float* __restrict__ target, int L, int F,
                                       int windows, float inv_nmask, long long n, float* gout) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = (int)((i / F) % L);
    gout[i] = l < windows ? (2.0f * ((trend[i] + season[i]) - target[i])) * inv_nmask : 0.0f;
}

// one Adagrad-from-scratch step on the whole tensor (torch.optim.Adagrad, fresh state, :384-397):
//   g = g_x + (coef / B) (2 (p - mean));   state_sum = g g;   p += ((-lr) g) / (sqrt(state_sum) + 1e-10)
__global__ void dts_langevin_step_kernel(const float* __restrict__ gx, const float* __restrict__ mean, float coef_over_b, float lr, long long n, float* p) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float pv = p[i];
    const float g = gx[i] + coef_over_b * (2.0f * (pv - mean[i]));
    const float std = sqrtf(g * g) + 1e-10f;
    p[i] = pv + ((-lr) * g) / std;
}

// img = observed ? sqrt_ac * target + sqrt_1mac * noise : param      (q_sample overwrite, :305-306)
__global__ void dts_overwrite_kernel(const float* __restrict__ param, const float* __restrict__ target, const float* __restrict__ noise, PairCoef k, int L,
                                     int F, int windows, long long n, float* img) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = (int)((i / F) % L);
    img[i] = l < windows ? k.sqrt_ac * target[i] + k.sqrt_1mac * noise[i] : param[i];
}

// rows of the batch: target[r][l][f] = l < windows ? history[w][v][l][f] : 0, with r = (w*par + s)*V + v (sample-major tiling of
// DiffusionTS_model.py:92); n = rows * L * F
__global__ void dts_build_target_kernel(const float* __restrict__ hist /*[W][V][windows][F]*/, int V, int par, int L, int F, int windows, long long row0,
                                        long long n, float* target) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = (int)(i % F);
    const int l = (int)((i / F) % L);
    const long long r = row0 + i / ((long long)F * L);
    const int v = (int)(r % V);
    const long long w = r / ((long long)V * par);
    target[i] = l < windows ? hist[((w * V + v) * windows + l) * F + f] : 0.0f;
}

// final img (observed part = target) -> samples[w][v'][o][f][chunk*par + p'] with the reference's un-tiling quirk
// (DiffusionTS_model.py:102-104): batch row r of a window is read as (v', p') = (r / par, r % par) although it was tiled
// sample-major (r = s*V + v).
__global__ void dts_gather_kernel(const float* __restrict__ x_start, int V, int par, int L, int F, int windows, int S, int chunk, long long row0,
                                  long long n, float* samples /*[W][V][O][F][S]*/) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // ((row - row0)*O + o)*F + f, row = w*V*par + r
    if (i >= n) return;
    const int O = L - windows;
    const int f = (int)(i % F);
    const int o = (int)((i / F) % O);
    const long long lr = i / ((long long)F * O);
    const long long wr = row0 + lr;
    const int r = (int)(wr % ((long long)V * par));
    const long long w = wr / ((long long)V * par);
    const int vq = r / par, pq = r - vq * par;
    samples[((((w * V + vq) * O + o) * F + f) * (long long)S) + chunk * par + pq] = x_start[((lr * L) + windows + o) * F + f];
}

// Philox normals for a [n] tensor: element i uses normal (i & 3) of block i >> 2, stream `stream`
__global__ void dts_philox_kernel(unsigned seed_lo, unsigned seed_hi, unsigned stream, long long q0, long long n, float* out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // block of 4 (q0 + q: position in the whole tensor)
    if (4 * q >= n) return;
    float z[4];
    chain::philox_normals((unsigned long long)(q0 + q), stream, seed_lo, seed_hi, 0x44545300u, z);
    for (int e = 0; e < 4 && 4 * q + e < n; ++e) out[4 * q + e] = z[e];
}

}  // namespace dts
