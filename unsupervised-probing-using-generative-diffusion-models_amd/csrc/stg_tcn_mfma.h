// DiffSTG / NsDiff_spatial ResidualBlock front half on the f16 matrix pipe (reference DiffSTG/ugnet.py:55-76, 115-126): tcn1 (causal k = 3
// convolution + 1x1 skip) + time-embedding projection, tcn2 (+ identity skip), LayerNorm over the channels, and the block's 1x1 shortcut
// when it is a convolution.  Included by stg.hip.
//
// Both convolutions are GEMMs  D[co][t] = sum_k A[co][k] B[k][t]  with k = (tap, input channel) and B[k][t] = x[channel][t + tap - 2]: M = co <= 16
// rows, N = 16 time steps per tile, K = 3 ci' (ci' = ci rounded up to 8) in chunks of 32 on v_mfma_f32_16x16x32_f16.  fp32-equivalent accuracy by
// the two-term f16 split of nst_gemm.h: x = hi + lo (hi = 11 leading bits, exact in f16), three products hi*hi + hi*lo + lo*hi, fp32 accumulate.
//   * The 1x1 skips ride in the current-time tap of the convolution weights (W1[.][.][2] += tcn1.shortcut or I, W2[.][.][2] += I), folded at load.
//   * A operands (weights, <= 48 VGPRs: conv1, conv2, block shortcut; hi and lo) stay in registers for the life of the workgroup.
//   * B operands come from an LDS image of the row segment TRANSPOSED to [time][channel] f16 (hi and lo planes), so the 8 consecutive k of a lane
//     are 8 channels at one time step = one ds_read_b128; the causal shift is an address offset.
//   * h1 leaves the accumulators (column = time on the lane, rows = 4 channels per lane group) as 8-byte LDS writes into the same transposed
//     layout; LayerNorm over the channels of a time step = 4 registers + two cross-group shuffles.
// The VALU version this replaces (stg_tcn_ln_kernel) ran at 17 TFLOP/s, bound by the LDS reads that fed its FMAs (profiles/r03).
// Activations are split UNSCALED: they are LayerNorm outputs, their convolutions and block sums, O(1), and every product sums to an O(1)
// result, so the reduced precision of the (subnormal) low half of elements below 2^-2 is an absolute error of 6e-8 per term.  Pre-scaling them
// by 2^6 was measured: same parity figures (1.5e-5 .. 2.8e-5 at the C5 dimensions), no gain.
#pragma once

namespace {

typedef _Float16 th8 __attribute__((ext_vector_type(8)));
typedef _Float16 th4 __attribute__((ext_vector_type(4)));
typedef float tf4 __attribute__((ext_vector_type(4)));

struct TcnMArgs {
    const float* src1; int c1;      // [N][c1][T]
    const float* src2; int c2;      // concatenated skip [N][c2][T] | null
    const th8* a1;                  // conv1 (+ skip) A fragments: [NC1][2 (hi, lo)][64 lanes] of 8 halves, scaled by 2^s1
    const th8* a2;                  // conv2 (+ identity) A fragments: [NC2][2][64]
    const th8* asb;                 // block shortcut A fragments: [1][2][64] | null
    float inv1, inv2, invsb;        // 2^-s
    const float *b1s;               // [co] tcn1.conv.bias (+ tcn1.shortcut.bias)
    const float *wt, *bt;           // t_conv [co][d_h], [co]
    const float* temb;              // [d_h]
    const float *b2;                // [co]
    const float *lnw, *lnb;         // [co]
    const float* bsb;               // [co] block shortcut bias | null
    float* out;                     // [N][co][T]
    float* sc;                      // [N][co][T] block shortcut output | null
    int co, T, d_h;
    int cip1, cip2;                 // padded input channels of conv1 / conv2 (multiples of 8)
    long long rows;
    int tseg, nseg;                 // time steps per item (multiple of 16 unless one segment covers the row), items per row
};

__device__ __forceinline__ void tcn_split4(const float (&v)[4], th4& hi, th4& lo) {
    float h[4], l[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        h[c] = __uint_as_float(__float_as_uint(v[c]) & 0xFFFFE000u);
        l[c] = v[c] - h[c];
    }
    typedef __fp16 pk2 __attribute__((ext_vector_type(2)));
    const pk2 h01 = __builtin_amdgcn_cvt_pkrtz(h[0], h[1]), h23 = __builtin_amdgcn_cvt_pkrtz(h[2], h[3]);
    const pk2 l01 = __builtin_amdgcn_cvt_pkrtz(l[0], l[1]), l23 = __builtin_amdgcn_cvt_pkrtz(l[2], l[3]);
    hi = th4{(_Float16)h01[0], (_Float16)h01[1], (_Float16)h23[0], (_Float16)h23[1]};
    lo = th4{(_Float16)l01[0], (_Float16)l01[1], (_Float16)l23[0], (_Float16)l23[1]};
}

#ifndef TCN_BATCH
#define TCN_BATCH 4                 // staging items (4 channels x 1 step) whose global loads a thread issues back to back
#endif

#ifndef TCN_PROBE
#define TCN_PROBE 0                 // timing probes (tools/r4_tcn_probe.sh only; results are wrong): 1 no global stores, 2 no global loads
#endif
#define TCN_MMA(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (acc), 0, 0, 0)

// NC1 / NC2: 32-wide k chunks of conv1 / conv2 (ceil(3 cip / 32)); SB: the block shortcut is a 1x1 convolution
template <int NC1, int NC2, bool SB>
__global__ __launch_bounds__(256) void stg_tcn_mfma_kernel(const TcnMArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
    const int ci = p.c1 + p.c2, co = p.co, T = p.T;
    const int P1 = p.cip1 + 8, P2 = p.cip2 + 8;                       // halves per time step of the two LDS images (16-byte multiples)
    const int nt1 = (p.tseg + 2 + 15) >> 4;                            // 16-step tiles of h1 on [t0 - 2, t1)
    const int XR = 16 * nt1 + 2, HR = 16 * nt1 + 16;                   // rows (time steps) of the x / h1 images
    _Float16* xh = reinterpret_cast<_Float16*>(smraw);                 // [XR][P1]   x at time t0 - 4 + row
    _Float16* xl = xh + (size_t)XR * P1;
    _Float16* hh = xl + (size_t)XR * P1;                               // [HR][P2]   h1 at time t0 - 2 + row
    _Float16* hl = hh + (size_t)HR * P2;
    float* tvs = reinterpret_cast<float*>(hl + (size_t)HR * P2);       // [16] bias of h1: tcn1 biases + time-embedding projection
    // ---- once per workgroup: A fragments into registers, the h1 bias vector ----
    th8 A1[NC1][2], A2[NC2][2], ASB[2];
#pragma unroll
    for (int c = 0; c < NC1; ++c) { A1[c][0] = p.a1[(c * 2 + 0) * 64 + lane]; A1[c][1] = p.a1[(c * 2 + 1) * 64 + lane]; }
#pragma unroll
    for (int c = 0; c < NC2; ++c) { A2[c][0] = p.a2[(c * 2 + 0) * 64 + lane]; A2[c][1] = p.a2[(c * 2 + 1) * 64 + lane]; }
    if (SB) { ASB[0] = p.asb[lane]; ASB[1] = p.asb[64 + lane]; }
    if (tid < 16) {
        float a = 0.0f;
        if (tid < co) {
            a = p.bt[tid];
            for (int d = 0; d < p.d_h; ++d) a += p.wt[tid * p.d_h + d] * p.temb[d];
            a += p.b1s[tid];
        }
        tvs[tid] = a;
    }
    // per-lane constants of the B-operand addressing: k-group q = 4 chunk + g -> (tap, first channel)
    const int G1 = p.cip1 >> 3, G2 = p.cip2 >> 3;
    int off1[NC1], off2[NC2];
#pragma unroll
    for (int c = 0; c < NC1; ++c) {
        int q = 4 * c + g;
        if (q >= 3 * G1) q = 3 * G1 - 1;                              // padding k-groups: their A columns are zero, any valid address will do
        off1[c] = (q / G1) * P1 + 8 * (q % G1);
    }
#pragma unroll
    for (int c = 0; c < NC2; ++c) {
        int q = 4 * c + g;
        if (q >= 3 * G2) q = 3 * G2 - 1;
        off2[c] = (q / G2) * P2 + 8 * (q % G2);
    }
    const int offsb = 2 * P1 + 8 * (g < G1 ? g : G1 - 1);             // block shortcut: the current-time tap only (K = cip1 <= 32: one chunk)
    const int cq = p.cip1 >> 2;                                       // channel quads of the x image
    const unsigned xr_magic = 0xFFFFFFFFu / (unsigned)XR + 1u;        // i / XR = umulhi(i, xr_magic) for the i < 2^16 this kernel sees
    float lw[4], lb[4], b2v[4], sbv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * g + r;
        lw[r] = c < co ? p.lnw[c] : 0.0f;
        lb[r] = c < co ? p.lnb[c] : 0.0f;
        b2v[r] = c < co ? p.b2[c] : 0.0f;
        sbv[r] = (SB && c < co) ? p.bsb[c] : 0.0f;
    }
    const float inv_co = 1.0f / (float)co;

    for (long long item = blockIdx.x; item < p.rows * p.nseg; item += gridDim.x) {
        const long long n = item / p.nseg;
        const int t0 = (int)(item - n * p.nseg) * p.tseg, t1 = min(T, t0 + p.tseg), tl = t1 - t0;
        __syncthreads();                                              // (the previous item is done with the images; first item: tvs is written)
        // ---- stage x[t0 - 4 .. ) transposed: thread -> (channel quad, time step); zeros before the row starts, after t1 and in padded channels ----
        // Round 4 (late): the 16 loads of a thread's batch are UNCONDITIONAL on clamped addresses and the zeroing is a select where the value is
        // split (a guarded load is an exec-mask branch of its own with a full wait behind it: 44 branches and ~500 vector instructions per batch
        // before), item -> (quad, step) by a multiply-high instead of a division, a quad's four channels one base + stride T (the first source's
        // channel count is a multiple of 4 in every UGnet this engine builds; otherwise, and in the probe builds, the previous loop below).
        if (!TCN_PROBE && (p.c1 & 3) == 0) {
            const int nitems = cq * XR;
            const float* rb1 = p.src1 + (size_t)n * p.c1 * T;          // (uniform)
            const float* rb2 = p.c2 ? p.src2 + (size_t)n * p.c2 * T : rb1;
            const int wf = __builtin_amdgcn_readfirstlane(tid & ~63);   // this wave's first thread: a batch slot none of its lanes fills is skipped (uniform)
            for (int i0 = tid; i0 < nitems; i0 += 256 * TCN_BATCH) {
                float v[TCN_BATCH][4];
                int keep[TCN_BATCH], wo[TCN_BATCH];
#pragma unroll
                for (int u = 0; u < TCN_BATCH; ++u) {
                    keep[u] = 0; wo[u] = -1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = 0.0f;
                    if (wf + (i0 - tid) + 256 * u >= nitems) continue;
                    const int i = min(i0 + 256 * u, nitems - 1);
                    const int q = (int)__umulhi((unsigned)i, xr_magic), row = i - q * XR, t = t0 - 4 + row;
                    const int c0 = min(4 * q, ci - 1);                   // (a padded quad reads the last real channel and keeps nothing)
                    const bool from1 = c0 < p.c1;
                    const float* base = from1 ? rb1 + (size_t)c0 * T : rb2 + (size_t)(c0 - p.c1) * T;
                    const int lim = (from1 ? p.c1 : ci) - 1 - c0;        // channels of this source behind c0
                    const int tc = min(max(t, 0), T - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = base[(size_t)min(j, lim) * T + tc];
                    keep[u] = (i0 + 256 * u < nitems && t >= 0 && t < t1) ? min(ci - 4 * q, 4) : 0;      // leading channels of the quad that are real
                    wo[u] = i0 + 256 * u < nitems ? row * P1 + 4 * q : -1;
                }
#pragma unroll
                for (int u = 0; u < TCN_BATCH; ++u) {
                    if (wf + (i0 - tid) + 256 * u >= nitems) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = j < keep[u] ? v[u][j] : 0.0f;
                    th4 hi, lo;
                    tcn_split4(v[u], hi, lo);
                    if (wo[u] >= 0) {
                        *reinterpret_cast<th4*>(xh + wo[u]) = hi;
                        *reinterpret_cast<th4*>(xl + wo[u]) = lo;
                    }
                }
            }
        } else
        for (int i0 = tid; i0 < cq * XR; i0 += 256 * TCN_BATCH) {    // TCN_BATCH thread items' loads are issued before the first LDS write
            float v[TCN_BATCH][4];
#pragma unroll
            for (int u = 0; u < TCN_BATCH; ++u) {
                const int i = i0 + 256 * u;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[u][j] = 0.0f;
                if (i < cq * XR) {
                    const int q = i / XR, row = i - q * XR, t = t0 - 4 + row;
                    if (t >= 0 && t < t1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = 4 * q + j;
                            if (TCN_PROBE & 2) v[u][j] = 0.25f * (float)(c + t);
                            else if (c < p.c1) v[u][j] = p.src1[(n * p.c1 + c) * T + t];
                            else if (c < ci) v[u][j] = p.src2[(n * p.c2 + (c - p.c1)) * T + t];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < TCN_BATCH; ++u) {
                const int i = i0 + 256 * u;
                if (i < cq * XR) {
                    const int q = i / XR, row = i - q * XR;
                    th4 hi, lo;
                    tcn_split4(v[u], hi, lo);
                    *reinterpret_cast<th4*>(xh + row * P1 + 4 * q) = hi;
                    *reinterpret_cast<th4*>(xl + row * P1 + 4 * q) = lo;
                }
            }
        }
        __syncthreads();
        // ---- h1 on [t0 - 2, t1): tile i covers times t0 - 2 + 16 i + n16; x rows 16 i + n16 + tap ----
        for (int i = wave; i < nt1; i += 4) {
            const int xrow = 16 * i + n16;
            tf4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < NC1; ++c) {
                const th8 bh = *reinterpret_cast<const th8*>(xh + xrow * P1 + off1[c]);
                const th8 bl = *reinterpret_cast<const th8*>(xl + xrow * P1 + off1[c]);
                TCN_MMA(acc, A1[c][1], bh);
                TCN_MMA(acc, A1[c][0], bl);
                TCN_MMA(acc, A1[c][0], bh);
            }
            const int t = t0 - 2 + 16 * i + n16;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (t >= 0 && 4 * g + r < co) ? acc[r] * p.inv1 + tvs[4 * g + r] : 0.0f;   // h1 is ZERO before the row starts (tcn2's padding)
            th4 hi, lo;
            tcn_split4(v, hi, lo);
            if (4 * g < p.cip2) {
                *reinterpret_cast<th4*>(hh + (16 * i + n16) * P2 + 4 * g) = hi;
                *reinterpret_cast<th4*>(hl + (16 * i + n16) * P2 + 4 * g) = lo;
            }
            if (SB) {                                                 // block shortcut of the current time step (x row + 2), times [t0, t1) only
                tf4 sb = {0.0f, 0.0f, 0.0f, 0.0f};
                const th8 bh = *reinterpret_cast<const th8*>(xh + xrow * P1 + offsb);
                const th8 bl = *reinterpret_cast<const th8*>(xl + xrow * P1 + offsb);
                TCN_MMA(sb, ASB[1], bh);
                TCN_MMA(sb, ASB[0], bl);
                TCN_MMA(sb, ASB[0], bh);
                if (t >= t0 && t < t1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < co && (!(TCN_PROBE & 1) || sb[r] == 1.2345e30f)) p.sc[(n * co + 4 * g + r) * T + t] = sb[r] * p.invsb + sbv[r];
                }
            }
        }
        __syncthreads();
        // ---- output on [t0, t1): tile j covers times t0 + 16 j + n16; h1 rows 16 j + n16 + tap ----
        const int nt2 = (tl + 15) >> 4;
        for (int j = wave; j < nt2; j += 4) {
            const int hrow = 16 * j + n16;
            tf4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < NC2; ++c) {
                const th8 bh = *reinterpret_cast<const th8*>(hh + hrow * P2 + off2[c]);
                const th8 bl = *reinterpret_cast<const th8*>(hl + hrow * P2 + off2[c]);
                TCN_MMA(acc, A2[c][1], bh);
                TCN_MMA(acc, A2[c][0], bl);
                TCN_MMA(acc, A2[c][0], bh);
            }
            float v[4], s = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = (4 * g + r < co) ? acc[r] * p.inv2 + b2v[r] : 0.0f;
                s += v[r];
            }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mean = s * inv_co;                            // LayerNorm([1, co]) over the channels of this time step
            float qq = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = (4 * g + r < co) ? v[r] - mean : 0.0f;
                qq += v[r] * v[r];
            }
            qq += __shfl_xor(qq, 16);
            qq += __shfl_xor(qq, 32);
            const float rstd = 1.0f / sqrtf(qq * inv_co + 1e-5f);
            const int t = t0 + 16 * j + n16;
            if (t < t1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < co && (!(TCN_PROBE & 1) || v[r] == 1.2345e30f)) p.out[(n * co + 4 * g + r) * T + t] = v[r] * rstd * lw[r] + lb[r];
            }
        }
    }
}

}  // namespace
