// Shared device helpers of the NsDiff chain kernels (fp32-MFMA and f16x3-MFMA variants).
#pragma once
#include "mpv_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace chain {

constexpr int HID = MPV_HIDDEN;
constexpr int WROW = 132;  // padded LDS row stride in floats: (33*i) mod 16 distinct over 16 lanes
#ifndef MPV_WAVES
#define MPV_WAVES 8
#endif
constexpr int WAVES = MPV_WAVES;
constexpr int THREADS = WAVES * 64;
constexpr int TILE = 32;   // chains per wave

// LDS carve (floats); every offset is a multiple of 4 floats (16 B)
constexpr int L_W2 = 0;
constexpr int L_W3 = L_W2 + HID * WROW;
constexpr int L_W1T = L_W3 + HID * WROW;              // [3F][128]
constexpr int L_B1 = L_W1T + 3 * MPV_MAX_NF * HID;
constexpr int L_B2 = L_B1 + HID;
constexpr int L_B3 = L_B2 + HID;
constexpr int L_W4 = L_B3 + HID;                      // [F][128]
constexpr int L_WS = L_W4 + MPV_MAX_NF * HID;         // [F][128]
constexpr int L_HB = L_WS + MPV_MAX_NF * HID;         // b4[4], bs[4]
constexpr int L_FLOATS = L_HB + 8;

struct ChainArgs {
    const float *w1t, *w2, *w3, *b1, *b2, *b3, *e1, *e2, *e3, *w4, *ws, *b4, *bs, *coef;
    // f16x3 variant: split weights [128][128] f16 (columns permuted for the A fragment), scaled biases,
    // folded per-step scale tables, sigma head pre-multiplied by ln 2
    const unsigned short *w2h, *w2l, *w3h, *w3l;
    const unsigned short* w1img;   // layer 1 as MFMA k-slots: [128][16*NK] f16 (nsdiff_chain_f16.hip::L1)
    const unsigned *b2p, *b3p;     // layer 2/3 biases, (hi | lo << 16) f16 pairs of b * 2^sB
    const float *g1, *g2, *g3, *wsl;
    const unsigned *gs2, *gs3;     // g2, g3 as bf16 outer-product A operands: [T*128 + 128 zero entries][4 words]
    const float* w4h;              // eps head as the split-f16 kernel stages it (x ln2 for TMDM, whose activations are not normalised)
    float bone2, bone3;            // bias k-slot constant of layers 2, 3: 2^(sW - sB)  (x the B-operand scale)
    float l1_scale;                // 2^(s1 + 4): scale of the layer-1 accumulators
    unsigned hi_mask;              // 0xFFFFE000: keeps the 11 leading significand bits of an fp32
    const float* y0hat;
    const float* gx;
    const float* noise;
    float* samples;
    long long n_chains;
    long long n_tiles;
    long long noise_kstride;   // B*P*O*F
    int T, B, O, S, P, C;
    unsigned seed_lo, seed_hi, off_lo;
    unsigned long long* tile_counter;   // f16 kernel: next unclaimed tile (waves pull tiles, so both waves of a SIMD stay busy to the end)
    // ---- fused summary epilogue of the f16 kernel (all zero / NULL = off) ----
    double* acc;                 // global [W][F][2] fp64: sum over rows of var_row, sum over rows of mean_row (zeroed by the launcher)
    const float *inv_std, *inv_mean;   // [F] inverse scaling inside the summary, or NULL
    int seg;                     // lanes of a tile that share a (b,o,f) row: min(S, 32), a power of two
    int sub;                     // consecutive tiles that make up one row: max(1, S / 32)
    int claim;                   // tiles per work-queue claim: a multiple of `sub` (8 by default)
    int acc_windows;             // windows of this launch (rows of the LDS accumulator table)
    long long chain_base;        // added to the chain index in the Philox key (launches over window groups)
    unsigned long long* stats;   // [0] += steps re-run through chain_step_safe
};

// ---- softplus(x) = x > 20 ? x : log1p(exp(x))  (torch.nn.functional.softplus defaults) ----
// Raw v_exp_f32 / v_log_f32 (no denormal pre-scaling): exp2 underflowing to 0 gives softplus 0
// where the true value is < 2^-126, and 1+e is always a normal number.  For e < 2^-12 the
// series e - e^2/2 replaces log(1+e), whose argument would have lost e's low bits.
__device__ __forceinline__ float softplus_dev(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
    float l = __builtin_amdgcn_logf(1.0f + e) * 0.693147180559945309f;
    const float small = __builtin_fmaf(-0.5f * e, e, e);
    l = (e < 2.44140625e-4f) ? small : l;
    return (x > 20.0f) ? x : l;
}

// ---- Philox4x32-10 + Box-Muller -------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void philox_normals(unsigned long long chain, unsigned k, unsigned seed_lo, unsigned seed_hi,
                                               unsigned off_lo, float (&z)[4]) {
    unsigned x[4];
    philox4x32_10((unsigned)chain, (unsigned)(chain >> 32), k, off_lo, seed_lo, seed_hi, x);
    const float inv24 = 5.9604644775390625e-8f;  // 2^-24
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float u1 = ((float)(x[2 * i] >> 8) + 0.5f) * inv24;      // (0,1)
        const float u2 = (float)(x[2 * i + 1] >> 8) * inv24;            // [0,1) revolutions
        const float rad = sqrtf(-2.0f * logf(u1));
        z[2 * i] = rad * __builtin_amdgcn_cosf(u2);
        z[2 * i + 1] = rad * __builtin_amdgcn_sinf(u2);
    }
}


// Normal draw k (0 = z_T, k = T - t for step t) of one chain, feature f = z[f].  One Philox block gives four
// normals; when F divides 4 they are used up over 4/F consecutive draws (element (k*F + f) & 3 of block
// (k*F) >> 2, kept in `cache` between steps), otherwise block k serves draw k.
template <int F>
__device__ __forceinline__ void step_normals(unsigned long long chain, unsigned k, unsigned seed_lo, unsigned seed_hi,
                                             unsigned off_lo, float (&cache)[4], float (&z)[4]) {
    if (F == 1 || F == 2) {
        const unsigned idx = k * F;
        if ((idx & 3u) == 0u) philox_normals(chain, idx >> 2, seed_lo, seed_hi, off_lo, cache);
        const unsigned e = idx & 3u;
        if (F == 1) {
            z[0] = e == 0u ? cache[0] : (e == 1u ? cache[1] : (e == 2u ? cache[2] : cache[3]));
        } else {
            z[0] = e ? cache[2] : cache[0];
            z[1] = e ? cache[3] : cache[1];
        }
    } else {
        philox_normals(chain, k, seed_lo, seed_hi, off_lo, z);
    }
}

// Next unclaimed 32-chain tile (wave-uniform).  Waves PULL tiles: the older wave of a SIMD wins issue arbitration and runs
// ~2x faster than its partner, so a static split leaves every SIMD with a single wave for the last third of a launch.
// Every wave's claims eventually land at or beyond n_tiles, so the grid always drains.
__device__ __forceinline__ long long claim_tile(unsigned long long* counter, int lane) {
    unsigned long long claim = 0;
    if (lane == 0) claim = atomicAdd(counter, 1ull);
    return (long long)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(claim >> 32)) << 32) |
                       __builtin_amdgcn_readfirstlane((unsigned)claim));
}

// hidden-unit index of register (4q+e) of out-tile j in half-wave h is 32j + 8q + 4h + e
__device__ __forceinline__ int hbase(int j, int q, int h) { return 32 * j + 8 * q + 4 * h; }

struct StepCoef {
    float c1, c2, q1, q2, c4, l4, l2, n1, bt, invr, omr, aoma, bt1, a, sp, sa, k1, k2;
};

// One element of p_sample / p_sample_t_1to0 (nsdiff_utils.py:131-157, 209-239).  Contraction is
// off so each product and sum rounds once, as the reference's eager fp32 tensor ops do.
// FAST (split-f16 sampler only, whose network output already differs from fp32 in the seventh digit): v_sqrt_f32 for the
// three square roots and one Newton-refined v_rcp_f32 per denominator instead of the IEEE division / square-root sequences
// (each <= 1-2 ulp; a negative discriminant still gives NaN, as in the reference).
template <bool FAST = false>
__device__ __forceinline__ float posterior(const StepCoef& c, float y, float mu, float g, float eps, float sig, float z,
                                           bool last) {
#pragma clang fp contract(off)
    auto root = [](float x) { return FAST ? __builtin_amdgcn_sqrtf(x) : sqrtf(x); };
    auto recip = [](float d) {
        const float r = __builtin_amdgcn_rcpf(d);
        return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    };
    const float lam1 = c.c1 * g - sig * c.c2;
    const float lam2 = g * g * c.q1 * c.q2 - sig * g * c.c4;
    const float num = -lam1 + root(lam1 * lam1 - c.l4 * lam2);              // NaN if disc < 0, as reference
    const float sy0 = FAST ? num * recip(c.l2) : num / c.l2;
    const float nz = c.n1 * g + c.bt * sy0;
    const float y0r = c.invr * (y - c.omr * mu - eps * root(nz));
    if (last) return y0r;
    const float s1 = c.q1 * g + c.aoma * sy0;
    const float s2 = c.q2 * g + c.bt1 * sy0;
    const float den = c.a * s2 + s1;
    float g0, g1, g2;
    if (FAST) {
        const float r = recip(den);
        g0 = c.sp * s1 * r;
        g1 = c.sa * s2 * r;
        g2 = (c.k1 * s2 + c.k2 * s1) * r;
    } else {
        g0 = c.sp * s1 / den;
        g1 = c.sa * s2 / den;
        g2 = (c.k1 * s2 + c.k2 * s1) / den;
    }
    return g0 * y0r + g1 * y + g2 * mu + root(sig) * z;
}


// TMDM / CARD reverse step (reference TMDM/tmdm_diffusion_utils.py:57-90, 94-104): every coefficient is a
// per-step scalar.  Columns: 0 invr = 1/sqrt(abar_t), 1 omr = 1 - sqrt(abar_t), 2 s = sqrt(1 - abar_t),
// 3..5 gamma_0..2, 6 sqrt(beta_hat_t)
__device__ __forceinline__ float posterior_tmdm(const float* cr, float y, float mu, float eps, float z, bool last) {
#pragma clang fp contract(off)
    const float y0r = cr[0] * (y - cr[1] * mu - eps * cr[2]);
    if (last) return y0r;
    return cr[3] * y0r + cr[4] * y + cr[5] * mu + cr[6] * z;
}

// acc = bias + W . hin on the fp32 matrix pipe; result copied back over hin
__device__ __forceinline__ void mfma_layer(const float* __restrict__ Ws, const float* __restrict__ bias,
                                           float (&hv)[64], int r, int h) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + hbase(j, q, h));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][q * 4 + e] = b[e];
        }
    }
    const float* wrow = Ws + r * WROW + 4 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(wrow + 32 * j * WROW + 32 * jt + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], hv[jt * 16 + q * 4 + e], acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) hv[j * 16 + i] = acc[j][i];
}

}  // namespace chain
