// NsDiff chain, split-f16 matrix-core variant for gfx950 (the default sampler).
//
// Two measured facts about gfx950 shape this kernel (tools/ubench/):
//  1. v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate and excludes the VALU while it executes, so an
//     exact-fp32 kernel (nsdiff_chain.hip) is bound by MFMA + VALU in series.  The f16 matrix pipe is 16x
//     faster and runs beside the VALU, so the 128x128 layers are evaluated as
//          W h  =  W_hi h_hi + W_hi h_lo + W_lo h_hi        (three v_mfma_f32_32x32x16_f16, fp32 accumulate)
//     with x = x_hi + x_lo a two-term fp16 split of the power-of-two-scaled fp32 value (>= 21 significand
//     bits kept; the dropped lo*lo term is 2^-21 relative): ~1e-6 relative on the samples.
//  2. A loop body of 16 KB or more is instruction-FETCH bound: it streams at 2 bytes/clk per SIMD (4 cycles
//     per 8-byte instruction) instead of 2.3 cycles per VALU instruction below 16 KB (ifetch.hip).  The step
//     loop is therefore written for code SIZE: layer 1 (K = 3F) and every bias ride on the matrix pipe as extra
//     k-slots, elementwise work uses 4-byte encodings, LDS addresses are base+immediate, and every rarely
//     needed guard lives in a cold block outside the hot path.
//
// Data flow as in nsdiff_chain.hip: a wave owns 32 chains and activations never change lanes.  For
// v_mfma_f32_32x32x16_f16 the B operand of k-step ks is eight f16 per lane; with k-steps enumerated as
// (tile, 8 consecutive accumulator registers) those are accumulator registers 8ks..8ks+7 of the lane (hidden
// unit 32*(ks>>1) + 16*(ks&1) + 8*(jj>>2) + 4*h + (jj&3) for element jj of half-wave h); the A operand reads
// the weight row with the matching column permutation (bits 2 and 3 swapped inside each 16-block; host).
//
// Elementwise work is done in the log2 domain: x' = x*log2(e) is folded into the per-step scale rows,
//   softplus(x)/ln2 = log2(1 + 2^x')                                   when |x| is provably moderate,
//                   = max( clamp(log2(1+e), (e-e^2/2)/ln2, e/ln2), x' ),  e = 2^min(x',126)   otherwise
// (the clamp is log1p's small-argument branch, the max is torch's threshold=20 branch).  F.normalize is scale
// invariant, so ln2 is never applied.
#include "chain_common.h"

using namespace chain;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HROW = 136;                       // LDS row stride of the f16 weight images, in halfs (272 B)
constexpr int IMG = HID * HROW * 2;             // bytes of one [128][136] image
constexpr float LOG2E = 1.44269504088896341f;
constexpr float HALF_LOG2E = 0.72134752044448170f;
constexpr float LN2 = 0.693147180559945309f;
constexpr float X1_LIMIT = 11.0f;               // fast layer-1 softplus while |x'| < 11 (|x| < 7.6)
constexpr float IN_LIMIT = 2048.0f;             // fast layer 1 while every |input| < 2^11 (f16 range after 2^4 scaling)

// layer 1 as k-slots of one (F=1) or two (F=2,3) MFMAs: 3F x (W_hi x_hi), 3F x (W_hi x_lo), 3F x (W_lo x_hi),
// and the two bias slots (b_hi, b_lo) x 2^4
template <int F> struct L1 {
    static constexpr int SLOTS = 9 * F + 2;
    static constexpr int NK = (SLOTS + 15) / 16;
};

template <int F> struct Lds {                   // byte offsets
    static constexpr int W2H = 0, W2L = IMG, W3H = 2 * IMG, W3L = 3 * IMG;
    static constexpr int W1 = 4 * IMG;                                   // [128][16*NK] halfs
    static constexpr int F32 = W1 + HID * 32 * L1<F>::NK;                // float region
    static constexpr int W4 = F32;                                       // [F][128]
    static constexpr int WSL = W4 + F * HID * 4;                         // [F][128]
    static constexpr int HB = WSL + F * HID * 4;                         // b4[4], bs[4]
    static constexpr int GAM = HB + 32;                                  // per wave [3][128] scale rows
    static constexpr int BYTES = GAM + WAVES * 3 * HID * 4;
    static_assert(BYTES <= 160 * 1024, "LDS budget");
};

// Pass boundaries: left to the scheduler by default (measured 5 % faster than hard fences); -DMPV_HARD_FENCE pins them.
#ifdef MPV_HARD_FENCE
#define MPV_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MPV_FENCE()
#endif

// All LDS reads go through `smem + 32-bit offset`: the opaque per-step copies of the base OFFSETS keep the
// loads inside the step loop without turning the pointers into generic (flat) ones.
#define LDS_F4(smem, off) (*reinterpret_cast<const f32x4*>((smem) + (off)))
#define LDS_H8(smem, off) (*reinterpret_cast<const f16x8*>((smem) + (off)))

// ---- softplus(x)/ln2 from x' = x*log2(e), in place on one accumulator tile (16 registers); sum(l^2) -> ss[0..3]
// Each stage is a PASS over the 16 independent registers (the dependent chain mul -> exp -> add -> log -> fma of
// one element is spread over five passes, so neighbouring instructions are independent).
template <bool SAFE>
__device__ __forceinline__ void softplus_tile(f32x16& a, const unsigned char* smem, unsigned grow, int j, float (&ss)[4]) {
    f32x4 gm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gm[q] = LDS_F4(smem, grow + (32 * j + 8 * q) * 4);
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] *= gm[i >> 2][i & 3];                       // x'
    MPV_FENCE();
    if (!SAFE) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
        MPV_FENCE();
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] += 1.0f;
        MPV_FENCE();
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_logf(a[i]);
        MPV_FENCE();
    } else {
        f32x16 e, c, lo;
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(a[i], -3.0e38f, 126.0f));
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_logf(1.0f + e[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) lo[i] = e[i] * __builtin_fmaf(e[i], -HALF_LOG2E, LOG2E);   // (e - e^2/2)/ln2
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_fmed3f(c[i], lo[i], e[i] * LOG2E);  // <= e/ln2
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_fmed3f(c[i], a[i], 3.0e38f);        // max(., x')
        MPV_FENCE();
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) ss[i & 3] = __builtin_fmaf(a[i], a[i], ss[i & 3]);
    MPV_FENCE();
}

template <bool SAFE>
__device__ __forceinline__ float activate_log2(f32x16 (&acc)[4], const unsigned char* smem, unsigned grow) {
    float ss[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j) softplus_tile<SAFE>(acc[j], smem, grow, j, ss);
    float t = (ss[0] + ss[1]) + (ss[2] + ss[3]);
    t += __shfl_xor(t, 32);
    return 1.0f / fmaxf(sqrtf(t), 1e-12f / LN2);           // 1 / max(||l||, F.normalize eps / ln2)
}

// t = a*scale in [0, 2^12] -> f16 pair (hi, lo) with hi + lo = t to >= 22 significand bits.
//  default      : hi = t with the low 13 significand bits cleared, lo = RTZ_f16(t - hi)
//                 (v_mul, v_and with an SGPR mask, v_sub, v_cvt_pkrtz: 4-byte encodings)
//  MPV_SPLIT_MIX: hi = RNE_f16(t), lo = RNE_f16(fma(a, scale, -hi)) -- two v_fma_mix{lo,hi}_f16 per element;
//                 1.5 fewer instructions per element but measured 1 % slower (VOP3P issue cost)
__device__ __forceinline__ void split_f16(f32x16 (&acc)[4], float scale, unsigned mask, f16x8 (&bh)[8], f16x8 (&bl)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x16& a = acc[j];
#ifndef MPV_SPLIT_MIX
        f32x16 hi;
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] *= scale;
        MPV_FENCE();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = a[i];                       // (a vector-element lvalue must not go through bit_cast)
            hi[i] = __uint_as_float(__float_as_uint(v) & mask);
        }
        MPV_FENCE();
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] -= hi[i];
        MPV_FENCE();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const auto ph = __builtin_amdgcn_cvt_pkrtz(hi[2 * p], hi[2 * p + 1]);
            const auto pl = __builtin_amdgcn_cvt_pkrtz(a[2 * p], a[2 * p + 1]);
            const int ks = 2 * j + (p >> 2), q = p & 3;
            bh[ks][2 * q] = (_Float16)ph[0]; bh[ks][2 * q + 1] = (_Float16)ph[1];
            bl[ks][2 * q] = (_Float16)pl[0]; bl[ks][2 * q + 1] = (_Float16)pl[1];
        }
#else
        (void)mask;
        u32x4 ph[2], pl[2];
#pragma unroll
        for (int p = 0; p < 8; ++p) {                       // hi = RNE_f16(a * scale), packed (even, odd)
            unsigned d;
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(d) : "v"(a[2 * p]), "v"(scale));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(d) : "v"(a[2 * p + 1]), "v"(scale));
            ph[p >> 2][p & 3] = d;
        }
        MPV_FENCE();
#pragma unroll
        for (int p = 0; p < 8; ++p) {                       // lo = RNE_f16(fma(a, scale, -hi)), hi read as f16
            unsigned d;
            const unsigned h = ph[p >> 2][p & 3];
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a[2 * p]), "v"(scale), "v"(h));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                : "+v"(d) : "v"(a[2 * p + 1]), "v"(scale), "v"(h));
            pl[p >> 2][p & 3] = d;
        }
        bh[2 * j] = __builtin_bit_cast(f16x8, ph[0]); bh[2 * j + 1] = __builtin_bit_cast(f16x8, ph[1]);
        bl[2 * j] = __builtin_bit_cast(f16x8, pl[0]); bl[2 * j + 1] = __builtin_bit_cast(f16x8, pl[1]);
#endif
        MPV_FENCE();
    }
}

// acc = bias + (W_hi + W_lo)(h_hi + h_lo) (without lo*lo) on the f16 matrix pipe.  The bias enters as one more
// MFMA per tile: A = (b_hi, b_lo, 0...) of the lane's row (registers, zero in the upper half-wave), B = (c, c, 0...)
__device__ __forceinline__ void mfma_layer_f16(const unsigned char* smem, unsigned wbase, int img_hi, int img_lo, const unsigned (&abias)[4],
                                               const f16x8& bone, const f16x8 (&bh)[8], const f16x8 (&bl)[8],
                                               f32x16 (&acc)[4]) {
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32x4 ab = {abias[j], 0u, 0u, 0u};
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ab), bone, zero, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int off = j * 32 * (HROW * 2) + ks * 32;
            const f16x8 ah = LDS_H8(smem, wbase + img_hi + off);
            const f16x8 al = LDS_H8(smem, wbase + img_lo + off);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc[j], 0, 0, 0);
        }
    }
}

// ---- layer 1 ---------------------------------------------------------------------------------------------
// slot s of the k dimension multiplies A[row][s] (host image) with this per-chain value
template <int F>
__device__ __forceinline__ _Float16 l1_slot(int s, const _Float16 (&xh)[3 * F], const _Float16 (&xl)[3 * F]) {
    if (s < 3 * F) return xh[s];
    if (s < 6 * F) return xl[s - 3 * F];
    if (s < 9 * F) return xh[s - 6 * F];
    if (s < 9 * F + 2) return (_Float16)16.0f;            // bias slots: (b_hi, b_lo) x 2^4
    return (_Float16)0.0f;
}

template <int F>
__device__ __forceinline__ void layer1_mfma(const unsigned char* smem, unsigned w1row, const float (&in)[3 * F], int h, f32x16 (&acc)[4]) {
    _Float16 xh[3 * F], xl[3 * F];
#pragma unroll
    for (int c = 0; c < 3 * F; ++c) {
        const float u = in[c] * 16.0f;                     // |u| < 2^15 guaranteed by the caller's range check
        xh[c] = (_Float16)u;
        xl[c] = (_Float16)(u - (float)xh[c]);
    }
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f16x8 b[L1<F>::NK];
#pragma unroll
    for (int k = 0; k < L1<F>::NK; ++k)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const _Float16 v0 = l1_slot<F>(16 * k + jj, xh, xl), v1 = l1_slot<F>(16 * k + 8 + jj, xh, xl);
            b[k][jj] = h ? v1 : v0;
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < L1<F>::NK; ++k) {
            const f16x8 a = LDS_H8(smem, w1row + (j * 32) * (32 * L1<F>::NK) + k * 32);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[k], k == 0 ? zero : acc[j], 0, 0, 0);
        }
    }
}

// cold path: exact fp32 layer 1 on the VALU from the global fp32 weights, result scaled like the MFMA path
template <int F>
__device__ __forceinline__ void layer1_valu(const float* __restrict__ w1t, const float* __restrict__ b1, const float (&in)[3 * F],
                                         int h, float out_scale, f32x16 (&acc)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = 32 * j + (i & 3) + 8 * (i >> 2) + 4 * h;
            float a = b1[idx];
#pragma unroll
            for (int c = 0; c < 3 * F; ++c) a = __builtin_fmaf(w1t[c * HID + idx], in[c], a);
            acc[j][i] = a * out_scale;
        }
}

template <int F> struct Vec { float v[F]; };
struct Vec4 { float v[4]; };

// one Philox block -> four normals, out of line: it runs once per 4/F steps and would otherwise sit in the hot loop
__device__ __noinline__ Vec4 philox_block(unsigned long long chain, unsigned block, unsigned seed_lo, unsigned seed_hi, unsigned off_lo) {
    Vec4 r;
    philox_normals(chain, block, seed_lo, seed_hi, off_lo, r.v);
    return r;
}

template <int F>
__device__ __forceinline__ void next_normals(unsigned long long chain, unsigned k, const ChainArgs& p, float (&cache)[4], float (&z)[4]) {
    if (F == 1 || F == 2) {
        const unsigned idx = k * F;
        if ((idx & 3u) == 0u) {
            const Vec4 b = philox_block(chain, idx >> 2, p.seed_lo, p.seed_hi, p.off_lo);
#pragma unroll
            for (int i = 0; i < 4; ++i) cache[i] = b.v[i];
        }
        const unsigned e = idx & 3u;
        if (F == 1) {
            z[0] = e == 0u ? cache[0] : (e == 1u ? cache[1] : (e == 2u ? cache[2] : cache[3]));
        } else {
            z[0] = e ? cache[2] : cache[0];
            z[1] = e ? cache[3] : cache[1];
        }
    } else {
        const Vec4 b = philox_block(chain, k, p.seed_lo, p.seed_hi, p.off_lo);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = b.v[i];
    }
}

struct StepCtx {                 // everything a step needs besides the chain state (all by value / LDS offsets)
    const unsigned char* smem;
    const float *coef, *w1t, *b1;   // global: per-step coefficient rows; fp32 layer-1 weights for the exact path
    float l1_scale;
    unsigned a_base, a_base3, w1_base, f_base, g_base;   // a_base3 = a_base + W3H: keeps every immediate < 64 KB
    unsigned ab2[4], ab3[4];
    float bone2, bone3;
    unsigned hi_mask;
    int h;
};

// One reverse step for 32 chains.  SAFE = false is the hot path: layer 1 on the matrix pipe and log2(1 + 2^x') for
// every softplus; it also returns `ok` = every guard passed (inputs inside the f16 range of the layer-1 split, every
// activation vector finite with norm > 1, i.e. no exp2 overflow and no all-tiny vector whose direction the plain
// log2(1+e) would lose).  SAFE = true is the exact evaluation (fp32 VALU layer 1, guarded softplus); the kernel
// re-runs a step through it (a cold, out-of-line call) in the rare case `ok` is false.
template <int F, bool SAFE>
__device__ __forceinline__ Vec<F> chain_step(const StepCtx& k, int t, const Vec<F>& y, const Vec<F>& mu,
                                             const Vec<F>& g, const Vec<F>& z, bool& ok) {
    using LD = Lds<F>;
    const float* cr = k.coef + t * MPV_COEF_COLS;
    f32x16 acc[4];
    f16x8 bh[8], bl[8];
    float in[3 * F], insum = 0.0f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        in[f] = y.v[f]; in[F + f] = mu.v[f]; in[2 * F + f] = g.v[f];
    }
    float inv, nrm_lo, nrm_hi;
    if (SAFE) {
        layer1_valu<F>(k.w1t, k.b1, in, k.h, k.l1_scale, acc);
    } else {
#pragma unroll
        for (int c = 0; c < 3 * F; ++c) insum += fabsf(in[c]);
        ok = insum < IN_LIMIT;                               // false for NaN/inf as well
        layer1_mfma<F>(k.smem, k.w1_base, in, k.h, acc);
    }
    inv = activate_log2<SAFE>(acc, k.smem, k.g_base);
    nrm_lo = inv;
    nrm_hi = inv;
    split_f16(acc, inv * 4096.0f, k.hi_mask, bh, bl);          // activations in [0,1] scaled by 2^12
    f16x8 bone = {0, 0, 0, 0, 0, 0, 0, 0};
    bone[0] = bone[1] = (_Float16)k.bone2;
    mfma_layer_f16(k.smem, k.a_base, LD::W2H, LD::W2L, k.ab2, bone, bh, bl, acc);
    inv = activate_log2<SAFE>(acc, k.smem, k.g_base + HID * 4);
    nrm_lo = fminf(nrm_lo, inv);
    nrm_hi = fmaxf(nrm_hi, inv);
    split_f16(acc, inv * 4096.0f, k.hi_mask, bh, bl);
    bone[0] = bone[1] = (_Float16)k.bone3;
    mfma_layer_f16(k.smem, k.a_base3, 0, LD::W3L - LD::W3H, k.ab3, bone, bh, bl, acc);
    inv = activate_log2<SAFE>(acc, k.smem, k.g_base + 2 * HID * 4);
    nrm_lo = fminf(nrm_lo, inv);
    nrm_hi = fmaxf(nrm_hi, inv);
    // inv = 1/||l||: guards  ||l|| > 1  and finite  <=>  0 < inv < 1   (NaN fails both)
    if (!SAFE) ok = ok && (nrm_lo > 0.0f) && (nrm_hi < 1.0f);
    // ---- heads on h3 = acc*inv in [0,1]: eps = W4 h3 + b4 ; sigma = softplus((Ws ln2) softplus(h3)/ln2 + bs)
    float ea[F][4], sa[F][4];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) ea[f][i] = sa[f][i] = 0.0f;
    const float inv_l2e = inv * LOG2E;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x16& a = acc[j];
        f32x16 sp;
#pragma unroll
        for (int i = 0; i < 16; ++i) sp[i] = a[i] * inv_l2e;
        MPV_FENCE();
#pragma unroll
        for (int i = 0; i < 16; ++i) sp[i] = __builtin_amdgcn_exp2f(sp[i]);       // 1 + e^h in [2, 3.72]: no
        MPV_FENCE();                                                                //   small-argument or
#pragma unroll                                                                      //   overflow cases
        for (int i = 0; i < 16; ++i) sp[i] += 1.0f;
        MPV_FENCE();
#pragma unroll
        for (int i = 0; i < 16; ++i) sp[i] = __builtin_amdgcn_logf(sp[i]);
        MPV_FENCE();
#pragma unroll
        for (int f = 0; f < F; ++f) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = LDS_F4(k.smem, k.f_base + (LD::W4 - LD::F32) + (f * HID + 32 * j + 8 * q) * 4);
                const f32x4 wsg = LDS_F4(k.smem, k.f_base + (LD::WSL - LD::F32) + (f * HID + 32 * j + 8 * q) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ea[f][e] = __builtin_fmaf(w4[e], a[q * 4 + e], ea[f][e]);       // un-normalised: * inv below
                    sa[f][e] = __builtin_fmaf(wsg[e], sp[q * 4 + e], sa[f][e]);
                }
            }
        }
        MPV_FENCE();
    }
    const StepCoef c = {cr[0], cr[1], cr[2],  cr[3],  cr[4],  cr[5],  cr[6],  cr[7],  cr[8],
                        cr[9], cr[10], cr[11], cr[12], cr[13], cr[14], cr[15], cr[16], cr[17]};
    const float* hb = reinterpret_cast<const float*>(k.smem + LD::HB);
    Vec<F> out;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const float ep = ((ea[f][0] + ea[f][1]) + (ea[f][2] + ea[f][3])) * inv;
        const float sp = (sa[f][0] + sa[f][1]) + (sa[f][2] + sa[f][3]);
        const float eps = ep + __shfl_xor(ep, 32) + hb[f];
        const float sig = softplus_dev(sp + __shfl_xor(sp, 32) + hb[4 + f]);
        out.v[f] = posterior(c, y.v[f], mu.v[f], g.v[f], eps, sig, z.v[f], t == 0);
    }
    return out;
}

template <int F>
__device__ __noinline__ Vec<F> chain_step_safe(StepCtx k, int t, Vec<F> y, Vec<F> mu, Vec<F> g, Vec<F> z) {
    bool ok;
    return chain_step<F, true>(k, t, y, mu, g, z, ok);
}

template <int F, bool PHILOX>
__global__ __launch_bounds__(THREADS, WAVES / 4) void nsdiff_chain_f16_kernel(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using LD = Lds<F>;

    // ---- stage the weights once per workgroup (rows padded from 256 B to 272 B) ----------------
    for (int i = threadIdx.x; i < HID * 16; i += THREADS) {          // 16-byte pieces: 16 per row
        const int row = i >> 4, piece = i & 15;
        const int dst = row * (HROW * 2) + piece * 16, src = row * HID + piece * 8;
        *reinterpret_cast<uint4*>(smem + LD::W2H + dst) = *reinterpret_cast<const uint4*>(p.w2h + src);
        *reinterpret_cast<uint4*>(smem + LD::W2L + dst) = *reinterpret_cast<const uint4*>(p.w2l + src);
        *reinterpret_cast<uint4*>(smem + LD::W3H + dst) = *reinterpret_cast<const uint4*>(p.w3h + src);
        *reinterpret_cast<uint4*>(smem + LD::W3L + dst) = *reinterpret_cast<const uint4*>(p.w3l + src);
    }
    for (int i = threadIdx.x; i < HID * 2 * L1<F>::NK; i += THREADS)     // W1 image, 16-byte pieces
        *reinterpret_cast<uint4*>(smem + LD::W1 + i * 16) = *reinterpret_cast<const uint4*>(p.w1img + i * 8);
    {
        float* f = reinterpret_cast<float*>(smem);
        for (int i = threadIdx.x; i < F * HID; i += THREADS) {
            f[LD::W4 / 4 + i] = p.w4[i];
            f[LD::WSL / 4 + i] = p.wsl[i];
        }
        if (threadIdx.x < F) {
            f[LD::HB / 4 + threadIdx.x] = p.b4[threadIdx.x];
            f[LD::HB / 4 + 4 + threadIdx.x] = p.bs[threadIdx.x];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r_lane = lane & 31, h_lane = lane >> 5;
    StepCtx k0;
    k0.smem = smem;
    k0.coef = p.coef; k0.w1t = p.w1t; k0.b1 = p.b1; k0.l1_scale = p.l1_scale;
    // per-lane LDS base offsets: everything in the step is base + immediate
    k0.a_base = r_lane * (HROW * 2) + h_lane * 16;                               // weight images: row, k-group
    k0.a_base3 = k0.a_base + LD::W3H;
    k0.w1_base = LD::W1 + r_lane * (32 * L1<F>::NK) + h_lane * 16;
    k0.f_base = LD::F32 + h_lane * 16;                                           // float tables at hidden 4h
    k0.g_base = LD::GAM + wave * (3 * HID * 4) + h_lane * 16;                    // this wave's scale rows
    // biases of layers 2/3 as MFMA A operands (row = tile row of this lane; only k-slots 0,1 of half-wave 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        k0.ab2[j] = h_lane ? 0u : p.b2p[32 * j + r_lane];
        k0.ab3[j] = h_lane ? 0u : p.b3p[32 * j + r_lane];
    }
    k0.bone2 = p.bone2;
    k0.bone3 = p.bone3;
    k0.hi_mask = p.hi_mask;                                                       // 0xFFFFE000, stays in an SGPR
    k0.h = h_lane;
    unsigned char* const gam_w = smem + LD::GAM + wave * (3 * HID * 4);
    const int gl = (lane & 31) * 16;                                              // scale-row staging: 16 B per lane

    for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < p.n_tiles; tile += (long long)gridDim.x * WAVES) {
        const long long n_raw = tile * TILE + r_lane;
        const bool valid = n_raw < p.n_chains;
        const long long n = valid ? n_raw : p.n_chains - 1;
        const int s = (int)(n % p.S);
        const long long row = n / p.S;            // (w*B + b)*O + o
        const int o = (int)(row % p.O);
        const long long gi = row / p.O;           // w*B + b
        const int b = (int)(gi % p.B);
        const long long w = gi / p.B;

        Vec<F> g, mu, y, zv;
        float z[4], zc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int f = 0; f < F; ++f) {
            g.v[f] = p.gx[row * F + f];
            mu.v[f] = p.y0hat ? p.y0hat[row * F + f] : 0.0f;
        }
        long long nbase = 0;
        if (!PHILOX) {
            const int c = s / p.P, pp = s - c * p.P;
            nbase = (((w * p.C + c) * p.T) * ((long long)p.B * p.P) + (long long)b * p.P + pp) * ((long long)p.O * F) +
                    (long long)o * F;
        }
        if (PHILOX) {
            next_normals<F>((unsigned long long)n, 0u, p, zc, z);
        } else {
#pragma unroll
            for (int f = 0; f < F; ++f) z[f] = p.noise[nbase + f];
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
#pragma clang fp contract(off)
            y.v[f] = sqrtf(g.v[f]) * z[f] + mu.v[f];
        }
        if (lane < 32) {                                   // scale rows of the first step t = T-1
            const long long g0 = (long long)(p.T - 1) * HID + (gl >> 2);
            *reinterpret_cast<f32x4*>(gam_w + gl) = *reinterpret_cast<const f32x4*>(p.g1 + g0);
            *reinterpret_cast<f32x4*>(gam_w + HID * 4 + gl) = *reinterpret_cast<const f32x4*>(p.g2 + g0);
            *reinterpret_cast<f32x4*>(gam_w + 2 * HID * 4 + gl) = *reinterpret_cast<const f32x4*>(p.g3 + g0);
        }
#pragma nounroll
        for (int t = p.T - 1; t >= 0; --t) {
            // The LDS operands are loop invariant in t; left alone, LICM hoists hundreds of them out of the step
            // loop and spills them.  Passing the few base OFFSETS through an opaque move keeps the loads in the
            // loop while the address arithmetic itself stays hoisted.
            StepCtx k = k0;
            asm volatile("" : "+v"(k.a_base), "+v"(k.a_base3), "+v"(k.w1_base), "+v"(k.f_base), "+v"(k.g_base));
            // prefetch the next step's scale rows into registers now, park them in LDS at the end of the step
            f32x4 nx1 = {0, 0, 0, 0}, nx2 = nx1, nx3 = nx1;
            if (t > 0 && lane < 32) {
                const long long g0 = (long long)(t - 1) * HID + (gl >> 2);
                nx1 = *reinterpret_cast<const f32x4*>(p.g1 + g0);
                nx2 = *reinterpret_cast<const f32x4*>(p.g2 + g0);
                nx3 = *reinterpret_cast<const f32x4*>(p.g3 + g0);
            }
            if (t > 0) {
                if (PHILOX) {
                    next_normals<F>((unsigned long long)n, (unsigned)(p.T - t), p, zc, z);
                } else {
#pragma unroll
                    for (int f = 0; f < F; ++f) z[f] = p.noise[nbase + (long long)(p.T - t) * p.noise_kstride + f];
                }
            }
#pragma unroll
            for (int f = 0; f < F; ++f) zv.v[f] = z[f];
#ifdef MPV_PAD_NOPS    // experiment: grow the hot loop by 4*MPV_PAD_NOPS bytes to probe the 16 KB instruction-fetch cliff
#pragma unroll
            for (int i = 0; i < MPV_PAD_NOPS; ++i) asm volatile("s_nop 0");
#endif
            bool ok = true;
            Vec<F> yn = chain_step<F, false>(k, t, y, mu, g, zv, ok);
            if (__builtin_expect(!__all(ok), 0)) yn = chain_step_safe<F>(k, t, y, mu, g, zv);
            y = yn;
            if (t > 0 && lane < 32) {                       // all reads of this step's rows are done (same wave, in order)
                *reinterpret_cast<f32x4*>(gam_w + gl) = nx1;
                *reinterpret_cast<f32x4*>(gam_w + HID * 4 + gl) = nx2;
                *reinterpret_cast<f32x4*>(gam_w + 2 * HID * 4 + gl) = nx3;
            }
        }
        if (valid && h_lane == 0) {
#pragma unroll
            for (int f = 0; f < F; ++f) p.samples[(row * F + f) * p.S + s] = y.v[f];
        }
    }
}

template <int F>
int launch_f16_f(mpv_ctx* ctx, const ChainArgs& a, bool philox, int grid) {
    auto kp = nsdiff_chain_f16_kernel<F, true>;
    auto kn = nsdiff_chain_f16_kernel<F, false>;
    constexpr int lds = Lds<F>::BYTES;
    if ((size_t)lds > ctx->max_lds) return mpv_fail(ctx, MPV_EINVAL, "device LDS too small for the resident-weight chain kernel");
    MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(philox ? kp : kn),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (philox)
        hipLaunchKernelGGL(kp, dim3(grid), dim3(THREADS), lds, ctx->stream, a);
    else
        hipLaunchKernelGGL(kn, dim3(grid), dim3(THREADS), lds, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

}  // namespace

int launch_chain_f16(mpv_ctx* ctx, const chain::ChainArgs& a, bool philox, int grid, int nf) {
    switch (nf) {
        case 1: return launch_f16_f<1>(ctx, a, philox, grid);
        case 2: return launch_f16_f<2>(ctx, a, philox, grid);
        case 3: return launch_f16_f<3>(ctx, a, philox, grid);
        default: return mpv_fail(ctx, MPV_EINVAL, "the f16x3 sampler supports nf 1..3 (use MPV_GEMM_F32 for nf = 4)");
    }
}
