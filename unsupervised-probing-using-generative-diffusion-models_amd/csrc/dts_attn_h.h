// DiffusionTS attention on the f16 matrix pipe at fp32-equivalent accuracy (head width 16 = the shipped 64 / 4), forward, dQ and dK-dV.
// Same structure and arithmetic order as the exact-fp32 kernels of dts_kernels.h (one workgroup per (sequence, head), 16-query / 16-key tiles,
// online softmax in base 2, the score tile in accumulator layout reused as the next product's B operand) with every 16 x 16 x 16 contraction as
// ONE v_mfma_f32_16x16x16_f16 per term of the two-term split x = hi + lo of power-of-two scaled operands (22 significant bits; products
// hi*hi + hi*lo + lo*hi, fp32 accumulate; the dropped lo*lo is 2^-22 relative) instead of four v_mfma_f32_16x16x4_f32: 3 x 16.5 cycles against 4 x 32
// (tools/ubench/mfma_rates.hip).  The fp32 kernels spent 56 % (forward), 70 % (dQ) and 76 % (dK-dV) of their time on the matrix pipe.
// K / V (Q / dO in the dK-dV pass) are split ONCE while they are staged into LDS, as f16 hi / lo planes, row-major [tokens][16].  Operands indexed by
// (token, 4 head columns) are one ds_read_b64 per plane.  Operands indexed by (head column, 4 tokens) -- K^T in dQ, Q^T and dO^T in dK-dV -- come out
// of the SAME planes through gfx950's transposed LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns and each lane receives one
// column): one instruction per plane where four 2-byte reads and two v_perm were needed (dK-dV: 151 -> 127 instructions per query tile, 540 -> 485 us
// per launch at configs[2]); a transposed copy would cost a workgroup per CU, and these kernels live on occupancy.  The forward kernel, where V is used
// in no other way, keeps a transposed plane [16][tokens] read with ds_read_b64 (measured: 274 us against 285 us with transposed reads of a row-major V).
#pragma once
#include <type_traits>

namespace dts {

typedef _Float16 ah4 __attribute__((ext_vector_type(4)));
#define DTS_MMH(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x16f16((a), (b), (acc), 0, 0, 0)
// acc += (ah + al) (bh + bl) without the lo * lo term, smallest terms first
#define DTS_MMH3(acc, ah, al, bh, bl) \
    do {                              \
        DTS_MMH(acc, al, bh);         \
        DTS_MMH(acc, ah, bl);         \
        DTS_MMH(acc, ah, bh);         \
    } while (0)

// the same on v_mfma_f32_16x16x32_f16: eight k per lane (k = 8 (lane >> 4) + j), twice the work of the x16 shape in the same 16.5 cycles
typedef _Float16 ah8 __attribute__((ext_vector_type(8)));
#define DTS_MMH32(acc, a, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (acc), 0, 0, 0)
#define DTS_MMH3_32(acc, ah, al, bh, bl) \
    do {                                 \
        DTS_MMH32(acc, al, bh);          \
        DTS_MMH32(acc, ah, bl);          \
        DTS_MMH32(acc, ah, bh);          \
    } while (0)
__device__ __forceinline__ ah8 ah8_of(const ah4& a, const ah4& b) { return ah8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

#ifndef DTS_SPLIT_RNE
#define DTS_SPLIT_RNE 0              // 1: both halves rounded to nearest (24 bits, 4 + conversions per element); 0: truncating split (22 bits, 3 per element)
#endif
constexpr int AH_RS = 16;            // halves per row of a row-major plane ([tokens][16]; 32-byte rows: the ds_read_b64 of 16 rows is 2-way conflicted)
constexpr int AH_MAXQ = 4;           // float4 blocks of a head tile a thread holds between the max pass and the split pass: tiles of <= 256 tokens
constexpr float AH_PSCALE = 8192.0f; // probabilities (<= 1) are split as p * 2^13 ...
constexpr float AH_PLOG = 13.0f;     // ... formed directly as exp2(s - m + 13)

// x = hi + lo: hi = the 11 leading significand bits (exact in f16), lo = f16(x - hi) truncated: 22 significant bits as long as lo is a NORMAL
// f16, i.e. for |x| >= 2^-2.  f16 has five exponent bits, so every operand is first multiplied by a power of two that puts the largest element of
// its tile / column at 2^13..2^14 (pow2_scale): elements down to 2^-15 of the largest keep all 22 bits, smaller ones lose precision only relative
// to themselves, and gradients carrying a 1 / (batch x length) normaliser (1e-9 at BASELINE configs[2]) are as safe as O(1) activations.  The
// powers of two are divided out of the fp32 accumulators (exactly).  Measured on the full-size replay test (tests/test_full_size_gpu.py, 16
// sequences of a 3200-sequence batch, 5 steps): WITHOUT the scaling the low halves of small operands flush to zero and 3 of 16 sequences stay
// on the oracle's trajectory; with it 14 of 16 (the exact-fp32 kernels: 13 of 16); rounding both halves to nearest instead (DTS_SPLIT_RNE, 24
// bits, one more instruction per element) also gives 14 of 16 and is 4 % slower.
__device__ __forceinline__ void attn_split4(const float v0, const float v1, const float v2, const float v3, ah4& hi, ah4& lo) {
    const float v[4] = {v0, v1, v2, v3};
#if DTS_SPLIT_RNE
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        hi[c] = (_Float16)v[c];
        lo[c] = (_Float16)(v[c] - (float)hi[c]);
    }
#else
    float h[4], l[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        h[c] = __uint_as_float(__float_as_uint(v[c]) & 0xFFFFE000u);   // 11 leading bits: exact in f16
        l[c] = v[c] - h[c];
    }
    typedef __fp16 pk2 __attribute__((ext_vector_type(2)));
    const pk2 h01 = __builtin_amdgcn_cvt_pkrtz(h[0], h[1]), h23 = __builtin_amdgcn_cvt_pkrtz(h[2], h[3]);
    const pk2 l01 = __builtin_amdgcn_cvt_pkrtz(l[0], l[1]), l23 = __builtin_amdgcn_cvt_pkrtz(l[2], l[3]);
    hi = ah4{(_Float16)h01[0], (_Float16)h01[1], (_Float16)h23[0], (_Float16)h23[1]};
    lo = ah4{(_Float16)l01[0], (_Float16)l01[1], (_Float16)l23[0], (_Float16)l23[1]};
#endif
}

__device__ __forceinline__ float pow2_scale(float mx) {
    if (!(mx > 0.0f) || !(mx < 3.0e38f)) return 1.0f;                   // all-zero tile, or inf / NaN (which then propagate as they would in fp32)
    int e;
    (void)frexpf(mx, &e);                                               // mx = m 2^e, m in [0.5, 1)
    return ldexpf(1.0f, 14 - e);
}

// the same with the reciprocal (exact), for products whose scale is divided out again per tile
__device__ __forceinline__ float pow2_scale(float mx, float& inv) {
    inv = 1.0f;
    if (!(mx > 0.0f) || !(mx < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(mx, &e);
    inv = ldexpf(1.0f, e - 14);
    return ldexpf(1.0f, 14 - e);
}

// ---- the four lanes of a column (r16, r16 + 16, r16 + 32, r16 + 48) without the LDS crossbar (round 4) -------------------------------------------
// gfx950's v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane rows of two registers in the VALU: with both operands = v the pair holds
// (v, v xor-16) resp. (v, v xor-32) afterwards.  __shfl_xor(v, 16 / 32) compiles to ds_bpermute_b32 + s_waitcnt: an LDS round trip of > 100 cycles
// in the middle of every tile step's dependency chain; the kernels that live on few waves per SIMD (dts_attn_bwd_h_kernel) are bound by that chain.
__device__ __forceinline__ float vmax_raw(float a, float b) {          // v_max_f32 without the canonicalising v_max(v, v) clang puts in front of fmaxf
    float r;                                                           // for values that come out of a permute (NaNs still propagate: IEEE mode off is
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));            // not assumed -- v_max_f32 returns the non-NaN operand, as fmaxf does)
    return r;
}
#ifndef DTS_COL4_SWAP
#define DTS_COL4_SWAP 0              // measured (round 4, 10-step c3 plan): the swap forms are 2 % SLOWER in the issue-bound kernels (forward 273 -> 281 us, dQ 373
#endif                               // -> 382: each swap needs a v_mov of its second operand) and shorten the merged backward's chain (935 -> 875 us with the
                                     // one-tile-behind dQ product); kept as a compile-time choice
__device__ __forceinline__ float col4_max(float v) {
#if !DTS_COL4_SWAP
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
#else
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax_raw(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax_raw(__uint_as_float(b[0]), __uint_as_float(b[1]));
#endif
}
__device__ __forceinline__ float col4_sum(float v) {                   // (v + v^16) + (that ^ 32): the association of the two-shuffle form
#if !DTS_COL4_SWAP
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
#else
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
#endif
}

__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}

// A head tile [n][16] (token stride ld) in registers: thread i holds the float4 blocks i, i + 256, ... (block = token * 4 + column quad).
struct HeadRegs { f32x4 v[AH_MAXQ]; };

__device__ __forceinline__ float head_load(const float* src, int ld, int n, int np, HeadRegs& h) {
    float m = 0.0f;
#pragma unroll
    for (int u = 0; u < AH_MAXQ; ++u) {
        const int i = threadIdx.x + 256 * u, r = i >> 2, d = 4 * (i & 3);
        h.v[u] = (i < np * 4 && r < n) ? *reinterpret_cast<const f32x4*>(src + (long long)r * ld + d) : f32x4{0, 0, 0, 0};
        m = fmaxf(m, fmaxf(fmaxf(fabsf(h.v[u][0]), fabsf(h.v[u][1])), fmaxf(fabsf(h.v[u][2]), fabsf(h.v[u][3]))));
    }
    return m;
}

// registers (x scale) -> f16 hi / lo planes: row-major [np][AH_RS] and / or transposed [16][np]
__device__ __forceinline__ void head_store(const HeadRegs& h, float scale, int np, _Float16* rh, _Float16* rl, _Float16* th, _Float16* tl) {
#pragma unroll
    for (int u = 0; u < AH_MAXQ; ++u) {
        const int i = threadIdx.x + 256 * u, r = i >> 2, d = 4 * (i & 3);
        if (i >= np * 4) break;
        ah4 hi, lo;
        attn_split4(h.v[u][0] * scale, h.v[u][1] * scale, h.v[u][2] * scale, h.v[u][3] * scale, hi, lo);
        if (rh) {
            *reinterpret_cast<ah4*>(rh + r * AH_RS + d) = hi;
            *reinterpret_cast<ah4*>(rl + r * AH_RS + d) = lo;
        }
        if (th) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                th[(d + e) * np + r] = hi[e];
                tl[(d + e) * np + r] = lo[e];
            }
        }
    }
}

// workgroup-wide maxima of NT per-thread values -> power-of-two scales (one barrier pair; red: LDS [4][NT])
template <int NT>
__device__ __forceinline__ void wg_scales(float (&m)[NT], float* red, float (&sc)[NT]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        m[k] = wave_max(m[k]);
        if (lane == 0) red[wave * NT + k] = m[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NT; ++k) sc[k] = pow2_scale(fmaxf(fmaxf(red[k], red[NT + k]), fmaxf(red[2 * NT + k], red[3 * NT + k])));
}

// Four tokens' values of one head column out of a row-major plane, p[(r0 + j) * AH_RS + col], j = 0..3, for all 16 columns at once as ONE gfx950
// transposed LDS read (ds_read_b64_tr_b16): a group of 16 lanes reads the block rows r0 .. r0 + 3 x 16 columns and lane i of the group receives
// column i of the four rows.  Lane 4 q + p of the group supplies the address of row r0 + q, columns 4 p ..: `lane_addr` below.  All 64 lanes must be
// active (the gather crosses lanes).
__device__ __forceinline__ ah4 column4_tr(const _Float16* lane_addr) {
    typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    const fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(lane_addr));
    return __builtin_bit_cast(ah4, v);
}

__device__ __forceinline__ void load_split4(const float* p, float scale, ah4& hi, ah4& lo) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    attn_split4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale, hi, lo);
}

// the 16 head columns of ONE token as a B-operand column (lane (r16, g) holds columns 4 g .. 4 g + 3 of token r16): split with the token's own
// power of two (its four lanes agree on it with two shuffles; returned) -- no pass over the head tile is needed for operands that enter as columns
__device__ __forceinline__ float load_split_col(const float* p, ah4& hi, ah4& lo) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    const float m = col4_max(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    const float sc = pow2_scale(m);
    attn_split4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc, hi, lo);
    return sc;
}

// the same for values already in registers
__device__ __forceinline__ float split_col(const f32x4& v, ah4& hi, ah4& lo) {
    const float m = col4_max(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    const float sc = pow2_scale(m);
    attn_split4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc, hi, lo);
    return sc;
}

// a 16 x 16 tile of gradient-like values in accumulator layout [row 4 g + r][column r16] -> split with a power of two per COLUMN (the four lanes of
// a column agree on it with two shuffles; returned).  A column of the B operand scales the same column of the product, which is on the same lanes:
// the caller divides it out of the zero-initialised product that consumes the tile.
__device__ __forceinline__ float tile_split(const f32x4& t, ah4& hi, ah4& lo, float& inv) {
    const float m = col4_max(fmaxf(fmaxf(fabsf(t[0]), fabsf(t[1])), fmaxf(fabsf(t[2]), fabsf(t[3]))));
    const float sc = pow2_scale(m, inv);
    attn_split4(t[0] * sc, t[1] * sc, t[2] * sc, t[3] * sc, hi, lo);
    return sc;
}

#ifndef AH_FWD_NQ
#define AH_FWD_NQ 2                  // query tiles a wave of the forward kernel runs side by side (1 = the round-3 loop)
#endif
__global__ __launch_bounds__(256) void dts_attn_fwd_h_kernel(const AttnArgs p) {
    constexpr int HD = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int Lkp = (p.Lk + 15) & ~15;
    _Float16* Kh = reinterpret_cast<_Float16*>(ldsb);      // [Lkp][AH_RS]
    _Float16* Kl = Kh + Lkp * AH_RS;
    _Float16* VTh = Kl + Lkp * AH_RS;                      // [16][Lkp]
    _Float16* VTl = VTh + 16 * Lkp;
    float* red = reinterpret_cast<float*>(VTl + 16 * Lkp); // [4][2]
    const float* Qb = p.Q + (long long)b * p.Lq * p.ldq + h * p.hs;
    HeadRegs kr, vr;
    float mx[2], sc[2];
    mx[0] = head_load(p.K + (long long)b * p.Lk * p.ldk + h * p.hs, p.ldk, p.Lk, Lkp, kr);
    mx[1] = head_load(p.V + (long long)b * p.Lk * p.ldv + h * p.hs, p.ldv, p.Lk, Lkp, vr);
    wg_scales<2>(mx, red, sc);
    head_store(kr, sc[0], Lkp, Kh, Kl, nullptr, nullptr);
    head_store(vr, sc[1], Lkp, nullptr, nullptr, VTh, VTl);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const float c0 = p.scale * LOG2E / sc[0];
    const float oinv = 1.0f / sc[1];                                   // O = (2^13 sV O) / (sV 2^13 l)
    const int nkt = Lkp >> 4, tail = (p.Lk & 15) ? nkt - 1 : nkt;      // key tile with padded keys (only that one is masked)
    // 13 query tiles over 4 waves = 4 + 3 + 3 + 3: the wave with the extra tile rotates from workgroup to workgroup, or SIMD 0 (wave 0 of every
    // resident workgroup) carries a third more than the others.
    // Round 4: a wave runs TWO of its query tiles (jt, jt + 4) through the key tiles together.  A (query tile, key tile) pair is one serial chain --
    // LDS read -> 3 dependent MFMAs -> maximum across lanes -> exp2 -> split -> 3 dependent MFMAs, ~1600 cycles end to end -- and with the five waves per
    // SIMD the LDS budget allows the kernel ran at 320 cycles per pair: latency-bound (the issue limit for its 64 instructions is ~130).  Two independent
    // chains per wave share the K / V fragment reads and fill each other's stalls.
    const int nqt = (p.Lq + 15) >> 4;
    const _Float16* kp = Kh + r16 * AH_RS + 4 * g;
    const _Float16* vp = VTh + r16 * Lkp + 4 * g;
    const int klo = Lkp * AH_RS, vlo = 16 * Lkp;
    for (int jt0 = (wave + blockIdx.x) & 3; jt0 < nqt; jt0 += 4 * AH_FWD_NQ) {
        int j[AH_FWD_NQ];
        bool jl[AH_FWD_NQ];
        ah4 qh[AH_FWD_NQ], ql[AH_FWD_NQ];
        float c[AH_FWD_NQ], mr[AH_FWD_NQ], l[AH_FWD_NQ];
        f32x4 o[AH_FWD_NQ];
#pragma unroll
        for (int u = 0; u < AH_FWD_NQ; ++u) {
            j[u] = 16 * (jt0 + 4 * u) + r16;
            jl[u] = j[u] < p.Lq;                                       // (a tile beyond the last one has no valid query at all)
            c[u] = c0 / load_split_col(Qb + (long long)(jl[u] ? j[u] : 0) * p.ldq + 4 * g, qh[u], ql[u]);      // the scores come out of the MFMA times sK sQ(query)
            mr[u] = -INFINITY;
            l[u] = 0.0f;
            o[u] = f32x4{0, 0, 0, 0};
        }
        // online softmax on the RAW scores (c > 0): running maximum mr of s, probabilities 2^13 exp2(c (s - mr)) as one fused multiply-add per score
        auto key_tile = [&](int it, auto masked, auto nact) {
            constexpr int NA = decltype(nact)::value;
            const ah4 kh = *reinterpret_cast<const ah4*>(kp + 16 * AH_RS * it);
            const ah4 kl = *reinterpret_cast<const ah4*>(kp + klo + 16 * AH_RS * it);
            const ah4 vh = *reinterpret_cast<const ah4*>(vp + 16 * it);
            const ah4 vl = *reinterpret_cast<const ah4*>(vp + vlo + 16 * it);
            f32x4 s[NA];
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                s[u] = f32x4{0, 0, 0, 0};
                DTS_MMH3(s[u], kh, kl, qh[u], ql[u]);                   // sK sQ S^T[key 4 g + r][query r16]
            }
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                if constexpr (decltype(masked)::value) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * it + 4 * g + r >= p.Lk) s[u][r] = -INFINITY;
                }
                const float tmax = col4_max(fmaxf(fmaxf(s[u][0], s[u][1]), fmaxf(s[u][2], s[u][3])));
                const float mn = fmaxf(mr[u], tmax);                   // finite from the first tile on (Lk >= 1)
                const float alpha = __builtin_amdgcn_exp2f((mr[u] - mn) * c[u]);
                const float nb = __builtin_fmaf(-mn, c[u], AH_PLOG);   // probabilities come out times 2^13 (the split's scale); l carries the same factor
#pragma unroll
                for (int r = 0; r < 4; ++r) s[u][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], c[u], nb));
                const float ps = (s[u][0] + s[u][1]) + (s[u][2] + s[u][3]);
                l[u] = l[u] * alpha + ps;
                o[u] *= alpha;
                mr[u] = mn;
            }
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                ah4 ph, pl;
                attn_split4(s[u][0], s[u][1], s[u][2], s[u][3], ph, pl);
                DTS_MMH3(o[u], vh, vl, ph, pl);                         // 2^13 sV O^T[d 4 g + r][query r16]
            }
        };
        auto run = [&](auto nact) {
            for (int it = 0; it < tail; ++it) key_tile(it, std::false_type{}, nact);
            if (tail < nkt) key_tile(tail, std::true_type{}, nact);
        };
        int nact = 0;
#pragma unroll
        for (int u = 0; u < AH_FWD_NQ; ++u) nact += (jt0 + 4 * u < nqt) ? 1 : 0;          // wave-uniform
        if (AH_FWD_NQ >= 4 && nact == 4) run(std::integral_constant<int, (AH_FWD_NQ >= 4 ? 4 : 1)>{});
        else if (AH_FWD_NQ >= 3 && nact == 3) run(std::integral_constant<int, (AH_FWD_NQ >= 3 ? 3 : 1)>{});
        else if (AH_FWD_NQ >= 2 && nact == 2) run(std::integral_constant<int, (AH_FWD_NQ >= 2 ? 2 : 1)>{});
        else run(std::integral_constant<int, 1>{});
#pragma unroll
        for (int u = 0; u < AH_FWD_NQ; ++u) {
            if (u >= nact) break;
            const float lt = col4_sum(l[u]);
            if (jl[u]) {
                const float inv = oinv / lt;
                float* dst = p.out + ((long long)b * p.Lq + j[u]) * p.ldout + h * p.hs + 4 * g;
                *reinterpret_cast<f32x4*>(dst) = o[u] * inv;
                if (g == 0 && p.lse) p.lse[((long long)b * p.H + h) * p.Lq + j[u]] = __builtin_fmaf(mr[u], c[u], __builtin_amdgcn_logf(lt) - AH_PLOG);
            }
        }
    }
}

// tiles a wave runs side by side (1 or 2).  Measured, same box, 10-step c3 plan (profiles/r04/c3_attn_ilp.txt): forward 280.5 -> 265.6 us with two
// query tiles (three: 263.1, four: 266-269); dK-dV 493.8 -> 476.6 with two key tiles (112 VGPRs: four workgroups per CU instead of five, still ahead);
// dQ 386.1 -> 397.2 with two (97 VGPRs cost it its fifth workgroup for nothing): one.
#ifndef AH_DQ_NQ
#define AH_DQ_NQ 1
#endif
#ifndef AH_DKV_NQ
#define AH_DKV_NQ 2
#endif
__global__ __launch_bounds__(256) void dts_attn_dq_h_kernel(const AttnArgs p) {
    constexpr int HD = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int Lkp = (p.Lk + 15) & ~15;
    _Float16* Kh = reinterpret_cast<_Float16*>(ldsb);      // [Lkp][AH_RS]
    _Float16* Kl = Kh + Lkp * AH_RS;
    _Float16* Vh = Kl + Lkp * AH_RS;
    _Float16* Vl = Vh + Lkp * AH_RS;
    float* red = reinterpret_cast<float*>(Vl + Lkp * AH_RS);      // [4][2]
    const float* Qb = p.Q + (long long)b * p.Lq * p.ldq + h * p.hs;
    const float* Db = p.dO + (long long)b * p.Lq * p.ldo + h * p.hs;
    HeadRegs kr, vr;
    float mx[2], sc[2];
    mx[0] = head_load(p.K + (long long)b * p.Lk * p.ldk + h * p.hs, p.ldk, p.Lk, Lkp, kr);
    mx[1] = head_load(p.V + (long long)b * p.Lk * p.ldv + h * p.hs, p.ldv, p.Lk, Lkp, vr);
    wg_scales<2>(mx, red, sc);
    head_store(kr, sc[0], Lkp, Kh, Kl, nullptr, nullptr);
    head_store(vr, sc[1], Lkp, Vh, Vl, nullptr, nullptr);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const float c0 = p.scale * LOG2E / sc[0];
    const float kinv = 1.0f / sc[0];
    const int nqt = (p.Lq + 15) >> 4;
    const int nkt = Lkp >> 4, tail = (p.Lk & 15) ? nkt - 1 : nkt;      // key tile with padded keys (only that one is masked)
    const int plane = Lkp * AH_RS;                                     // Kh -> Kl -> Vh -> Vl are consecutive planes
    const _Float16* kp = Kh + r16 * AH_RS + 4 * g;
    const _Float16* kc = Kh + (4 * g + (r16 >> 2)) * AH_RS + 4 * (r16 & 3);      // transposed reads of the key block 4 g .. 4 g + 3: this lane's address
    for (int jt0 = (wave + blockIdx.x) & 3; jt0 < nqt; jt0 += 4 * AH_DQ_NQ) {      // (rotated, and AH_DQ_NQ query tiles side by side: see the forward kernel)
        int j[AH_DQ_NQ];
        bool jl[AH_DQ_NQ];
        ah4 qh[AH_DQ_NQ], ql[AH_DQ_NQ], dh[AH_DQ_NQ], dl[AH_DQ_NQ];
        float c[AH_DQ_NQ], del[AH_DQ_NQ], dpinv[AH_DQ_NQ], lse[AH_DQ_NQ];
        f32x4 dq[AH_DQ_NQ];
#pragma unroll
        for (int u = 0; u < AH_DQ_NQ; ++u) {
            j[u] = 16 * (jt0 + 4 * u) + r16;
            jl[u] = j[u] < p.Lq;
            c[u] = c0 / load_split_col(Qb + (long long)(jl[u] ? j[u] : 0) * p.ldq + 4 * g, qh[u], ql[u]);
            // delta[query] = dO[query] . O[query] is formed here, from the dO row the kernel loads anyway and the matching O row (four lanes per query, two
            // shuffles), and left in p.delta for the dK-dV pass: the separate pass over dO and O (dts_attn_delta_kernel) is gone in this mode
            const f32x4 dov = *reinterpret_cast<const f32x4*>(Db + (long long)(jl[u] ? j[u] : 0) * p.ldo + 4 * g);
            const f32x4 ov = *reinterpret_cast<const f32x4*>(p.O + ((long long)b * p.Lq + (jl[u] ? j[u] : 0)) * p.ldo + h * p.hs + 4 * g);
            del[u] = col4_sum((dov[0] * ov[0] + dov[1] * ov[1]) + (dov[2] * ov[2] + dov[3] * ov[3]));
            if (jl[u] && g == 0) const_cast<float*>(p.delta)[((long long)b * p.H + h) * p.Lq + j[u]] = del[u];
            dpinv[u] = 1.0f / (sc[1] * split_col(dov, dh[u], dl[u]));      // V . dO comes out times sV sdO(query)
            lse[u] = p.lse[((long long)b * p.H + h) * p.Lq + (jl[u] ? j[u] : 0)];
            dq[u] = f32x4{0, 0, 0, 0};
        }
        auto key_tile = [&](int it, auto masked, auto nact) {
            constexpr int NA = decltype(nact)::value;
            const _Float16* kr = kp + 16 * AH_RS * it;
            const ah4 kh = *reinterpret_cast<const ah4*>(kr), kl = *reinterpret_cast<const ah4*>(kr + plane);
            const ah4 vh = *reinterpret_cast<const ah4*>(kr + 2 * plane), vl = *reinterpret_cast<const ah4*>(kr + 3 * plane);
            const _Float16* kq = kc + 16 * AH_RS * it;
            const ah4 th = column4_tr(kq), tl = column4_tr(kq + plane);                                    // sK K^T[d = r16][keys 4 g ..]
            f32x4 s[NA], t2[NA];
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                s[u] = f32x4{0, 0, 0, 0};
                t2[u] = f32x4{0, 0, 0, 0};
                DTS_MMH3(s[u], kh, kl, qh[u], ql[u]);
                DTS_MMH3(t2[u], vh, vl, dh[u], dl[u]);
            }
#pragma unroll
            for (int u = 0; u < NA; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(s[u][r] * c[u] - lse[u]);
                    s[u][r] = pr * (t2[u][r] * dpinv[u] - del[u]) * p.scale;
                }
                if constexpr (decltype(masked)::value) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * it + 4 * g + r >= p.Lk) s[u][r] = 0.0f;
                }
                ah4 sh, sl;
                float issc;
                (void)tile_split(s[u], sh, sl, issc);                   // dS^T of this tile times its own power of two
                f32x4 part = {0, 0, 0, 0};
                DTS_MMH3(part, th, tl, sh, sl);
                dq[u] += part * (kinv * issc);
            }
        };
        auto run = [&](auto nact) {
            for (int it = 0; it < tail; ++it) key_tile(it, std::false_type{}, nact);
            if (tail < nkt) key_tile(tail, std::true_type{}, nact);
        };
        int nact = 0;
#pragma unroll
        for (int u = 0; u < AH_DQ_NQ; ++u) nact += (jt0 + 4 * u < nqt) ? 1 : 0;          // wave-uniform
        if (AH_DQ_NQ >= 2 && nact == 2) run(std::integral_constant<int, (AH_DQ_NQ >= 2 ? 2 : 1)>{});
        else run(std::integral_constant<int, 1>{});
#pragma unroll
        for (int u = 0; u < AH_DQ_NQ; ++u)
            if (u < nact && jl[u]) *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Lq + j[u]) * p.ldout + h * p.hs + 4 * g) = dq[u];
    }
}

__global__ __launch_bounds__(256) void dts_attn_dkv_h_kernel(const AttnArgs p) {
    constexpr int HD = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int Lqp = (p.Lq + 15) & ~15;
    _Float16* Qh = reinterpret_cast<_Float16*>(ldsb);      // [Lqp][AH_RS]
    _Float16* Ql = Qh + Lqp * AH_RS;
    _Float16* Dh = Ql + Lqp * AH_RS;                       // dO
    _Float16* Dl = Dh + Lqp * AH_RS;
    float* lsel = reinterpret_cast<float*>(Dl + Lqp * AH_RS);    // [Lqp]
    float* dell = lsel + Lqp;                              // [Lqp]
    float* red = dell + Lqp;                               // [4][2]
    const float* Kb = p.K + (long long)b * p.Lk * p.ldk + h * p.hs;
    const float* Vb = p.V + (long long)b * p.Lk * p.ldv + h * p.hs;
    HeadRegs qr, dr;
    float mx[2], sc[2];
    mx[0] = head_load(p.Q + (long long)b * p.Lq * p.ldq + h * p.hs, p.ldq, p.Lq, Lqp, qr);
    mx[1] = head_load(p.dO + (long long)b * p.Lq * p.ldo + h * p.hs, p.ldo, p.Lq, Lqp, dr);
    wg_scales<2>(mx, red, sc);
    head_store(qr, sc[0], Lqp, Qh, Ql, nullptr, nullptr);
    head_store(dr, sc[1], Lqp, Dh, Dl, nullptr, nullptr);
    for (int i = threadIdx.x; i < Lqp; i += 256) {
        lsel[i] = i < p.Lq ? p.lse[((long long)b * p.H + h) * p.Lq + i] : INFINITY;     // padded queries: P = 0
        dell[i] = i < p.Lq ? p.delta[((long long)b * p.H + h) * p.Lq + i] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const float c0 = p.scale * LOG2E / sc[0];
    const float dvinv = 1.0f / (sc[1] * AH_PSCALE), qinv = 1.0f / sc[0], pscale = p.scale / AH_PSCALE;
    const int nktk = (p.Lk + 15) >> 4;
    const int plane = Lqp * AH_RS;                                     // Qh -> Ql -> Dh -> Dl are consecutive planes
    const _Float16* qp = Qh + r16 * AH_RS + 4 * g;                     // row reads: query 16 it + r16, head columns 4 g ..
    const _Float16* qc = Qh + (4 * g + (r16 >> 2)) * AH_RS + 4 * (r16 & 3);      // transposed reads of the query block 16 it + 4 g ..: this lane's address
    for (int jt0 = (wave + blockIdx.x) & 3; jt0 < nktk; jt0 += 4 * AH_DKV_NQ) {     // (rotated, AH_DKV_NQ key tiles side by side: see the forward kernel)
        int j[AH_DKV_NQ];
        bool jl[AH_DKV_NQ];
        ah4 kh[AH_DKV_NQ], kl[AH_DKV_NQ], vh[AH_DKV_NQ], vl[AH_DKV_NQ];
        float c[AH_DKV_NQ], dpinv[AH_DKV_NQ];
        f32x4 dk[AH_DKV_NQ], dv[AH_DKV_NQ];
#pragma unroll
        for (int u = 0; u < AH_DKV_NQ; ++u) {
            j[u] = 16 * (jt0 + 4 * u) + r16;                          // key
            jl[u] = j[u] < p.Lk;
            c[u] = c0 / load_split_col(Kb + (long long)(jl[u] ? j[u] : 0) * p.ldk + 4 * g, kh[u], kl[u]);
            dpinv[u] = 1.0f / (sc[1] * load_split_col(Vb + (long long)(jl[u] ? j[u] : 0) * p.ldv + 4 * g, vh[u], vl[u]));
            dk[u] = f32x4{0, 0, 0, 0};
            dv[u] = f32x4{0, 0, 0, 0};
        }
        auto run = [&](auto nact) {
            constexpr int NA = decltype(nact)::value;
            for (int it = 0; it < (Lqp >> 4); ++it) {
                const _Float16* qr = qp + 16 * AH_RS * it;
                const ah4 qh = *reinterpret_cast<const ah4*>(qr), ql = *reinterpret_cast<const ah4*>(qr + plane);
                const ah4 dh = *reinterpret_cast<const ah4*>(qr + 2 * plane), dl = *reinterpret_cast<const ah4*>(qr + 3 * plane);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsel + 16 * it + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dell + 16 * it + 4 * g);
                const _Float16* qq = qc + 16 * AH_RS * it;
                const ah4 qth = column4_tr(qq), qtl = column4_tr(qq + plane);
                const ah4 dth = column4_tr(qq + 2 * plane), dtl = column4_tr(qq + 3 * plane);                  // sdO dO^T[d = r16][queries 4 g ..]
                f32x4 s[NA], t2[NA];
#pragma unroll
                for (int u = 0; u < NA; ++u) {
                    s[u] = f32x4{0, 0, 0, 0};
                    t2[u] = f32x4{0, 0, 0, 0};
                    DTS_MMH3(s[u], qh, ql, kh[u], kl[u]);               // sQ sK S[query 4 g + r][key r16]
                    DTS_MMH3(t2[u], dh, dl, vh[u], vl[u]);
                }
#pragma unroll
                for (int u = 0; u < NA; ++u) {
                    f32x4 pr, ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pr[r] = __builtin_amdgcn_exp2f(s[u][r] * c[u] - (l4[r] - AH_PLOG));        // 2^13 P
                        ds[r] = pr[r] * (t2[u][r] * dpinv[u] - d4[r]) * pscale;
                    }
                    ah4 ph, pl, sh, sl;
                    attn_split4(pr[0], pr[1], pr[2], pr[3], ph, pl);
                    float issc;
                    (void)tile_split(ds, sh, sl, issc);
                    DTS_MMH3(dv[u], dth, dtl, ph, pl);
                    f32x4 part = {0, 0, 0, 0};
                    DTS_MMH3(part, qth, qtl, sh, sl);
                    dk[u] += part * (qinv * issc);
                }
            }
        };
        int nact = 0;
#pragma unroll
        for (int u = 0; u < AH_DKV_NQ; ++u) nact += (jt0 + 4 * u < nktk) ? 1 : 0;         // wave-uniform
        if (AH_DKV_NQ >= 2 && nact == 2) run(std::integral_constant<int, (AH_DKV_NQ >= 2 ? 2 : 1)>{});
        else run(std::integral_constant<int, 1>{});
#pragma unroll
        for (int u = 0; u < AH_DKV_NQ; ++u) {
            if (u < nact && jl[u]) {
                *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Lk + j[u]) * p.ldout + h * p.hs + 4 * g) = dk[u];
                *reinterpret_cast<f32x4*>(p.out2 + ((long long)b * p.Lk + j[u]) * p.ldout2 + h * p.hs + 4 * g) = dv[u] * dvinv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: dQ, dK and dV in ONE pass.  The two-pass backward (dts_attn_dq_h_kernel + dts_attn_dkv_h_kernel) forms S, P, dP and dS of every
// (query tile, key tile) pair twice -- 101 + 127 vector instructions per pair, and reads Q / K / V / dO / O twice.  This kernel is the dK-dV pass
// (a wave owns key tiles, loops over the query tiles; S, dS in layout [query 4 g + r][key r16]) that ALSO forms the pair's contribution to dQ:
// dQ^T[d][q] += sum_key K^T[d][key] dS[q][key] contracts over the key, which sits on the lanes of the dS tile, so the fp32 tile goes through a
// 16 x 16 per-wave LDS scratch (four ds_write_b32 + one ds_read_b128 per lane: lane (q r16, g) gets dS[q][keys 4 g ..] = the B operand) and is split
// there with ONE power of two per tile (the smallest of the tile's per-key column scales of the dK product: a 16-lane DPP row reduction on the
// exponent bits); A = K^T of the wave's key tile [d = r16][keys 4 g ..], loaded once per key tile with its own per-tile power of two.  The
// partial dQ^T tiles of a wave's key tiles wait in registers (AH_MAXT x 4: the query-tile loop is unrolled) and are added over the four key-tile
// residue classes in class order through LDS at the end -- a fixed order, so the result is bitwise repeatable and independent of the batch a
// sequence runs in.  delta = dO . O is formed while dO is staged.  About 160 instructions per pair instead of 228; one pass over the operands.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int AH_MAXT = 16;          // query tiles of a head tile (<= 256 tokens: attn()'s bound for the split-f16 kernels)
constexpr int AH_SCR = 20;           // floats per row of the transposition scratch: 80-byte rows, conflict-free b32 writes and b128 reads
constexpr int AH_XT = 4;             // query tiles per round of the final dQ exchange ([AH_XT][3][4][64] floats of LDS)

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
// min / max over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1) of POSITIVE floats, on their bit patterns (no NaN canonicalisation)
__device__ __forceinline__ float row16_min_pos(float v) {
    int x = __float_as_int(v);
    x = min(x, dpp_i<0x128>(x)); x = min(x, dpp_i<0x124>(x)); x = min(x, dpp_i<0x122>(x)); x = min(x, dpp_i<0x121>(x));
    return __int_as_float(x);
}
__device__ __forceinline__ float row16_max_pos(float v) {
    int x = __float_as_int(v);
    x = max(x, dpp_i<0x128>(x)); x = max(x, dpp_i<0x124>(x)); x = max(x, dpp_i<0x122>(x)); x = max(x, dpp_i<0x121>(x));
    return __int_as_float(x);
}

inline size_t attn_bwd_h_lds(int Lq) {
    const size_t Lqp = (size_t)((Lq + 15) & ~15);
    const size_t main_bytes = sizeof(_Float16) * 4 * Lqp * AH_RS + sizeof(float) * (2 * Lqp + 8 + 4 * 2 * 16 * AH_SCR);
    const size_t xchg_bytes = sizeof(float) * AH_XT * 3 * 4 * 64;
    return (main_bytes > xchg_bytes ? main_bytes : xchg_bytes) + 64;
}

#ifndef DTS_BWD_WAVES
#define DTS_BWD_WAVES 2              // waves per SIMD the register allocator is held to (2: <= 256 VGPRs, no spills; 3: <= 168 VGPRs, 61 spilled at NT = 13)
#endif
// NT: compile-time bound on the query tiles (the dQ partials dq[NT] must be statically indexed: the query-tile loop is unrolled)
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DTS_BWD_WAVES))) void dts_attn_bwd_h_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int Lqp = (p.Lq + 15) & ~15;
    _Float16* Qh = reinterpret_cast<_Float16*>(ldsb);      // [Lqp][AH_RS]
    _Float16* Ql = Qh + Lqp * AH_RS;
    _Float16* Dh = Ql + Lqp * AH_RS;                       // dO
    _Float16* Dl = Dh + Lqp * AH_RS;
    float* lsel = reinterpret_cast<float*>(Dl + Lqp * AH_RS);    // [Lqp]
    float* dell = lsel + Lqp;                              // [Lqp]
    float* red = dell + Lqp;                               // [4][2]
    float* scr = red + 8;                                  // [4 waves][2][16][AH_SCR]
    const float* Kb = p.K + (long long)b * p.Lk * p.ldk + h * p.hs;
    const float* Vb = p.V + (long long)b * p.Lk * p.ldv + h * p.hs;
    float mx[2], sc[2];
    {
        HeadRegs qr, dr;
        mx[0] = head_load(p.Q + (long long)b * p.Lq * p.ldq + h * p.hs, p.ldq, p.Lq, Lqp, qr);
        mx[1] = head_load(p.dO + (long long)b * p.Lq * p.ldo + h * p.hs, p.ldo, p.Lq, Lqp, dr);
        // delta[q] = dO[q] . O[q]: the four threads of a row hold its four column quads (the association of the two-pass kernels: bitwise equal)
#pragma unroll
        for (int u = 0; u < AH_MAXQ; ++u) {
            const int i = threadIdx.x + 256 * u, r = i >> 2, d = 4 * (i & 3);
            float part = 0.0f;
            if (i < Lqp * 4 && r < p.Lq) {
                const f32x4 ov = *reinterpret_cast<const f32x4*>(p.O + ((long long)b * p.Lq + r) * p.ldo + h * p.hs + d);
                part = (dr.v[u][0] * ov[0] + dr.v[u][1] * ov[1]) + (dr.v[u][2] * ov[2] + dr.v[u][3] * ov[3]);
            }
            part += __shfl_xor(part, 1);
            part += __shfl_xor(part, 2);
            if (i < Lqp * 4 && (i & 3) == 0) dell[r] = part;
        }
        wg_scales<2>(mx, red, sc);
        head_store(qr, sc[0], Lqp, Qh, Ql, nullptr, nullptr);
        head_store(dr, sc[1], Lqp, Dh, Dl, nullptr, nullptr);
    }
    for (int i = threadIdx.x; i < Lqp; i += 256) lsel[i] = i < p.Lq ? p.lse[((long long)b * p.H + h) * p.Lq + i] : INFINITY;     // padded queries: P = 0
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    // key tiles by residue class: class vw = jt & 3 belongs to wave (vw - blockIdx) & 3 -- the class with the extra tile rotates over the SIMDs from
    // workgroup to workgroup, while sums are formed per CLASS, so the arithmetic does not depend on the rotation
    const int vw = (wave + blockIdx.x) & 3;
    const float c0 = p.scale * LOG2E / sc[0];
    const float dvinv = 1.0f / (sc[1] * AH_PSCALE), qinv = 1.0f / sc[0], pscale = p.scale / AH_PSCALE;
    const int nqt = Lqp >> 4, nkt = (p.Lk + 15) >> 4;
    float* myscr = scr + wave * 2 * 16 * AH_SCR;
    float* scr_w = myscr + (4 * g) * AH_SCR + r16;                     // this lane's first write slot: dS[query 4 g + r][key r16]
    const float* scr_r = myscr + r16 * AH_SCR + 4 * g;                 // ... and its read: dS[query r16][keys 4 g ..]
    f32x4 dq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) dq[t] = f32x4{0, 0, 0, 0};
    for (int jt = vw; jt < nkt; jt += 4) {
        const int j = 16 * jt + r16;                                  // key
        const bool jl = j < p.Lk;
        ah4 kh, kl, vh, vl;
        const float c = c0 / load_split_col(Kb + (long long)(jl ? j : 0) * p.ldk + 4 * g, kh, kl);
        const float dpinv = 1.0f / (sc[1] * load_split_col(Vb + (long long)(jl ? j : 0) * p.ldv + 4 * g, vh, vl));
        // K^T of the tile, [d = r16][keys 16 jt + 4 g ..], one power of two for the tile
        float kt[4], km = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 16 * jt + 4 * g + e;
            kt[e] = key < p.Lk ? Kb[(long long)key * p.ldk + r16] : 0.0f;
            km = fmaxf(km, fabsf(kt[e]));
        }
        float ktinv;
        const float ksc = pow2_scale(wave_max(km), ktinv);
        ah4 kth, ktl;
        attn_split4(kt[0] * ksc, kt[1] * ksc, kt[2] * ksc, kt[3] * ksc, kth, ktl);
        f32x4 dk = {0, 0, 0, 0}, dv = {0, 0, 0, 0};
        const int plane = Lqp * AH_RS;                                 // Qh -> Ql -> Dh -> Dl are consecutive planes
        const _Float16* qp = Qh + r16 * AH_RS + 4 * g;                 // row reads: query 16 it + r16, head columns 4 g ..
        const _Float16* qc = Qh + (4 * g + (r16 >> 2)) * AH_RS + 4 * (r16 & 3);      // transposed reads of the query block 16 it + 4 g ..: this lane's address
        // The dQ product of a pair runs ONE query tile behind the rest: the transposed tile is read back from the scratch at the top of the next step,
        // with that step's other LDS reads, so the write -> read round trip is off the dependency chain (two scratch tiles per wave, alternating).
        float tsc_prev = 1.0f, tinv_prev = 1.0f;
        auto dq_step = [&](const f32x4& dst, f32x4& acc) {
            ah4 th, tl;
            attn_split4(dst[0] * tsc_prev, dst[1] * tsc_prev, dst[2] * tsc_prev, dst[3] * tsc_prev, th, tl);
            f32x4 pq = {0, 0, 0, 0};
            DTS_MMH3(pq, kth, ktl, th, tl);                             // sKt sT dQ^T[d 4 g + r][query r16]
            acc += pq * (ktinv * tinv_prev);
        };
#pragma unroll
        for (int it = 0; it <= NT; ++it) {
            if (it < NT && it < nqt) {
                const _Float16* qrow = qp + 16 * AH_RS * it;
                const ah4 qh = *reinterpret_cast<const ah4*>(qrow), ql = *reinterpret_cast<const ah4*>(qrow + plane);
                const ah4 dh = *reinterpret_cast<const ah4*>(qrow + 2 * plane), dl = *reinterpret_cast<const ah4*>(qrow + 3 * plane);
                f32x4 dprev = {0, 0, 0, 0};
                if (it > 0) dprev = *reinterpret_cast<const f32x4*>(scr_r + ((it - 1) & 1) * 16 * AH_SCR);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsel + 16 * it + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dell + 16 * it + 4 * g);
                const _Float16* qq = qc + 16 * AH_RS * it;
                const ah4 qth = column4_tr(qq), qtl = column4_tr(qq + plane);
                const ah4 dth = column4_tr(qq + 2 * plane), dtl = column4_tr(qq + 3 * plane);                  // sdO dO^T[d = r16][queries 4 g ..]
                f32x4 s = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
                DTS_MMH3(s, qh, ql, kh, kl);                                // sQ sK S[query 4 g + r][key r16]
                DTS_MMH3(t2, dh, dl, vh, vl);
                f32x4 pr, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pr[r] = __builtin_amdgcn_exp2f(s[r] * c - (l4[r] - AH_PLOG));        // 2^13 P
                    ds[r] = pr[r] * (t2[r] * dpinv - d4[r]) * pscale;
                }
                if (!jl) ds = f32x4{0, 0, 0, 0};                            // a padded key column (its dK / dV are never stored) must not reach dQ
                float* sw = scr_w + (it & 1) * 16 * AH_SCR;
#pragma unroll
                for (int r = 0; r < 4; ++r) sw[r * AH_SCR] = ds[r];
                ah4 ph, pl, sh, sl;
                attn_split4(pr[0], pr[1], pr[2], pr[3], ph, pl);
                float issc;
                const float csc = tile_split(ds, sh, sl, issc);
                DTS_MMH3(dv, dth, dtl, ph, pl);
                f32x4 part = {0, 0, 0, 0};
                DTS_MMH3(part, qth, qtl, sh, sl);
                dk += part * (qinv * issc);
                if (it > 0) dq_step(dprev, dq[it > 0 ? it - 1 : 0]);
                tsc_prev = row16_min_pos(csc);                              // one power of two for the transposed tile: the smallest column scale
                tinv_prev = row16_max_pos(issc);
            } else if (it > 0 && it == nqt) {                               // the last tile's dQ product
                const f32x4 dprev = *reinterpret_cast<const f32x4*>(scr_r + ((it - 1) & 1) * 16 * AH_SCR);
                dq_step(dprev, dq[it > 0 ? it - 1 : 0]);
            }
        }
        if (jl) {
            *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Lk + j) * p.ldout + h * p.hs + 4 * g) = dk;
            *reinterpret_cast<f32x4*>(p.out2 + ((long long)b * p.Lk + j) * p.ldout2 + h * p.hs + 4 * g) = dv * dvinv;
        }
    }
    // dQ tile t = sum over the key-tile classes 0..3 in that order; class t & 3 collects it
    __syncthreads();
    float* xb = reinterpret_cast<float*>(ldsb);                            // [AH_XT][3][4][64]
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += AH_XT) {
        if (t0 < nqt) {
#pragma unroll
            for (int tt = 0; tt < AH_XT; ++tt) {
                const int t = t0 + tt, owner = t & 3;
                if (t < NT && t < nqt && vw != owner) {
                    float* dst = xb + ((tt * 3 + ((vw - owner - 1) & 3)) * 4) * 64 + lane;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r * 64] = dq[t < NT ? t : 0][r];
                }
            }
            __syncthreads();
#pragma unroll
            for (int tt = 0; tt < AH_XT; ++tt) {
                const int t = t0 + tt, owner = t & 3;
                if (t < NT && t < nqt && vw == owner) {
                    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if (w == owner) {
                            acc += dq[t < NT ? t : 0];
                        } else {
                            const float* src = xb + ((tt * 3 + ((w - owner - 1) & 3)) * 4) * 64 + lane;
                            acc += f32x4{src[0], src[64], src[128], src[192]};
                        }
                    }
                    const int q = 16 * t + r16;
                    if (q < p.Lq) *reinterpret_cast<f32x4*>(p.out3 + ((long long)b * p.Lq + q) * p.ldout3 + h * p.hs + 4 * g) = acc;
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The fused MLP of a block and its input-gradient mirror (dts_mlp_fwd_kernel / dts_mlp_bwd_kernel of dts_kernels.h) on the f16 matrix pipe,
// same structure: a wave owns 16-token tiles, the 16 M-wide hidden tile leaves the first product in accumulator layout [hidden unit 4 g + r][token]
// and IS the B operand of the second product (k = 4 g + j of v_mfma_f32_16x16x16_f16) after GELU (or gelu') and the split.  Weights: f16 hi / lo
// planes in LDS (split while they are staged, once per workgroup; the same 2 + 2 bytes per weight as fp32).  384 MFMAs of 16.5 cycles per tile
// against 512 of 32; the fp32 kernels were 65 % matrix-pipe bound.
// ---------------------------------------------------------------------------------------------------------------------
// weights [M][D] and [D][M] -> f16 hi / lo planes, each matrix times its own power of two (returned in sa, sb); red: LDS [4][2]
template <int DT, int MT>
__device__ __forceinline__ void mlp_stage_weights(const float* Wa /*[M][D]*/, const float* Wb /*[D][M]*/, _Float16* Ah, _Float16* Al, _Float16* Bh, _Float16* Bl,
                                                  float* red, float& sa, float& sb) {
    constexpr int D = 16 * DT, M = 16 * MT, KSH = D + 8, MSH = M + 8;
    float mx[2] = {0.0f, 0.0f}, sc[2];
    for (int i = threadIdx.x; i < M * (D / 4); i += 64 * MLP_WAVES) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(Wa + 4 * (size_t)i), b = *reinterpret_cast<const f32x4*>(Wb + 4 * (size_t)i);
        mx[0] = fmaxf(mx[0], fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
        mx[1] = fmaxf(mx[1], fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3]))));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        mx[k] = wave_max(mx[k]);
        if (lane == 0) red[wave * 2 + k] = mx[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float m = 0.0f;
        for (int w = 0; w < MLP_WAVES; ++w) m = fmaxf(m, red[w * 2 + k]);
        sc[k] = pow2_scale(m);
    }
    sa = sc[0];
    sb = sc[1];
    for (int i = threadIdx.x; i < M * (D / 4); i += 64 * MLP_WAVES) {
        const int n = i / (D / 4), c = i - n * (D / 4);
        ah4 hi, lo;
        load_split4(Wa + (size_t)n * D + 4 * c, sa, hi, lo);
        *reinterpret_cast<ah4*>(Ah + n * KSH + 4 * c) = hi;
        *reinterpret_cast<ah4*>(Al + n * KSH + 4 * c) = lo;
    }
    // Wb's columns are the k of the second product, whose B operand is the hidden tile in accumulator layout.  With 32-wide products a lane's eight k
    // are its accumulator quads of TWO neighbouring 16-blocks of hidden units: inside every 32-column group the quad (block w, rows 4 q ..) sits at 8 q + 4 w.
    constexpr bool K32 = (DT % 2 == 0) && (MT % 2 == 0);
    for (int i = threadIdx.x; i < D * (M / 4); i += 64 * MLP_WAVES) {
        const int n = i / (M / 4), c = i - n * (M / 4);
        ah4 hi, lo;
        load_split4(Wb + (size_t)n * M + 4 * c, sb, hi, lo);
        const int col = K32 ? 32 * (c >> 3) + 8 * (c & 3) + 4 * ((c >> 2) & 1) : 4 * c;
        *reinterpret_cast<ah4*>(Bh + n * MSH + col) = hi;
        *reinterpret_cast<ah4*>(Bl + n * MSH + col) = lo;
    }
}

// column of a token row that float4 number s of lane group g holds: 16-wide products: 16 s + 4 g; 32-wide: the lane's eight k of chunk s / 2 are
// 32 (s / 2) + 8 g .. + 7, as two float4
template <bool K32>
__device__ __forceinline__ int mlp_col(int s, int g) { return K32 ? 32 * (s >> 1) + 8 * g + 4 * (s & 1) : 16 * s + 4 * g; }

// power of two for the values a token's four lanes hold (n f32x4 each): the B-operand column of that token
template <int N>
__device__ __forceinline__ float token_scale(const f32x4 (&v)[N]) {
    float m = 0.0f;
#pragma unroll
    for (int s = 0; s < N; ++s) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[s][0]), fabsf(v[s][1])), fmaxf(fabsf(v[s][2]), fabsf(v[s][3]))));
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    return pow2_scale(m);
}

constexpr int MLP_HB = 4;           // hidden tiles formed, activated, split and consumed together (8 in the fp32 kernels; 4 keeps these under 256 VGPRs)

template <int DT, int MT>
__global__ __launch_bounds__(64 * MLP_WAVES) void dts_mlp_fwd_h_kernel(const MlpArgs p) {
    constexpr int D = 16 * DT, M = 16 * MT, KSH = D + 8, MSH = M + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    _Float16* W0h = reinterpret_cast<_Float16*>(ldsb);     // [M][KSH]
    _Float16* W0l = W0h + M * KSH;
    _Float16* W2h = W0l + M * KSH;                          // [D][MSH]
    _Float16* W2l = W2h + D * MSH;
    float* red = reinterpret_cast<float*>(W2l + D * MSH);  // [MLP_WAVES][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    float s0, s2;
    mlp_stage_weights<DT, MT>(p.W0, p.W2, W0h, W0l, W2h, W2l, red, s0, s2);
    __syncthreads();
    const float i0 = 1.0f / s0, i2 = 1.0f / s2;
    constexpr int HB = MT < MLP_HB ? MT : MLP_HB;
    constexpr bool K32 = (DT % 2 == 0) && (MT % 2 == 0) && (HB % 2 == 0);      // both products on v_mfma_f32_16x16x32_f16
    f32x4 lw[DT], lb[DT];
#pragma unroll
    for (int s = 0; s < DT; ++s) {
        lw[s] = *reinterpret_cast<const f32x4*>(p.lnw + mlp_col<K32>(s, g));
        lb[s] = *reinterpret_cast<const f32x4*>(p.lnb + mlp_col<K32>(s, g));
    }
    const long long n_tiles = (p.rows + 15) / 16;
    // one workgroup per CU (the weights fill its LDS), two waves per SIMD: the X tile of the wave's next iteration is requested before this one's products
    const long long tile0 = (long long)blockIdx.x * MLP_WAVES + wave, tstep = (long long)gridDim.x * MLP_WAVES;
    auto load_tile = [&](long long tile, f32x4 (&dst)[DT]) {
        const long long tok = tile * 16 + r16;
        const bool live = tile < n_tiles && tok < p.rows;
        const float* xr = p.X + (live ? tok : 0) * (long long)D;
#pragma unroll
        for (int s = 0; s < DT; ++s) dst[s] = live ? *reinterpret_cast<const f32x4*>(xr + mlp_col<K32>(s, g)) : f32x4{0, 0, 0, 0};
    };
    f32x4 xn[DT];
    load_tile(tile0, xn);
    for (long long tile = tile0; tile < n_tiles; tile += tstep) {
        asm volatile("" ::: "memory");        // the weight fragments are re-read from LDS every tile
        const long long tok = tile * 16 + r16;
        const bool live = tok < p.rows;
        f32x4 xb[DT];
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s < DT; ++s) {
            xb[s] = xn[s];
            sum += (xb[s][0] + xb[s][1]) + (xb[s][2] + xb[s][3]);
        }
        load_tile(tile + tstep, xn);
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
        const float sx = token_scale<DT>(xb);
        const float ix = i0 / sx;
        ah4 xh[DT], xl[DT];
#pragma unroll
        for (int s = 0; s < DT; ++s) attn_split4(xb[s][0] * sx, xb[s][1] * sx, xb[s][2] * sx, xb[s][3] * sx, xh[s], xl[s]);
        f32x4 out[DT];
#pragma unroll
        for (int mt = 0; mt < DT; ++mt) out[mt] = f32x4{0, 0, 0, 0};
#pragma unroll 1
        for (int hb = 0; hb < MT; hb += HB) {                        // (not unrolled: the compiler otherwise hoists the next blocks' weight fragments, 49 spilled VGPRs)
            f32x4 u[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int nt = hb + j;
                f32x4 acc = {0, 0, 0, 0};
                if constexpr (K32) {
                    const int wo = (16 * nt + r16) * KSH + 8 * g;
#pragma unroll
                    for (int s = 0; s < DT / 2; ++s) {
                        const ah8 ah = *reinterpret_cast<const ah8*>(W0h + wo + 32 * s), al = *reinterpret_cast<const ah8*>(W0l + wo + 32 * s);
                        DTS_MMH3_32(acc, ah, al, ah8_of(xh[2 * s], xh[2 * s + 1]), ah8_of(xl[2 * s], xl[2 * s + 1]));
                    }
                } else {
                    const int wo = (16 * nt + r16) * KSH + 4 * g;
#pragma unroll
                    for (int s = 0; s < DT; ++s) {
                        const ah4 ah = *reinterpret_cast<const ah4*>(W0h + wo + 16 * s), al = *reinterpret_cast<const ah4*>(W0l + wo + 16 * s);
                        DTS_MMH3(acc, ah, al, xh[s], xl[s]);
                    }
                }
                acc = acc * ix + *reinterpret_cast<const f32x4*>(p.b0 + 16 * nt + 4 * g);
                if (p.U && live) *reinterpret_cast<f32x4*>(p.U + tok * (long long)M + 16 * nt + 4 * g) = acc;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = gelu_f(acc[e]);
                u[j] = acc;
            }
            const float su = token_scale<HB>(u);
            ah4 uh[HB], ul[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) attn_split4(u[j][0] * su, u[j][1] * su, u[j][2] * su, u[j][3] * su, uh[j], ul[j]);
            const float iu = i2 / su;
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) {
                f32x4 part = {0, 0, 0, 0};
                if constexpr (K32) {
#pragma unroll
                    for (int j = 0; j < HB; j += 2) {
                        const int wo = (16 * mt + r16) * MSH + 16 * (hb + j) + 8 * g;
                        const ah8 ah = *reinterpret_cast<const ah8*>(W2h + wo), al = *reinterpret_cast<const ah8*>(W2l + wo);
                        DTS_MMH3_32(part, ah, al, ah8_of(uh[j], uh[j + 1]), ah8_of(ul[j], ul[j + 1]));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < HB; ++j) {
                        const int wo = (16 * mt + r16) * MSH + 16 * (hb + j) + 4 * g;
                        const ah4 ah = *reinterpret_cast<const ah4*>(W2h + wo), al = *reinterpret_cast<const ah4*>(W2l + wo);
                        DTS_MMH3(part, ah, al, uh[j], ul[j]);
                    }
                }
                out[mt] += part * iu;
            }
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

// Round 4: the same fused MLP with TWO 16-token tiles per wave iteration.  The one-tile kernel re-reads both weight matrices from LDS for every tile --
// 128 ds_read_b128 (128 KB) per 192 MFMAs -- and with eight waves on one LDS port that is 8192 LDS cycles against 6144 matrix-pipe cycles per SIMD: it
// is LDS-bound.  Two tiles share every weight fragment (half the LDS bytes per MFMA) and give the wave two independent chains; the hidden units are
// formed in blocks of HB2 = 2 tiles so that both tiles' blocks fit the register file.
constexpr int MLP_HB2 = 2;
template <int DT, int MT>
__global__ __launch_bounds__(64 * MLP_WAVES) void dts_mlp_fwd_h2_kernel(const MlpArgs p) {
    constexpr int D = 16 * DT, M = 16 * MT, KSH = D + 8, MSH = M + 8, NTK = 2;
    static_assert(DT % 2 == 0 && MT % 2 == 0, "the two-tile kernel is written for the 16x16x32 products");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    _Float16* W0h = reinterpret_cast<_Float16*>(ldsb);     // [M][KSH]
    _Float16* W0l = W0h + M * KSH;
    _Float16* W2h = W0l + M * KSH;                          // [D][MSH]
    _Float16* W2l = W2h + D * MSH;
    float* red = reinterpret_cast<float*>(W2l + D * MSH);  // [MLP_WAVES][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    float s0, s2;
    mlp_stage_weights<DT, MT>(p.W0, p.W2, W0h, W0l, W2h, W2l, red, s0, s2);
    __syncthreads();
    const float i0 = 1.0f / s0, i2 = 1.0f / s2;
    constexpr int HB = MLP_HB2;
    f32x4 lw[DT], lb[DT];
#pragma unroll
    for (int s = 0; s < DT; ++s) {
        lw[s] = *reinterpret_cast<const f32x4*>(p.lnw + mlp_col<true>(s, g));
        lb[s] = *reinterpret_cast<const f32x4*>(p.lnb + mlp_col<true>(s, g));
    }
    const long long n_tiles = (p.rows + 15) / 16, n_pairs = (n_tiles + NTK - 1) / NTK;
    const long long pair0 = (long long)blockIdx.x * MLP_WAVES + wave, pstep = (long long)gridDim.x * MLP_WAVES;
    for (long long pair = pair0; pair < n_pairs; pair += pstep) {
        asm volatile("" ::: "memory");        // the weight fragments are re-read from LDS every pair
        long long tok[NTK];
        bool live[NTK];
        ah4 xh[NTK][DT], xl[NTK][DT];
        float ix[NTK];
#pragma unroll
        for (int t = 0; t < NTK; ++t) {
            tok[t] = (pair * NTK + t) * 16 + r16;
            live[t] = tok[t] < p.rows;
            const float* xr = p.X + (live[t] ? tok[t] : 0) * (long long)D;
            f32x4 xb[DT];
            float sum = 0.0f;
#pragma unroll
            for (int s = 0; s < DT; ++s) {
                xb[s] = live[t] ? *reinterpret_cast<const f32x4*>(xr + mlp_col<true>(s, g)) : f32x4{0, 0, 0, 0};
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
            const float sx = token_scale<DT>(xb);
            ix[t] = i0 / sx;
#pragma unroll
            for (int s = 0; s < DT; ++s) attn_split4(xb[s][0] * sx, xb[s][1] * sx, xb[s][2] * sx, xb[s][3] * sx, xh[t][s], xl[t][s]);
        }
        f32x4 out[NTK][DT];
#pragma unroll
        for (int t = 0; t < NTK; ++t)
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) out[t][mt] = f32x4{0, 0, 0, 0};
#pragma unroll 1
        for (int hb = 0; hb < MT; hb += HB) {
            f32x4 u[NTK][HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int nt = hb + j;
                f32x4 acc[NTK];
#pragma unroll
                for (int t = 0; t < NTK; ++t) acc[t] = f32x4{0, 0, 0, 0};
                const int wo = (16 * nt + r16) * KSH + 8 * g;
#pragma unroll
                for (int s = 0; s < DT / 2; ++s) {
                    const ah8 ah = *reinterpret_cast<const ah8*>(W0h + wo + 32 * s), al = *reinterpret_cast<const ah8*>(W0l + wo + 32 * s);
#pragma unroll
                    for (int t = 0; t < NTK; ++t) DTS_MMH3_32(acc[t], ah, al, ah8_of(xh[t][2 * s], xh[t][2 * s + 1]), ah8_of(xl[t][2 * s], xl[t][2 * s + 1]));
                }
                const f32x4 b0v = *reinterpret_cast<const f32x4*>(p.b0 + 16 * nt + 4 * g);
#pragma unroll
                for (int t = 0; t < NTK; ++t) {
                    acc[t] = acc[t] * ix[t] + b0v;
                    if (p.U && live[t]) *reinterpret_cast<f32x4*>(p.U + tok[t] * (long long)M + 16 * nt + 4 * g) = acc[t];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] = gelu_f(acc[t][e]);
                    u[t][j] = acc[t];
                }
            }
            ah4 uh[NTK][HB], ul[NTK][HB];
            float iu[NTK];
#pragma unroll
            for (int t = 0; t < NTK; ++t) {
                const float su = token_scale<HB>(u[t]);
#pragma unroll
                for (int j = 0; j < HB; ++j) attn_split4(u[t][j][0] * su, u[t][j][1] * su, u[t][j][2] * su, u[t][j][3] * su, uh[t][j], ul[t][j]);
                iu[t] = i2 / su;
            }
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) {
#pragma unroll
                for (int j = 0; j < HB; j += 2) {
                    const int wo = (16 * mt + r16) * MSH + 16 * (hb + j) + 8 * g;
                    const ah8 ah = *reinterpret_cast<const ah8*>(W2h + wo), al = *reinterpret_cast<const ah8*>(W2l + wo);
#pragma unroll
                    for (int t = 0; t < NTK; ++t) {
                        f32x4 part = {0, 0, 0, 0};
                        DTS_MMH3_32(part, ah, al, ah8_of(uh[t][j], uh[t][j + 1]), ah8_of(ul[t][j], ul[t][j + 1]));
                        out[t][mt] += part * iu[t];
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NTK; ++t) {
            if (live[t]) {
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) {
                    const int n0 = 16 * mt + 4 * g;
                    f32x4 v = out[t][mt] + *reinterpret_cast<const f32x4*>(p.b2 + n0);
                    v += *reinterpret_cast<const f32x4*>(p.X + tok[t] * (long long)D + n0);
                    *reinterpret_cast<f32x4*>(p.Y + tok[t] * (long long)D + n0) = v;
                }
            }
        }
    }
}

template <int DT, int MT>
__global__ __launch_bounds__(64 * MLP_WAVES) void dts_mlp_bwd_h_kernel(const MlpBwdArgs p) {
    constexpr int D = 16 * DT, M = 16 * MT, KSH = D + 8, MSH = M + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    _Float16* W2h = reinterpret_cast<_Float16*>(ldsb);     // [M][KSH]   W2^T
    _Float16* W2l = W2h + M * KSH;
    _Float16* W0h = W2l + M * KSH;                          // [D][MSH]   W0^T
    _Float16* W0l = W0h + D * MSH;
    float* red = reinterpret_cast<float*>(W0l + D * MSH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    float s2, s0;
    mlp_stage_weights<DT, MT>(p.W2T, p.W0T, W2h, W2l, W0h, W0l, red, s2, s0);
    __syncthreads();
    const float i2 = 1.0f / s2, i0 = 1.0f / s0;
    const long long n_tiles = (p.rows + 15) / 16;
    constexpr int HB = MT < MLP_HB ? MT : MLP_HB;
    constexpr bool K32 = (DT % 2 == 0) && (MT % 2 == 0) && (HB % 2 == 0);      // both products on v_mfma_f32_16x16x32_f16
    // two waves per SIMD: the G tile of the wave's next iteration and the saved pre-activations of the next block of hidden units are requested a
    // phase ahead of their use
    const long long tile0 = (long long)blockIdx.x * MLP_WAVES + wave, tstep = (long long)gridDim.x * MLP_WAVES;
    auto load_g = [&](long long tile, f32x4 (&dst)[DT]) {
        const long long tok = tile * 16 + r16;
        const bool live = tile < n_tiles && tok < p.rows;
        const float* gr = p.G + (live ? tok : 0) * (long long)D;
#pragma unroll
        for (int s = 0; s < DT; ++s) dst[s] = live ? *reinterpret_cast<const f32x4*>(gr + mlp_col<K32>(s, g)) : f32x4{0, 0, 0, 0};
    };
    auto load_u = [&](long long tile, int hb, f32x4 (&dst)[HB]) {
        const long long tok = tile * 16 + r16;
        const bool live = tile < n_tiles && tok < p.rows && hb < MT;
#pragma unroll
        for (int j = 0; j < HB; ++j)
            dst[j] = live ? *reinterpret_cast<const f32x4*>(p.U + tok * (long long)M + 16 * (hb + j) + 4 * g) : f32x4{0, 0, 0, 0};
    };
    f32x4 gn[DT], un[HB];
    load_g(tile0, gn);
    load_u(tile0, 0, un);
    for (long long tile = tile0; tile < n_tiles; tile += tstep) {
        asm volatile("" ::: "memory");
        const long long tok = tile * 16 + r16;
        const bool live = tok < p.rows;
        f32x4 gb[DT];
#pragma unroll
        for (int s = 0; s < DT; ++s) gb[s] = gn[s];
        load_g(tile + tstep, gn);
        const float sg = token_scale<DT>(gb);                         // gradients carry the batch normaliser (1e-9 at full size): the token's own power of two
        const float ig = i2 / sg;
        ah4 gh[DT], gl[DT];
#pragma unroll
        for (int s = 0; s < DT; ++s) attn_split4(gb[s][0] * sg, gb[s][1] * sg, gb[s][2] * sg, gb[s][3] * sg, gh[s], gl[s]);
        f32x4 out[DT];
#pragma unroll
        for (int mt = 0; mt < DT; ++mt) out[mt] = f32x4{0, 0, 0, 0};
#pragma unroll 1
        for (int hb = 0; hb < MT; hb += HB) {
            f32x4 e[HB], uc[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) uc[j] = un[j];
            if (hb + HB < MT) load_u(tile, hb + HB, un);
            else load_u(tile + tstep, 0, un);
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int nt = hb + j;
                f32x4 acc = {0, 0, 0, 0};
                if constexpr (K32) {
                    const int wo = (16 * nt + r16) * KSH + 8 * g;
#pragma unroll
                    for (int s = 0; s < DT / 2; ++s) {
                        const ah8 ah = *reinterpret_cast<const ah8*>(W2h + wo + 32 * s), al = *reinterpret_cast<const ah8*>(W2l + wo + 32 * s);
                        DTS_MMH3_32(acc, ah, al, ah8_of(gh[2 * s], gh[2 * s + 1]), ah8_of(gl[2 * s], gl[2 * s + 1]));
                    }
                } else {
                    const int wo = (16 * nt + r16) * KSH + 4 * g;
#pragma unroll
                    for (int s = 0; s < DT; ++s) {
                        const ah4 ah = *reinterpret_cast<const ah4*>(W2h + wo + 16 * s), al = *reinterpret_cast<const ah4*>(W2l + wo + 16 * s);
                        DTS_MMH3(acc, ah, al, gh[s], gl[s]);
                    }
                }
                const f32x4 u = uc[j];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = acc[q] * ig * gelu_grad_f(u[q]);
                e[j] = acc;
            }
            const float se = token_scale<HB>(e);
            ah4 eh[HB], el[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) attn_split4(e[j][0] * se, e[j][1] * se, e[j][2] * se, e[j][3] * se, eh[j], el[j]);
            const float ie = i0 / se;
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) {
                f32x4 part = {0, 0, 0, 0};
                if constexpr (K32) {
#pragma unroll
                    for (int j = 0; j < HB; j += 2) {
                        const int wo = (16 * mt + r16) * MSH + 16 * (hb + j) + 8 * g;
                        const ah8 ah = *reinterpret_cast<const ah8*>(W0h + wo), al = *reinterpret_cast<const ah8*>(W0l + wo);
                        DTS_MMH3_32(part, ah, al, ah8_of(eh[j], eh[j + 1]), ah8_of(el[j], el[j + 1]));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < HB; ++j) {
                        const int wo = (16 * mt + r16) * MSH + 16 * (hb + j) + 4 * g;
                        const ah4 ah = *reinterpret_cast<const ah4*>(W0h + wo), al = *reinterpret_cast<const ah4*>(W0l + wo);
                        DTS_MMH3(part, ah, al, eh[j], el[j]);
                    }
                }
                out[mt] += part * ie;
            }
        }
        if (p.lnx) {                                                     // fused LayerNorm backward (one workgroup per CU: the registers are there)
            f32x4 lx[DT], old[DT];
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) {
                lx[mt] = live ? *reinterpret_cast<const f32x4*>(p.lnx + tok * (long long)D + 16 * mt + 4 * g) : f32x4{0, 0, 0, 0};
                old[mt] = live ? *reinterpret_cast<const f32x4*>(p.Y + tok * (long long)D + 16 * mt + 4 * g) : f32x4{0, 0, 0, 0};
            }
            float sum = 0.0f;
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) sum += (lx[mt][0] + lx[mt][1]) + (lx[mt][2] + lx[mt][3]);
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float mean = sum / (float)D;
            float q = 0.0f;
#pragma unroll
            for (int mt = 0; mt < DT; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lx[mt][e] -= mean;
                    q += lx[mt][e] * lx[mt][e];
                }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
            float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) {
                const f32x4 gw = *reinterpret_cast<const f32x4*>(p.lng + 16 * mt + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lx[mt][e] *= rstd;                                   // xhat
                    out[mt][e] *= gw[e];
                    m1 += out[mt][e];
                    m2 += out[mt][e] * lx[mt][e];
                }
            }
            m1 += __shfl_xor(m1, 16);
            m1 += __shfl_xor(m1, 32);
            m2 += __shfl_xor(m2, 16);
            m2 += __shfl_xor(m2, 32);
            m1 /= (float)D;
            m2 /= (float)D;
            if (live) {
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = old[mt][e] + rstd * (out[mt][e] - m1 - lx[mt][e] * m2);
                    *reinterpret_cast<f32x4*>(p.Y + tok * (long long)D + 16 * mt + 4 * g) = v;
                }
            }
        } else if (live) {
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) *reinterpret_cast<f32x4*>(p.Y + tok * (long long)D + 16 * mt + 4 * g) = out[mt];
        }
    }
}


// The input-gradient mirror with two 16-token tiles per wave iteration (see dts_mlp_fwd_h2_kernel): every W2^T / W0^T fragment read from LDS serves both tiles.
template <int DT, int MT>
__global__ __launch_bounds__(64 * MLP_WAVES) void dts_mlp_bwd_h2_kernel(const MlpBwdArgs p) {
    constexpr int D = 16 * DT, M = 16 * MT, KSH = D + 8, MSH = M + 8, NTK = 2, HB = MLP_HB2;
    static_assert(DT % 2 == 0 && MT % 2 == 0, "the two-tile kernel is written for the 16x16x32 products");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    _Float16* W2h = reinterpret_cast<_Float16*>(ldsb);     // [M][KSH]   W2^T
    _Float16* W2l = W2h + M * KSH;
    _Float16* W0h = W2l + M * KSH;                          // [D][MSH]   W0^T
    _Float16* W0l = W0h + D * MSH;
    float* red = reinterpret_cast<float*>(W0l + D * MSH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    float s2, s0;
    mlp_stage_weights<DT, MT>(p.W2T, p.W0T, W2h, W2l, W0h, W0l, red, s2, s0);
    __syncthreads();
    const float i2 = 1.0f / s2, i0 = 1.0f / s0;
    const long long n_tiles = (p.rows + 15) / 16, n_pairs = (n_tiles + NTK - 1) / NTK;
    const long long pair0 = (long long)blockIdx.x * MLP_WAVES + wave, pstep = (long long)gridDim.x * MLP_WAVES;
    for (long long pair = pair0; pair < n_pairs; pair += pstep) {
        asm volatile("" ::: "memory");
        long long tok[NTK];
        bool live[NTK];
        ah4 gh[NTK][DT], gl[NTK][DT];
        float ig[NTK];
#pragma unroll
        for (int t = 0; t < NTK; ++t) {
            tok[t] = (pair * NTK + t) * 16 + r16;
            live[t] = tok[t] < p.rows;
            const float* gr = p.G + (live[t] ? tok[t] : 0) * (long long)D;
            f32x4 gb[DT];
#pragma unroll
            for (int s = 0; s < DT; ++s) gb[s] = live[t] ? *reinterpret_cast<const f32x4*>(gr + mlp_col<true>(s, g)) : f32x4{0, 0, 0, 0};
            const float sg = token_scale<DT>(gb);                     // gradients carry the batch normaliser (1e-9 at full size): the token's own power of two
            ig[t] = i2 / sg;
#pragma unroll
            for (int s = 0; s < DT; ++s) attn_split4(gb[s][0] * sg, gb[s][1] * sg, gb[s][2] * sg, gb[s][3] * sg, gh[t][s], gl[t][s]);
        }
        f32x4 out[NTK][DT];
#pragma unroll
        for (int t = 0; t < NTK; ++t)
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) out[t][mt] = f32x4{0, 0, 0, 0};
#pragma unroll 1
        for (int hb = 0; hb < MT; hb += HB) {
            f32x4 uc[NTK][HB], e[NTK][HB];
#pragma unroll
            for (int t = 0; t < NTK; ++t)
#pragma unroll
                for (int j = 0; j < HB; ++j)
                    uc[t][j] = live[t] ? *reinterpret_cast<const f32x4*>(p.U + tok[t] * (long long)M + 16 * (hb + j) + 4 * g) : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int nt = hb + j;
                f32x4 acc[NTK];
#pragma unroll
                for (int t = 0; t < NTK; ++t) acc[t] = f32x4{0, 0, 0, 0};
                const int wo = (16 * nt + r16) * KSH + 8 * g;
#pragma unroll
                for (int s = 0; s < DT / 2; ++s) {
                    const ah8 ah = *reinterpret_cast<const ah8*>(W2h + wo + 32 * s), al = *reinterpret_cast<const ah8*>(W2l + wo + 32 * s);
#pragma unroll
                    for (int t = 0; t < NTK; ++t) DTS_MMH3_32(acc[t], ah, al, ah8_of(gh[t][2 * s], gh[t][2 * s + 1]), ah8_of(gl[t][2 * s], gl[t][2 * s + 1]));
                }
#pragma unroll
                for (int t = 0; t < NTK; ++t) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[t][q] = acc[t][q] * ig[t] * gelu_grad_f(uc[t][j][q]);
                    e[t][j] = acc[t];
                }
            }
            ah4 eh[NTK][HB], el[NTK][HB];
            float ie[NTK];
#pragma unroll
            for (int t = 0; t < NTK; ++t) {
                const float se = token_scale<HB>(e[t]);
#pragma unroll
                for (int j = 0; j < HB; ++j) attn_split4(e[t][j][0] * se, e[t][j][1] * se, e[t][j][2] * se, e[t][j][3] * se, eh[t][j], el[t][j]);
                ie[t] = i0 / se;
            }
#pragma unroll
            for (int mt = 0; mt < DT; ++mt) {
#pragma unroll
                for (int j = 0; j < HB; j += 2) {
                    const int wo = (16 * mt + r16) * MSH + 16 * (hb + j) + 8 * g;
                    const ah8 ah = *reinterpret_cast<const ah8*>(W0h + wo), al = *reinterpret_cast<const ah8*>(W0l + wo);
#pragma unroll
                    for (int t = 0; t < NTK; ++t) {
                        f32x4 part = {0, 0, 0, 0};
                        DTS_MMH3_32(part, ah, al, ah8_of(eh[t][j], eh[t][j + 1]), ah8_of(el[t][j], el[t][j + 1]));
                        out[t][mt] += part * ie[t];
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NTK; ++t) {
            if (p.lnx) {                                                 // fused LayerNorm backward, as in the one-tile kernel
                f32x4 lx[DT], old[DT];
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) {
                    lx[mt] = live[t] ? *reinterpret_cast<const f32x4*>(p.lnx + tok[t] * (long long)D + 16 * mt + 4 * g) : f32x4{0, 0, 0, 0};
                    old[mt] = live[t] ? *reinterpret_cast<const f32x4*>(p.Y + tok[t] * (long long)D + 16 * mt + 4 * g) : f32x4{0, 0, 0, 0};
                }
                float sum = 0.0f;
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) sum += (lx[mt][0] + lx[mt][1]) + (lx[mt][2] + lx[mt][3]);
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float mean = sum / (float)D;
                float q = 0.0f;
#pragma unroll
                for (int mt = 0; mt < DT; ++mt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        lx[mt][c] -= mean;
                        q += lx[mt][c] * lx[mt][c];
                    }
                q += __shfl_xor(q, 16);
                q += __shfl_xor(q, 32);
                const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
                float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) {
                    const f32x4 gw = *reinterpret_cast<const f32x4*>(p.lng + 16 * mt + 4 * g);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        lx[mt][c] *= rstd;                               // xhat
                        out[t][mt][c] *= gw[c];
                        m1 += out[t][mt][c];
                        m2 += out[t][mt][c] * lx[mt][c];
                    }
                }
                m1 += __shfl_xor(m1, 16);
                m1 += __shfl_xor(m1, 32);
                m2 += __shfl_xor(m2, 16);
                m2 += __shfl_xor(m2, 32);
                m1 /= (float)D;
                m2 /= (float)D;
                if (live[t]) {
#pragma unroll
                    for (int mt = 0; mt < DT; ++mt) {
                        f32x4 v;
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = old[mt][c] + rstd * (out[t][mt][c] - m1 - lx[mt][c] * m2);
                        *reinterpret_cast<f32x4*>(p.Y + tok[t] * (long long)D + 16 * mt + 4 * g) = v;
                    }
                }
            } else if (live[t]) {
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) *reinterpret_cast<f32x4*>(p.Y + tok[t] * (long long)D + 16 * mt + 4 * g) = out[t][mt];
            }
        }
    }
}



// ---------------------------------------------------------------------------------------------------------------------
// Token-wise Linear (dts_lin_wave_kernel of dts_kernels.h: same arguments, prologues, epilogues, persistent grid and load order) on the f16 matrix
// pipe: K a multiple of 32, X and W as power-of-two scaled two-term f16 splits (X per token, W per matrix while it is staged into LDS as hi / lo
// planes -- the same bytes as its fp32 image), three v_mfma_f32_16x16x32_f16 per 32 k.  A lane holds the eight consecutive columns 32 c + 8 g ..
// of its token (two float4 loads), which is the B operand as it stands.  The exact kernel's 16 x 16 x 4 fp32 products were 40 us of its 118 us per launch.
// ---------------------------------------------------------------------------------------------------------------------
// LNB: the fused LayerNorm-backward epilogue (LinArgs::lnx), a separate instance so that the plain one keeps its registers (124 VGPRs: two workgroups per CU)
template <int KT, bool LNB = false>
__global__ __launch_bounds__(64 * LINW_WAVES) void dts_lin_wave_h_kernel(const LinArgs p) {
    static_assert(KT % 2 == 0, "K must be a multiple of 32");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    constexpr int K = 16 * KT, KSH = K + 8;
    _Float16* Wh = reinterpret_cast<_Float16*>(ldsb);      // [N][KSH]
    _Float16* Wlo = Wh + p.N * KSH;
    float* red = reinterpret_cast<float*>(Wlo + p.N * KSH);  // [LINW_WAVES]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    float mx = 0.0f;
    for (int i = tid; i < p.N * (K / 4); i += 64 * LINW_WAVES) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(p.W + 4 * (size_t)i);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fmaxf(fabsf(w[2]), fabsf(w[3]))));
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = 0.0f;
    for (int w = 0; w < LINW_WAVES; ++w) mx = fmaxf(mx, red[w]);
    const float sw = pow2_scale(mx), iw = 1.0f / sw;
    for (int i = tid; i < p.N * (K / 4); i += 64 * LINW_WAVES) {
        const int n = i / (K / 4), c = i - n * (K / 4);
        ah4 hi, lo;
        load_split4(p.W + (size_t)n * K + 4 * c, sw, hi, lo);
        *reinterpret_cast<ah4*>(Wh + n * KSH + 4 * c) = hi;
        *reinterpret_cast<ah4*>(Wlo + n * KSH + 4 * c) = lo;
    }
    __syncthreads();
    const long long n_tiles = (p.rows + 15) / 16;
    const long long ys = p.yblk ? p.yblk : 16;
    // float4 number s of the lane: columns 32 (s / 2) + 8 g + 4 (s & 1) ..; in the 16-column blocked layout that is block 2 (s / 2) + (g >> 1)
    auto x_off = [&](int s) -> long long {
        const int col = 32 * (s >> 1) + 8 * g + 4 * (s & 1);
        return p.xblk ? (long long)(col >> 4) * p.xblk + (col & 15) : col;
    };
    constexpr bool AHEAD = KT <= 4;
    const long long tile0 = (long long)blockIdx.x * LINW_WAVES + wave, tstep = (long long)gridDim.x * LINW_WAVES;
    auto load_tile = [&](long long tile, f32x4 (&dst)[KT]) {
        const long long tok = tile * 16 + r16;
        const bool live = tile < n_tiles && tok < p.rows;
        const float* xr = p.X + (live ? tok : 0) * (long long)p.ldx;
#pragma unroll
        for (int s = 0; s < KT; ++s) dst[s] = live ? *reinterpret_cast<const f32x4*>(xr + x_off(s)) : f32x4{0, 0, 0, 0};
    };
    f32x4 xn[AHEAD ? KT : 1];
    if constexpr (AHEAD) load_tile(tile0, xn);
    for (long long tile = tile0; tile < n_tiles; tile += tstep) {
        asm volatile("" ::: "memory");              // weight fragments stay in LDS
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
                const int col = 32 * (s >> 1) + 8 * g + 4 * (s & 1);
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.P1 + col), b = *reinterpret_cast<const f32x4*>(p.P2 + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = xb[s][e] * rstd;
                    xb[s][e] = (p.pro == PRO_ADALN) ? xh * (1.0f + a[e]) + b[e] : xh * a[e] + b[e];
                }
            }
        } else if (p.pro == PRO_GELUGRAD) {
            const float* ur = p.P1 + (live ? tok : 0) * (long long)p.ldp;
#pragma unroll
            for (int s = 0; s < KT; ++s) {
                const f32x4 u = live ? *reinterpret_cast<const f32x4*>(ur + 32 * (s >> 1) + 8 * g + 4 * (s & 1)) : f32x4{0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 4; ++e) xb[s][e] *= gelu_grad_f(u[e]);
            }
        } else if (p.pro == PRO_GELU) {
#pragma unroll
            for (int s = 0; s < KT; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) xb[s][e] = gelu_f(xb[s][e]);
        }
        const float sx = token_scale<KT>(xb);
        const float ix = iw / sx;
        ah8 xh[KT / 2], xl[KT / 2];
#pragma unroll
        for (int c = 0; c < KT / 2; ++c) {
            ah4 h0, l0, h1, l1;
            attn_split4(xb[2 * c][0] * sx, xb[2 * c][1] * sx, xb[2 * c][2] * sx, xb[2 * c][3] * sx, h0, l0);
            attn_split4(xb[2 * c + 1][0] * sx, xb[2 * c + 1][1] * sx, xb[2 * c + 1][2] * sx, xb[2 * c + 1][3] * sx, h1, l1);
            xh[c] = ah8_of(h0, h1);
            xl[c] = ah8_of(l0, l1);
        }
        // output blocks four at a time: everything the four epilogues add (residual, accumulation target) is requested up front, ahead of the next
        // tile's X (see dts_lin_wave_kernel)
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
            f32x4 lx[4];                                                 // fused LayerNorm backward: the LayerNorm's input row, requested with the addends
            if constexpr (LNB) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    lx[i] = (live && i < NT) ? *reinterpret_cast<const f32x4*>(p.lnx + tok * (long long)p.N + 16 * i + 4 * g) : f32x4{0, 0, 0, 0};
            }
            if constexpr (AHEAD) {
                if (nt0 == 0) load_tile(tile + tstep, xn);
            }
            f32x4 res[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nt = nt0 + i;
                res[i] = f32x4{0, 0, 0, 0};
                if (nt >= NT) break;
                const int n0 = 16 * nt + 4 * g;
                f32x4 acc = {0, 0, 0, 0};
                const int wo = (16 * nt + r16) * KSH + 8 * g;
#pragma unroll
                for (int c = 0; c < KT / 2; ++c) {
                    const ah8 ah = *reinterpret_cast<const ah8*>(Wh + wo + 32 * c), al = *reinterpret_cast<const ah8*>(Wlo + wo + 32 * c);
                    DTS_MMH3_32(acc, ah, al, xh[c], xl[c]);
                }
                f32x4 v = acc * ix;
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n0);
                if constexpr (LNB) {
                    res[i] = v;                                          // finished below, once the token's whole row is there
                } else if (live) {
                    v += addend[i];
                    *reinterpret_cast<f32x4*>(p.Y + tok * (long long)p.ldy + ys * nt + 4 * g) = v;
                }
            }
            if constexpr (LNB) {                                         // (N <= 64: one pass of this loop holds the row; the launcher checks)
                const float inv_n = 1.0f / (float)p.N;
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) sum += (lx[i][0] + lx[i][1]) + (lx[i][2] + lx[i][3]);
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float mean = sum * inv_n;
                float q = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lx[i][e] = i < NT ? lx[i][e] - mean : 0.0f;
                        q += lx[i][e] * lx[i][e];
                    }
                q += __shfl_xor(q, 16);
                q += __shfl_xor(q, 32);
                const float rstd = 1.0f / sqrtf(q * inv_n + 1e-5f);
                float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i >= NT) break;
                    const f32x4 gw = *reinterpret_cast<const f32x4*>(p.lng + 16 * i + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lx[i][e] *= rstd;                                // xhat
                        res[i][e] *= p.ln_ada ? 1.0f + gw[e] : gw[e];
                        m1 += res[i][e];
                        m2 += res[i][e] * lx[i][e];
                    }
                }
                m1 += __shfl_xor(m1, 16);
                m1 += __shfl_xor(m1, 32);
                m2 += __shfl_xor(m2, 16);
                m2 += __shfl_xor(m2, 32);
                m1 *= inv_n;
                m2 *= inv_n;
                if (live) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i >= NT) break;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = rstd * (res[i][e] - m1 - lx[i][e] * m2);
                        v += addend[i];                                  // (accumulate: the old value of Y)
                        *reinterpret_cast<f32x4*>(p.Y + tok * (long long)p.ldy + ys * i + 4 * g) = v;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-sequence mixing along the token axis (dts_seqmix_kernel of dts_kernels.h: Y_b[m][d] = bias[m] + sum_k W[m][k] X_b[k][d]) on the f16 matrix pipe:
// W arrives pre-split (hi / lo f16 planes [Mp][Kp32], one power of two for the matrix, made at load), X_b is scaled by its own power of two and
// split while it is staged into LDS as row-major planes [Kp32][D + 8]; the B operand of a 32-wide product -- eight consecutive k (tokens) of one
// channel -- is two transposed LDS reads (ds_read_b64_tr_b16) per plane.  A wave owns every fourth 16-row block of W and keeps the X fragments of a
// k-chunk in registers across its blocks.  The exact kernel was bound by its 16 x 16 x 4 fp32 products (135 us of matrix pipe in 286 us per launch).
// ---------------------------------------------------------------------------------------------------------------------
struct SeqmixHArgs {
    const _Float16* Wh; const _Float16* Wl; int Mp, Kp32; float isw;     // planes [Mp][Kp32]; isw = 1 / (the matrix's power of two)
    const float* bias;              // [M] | null
    const float* X; long long xs; int ldx; int K;     // X_b = X + b*xs, rows k < K, row stride ldx
    float* Y; long long ys; int ldy; int M;           // Y_b = Y + b*ys, rows m < M
};

template <int DT>
__global__ __launch_bounds__(256) void dts_seqmix_h_kernel(const SeqmixHArgs p) {
    constexpr int D = 16 * DT, XS = D + 8, MAXB = 4;                   // MAXB: 16-row blocks of W per wave (Mp <= 256)
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    _Float16* Xh = reinterpret_cast<_Float16*>(ldsb);      // [Kp32][XS]
    _Float16* Xl = Xh + p.Kp32 * XS;
    float* red = reinterpret_cast<float*>(Xl + p.Kp32 * XS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const float* xb = p.X + (long long)blockIdx.x * p.xs;
    float mx = 0.0f;
    for (int i = tid; i < p.K * (D / 4); i += 256) {
        const int k = i / (D / 4), c = i - k * (D / 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long long)k * p.ldx + 4 * c);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    const float sx = pow2_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    for (int i = tid; i < p.Kp32 * (D / 4); i += 256) {
        const int k = i / (D / 4), c = i - k * (D / 4);
        ah4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
        if (k < p.K) load_split4(xb + (long long)k * p.ldx + 4 * c, sx, hi, lo);
        *reinterpret_cast<ah4*>(Xh + k * XS + 4 * c) = hi;
        *reinterpret_cast<ah4*>(Xl + k * XS + 4 * c) = lo;
    }
    __syncthreads();
    const int nmt = p.Mp >> 4;
    f32x4 acc[MAXB][DT];
#pragma unroll
    for (int i = 0; i < MAXB; ++i)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[i][dt] = f32x4{0, 0, 0, 0};
    const int plane = p.Kp32 * XS;
    const _Float16* xt = Xh + (8 * g + (r16 >> 2)) * XS + 4 * (r16 & 3);      // transposed reads: this lane's address in the block rows 8 g .. of a chunk
    for (int c = 0; c < (p.Kp32 >> 5); ++c) {
        ah8 ah[MAXB], al[MAXB];
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            const int mt = wave + 4 * i;
            const long long wo = (long long)(16 * (mt < nmt ? mt : 0) + r16) * p.Kp32 + 32 * c + 8 * g;
            ah[i] = *reinterpret_cast<const ah8*>(p.Wh + wo);
            al[i] = *reinterpret_cast<const ah8*>(p.Wl + wo);
        }
        ah8 bh[DT], bl[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const _Float16* q = xt + 32 * c * XS + 16 * dt;
            bh[dt] = ah8_of(column4_tr(q), column4_tr(q + 4 * XS));
            bl[dt] = ah8_of(column4_tr(q + plane), column4_tr(q + plane + 4 * XS));
        }
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            if (wave + 4 * i >= nmt) break;                              // (wave-uniform)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) DTS_MMH3_32(acc[i][dt], ah[i], al[i], bh[dt], bl[dt]);
        }
    }
    const float inv = p.isw / sx;
    float* yb = p.Y + (long long)blockIdx.x * p.ys;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int mt = wave + 4 * i;
        if (mt >= nmt) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * mt + 4 * g + r;
            if (m < p.M) {
                const float bm = p.bias ? p.bias[m] : 0.0f;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) yb[(long long)m * p.ldy + 16 * dt + r16] = acc[i][dt][r] * inv + bm;
            }
        }
    }
}

}  // namespace dts
