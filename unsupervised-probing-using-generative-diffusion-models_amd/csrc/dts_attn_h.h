// DiffusionTS attention on the f16 matrix pipe at fp32-equivalent accuracy (head width 16 = the shipped 64 / 4), forward, dQ and dK-dV.
// Same structure and arithmetic order as the exact-fp32 kernels of dts_kernels.h (one workgroup per (sequence, head), 16-query / 16-key tiles,
// online softmax in base 2, the score tile in accumulator layout reused as the next product's B operand) with every 16 x 16 x 16 contraction as
// ONE v_mfma_f32_16x16x16_f16 per term of the two-term split x = hi + lo (both halves rounded to nearest: 24 significant bits; products
// hi*hi + hi*lo + lo*hi, fp32 accumulate; the dropped lo*lo is 2^-24 relative) instead of four v_mfma_f32_16x16x4_f32: 3 x 16.5 cycles against 4 x 32
// (tools/ubench/mfma_rates.hip).  The fp32 kernels spent 56 % (forward), 70 % (dQ) and 76 % (dK-dV) of their time on the matrix pipe.
// K / V (Q / dO in the dK-dV pass) are split ONCE while they are staged into LDS, as f16 hi / lo planes.  Operands indexed by (token, 4 head
// columns) are one ds_read_b64 per plane out of the row-major planes; operands indexed by (head column, 4 tokens) are read column-wise out of the
// SAME planes in the backward kernels (four 2-byte reads per plane; a transposed copy would cost a workgroup per CU, and these kernels live on
// occupancy), and from a transposed plane in the forward kernel, where V is used in no other way.
#pragma once

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

constexpr int AH_RS = 16;            // halves per row of a row-major plane ([tokens][16]; 32-byte rows: the ds_read_b64 of 16 rows is 2-way conflicted)

// x = hi + lo with hi = f16(x) and lo = f16(x - hi), BOTH rounded to nearest: |x - hi - lo| <= 2^-24 |x|, the precision of the fp32 operand itself
// (the truncating split of nst_gemm.h keeps 22 bits; the sign-descent sampler is sensitive to that difference: with it only 3 of 16 replayed
// sequences of the full-size 5-step test stayed on the oracle's trajectory, against 13 of 16 for the exact-fp32 kernels).
__device__ __forceinline__ void attn_split4(const float v0, const float v1, const float v2, const float v3, ah4& hi, ah4& lo) {
    const float v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        hi[c] = (_Float16)v[c];
        lo[c] = (_Float16)(v[c] - (float)hi[c]);
    }
}

// rows [n][16] of one head (token stride ld) -> f16 hi / lo planes: row-major [np][AH_RS] (rows >= n zero) and / or transposed [16][np]
__device__ __forceinline__ void stage_head_h(const float* src, int ld, int n, int np, _Float16* rh, _Float16* rl, _Float16* th, _Float16* tl) {
    for (int i = threadIdx.x; i < np * 4; i += 256) {                     // four head columns of a token per thread and pass
        const int r = i >> 2, d = 4 * (i & 3);
        const f32x4 v = r < n ? *reinterpret_cast<const f32x4*>(src + (long long)r * ld + d) : f32x4{0, 0, 0, 0};
        ah4 hi, lo;
        attn_split4(v[0], v[1], v[2], v[3], hi, lo);
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

// four tokens' values of head column `col` out of a row-major plane: p[(row0 + j) * AH_RS + col], j = 0..3
__device__ __forceinline__ ah4 column4(const _Float16* p, int row0, int col) {
    const _Float16* q = p + row0 * AH_RS + col;
    return ah4{q[0], q[AH_RS], q[2 * AH_RS], q[3 * AH_RS]};
}

__device__ __forceinline__ void load_split4(const float* p, ah4& hi, ah4& lo) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    attn_split4(v[0], v[1], v[2], v[3], hi, lo);
}

__global__ __launch_bounds__(256) void dts_attn_fwd_h_kernel(const AttnArgs p) {
    constexpr int HD = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int Lkp = (p.Lk + 15) & ~15;
    _Float16* Kh = reinterpret_cast<_Float16*>(ldsb);      // [Lkp][AH_RS]
    _Float16* Kl = Kh + Lkp * AH_RS;
    _Float16* VTh = Kl + Lkp * AH_RS;                      // [16][Lkp]
    _Float16* VTl = VTh + 16 * Lkp;
    stage_head_h(p.K + (long long)b * p.Lk * p.ldk + h * HD, p.ldk, p.Lk, Lkp, Kh, Kl, nullptr, nullptr);
    stage_head_h(p.V + (long long)b * p.Lk * p.ldv + h * HD, p.ldv, p.Lk, Lkp, nullptr, nullptr, VTh, VTl);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const float c = p.scale * LOG2E;
    for (int jt = wave; jt < ((p.Lq + 15) >> 4); jt += 4) {
        const int j = 16 * jt + r16;
        const bool jl = j < p.Lq;
        ah4 qh, ql;
        load_split4(p.Q + ((long long)b * p.Lq + (jl ? j : 0)) * p.ldq + h * HD + 4 * g, qh, ql);
        float m = -INFINITY, l = 0.0f;
        f32x4 o = {0, 0, 0, 0};
        for (int it = 0; it < (Lkp >> 4); ++it) {
            const ah4 kh = *reinterpret_cast<const ah4*>(Kh + (16 * it + r16) * AH_RS + 4 * g);
            const ah4 kl = *reinterpret_cast<const ah4*>(Kl + (16 * it + r16) * AH_RS + 4 * g);
            f32x4 s = {0, 0, 0, 0};
            DTS_MMH3(s, kh, kl, qh, ql);                                // S^T[key 4 g + r][query r16]
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
            ah4 ph, pl;
            attn_split4(s[0], s[1], s[2], s[3], ph, pl);
            const ah4 vh = *reinterpret_cast<const ah4*>(VTh + r16 * Lkp + 16 * it + 4 * g);
            const ah4 vl = *reinterpret_cast<const ah4*>(VTl + r16 * Lkp + 16 * it + 4 * g);
            DTS_MMH3(o, vh, vl, ph, pl);                                // O^T[d 4 g + r][query r16]
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        if (jl) {
            const float inv = 1.0f / l;
            float* dst = p.out + ((long long)b * p.Lq + j) * p.ldout + h * HD + 4 * g;
            *reinterpret_cast<f32x4*>(dst) = o * inv;
            if (g == 0 && p.lse) p.lse[((long long)b * p.H + h) * p.Lq + j] = m + __builtin_amdgcn_logf(l);
        }
    }
}

__global__ __launch_bounds__(256) void dts_attn_dq_h_kernel(const AttnArgs p) {
    constexpr int HD = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int Lkp = (p.Lk + 15) & ~15;
    _Float16* Kh = reinterpret_cast<_Float16*>(ldsb);      // [Lkp][AH_RS]
    _Float16* Kl = Kh + Lkp * AH_RS;
    _Float16* Vh = Kl + Lkp * AH_RS;
    _Float16* Vl = Vh + Lkp * AH_RS;
    stage_head_h(p.K + (long long)b * p.Lk * p.ldk + h * HD, p.ldk, p.Lk, Lkp, Kh, Kl, nullptr, nullptr);
    stage_head_h(p.V + (long long)b * p.Lk * p.ldv + h * HD, p.ldv, p.Lk, Lkp, Vh, Vl, nullptr, nullptr);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const float c = p.scale * LOG2E;
    for (int jt = wave; jt < ((p.Lq + 15) >> 4); jt += 4) {
        const int j = 16 * jt + r16;
        const bool jl = j < p.Lq;
        const long long tok = (long long)b * p.Lq + (jl ? j : 0);
        ah4 qh, ql, dh, dl;
        load_split4(p.Q + tok * p.ldq + h * HD + 4 * g, qh, ql);
        load_split4(p.dO + tok * p.ldo + h * HD + 4 * g, dh, dl);
        const float lse = p.lse[((long long)b * p.H + h) * p.Lq + (jl ? j : 0)];
        const float del = p.delta[((long long)b * p.H + h) * p.Lq + (jl ? j : 0)];
        f32x4 dq = {0, 0, 0, 0};
        for (int it = 0; it < (Lkp >> 4); ++it) {
            const int ro = (16 * it + r16) * AH_RS + 4 * g;
            const ah4 kh = *reinterpret_cast<const ah4*>(Kh + ro), kl = *reinterpret_cast<const ah4*>(Kl + ro);
            const ah4 vh = *reinterpret_cast<const ah4*>(Vh + ro), vl = *reinterpret_cast<const ah4*>(Vl + ro);
            f32x4 s = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
            DTS_MMH3(s, kh, kl, qh, ql);
            DTS_MMH3(t2, vh, vl, dh, dl);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = (16 * it + 4 * g + r < p.Lk) ? __builtin_amdgcn_exp2f(s[r] * c - lse) : 0.0f;
                s[r] = pr * (t2[r] - del) * p.scale;
            }
            ah4 sh, sl;
            attn_split4(s[0], s[1], s[2], s[3], sh, sl);
            const ah4 th = column4(Kh, 16 * it + 4 * g, r16), tl = column4(Kl, 16 * it + 4 * g, r16);      // K^T[d = r16][keys 4 g ..]
            DTS_MMH3(dq, th, tl, sh, sl);
        }
        if (jl) *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Lq + j) * p.ldout + h * HD + 4 * g) = dq;
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
    stage_head_h(p.Q + (long long)b * p.Lq * p.ldq + h * HD, p.ldq, p.Lq, Lqp, Qh, Ql, nullptr, nullptr);
    stage_head_h(p.dO + (long long)b * p.Lq * p.ldo + h * HD, p.ldo, p.Lq, Lqp, Dh, Dl, nullptr, nullptr);
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
        ah4 kh, kl, vh, vl;
        load_split4(p.K + tok * p.ldk + h * HD + 4 * g, kh, kl);
        load_split4(p.V + tok * p.ldv + h * HD + 4 * g, vh, vl);
        f32x4 dk = {0, 0, 0, 0}, dv = {0, 0, 0, 0};
        for (int it = 0; it < (Lqp >> 4); ++it) {
            const int ro = (16 * it + r16) * AH_RS + 4 * g;
            const ah4 qh = *reinterpret_cast<const ah4*>(Qh + ro), ql = *reinterpret_cast<const ah4*>(Ql + ro);
            const ah4 dh = *reinterpret_cast<const ah4*>(Dh + ro), dl = *reinterpret_cast<const ah4*>(Dl + ro);
            f32x4 s = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
            DTS_MMH3(s, qh, ql, kh, kl);                                // S[query 4 g + r][key r16]
            DTS_MMH3(t2, dh, dl, vh, vl);
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsel + 16 * it + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dell + 16 * it + 4 * g);
            f32x4 pr, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[r] = __builtin_amdgcn_exp2f(s[r] * c - l4[r]);
                ds[r] = pr[r] * (t2[r] - d4[r]) * p.scale;
            }
            ah4 ph, pl, sh, sl;
            attn_split4(pr[0], pr[1], pr[2], pr[3], ph, pl);
            attn_split4(ds[0], ds[1], ds[2], ds[3], sh, sl);
            const ah4 dth = column4(Dh, 16 * it + 4 * g, r16), dtl = column4(Dl, 16 * it + 4 * g, r16);      // dO^T[d = r16][queries 4 g ..]
            const ah4 qth = column4(Qh, 16 * it + 4 * g, r16), qtl = column4(Ql, 16 * it + 4 * g, r16);
            DTS_MMH3(dv, dth, dtl, ph, pl);
            DTS_MMH3(dk, qth, qtl, sh, sl);
        }
        if (jl) {
            *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Lk + j) * p.ldout + h * HD + 4 * g) = dk;
            *reinterpret_cast<f32x4*>(p.out2 + ((long long)b * p.Lk + j) * p.ldout2 + h * HD + 4 * g) = dv;
        }
    }
}

}  // namespace dts
