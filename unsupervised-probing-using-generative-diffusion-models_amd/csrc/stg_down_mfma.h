// DiffSTG / NsDiff_spatial, the "downsampling" product of a residual block (DiffSTG/ugnet.py:100-108, NsDiff/ugnet.py:101-109: the k = T + 1
// convolution, here Y[rows][N] = X[rows][K] W[N][K]^T + bias with K = channels x time = 800..3200 and N = Td_h x channels <= 160), round 4.
//
// Why its own kernel: on the general split-f16 GEMM (nst_gemm.h) these products are 200-400 workgroups on 256 CUs, each streaming its 128 rows of X
// through LDS 16 KB at a time between two workgroup barriers, padded to 128 output columns: 2.0-2.9 TB/s of an operand that is read exactly once
// (profiles/r04/down_kernel.txt; split-K measured negative, stg.hip down_ksplit).  The output is narrow enough for a wave to own WHOLE rows of it:
//   * eight waves (two per SIMD), each with 16 rows x ALL N columns (<= 10 accumulator tiles of v_mfma_f32_16x16x32_f16).  A wave stages ITS OWN rows
//     of X: 4 coalesced 16-byte loads per lane and 64-wide k stage (an instruction covers 4 rows x 256 contiguous bytes), two stages in flight in
//     registers (64 KB per CU), split into two f16 terms on the way into a wave-private LDS tile (the nst_gemm.h split: hi = 11 significant bits,
//     lo = x - hi; hi*hi + hi*lo + lo*hi in fp32 accumulators) and read back as MFMA operands -- no workgroup barrier on the X path, a wave's LDS
//     operations are ordered among themselves;
//   * the pre-split weight planes go through LDS once per workgroup (64 k per stage, double buffered, one barrier per stage), shared by all waves;
//   * waves 4-7 run half a stage ahead on the X path (they multiply first and split the next stage afterwards), so that on every SIMD one wave's
//     vector / LDS-store phase runs under the other's products;
//   * the operand reads are software-pipelined by hand (fragments of the next two column tiles requested before the products of these two, order
//     pinned with sched_barrier) and use a chunk swizzle made for 16-row fragments (dn_off below: SQ_LDS_BANK_CONFLICT = 0, against 50 % of the
//     LDS cycles with nst_gemm.h's h_off);
//   * the weights are the MFMA's M index, so a lane ends with 4 consecutive output columns of one row: 16-byte stores, no staging of the output.
// Measured alone (tools/ubench/gemm_h3 GEMM_DOWN=1, 25 600 rows): N = 80: K = 3200 125 -> 78-84 us, K = 1600 60 -> 38 us, K = 800 34 -> 23 us; N = 160,
// K = 3200: 165 -> 123-157 us.  Inside the sampler: c5 9.37 -> 9.88 windows/s, NsDiff_spatial 25.1 -> 25.55 (same box, MPV_STG_DOWN=0 for the former).
// What the versions before this one taught (same file, same shapes):
//   1. X loaded straight into the operand layout (lane l holds row l & 15, 8 floats at k = 8 (l >> 4)): no LDS for X at all -- 93-108 us, no better
//      with more loads in flight or two workgroups per CU.  Not HBM: a kernel that only READS X in 256-byte row pieces, two stages in flight, does it
//      in 58 us (5.6 TB/s; tools/ubench/row_pieces.hip).  The time was the serial chain of a stage at one wave per SIMD: probes without the products,
//      without the split, without the weight path and without any global load add up to the whole (42 + 30 + 20 of 92 us).
//   2. A guarded load (`k < K ? load : 0`) compiles to an exec-mask branch and the compiler then waits vmcnt(0) behind it every stage; a select applied
//      where the value is loaded waits for the load.  Loads here are unconditional on clamped addresses, zeroing happens where the value is used.
//   3. Wider accumulator sets (32 rows x 160 columns per wave): 388 registers, the compiler sinks the weight loads to their use -- slower than two
//      column groups; two column groups re-read X (no faster than the general kernel).  16 rows x 160 columns at two waves per SIMD: 218 registers.
#pragma once
#include "nst_gemm.h"

namespace {

typedef float dn4f __attribute__((ext_vector_type(4)));

constexpr int DN_KC = 64;            // k per stage (two MFMA k-steps)
constexpr int DN_WAVES = 8;          // two per SIMD: one wave's split / LDS phase runs under the other's products
constexpr int DN_MT = 1;             // 16-row tiles per wave
constexpr int DN_ROWS = DN_WAVES * 16 * DN_MT;     // rows per workgroup
constexpr int DN_MAX_N = 160;

// Operand tiles in LDS: [rows][32 halves], 64-byte rows of four 16-byte chunks, as in nst_gemm.h -- but h_off's chunk swizzle is made for the 32-row
// fragments of v_mfma_*_32x32x16 (lane & 31 = row).  Here a fragment is 16 rows x 4 chunks (lane & 15 = row, lane >> 4 = chunk), and ds_read_b128
// serves the lanes {0-3, 12-15, 20-27} in one LDS cycle (MI355X_MICROARCH.md, LDS table): rows 0-3 and 12-15 at chunk c, rows 4-11 at chunk c + 1.
// With h_off those are 2-way bank conflicts on every operand read (measured: the products of a stage took 1900 cycles for 60 MFMAs); the chunk
// permutation g(row >> 2) = {0, 2, 3, 1} makes all four lane groups conflict-free.
__device__ __forceinline__ int dn_off(int row, int half) {
    return row * 32 + ((((half >> 3) ^ (0x78 >> (2 * ((row >> 2) & 3)))) & 3) << 3) + (half & 7);
}
constexpr int DN_PAD = 32;           // halves between the sub-tiles of a plane: k-steps 0 and 1 of a row land 64 bytes apart (mod 128: stores are banked
                                     // mod 32 dwords), so a store group that writes both halves of a row's 64 k hits distinct banks

// NT: 16-column tiles of the output a workgroup computes (weight rows beyond N are clamped on load and masked on store).
// Uses NstGemmHArgs (X, ldx, Wh, Wl, inv_scale, bias, Y, ldy, M, N, K, ldw); act / R / split-K fields must be unset.
// PROBE (tools/ubench/gemm_h3.hip only; results are wrong): 1 no products, 2 no split / LDS write of X, 4 no weight path in the loop, 8 no X loads in the loop
template <int NT, int PROBE = 0>
__global__ __launch_bounds__(64 * DN_WAVES, 1) void stg_down_h3_kernel(const NstGemmHArgs p) {
    constexpr int SUB = 16 * NT * 32 + DN_PAD;                 // halves of one [16 NT][32] weight sub-tile (one MFMA k-step of one plane)
    constexpr int BUF = 4 * SUB;                          // halves per weight stage: planes (hi, lo) x k-steps (0, 1)
    constexpr int XSUB = 16 * DN_MT * 32 + DN_PAD;              // halves of one [32 rows][32] activation sub-tile
    extern __shared__ __attribute__((aligned(16))) _Float16 wt[];                     // [2 stages][BUF] weights, then [waves][4 XSUB] activations
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    _Float16* xs = wt + 2 * BUF + wave * 4 * XSUB;        // this wave's tile: planes (hi, lo) x k-steps (0, 1)
    const long long mblock = blockIdx.x;
    constexpr int n0 = 0;
    const long long m0 = mblock * DN_ROWS + wave * (16 * DN_MT);
    // ---- activations: load j of a stage = rows m0 + 4 j + lg, 4 floats at k = 64 kc + 4 lr ------------------------------------------------------
    // (every load of the k loop is unconditional -- a guarded load becomes an exec-mask branch, and the compiler then drains ALL loads in flight
    // before the first use behind it.  Beyond K the address is clamped into the row and the value replaced by 0 when it is split.)
    constexpr int XL = 4 * DN_MT;                         // loads per lane and stage
    const float* xbase = p.X + (size_t)mblock * DN_ROWS * p.ldx;             // (uniform; a workgroup's 128 rows span far less than 4 GB)
    unsigned xoff[XL];                                    // byte offset of row 4 j + lg of this wave
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        long long r = m0 + 4 * j + lg;
        if (r >= p.M) r = p.M - 1;
        xoff[j] = (unsigned)((r - mblock * DN_ROWS) * p.ldx * 4);
    }
    struct XStage { dn4f v[XL]; };
    auto fetch_x = [&](XStage& st, int kc) {
        const int k = kc * DN_KC + 4 * lr, kk = k < p.K ? k : p.K - 4;       // K is a multiple of 4: a 16-byte piece is inside the row or beyond it
#pragma unroll
        for (int j = 0; j < XL; ++j) st.v[j] = *reinterpret_cast<const dn4f*>(reinterpret_cast<const char*>(xbase) + (xoff[j] + 4u * (unsigned)kk));
    };
    int xw_off[XL];                                       // where load j lands: sub-tile (4 lr) >> 5, row 4 j + lg, halves (4 lr) & 31 ..+3
#pragma unroll
    for (int j = 0; j < XL; ++j) xw_off[j] = ((4 * lr) >> 5) * XSUB + dn_off(4 * j + lg, (4 * lr) & 31);
    auto stash_x = [&](const XStage& st, int kc) {
        const bool tail = (kc + 1) * DN_KC > p.K, in = kc * DN_KC + 4 * lr < p.K;     // (tail: uniform -- the selects cost a stage that is wholly inside K nothing)
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            dn4f v = st.v[j];
            if (tail && !in) v = dn4f{0.0f, 0.0f, 0.0f, 0.0f};
            float hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                hi[c] = __uint_as_float(__float_as_uint(v[c]) & 0xFFFFE000u);
                lo[c] = v[c] - hi[c];
            }
            u2 ph, pl;
            ph[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(hi[0], hi[1]));
            ph[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(hi[2], hi[3]));
            pl[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo[0], lo[1]));
            pl[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo[2], lo[3]));
            *reinterpret_cast<u2*>(xs + xw_off[j]) = ph;
            *reinterpret_cast<u2*>(xs + 2 * XSUB + xw_off[j]) = pl;
        }
    };
    // ---- weights: stage kc = rows n < 16 NT, halves [64 kc, 64 kc + 64) of both planes = 256 NT 16-byte chunks, NT per thread --------------
    constexpr int WCH = 256 * NT, WL = (WCH + 64 * DN_WAVES - 1) / (64 * DN_WAVES);     // 16-byte chunks of a stage, loads per thread
    struct WStage { h8 v[WL]; };
    auto wchunk = [&](int j, int& plane, int& n, int& c) {
        int id = tid + 64 * DN_WAVES * j;
        if (id >= WCH) id = WCH - 1;                      // (odd NT: the last round has spare threads; they repeat the last chunk)
        plane = id / (128 * NT);
        const int rem = id % (128 * NT);
        n = rem >> 3;
        c = rem & 7;
    };
    auto fetch_w = [&](WStage& wreg, int kc) {
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            int plane, n, c;
            wchunk(j, plane, n, c);
            n += n0;
            int k = kc * DN_KC + 8 * c;
            if (n >= p.N) n = p.N - 1;
            if (k >= p.ldw) k = p.ldw - 8;                // (second k-step of the last stage when ldw is an odd multiple of 32: X is 0 there)
            wreg.v[j] = *reinterpret_cast<const h8*>((plane ? p.Wl : p.Wh) + (size_t)n * p.ldw + k);
        }
    };
    auto stash_w = [&](const WStage& wreg, int buf) {
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            int plane, n, c;
            wchunk(j, plane, n, c);
            *reinterpret_cast<h8*>(wt + buf * BUF + (2 * plane + (c >> 2)) * SUB + dn_off(n, 8 * (c & 3))) = wreg.v[j];
        }
    };
    dn4f acc[DN_MT][NT];
#pragma unroll
    for (int i = 0; i < DN_MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = dn4f{0.0f, 0.0f, 0.0f, 0.0f};
    const int f_off = dn_off(lr, 8 * lg);                 // operand fragment of row lr of a 16-row tile (tile bases are multiples of 16 rows: the swizzle depends on lr alone)
    // (one wave per SIMD: nothing else hides an LDS read's latency, so the weight fragments of tile t + 1 are requested before the products of tile
    // t, and the order is pinned -- left to itself the compiler reads each pair right before its first product: 1750 cycles per stage for 60 MFMAs)
    auto multiply = [&](int buf) {
        const _Float16* wb_ = wt + buf * BUF + f_off;
        h8 xh[2][DN_MT], xl[2][DN_MT];
#pragma unroll
        for (int i = 0; i < DN_MT; ++i) {
            xh[0][i] = *reinterpret_cast<const h8*>(xs + i * 16 * 32 + f_off);
            xl[0][i] = *reinterpret_cast<const h8*>(xs + 2 * XSUB + i * 16 * 32 + f_off);
        }
        // units u = (k-step, column tile), two at a time: their products alternate between two accumulators (three dependent MFMAs back to back
        // wait for each other), the fragments of the next two units are requested before the products of these two
        h8 wq[2][2][2];                                   // [ring slot][unit of the pair][hi, lo]
        auto wread = [&](int u, h8 (&w)[2]) {
            const int s = u / NT, t = u % NT;
            w[0] = *reinterpret_cast<const h8*>(wb_ + s * SUB + t * 16 * 32);
            w[1] = *reinterpret_cast<const h8*>(wb_ + (2 + s) * SUB + t * 16 * 32);
        };
        wread(0, wq[0][0]);
        wread(1, wq[0][1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NT; ++q) {                    // pair q = units 2 q, 2 q + 1
            if (q + 1 < NT) {
                wread(2 * q + 2, wq[(q + 1) & 1][0]);
                wread(2 * q + 3, wq[(q + 1) & 1][1]);
            }
            if (q == 0) {
#pragma unroll
                for (int i = 0; i < DN_MT; ++i) {
                    xh[1][i] = *reinterpret_cast<const h8*>(xs + XSUB + i * 16 * 32 + f_off);
                    xl[1][i] = *reinterpret_cast<const h8*>(xs + 3 * XSUB + i * 16 * 32 + f_off);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int u0 = 2 * q, u1 = 2 * q + 1, s0 = u0 / NT, t0 = u0 % NT, s1 = u1 / NT, t1 = u1 % NT;
            const h8 (&w0)[2] = wq[q & 1][0];
            const h8 (&w1)[2] = wq[q & 1][1];
#pragma unroll
            for (int i = 0; i < DN_MT; ++i) {
                acc[i][t0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[1], xh[s0][i], acc[i][t0], 0, 0, 0);
                acc[i][t1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[1], xh[s1][i], acc[i][t1], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < DN_MT; ++i) {
                acc[i][t0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[0], xl[s0][i], acc[i][t0], 0, 0, 0);
                acc[i][t1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[0], xl[s1][i], acc[i][t1], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < DN_MT; ++i) {
                acc[i][t0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[0], xh[s0][i], acc[i][t0], 0, 0, 0);
                acc[i][t1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[0], xh[s1][i], acc[i][t1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int nkc = (p.ldw + DN_KC - 1) / DN_KC;
    XStage sa, sb;
    WStage wa;
    auto last = [&](int kc) { return kc < nkc ? kc : nkc - 1; };
    // Waves w and w + 4 share a SIMD, and the barrier of a stage keeps all eight in step: if every wave split its activations first and multiplied
    // second, the matrix pipe would idle through the first phase and the vector pipe through the second (measured: 78 us = the sum of the phases).
    // So waves 4-7 run half a stage ahead on the X path: they multiply stage kc FIRST (their tile was written at the end of the previous stage) and
    // split stage kc + 1 into the same tile afterwards (wave-private and in program order: no second buffer), under the products of waves 0-3.
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= DN_WAVES / 2;
    fetch_w(wa, 0);
    fetch_x(sa, 0);
    if (late) {
        stash_x(sa, 0);
        fetch_x(sa, last(1));                             // waves 4-7: sa holds the odd stages, sb the even ones from 2 on
        fetch_x(sb, last(2));
    } else {
        fetch_x(sb, last(1));
    }
    stash_w(wa, 0);
    fetch_w(wa, last(1));
    __syncthreads();
    // stage kc: park the weights of kc + 1 (requested a stage ago) in the other buffer and request kc + 2 into their registers; waves 0-3 split the
    // activations of kc into their tile and request kc + 2 into the registers, then multiply; waves 4-7 multiply, then split kc + 1 and request
    // kc + 3; one barrier.  (Past the last stage the requests repeat it: no branch inside a phase.  A second register set of weights, requested two
    // stages ahead, changed nothing.)
#define DN_STAGE(ST, KC_)                                         \
    do {                                                          \
        const int kc_ = (KC_);                                    \
        if (!(PROBE & 4)) stash_w(wa, (kc_ + 1) & 1);             \
        if (!(PROBE & 4)) fetch_w(wa, last(kc_ + 2));             \
        if (!late) {                                              \
            if (!(PROBE & 2)) stash_x(ST, kc_);                   \
            if (!(PROBE & 8)) fetch_x(ST, last(kc_ + 2));         \
            __builtin_amdgcn_wave_barrier();                      \
        }                                                         \
        if (!(PROBE & 1)) multiply(kc_ & 1);                      \
        if (late) {                                               \
            __builtin_amdgcn_wave_barrier();                      \
            if (!(PROBE & 2)) stash_x(ST, kc_ + 1);               \
            if (!(PROBE & 8)) fetch_x(ST, last(kc_ + 3));         \
        }                                                         \
        __syncthreads();                                          \
    } while (0)
    for (int kc = 0; kc < nkc; kc += 2) {
        DN_STAGE(sa, kc);
        if (kc + 1 >= nkc) break;
        DN_STAGE(sb, kc + 1);
    }
#undef DN_STAGE
    // ---- epilogue: lane holds row m0 + 16 i + lr, columns n0 + 16 t + 4 lg ..+3 -------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < DN_MT; ++i) {
        const long long r = m0 + 16 * i + lr;
        if (r >= p.M) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + 16 * t + 4 * lg;
            if (n >= p.N) continue;                       // N is a multiple of 4
            dn4f y = acc[i][t] * p.inv_scale;
            if (p.bias) y += *reinterpret_cast<const dn4f*>(p.bias + n);
            *reinterpret_cast<dn4f*>(p.Y + (size_t)r * p.ldy + n) = y;
        }
    }
}

// True when the product fits this kernel (the caller falls back to nst_gemm_h3_launch otherwise).  MPV_STG_DOWN: 0 never, 1 whenever the shape
// allows (tests: short K too), unset: long-K products only.
inline bool stg_down_fits(const NstGemmHArgs& a) {
    const char* env = getenv("MPV_STG_DOWN");
    const int mode = env ? atoi(env) : -1;
    if (mode == 0) return false;
    if (a.N > DN_MAX_N || a.N % 4 != 0 || a.K % 4 != 0 || a.ldx % 4 != 0 || a.ldy % 4 != 0) return false;
    if (a.act != 0 || a.R != nullptr || a.ksplit > 1 || a.xparts > 1 || a.xbias != nullptr) return false;
    if ((reinterpret_cast<uintptr_t>(a.X) | reinterpret_cast<uintptr_t>(a.Y) | reinterpret_cast<uintptr_t>(a.bias)) & 15) return false;
    return mode == 1 || a.K >= 512;
}

inline void stg_down_h3_launch(const NstGemmHArgs& a, hipStream_t stream) {
    const int nt = (a.N + 15) / 16;
    const unsigned grid = (unsigned)((a.M + DN_ROWS - 1) / DN_ROWS);
#define DN_LAUNCH(NT_)                                                                                                                       \
    do {                                                                                                                                     \
        constexpr int lds = (2 * 4 * (16 * (NT_) * 32 + DN_PAD) + DN_WAVES * 4 * (16 * DN_MT * 32 + DN_PAD)) * (int)sizeof(_Float16);                                                              \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(stg_down_h3_kernel<NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        (void)attr;                                                                                                                          \
        hipLaunchKernelGGL((stg_down_h3_kernel<NT_>), dim3(grid), dim3(64 * DN_WAVES), lds, stream, a);                                                \
    } while (0)
    if (nt <= 1) DN_LAUNCH(1);
    else if (nt <= 2) DN_LAUNCH(2);
    else if (nt <= 3) DN_LAUNCH(3);
    else if (nt <= 4) DN_LAUNCH(4);
    else if (nt <= 5) DN_LAUNCH(5);
    else if (nt <= 6) DN_LAUNCH(6);
    else if (nt <= 8) DN_LAUNCH(8);
    else DN_LAUNCH(10);
#undef DN_LAUNCH
}

}  // namespace
