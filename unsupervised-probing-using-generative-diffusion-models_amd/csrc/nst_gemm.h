// f(x) backbone, Linear layers: Y[M][N] = act(X[M][K] W[N][K]^T + bias) + R on the matrix cores of gfx950 (included by nst.hip and by
// tools/ubench/gemm_h3.hip, which times the kernels alone).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float nst_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// ---------------------------------------------------------------------------------------------------------------------
// GEMM  Y[M][N] = act(X[M][K] W[N][K]^T + bias) + R
// ---------------------------------------------------------------------------------------------------------------------
struct NstGemmArgs {
    const float* X; long long ldx;
    const float* W;           // [N][K]
    const float* bias;        // [N] | null
    const float* R; long long ldr;     // residual | null
    float* Y; long long ldy;
    long long M;
    int N, K;
    int act;                  // 0 none, 1 relu, 2 gelu, 3 multiply by gelu'(R) instead of adding R
};

constexpr int G_BM = 128, G_BN = 128, G_BK = 32, G_LD = G_BK + 4;

// Workgroup id -> output tile.  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: XCD x takes the x-th
// contiguous eighth of the tile list, in which the N tiles of one 128-row block of X are neighbours -- they run at the same time
// on the same XCD, so the X block is fetched from HBM once and served to the others by that L2 (the weights, a few MB, stay
// resident in every L2).  Returns false for the padding workgroups of the last chunk.
template <int BM = G_BM>
__device__ __forceinline__ bool gemm_tile(long long M, int N, long long& m0, int& n0, long long id = -1) {
    const long long n_n = (N + G_BN - 1) / G_BN, total = ((M + BM - 1) / BM) * n_n;
    const long long per = (total + 7) / 8;
    if (id < 0) id = blockIdx.x;
    const long long t = (id & 7) * per + (id >> 3);
    if ((id >> 3) >= per || t >= total) return false;
    m0 = (t / n_n) * BM;
    n0 = (int)(t % n_n) * G_BN;
    return true;
}

constexpr int C_LD = G_BN + 4;                  // floats per row of the output tile staged in LDS

// Output tile: accumulators -> LDS [128][C_LD] -> rows of 512 contiguous bytes (a wave writes two whole rows per instruction; the
// accumulator layout itself would give 128-byte pieces of 64 different rows).  Bias, activation and residual are applied on the way out.
// N must be a multiple of 4.
template <bool PROBE_NO_STORE = false, int MT = 2, bool HALF_STAGE = false>      // MT: 32-row accumulator tiles per wave (the workgroup tile has BM = 64 MT rows)
__device__ __forceinline__ void gemm_store_tile(float* Cs, const v16f (&acc)[MT][2], float scale, long long m0, int n0, long long M, int N, const float* bias,
                                                int act, const float* R, long long ldr, float* Y, long long ldy) {
    constexpr int BM = 64 * MT;
    // MT = 4 (256-row workgroup tiles, round 4) and HALF_STAGE: the staging buffer holds half the tile's rows, so the two wave rows (wm = 0, 1) go
    // through it one after the other
    constexpr int HALVES = (MT == 4 || HALF_STAGE) ? 2 : 1, HR = BM / HALVES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int c4e = (tid & 31) * 4;
    const bool add_r = R && act != 3 && n0 + c4e < N;
    const int c4 = (tid & 31) * 4, n = n0 + c4;
    v4f b4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (bias && n < N) b4 = *reinterpret_cast<const v4f*>(bias + n);
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
        const long long mh = m0 + (long long)half * HR;
        // the residual tile is requested first, all row pieces of this thread at once: the loads fly while the accumulators go through LDS (a
        // product with a short K -- DiffSTG's up-sampling maps, K = 80..160 against N = 3200 -- is all epilogue: 237 us per launch with the loads
        // issued four at a time inside the store loop)
        v4f rr[HR / 8];
        if (add_r) {
#pragma unroll
            for (int it = 0; it < HR / 8; ++it) {
                const long long m = mh + it * 8 + (tid >> 5);
                rr[it] = m < M ? *reinterpret_cast<const v4f*>(R + (size_t)m * ldr + n0 + c4e) : v4f{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
        __syncthreads();                              // every wave is done with the operand tiles (resp. with the previous half's rows)
        if (HALVES == 1 || wm == half) {
            const int wrow = HALVES == 1 ? wm * 32 * MT : 0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * C_LD + wn * 64 + j * 32 + li] = acc[i][j][r] * scale;
        }
        __syncthreads();
        if (n >= N) continue;
#pragma unroll
        for (int it = 0; it < HR / 8; ++it) {
            const int row = it * 8 + (tid >> 5);
            const long long m = mh + row;
            if (m >= M) break;
            v4f y = *reinterpret_cast<const v4f*>(Cs + row * C_LD + c4) + b4;
            if (act == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) y[c] = fmaxf(y[c], 0.0f);
            } else if (act == 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) y[c] = nst_gelu(y[c]);
            }
            if (act == 3) {                           // y *= gelu'(R[m][n]): the GELU backward of an MLP, R = the saved pre-activation
                const v4f u = *reinterpret_cast<const v4f*>(R + (size_t)m * ldr + n);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    y[c] *= 0.5f * (1.0f + erff(u[c] * 0.70710678118654752f)) + u[c] * 0.39894228040143268f * __expf(-0.5f * u[c] * u[c]);
            } else if (add_r) y += rr[it];
            if (!PROBE_NO_STORE || y[0] == 1.2345e30f) *reinterpret_cast<v4f*>(Y + (size_t)m * ldy + n) = y;
        }
    }
}

__global__ __launch_bounds__(256, 2) void nst_gemm_kernel(const NstGemmArgs p) {
    extern __shared__ float sm[];
    float* As = sm;                               // [2][G_BM][G_LD]
    float* Bs = sm + 2 * G_BM * G_LD;             // [2][G_BN][G_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    long long m0;
    int n0;
    if (!gemm_tile(p.M, p.N, m0, n0)) return;
    // global -> register staging: thread covers rows (tid>>3) + 32 j, j = 0..3, k quad (tid & 7)
    const int lr = tid >> 3, lk = (tid & 7) * 4;
    const float* xa[4];
    const float* wb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long long r = m0 + lr + 32 * j;
        if (r >= p.M) r = p.M - 1;
        xa[j] = p.X + (size_t)r * p.ldx + lk;
        int n = n0 + lr + 32 * j;
        if (n >= p.N) n = p.N - 1;
        wb[j] = p.W + (size_t)n * p.K + lk;
    }
    v4f ra[4], rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ra[j] = *reinterpret_cast<const v4f*>(xa[j]); rb[j] = *reinterpret_cast<const v4f*>(wb[j]); }
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<v4f*>(As + (size_t)(lr + 32 * j) * G_LD + lk) = ra[j];
        *reinterpret_cast<v4f*>(Bs + (size_t)(lr + 32 * j) * G_LD + lk) = rb[j];
    }
    __syncthreads();
    const int nk = p.K / G_BK;
    const int li = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ra[j] = *reinterpret_cast<const v4f*>(xa[j] + (size_t)(kt + 1) * G_BK);
                rb[j] = *reinterpret_cast<const v4f*>(wb[j] + (size_t)(kt + 1) * G_BK);
            }
        }
        const float* a_base = As + (size_t)cur * G_BM * G_LD + (size_t)(wm * 64 + li) * G_LD + 4 * lh;
        const float* b_base = Bs + (size_t)cur * G_BN * G_LD + (size_t)(wn * 64 + li) * G_LD + 4 * lh;
#pragma unroll
        for (int k8 = 0; k8 < G_BK / 8; ++k8) {
            v4f a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const v4f*>(a_base + (size_t)i * 32 * G_LD + 8 * k8);
                b[i] = *reinterpret_cast<const v4f*>(b_base + (size_t)i * 32 * G_LD + 8 * k8);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            float* an = As + (size_t)(cur ^ 1) * G_BM * G_LD;
            float* bn = Bs + (size_t)(cur ^ 1) * G_BN * G_LD;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<v4f*>(an + (size_t)(lr + 32 * j) * G_LD + lk) = ra[j];
                *reinterpret_cast<v4f*>(bn + (size_t)(lr + 32 * j) * G_LD + lk) = rb[j];
            }
        }
        __syncthreads();
    }
    gemm_store_tile(sm, acc, 1.0f, m0, n0, p.M, p.N, p.bias, p.act, p.R, p.ldr, p.Y, p.ldy);      // LDS is large enough: 2 (128 + 128) x 36 floats
}

// The same GEMM on the f16 matrix pipe at fp32-equivalent accuracy (the split the chain sampler uses): every
// operand is the sum of two f16 numbers, x = hi + lo with hi = x truncated to 11 significant bits (exact in f16) and lo = x - hi,
// and the product is hi*hi + hi*lo + lo*hi in fp32 accumulators (the dropped lo*lo term is 2^-22 relative).  Weights are split
// once at load, scaled by a power of two per matrix so that lo stays a normal f16; activations are split while the tile is
// written to LDS.  Three v_mfma_f32_32x32x16_f16 per k-step of 16: 5.3x the fp32 pipe's rate.  Inputs must fit f16's range
// (|x| < 65504): every GEMM of this network reads a LayerNorm output, an attention average of such, or a GELU/ReLU of one.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

struct NstGemmHArgs {
    const float* X; long long ldx;
    const _Float16* Wh; const _Float16* Wl;     // [N][ldw] each, zero beyond column K
    float inv_scale;                            // 1 / (power of two the weights were scaled by)
    const float* bias;
    const float* R; long long ldr;
    float* Y; long long ldy;
    long long M;
    int N, K;                                   // K a multiple of 4; the k loop runs over ldw (K rounded up to 32), X columns >= K read as 0
    int act;
    int ldw;
    // Round 4, split-K with a fixed summation order (DiffSTG's long-K, narrow-N "downsampling" products: 25 600 rows x N = 80..160 are 200-400 tiles
    // on 256 CUs, each streaming K = 800..3200 with 32 KB in flight -- concurrency-bound at 1.6-2.4 TB/s).  ksplit > 1: the k-tiles of an output tile
    // are dealt to ksplit workgroups in contiguous chunks; workgroup s writes its partial sums (x inv_scale; NO bias / activation / residual) to
    // Y + s * ypart.  The consumer reads them back through xparts / xpart / xbias: its A operand is ((X_0 + X_1) + ...) + xbias[k], always in that
    // order, so the result is bitwise repeatable.
    int ksplit = 1;
    long long ypart = 0;
    int xparts = 1;
    long long xpart = 0;
    const float* xbias = nullptr;
};

// Operand tiles in LDS: rows of G_BK = 32 halves = four 16-byte chunks.  Round 3 padded a row to 80 bytes, which makes the operand reads (16 lanes x
// 16 bytes, consecutive rows) conflict-free but not the tile writes (8 or 16 bytes per lane, 64 contiguous bytes per row, next row 80 bytes on: rows 0
// and 3 of a 256-byte pass share banks) -- PMC on the QKV product: SQ_LDS_BANK_CONFLICT = 31 % of SQ_LDS_IDX_ACTIVE (profiles/r04/gemm_h3_ubench.txt).
// Round 4: unpadded 64-byte rows with the chunk index XOR-ed by (row >> 2) & 3 -- writes cover whole 64-byte rows (any chunk order), and the 16 rows
// of a read pass hit bank group 16 (row & 3) + 4 (chunk ^ (row >> 2)): all distinct.  H_SWZ 0 restores the padded layout.
#ifndef H_SWZ
#define H_SWZ 1
#endif
constexpr int H_LD = H_SWZ ? G_BK : G_BK + 8;   // halves per LDS row
__device__ __forceinline__ int h_off(int row, int half) {        // offset (halves) of element `half` (a multiple of 4) of row `row`
#if H_SWZ
    return row * H_LD + ((((half >> 3) ^ (row >> 2)) & 3) << 3) + (half & 7);
#else
    return row * H_LD + half;
#endif
}

// PROBE (tools/ubench/gemm_h3.hip only): 1 no epilogue stores, 2 one k-tile, 4 no global loads inside the loop
// BM = 128 (four waves of 64 x 64; the shipped shape) or 64 (four waves of 32 x 64: half the registers and LDS, three workgroups per CU; see
// nst_gemm_h3_launch).
// BM = 256 (round 4: four waves of 128 x 64): the k loop of the 128-row shape runs at half the matrix pipe's pace because the CU's ONE LDS port is as
// busy as its four matrix pipes -- per 32-wide k-tile a workgroup reads 4 waves x 16 ds_read_b128 (64 KB) and writes 32 KB of split operands for
// 96 MFMAs, and two resident workgroups make that 1536 LDS cycles per 1536 MFMA cycles per SIMD.  A 128 x 64 wave tile needs 24 reads per 48 MFMAs
// instead of 32, and the 256 x 128 workgroup tile stages 384 operand rows per 2x the products instead of 512: 0.75x the LDS traffic per MFMA.  One
// register set of staged tiles (the k-tile's compute phase is twice as long); the output goes through the 128-row staging buffer in two halves.
// SK: the split-K producer / consumer paths (ksplit, xparts, xbias) exist only in this instantiation -- in the plain one they cost 6 VGPRs the 128-row
// shape does not have (250 -> 256 + 9 spilled: DiffSTG's step 9.2 -> 7.2 windows/s when they were runtime branches of the one kernel).
// LITE (BM = 128 only): one register set of staged tiles and a 64-row output staging buffer -- <= 168 VGPRs and 34 KB of LDS, so THREE workgroups share
// a CU (three waves per SIMD) where the full shape has two.
template <int PROBE = 0, int BM = G_BM, bool SK = false, bool LITE = false>
#ifndef NST_LITE_WGS
#define NST_LITE_WGS 3
#endif
__global__ __launch_bounds__(256, LITE ? NST_LITE_WGS : (BM == 64 ? 3 : 2)) void nst_gemm_h3_kernel(const NstGemmHArgs p) {
    constexpr int MT = BM / 64;
    constexpr int SR = BM == 256 ? 128 : (LITE ? BM / 2 : BM);                  // rows of the output staging buffer
    static_assert(!LITE || BM == 128, "LITE is a variant of the 128-row shape");
    static_assert(BM == 64 || BM == 128 || BM == 256, "workgroup tile rows");
    static_assert((2 * BM + 2 * G_BN) * H_LD * 2 <= SR * C_LD * 4, "the operand tiles fit the output tile's LDS");
    __shared__ __attribute__((aligned(16))) float smem[SR * C_LD];             // operand tiles, then the output tile
    _Float16* Ah = reinterpret_cast<_Float16*>(smem);
    _Float16* Al = Ah + BM * H_LD;
    _Float16* Bh = Al + BM * H_LD;
    _Float16* Bl = Bh + G_BN * H_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    long long m0;
    int n0;
    const int ksplit = (SK && p.ksplit > 1) ? p.ksplit : 1, split = ksplit > 1 ? (int)(blockIdx.x % ksplit) : 0;
    if (!gemm_tile<BM>(p.M, p.N, m0, n0, ksplit > 1 ? (long long)(blockIdx.x / ksplit) : -1)) return;
    const int nk_all = (PROBE & 2) ? 1 : p.ldw / G_BK, kchunk = (nk_all + ksplit - 1) / ksplit;
    const int kt0 = split * kchunk, k0 = kt0 * G_BK;       // this workgroup's k-tiles: [kt0, kt0 + nk)
    const int lr = tid >> 3, lk = (tid & 7) * 4;           // X: rows lr + 32 j, 4 floats at lk
    const int wr = tid >> 2, wk = (tid & 3) * 8;           // W: rows wr + 64 j, 8 halves at wk
    const float* xa[2 * MT];
    const _Float16* wh[2];
    const _Float16* wl[2];
#pragma unroll
    for (int j = 0; j < 2 * MT; ++j) {
        long long r = m0 + lr + 32 * j;
        if (r >= p.M) r = p.M - 1;
        xa[j] = p.X + (size_t)r * p.ldx + lk + k0;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int n = n0 + wr + 64 * j;
        if (n >= p.N) n = p.N - 1;
        wh[j] = p.Wh + (size_t)n * p.ldw + wk + k0;
        wl[j] = p.Wl + (size_t)n * p.ldw + wk + k0;
    }
    // Two register sets of staged tiles, so that a global load has two compute phases to land before it is written to LDS.
    struct Stage { v4f x[2 * MT]; h8 h[2], l[2]; };
    Stage sa, sb;                                          // (BM = 256 and the split-K instance never touch sb: dead, no registers)
    auto fetch = [&](Stage& st, int kt) {
#pragma unroll
        for (int j = 0; j < 2 * MT; ++j) {
            if (k0 + kt * G_BK + lk < p.K) {
                v4f v = *reinterpret_cast<const v4f*>(xa[j] + (size_t)kt * G_BK);
                if constexpr (SK) {
                    for (int s = 1; s < p.xparts; ++s) v += *reinterpret_cast<const v4f*>(xa[j] + (size_t)s * p.xpart + (size_t)kt * G_BK);   // partial sums of a split-K producer
                    if (p.xbias) v += *reinterpret_cast<const v4f*>(p.xbias + k0 + kt * G_BK + lk);
                }
                st.x[j] = v;
            } else {
                st.x[j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            st.h[j] = *reinterpret_cast<const h8*>(wh[j] + (size_t)kt * G_BK);
            st.l[j] = *reinterpret_cast<const h8*>(wl[j] + (size_t)kt * G_BK);
        }
    };
    auto stash = [&](const Stage& st) {                    // split the activations, write the tile to LDS
#pragma unroll
        for (int j = 0; j < 2 * MT; ++j) {
            float hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = st.x[j][c];
                hi[c] = __uint_as_float(__float_as_uint(v) & 0xFFFFE000u);
                lo[c] = v - hi[c];
            }
            u2 ph, pl;
            ph[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(hi[0], hi[1]));
            ph[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(hi[2], hi[3]));
            pl[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo[0], lo[1]));
            pl[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo[2], lo[3]));
            *reinterpret_cast<u2*>(Ah + h_off(lr + 32 * j, lk)) = ph;
            *reinterpret_cast<u2*>(Al + h_off(lr + 32 * j, lk)) = pl;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<h8*>(Bh + h_off(wr + 64 * j, wk)) = st.h[j];
            *reinterpret_cast<h8*>(Bl + h_off(wr + 64 * j, wk)) = st.l[j];
        }
    };
    v16f acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nk = kt0 >= nk_all ? 0 : (kt0 + kchunk <= nk_all ? kchunk : nk_all - kt0);      // (0: an empty split still writes its zero partial)
    const int li = lane & 31, lh = lane >> 5;
    // (row bases are multiples of 32, so the swizzle of a lane's rows depends on li alone: one pair of offsets per k-step, fixed for the kernel)
    const int a_row = wm * 32 * MT + li, b_row = wn * 64 + li;
    int a_off[G_BK / 16], b_off[G_BK / 16];
#pragma unroll
    for (int ks = 0; ks < G_BK / 16; ++ks) {
        a_off[ks] = h_off(a_row, 16 * ks + 8 * lh);
        b_off[ks] = h_off(b_row, 16 * ks + 8 * lh);
    }
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < G_BK / 16; ++ks) {
            h8 ah[MT], al[MT], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = *reinterpret_cast<const h8*>(Ah + a_off[ks] + i * 32 * H_LD);
                al[i] = *reinterpret_cast<const h8*>(Al + a_off[ks] + i * 32 * H_LD);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bh[i] = *reinterpret_cast<const h8*>(Bh + b_off[ks] + i * 32 * H_LD);
                bl[i] = *reinterpret_cast<const h8*>(Bl + b_off[ks] + i * 32 * H_LD);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    const bool partial = ksplit > 1;
    const float* e_bias = partial ? nullptr : p.bias;
    const float* e_R = partial ? nullptr : p.R;
    const int e_act = partial ? 0 : p.act;
    float* e_Y = partial ? p.Y + (size_t)split * p.ypart : p.Y;
    if (nk == 0) {
        gemm_store_tile<(PROBE & 1) != 0, MT, LITE>(smem, acc, p.inv_scale, m0, n0, p.M, p.N, e_bias, e_act, e_R, p.ldr, e_Y, p.ldy);
        return;
    }
    fetch(sa, 0);
    stash(sa);
    if (nk > 1) fetch(sa, 1);
    if constexpr (BM == 256 || SK || LITE) {               // one register set: tile kt + 2 is requested once tile kt + 1 has left the registers
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            compute();                                     // tile kt
            if (kt + 1 >= nk) break;
            __syncthreads();
            stash(sa);                                     // tile kt + 1
            __syncthreads();
            if (!(PROBE & 4) && kt + 2 < nk) fetch(sa, kt + 2);
        }
        gemm_store_tile<(PROBE & 1) != 0, MT, LITE>(smem, acc, p.inv_scale, m0, n0, p.M, p.N, e_bias, e_act, e_R, p.ldr, e_Y, p.ldy);
        return;
    }
    if (nk > 2) fetch(sb, 2);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        compute();                                         // tile kt
        if (kt + 1 >= nk) break;
        __syncthreads();
        stash(sa);                                         // tile kt + 1
        __syncthreads();
        if (!(PROBE & 4) && kt + 3 < nk) fetch(sa, kt + 3);
        compute();                                         // tile kt + 1
        if (kt + 2 >= nk) break;
        __syncthreads();
        stash(sb);                                         // tile kt + 2
        __syncthreads();
        if (!(PROBE & 4) && kt + 4 < nk) fetch(sb, kt + 4);
    }
    gemm_store_tile<(PROBE & 1) != 0, MT>(smem, acc, p.inv_scale, m0, n0, p.M, p.N, e_bias, e_act, e_R, p.ldr, e_Y, p.ldy);
}

// Launch on `stream`.  MPV_GEMM_BM=64 selects the 64-row workgroup tiles for every product (a tuning aid: at DiffSTG's BASELINE shapes they are
// 5 % faster on the N = 80 products and 40 % slower on N = 160, K = 3200; the whole step 8.7 against 9.1 windows/s with 128 rows).
inline void nst_gemm_h3_launch(const NstGemmHArgs& a, hipStream_t stream) {
    const long long n_n = (a.N + G_BN - 1) / G_BN;
    static const int bm_env = getenv("MPV_GEMM_BM") ? atoi(getenv("MPV_GEMM_BM")) : 0;
    // 64-row tiles (MPV_GEMM_BM=64, a tuning aid): alone on the device they win on DiffSTG's single-N-tile "downsampling" products (N = 80: 56.9 -> 46.7 us
    // at K = 1600 with the swizzled operand tiles), inside the sampler step they lose -- c5 8.68 windows/s with them on those products against 9.22
    // with 128 rows everywhere (profiles/r04/gemm_h3_ubench.txt) -- so they are not chosen automatically.
    const bool rows64 = bm_env == 64;
    // 256-row tiles for the wide products (f(x)'s fused QKV: N = 3 d_model) when they still fill the chip several times over.  Measured on one
    // 216 800-row pass (tools/ubench/gemm_h3, profiles/r04/gemm_h3_ubench.txt): N = 1536, K = 512: 1.179 -> 1.053 ms (289 -> 324 TFLOP/s);
    // N = 512: 0.428 -> 0.417; N = 256: 0.230 -> 0.237; N = 512, K = 256: 0.278 -> 0.306 (the longer prologue / epilogue of a 256-row tile outweighs
    // the leaner k-loop once a row block has few N tiles) -- so only N >= 1024.  MPV_GEMM_BM=128 keeps the round-3 shape for A/B runs.
    const long long tiles256 = ((a.M + 255) / 256) * n_n;
    if (a.ksplit > 1 || a.xparts > 1 || a.xbias) {        // split-K producer / consumer: 128-row tiles, ksplit workgroups per tile (consecutive block ids)
        const long long tiles = ((a.M + G_BM - 1) / G_BM) * n_n;
        hipLaunchKernelGGL((nst_gemm_h3_kernel<0, 128, true>), dim3((unsigned)(8 * ((tiles + 7) / 8) * (a.ksplit > 1 ? a.ksplit : 1))), dim3(256), 0, stream, a);
        return;
    }
    // (and K >= 256: DiffSTG's up-sampling products, N = 3200 at K = 80..160, are all epilogue -- 206 us per launch with 256-row tiles against 165)
    if (bm_env == 256) {                                  // (superseded as a default by the three-workgroup shape below: QKV 1.077 ms against 0.984)
        hipLaunchKernelGGL((nst_gemm_h3_kernel<0, 256>), dim3((unsigned)(8 * ((tiles256 + 7) / 8))), dim3(256), 0, stream, a);
        return;
    }
    if (rows64) {
        const long long tiles = ((a.M + 63) / 64) * n_n;
        hipLaunchKernelGGL((nst_gemm_h3_kernel<0, 64>), dim3((unsigned)(8 * ((tiles + 7) / 8))), dim3(256), 0, stream, a);
    } else if (bm_env == 128) {                           // the round-3 shape: two register sets of staged tiles, whole-tile output staging, 2 workgroups per CU
        const long long tiles = ((a.M + G_BM - 1) / G_BM) * n_n;
        hipLaunchKernelGGL((nst_gemm_h3_kernel<0, 128>), dim3((unsigned)(8 * ((tiles + 7) / 8))), dim3(256), 0, stream, a);
    } else {
        // Default (round 4): the same 128 x 128 tile with ONE register set and a half-tile output staging buffer -- 164 VGPRs, 34 KB of LDS, three
        // workgroups per CU.  tools/ubench/gemm_h3 on a 216 800-row pass: QKV 1.165 -> 0.984 ms (347 TFLOP/s), out + residual 0.418 -> 0.376,
        // FFN 0.222 -> 0.199 and 0.278 -> 0.243; the k loop alone 0.765 -> 0.636 ms.  (Four per CU would need <= 128 VGPRs: 65 spilled, 3.9 ms.)
        const long long tiles = ((a.M + G_BM - 1) / G_BM) * n_n;
        hipLaunchKernelGGL((nst_gemm_h3_kernel<0, 128, false, true>), dim3((unsigned)(8 * ((tiles + 7) / 8))), dim3(256), 0, stream, a);
    }
}

}  // namespace
