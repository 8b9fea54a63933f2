// DiffusionTS on gfx950: weights, workspace, the forward / input-gradient schedules of the 9-block transformer and the DDIM +
// Langevin-infill sampler (reference DiffusionTS_model.py:72-109, DiffusionTS.py:152-160, 277-310, 359-407), behind the
// mpv_dts_* entry points of include/mpv_hip.h.  The kernels are in dts_kernels.h (MFMA) and dts_small.h (VALU).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

#include "dts_small.h"
#include "nst_gemm.h"

using namespace dts;

namespace {

struct EncW { const float *ln2w, *ln2b, *wqkv, *bqkv, *wp, *bp, *w0, *b0, *w2, *b2, *wqkv_t, *wp_t, *w0_t, *w2_t; };
struct DecW {
    const float *ln2w, *ln2b;
    const float *wqkv1, *bqkv1, *wp1, *bp1, *wqkv1_t, *wp1_t;
    const float *wq2, *bq2, *wkv2, *bkv2, *wp2, *bp2, *wq2_t, *wkv2_t, *wp2_t;
    const float *tw0, *tb0, *tw3, *tb3;
    const float *w0, *b0, *w2, *b2, *w0_t, *w2_t;
    // proj (Conv1d L -> 2L over the token axis) is never applied on its own: its first half is folded into the trend block's first
    // convolution (tw0 / tb0x below are the composites) and its second half into the real-DFT basis of the Fourier layer
    const float *wdft, *bdft, *wdft_t;             // [2 NFp][Lp] = DFT basis x proj[L:], its row bias, and the transpose [L][2 NFp]
    const _Float16 *wdft_h = nullptr, *wdft_l = nullptr;   // the same matrix as scaled f16 hi / lo planes [2 NFp][Lp32] (split-f16 token mixing)
    float wdft_isw = 1.0f;                         // 1 / its power of two
    const float *wlin, *blin;
};

}  // namespace

struct DtsEngine {
    mpv_dts_desc d{};
    int L = 0, F = 0, D = 0, H = 0, hd = 0, NF = 0, NFp = 0, topk = 0, Lp = 0, L2p = 0, n_ada = 0, M = 0;
    float* blob = nullptr;
    std::vector<EncW> enc;
    std::vector<DecW> dec;
    const float *wemb = nullptr, *bemb = nullptr, *winv = nullptr, *binv = nullptr, *wcs = nullptr, *wcm = nullptr, *pe_enc = nullptr,
                *pe_dec = nullptr, *ada = nullptr, *freq = nullptr, *poly = nullptr, *cosT = nullptr, *sinT = nullptr;
    // workspace (one allocation, grow-only)
    float* ws = nullptr;
    size_t ws_floats = 0;
    int cap_B = 0;
    // sub-buffers for batch cap_B (set by carve())
    std::vector<float*> xe, xd;                      // block inputs: [n_enc+1], [n_dec+1]
    struct EncS { float *qkv, *att, *lse, *x1, *u; };
    struct DecS { float *qkv1, *att1, *lse1, *x1, *q2, *kv2, *att2, *lse2, *x2, *u, *t3; int* fidx; };
    std::vector<EncS> es;
    std::vector<DecS> dsv;
    float *spec = nullptr, *famp = nullptr, *fph = nullptr, *x3 = nullptr, *trend_sum = nullptr, *season_sum = nullptr,
          *means = nullptr, *trend = nullptr, *season = nullptr;
    float *gA = nullptr, *gB = nullptr, *gC = nullptr, *gwide = nullptr, *gqkv = nullptr, *gkv = nullptr, *genc = nullptr,
          *delta = nullptr, *gseason = nullptr, *gmeans = nullptr, *gout = nullptr, *gx = nullptr;
    // sampler state [B][L][F]
    float *img = nullptr, *target = nullptr, *xstart = nullptr, *pmean = nullptr, *noise = nullptr;
    // Round 4: the three consumers of a decoder block's x2 -- trend block, Fourier layer (token mixing, selection, synthesis), MLP -- are independent
    // until the mean removal; the first two are short latency-bound kernels (1.2-3 TB/s alone on the device), so they run on two auxiliary streams
    // beside the MLP (fork / join with events; every buffer is written by one branch only, so the arithmetic is unchanged).  MPV_DTS_STREAMS=0: one stream.
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    bool snapshots = false;                          // mpv_dts_vjp: keep the gradient at the encoder output for mpv_dts_debug_tensor
    // Attention operands (Q | K | V, attention output, and their gradients) in 16-column BLOCKED layout [block][tokens][16] when a head is 16
    // wide: a (sequence, head) tile is then 12.8 KB of contiguous memory instead of 64-byte pieces of 768-byte rows (half of every 128-byte
    // line unused: measured 1.16 GB per forward attention launch against 0.66 GB algorithmic).  MPV_DTS_ROWMAJOR=1 keeps the round-2 layout.
    bool blocked = false;
};

namespace {

// ---- launch helpers ------------------------------------------------------------------------------------------------
template <class K>
int set_lds(mpv_ctx* ctx, K kernel, size_t bytes) {
    if (bytes > ctx->max_lds) return mpv_fail(ctx, MPV_EINVAL, "DiffusionTS: a kernel needs more LDS than the device has (sequence too long)");
    MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MPV_OK;
}

// Linear layers follow the engine's gemm mode (set by forward / backward / the sampler on entry): split-f16 products unless MPV_GEMM_F32
thread_local bool t_lin_f16 = false;

int lin(mpv_ctx* ctx, const float* X, int ldx, const float* W, const float* bias, int K, int N, long long rows, float* Y, int ldy, int pro = PRO_NONE,
        const float* P1 = nullptr, const float* P2 = nullptr, int ldp = 0, const float* R = nullptr, int ldr = 0, int accumulate = 0, long long xblk = 0,
        long long yblk = 0, const float* lnx = nullptr, const float* lng = nullptr, int ln_ada = 0) {
    LinArgs a{X, ldx, W, bias, R, ldr, Y, ldy, P1, P2, ldp, rows, K, N, pro, accumulate, xblk, yblk};
    a.lnx = lnx; a.lng = lng; a.ln_ada = ln_ada;
    // (lnx: the fused LayerNorm backward of dts_lin_wave_h_kernel; -100 = not available for this shape / mode, the caller runs the two kernels)
    if (lnx && !(t_lin_f16 && !getenv("MPV_DTS_LIN_F32") && !getenv("MPV_DTS_LIN_TILE") && !getenv("MPV_DTS_NO_LNFUSE") && K % 32 == 0 && N % 16 == 0 && N <= 64 &&
                 !R && !yblk && ldy == N))
        return -100;
    static const bool tile_kernel = getenv("MPV_DTS_LIN_TILE") != nullptr;              // validation aid: the workgroup-tile kernel for every call
    static const bool lin_exact = getenv("MPV_DTS_LIN_F32") != nullptr;                 // A/B aid: keep the Linear layers on the fp32 matrix pipe
    if (t_lin_f16 && !lin_exact && !tile_kernel && K % 32 == 0 && N % 16 == 0) {        // split-f16 products (gemm mode f16x3), dts_attn_h.h
        const size_t wl = 2 * sizeof(_Float16) * (size_t)N * (K + 8) + sizeof(float) * LINW_WAVES;
        const long long wtiles = ((rows + 15) / 16 + LINW_WAVES - 1) / LINW_WAVES;
        const dim3 wblock(64 * LINW_WAVES);
#define DTS_LINWH_(KT_, LNB_)                                                                                                        \
    do {                                                                                                                             \
        if (int rc = set_lds(ctx, dts_lin_wave_h_kernel<KT_, LNB_>, wl)) return rc;                                                  \
        static thread_local std::map<size_t, int> known;                      /* LDS bytes -> resident workgroups per CU */         \
        int& resident = known[wl];                                                                                                   \
        if (!resident)                                                                                                               \
            MPV_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, dts_lin_wave_h_kernel<KT_, LNB_>, 64 * LINW_WAVES, wl)); \
        const dim3 wgrid((unsigned)std::min<long long>(wtiles, (long long)std::max(1, resident) * ctx->cu_count));                   \
        hipLaunchKernelGGL((dts_lin_wave_h_kernel<KT_, LNB_>), wgrid, wblock, wl, ctx->stream, a);                                   \
        MPV_HIP_TRY(ctx, hipGetLastError());                                                                                         \
        return MPV_OK;                                                                                                               \
    } while (0)
#define DTS_LINWH(KT_)                  \
    do {                                \
        if (lnx) DTS_LINWH_(KT_, true); \
        else DTS_LINWH_(KT_, false);    \
    } while (0)
        switch (K / 16) {
            case 2: DTS_LINWH(2);
            case 4: DTS_LINWH(4);
            case 6: DTS_LINWH(6);
            case 8: DTS_LINWH(8);
            case 12: DTS_LINWH(12);
            case 16: DTS_LINWH(16);
            default: break;
        }
#undef DTS_LINWH
#undef DTS_LINWH_
    }
    if (!tile_kernel && K % 16 == 0 && N % 16 == 0) {
        const size_t wl = sizeof(float) * (size_t)N * (K + 4);
        const long long wtiles = ((rows + 15) / 16 + LINW_WAVES - 1) / LINW_WAVES;
        // persistent grid = exactly the workgroups that are resident at once (registers and LDS decide: 2 per CU at K = 64, 1 at K >= 128); a
        // larger grid leaves a partial second round (K = 64: 150 -> 130 us per launch at 3200 x 200 tokens)
        static const int wcap = getenv("MPV_DTS_LINW_WGS") ? std::max(1, atoi(getenv("MPV_DTS_LINW_WGS"))) : 0;        // tuning aid: force it
        const dim3 wblock(64 * LINW_WAVES);
#define DTS_LINW(KT_)                                                                                                              \
    do {                                                                                                                           \
        if (int rc = set_lds(ctx, dts_lin_wave_kernel<KT_>, wl)) return rc;                                                        \
        static thread_local std::map<size_t, int> known;                      /* LDS bytes -> resident workgroups per CU */       \
        int& resident = known[wl];                                                                                                 \
        if (!resident)                                                                                                             \
            MPV_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, dts_lin_wave_kernel<KT_>, 64 * LINW_WAVES, wl));  \
        const long long wper = wcap ? wcap : std::max(1, resident);                                                                \
        const dim3 wgrid((unsigned)std::min<long long>(wtiles, wper * ctx->cu_count));                                             \
        hipLaunchKernelGGL(dts_lin_wave_kernel<KT_>, wgrid, wblock, wl, ctx->stream, a);                                           \
        MPV_HIP_TRY(ctx, hipGetLastError());                                                                                       \
        return MPV_OK;                                                                                                             \
    } while (0)
        switch (K / 16) {
            case 1: DTS_LINW(1);
            case 2: DTS_LINW(2);
            case 3: DTS_LINW(3);
            case 4: DTS_LINW(4);
            case 6: DTS_LINW(6);
            case 8: DTS_LINW(8);
            case 12: DTS_LINW(12);
            case 16: DTS_LINW(16);
            default: break;                       // other widths: the workgroup-tile kernel below
        }
#undef DTS_LINW
    }
    if (xblk || yblk) return mpv_fail(ctx, MPV_ESTATE, "DiffusionTS: blocked operands need a dts_lin_wave_kernel instance");
    const size_t lds = sizeof(float) * (size_t)(N + LIN_TM) * (K + 4);
    if (int rc = set_lds(ctx, dts_lin_kernel, lds)) return rc;
    const long long tiles = (rows + LIN_TM - 1) / LIN_TM;
    const long long per_cu = std::max<long long>(1, std::min<long long>(4, (long long)ctx->max_lds / (long long)lds));
    const int grid = (int)std::min<long long>(tiles, per_cu * ctx->cu_count);
    hipLaunchKernelGGL(dts_lin_kernel, dim3(grid), dim3(LIN_THREADS), lds, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

// Y = X W^T (W [N][K]) on the tiled exact-fp32 MFMA GEMM of nst_gemm.h (128x128x32 LDS tiles, two workgroups per CU), for the two
// products of an MLP's backward whose operands are too wide for dts_lin_kernel's "all of W + a 64-token tile in LDS" to leave more than
// one workgroup per CU.  act 3: the epilogue multiplies by gelu'(U).
int wide_gemm(mpv_ctx* ctx, const float* X, int ldx, const float* W, int K, int N, long long rows, float* Y, int ldy, const float* U = nullptr, int ldu = 0) {
    NstGemmArgs a{X, ldx, W, nullptr, U, ldu, Y, ldy, rows, N, K, U ? 3 : 0};
    const size_t lds = sizeof(float) * 2 * (G_BM + G_BN) * G_LD;
    if (int rc = set_lds(ctx, nst_gemm_kernel, lds)) return rc;
    const long long tiles = ((rows + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
    hipLaunchKernelGGL(nst_gemm_kernel, dim3((unsigned)(8 * ((tiles + 7) / 8))), dim3(256), lds, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

// split-f16 token mixing (dts_seqmix_h_kernel): W as pre-split planes [Mp][Kp32]
int seqmix_h(mpv_ctx* ctx, int D, const _Float16* Wh, const _Float16* Wl, float isw, int Mp, int Kp32, const float* bias, const float* X, long long xs, int ldx,
             int K, float* Y, long long ys, int ldy, int M, int B) {
    if (Mp > 256) return -100;
    SeqmixHArgs a{Wh, Wl, Mp, Kp32, isw, bias, X, xs, ldx, K, Y, ys, ldy, M};
    const size_t lds = 2 * sizeof(_Float16) * (size_t)Kp32 * (D + 8) + 64;
    if (D == 64) {
        if (int rc = set_lds(ctx, dts_seqmix_h_kernel<4>, lds)) return rc;
        hipLaunchKernelGGL(dts_seqmix_h_kernel<4>, dim3(B), dim3(256), lds, ctx->stream, a);
    } else if (D == 32) {
        if (int rc = set_lds(ctx, dts_seqmix_h_kernel<2>, lds)) return rc;
        hipLaunchKernelGGL(dts_seqmix_h_kernel<2>, dim3(B), dim3(256), lds, ctx->stream, a);
    } else if (D == 16) {
        if (int rc = set_lds(ctx, dts_seqmix_h_kernel<1>, lds)) return rc;
        hipLaunchKernelGGL(dts_seqmix_h_kernel<1>, dim3(B), dim3(256), lds, ctx->stream, a);
    } else {
        return -100;
    }
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

int seqmix(mpv_ctx* ctx, int D, const float* W, int Mp, int Kp, const float* bias, const float* X, long long xs, int ldx, int K, float* Y,
           long long ys, int ldy, int M, int B, int accumulate = 0) {
    SeqmixArgs a{W, Mp, Kp, bias, X, xs, ldx, K, Y, ys, ldy, M, accumulate};
    const size_t lds = sizeof(float) * (size_t)Kp * (D + 4);
    if (D == 64) {
        if (int rc = set_lds(ctx, dts_seqmix_kernel<4>, lds)) return rc;
        hipLaunchKernelGGL(dts_seqmix_kernel<4>, dim3(B), dim3(256), lds, ctx->stream, a);
    } else if (D == 32) {
        if (int rc = set_lds(ctx, dts_seqmix_kernel<2>, lds)) return rc;
        hipLaunchKernelGGL(dts_seqmix_kernel<2>, dim3(B), dim3(256), lds, ctx->stream, a);
    } else if (D == 16) {
        if (int rc = set_lds(ctx, dts_seqmix_kernel<1>, lds)) return rc;
        hipLaunchKernelGGL(dts_seqmix_kernel<1>, dim3(B), dim3(256), lds, ctx->stream, a);
    } else {
        return mpv_fail(ctx, MPV_EINVAL, "DiffusionTS: d_model must be 16, 32 or 64");
    }
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

template <int E>
int attn_launch(mpv_ctx* ctx, int mode, const AttnArgs& a, int B) {
    const int Lkp = (a.Lk + 15) & ~15, Lqp = (a.Lq + 15) & ~15, RS = 4 * E + 4;
    const dim3 grid(B * a.H), block(256);
    if (mode == 0) {
        const size_t lds = sizeof(float) * ((size_t)Lkp * RS + 16 * (Lkp + 4));
        if (int rc = set_lds(ctx, dts_attn_fwd_kernel<E>, lds)) return rc;        // (a one-pass softmax with the 13 score tiles of a query tile in
        hipLaunchKernelGGL(dts_attn_fwd_kernel<E>, grid, block, lds, ctx->stream, a);   //  registers measured 15 % slower than this online one)
    } else if (mode == 1) {
        const size_t lds = sizeof(float) * (2 * (size_t)Lkp * RS);
        if (int rc = set_lds(ctx, dts_attn_dq_kernel<E>, lds)) return rc;
        hipLaunchKernelGGL(dts_attn_dq_kernel<E>, grid, block, lds, ctx->stream, a);
    } else {
        const size_t lds = sizeof(float) * (2 * (size_t)Lqp * RS + 2 * Lqp);
        if (int rc = set_lds(ctx, dts_attn_dkv_kernel<E>, lds)) return rc;
        hipLaunchKernelGGL(dts_attn_dkv_kernel<E>, grid, block, lds, ctx->stream, a);
    }
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

// the split-f16 kernels of dts_attn_h.h (head width 16): LDS = f16 hi / lo planes of K (+ V transposed / row-major), or of Q and dO
int attn_launch_h(mpv_ctx* ctx, int mode, const AttnArgs& a, int B) {
    const int Lkp = (a.Lk + 15) & ~15, Lqp = (a.Lq + 15) & ~15;
    const dim3 grid(B * a.H), block(256);
    if (mode == 0) {
        const size_t lds = sizeof(_Float16) * (2 * (size_t)Lkp * AH_RS + 2 * 16 * (size_t)Lkp) + 64;
        if (int rc = set_lds(ctx, dts_attn_fwd_h_kernel, lds)) return rc;
        hipLaunchKernelGGL(dts_attn_fwd_h_kernel, grid, block, lds, ctx->stream, a);
    } else if (mode == 1) {
        const size_t lds = sizeof(_Float16) * (4 * (size_t)Lkp * AH_RS) + 64;
        if (int rc = set_lds(ctx, dts_attn_dq_h_kernel, lds)) return rc;
        hipLaunchKernelGGL(dts_attn_dq_h_kernel, grid, block, lds, ctx->stream, a);
    } else if (mode == 2) {
        const size_t lds = sizeof(_Float16) * (4 * (size_t)Lqp * AH_RS) + sizeof(float) * 2 * Lqp + 64;
        if (int rc = set_lds(ctx, dts_attn_dkv_h_kernel, lds)) return rc;
        hipLaunchKernelGGL(dts_attn_dkv_h_kernel, grid, block, lds, ctx->stream, a);
    } else {                                                                  // 3: dQ, dK, dV in one pass (round 4)
        const size_t lds = attn_bwd_h_lds(a.Lq);
        const int nqt = Lqp >> 4;
#define DTS_ABWD(NT_)                                                                         \
    do {                                                                                      \
        if (int rc = set_lds(ctx, dts_attn_bwd_h_kernel<NT_>, lds)) return rc;                \
        hipLaunchKernelGGL(dts_attn_bwd_h_kernel<NT_>, grid, block, lds, ctx->stream, a);     \
    } while (0)
        if (nqt <= 4) DTS_ABWD(4);
        else if (nqt <= 8) DTS_ABWD(8);
        else if (nqt <= 13) DTS_ABWD(13);
        else DTS_ABWD(AH_MAXT);
#undef DTS_ABWD
    }
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

int attn(mpv_ctx* ctx, int mode, const AttnArgs& a, int B, bool f16) {
    if (f16 && a.hd == 16 && a.Lk <= 64 * AH_MAXQ && a.Lq <= 64 * AH_MAXQ) return attn_launch_h(ctx, mode, a, B);   // (a thread holds AH_MAXQ blocks of a staged head tile)
    switch (a.hd) {
        case 4: return attn_launch<1>(ctx, mode, a, B);
        case 8: return attn_launch<2>(ctx, mode, a, B);
        case 16: return attn_launch<4>(ctx, mode, a, B);
        default: return mpv_fail(ctx, MPV_EINVAL, "DiffusionTS: head dimension must be 4, 8 or 16");
    }
}

int ln_bwd(mpv_ctx* ctx, int D, const float* g, const float* x, const float* gam, int ada, float* dx, int accumulate, long long rows) {
    const dim3 grid((unsigned)((rows + 15) / 16)), block(256);
    if (D == 64) hipLaunchKernelGGL(dts_ln_bwd_kernel<4>, grid, block, 0, ctx->stream, g, x, gam, ada, dx, accumulate, rows);
    else if (D == 32) hipLaunchKernelGGL(dts_ln_bwd_kernel<2>, grid, block, 0, ctx->stream, g, x, gam, ada, dx, accumulate, rows);
    else hipLaunchKernelGGL(dts_ln_bwd_kernel<1>, grid, block, 0, ctx->stream, g, x, gam, ada, dx, accumulate, rows);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

int mlp_fwd(mpv_ctx* ctx, int D, int M, const float* X, const float* W0, const float* b0, const float* W2, const float* b2, const float* lnw,
            const float* lnb, float* Y, float* U, long long rows, bool f16) {
    MlpArgs a{X, W0, b0, W2, b2, lnw, lnb, Y, U, rows};
    const size_t lds = f16 ? 2 * sizeof(_Float16) * ((size_t)M * (D + 8) + (size_t)D * (M + 8)) + sizeof(float) * 2 * MLP_WAVES
                           : sizeof(float) * ((size_t)M * (D + 4) + (size_t)D * (M + 4));
    const long long tiles = ((rows + 15) / 16 + MLP_WAVES - 1) / MLP_WAVES;              // 16-token tiles, MLP_WAVES of them per workgroup pass
    const long long per_cu = std::max<long long>(1, std::min<long long>(2, (long long)ctx->max_lds / (long long)lds));
    const int grid = (int)std::min<long long>(tiles, per_cu * ctx->cu_count);
    const bool two_tiles = !(getenv("MPV_DTS_MLP_TILES") && atoi(getenv("MPV_DTS_MLP_TILES")) == 1);       // (A/B aid: 1 = the one-tile kernel)
#define DTS_MLP(DT_, MT_)                                                                              \
    do {                                                                                               \
        if (f16 && two_tiles && (DT_) % 2 == 0) {                                                      \
            if (int rc = set_lds(ctx, dts_mlp_fwd_h2_kernel<(DT_) % 2 ? 2 : (DT_), MT_>, lds)) return rc; \
            const long long pairs = (((rows + 15) / 16 + 1) / 2 + MLP_WAVES - 1) / MLP_WAVES;         \
            const int grid2 = (int)std::min<long long>(pairs, per_cu * ctx->cu_count);                \
            hipLaunchKernelGGL((dts_mlp_fwd_h2_kernel<(DT_) % 2 ? 2 : (DT_), MT_>), dim3(grid2), dim3(64 * MLP_WAVES), lds, ctx->stream, a); \
        } else if (f16) {                                                                              \
            if (int rc = set_lds(ctx, dts_mlp_fwd_h_kernel<DT_, MT_>, lds)) return rc;                 \
            hipLaunchKernelGGL((dts_mlp_fwd_h_kernel<DT_, MT_>), dim3(grid), dim3(64 * MLP_WAVES), lds, ctx->stream, a); \
        } else {                                                                                       \
            if (int rc = set_lds(ctx, dts_mlp_fwd_kernel<DT_, MT_>, lds)) return rc;                   \
            hipLaunchKernelGGL((dts_mlp_fwd_kernel<DT_, MT_>), dim3(grid), dim3(64 * MLP_WAVES), lds, ctx->stream, a); \
        }                                                                                              \
    } while (0)
    if (D == 64 && M == 256) DTS_MLP(4, 16);
    else if (D == 32 && M == 128) DTS_MLP(2, 8);
    else if (D == 16 && M == 64) DTS_MLP(1, 4);
    else return -100;                                                     // no fused instance: the caller falls back to two dts_lin launches
#undef DTS_MLP
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

// d_ln = W0^T (gelu'(u) * (W2^T g)) in one kernel (see dts_mlp_bwd_kernel); -100 when no fused instance exists
// lnx / lng: the fused LayerNorm backward of dts_mlp_bwd_h_kernel (Y accumulates the LayerNorm's input gradient; f16 instances only: -100 otherwise)
int mlp_bwd(mpv_ctx* ctx, int D, int M, const float* G, const float* U, const float* W2T, const float* W0T, float* Y, long long rows, bool f16,
            const float* lnx = nullptr, const float* lng = nullptr) {
    MlpBwdArgs a{G, U, W2T, W0T, Y, rows};
    if (lnx && (!f16 || getenv("MPV_DTS_NO_LNFUSE"))) return -100;
    a.lnx = lnx; a.lng = lng;
    const size_t lds = f16 ? 2 * sizeof(_Float16) * ((size_t)M * (D + 8) + (size_t)D * (M + 8)) + sizeof(float) * 2 * MLP_WAVES
                           : sizeof(float) * ((size_t)M * (D + 4) + (size_t)D * (M + 4));
    const long long tiles = ((rows + 15) / 16 + MLP_WAVES - 1) / MLP_WAVES;
    const long long per_cu = std::max<long long>(1, std::min<long long>(2, (long long)ctx->max_lds / (long long)lds));
    const int grid = (int)std::min<long long>(tiles, per_cu * ctx->cu_count);
    const bool two_tiles = getenv("MPV_DTS_MLPB_TILES") && atoi(getenv("MPV_DTS_MLPB_TILES")) == 2;       // measured slower (331 vs 304 us, profiles/r04/c3_mlp_two_tiles.txt): opt-in
#define DTS_MLPB(DT_, MT_)                                                                                           \
    do {                                                                                                            \
        if (f16 && two_tiles && (DT_) % 2 == 0) {                                                                   \
            if (int rc = set_lds(ctx, dts_mlp_bwd_h2_kernel<(DT_) % 2 ? 2 : (DT_), MT_>, lds)) return rc;           \
            const long long pairs = (((rows + 15) / 16 + 1) / 2 + MLP_WAVES - 1) / MLP_WAVES;                      \
            const int grid2 = (int)std::min<long long>(pairs, per_cu * ctx->cu_count);                             \
            hipLaunchKernelGGL((dts_mlp_bwd_h2_kernel<(DT_) % 2 ? 2 : (DT_), MT_>), dim3(grid2), dim3(64 * MLP_WAVES), lds, ctx->stream, a); \
        } else if (f16) {                                                                                           \
            if (int rc = set_lds(ctx, dts_mlp_bwd_h_kernel<DT_, MT_>, lds)) return rc;                              \
            hipLaunchKernelGGL((dts_mlp_bwd_h_kernel<DT_, MT_>), dim3(grid), dim3(64 * MLP_WAVES), lds, ctx->stream, a); \
        } else {                                                                                                    \
            if (int rc = set_lds(ctx, dts_mlp_bwd_kernel<DT_, MT_>, lds)) return rc;                                \
            hipLaunchKernelGGL((dts_mlp_bwd_kernel<DT_, MT_>), dim3(grid), dim3(64 * MLP_WAVES), lds, ctx->stream, a); \
        }                                                                                                           \
    } while (0)
    if (D == 64 && M == 256) DTS_MLPB(4, 16);
    else if (D == 32 && M == 128) DTS_MLPB(2, 8);
    else if (D == 16 && M == 64) DTS_MLPB(1, 4);
    else return -100;
#undef DTS_MLPB
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

// x_out = x + mlp(ln2(x)); u (pre-GELU hidden) is kept only when the backward pass will read it
int mlp_block(mpv_ctx* ctx, DtsEngine& e, const float* x, const float* lnw, const float* lnb, const float* w0, const float* b0, const float* w2,
              const float* b2, float* u, float* out, long long T, bool save);

inline dim3 g1d(long long n, int threads = 256) { return dim3((unsigned)((n + threads - 1) / threads)); }

#define DTS_TRY(expr)            \
    do {                         \
        if (int rc_ = (expr)) return rc_; \
    } while (0)

// ---- workspace -------------------------------------------------------------------------------------------------------
int carve(mpv_ctx* ctx, DtsEngine& e, int B) {
    if (B <= e.cap_B) return MPV_OK;
    const size_t T = (size_t)B * e.L, D = e.D, F = e.F, H = e.H;
    const size_t BH = (size_t)B * H * e.L;
    size_t total = 0;
    auto take = [&](size_t n) { const size_t at = total; total += (n + 3) & ~size_t(3); return at; };
    std::vector<size_t> o_xe(e.d.n_enc + 1), o_xd(e.d.n_dec + 1);
    for (auto& o : o_xe) o = take(T * D);
    for (auto& o : o_xd) o = take(T * D);
    struct EO { size_t qkv, att, lse, x1, u; };
    struct DO_ { size_t qkv1, att1, lse1, x1, q2, kv2, att2, lse2, x2, u, t3, fidx; };
    std::vector<EO> eo(e.d.n_enc);
    std::vector<DO_> dofs(e.d.n_dec);
    for (auto& o : eo) o = EO{take(T * 3 * D), take(T * D), take(BH), take(T * D), take(T * e.M)};
    for (auto& o : dofs)
        o = DO_{take(T * 3 * D), take(T * D), take(BH), take(T * D), take(T * D), take(T * 2 * D), take(T * D), take(BH), take(T * D),
                take(T * e.M), take((size_t)B * 3 * D), take((size_t)B * e.topk * D)};
    const size_t o_spec = take((size_t)B * 2 * e.NFp * D), o_famp = take((size_t)B * e.topk * D),
                 o_fph = take((size_t)B * e.topk * D), o_x3 = take(T * D), o_ts = take(T * F), o_ss = take(T * D),
                 o_means = take((size_t)B * e.d.n_dec * F), o_trend = take(T * F), o_season = take(T * F);
    const size_t o_gA = take(T * D), o_gB = take(T * D), o_gC = take(T * D), o_gw = take(T * e.M), o_gqkv = take(T * 3 * D), o_gkv = take(T * 2 * D),
                 o_genc = take(T * D), o_delta = take(BH), o_gs = take(T * D), o_gm = take((size_t)B * e.d.n_dec * F),
                 o_gout = take(T * F), o_gx = take(T * F);
    const size_t o_img = take(T * F), o_tgt = take(T * F), o_xs = take(T * F), o_pm = take(T * F), o_nz = take(T * F);
    if (e.ws) {
        MPV_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(e.ws);
        e.ws = nullptr;
        e.cap_B = 0;
    }
    if (hipMalloc(&e.ws, total * sizeof(float)) != hipSuccess) return mpv_fail(ctx, MPV_ENOMEM, "DiffusionTS workspace");
    e.ws_floats = total;
    e.cap_B = B;
    float* w = e.ws;
    e.xe.resize(o_xe.size());
    e.xd.resize(o_xd.size());
    for (size_t i = 0; i < o_xe.size(); ++i) e.xe[i] = w + o_xe[i];
    for (size_t i = 0; i < o_xd.size(); ++i) e.xd[i] = w + o_xd[i];
    e.es.resize(eo.size());
    e.dsv.resize(dofs.size());
    for (size_t i = 0; i < eo.size(); ++i) e.es[i] = DtsEngine::EncS{w + eo[i].qkv, w + eo[i].att, w + eo[i].lse, w + eo[i].x1, w + eo[i].u};
    for (size_t i = 0; i < dofs.size(); ++i) {
        const DO_& o = dofs[i];
        e.dsv[i] = DtsEngine::DecS{w + o.qkv1, w + o.att1, w + o.lse1, w + o.x1, w + o.q2, w + o.kv2, w + o.att2, w + o.lse2, w + o.x2, w + o.u, w + o.t3,
                                   reinterpret_cast<int*>(w + o.fidx)};
    }
    e.spec = w + o_spec; e.famp = w + o_famp; e.fph = w + o_fph; e.x3 = w + o_x3; e.trend_sum = w + o_ts; e.season_sum = w + o_ss;
    e.means = w + o_means; e.trend = w + o_trend; e.season = w + o_season;
    e.gA = w + o_gA; e.gB = w + o_gB; e.gC = w + o_gC; e.gwide = w + o_gw; e.gqkv = w + o_gqkv; e.gkv = w + o_gkv; e.genc = w + o_genc;
    e.delta = w + o_delta; e.gseason = w + o_gs; e.gmeans = w + o_gm; e.gout = w + o_gout; e.gx = w + o_gx;
    e.img = w + o_img; e.target = w + o_tgt; e.xstart = w + o_xs; e.pmean = w + o_pm; e.noise = w + o_nz;
    return MPV_OK;
}

int mlp_block(mpv_ctx* ctx, DtsEngine& e, const float* x, const float* lnw, const float* lnb, const float* w0, const float* b0, const float* w2,
              const float* b2, float* u, float* out, long long T, bool save) {
    const int rc = mlp_fwd(ctx, e.D, e.M, x, w0, b0, w2, b2, lnw, lnb, out, save ? u : nullptr, T, e.d.gemm != MPV_GEMM_F32);
    if (rc != -100) return rc;
    DTS_TRY(lin(ctx, x, e.D, w0, b0, e.D, e.M, T, u, e.M, PRO_LN, lnw, lnb));
    return lin(ctx, u, e.M, w2, b2, e.M, e.D, T, out, e.D, PRO_GELU, nullptr, nullptr, 0, x, e.D);
}

// the auxiliary streams of the decoder's fork (created on first use); false: everything on ctx->stream
bool aux_streams(mpv_ctx* ctx, DtsEngine& e) {
    static const bool off = getenv("MPV_DTS_STREAMS") && atoi(getenv("MPV_DTS_STREAMS")) == 0;
    if (off) return false;
    if (e.aux[0]) return true;
    for (int i = 0; i < 2; ++i) {
        if (hipStreamCreateWithFlags(&e.aux[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e.ev_join[i], hipEventDisableTiming) != hipSuccess) {
            e.aux[0] = e.aux[1] = nullptr;
            return false;
        }
    }
    if (hipEventCreateWithFlags(&e.ev_fork, hipEventDisableTiming) != hipSuccess) { e.aux[0] = e.aux[1] = nullptr; return false; }
    return true;
}
struct StreamSwap {                                   // ctx->stream = s for the scope (the launch helpers read it)
    mpv_ctx* ctx; hipStream_t keep;
    StreamSwap(mpv_ctx* c, hipStream_t s) : ctx(c), keep(c->stream) { c->stream = s; }
    ~StreamSwap() { ctx->stream = keep; }
};

// ---- forward: Transformer.forward (transformer.py:422-438); save = the input-gradient pass follows ----------------------------
int forward(mpv_ctx* ctx, DtsEngine& e, const float* x, int B, int t, bool save) {
    const int L = e.L, F = e.F, D = e.D, H = e.H, M = e.M;
    t_lin_f16 = e.d.gemm != MPV_GEMM_F32;
    const long long T = (long long)B * L;
    const float scale = 1.0f / std::sqrt((float)e.hd);
    const float* ada_t = e.ada + (size_t)t * e.n_ada * 2 * D;
    hipStream_t st = ctx->stream;
    // blocked attention operands: block stride blk = T * 16 floats, every leading dimension 16, head h = block h (K: block H + h, V: 2 H + h)
    const long long blk = e.blocked ? T * 16 : 0, hs = e.blocked ? T * 16 : e.hd;
    const long long qo = e.blocked ? (long long)H * T * 16 : D;         // offset of the K part behind the Q part (V: twice that)
    const int l3 = e.blocked ? 16 : 3 * D, l2 = e.blocked ? 16 : 2 * D, l1 = e.blocked ? 16 : D;
    hipLaunchKernelGGL(dts_embed_fwd_kernel, g1d(T * D), dim3(256), 0, st, x, e.wemb, e.bemb, e.pe_enc, e.pe_dec, L, F, D, T * D, e.xe[0], e.xd[0]);
    MPV_HIP_TRY(ctx, hipGetLastError());
    int ada_i = 0;
    for (int i = 0; i < e.d.n_enc; ++i, ++ada_i) {                       // EncoderBlock.forward :235-239
        const EncW& w = e.enc[i];
        const DtsEngine::EncS& s = e.es[i];
        const float* sc = ada_t + (size_t)ada_i * 2 * D;
        DTS_TRY(lin(ctx, e.xe[i], D, w.wqkv, w.bqkv, D, 3 * D, T, s.qkv, l3, PRO_ADALN, sc, sc + D, 0, nullptr, 0, 0, 0, blk));
        AttnArgs a{s.qkv, s.qkv + qo, s.qkv + 2 * qo, l3, l3, l3, nullptr, nullptr, l1, s.att, l1, nullptr, 0, s.lse, nullptr, H, L, L, e.hd, scale, hs};
        DTS_TRY(attn(ctx, 0, a, B, e.d.gemm != MPV_GEMM_F32));
        DTS_TRY(lin(ctx, s.att, l1, w.wp, w.bp, D, D, T, s.x1, D, PRO_NONE, nullptr, nullptr, 0, e.xe[i], D, 0, blk));
        DTS_TRY(mlp_block(ctx, e, s.x1, w.ln2w, w.ln2b, w.w0, w.b0, w.w2, w.b2, s.u, e.xe[i + 1], T, save));
    }
    const float* encout = e.xe[e.d.n_enc];
    MPV_HIP_TRY(ctx, hipMemsetAsync(e.trend_sum, 0, sizeof(float) * T * F, st));
    MPV_HIP_TRY(ctx, hipMemsetAsync(e.season_sum, 0, sizeof(float) * T * D, st));
    for (int i = 0; i < e.d.n_dec; ++i, ada_i += 2) {                    // DecoderBlock.forward :323-332
        const DecW& w = e.dec[i];
        const DtsEngine::DecS& s = e.dsv[i];
        const float* sc1 = ada_t + (size_t)ada_i * 2 * D;
        const float* sc2 = sc1 + 2 * D;
        DTS_TRY(lin(ctx, e.xd[i], D, w.wqkv1, w.bqkv1, D, 3 * D, T, s.qkv1, l3, PRO_ADALN, sc1, sc1 + D, 0, nullptr, 0, 0, 0, blk));
        AttnArgs a1{s.qkv1, s.qkv1 + qo, s.qkv1 + 2 * qo, l3, l3, l3, nullptr, nullptr, l1, s.att1, l1, nullptr, 0, s.lse1, nullptr, H, L, L, e.hd, scale, hs};
        DTS_TRY(attn(ctx, 0, a1, B, e.d.gemm != MPV_GEMM_F32));
        DTS_TRY(lin(ctx, s.att1, l1, w.wp1, w.bp1, D, D, T, s.x1, D, PRO_NONE, nullptr, nullptr, 0, e.xd[i], D, 0, blk));
        DTS_TRY(lin(ctx, s.x1, D, w.wq2, w.bq2, D, D, T, s.q2, l1, PRO_ADALN, sc2, sc2 + D, 0, nullptr, 0, 0, 0, blk));
        DTS_TRY(lin(ctx, encout, D, w.wkv2, w.bkv2, D, 2 * D, T, s.kv2, l2, PRO_NONE, nullptr, nullptr, 0, nullptr, 0, 0, 0, blk));
        AttnArgs a2{s.q2, s.kv2, s.kv2 + qo, l1, l2, l2, nullptr, nullptr, l1, s.att2, l1, nullptr, 0, s.lse2, nullptr, H, L, L, e.hd, scale, hs};
        DTS_TRY(attn(ctx, 0, a2, B, e.d.gemm != MPV_GEMM_F32));
        DTS_TRY(lin(ctx, s.att2, l1, w.wp2, w.bp2, D, D, T, s.x2, D, PRO_NONE, nullptr, nullptr, 0, s.x1, D, 0, blk));
        // proj (Conv1d(L -> 2L, k=1) over the token axis) folded into its two consumers: trend block on x2 with composite taps, and
        // the selected spectrum as ONE token mix with (DFT basis x proj[L:])
        const bool fork = aux_streams(ctx, e);
        hipStream_t s_trend = st, s_four = st;
        if (fork) {
            MPV_HIP_TRY(ctx, hipEventRecord(e.ev_fork, st));
            MPV_HIP_TRY(ctx, hipStreamWaitEvent(e.aux[0], e.ev_fork, 0));
            MPV_HIP_TRY(ctx, hipStreamWaitEvent(e.aux[1], e.ev_fork, 0));
            s_trend = e.aux[0];
            s_four = e.aux[1];
        }
        hipLaunchKernelGGL(dts_trend_fwd_kernel, dim3(B), dim3(256), sizeof(float) * (15 * D + 3 * F), s_trend, s.x2, (long long)L * D, w.tw0, w.tb0, w.tw3, w.tb3,
                           e.poly, L, F, D, s.t3, e.trend_sum);
        MPV_HIP_TRY(ctx, hipGetLastError());
        if (fork) MPV_HIP_TRY(ctx, hipEventRecord(e.ev_join[0], s_trend));
        {
            static const bool mix_exact = getenv("MPV_DTS_MIX_F32") != nullptr;          // A/B aid: the token mixing on the fp32 matrix pipe
            StreamSwap on_branch(ctx, s_four);                                            // the launch helpers enqueue on ctx->stream
            int rc = -100;
            if (e.d.gemm != MPV_GEMM_F32 && !mix_exact)
                rc = seqmix_h(ctx, D, w.wdft_h, w.wdft_l, w.wdft_isw, 2 * e.NFp, (L + 31) & ~31, w.bdft, s.x2, (long long)L * D, D, L, e.spec, 2LL * e.NFp * D, D,
                              2 * e.NFp, B);
            if (rc == -100) rc = seqmix(ctx, D, w.wdft, 2 * e.NFp, e.Lp, w.bdft, s.x2, (long long)L * D, D, L, e.spec, 2LL * e.NFp * D, D, 2 * e.NFp, B);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(dts_fourier_select_kernel, g1d((long long)B * D), dim3(256), 0, s_four, e.spec, e.NF, e.NFp, D, e.topk, (long long)B * D, s.fidx,
                           e.famp, e.fph);
        hipLaunchKernelGGL(dts_fourier_synth_kernel, dim3(B), dim3(256), 0, s_four, s.fidx, e.famp, e.fph, e.freq + 1, L, D, e.topk, e.season_sum);
        MPV_HIP_TRY(ctx, hipGetLastError());
        if (fork) MPV_HIP_TRY(ctx, hipEventRecord(e.ev_join[1], s_four));
        DTS_TRY(mlp_block(ctx, e, s.x2, w.ln2w, w.ln2b, w.w0, w.b0, w.w2, w.b2, s.u, e.x3, T, save));
        if (fork) {
            MPV_HIP_TRY(ctx, hipStreamWaitEvent(st, e.ev_join[0], 0));
            MPV_HIP_TRY(ctx, hipStreamWaitEvent(st, e.ev_join[1], 0));
        }
        hipLaunchKernelGGL(dts_meansub_fwd_kernel, dim3(B), dim3(256), sizeof(float) * (256 + D), st, e.x3, w.wlin, w.blin, L, F, D, e.xd[i + 1],
                           e.means + (size_t)i * F, e.d.n_dec * F);
        MPV_HIP_TRY(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(dts_tail_fwd_kernel, dim3(B), dim3(256), sizeof(float) * (2 * L * F + 2 * F), st, e.xd[e.d.n_dec], e.season_sum, e.trend_sum, e.means,
                       e.winv, e.binv, e.wcs, e.wcm, L, F, D, e.d.combine_taps, e.d.n_dec, e.trend, e.season);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

// attention backward of one layer: dQ into dq (ldq), dK/dV into dk/dv (ldkv)
int attn_bwd(mpv_ctx* ctx, DtsEngine& e, int B, const float* Q, int ldq, const float* K, const float* V, int ldkv, const float* O, const float* dO,
             float* lse, float* dq, int lddq, float* dk, float* dv, int lddkv) {
    const long long n = (long long)B * e.H * e.L;
    const float scale = 1.0f / std::sqrt((float)e.hd);
    const long long hs = e.blocked ? (long long)B * e.L * 16 : e.hd;
    const int ldo = e.blocked ? 16 : e.D;
    const bool f16 = e.d.gemm != MPV_GEMM_F32 && e.hd == 16 && e.L <= 64 * AH_MAXQ;   // (attn()'s condition) the split-f16 dQ kernel forms delta itself, for dK-dV too
    if (!f16) {
        hipLaunchKernelGGL(dts_attn_delta_kernel, g1d(n), dim3(256), 0, ctx->stream, dO, O, ldo, hs, e.H, e.L, e.hd, n, e.delta);   // n = tokens * heads
        MPV_HIP_TRY(ctx, hipGetLastError());
    }
    AttnArgs a{Q, K, V, ldq, ldkv, ldkv, dO, O, ldo, dq, lddq, nullptr, 0, lse, e.delta, e.H, e.L, e.L, e.hd, scale, hs};
    // Round 4: dQ + dK + dV in ONE pass over the (query tile, key tile) pairs (dts_attn_bwd_h_kernel).  Measured at configs[2] (profiles/r04/
    // c3_attn_bwd_variants.txt): 875 us per launch against 477 + 382 for the two-pass pair -- its 64 dQ-partial registers leave two waves per SIMD where
    // the pair runs five, and at two waves the 15 MFMAs of a pair no longer hide behind the vector work (6.0 cycles per instruction against 4.3);
    // held to three waves it spills the partials (1358 us).  It moves 1 GB less per launch, so it is kept as MPV_DTS_ATTN_ONE_PASS=1; the default is the pair.
    const bool one_pass = getenv("MPV_DTS_ATTN_ONE_PASS") != nullptr;          // (read per call: tests switch it)
    if (f16 && one_pass) {
        a.out = dk; a.ldout = lddkv; a.out2 = dv; a.ldout2 = lddkv; a.out3 = dq; a.ldout3 = lddq;
        return attn_launch_h(ctx, 3, a, B);
    }
    DTS_TRY(attn(ctx, 1, a, B, e.d.gemm != MPV_GEMM_F32));
    a.out = dk; a.ldout = lddkv; a.out2 = dv; a.ldout2 = lddkv;
    DTS_TRY(attn(ctx, 2, a, B, e.d.gemm != MPV_GEMM_F32));
    return MPV_OK;
}

// ---- input gradient: gout = dLoss/d(trend + season_error) [B][L][F] -> gx = dLoss/dx [B][L][F]; uses the saved forward --------
int backward(mpv_ctx* ctx, DtsEngine& e, const float* gout, float* gx, int B, int t) {
    const int L = e.L, F = e.F, D = e.D, M = e.M;
    t_lin_f16 = e.d.gemm != MPV_GEMM_F32;
    const long long T = (long long)B * L;
    const float* ada_t = e.ada + (size_t)t * e.n_ada * 2 * D;
    hipStream_t st = ctx->stream;
    const long long blk = e.blocked ? T * 16 : 0;                       // blocked attention operands, as in forward()
    const long long qo = e.blocked ? (long long)e.H * T * 16 : D;
    const int l3 = e.blocked ? 16 : 3 * D, l2 = e.blocked ? 16 : 2 * D, l1 = e.blocked ? 16 : D;
    // dx += LayerNorm-backward(x, gamma)(G Wt^T): one kernel where the split-f16 Linear has the fused epilogue, else the product into gC and dts_ln_bwd_kernel
    auto lin_ln = [&](const float* G, int ldg, const float* Wt, int K, const float* x, const float* gam, int ada, float* dx) -> int {
        const int rc = lin(ctx, G, ldg, Wt, nullptr, K, D, T, dx, D, PRO_NONE, nullptr, nullptr, 0, nullptr, 0, 1, blk, 0, x, gam, ada);
        if (rc != -100) return rc;
        DTS_TRY(lin(ctx, G, ldg, Wt, nullptr, K, D, T, e.gC, D, PRO_NONE, nullptr, nullptr, 0, nullptr, 0, 0, blk));
        return ln_bwd(ctx, D, e.gC, x, gam, ada, dx, 1, T);
    };
    // tail: d_h (decoder output) -> gA, d_season, d_means
    hipLaunchKernelGGL(dts_tail_bwd_kernel, dim3(B), dim3(256), 0, st, gout, e.winv, e.wcs, e.wcm, L, F, D, e.d.combine_taps, e.d.n_dec, e.gA, e.gseason,
                       e.gmeans);
    MPV_HIP_TRY(ctx, hipGetLastError());
    MPV_HIP_TRY(ctx, hipMemsetAsync(e.genc, 0, sizeof(float) * T * D, st));
    float* gcur = e.gA;                                                  // gradient w.r.t. the current block OUTPUT
    float* gnext = e.gB;
    for (int i = e.d.n_dec - 1; i >= 0; --i) {
        const DecW& w = e.dec[i];
        const DtsEngine::DecS& s = e.dsv[i];
        const int ada_i = e.d.n_enc + 2 * i;
        const float* sc1 = ada_t + (size_t)ada_i * 2 * D;
        const float* sc2 = sc1 + 2 * D;
        // out = x3 - mean(x3), mean_i = Linear(mean): d_x3 -> gnext
        hipLaunchKernelGGL(dts_meansub_bwd_kernel, dim3(B), dim3(256), sizeof(float) * (256 + D), st, gcur, w.wlin, e.gmeans + (size_t)i * F, e.d.n_dec * F, L, F, D, gnext);
        MPV_HIP_TRY(ctx, hipGetLastError());
        std::swap(gcur, gnext);                                          // gcur = d_x3
        // x3 = x2 + mlp(ln2(x2))
        // d_ln = W0^T (gelu'(u) * (W2^T g)), hidden gradient on chip; in split-f16 mode the kernel also finishes ln2's input gradient into gcur
        const int both = mlp_bwd(ctx, D, M, gcur, s.u, w.w2_t, w.w0_t, gcur, T, e.d.gemm != MPV_GEMM_F32, s.x2, w.ln2w);
        if (both != MPV_OK && both != -100) return both;
        const int fused = both == MPV_OK ? MPV_OK : mlp_bwd(ctx, D, M, gcur, s.u, w.w2_t, w.w0_t, e.gC, T, e.d.gemm != MPV_GEMM_F32);
        if (fused == MPV_OK) {
        } else if (fused != -100) {
            return fused;
        } else if (D % G_BK == 0 && M % G_BK == 0) {                           // d_hidden = (g W2) * gelu'(u), then d_ln = d_hidden W0
            DTS_TRY(wide_gemm(ctx, gcur, D, w.w2_t, D, M, T, e.gwide, M, s.u, M));
            DTS_TRY(wide_gemm(ctx, e.gwide, M, w.w0_t, M, D, T, e.gC, D));
        } else {
            DTS_TRY(lin(ctx, gcur, D, w.w2_t, nullptr, D, M, T, e.gwide, M));
            DTS_TRY(lin(ctx, e.gwide, M, w.w0_t, nullptr, M, D, T, e.gC, D, PRO_GELUGRAD, s.u, nullptr, M));
        }
        if (both != MPV_OK) DTS_TRY(ln_bwd(ctx, D, e.gC, s.x2, w.ln2w, 0, gcur, 1, T));     // gcur = d_x2 (so far)
        // trend / season branches through proj
        hipLaunchKernelGGL(dts_trend_bwd_kernel, dim3(B), dim3(256), sizeof(float) * (3 * F + 3 * D), st, gout, s.t3, w.tw0, w.tw3, e.poly, L, F, D, gcur,
                           (long long)L * D);
        hipLaunchKernelGGL(dts_fourier_bwd_kernel, dim3(B), dim3(256), sizeof(float) * (10 * e.topk * D + 2 * L), st, e.gseason, s.fidx, e.cosT, e.sinT, w.wdft_t, L, D,
                           e.topk, e.NFp, gcur, (long long)L * D);
        MPV_HIP_TRY(ctx, hipGetLastError());
        // x2 = x1 + proj2(attn2(q2(ln1_1(x1)), kv2(enc)))
        DTS_TRY(lin(ctx, gcur, D, w.wp2_t, nullptr, D, D, T, e.gC, l1, PRO_NONE, nullptr, nullptr, 0, nullptr, 0, 0, 0, blk));      // d_att2
        DTS_TRY(attn_bwd(ctx, e, B, s.q2, l1, s.kv2, s.kv2 + qo, l2, s.att2, e.gC, s.lse2, e.gqkv, l1, e.gkv, e.gkv + qo, l2));
        DTS_TRY(lin(ctx, e.gkv, l2, w.wkv2_t, nullptr, 2 * D, D, T, e.genc, D, PRO_NONE, nullptr, nullptr, 0, nullptr, 0, 1, blk));
        DTS_TRY(lin_ln(e.gqkv, l1, w.wq2_t, D, s.x1, sc2, 1, gcur));    // gcur = d_x1
        // x1 = x0 + proj1(attn1(qkv1(ln1(x0))))
        DTS_TRY(lin(ctx, gcur, D, w.wp1_t, nullptr, D, D, T, e.gC, l1, PRO_NONE, nullptr, nullptr, 0, nullptr, 0, 0, 0, blk));
        DTS_TRY(attn_bwd(ctx, e, B, s.qkv1, l3, s.qkv1 + qo, s.qkv1 + 2 * qo, l3, s.att1, e.gC, s.lse1, e.gqkv, l3, e.gqkv + qo, e.gqkv + 2 * qo, l3));
        DTS_TRY(lin_ln(e.gqkv, l3, w.wqkv1_t, 3 * D, e.xd[i], sc1, 1, gcur));      // gcur = d_x0 = gradient w.r.t. the previous block's output
    }
    if (e.snapshots)                                                     // "g_enc" of mpv_dts_debug_tensor (gseason is free from here on)
        MPV_HIP_TRY(ctx, hipMemcpyAsync(e.gseason, e.genc, sizeof(float) * T * D, hipMemcpyDeviceToDevice, st));
    float* gdec0 = gcur;                                                 // d(h_dec0)
    float* genc = e.genc;                                                // d(encoder output), accumulated over the decoder blocks
    float* tmp = (gcur == e.gA) ? e.gB : e.gA;
    for (int i = e.d.n_enc - 1; i >= 0; --i) {
        const EncW& w = e.enc[i];
        const DtsEngine::EncS& s = e.es[i];
        const float* sc = ada_t + (size_t)i * 2 * D;
        const int both = mlp_bwd(ctx, D, M, genc, s.u, w.w2_t, w.w0_t, genc, T, e.d.gemm != MPV_GEMM_F32, s.x1, w.ln2w);
        if (both != MPV_OK && both != -100) return both;
        const int fused = both == MPV_OK ? MPV_OK : mlp_bwd(ctx, D, M, genc, s.u, w.w2_t, w.w0_t, e.gC, T, e.d.gemm != MPV_GEMM_F32);
        if (fused == MPV_OK) {
        } else if (fused != -100) {
            return fused;
        } else if (D % G_BK == 0 && M % G_BK == 0) {
            DTS_TRY(wide_gemm(ctx, genc, D, w.w2_t, D, M, T, e.gwide, M, s.u, M));
            DTS_TRY(wide_gemm(ctx, e.gwide, M, w.w0_t, M, D, T, e.gC, D));
        } else {
            DTS_TRY(lin(ctx, genc, D, w.w2_t, nullptr, D, M, T, e.gwide, M));
            DTS_TRY(lin(ctx, e.gwide, M, w.w0_t, nullptr, M, D, T, e.gC, D, PRO_GELUGRAD, s.u, nullptr, M));
        }
        if (both != MPV_OK) DTS_TRY(ln_bwd(ctx, D, e.gC, s.x1, w.ln2w, 0, genc, 1, T));    // genc = d_x1
        DTS_TRY(lin(ctx, genc, D, w.wp_t, nullptr, D, D, T, e.gC, l1, PRO_NONE, nullptr, nullptr, 0, nullptr, 0, 0, 0, blk));
        DTS_TRY(attn_bwd(ctx, e, B, s.qkv, l3, s.qkv + qo, s.qkv + 2 * qo, l3, s.att, e.gC, s.lse, e.gqkv, l3, e.gqkv + qo, e.gqkv + 2 * qo, l3));
        DTS_TRY(lin_ln(e.gqkv, l3, w.wqkv_t, 3 * D, e.xe[i], sc, 1, genc));        // genc = d_x0
    }
    (void)tmp;
    hipLaunchKernelGGL(dts_embed_bwd_kernel, g1d(T * F * 16), dim3(256), 0, st, genc, gdec0, e.wemb, L, F, D, T * F, gx);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

std::vector<float> transposed2(const float* w, int rows, int cols) {
    std::vector<float> t((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols + c];
    return t;
}
std::vector<float> padded(const float* w, int rows, int cols, int rp, int cp) {
    std::vector<float> t((size_t)rp * cp, 0.0f);
    for (int r = 0; r < rows; ++r) std::memcpy(&t[(size_t)r * cp], w + (size_t)r * cols, sizeof(float) * cols);
    return t;
}

}  // namespace

void dts_engine_destroy(DtsEngine* e) {
    if (!e) return;
    if (e->blob) (void)hipFree(e->blob);
    if (e->ws) (void)hipFree(e->ws);
    for (int i = 0; i < 2; ++i) {
        if (e->aux[i]) { (void)hipStreamSynchronize(e->aux[i]); (void)hipStreamDestroy(e->aux[i]); }
        if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    delete e;
}

extern "C" {

int mpv_dts_load(mpv_ctx* ctx, const mpv_dts_desc* d, const float* const* tensors, int32_t n_tensors, const float* ada_tables,
                 const float* rfftfreq, const float* poly) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_dts_load: ctx is NULL");
    if (!d || !tensors || !ada_tables || !rfftfreq || !poly) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_load: NULL argument");
    const int L = d->seq_len, F = d->nf, D = d->d_model, H = d->n_heads;
    if (L < 4 || F < 1 || F > 8 || (D != 16 && D != 32 && D != 64) || H < 1 || D % H != 0 || d->n_enc < 1 || d->n_dec < 1 || d->timesteps < 1 ||
        d->mlp_times < 1 || (d->combine_taps & 1) == 0 || d->combine_taps > L)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_load: need seq_len >= 4, 1 <= nf <= 8, d_model in {16,32,64}, n_heads | d_model, odd combine_taps");
    const int hd = D / H;
    if (hd != 4 && hd != 8 && hd != 16) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_load: head dimension must be 4, 8 or 16");
    const int want = 8 + 14 * d->n_enc + 30 * d->n_dec;
    if (n_tensors != want) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_load: wrong number of weight tensors");
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i]) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_load: NULL weight tensor");
    DeviceGuard dg_(ctx->device);
    MPV_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    dts_engine_destroy(ctx->dts);
    ctx->dts = nullptr;
    DtsEngine* e = new (std::nothrow) DtsEngine();
    if (!e) return mpv_fail(ctx, MPV_ENOMEM, "mpv_dts_load: out of host memory");
    e->d = *d;
    e->L = L; e->F = F; e->D = D; e->H = H; e->hd = hd; e->M = d->mlp_times * D;
    e->blocked = hd == 16 && !getenv("MPV_DTS_ROWMAJOR") && !getenv("MPV_DTS_LIN_TILE");
    e->NF = (L % 2 == 0) ? L / 2 - 1 : (L - 1) / 2;                      // rfft bins 1 .. (L/2 - 1 | (L-1)/2), transformer.py:66-73
    e->NFp = (e->NF + 15) & ~15;
    e->topk = (int)std::log((double)e->NF);                              // int(factor * log(length)), :93
    if (e->topk < 1 || e->topk > DTS_MAX_TOPK) { delete e; return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_load: sequence length gives an unsupported top-k"); }
    e->Lp = (L + 15) & ~15;
    e->L2p = (2 * L + 15) & ~15;
    e->n_ada = d->n_enc + 2 * d->n_dec;
    // ---- host staging of everything, one upload --------------------------------------------------------------------
    std::vector<float> host;
    auto put = [&](const float* p, size_t n) { const size_t at = host.size(); host.insert(host.end(), p, p + n); host.resize((host.size() + 3) & ~size_t(3)); return at; };
    auto putv = [&](const std::vector<float>& v) { return put(v.data(), v.size()); };
    auto cat = [&](std::initializer_list<std::pair<const float*, size_t>> parts) { std::vector<float> v; for (auto& p : parts) v.insert(v.end(), p.first, p.first + p.second); return v; };
    const float* const* T_ = tensors;
    const size_t DD = (size_t)D * D, MD = (size_t)e->M * D;
    std::vector<size_t> o;                                               // offsets in `host`, consumed in the same order below
    std::vector<float> dft_isw;                                          // per decoder block: 1 / (power of two of its split DFT composite)
    o.push_back(put(T_[0], (size_t)D * F * 3)); o.push_back(put(T_[1], D)); o.push_back(put(T_[2], (size_t)F * D * 3)); o.push_back(put(T_[3], F));
    o.push_back(put(T_[4], (size_t)F * D * d->combine_taps)); o.push_back(put(T_[5], d->n_dec)); o.push_back(put(T_[6], (size_t)L * D));
    o.push_back(put(T_[7], (size_t)L * D));
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<double> dcos(L), dsin(L);                                // angles reduced exactly mod L
    for (int m = 0; m < L; ++m) { dcos[m] = std::cos(two_pi * m / L); dsin[m] = std::sin(two_pi * m / L); }
    int k = 8;
    for (int i = 0; i < d->n_enc; ++i, k += 14) {
        // order: ln2.w, ln2.b, key.w, key.b, query.w, query.b, value.w, value.b, proj.w, proj.b, mlp0.w, mlp0.b, mlp2.w, mlp2.b
        const std::vector<float> wqkv = cat({{T_[k + 4], DD}, {T_[k + 2], DD}, {T_[k + 6], DD}});          // rows: query | key | value
        const std::vector<float> bqkv = cat({{T_[k + 5], (size_t)D}, {T_[k + 3], (size_t)D}, {T_[k + 7], (size_t)D}});
        o.push_back(put(T_[k], D)); o.push_back(put(T_[k + 1], D)); o.push_back(putv(wqkv)); o.push_back(putv(bqkv));
        o.push_back(put(T_[k + 8], DD)); o.push_back(put(T_[k + 9], D)); o.push_back(put(T_[k + 10], MD)); o.push_back(put(T_[k + 11], e->M));
        o.push_back(put(T_[k + 12], MD)); o.push_back(put(T_[k + 13], D));
        o.push_back(putv(transposed2(wqkv.data(), 3 * D, D))); o.push_back(putv(transposed2(T_[k + 8], D, D)));
        o.push_back(putv(transposed2(T_[k + 10], e->M, D))); o.push_back(putv(transposed2(T_[k + 12], D, e->M)));
    }
    for (int i = 0; i < d->n_dec; ++i, k += 30) {
        // order: ln2.w, ln2.b, attn1.{key,query,value,proj}.{w,b} (8), attn2.{key,query,value,proj}.{w,b} (8),
        //        trend.0.w, trend.0.b, trend.3.w, trend.3.b, mlp0.w, mlp0.b, mlp2.w, mlp2.b, proj.w, proj.b, linear.w, linear.b
        const float* const* A1 = T_ + k + 2;
        const float* const* A2 = T_ + k + 10;
        const std::vector<float> wqkv1 = cat({{A1[2], DD}, {A1[0], DD}, {A1[4], DD}});
        const std::vector<float> bqkv1 = cat({{A1[3], (size_t)D}, {A1[1], (size_t)D}, {A1[5], (size_t)D}});
        const std::vector<float> wkv2 = cat({{A2[0], DD}, {A2[4], DD}});
        const std::vector<float> bkv2 = cat({{A2[1], (size_t)D}, {A2[5], (size_t)D}});
        o.push_back(put(T_[k], D)); o.push_back(put(T_[k + 1], D));
        o.push_back(putv(wqkv1)); o.push_back(putv(bqkv1)); o.push_back(put(A1[6], DD)); o.push_back(put(A1[7], D));
        o.push_back(putv(transposed2(wqkv1.data(), 3 * D, D))); o.push_back(putv(transposed2(A1[6], D, D)));
        o.push_back(put(A2[2], DD)); o.push_back(put(A2[3], D)); o.push_back(putv(wkv2)); o.push_back(putv(bkv2)); o.push_back(put(A2[6], DD));
        o.push_back(put(A2[7], D));
        o.push_back(putv(transposed2(A2[2], D, D))); o.push_back(putv(transposed2(wkv2.data(), 2 * D, D))); o.push_back(putv(transposed2(A2[6], D, D)));
        // trend.0 (Conv1d L -> 3, k=3 along D) composed with proj[:L]:  Wt[p][l'][tap] = sum_l w0[p][l][tap] Wp[l][l'], and its bias per (p, d):
        // b0[p] + sum_tap [0 <= d+tap-1 < D] sum_l w0[p][l][tap] bp[l]   (zero padding along D drops the proj bias at the borders)
        {
            const float* w0 = T_[k + 18]; const float* b0 = T_[k + 19]; const float* Wp = T_[k + 26]; const float* bp = T_[k + 27];
            std::vector<float> wt((size_t)3 * L * 3), bx((size_t)3 * D);
            for (int p = 0; p < 3; ++p) {
                double cb[3] = {0, 0, 0};
                for (int tap = 0; tap < 3; ++tap) {
                    for (int lp = 0; lp < L; ++lp) {
                        double a = 0.0;
                        for (int l = 0; l < L; ++l) a += (double)w0[((size_t)p * L + l) * 3 + tap] * (double)Wp[(size_t)l * L + lp];
                        wt[((size_t)p * L + lp) * 3 + tap] = (float)a;
                    }
                    for (int l = 0; l < L; ++l) cb[tap] += (double)w0[((size_t)p * L + l) * 3 + tap] * (double)bp[l];
                }
                for (int dd = 0; dd < D; ++dd)
                    bx[(size_t)p * D + dd] = (float)((double)b0[p] + (dd > 0 ? cb[0] : 0.0) + cb[1] + (dd + 1 < D ? cb[2] : 0.0));
            }
            o.push_back(putv(wt)); o.push_back(putv(bx));
        }
        o.push_back(put(T_[k + 20], (size_t)F * D * 3));
        o.push_back(put(T_[k + 21], F));
        o.push_back(put(T_[k + 22], MD)); o.push_back(put(T_[k + 23], e->M)); o.push_back(put(T_[k + 24], MD)); o.push_back(put(T_[k + 25], D));
        o.push_back(putv(transposed2(T_[k + 22], e->M, D))); o.push_back(putv(transposed2(T_[k + 24], D, e->M)));
        // real-DFT basis of bins 1..NF (rows [0, NF) cos, rows [NFp, NFp + NF) -sin) composed with proj[L:] and its bias
        {
            const float* Wp = T_[k + 26] + (size_t)L * L; const float* bp = T_[k + 27] + L;
            std::vector<float> wd((size_t)2 * e->NFp * e->Lp, 0.0f), bd((size_t)2 * e->NFp, 0.0f), wdt((size_t)L * 2 * e->NFp, 0.0f);
            std::vector<double> row(L);
            for (int part = 0; part < 2; ++part)
                for (int kk = 0; kk < e->NF; ++kk) {
                    const int r = part * e->NFp + kk;
                    std::fill(row.begin(), row.end(), 0.0);
                    double bias = 0.0;
                    for (int t = 0; t < L; ++t) {
                        const int m = (int)(((long long)(kk + 1) * t) % L);
                        const double basis = part == 0 ? dcos[m] : -dsin[m];
                        bias += basis * (double)bp[t];
                        const float* wr = Wp + (size_t)t * L;
                        for (int lp = 0; lp < L; ++lp) row[lp] += basis * (double)wr[lp];
                    }
                    bd[r] = (float)bias;
                    for (int lp = 0; lp < L; ++lp) {
                        wd[(size_t)r * e->Lp + lp] = (float)row[lp];
                        wdt[(size_t)lp * 2 * e->NFp + r] = (float)row[lp];
                    }
                }
            o.push_back(putv(wd)); o.push_back(putv(bd)); o.push_back(putv(wdt));
            {   // two-term f16 split of the composite, one power of two for the matrix (max in [2^13, 2^14)), rows padded to a multiple of 32 columns
                const int Lp32 = (L + 31) & ~31, Mp = 2 * e->NFp;
                float mx = 0.0f;
                for (float v : wd) mx = std::fmax(mx, std::fabs(v));
                int ex = 0;
                if (mx > 0.0f) (void)std::frexp(mx, &ex);
                const float sw = mx > 0.0f ? std::ldexp(1.0f, 14 - ex) : 1.0f;
                std::vector<float> planes((size_t)Mp * Lp32, 0.0f);               // 2 planes x Mp x Lp32 halves
                _Float16* ph = reinterpret_cast<_Float16*>(planes.data());
                _Float16* pl = ph + (size_t)Mp * Lp32;
                for (int r = 0; r < Mp; ++r)
                    for (int lp = 0; lp < L; ++lp) {
                        const float x = wd[(size_t)r * e->Lp + lp] * sw;
                        unsigned u; std::memcpy(&u, &x, 4); u &= 0xFFFFE000u;
                        float hi; std::memcpy(&hi, &u, 4);
                        ph[(size_t)r * Lp32 + lp] = (_Float16)hi;
                        pl[(size_t)r * Lp32 + lp] = (_Float16)(x - hi);
                    }
                o.push_back(putv(planes));
                dft_isw.push_back(1.0f / sw);
            }
        }
        o.push_back(put(T_[k + 28], (size_t)F * D)); o.push_back(put(T_[k + 29], F));
    }
    const size_t o_ada = put(ada_tables, (size_t)d->timesteps * e->n_ada * 2 * D);
    const size_t o_freq = put(rfftfreq, (size_t)L / 2 + 1);
    const size_t o_poly = put(poly, (size_t)3 * L);
    std::vector<float> cosT(L), sinT(L);                                  // basis tables of the Fourier layer's transpose
    for (int m = 0; m < L; ++m) { cosT[m] = (float)dcos[m]; sinT[m] = (float)dsin[m]; }
    const size_t o_cos = putv(cosT), o_sin = putv(sinT);
    if (hipMalloc(&e->blob, host.size() * sizeof(float)) != hipSuccess) { delete e; return mpv_fail(ctx, MPV_ENOMEM, "mpv_dts_load: weight blob"); }
    if (hipMemcpy(e->blob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        dts_engine_destroy(e);
        return mpv_fail(ctx, MPV_EHIP, "mpv_dts_load: upload failed");
    }
    size_t at = 0;
    auto nx = [&]() { return (const float*)(e->blob + o[at++]); };
    e->wemb = nx(); e->bemb = nx(); e->winv = nx(); e->binv = nx(); e->wcs = nx(); e->wcm = nx(); e->pe_enc = nx(); e->pe_dec = nx();
    e->enc.resize(d->n_enc);
    for (auto& w : e->enc) {
        w.ln2w = nx(); w.ln2b = nx(); w.wqkv = nx(); w.bqkv = nx(); w.wp = nx(); w.bp = nx(); w.w0 = nx(); w.b0 = nx(); w.w2 = nx(); w.b2 = nx();
        w.wqkv_t = nx(); w.wp_t = nx(); w.w0_t = nx(); w.w2_t = nx();
    }
    e->dec.resize(d->n_dec);
    for (auto& w : e->dec) {
        w.ln2w = nx(); w.ln2b = nx();
        w.wqkv1 = nx(); w.bqkv1 = nx(); w.wp1 = nx(); w.bp1 = nx(); w.wqkv1_t = nx(); w.wp1_t = nx();
        w.wq2 = nx(); w.bq2 = nx(); w.wkv2 = nx(); w.bkv2 = nx(); w.wp2 = nx(); w.bp2 = nx(); w.wq2_t = nx(); w.wkv2_t = nx(); w.wp2_t = nx();
        w.tw0 = nx(); w.tb0 = nx(); w.tw3 = nx(); w.tb3 = nx();
        w.w0 = nx(); w.b0 = nx(); w.w2 = nx(); w.b2 = nx(); w.w0_t = nx(); w.w2_t = nx();
        w.wdft = nx(); w.bdft = nx(); w.wdft_t = nx();
        w.wdft_h = reinterpret_cast<const _Float16*>(nx()); w.wdft_l = w.wdft_h + (size_t)2 * e->NFp * ((L + 31) & ~31);
        w.wdft_isw = dft_isw[&w - e->dec.data()];
        w.wlin = nx(); w.blin = nx();
    }
    e->ada = e->blob + o_ada; e->freq = e->blob + o_freq; e->poly = e->blob + o_poly; e->cosT = e->blob + o_cos;
    e->sinT = e->blob + o_sin;
    ctx->dts = e;
    return MPV_OK;
}

static int dts_check(mpv_ctx* ctx, const char* fn, int32_t B, int32_t t) {
    if (!ctx->dts) return mpv_fail(ctx, MPV_ESTATE, std::string(fn) + ": call mpv_dts_load first");
    if (B < 1 || t < 0 || t >= ctx->dts->d.timesteps) return mpv_fail(ctx, MPV_EINVAL, std::string(fn) + ": need B >= 1 and 0 <= t < timesteps");
    return MPV_OK;
}

int mpv_dts_forward(mpv_ctx* ctx, const float* x, int32_t B, int32_t t, float* trend, float* season) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_dts_forward: ctx is NULL");
    if (int rc = dts_check(ctx, "mpv_dts_forward", B, t)) return rc;
    if (!x || !trend || !season) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_forward: NULL argument");
    DeviceGuard dg_(ctx->device);
    DtsEngine& e = *ctx->dts;
    DTS_TRY(carve(ctx, e, B));
    DTS_TRY(forward(ctx, e, x, B, t, false));
    const size_t n = sizeof(float) * (size_t)B * e.L * e.F;
    MPV_HIP_TRY(ctx, hipMemcpyAsync(trend, e.trend, n, hipMemcpyDeviceToDevice, ctx->stream));
    MPV_HIP_TRY(ctx, hipMemcpyAsync(season, e.season, n, hipMemcpyDeviceToDevice, ctx->stream));
    return MPV_OK;
}

int mpv_dts_vjp(mpv_ctx* ctx, const float* x, int32_t B, int32_t t, const float* g_out, float* g_x) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_dts_vjp: ctx is NULL");
    if (int rc = dts_check(ctx, "mpv_dts_vjp", B, t)) return rc;
    if (!x || !g_out || !g_x) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_vjp: NULL argument");
    DeviceGuard dg_(ctx->device);
    DtsEngine& e = *ctx->dts;
    DTS_TRY(carve(ctx, e, B));
    DTS_TRY(forward(ctx, e, x, B, t, true));
    e.snapshots = true;
    const int rc = backward(ctx, e, g_out, g_x, B, t);
    e.snapshots = false;
    return rc;
}

int mpv_dts_debug_tensor(mpv_ctx* ctx, const char* name, int32_t index, float* out, int64_t n_floats) {
    if (!ctx || !ctx->dts || !name || !out) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_debug_tensor: bad argument");
    DtsEngine& e = *ctx->dts;
    const std::string s(name);
    const float* src = nullptr;
    if (s == "enc_x" && index >= 0 && index <= e.d.n_enc) src = e.xe[index];
    else if (s == "dec_x" && index >= 0 && index <= e.d.n_dec) src = e.xd[index];
    else if (s == "trend_sum") src = e.trend_sum;
    else if (s == "season_sum") src = e.season_sum;
    else if (s == "means") src = e.means;
    else if (s == "g_enc") src = e.gseason;
    if (!src) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_debug_tensor: unknown tensor");
    DeviceGuard dg_(ctx->device);
    MPV_HIP_TRY(ctx, hipMemcpyAsync(out, src, sizeof(float) * (size_t)n_floats, hipMemcpyDeviceToDevice, ctx->stream));
    return MPV_OK;
}

int mpv_dts_sample(mpv_ctx* ctx, const mpv_dts_sample_shape* s, const float* history, const float* plan, const float* noise, float* samples) {
    if (!ctx) return mpv_fail(nullptr, MPV_EINVAL, "mpv_dts_sample: ctx is NULL");
    if (!ctx->dts) return mpv_fail(ctx, MPV_ESTATE, "mpv_dts_sample: call mpv_dts_load first");
    if (!s || !history || !plan || !samples) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_sample: NULL argument");
    DtsEngine& e = *ctx->dts;
    const int L = e.L, F = e.F;
    if (s->n_windows < 1 || s->nodes < 1 || s->n_samples < 1 || s->chunk < 1 || s->n_samples % s->chunk != 0 || s->windows < 1 || s->windows >= L ||
        s->n_pairs < 1)
        return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_sample: need n_windows, nodes >= 1, n_samples a positive multiple of chunk, 1 <= windows < seq_len, n_pairs >= 1");
    DeviceGuard dg_(ctx->device);
    const int par = s->chunk, V = s->nodes, W = s->n_windows;
    const long long rows = (long long)W * par * V;                       // sequences per chunk pass
    if (rows > 0x7fffffffLL / std::max(L, 1)) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_sample: batch too large for one call");
    // Sequences are independent, so a batch larger than the workspace budget runs in blocks of rows (results do not depend on the
    // block size: Philox is keyed by position).  Measured on BASELINE configs[2] (3200 sequences): blocks of 3200 / 1600 / 800 /
    // 400 / 200 rows take 14.9 / 15.3 / 16.6 / 20.4 / 28.9 s per window -- smaller blocks do NOT buy Infinity-Cache residency of the
    // per-layer activations, they only add launches, so the default keeps a whole evaluation_step chunk together (35 KB per token).
    long long blk = s->rows_per_pass > 0 ? s->rows_per_pass : 4096;
    if (const char* env = std::getenv("MPV_DTS_ROWS")) blk = std::max(1, std::atoi(env));
    if ((L * F) % 4) blk = (blk + 3) & ~3LL;                             // Philox blocks of four elements must not straddle row blocks
    blk = std::min(blk, rows);
    DTS_TRY(carve(ctx, e, (int)blk));
    hipStream_t st = ctx->stream;
    const long long n_full = rows * L * F;
    const int Bw = par * V;                                              // batch of ONE evaluation_step chunk: the loss normalisers
    const float inv_nmask = 1.0f / (float)((long long)Bw * s->windows * F);
    const float coef_over_b = s->infill_coef / (float)Bw;
    const unsigned seed_lo = (unsigned)s->seed, seed_hi = (unsigned)(s->seed >> 32);
    // injected noise: per chunk, draws in the reference's order, each [W][Bw][L][F]: img, then per pair with
    // time_next >= 0: `noise` (unused: sigma = 0), K Langevin epsilons (unused: coef_ = 0), q_sample noise
    long long draws_per_chunk = 1;
    for (int pi = 0; pi < s->n_pairs; ++pi)
        if (plan[pi * MPV_DTS_PLAN_COLS + 1] >= 0.0f) draws_per_chunk += 2 + (long long)plan[pi * MPV_DTS_PLAN_COLS + 6];
    for (long long r0 = 0; r0 < rows; r0 += blk) {
        const int B = (int)std::min(blk, rows - r0);
        const long long n = (long long)B * L * F, e0 = r0 * L * F;
        hipLaunchKernelGGL(dts_build_target_kernel, g1d(n), dim3(256), 0, st, history, V, par, L, F, s->windows, r0, n, e.target);
        MPV_HIP_TRY(ctx, hipGetLastError());
        for (int c = 0; c < s->n_samples / par; ++c) {
            long long draw = (long long)c * draws_per_chunk;
            unsigned stream_id = (unsigned)((long long)c * draws_per_chunk);   // chunks never share a Philox stream, however long the Langevin plan
            auto next_noise = [&](bool used, float* dst) -> int {        // consume one draw of the reference's stream
                if (noise) {
                    if (used) MPV_HIP_TRY(ctx, hipMemcpyAsync(dst, noise + draw * n_full + e0, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
                } else if (used) {
                    hipLaunchKernelGGL(dts_philox_kernel, g1d((n + 3) / 4), dim3(256), 0, st, seed_lo, seed_hi, stream_id, e0 / 4, n, dst);
                    MPV_HIP_TRY(ctx, hipGetLastError());
                }
                ++draw;
                ++stream_id;
                return MPV_OK;
            };
            DTS_TRY(next_noise(true, e.img));                            // img = randn(shape), :285
            for (int pi = 0; pi < s->n_pairs; ++pi) {
                const float* pr = plan + pi * MPV_DTS_PLAN_COLS;
                const int time = (int)pr[0];
                const bool last = pr[1] < 0.0f;
                const PairCoef kc{pr[2], pr[3], pr[4], pr[5], pr[8], pr[9]};
                if (time < 0 || time >= e.d.timesteps) return mpv_fail(ctx, MPV_EINVAL, "mpv_dts_sample: plan row with a time outside [0, timesteps)");
                DTS_TRY(forward(ctx, e, e.img, B, time, false));
                hipLaunchKernelGGL(dts_predict_kernel, g1d(n), dim3(256), 0, st, e.img, e.trend, e.season, kc, last ? 1 : 0, n, e.xstart, e.pmean);
                MPV_HIP_TRY(ctx, hipGetLastError());
                if (last) break;
                const float sigma = pr[10];                              // eta sqrt(...) of the pair; 0 for the DDIM sampler every shipped config uses
                DTS_TRY(next_noise(sigma != 0.0f, e.noise));             // `noise` of :300
                // img = pred_mean + sigma noise; Langevin refinement of the whole tensor (:359-407)
                if (sigma != 0.0f) {
                    hipLaunchKernelGGL(dts_add_noise_kernel, g1d(n), dim3(256), 0, st, e.pmean, e.noise, sigma, n, e.img);
                    MPV_HIP_TRY(ctx, hipGetLastError());
                } else {
                    MPV_HIP_TRY(ctx, hipMemcpyAsync(e.img, e.pmean, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
                }
                // sigma != 0: both loss terms of langevin_fn are divided by sigma (:392-395), i.e. their gradients are
                const float inv_nmask_s = sigma != 0.0f ? inv_nmask / sigma : inv_nmask, coef_s = sigma != 0.0f ? coef_over_b / sigma : coef_over_b;
                const int K = (int)pr[6];
                for (int it = 0; it < K; ++it) {
                    DTS_TRY(forward(ctx, e, e.img, B, time, true));
                    hipLaunchKernelGGL(dts_infill_grad_kernel, g1d(n), dim3(256), 0, st, e.trend, e.season, e.target, L, F, s->windows, inv_nmask_s, n, e.gout);
                    MPV_HIP_TRY(ctx, hipGetLastError());
                    DTS_TRY(backward(ctx, e, e.gout, e.gx, B, time));
                    hipLaunchKernelGGL(dts_langevin_step_kernel, g1d(n), dim3(256), 0, st, e.gx, e.pmean, coef_s, pr[7], n, e.img);
                    MPV_HIP_TRY(ctx, hipGetLastError());
                    DTS_TRY(next_noise(false, nullptr));                 // epsilon of :403, multiplied by coef_ = 0
                }
                DTS_TRY(next_noise(true, e.noise));                      // q_sample's noise, :305
                hipLaunchKernelGGL(dts_overwrite_kernel, g1d(n), dim3(256), 0, st, e.img, e.target, e.noise, kc, L, F, s->windows, n, e.img);
                MPV_HIP_TRY(ctx, hipGetLastError());
            }
            const long long no = (long long)B * (L - s->windows) * F;
            hipLaunchKernelGGL(dts_gather_kernel, g1d(no), dim3(256), 0, st, e.xstart, V, par, L, F, s->windows, s->n_samples, c, r0, no, samples);
            MPV_HIP_TRY(ctx, hipGetLastError());
        }
    }
    return MPV_OK;
}

}  // extern "C"
