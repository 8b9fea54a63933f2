// Shared internals of libmpv_hip (not part of the ABI; the ABI is include/mpv_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "mpv_hip.h"

struct NsdiffModel {
    bool loaded = false;
    int steps = 0, nf = 0;
    float* blob = nullptr;        // one device allocation, sub-pointers below
    const float *w1t, *w2, *w3, *b1, *b2, *b3, *e1, *e2, *e3, *w4, *ws, *b4, *bs, *coef;
    // split-f16 matrix-core variant (see nsdiff_chain_f16.hip)
    float* blob16 = nullptr;
    const unsigned short *w2h, *w2l, *w3h, *w3l, *w1img;
    const unsigned *b2p, *b3p, *gs2, *gs3;
    const float *g1, *g2, *g3, *wsl, *w4h;
    float bone2 = 1.0f, bone3 = 1.0f, l1_scale = 1.0f;
};

struct SigmaModel {
    bool loaded = false;
    mpv_sigma_desc d{};
    float* blob = nullptr;
    // transposed ([in][out]) weights for coalesced reads, LayerNorm affine as given
    const float *w0t, *b0, *ln0w, *ln0b, *w1t, *b1, *ln1w, *ln1b, *w2t, *b2;
};

struct DtsEngine;                      // DiffusionTS weights + workspace (dts.hip)
void dts_engine_destroy(DtsEngine* e);

struct StgEngine;                      // DiffSTG weights + workspace (stg.hip)
void stg_engine_destroy(StgEngine* e);

struct NstEngine;                      // ns-Transformer f(x) weights + workspace (nst.hip)
void nst_engine_destroy(NstEngine* e);

struct DeviceGuard {     // run on ctx->device without disturbing the caller's current device
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

struct mpv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    NsdiffModel ns;
    NsdiffModel tm;                // TMDM sampler weights (fp32 fields only)
    SigmaModel sg;
    DtsEngine* dts = nullptr;
    StgEngine* stg = nullptr;
    StgEngine* nss = nullptr;          // NsDiff_spatial: the same engine type in mode 1 (stg.hip)
    NstEngine* nst = nullptr;
    double* reduce_ws = nullptr;   // partial sums of mpv_reduce
    unsigned long long* tile_counter = nullptr;   // work queue head of the split-f16 chain kernel (zeroed per launch)
    unsigned long long* chain_stats = nullptr;    // [0] steps re-run through the exact routine (device counter)
    unsigned long long tile_steps = 0;            // reverse steps launched on the split-f16 kernel (host count)
    double* summary_acc = nullptr;                // fused summary accumulators [W][F][2]
    size_t summary_acc_bytes = 0;
    float* sample_ws = nullptr;                   // samples of a summary-only call that cannot use the fused epilogue
    size_t sample_ws_bytes = 0;
    size_t reduce_ws_bytes = 0;
    // chain-kernel timing (HIP events on the launch stream)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    int cu_count = 256;
    size_t max_lds = 0;
};

int mpv_fail(mpv_ctx* ctx, int code, const std::string& msg);
int mpv_reset_tile_counter(mpv_ctx* ctx);   // allocate on first use, zero it on the context's stream

#define MPV_HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return mpv_fail(ctx, MPV_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// launchers implemented next to their kernels
int launch_nsdiff_chain(mpv_ctx* ctx, const mpv_sample_shape& s, const float* y0hat, const float* gx,
                        const float* noise, const mpv_sample_out& out);
namespace chain { struct ChainArgs; }
int launch_chain_f16(mpv_ctx* ctx, const chain::ChainArgs& a, bool philox, int grid, int nf, int products, bool tmdm);
int launch_tmdm_chain(mpv_ctx* ctx, const mpv_sample_shape& s, const float* y0hat, const float* noise, const mpv_sample_out& out);
int chain_f16_summary_capacity(const mpv_ctx* ctx, int nf);
int launch_biomass_sim(mpv_ctx* ctx, const mpv_biomass_desc& d, const int32_t* row_ptr, const int32_t* col_idx, const float* x0,
                       float* series_out);
int launch_network_sde(mpv_ctx* ctx, const mpv_sde_desc& d, const int32_t* row_ptr, const int32_t* col_idx, const float* x0, float* series_out);
int launch_slbp_sim(mpv_ctx* ctx, const mpv_slbp_desc& d, const float* x0, float* series_out);
int launch_philox_dump(mpv_ctx* ctx, uint64_t seed, uint64_t offset, int64_t n_chains, float* out);
int launch_sigma_head(mpv_ctx* ctx, const float* series, int V, int64_t L, int n_windows, int64_t first, int64_t step,
                      const float* mean, const float* std, float add_eps, float* gx);
int launch_mpv_reduce(mpv_ctx* ctx, const float* samples, int W, int64_t R, int F, int S, const float* std,
                      const float* mean, float* mpv, float* pred_mean, float* mpv_feat);
