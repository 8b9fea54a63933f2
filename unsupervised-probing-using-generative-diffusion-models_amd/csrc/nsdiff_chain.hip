// Fused NsDiff reverse-diffusion chain for gfx950 (MI355X).
//
// One launch runs, for every chain (window w, node b, horizon o, sample s), ALL T steps of
// p_sample_loop (reference NsDiff/nsdiff_utils.py:271-284): per step the conditional MLP
// (NsDiff/denoise.py:35-51), the quadratic posterior (nsdiff_utils.py:111-158) and the
// noise injection, and writes only y_0.  The state y never leaves registers.
//
// Mapping to the hardware
//   * a wave owns 32 chains (lane&31 = chain, consecutive chains = consecutive samples s of
//     one (w,b,o) row, so the final store is a coalesced 128-B segment per feature);
//   * the two 128x128 layers run on the exact-fp32 matrix pipe, v_mfma_f32_32x32x2_f32:
//     D[out][chain] += W[out][k] * H[k][chain].  A = weights (lane&31 = out row of the tile,
//     lane>>5 = k), B = activations (lane&31 = chain, lane>>5 = k).  The accumulator layout
//     (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31) IS the B-operand layout of the
//     next layer once k-steps are enumerated as (tile, reg): register `reg` of half-wave h holds
//     hidden unit 32*tile + (reg&3) + 8*(reg>>2) + 4*h, and a k-step takes the same register
//     from both halves.  So the activations never move between lanes; only the sum of squares
//     for F.normalize crosses the two half-waves (one DPP/permute per layer);
//   * W2 and W3 live in LDS for the lifetime of the workgroup in their native [out][in]
//     layout with a 132-float row stride: one ds_read_b128 per lane fetches the A operands of
//     four consecutive k-steps, bank-conflict free;
//   * layer 1 (K = 3F), the two heads (N = F) and all elementwise work run on the VALU,
//     which the second wave of each SIMD overlaps with the first wave's MFMA phase.
//
// Algorithmic work per chain-step: 2*128*(128+128) + 2*128*3F + 2*2*128*F = 256*(256+5F) flop
// (SURVEY.md section 8d), ~98% of it in the two MFMA layers -> roofline: fp32 MFMA peak.
#include <algorithm>
#include <cstdlib>

#include "chain_common.h"

using namespace chain;

namespace {

// v <- normalize(softplus(gamma * v)) over the 128 hidden units of each chain (TMDM: softplus only,
// reference TMDM/tmdm_model.py:59-61)
template <bool NORMALIZE>
__device__ __forceinline__ void activate(float (&v)[64], const float* __restrict__ gamma_row, int h) {
    float ss = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma_row + hbase(j, q, h));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = softplus_dev(gm[e] * v[j * 16 + q * 4 + e]);
                v[j * 16 + q * 4 + e] = s;
                ss = __builtin_fmaf(s, s, ss);
            }
        }
    }
    if (!NORMALIZE) return;
    ss += __shfl_xor(ss, 32);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize eps
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] *= inv;
}

// TMDM = false: NsDiff (inputs y, y0hat, g; normalised activations; eps and sigma heads; quadratic posterior).
// TMDM = true : the CARD-style sampler of TMDM (inputs y, y0hat; plain softplus; eps head; scalar posterior).
template <int F, bool PHILOX, bool TMDM>
__global__ __launch_bounds__(THREADS, WAVES / 4) void nsdiff_chain_kernel(const ChainArgs p) {
    constexpr int NIN = TMDM ? 2 * F : 3 * F;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    // ---- stage the weights once per workgroup -------------------------------------------
    for (int i = threadIdx.x; i < HID * HID / 4; i += THREADS) {
        const int row = i >> 5, c4 = (i & 31) * 4;
        *reinterpret_cast<f32x4*>(lds + L_W2 + row * WROW + c4) = *reinterpret_cast<const f32x4*>(p.w2 + row * HID + c4);
        *reinterpret_cast<f32x4*>(lds + L_W3 + row * WROW + c4) = *reinterpret_cast<const f32x4*>(p.w3 + row * HID + c4);
    }
    for (int i = threadIdx.x; i < NIN * HID; i += THREADS) lds[L_W1T + i] = p.w1t[i];
    for (int i = threadIdx.x; i < HID; i += THREADS) {
        lds[L_B1 + i] = p.b1[i];
        lds[L_B2 + i] = p.b2[i];
        lds[L_B3 + i] = p.b3[i];
    }
    for (int i = threadIdx.x; i < F * HID; i += THREADS) {
        lds[L_W4 + i] = p.w4[i];
        if (!TMDM) lds[L_WS + i] = p.ws[i];
    }
    if (threadIdx.x < F) {
        lds[L_HB + threadIdx.x] = p.b4[threadIdx.x];
        if (!TMDM) lds[L_HB + 4 + threadIdx.x] = p.bs[threadIdx.x];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r_lane = lane & 31, h_lane = lane >> 5;

    for (;;) {
        const long long tile = claim_tile(p.tile_counter, lane);
        if (tile >= p.n_tiles) break;
        const long long n_raw = tile * TILE + r_lane;
        const bool valid = n_raw < p.n_chains;
        const long long n = valid ? n_raw : p.n_chains - 1;
        const int s = (int)(n % p.S);
        const long long row = n / p.S;            // (w*B + b)*O + o
        const int o = (int)(row % p.O);
        const long long gi = row / p.O;           // w*B + b
        const int b = (int)(gi % p.B);
        const long long w = gi / p.B;

        float g[F], mu[F], y[F], z[4], zc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int f = 0; f < F; ++f) {
            g[f] = TMDM ? 1.0f : p.gx[row * F + f];          // TMDM: y_T = z + mean (tmdm_diffusion_utils.py:109-110)
            mu[f] = p.y0hat ? p.y0hat[row * F + f] : 0.0f;
        }
        long long nbase = 0;
        if (!PHILOX) {
            const int c = s / p.P, pp = s - c * p.P;
            nbase = (((w * p.C + c) * p.T) * ((long long)p.B * p.P) + (long long)b * p.P + pp) * ((long long)p.O * F) +
                    (long long)o * F;
        }
        // y_T = sqrt(g) * z_T + mu       (nsdiff_utils.py:273-274)
        if (PHILOX) {
            step_normals<F>((unsigned long long)n, 0u, p.seed_lo, p.seed_hi, p.off_lo, zc, z);
        } else {
#pragma unroll
            for (int f = 0; f < F; ++f) z[f] = p.noise[nbase + f];
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
#pragma clang fp contract(off)
            y[f] = sqrtf(g[f]) * z[f] + mu[f];
        }

#pragma nounroll
        for (int t = p.T - 1; t >= 0; --t) {
            // Every LDS operand below is loop-invariant in t; left alone, LICM hoists ~600 of
            // them out of the step loop and spills them to scratch.  Re-materialise the lane
            // coordinates through an opaque move each step so the loads stay in the loop.
            int h = h_lane, r = r_lane;
            asm volatile("" : "+v"(h), "+v"(r));
            // the step's normal draw (k = T - t); unused at t == 0
            if (t > 0) {
                if (PHILOX) {
                    step_normals<F>((unsigned long long)n, (unsigned)(p.T - t), p.seed_lo, p.seed_hi, p.off_lo, zc, z);
                } else {
#pragma unroll
                    for (int f = 0; f < F; ++f) z[f] = p.noise[nbase + (long long)(p.T - t) * p.noise_kstride + f];
                }
            }
            // ---- layer 1 on the VALU: h = b1 + W1 . [y, y0hat, g] ---------------------------
            float hv[64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int base = hbase(j, q, h);
                    f32x4 a = *reinterpret_cast<const f32x4*>(lds + L_B1 + base);
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        a += *reinterpret_cast<const f32x4*>(lds + L_W1T + f * HID + base) * y[f];
                        a += *reinterpret_cast<const f32x4*>(lds + L_W1T + (F + f) * HID + base) * mu[f];
                        if (!TMDM) a += *reinterpret_cast<const f32x4*>(lds + L_W1T + (2 * F + f) * HID + base) * g[f];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[j * 16 + q * 4 + e] = a[e];
                }
            }
            activate<!TMDM>(hv, p.e1 + (long long)t * HID, h);
            // ---- layers 2, 3 on the fp32 matrix pipe ---------------------------------------
            mfma_layer(lds + L_W2, lds + L_B2, hv, r, h);
            activate<!TMDM>(hv, p.e2 + (long long)t * HID, h);
            mfma_layer(lds + L_W3, lds + L_B3, hv, r, h);
            activate<!TMDM>(hv, p.e3 + (long long)t * HID, h);
            // ---- heads: eps = W4 h + b4 ; sigma = softplus(Ws softplus(h) + bs) --------------
            float ea[F], sa[F];
#pragma unroll
            for (int f = 0; f < F; ++f) ea[f] = sa[f] = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int base = hbase(j, q, h);
                    float sp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (!TMDM) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) sp[e] = softplus_dev(hv[j * 16 + q * 4 + e]);
                    }
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(lds + L_W4 + f * HID + base);
                        f32x4 wsg = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (!TMDM) wsg = *reinterpret_cast<const f32x4*>(lds + L_WS + f * HID + base);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ea[f] = __builtin_fmaf(w4[e], hv[j * 16 + q * 4 + e], ea[f]);
                            if (!TMDM) sa[f] = __builtin_fmaf(wsg[e], sp[e], sa[f]);
                        }
                    }
                }
            }
            // ---- posterior ----------------------------------------------------------------
            const float* cr = p.coef + t * MPV_COEF_COLS;
            const StepCoef c = {cr[0], cr[1], cr[2],  cr[3],  cr[4],  cr[5],  cr[6],  cr[7],  cr[8],
                                cr[9], cr[10], cr[11], cr[12], cr[13], cr[14], cr[15], cr[16], cr[17]};
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float eps = ea[f] + __shfl_xor(ea[f], 32) + lds[L_HB + f];
                if (TMDM) {
                    y[f] = posterior_tmdm(cr, y[f], mu[f], eps, z[f], t == 0);
                } else {
                    const float sig = softplus_dev(sa[f] + __shfl_xor(sa[f], 32) + lds[L_HB + 4 + f]);
                    y[f] = posterior(c, y[f], mu[f], g[f], eps, sig, z[f], t == 0);
                }
            }
        }
        if (valid && h_lane == 0) {
#pragma unroll
            for (int f = 0; f < F; ++f) p.samples[(row * F + f) * p.S + s] = y[f];
        }
    }
}

template <int F>
__global__ void philox_dump_kernel(float* out, long long n_chains, int T, unsigned seed_lo, unsigned seed_hi,
                                   unsigned off_lo) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_chains) return;
    float z[4], zc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < T; ++k) {                       // same sequential consumption as the sampler
        step_normals<F>((unsigned long long)n, (unsigned)k, seed_lo, seed_hi, off_lo, zc, z);
        for (int f = 0; f < F; ++f) out[(n * T + k) * F + f] = z[f];
    }
}

template <int F, bool TMDM = false>
int launch_f(mpv_ctx* ctx, const ChainArgs& a, bool philox, int grid) {
    const size_t lds_bytes = L_FLOATS * sizeof(float);
    auto kp = nsdiff_chain_kernel<F, true, TMDM>;
    auto kn = nsdiff_chain_kernel<F, false, TMDM>;
    MPV_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(philox ? kp : kn),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (int rc = mpv_reset_tile_counter(ctx)) return rc;
    ChainArgs q = a;
    q.tile_counter = ctx->tile_counter;
    if (philox)
        hipLaunchKernelGGL(kp, dim3(grid), dim3(THREADS), lds_bytes, ctx->stream, q);
    else
        hipLaunchKernelGGL(kn, dim3(grid), dim3(THREADS), lds_bytes, ctx->stream, q);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

}  // namespace

static void fill_model_args(ChainArgs& a, const NsdiffModel& m) {
    a.w1t = m.w1t; a.w2 = m.w2; a.w3 = m.w3; a.b1 = m.b1; a.b2 = m.b2; a.b3 = m.b3;
    a.e1 = m.e1; a.e2 = m.e2; a.e3 = m.e3; a.w4 = m.w4; a.ws = m.ws; a.b4 = m.b4; a.bs = m.bs; a.coef = m.coef;
    a.w2h = m.w2h; a.w2l = m.w2l; a.w3h = m.w3h; a.w3l = m.w3l;
    a.w1img = m.w1img; a.b2p = m.b2p; a.b3p = m.b3p; a.g1 = m.g1; a.g2 = m.g2; a.g3 = m.g3; a.wsl = m.wsl; a.w4h = m.w4h; a.gs2 = m.gs2; a.gs3 = m.gs3;
    a.bone2 = m.bone2; a.bone3 = m.bone3; a.l1_scale = m.l1_scale; a.hi_mask = 0xFFFFE000u;
}

// grow-only context workspace
template <class T>
static int ensure_ws(mpv_ctx* ctx, T** ptr, size_t* have, size_t need, const char* what) {
    if (need <= *have) return MPV_OK;
    if (*ptr) {
        MPV_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(*ptr);
    }
    *ptr = nullptr;
    *have = 0;
    if (hipMalloc(ptr, need) != hipSuccess) return mpv_fail(ctx, MPV_ENOMEM, what);
    *have = need;
    return MPV_OK;
}

__global__ void summary_finalize_kernel(const double* acc, int W, int F, long long R, float* mpv, float* pred_mean, float* mpv_feat) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    double vt = 0.0, mt = 0.0;
    for (int f = 0; f < F; ++f) {
        const double v = acc[(w * F + f) * 2], m = acc[(w * F + f) * 2 + 1];
        if (mpv_feat) mpv_feat[w * F + f] = (float)(v / (double)R);
        vt += v;
        mt += m;
    }
    if (mpv) mpv[w] = (float)(vt / ((double)R * F));
    if (pred_mean) pred_mean[w] = (float)(mt / ((double)R * F));
}

// Both samplers: geometry, the choice between the fused summary epilogue and sample + mpv_reduce, launches over
// window groups when the LDS accumulator table cannot hold every window.
static int run_chain(mpv_ctx* ctx, const NsdiffModel& m, bool tmdm, const mpv_sample_shape& s, const float* y0hat, const float* gx,
                     const float* noise, const mpv_sample_out& out) {
    const int F = m.nf;
    const long long per_window = (long long)s.nodes * s.pred_len * s.n_samples;        // chains
    if ((long long)s.n_windows * per_window == 0) return MPV_OK;
    const bool want_summary = out.mpv || out.pred_mean || out.mpv_feat;
    const bool f16 = s.gemm != MPV_GEMM_F32;
    const int S = s.n_samples;
    const bool s_ok = (S <= 32 && (S & (S - 1)) == 0) || (S % 32 == 0);
    const int cap = (want_summary && f16 && s_ok) ? chain_f16_summary_capacity(ctx, F) : 0;
    const bool fused = cap > 0;
    float* samples = out.samples;
    if (want_summary && !fused && !samples) {
        if (int rc = ensure_ws(ctx, &ctx->sample_ws, &ctx->sample_ws_bytes, sizeof(float) * (size_t)s.n_windows * per_window * F,
                               "sample workspace")) return rc;
        samples = ctx->sample_ws;
    }
    if (fused) {
        const size_t need = sizeof(double) * (size_t)s.n_windows * F * 2;
        if (int rc = ensure_ws(ctx, &ctx->summary_acc, &ctx->summary_acc_bytes, need, "summary accumulators")) return rc;
        MPV_HIP_TRY(ctx, hipMemsetAsync(ctx->summary_acc, 0, need, ctx->stream));
    }
    if (!ctx->chain_stats) {
        MPV_HIP_TRY(ctx, hipMalloc(&ctx->chain_stats, sizeof(unsigned long long)));
        MPV_HIP_TRY(ctx, hipMemsetAsync(ctx->chain_stats, 0, sizeof(unsigned long long), ctx->stream));
    }
    if ((size_t)L_FLOATS * sizeof(float) > ctx->max_lds)
        return mpv_fail(ctx, MPV_EINVAL, "device LDS too small for the resident-weight chain kernel");
    const bool philox = (noise == nullptr);

    struct EventPair {                                                   // destroyed on every early return; handed to ctx->events on success
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~EventPair() {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } ev;
    if (ctx->timing) {
        MPV_HIP_TRY(ctx, hipEventCreate(&ev.e0));
        MPV_HIP_TRY(ctx, hipEventCreate(&ev.e1));
        MPV_HIP_TRY(ctx, hipEventRecord(ev.e0, ctx->stream));
    }
    const int group = fused ? cap : s.n_windows;
    for (int w0 = 0; w0 < s.n_windows; w0 += group) {
        const int wn = (s.n_windows - w0 < group) ? s.n_windows - w0 : group;
        const long long rows0 = (long long)w0 * s.nodes * s.pred_len;               // (w,b,o) rows before this group
        ChainArgs a{};
        fill_model_args(a, m);
        a.y0hat = y0hat ? y0hat + rows0 * F : nullptr;
        a.gx = gx ? gx + rows0 * F : nullptr;
        a.noise = noise ? noise + (long long)w0 * per_window * m.steps * F : nullptr;
        a.samples = samples ? samples + rows0 * F * S : nullptr;
        a.T = m.steps; a.B = s.nodes; a.O = s.pred_len; a.S = S; a.P = s.chunk; a.C = S / s.chunk;
        a.n_chains = (long long)wn * per_window;
        a.n_tiles = (a.n_chains + TILE - 1) / TILE;
        a.noise_kstride = (long long)s.nodes * s.chunk * s.pred_len * F;
        a.seed_lo = (unsigned)s.seed;
        a.seed_hi = (unsigned)(s.seed >> 32) + (unsigned)(s.offset >> 32);
        a.off_lo = (unsigned)s.offset;
        a.chain_base = rows0 * S;
        a.sub = 1;
        a.seg = 1;
        a.stats = ctx->chain_stats;
        if (fused) {
            a.acc = ctx->summary_acc + (size_t)w0 * F * 2;
            a.inv_std = out.std; a.inv_mean = out.mean;
            a.seg = S < 32 ? S : 32;
            a.sub = S <= 32 ? 1 : S / 32;
            a.acc_windows = wn;
        }
        int per_claim = 8;                                               // tiles per work-queue atomic (tuning knob: MPV_CHAIN_CLAIM)
        if (const char* env = getenv("MPV_CHAIN_CLAIM")) per_claim = std::max(1, atoi(env));
        a.claim = f16 ? a.sub * std::max(1, per_claim / a.sub) : 1;     // (the exact-fp32 kernel claims tile by tile)
        const long long wg_needed = ((a.n_tiles + a.claim - 1) / a.claim + WAVES - 1) / WAVES;
        const int grid = (int)(wg_needed < ctx->cu_count ? wg_needed : ctx->cu_count);   // 1 workgroup per CU (LDS-bound)
        int rc;
        if (f16) {
            rc = launch_chain_f16(ctx, a, philox, grid, F, s.gemm == MPV_GEMM_F16X3 ? 3 : (s.gemm == MPV_GEMM_F16X2 ? 2 : 1), tmdm);
            ctx->tile_steps += (unsigned long long)a.n_tiles * m.steps;
        } else if (tmdm) {
            switch (F) {
                case 1: rc = launch_f<1, true>(ctx, a, philox, grid); break;
                case 2: rc = launch_f<2, true>(ctx, a, philox, grid); break;
                case 3: rc = launch_f<3, true>(ctx, a, philox, grid); break;
                case 4: rc = launch_f<4, true>(ctx, a, philox, grid); break;
                default: return mpv_fail(ctx, MPV_EINVAL, "nf must be 1..4");
            }
        } else {
            switch (F) {
                case 1: rc = launch_f<1>(ctx, a, philox, grid); break;
                case 2: rc = launch_f<2>(ctx, a, philox, grid); break;
                case 3: rc = launch_f<3>(ctx, a, philox, grid); break;
                case 4: rc = launch_f<4>(ctx, a, philox, grid); break;
                default: return mpv_fail(ctx, MPV_EINVAL, "nf must be 1..4");
            }
        }
        if (rc != MPV_OK) return rc;
    }
    if (ctx->timing) {
        MPV_HIP_TRY(ctx, hipEventRecord(ev.e1, ctx->stream));
        ctx->events.emplace_back(ev.e0, ev.e1);
        ev.e0 = ev.e1 = nullptr;
    }
    if (!want_summary) return MPV_OK;
    const long long R = (long long)s.nodes * s.pred_len;
    if (fused) {
        hipLaunchKernelGGL(summary_finalize_kernel, dim3((s.n_windows + 63) / 64), dim3(64), 0, ctx->stream, ctx->summary_acc,
                           s.n_windows, F, R, out.mpv, out.pred_mean, out.mpv_feat);
        MPV_HIP_TRY(ctx, hipGetLastError());
        return MPV_OK;
    }
    // unfused: the stand-alone reduction over the sample tensor (it needs both scalar outputs; route missing ones to scratch)
    if (int rc = ensure_ws(ctx, &ctx->summary_acc, &ctx->summary_acc_bytes, sizeof(double) * (size_t)s.n_windows * 2, "summary scratch")) return rc;
    float* scratch = reinterpret_cast<float*>(ctx->summary_acc);
    return launch_mpv_reduce(ctx, samples, s.n_windows, R, F, S, out.std, out.mean, out.mpv ? out.mpv : scratch,
                             out.pred_mean ? out.pred_mean : scratch + s.n_windows, out.mpv_feat);
}

int launch_nsdiff_chain(mpv_ctx* ctx, const mpv_sample_shape& s, const float* y0hat, const float* gx,
                        const float* noise, const mpv_sample_out& out) {
    return run_chain(ctx, ctx->ns, false, s, y0hat, gx, noise, out);
}

int launch_tmdm_chain(mpv_ctx* ctx, const mpv_sample_shape& s, const float* y0hat, const float* noise, const mpv_sample_out& out) {
    return run_chain(ctx, ctx->tm, true, s, y0hat, nullptr, noise, out);
}

int launch_philox_dump(mpv_ctx* ctx, uint64_t seed, uint64_t offset, int64_t n_chains, float* out) {
    const NsdiffModel& m = ctx->ns;
    if (n_chains == 0) return MPV_OK;
    const int threads = 256;
    const dim3 grid((unsigned)((n_chains + threads - 1) / threads)), block(threads);
    const unsigned lo = (unsigned)seed, hi = (unsigned)(seed >> 32) + (unsigned)(offset >> 32), off = (unsigned)offset;
    switch (m.nf) {
        case 1: hipLaunchKernelGGL(philox_dump_kernel<1>, grid, block, 0, ctx->stream, out, (long long)n_chains, m.steps, lo, hi, off); break;
        case 2: hipLaunchKernelGGL(philox_dump_kernel<2>, grid, block, 0, ctx->stream, out, (long long)n_chains, m.steps, lo, hi, off); break;
        case 3: hipLaunchKernelGGL(philox_dump_kernel<3>, grid, block, 0, ctx->stream, out, (long long)n_chains, m.steps, lo, hi, off); break;
        default: hipLaunchKernelGGL(philox_dump_kernel<4>, grid, block, 0, ctx->stream, out, (long long)n_chains, m.steps, lo, hi, off); break;
    }
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}
