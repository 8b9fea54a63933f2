// Network SDEs of the paper's three dynamics, Euler-Maruyama on device, for gfx950 (the step before the path).
//
//   biomass  dx_i = [ r x_i (1 - x_i/k) - eps(t) x_i^2/(x_i^2+1) + d sum_j A_ij (x_j - x_i) ] dt + eta dW_i ,  x clamped at 0
//            (reference dataset/spdata_sde_biomass_dynamic_gene.py:53-68; parameters as in paper_figures/common.py:41)
//   SIS      dx_i = [ -d x_i + eps(t) (1 - x_i) sum_j A_ij x_j ] dt + eta dW_i      (dataset/spdata_sde_SIS_dynamic_gene.py:44-52)
//   neuronal dx_i = [ -x_i + (1 - eps(t)) sum_j A_ij sigmoid(tau (x_j - mu)) ] dt + eta dW_i   (Wilson-Cowan,
//            dataset/spdata_sde_neuronal_dynamic_gene.py:50-63)
//   eps(t) a linear ramp over the recorded ticks after a burn-in at eps_start.
//
// Time is inherently serial, the network is small (V <= 1024): ONE workgroup per trajectory keeps the state in
// LDS (double buffered, one barrier per tick), a thread owns up to 4 nodes, the graph is CSR.  Normals are the
// same Philox4x32-10 + Box-Muller stream as the sampler's (counter = (node, tick>>2), element tick&3).  The
// sub-sampling of the harness (sample_time_series, reference diffusion_model_uncertainy.py:128-134) is fused into
// the store: only every `stride`-th tick is written, directly in the harness layout [V][L][1] that the sigma-head
// kernel reads, so a benchmark trajectory never exists at full resolution and never crosses PCIe.
#include "chain_common.h"

namespace {

constexpr int SIM_THREADS = 256;
constexpr int SIM_NPT = 4;                                    // nodes per thread

struct SimArgs {
    const int* row_ptr;
    const int* col_idx;
    const float* x0;        // [n_traj][V]
    float* out;             // [n_traj][V][L]
    int V, ticks, burn, stride, L;
    float r, k, d, dt, eps0, eps_step, noise_scale;     // SIS: d; neuronal: r = tau, k = mu
    unsigned seed_lo, seed_hi;
};

// MODEL 0 biomass, 1 SIS, 2 neuronal (Wilson-Cowan)
template <int MODEL>
__global__ __launch_bounds__(SIM_THREADS) void biomass_em_kernel(const SimArgs p) {
    extern __shared__ float xs[];                             // [2][V]
    const int traj = blockIdx.x;
    int node[SIM_NPT], beg[SIM_NPT], end[SIM_NPT];
    float x[SIM_NPT], zc[SIM_NPT][4];
#pragma unroll
    for (int q = 0; q < SIM_NPT; ++q) {
        node[q] = threadIdx.x + q * SIM_THREADS;
        const bool live = node[q] < p.V;
        beg[q] = live ? p.row_ptr[node[q]] : 0;
        end[q] = live ? p.row_ptr[node[q] + 1] : 0;
        x[q] = live ? p.x0[(long long)traj * p.V + node[q]] : 0.0f;
        if (live) xs[node[q]] = MODEL == 0 ? fmaxf(x[q], 0.0f) : (MODEL == 2 ? 1.0f / (1.0f + expf(-p.r * (x[q] - p.k))) : x[q]);
    }
    __syncthreads();
    const int total = p.burn + p.ticks;
    for (int t = 0; t < total; ++t) {
        const float* cur = xs + (t & 1) * p.V;
        float* nxt = xs + ((t + 1) & 1) * p.V;
        const int i = t - p.burn;                             // recorded tick index (negative during burn-in)
        const float eps = i < 0 ? p.eps0 : p.eps0 + (float)i * p.eps_step;
#pragma unroll
        for (int q = 0; q < SIM_NPT; ++q) {
            if (node[q] < p.V) {
                if ((t & 3) == 0)
                    chain::philox_normals((unsigned long long)traj * p.V + node[q], (unsigned)(t >> 2), p.seed_lo, p.seed_hi, 0u, zc[q]);
                const int e = t & 3;
                const float z = e == 0 ? zc[q][0] : (e == 1 ? zc[q][1] : (e == 2 ? zc[q][2] : zc[q][3]));
                // cur[] holds what the neighbours contribute: the clamped state (biomass), the state (SIS) or its sigmoid (neuronal)
                const float xc = cur[node[q]];
                float ax = 0.0f;
                for (int j = beg[q]; j < end[q]; ++j) ax += cur[p.col_idx[j]];
                {
#pragma clang fp contract(off)
                    float drift;
                    if (MODEL == 0) {
                        const float deg = (float)(end[q] - beg[q]);
                        drift = p.r * xc * (1.0f - xc / p.k) - eps * xc * xc / (xc * xc + 1.0f) + p.d * (ax - deg * xc);
                    } else if (MODEL == 1) {
                        drift = -p.d * x[q] + eps * (1.0f - x[q]) * ax;
                    } else {
                        drift = -x[q] + (1.0f - eps) * ax;
                    }
                    x[q] = x[q] + drift * p.dt + p.noise_scale * z;
                }
                nxt[node[q]] = MODEL == 0 ? fmaxf(x[q], 0.0f) : (MODEL == 2 ? 1.0f / (1.0f + expf(-p.r * (x[q] - p.k))) : x[q]);
                if (i >= 0 && (i % p.stride) == 0) p.out[((long long)traj * p.V + node[q]) * p.L + i / p.stride] = x[q];
            }
        }
        __syncthreads();
    }
}

// ---- SLBP (shallow-lake bream / pike, the single-series sensitivity data set): dataset/SLBP_dynamic_gene.py:26-58 --------------
//   dX = [ i_b + r X N/(N + H1) - c_b X^2 - p_r F_R Y ] dt + sqrt(2 D) dW_1 ,   F_R = X^2 / (X^2 + H4^2)
//   dY = [ i_p + c_e p_r F_R Y V/(V + H2) - m_p Y - c_p Y^2 ] dt + sqrt(0.2 D) dW_2 ,   V = K H3^2 / (X^2 + H3^2)
// The noise is additive, so the reference's Milstein scheme is Euler-Maruyama.  N(t): N_start through the burn-in, then a linear ramp
// to N_end over the recorded ticks.  Two variables and a serial time axis: one THREAD per trajectory.
struct SlbpArgs {
    mpv_slbp_desc d;
    const float* x0;        // [n_traj][2]
    float* out;             // [n_traj][L][2]
    int L;
    float n_step, sx, sy;   // nutrient increment per tick, noise scales sqrt(2 D dt), sqrt(0.2 D dt)
};

__global__ __launch_bounds__(64) void slbp_em_kernel(const SlbpArgs p) {
#pragma clang fp contract(off)
    const int traj = blockIdx.x * 64 + threadIdx.x;
    if (traj >= p.d.n_trajectories) return;
    const mpv_slbp_desc& d = p.d;
    float X = p.x0[2 * traj], Y = p.x0[2 * traj + 1];
    const float h3 = d.H3 * d.H3, h4 = d.H4 * d.H4, kh3 = d.K * h3;
    const int total = d.burn_ticks + d.ticks;
    float zc[4];
    for (int t = 0; t < total; ++t) {
        const int i = t - d.burn_ticks;
        const float N = i < 0 ? d.N_start : d.N_start + (float)i * p.n_step;
        if ((t & 1) == 0) chain::philox_normals((unsigned long long)traj, (unsigned)(t >> 1), (unsigned)d.seed, (unsigned)(d.seed >> 32), 0u, zc);
        const float z1 = (t & 1) ? zc[2] : zc[0], z2 = (t & 1) ? zc[3] : zc[1];
        const float x2 = X * X;
        const float V = kh3 / (x2 + h3);
        const float FR = x2 / (x2 + h4);
        const float dX = ((d.i_b + (d.r * X) * (N / (N + d.H1))) - d.c_b * x2) - (d.p_r * FR) * Y;
        const float dY = ((d.i_p + (((d.c_e * d.p_r) * FR) * Y) * (V / (V + d.H2))) - d.m_p * Y) - d.c_p * (Y * Y);
        X = (X + dX * d.dt) + p.sx * z1;
        Y = (Y + dY * d.dt) + p.sy * z2;
        if (i >= 0 && (i % d.stride) == 0) {
            float* o = p.out + ((long long)traj * p.L + i / d.stride) * 2;
            o[0] = X;
            o[1] = Y;
        }
    }
}

}  // namespace

int launch_slbp_sim(mpv_ctx* ctx, const mpv_slbp_desc& d, const float* x0, float* series_out) {
    SlbpArgs a{};
    a.d = d; a.x0 = x0; a.out = series_out;
    a.L = (d.ticks + d.stride - 1) / d.stride;
    a.n_step = d.ticks > 1 ? (d.N_end - d.N_start) / (float)(d.ticks - 1) : 0.0f;
    a.sx = sqrtf(2.0f * d.D) * sqrtf(d.dt);
    a.sy = sqrtf(0.2f * d.D) * sqrtf(d.dt);
    hipLaunchKernelGGL(slbp_em_kernel, dim3((d.n_trajectories + 63) / 64), dim3(64), 0, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

int launch_biomass_sim(mpv_ctx* ctx, const mpv_biomass_desc& d, const int32_t* row_ptr, const int32_t* col_idx, const float* x0,
                       float* series_out) {
    SimArgs a{};
    a.row_ptr = row_ptr; a.col_idx = col_idx; a.x0 = x0; a.out = series_out;
    a.V = d.nodes; a.ticks = d.ticks; a.burn = d.burn_ticks; a.stride = d.stride;
    a.L = (d.ticks + d.stride - 1) / d.stride;
    a.r = d.r; a.k = d.k; a.d = d.d; a.dt = d.dt; a.eps0 = d.eps_start;
    a.eps_step = d.ticks > 1 ? (d.eps_end - d.eps_start) / (float)(d.ticks - 1) : 0.0f;
    a.noise_scale = d.eta * sqrtf(d.dt);
    a.seed_lo = (unsigned)d.seed; a.seed_hi = (unsigned)(d.seed >> 32);
    hipLaunchKernelGGL(biomass_em_kernel<0>, dim3(d.n_trajectories), dim3(SIM_THREADS), 2 * sizeof(float) * d.nodes, ctx->stream, a);
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}

int launch_network_sde(mpv_ctx* ctx, const mpv_sde_desc& d, const int32_t* row_ptr, const int32_t* col_idx, const float* x0, float* series_out) {
    SimArgs a{};
    a.row_ptr = row_ptr; a.col_idx = col_idx; a.x0 = x0; a.out = series_out;
    a.V = d.nodes; a.ticks = d.ticks; a.burn = d.burn_ticks; a.stride = d.stride;
    a.L = (d.ticks + d.stride - 1) / d.stride;
    a.r = d.p0; a.k = d.p1; a.d = d.p2; a.dt = d.dt; a.eps0 = d.eps_start;
    a.eps_step = d.ticks > 1 ? (d.eps_end - d.eps_start) / (float)(d.ticks - 1) : 0.0f;
    a.noise_scale = d.eta * sqrtf(d.dt);
    a.seed_lo = (unsigned)d.seed; a.seed_hi = (unsigned)(d.seed >> 32);
    const dim3 grid(d.n_trajectories), block(SIM_THREADS);
    const size_t lds = 2 * sizeof(float) * d.nodes;
    switch (d.model) {
        case MPV_SDE_BIOMASS: hipLaunchKernelGGL(biomass_em_kernel<0>, grid, block, lds, ctx->stream, a); break;
        case MPV_SDE_SIS: hipLaunchKernelGGL(biomass_em_kernel<1>, grid, block, lds, ctx->stream, a); break;
        case MPV_SDE_NEURONAL: hipLaunchKernelGGL(biomass_em_kernel<2>, grid, block, lds, ctx->stream, a); break;
        default: return mpv_fail(ctx, MPV_EINVAL, "mpv_network_sde_simulate: unknown model");
    }
    MPV_HIP_TRY(ctx, hipGetLastError());
    return MPV_OK;
}
