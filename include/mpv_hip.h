/*
 * mpv_hip.h -- C ABI of the MI355X (gfx950) reverse-diffusion MPV sampling library.
 *
 * The reference (curious-child/Unsupervised-probing-Using-Generative-Diffusion-Models) is
 * pure Python/PyTorch and has no FFI; its boundary for this path is the duck-typed model
 * protocol consumed by evaluation_and_analysis/diffusion_model_uncertainy.py
 * (`evaluation_step`, `cond_pred_model_g`, `scaler_transform`).
 * This header is the thin native layer *beneath* that protocol: each entry point names
 * the reference function(s) whose arithmetic it replaces.  The Python host mirror
 * (package `models/Diffusion_model/NsDiff/NsDiff_model.py`) binds it with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MPV_E* code on failure, and never
 *     throws; mpv_last_error() gives the message of the last failure on that context
 *     (or of the last context-less failure when ctx == NULL);
 *   - the caller owns every buffer it passes.  "dev" pointers are device (HBM) pointers
 *     on the context's device, "host" pointers are ordinary host memory;
 *   - work is enqueued on the context's HIP stream and is asynchronous with respect to
 *     the host; the library allocates only context-private workspace (weights, tables,
 *     reduction scratch), freed by mpv_ctx_destroy;
 *   - a context is not thread-safe: one context per (process, device, stream);
 *   - all tensors are fp32, row-major, in the reference's own index order.
 */
#ifndef MPV_HIP_H
#define MPV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPV_ABI_VERSION 4

#define MPV_OK 0
#define MPV_EINVAL (-1)   /* bad argument / unsupported shape            */
#define MPV_ESTATE (-2)   /* model not loaded, wrong call order          */
#define MPV_EHIP (-3)     /* a HIP runtime call failed                   */
#define MPV_ENOMEM (-4)   /* workspace allocation failed                 */

#define MPV_HIDDEN 128     /* ConditionalGuidedModel width, denoise.py:30-32 */
#define MPV_MAX_NF 4       /* features per series the kernels are built for  */
#define MPV_COEF_COLS 20   /* per-step posterior coefficient row, see below  */

/* Evaluation of the two 128x128 layers of the epsilon network (everything else is fp32 VALU):
 *   MPV_GEMM_F16X3  exact two-term fp16 split of weights and activations, three f16 MFMAs with fp32
 *                   accumulation per product (W_hi h_hi + W_hi h_lo + W_lo h_hi; ~2^-21 relative per
 *                   product, i.e. ~1e-6 on the samples).  Default: ~3x faster, inside the 1e-4 parity budget.
 *   MPV_GEMM_F32    v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 FMA chain.                              */
#define MPV_GEMM_F16X3 0
#define MPV_GEMM_F32 1
/* Reduced-precision modes of the same kernel ("bf16 mode reported separately"); fp32 accumulate,
 * fp32 elementwise work, only the MFMA operands are narrower.  Not fp32-equivalent: report their parity beside the
 * number.   MPV_GEMM_F16X2: weights two-term f16, activations ONE f16 (round to nearest): two MFMAs per k-step.
 *           MPV_GEMM_F16  : weights and activations one f16 each: one MFMA per k-step.                       */
#define MPV_GEMM_F16X2 2
#define MPV_GEMM_F16 3

typedef struct mpv_ctx mpv_ctx;

int mpv_abi_version(void);

/* device: HIP device ordinal.  hip_stream: a hipStream_t (e.g. torch's current stream
 * handle) or NULL for the default stream. */
int mpv_ctx_create(int device, void* hip_stream, mpv_ctx** out);
void mpv_ctx_destroy(mpv_ctx* ctx);
int mpv_ctx_set_stream(mpv_ctx* ctx, void* hip_stream);
const char* mpv_last_error(const mpv_ctx* ctx);

/* ------------------------------------------------------------------------------------
 * NsDiff epsilon-network + reverse process
 * replaces: ConditionalGuidedModel.forward (NsDiff/denoise.py:35-51),
 *           NsDiff_net.forward (NsDiff/NsDiff_net.py:163-172),
 *           p_sample / p_sample_t_1to0 / p_sample_loop (NsDiff/nsdiff_utils.py:111-158,
 *           209-239, 271-284) and the tile/un-tile of evaluation_step
 *           (NsDiff/NsDiff_model.py:227-262, 470-494).
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t steps;  /* T = diffusion_steps = rows of the nn.Embedding tables           */
    int32_t nf;     /* F = dataset_nf (1..MPV_MAX_NF)                                  */
} mpv_nsdiff_desc;

/* Host pointers, PyTorch layouts ([out][in] row-major weights), i.e. exactly the
 * tensors of state_dict()["model.diffussion_model.*"]. */
typedef struct {
    const float* lin_w[3];   /* lin{1,2,3}.lin.weight   [128][3F], [128][128], [128][128] */
    const float* lin_b[3];   /* lin{1,2,3}.lin.bias     [128]                              */
    const float* embed[3];   /* lin{1,2,3}.embed.weight [T][128]                           */
    const float* eps_w;      /* lin4.weight      [F][128]                                  */
    const float* eps_b;      /* lin4.bias        [F]                                       */
    const float* sig_w;      /* sigma_lin.weight [F][128]                                  */
    const float* sig_b;      /* sigma_lin.bias   [F]                                       */
} mpv_nsdiff_weights;

/* coef: host [T][MPV_COEF_COLS] fp32, one row per timestep t, built by the host from the
 * schedule tables of NsDiff_net.__init__ (NsDiff_net.py:92-134) with the reference's own
 * fp32 operation order (columns, in terms of nsdiff_utils.py:131-157 names; a=alpha_t,
 * bt/bb = betas_tilde/bar[t], bt1/bb1 = betas_tilde_m_1/bar_m_1[t]):
 *   0 c1   = (1-a)^2*bt1 + a*(1-a)*(bb1-bt1)        lambda_1 = c1*g - sigma*c2
 *   1 c2   = a*bt1 + a*(1-a)
 *   2 q1   = (1-a)^2                                lambda_2 = g*g*q1*q2 - sigma*g*c4
 *   3 q2   = bb1-bt1
 *   4 c4   = a*bb1 - a*bt1 + (1-a)^2
 *   5 l4   = 4*lambda_0,  lambda_0 = a*(1-a)*bt1
 *   6 l2   = 2*lambda_0
 *   7 n1   = bb-bt                                  noise = n1*g + bt*Sigma_y0
 *   8 bt
 *   9 invr = 1/sqrt(1-s^2), s = one_minus_alphas_bar_sqrt[t]
 *  10 omr  = 1-sqrt(1-s^2)
 *  11 aoma = a*(1-a)                                Sigma_1 = q1*g + aoma*Sigma_y0
 *  12 bt1                                           Sigma_2 = q2*g + bt1*Sigma_y0
 *  13 a                                             D = a*Sigma_2 + Sigma_1
 *  14 sp   = sqrt(alphas_cumprod_prev[t])           gamma_0 = sp*Sigma_1/D
 *  15 sa   = sqrt(a)                                gamma_1 = sa*Sigma_2/D
 *  16 k1   = sa*(a-1)                               gamma_2 = (k1*Sigma_2 + k2*Sigma_1)/D
 *  17 k2   = 1-sp
 *  18,19 unused (zero)
 */
int mpv_nsdiff_load(mpv_ctx* ctx, const mpv_nsdiff_desc* desc, const mpv_nsdiff_weights* host_weights,
                    const float* host_coef);

typedef struct {
    int32_t n_windows;    /* W: windows batched in this call (reference: one evaluation_step each) */
    int32_t nodes;        /* B: rows of one window's batch (Node axis; 1 for SLBP)                 */
    int32_t pred_len;     /* O                                                                     */
    int32_t n_samples;    /* S = n_z_samples                                                       */
    int32_t chunk;        /* P = parallel_sample; S % P == 0.  Only fixes the noise draw order.   */
    int32_t gemm;         /* MPV_GEMM_F16X3 (0, default), MPV_GEMM_F32, MPV_GEMM_F16X2 or MPV_GEMM_F16: how the two 128x128 layers are evaluated */
    uint64_t seed;        /* Philox key when noise == NULL                                         */
    uint64_t offset;      /* Philox stream offset (counter word 3) when noise == NULL              */
} mpv_sample_shape;

/* Outputs of a sampling call.  Every pointer is nullable; at least one of `samples` / `mpv` must be given.
 *   samples  : dev [W][B][O][F][S] -- window w's slice is the reference's `outs` [B, pred_len, F, n_z_samples]
 *              (NsDiff_model.py:262-266).  NULL = do not write the sample tensor at all (curve-only runs: the
 *              reference keeps it only to feed summarize_pred_future_list).
 *   mpv      : dev [W]    mean over (b,o,f) of the population variance over s of x = y*std + mean
 *   pred_mean: dev [W]    mean of x over everything          (summarize_pred_future_list,
 *   mpv_feat : dev [W][F] the variance mean per feature       diffusion_model_uncertainy.py:286-303, 529-538)
 *   std/mean : dev [F] inverse scaling applied inside the summary only (_feature_inverse_transform :267-283), or both NULL.
 * With the split-f16 sampler (gemm != MPV_GEMM_F32, nf <= 3) and S a power of two <= 32 or a multiple of 32 the summary
 * is fused into the sampler's epilogue: a wave's 32 chains are whole (b,o,f) rows, sum x and sum x^2 are formed by
 * wavefront shuffles in fp64, accumulated per workgroup in LDS and flushed once per workgroup; nothing but W*F*16 bytes
 * per workgroup leaves the chip when samples == NULL.  Otherwise the library samples into a context-private workspace
 * (or `samples`) and runs mpv_reduce on it. */
typedef struct {
    float* samples;
    float* mpv;
    float* pred_mean;
    float* mpv_feat;
    const float* std;
    const float* mean;
} mpv_sample_out;

/* y0hat : dev [W][B][O][F] prior mean f(x) (= y_T_mean), or NULL for zeros (cond_var / Guassian).
 * gx    : dev [W][B][O][F] prior variance g(x) (already + EPS where the variant adds it).
 * noise : dev, or NULL.  When given it holds every normal draw of the reference in the
 *         reference's own order: for window w, for chunk c < S/P, for draw
 *         k < T (k = 0 is z_T, k >= 1 is the z of step t = T-k), a tensor [B*P][O][F] whose
 *         row b*P+p is sample s = c*P+p of node b:
 *             noise[((((w*(S/P) + c)*T + k)*(B*P) + b*P + p)*O + o)*F + f]
 *         When NULL, normals come from Philox4x32-10 keyed by `seed`, counter
 *         (chain_lo, chain_hi, block, offset) with chain = ((w*B+b)*O+o)*S+s, Box-Muller on
 *         the four output words.  Draw k, feature f uses normal (k*F+f)&3 of block (k*F)>>2
 *         when F is 1 or 2 (all four normals of a block are consumed), normal f of block k
 *         when F is 3 or 4.
 * out   : see mpv_sample_out.
 */
int mpv_nsdiff_sample(mpv_ctx* ctx, const mpv_sample_shape* shape, const float* y0hat, const float* gx,
                      const float* noise, const mpv_sample_out* out);

/* Counters of the split-f16 sampler since the last reset (synchronises the stream): tile_steps = reverse steps executed
 * (one per 32-chain tile per t), safe_steps = how many of them failed the end-of-step range guard and were re-run
 * through the exact out-of-line routine (nsdiff_chain_f16.hip::chain_step_safe). */
int mpv_chain_stats(mpv_ctx* ctx, uint64_t* tile_steps, uint64_t* safe_steps, int reset);

/* ------------------------------------------------------------------------------------
 * TMDM sampler (CARD-style reverse process around a supplied prior mean)
 * replaces: ConditionalGuidedModel.forward with cat_x & cat_y_pred (TMDM/tmdm_model.py:38-64),
 *           p_sample / p_sample_t_1to0 / p_sample_loop (TMDM/tmdm_diffusion_utils.py:57-119) and the
 *           tile/un-tile of TMDM_model.evaluation_step (TMDM/tmdm_adapter.py:131-155).
 * The prior y_0_hat = ns-Transformer VAE mean (third-party torch_timeseries arithmetic) is an INPUT.
 * Weights as mpv_nsdiff_weights with lin_w[0] = [128][2F], embed tables [T+1][128] (rows 0..T-1 used),
 * sig_w / sig_b ignored (may be NULL).
 * coef: host [T][MPV_COEF_COLS]: 0 invr = 1/sqrt(abar_t), 1 omr = 1-sqrt(abar_t), 2 s = sqrt(1-abar_t),
 *       3..5 gamma_0..2, 6 sqrt(beta_hat_t)  (tmdm_diffusion_utils.py:69-88), rest 0.
 * mpv_tmdm_sample: shape->pred_len is the SEQUENCE length the sampler runs on (label_len + pred_len);
 * y0hat dev [W][B][L][F]; noise as in mpv_nsdiff_sample (draw 0 = z_T, then one per step t = T-1..1);
 * out as in mpv_nsdiff_sample (samples dev [W][B][L][F][S]); shape->gemm is MPV_GEMM_F16X3 (nf <= 3; the same tile-pipelined kernel as NsDiff with
 * the normalisation switched off) or MPV_GEMM_F32.
 * ---------------------------------------------------------------------------------- */
int mpv_tmdm_load(mpv_ctx* ctx, const mpv_nsdiff_desc* desc, const mpv_nsdiff_weights* host_weights, const float* host_coef);
int mpv_tmdm_sample(mpv_ctx* ctx, const mpv_sample_shape* shape, const float* y0hat, const float* noise, const mpv_sample_out* out);

/* ------------------------------------------------------------------------------------
 * DiffusionTS: transformer x0-predictor + DDIM / Langevin-infill sampler
 * replaces: Transformer.forward and its blocks (DiffusionTS/diffusionts_transformer.py:12-438, AdaLayerNorm / Conv_MLP of
 *           diffusionts_model_utils.py:158-202), Diffusion_TS.output / model_predictions (DiffusionTS/DiffusionTS.py:147-160),
 *           the input gradient that langevin_fn takes through the transformer by autograd (:359-407), fast_sample_infill
 *           (:277-310), q_sample (:232-237) and the tile / un-tile of DiffusionTS_model.evaluation_step
 *           (DiffusionTS/DiffusionTS_model.py:72-109, including its sample-major tiling but node-major un-tiling).
 * Every dense contraction (QKV / proj / MLP Linear layers, softmax attention forward and backward, the Conv1d(L -> 2L, k=1)
 * over the token axis, the real DFT of the Fourier season layer) runs on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32).
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t seq_len;       /* L = windows + pred_len = n_channel = max_len                         */
    int32_t nf;            /* F = feature_size (1..8)                                              */
    int32_t d_model;       /* n_embd: 16, 32 or 64                                                 */
    int32_t n_heads;       /* d_model / n_heads in {4, 8, 16}                                      */
    int32_t n_enc, n_dec;  /* n_layer_enc, n_layer_dec                                             */
    int32_t mlp_times;     /* mlp_hidden_times                                                     */
    int32_t timesteps;     /* rows of the AdaLayerNorm table                                       */
    int32_t combine_taps;  /* kernel size of combine_s (1 or 5; circular 'same' convolution)       */
    int32_t gemm;          /* MPV_GEMM_F16X3 (0, default): attention products (head width 16) as two-term f16 splits on the f16 matrix pipe,
                              fp32-equivalent; MPV_GEMM_F32: every contraction on the exact-fp32 pipe (ABI v3 field) */
} mpv_dts_desc;

/* tensors: host pointers, PyTorch layouts, state_dict()["model.model.*"] in this order
 *   0 emb.sequential.1.weight [D][F][3]   1 emb...bias [D]   2 inverse.sequential.1.weight [F][D][3]   3 inverse...bias [F]
 *   4 combine_s.weight [F][D][taps]       5 combine_m.weight [n_dec]   6 pos_enc.pe [L][D]   7 pos_dec.pe [L][D]
 *   then per encoder block (14): ln2.{weight,bias}, attn.{key,query,value,proj}.{weight,bias}, mlp.0.{weight,bias}, mlp.2.{weight,bias}
 *   then per decoder block (30): ln2.{weight,bias}, attn1.{key,query,value,proj}.{weight,bias}, attn2.{key,query,value,proj}.{weight,bias},
 *        trend.trend.0.{weight [3][L][3],bias}, trend.trend.3.{weight [F][D][3],bias}, mlp.0.{weight,bias}, mlp.2.{weight,bias},
 *        proj.{weight [2L][L],bias}, linear.{weight [F][D],bias}
 * ada_tables: host [timesteps][n_enc + 2 n_dec][2 D]: cat(scale, shift) = Linear(SiLU(SinusoidalPosEmb(t))) of every AdaLayerNorm
 *   (order: encoder block i ln1; decoder block i ln1, ln1_1), evaluated by the host with the reference's fp32 ops
 *   (diffusionts_model_utils.py:49-61, 187-202).
 * rfftfreq: host [L/2 + 1] = torch.fft.rfftfreq(L).   poly: host [3][L] = TrendBlock.poly_space (transformer.py:26-27). */
int mpv_dts_load(mpv_ctx* ctx, const mpv_dts_desc* desc, const float* const* tensors, int32_t n_tensors, const float* ada_tables,
                 const float* rfftfreq, const float* poly);

/* (trend, season_error) = Transformer.forward(x [B][L][F], t) for one timestep t shared by the batch; all dev, [B][L][F]. */
int mpv_dts_forward(mpv_ctx* ctx, const float* x, int32_t B, int32_t t, float* trend, float* season_error);

/* Vector-Jacobian product of model_output = trend + season_error w.r.t. x: g_x = (d model_output / d x)^T g_out, all dev [B][L][F]
 * (what autograd gives langevin_fn).  The top-k bin SELECTION of the Fourier layer is piecewise constant, as under autograd. */
int mpv_dts_vjp(mpv_ctx* ctx, const float* x, int32_t B, int32_t t, const float* g_out, float* g_x);

/* Validation aid: copy an internal tensor of the last forward / vjp ("enc_x", "dec_x" with a block index; "trend_sum",
 * "season_sum", "means"; after mpv_dts_vjp also "g_enc", the gradient at the encoder output) into out (dev). */
int mpv_dts_debug_tensor(mpv_ctx* ctx, const char* name, int32_t index, float* out, int64_t n_floats);

#define MPV_DTS_PLAN_COLS 11
typedef struct {
    int32_t n_windows;     /* W evaluation_step calls batched in this call                                   */
    int32_t nodes;         /* V = batch rows of one evaluation_step                                           */
    int32_t windows;       /* observed length; pred_len = seq_len - windows                                   */
    int32_t n_samples;     /* S = n_z_samples                                                                 */
    int32_t chunk;         /* par = min(parallel_sample, n_z_samples); S % par == 0                           */
    int32_t n_pairs;       /* rows of `plan` = sampling_timesteps                                             */
    float infill_coef;     /* model_kwargs["coef"] (:87)                                                      */
    int32_t rows_per_pass; /* sequences run through the network together (0 = library default, 4096): results do not depend on it */
    uint64_t seed;         /* Philox key when noise == NULL                                                   */
} mpv_dts_sample_shape;

/* history: dev [W][V][windows][F] scaled windows.  plan: HOST [n_pairs][MPV_DTS_PLAN_COLS], one row per DDIM pair
 * (time, time_next) of fast_sample_infill, built by the host with the reference's fp32 ops: 0 time, 1 time_next,
 * 2 sqrt_recip_alphas_cumprod[time], 3 sqrt_recipm1_alphas_cumprod[time], 4 sqrt(alpha_next), 5 sqrt(1 - alpha_next) (eta = 0),
 * 6 K Langevin iterations, 7 their learning rate, 8 sqrt_alphas_cumprod[time], 9 sqrt_one_minus_alphas_cumprod[time], 10 sigma of the pair
 * (eta sqrt((1 - a/a_next)(1 - a_next)/(1 - a)); 0 for eta = 0, then column 5 is sqrt(1 - alpha_next); otherwise column 5 = sqrt(1 - alpha_next -
 * sigma^2), the step adds sigma * noise and both Langevin loss terms are divided by sigma, DiffusionTS.py:294-300, 392-395).
 * noise: dev or NULL.  When given: for chunk c < S/par, every normal draw of the reference in its order -- img,
 * then per pair with time_next >= 0: the step's `noise` (used when sigma != 0), K Langevin epsilons (drawn but unused: coef_ = 0), q_sample's noise --
 * each draw a tensor [W][par*V][L][F] (the W windows' streams side by side).  NULL: Philox4x32-10 for the draws that are used.
 * samples: dev [W][V][pred_len][F][S]; window w's slice is the reference's `outs`. */
int mpv_dts_sample(mpv_ctx* ctx, const mpv_dts_sample_shape* shape, const float* history, const float* plan, const float* noise,
                   float* samples);

/* ------------------------------------------------------------------------------------
 * DiffSTG: graph U-Net epsilon network + DDIM / DDPM sampler
 * replaces: UGnet.forward and its blocks (DiffSTG/ugnet.py:15-295: TimeEmbedding, TcnBlock, ResidualBlock, Down/Upsample),
 *           torch_geometric.nn.ResGatedGraphConv as selected by models/layer/gnn_conv.py:18-19 (restated from its published
 *           definition: the package is absent, that layer's parity is unpinned), gaussian_denoise_step / gaussian_posterior /
 *           duplicate_edge_index / evaluation_step (DiffSTG/graph_diffusion_model.py:46-101, 193-282).
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t T_h, T_p;      /* history / prediction length; T = T_h + T_p (even)                    */
    int32_t nf;            /* F                                                                    */
    int32_t d_h;           /* base channels = time-embedding width                                 */
    int32_t Td_h;          /* taps the k = T+1 convolution reduces the time axis to (even)         */
    int32_t timesteps;     /* diffusion_steps: rows 0..timesteps of the time-embedding table       */
    int32_t gemm;          /* MPV_GEMM_F16X3 (0, default): the k = T+1 convolutions and the graph-conv Linear maps as two-term f16
                              splits on the f16 matrix pipe, fp32-equivalent; MPV_GEMM_F32: exact fp32 pipe */
} mpv_stg_desc;

/* One step of UGnet.forward's module sequence (down path, middle, up path), host pointers in PyTorch layouts.
 * kind 0: ResidualBlock(c_in -> c_out) at time length T (c_in includes the concatenated skip when pop = 1); w[] =
 *    0 tcn1.conv.weight [co][ci][3][3]  1 tcn1.conv.bias  2 tcn1.shortcut.weight [co][ci] | NULL  3 tcn1.shortcut.bias | NULL
 *    4 t_conv.weight [co][d_h]  5 t_conv.bias  6 tcn2.conv.weight [co][co][3][3]  7 tcn2.conv.bias  8 norm.weight [co]  9 norm.bias
 *    10 downsampling.weight [co][co][1][T+1]  11 downsampling.bias
 *    12..17 spatial.gnn.lin_key / lin_query / lin_value .weight [C][C], .bias [C] (C = Td_h co)   18 lin_skip.weight [C][C]   19 gnn bias [C]
 *    20 upsampling.weight [co][co][1][T+1] (ConvTranspose2d: [in][out])  21 upsampling.bias  22 shortcut.weight [co][ci] | NULL  23 shortcut.bias | NULL
 * kind 1: Downsample conv (1,3) stride 2 pad 1: w[0] weight [c][c][1][3], w[1] bias.   kind 2: Upsample ConvTranspose (1,4) stride 2 pad 1:
 *    w[0] weight [c][c][1][4] ([in][out]), w[1] bias.
 * push: the step's output is appended to the skip stack (down path); pop: the block concatenates the popped skip behind its input.
 *
 * Round 4 (additive: no layout change): the other layers models/layer/gnn_conv.py:4-28 can put into a ResidualBlock's SpatialBlock, as further
 * residual-block kinds.  Everything but w[12..19] is as for kind 0; C = Td_h co; restated from torch_geometric's published definitions (unpinned):
 * kind MPV_STG_RES_SAGE_MEAN  SAGEConv(aggr="mean"): out_i = lin_l(mean_{j->i} x_j) + lin_r(x_i)
 * kind MPV_STG_RES_SUM        GraphConv(aggr="add") / SAGEConv(aggr="add" | "sum"): out_i = lin_rel(sum_{j->i} x_j) + lin_root(x_i)
 *      both: w[12] lin_l | lin_rel .weight [C][C]   w[13] its bias [C] | NULL   w[14] lin_r | lin_root .weight [C][C] | NULL (root_weight=False)
 * kind MPV_STG_RES_GCN        GCNConv (add_self_loops, symmetric normalisation by in-degree + 1):  w[12] lin.weight [C][C]   w[13] bias [C] | NULL
 * kind MPV_STG_RES_GAT        GATConv(heads=H, concat=True, add_self_loops=True), eval mode:  w[12] lin.weight [C][C] (row h D + d: head h)
 *      w[13] bias [C] | NULL   w[14] att_src [H][C/H]   w[15] att_dst [H][C/H]   w[16] -> two floats { H, negative_slope }
 * The block applies ReLU behind the layer (SpatialBlock.forward, DiffSTG/ugnet.py:36-46) in every kind. */
#define MPV_STG_RES_SAGE_MEAN 10
#define MPV_STG_RES_SUM 11
#define MPV_STG_RES_GCN 12
#define MPV_STG_RES_GAT 13
typedef struct {
    int32_t kind, c_in, c_out, T, pop, push;
    const float* w[24];
} mpv_stg_step;

/* head: host { x_proj.weight [d_h][F], x_proj.bias, out.0.weight [F][d_h], out.0.bias, out.1.weight [T][2T], out.1.bias }.
 * temb: host [timesteps + 1][d_h] = TimeEmbedding(t, d_h) for t = 0..timesteps, evaluated with the reference's fp32 ops (ugnet.py:15-33). */
int mpv_stg_load(mpv_ctx* ctx, const mpv_stg_desc* desc, const mpv_stg_step* steps, int32_t n_steps, const float* const* head, const float* temb);

/* The graph of ONE window (host int64, torch_geometric convention: edge k goes edge_src[k] -> edge_dst[k], messages are summed at the
 * target, an undirected edge is listed in both directions).  Stored as CSR by target; every (window, parallel sample) copy reuses it. */
int mpv_stg_set_graph(mpv_ctx* ctx, int32_t n_nodes, int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst);

/* eps = UGnet(x_t, t, (x_masked, edges)): xt, x_masked, eps_out dev [n_rows][T][F], n_rows a multiple of the graph's node count. */
int mpv_stg_forward(mpv_ctx* ctx, const float* xt, const float* x_masked, int64_t n_rows, int32_t t, float* eps_out);

#define MPV_STG_PLAN_COLS 6
typedef struct {
    int32_t n_windows;     /* W evaluation_step calls batched                                       */
    int32_t parallel;      /* parallel_sampling P: graph copies per window in one batch             */
    int32_t sequential;    /* sequential_sampling S: passes                                         */
    int32_t n_steps;       /* inference_diffusion_steps = rows of plan                              */
    uint64_t seed;
} mpv_stg_sample_shape;

/* history: dev [W][V][T_h][F] scaled.  plan: HOST float64 [n_steps][6]: 0 t1, 1 ddpm flag, 2..4 the posterior's Python-float
 * coefficients (ddim: x' = c1 (x - c2 eps) + c3 eps; ddpm: x' = c1 (x - c2 eps) + c3 z), 5 t2 (informative).
 * noise: dev or NULL: per sequential pass the draws of the reference in order (x_T, then one z per DDPM step), each [W][P*V][T][F].
 * samples: dev [W][V][T][F][P*S], sample index s*P + p; window w's slice is the reference's splitted_predict_x0 (F = 1). */
int mpv_stg_sample(mpv_ctx* ctx, const mpv_stg_sample_shape* shape, const float* history, const double* plan, const float* noise, float* samples);

/* ------------------------------------------------------------------------------------
 * NsDiff_spatial: NsDiff's graph U-Net epsilon / sigma network + the NsDiff reverse loop
 * replaces: UGnet.forward of models/Diffusion_model/NsDiff/ugnet.py:194-292 (x_proj over cat(y_t, y_0_hat, g), the ResidualBlock / Down /
 *           Upsample sequence of mpv_stg_step at time length pred_len, out = Conv2d(d_h, d_h, 1x1) + Linear(T, T), eps = lin4(.),
 *           sigma = softplus(sigma_lin(softplus(.)))), NsDiff_net_spatial.forward (NsDiff_net.py:175-260), and the sampling part of
 *           NsDiff_model_spatial.evaluation_step (NsDiff_model.py:695-790): node-major row tiling (row v*P + p) against the copy-major
 *           duplicated edge index (copy r / V, node r % V: reproduced as the reference has it), p_sample_loop with p_sample /
 *           p_sample_t_1to0 (nsdiff_utils.py:111-158, 209-239, 271-284).  f(x) (Model_spatial) is an input: y0hat.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t pred_len;      /* T = O: the sampled horizon, a multiple of 4 (and even at every U-Net resolution) */
    int32_t nf;            /* F                                                                    */
    int32_t d_h;           /* base channels = time-embedding width                                 */
    int32_t Td_h;          /* taps the k = T+1 convolution reduces the time axis to (even)         */
    int32_t steps;         /* diffusion_steps: timesteps 0..steps-1                                */
    int32_t gemm;          /* as mpv_stg_desc.gemm                                                 */
} mpv_nss_desc;

/* steps: as mpv_stg_load (time length pred_len at the top resolution).  head: host { x_proj.weight [d_h][3F], x_proj.bias, out.0.weight
 * [d_h][d_h], out.0.bias, out.1.weight [T][T], out.1.bias, lin4.weight [F][d_h], lin4.bias, sigma_lin.weight [F][d_h], sigma_lin.bias,
 * coef [steps][MPV_COEF_COLS] (the rows of mpv_nsdiff_load) }.  temb: host [steps][d_h] = TimeEmbedding(t, d_h), t = 0..steps-1. */
int mpv_nss_load(mpv_ctx* ctx, const mpv_nss_desc* desc, const mpv_stg_step* steps, int32_t n_steps, const float* const* head, const float* temb);

/* The graph of ONE window, as mpv_stg_set_graph. */
int mpv_nss_set_graph(mpv_ctx* ctx, int32_t n_nodes, int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst);

/* (eps, sigma) = UGnet(y_t, y_0_hat, g, t, edges): all dev [n_rows][T][F], n_rows a multiple of the graph's node count (consecutive
 * blocks of n_nodes rows are the graph copies). */
int mpv_nss_forward(mpv_ctx* ctx, const float* y, const float* y0hat, const float* gx, int64_t n_rows, int32_t t, float* eps_out, float* sigma_out);

typedef struct {
    int32_t n_windows;     /* W evaluation_step calls batched                                       */
    int32_t nodes;         /* V = the graph's node count                                            */
    int32_t n_samples;     /* S = n_z_samples                                                       */
    int32_t chunk;         /* P = parallel_sample: graph copies per window in one pass; S % P == 0  */
    uint64_t seed;
    int64_t rows_per_pass; /* ABI v3: windows are processed in blocks of at most this many node rows
                              (P*V rows per window, at least one window per block); 0 = the library's
                              default (32768).  Draws are keyed by (window, chunk, position), so the
                              result does not depend on the blocking.                                */
} mpv_nss_sample_shape;

/* y0hat, gx: dev [W][V][O][F] (f(x) and the sigma head's g(x) per node).  noise: dev or NULL: the reference's draws in its order, laid out as
 * for mpv_nsdiff_sample: [W][S/P][steps][V*P][O][F] (draw 0 = z_T, then one per step t = steps-1 .. 1, each AFTER the model call);
 * NULL: Philox4x32-10.  samples: dev [W][V][O][F][S], sample index c*P + p; window w's slice is the reference's `outs`. */
int mpv_nss_sample(mpv_ctx* ctx, const mpv_nss_sample_shape* shape, const float* y0hat, const float* gx, const float* noise, float* samples);

/* ------------------------------------------------------------------------------------
 * Synthetic input generation on device (the step before the path)
 * replaces: the harvested-biomass network simulator dataset/spdata_sde_biomass_dynamic_gene.py:53-68 (drift f, noise g;
 *           there integrated by torchsde) + normalize_time_series / sample_time_series of the harness
 *           (diffusion_model_uncertainy.py:106-134): Euler-Maruyama with step dt, burn_ticks at eps_start, then `ticks`
 *           recorded ticks along a linear ramp eps_start -> eps_end; every `stride`-th recorded tick is stored.
 * row_ptr / col_idx: dev CSR adjacency of the undirected graph (int32, [nodes+1] / [nnz]); x0 dev [n_traj][nodes];
 * series_out dev [n_traj][nodes][ceil(ticks/stride)] = the harness's resident layout [V][L][F=1].
 * Normal of (trajectory, node, tick) = element tick&3 of Philox block (chain = traj*nodes + node, k = tick>>2), the
 * generator of mpv_nsdiff_philox_normals with offset 0.  nodes <= 1024.
 * ---------------------------------------------------------------------------------- */
typedef struct mpv_biomass_desc {
    int32_t n_trajectories, nodes, ticks, burn_ticks, stride;
    float r, k, d, eta, dt, eps_start, eps_end;
    uint64_t seed;
} mpv_biomass_desc;
int mpv_biomass_simulate(mpv_ctx* ctx, const mpv_biomass_desc* desc, const int32_t* row_ptr, const int32_t* col_idx,
                         const float* x0, float* series_out);

/* The same generator for all three network dynamics of the paper.  model: MPV_SDE_BIOMASS (p0 = r, p1 = k, p2 = d: the call above),
 * MPV_SDE_SIS (p2 = d: dx = [-d x + eps(t) (1 - x) A x] dt + eta dW, dataset/spdata_sde_SIS_dynamic_gene.py:44-52),
 * MPV_SDE_NEURONAL (p0 = tau, p1 = mu: dx = [-x + (1 - eps(t)) A sigmoid(tau (x - mu))] dt + eta dW, Wilson-Cowan,
 * dataset/spdata_sde_neuronal_dynamic_gene.py:50-63).  Same graph / Philox / sub-sampling conventions. */
#define MPV_SDE_BIOMASS 0
#define MPV_SDE_SIS 1
#define MPV_SDE_NEURONAL 2
typedef struct mpv_sde_desc {
    int32_t model, n_trajectories, nodes, ticks, burn_ticks, stride;
    float p0, p1, p2, eta, dt, eps_start, eps_end;
    uint64_t seed;
} mpv_sde_desc;
int mpv_network_sde_simulate(mpv_ctx* ctx, const mpv_sde_desc* desc, const int32_t* row_ptr, const int32_t* col_idx, const float* x0,
                             float* series_out);

/* ------------------------------------------------------------------------------------
 * f(x): the mean backbone of NsDiff, and TMDM's prior
 * replaces: ns_Transformer.Model.forward (NsDiff/mu_backbone.py:150-183, Projector :12-41), called at NsDiff_model.py:222 /
 *           :442-446 once per window, and TMDM's prior (TMDM/tmdm_ns_transformer.py:136-172, called at tmdm_adapter.py:123)
 *           in eval mode; with scaler_transform and build_sliding_windows folded into the load like mpv_sigma_head.
 * The encoder / decoder / attention / embedding layers are torch-timeseries==0.1.10 classes the reference only imports;
 * they are restated from the published Non-stationary Transformer (oracle/ns_transformer_oracle.py): PARITY UNPINNED.
 *
 * tensors (host, fp32, state_dict order of the `cond_pred_model.` sub-module):
 *   enc_embedding.value_embedding.tokenConv.weight [D][F][3], enc_embedding.position_embedding.pe [pe_rows][D], the same two of
 *   dec_embedding; per encoder layer: attention.{query,key,value,out}_projection.{weight,bias} (8), conv1.{weight [d_ff][D],bias},
 *   conv2.{weight [D][d_ff],bias}, norm1.{weight,bias}, norm2.{weight,bias}; encoder.norm.{weight,bias}; per decoder layer:
 *   self_attention (8), cross_attention (8), conv1, conv2 (4), norm1..3 (6); decoder.norm (2); decoder.projection.{weight,bias};
 *   tau_learner: series_conv.weight [seq_len][3], then backbone.{0,2,..}.{weight,bias} for the hidden layers and the bias-free
 *   output weight; delta_learner likewise; with latent: z_mean.0, z_mean.2, z_out.0, z_out.2 {weight,bias}.
 * ---------------------------------------------------------------------------------- */
typedef struct mpv_nst_desc {
    int32_t enc_in, c_out;            /* F (c_out must equal enc_in: the output is de-normalised with the input statistics) */
    int32_t seq_len, label_len, pred_len;
    int32_t d_model, n_heads, d_ff, e_layers, d_layers;
    int32_t activation;               /* 0 relu, 1 gelu (erf) */
    int32_t p_layers;                 /* p_hidden_layers, 1..4 */
    int32_t p_hidden[4];              /* p_hidden_dims */
    int32_t latent;                   /* 1: TMDM prior (encoder output -> z_mean -> z_out -> decoder) */
    int32_t pe_rows;                  /* rows of the two position tables passed */
    int32_t fT_h;                     /* ABI v3.  > 0 (even): Model_spatial -- the encoder output goes through Conv2d(D, D, (1, seq_len + 1),
                                         pad fT_h / 2) to fT_h steps, `spatial_layers` x relu(ResGatedGraphConv(fT_h D -> fT_h D)) over the graph of
                                         mpv_nst_set_graph, and ConvTranspose2d back to seq_len steps before the decoder cross-attends to it
                                         (NsDiff/mu_backbone.py:186-340, called at NsDiff_model.py:744).  0: the plain Model.               */
    int32_t spatial_layers;
    int32_t spatial_gnn;              /* ABI v4.  The graph convolution of the spatial layers (f_gnn_name): 0 ResGatedGraphConv, or MPV_STG_RES_SAGE_MEAN /
                                         _SUM / _GCN / _GAT with the same 8 tensor slots per layer as mpv_stg_step.w[12..19] (NULL where the kind has none). */
    int32_t spatial_heads;            /* GATConv heads (f_gnn_param["heads"]); otherwise ignored                                         */
    float spatial_slope;              /* GATConv negative_slope                                                                            */
} mpv_nst_desc;
/* With fT_h > 0 the tensor list continues: downsampling.{weight [D][D][1][seq_len+1], bias [D]}; per spatial layer spatial_encoder.N.gnn.
 * {lin_key,lin_query,lin_value}.{weight [C][C], bias [C]}, lin_skip.weight [C][C], bias [C] (C = fT_h D; torch_geometric's ResGatedGraphConv
 * parameter names: the package is absent, the layer is restated -- unpinned like the rest of f(x)); upsampling.{weight [D][D][1][seq_len+1]
 * (ConvTranspose2d: [in][out]), bias [D]}. */
int mpv_nst_load(mpv_ctx* ctx, const mpv_nst_desc* desc, const float* const* tensors, int32_t n_tensors);

/* Model_spatial only: the graph its convolutions run over -- replaces the `edge_index` argument of Model_spatial.forward (mu_backbone.py:296).
 * edge_src / edge_dst: host [n_edges] (edge_index[0] = source j, edge_index[1] = target i).  Every window of an mpv_nst_forward call is one
 * copy of this graph: `nodes` there must equal n_nodes. */
int mpv_nst_set_graph(mpv_ctx* ctx, int32_t n_nodes, int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst);

/* series: dev [nodes][series_len][F]; window w covers [first + w*step, +seq_len) of every node; mean/std: dev [F] scaler fused into
 * the load or both NULL.  y0hat: dev [n_windows*nodes][pred_len][F] (full = 0: the reference's first return value) or
 * [n_windows*nodes][label_len+pred_len][F] (full = 1: its second).  gemm: MPV_GEMM_F16X3 (two-term f16 split of weights and
 * activations, fp32-equivalent, 3 f16 MFMAs per product) or MPV_GEMM_F32 (exact fp32 pipe) for the Linear layers; attention,
 * LayerNorm and everything else are fp32 either way.  rows_per_pass: window rows per pass over the layers (0 = sized to a 6 GB
 * workspace). */
int mpv_nst_forward(mpv_ctx* ctx, const float* series, int32_t nodes, int64_t series_len, int32_t n_windows, int64_t first, int64_t step,
                    const float* mean, const float* std, int32_t full, int32_t gemm, int64_t rows_per_pass, float* y0hat);

/* Validation helper: intermediate of the LAST pass of the last mpv_nst_forward call.  which: 0 normalised window [B][S][F],
 * 1 statistics [B][2][F] (mean, std), 2 tau [B], 3 delta [B][S], 4 what the decoder cross-attends to [B][S][D] (the encoder output; after the
 * latent bottleneck / the spatial stage when present), 5 (Model_spatial) the features after the last graph convolution [B][fT_h][D]. */
int mpv_nst_debug_tensor(mpv_ctx* ctx, int32_t which, float* out_dev, int64_t capacity);

/* The single-series generator of the SLBP sensitivity data sets (shallow-lake bream X / pike Y; dataset/SLBP_dynamic_gene.py:26-58, parameter
 * table :10-24): dX = [i_b + r X N/(N+H1) - c_b X^2 - p_r F_R Y] dt + sqrt(2D) dW, dY = [i_p + c_e p_r F_R Y V/(V+H2) - m_p Y - c_p Y^2] dt +
 * sqrt(0.2 D) dW, F_R = X^2/(X^2+H4^2), V = K H3^2/(X^2+H3^2).  Additive noise: the reference's Milstein scheme reduces to Euler-Maruyama.
 * N(t) = N_start through burn_ticks, then a linear ramp to N_end over `ticks`.  x0: dev [n_trajectories][2]; series_out: dev
 * [n_trajectories][ceil(ticks/stride)][2] (every stride-th tick, the harness's [T][F] single-series layout). */
typedef struct mpv_slbp_desc {
    int32_t n_trajectories, ticks, burn_ticks, stride;
    float i_b, i_p, r, H1, H2, H3, H4, c_b, c_p, p_r, c_e, m_p, K, D;
    float dt, N_start, N_end;
    uint64_t seed;
} mpv_slbp_desc;
int mpv_slbp_simulate(mpv_ctx* ctx, const mpv_slbp_desc* desc, const float* x0, float* series_out);

/* Debug/validation helper: write the Philox normals mpv_nsdiff_sample would consume, in the
 * dense layout [n_chains][T][F] (chain-major), so a CPU checker can replay them. */
int mpv_nsdiff_philox_normals(mpv_ctx* ctx, uint64_t seed, uint64_t offset, int64_t n_chains, float* out_dev);

/* ------------------------------------------------------------------------------------
 * sigma head g(x)
 * replaces: wv_sigma_trailing (NsDiff/sigma.py:34-72) + SigmaEstimation.forward
 *           (NsDiff/g_backbone.py:49-72), with scaler_transform (NsDiff_model.py:106-107)
 *           and build_sliding_windows (diffusion_model_uncertainy.py:137-147) folded
 *           into the load.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t seq_len;    /* windows (T of the head's input)                     */
    int32_t rolling;    /* kernel_size k of the trailing variance              */
    int32_t pred_len;   /* O                                                   */
    int32_t nf;         /* F                                                   */
    int32_t hidden;     /* 512                                                 */
} mpv_sigma_desc;

typedef struct {            /* host pointers: state_dict()["cond_pred_model_g.mlp.*"] */
    const float* w0;  const float* b0;       /* mlp.0  [hidden][seq_len-rolling], [hidden] */
    const float* ln0_w; const float* ln0_b;  /* mlp.2  [F][hidden]                         */
    const float* w1;  const float* b1;       /* mlp.3  [hidden][hidden]                    */
    const float* ln1_w; const float* ln1_b;  /* mlp.5  [F][hidden]                         */
    const float* w2;  const float* b2;       /* mlp.6  [pred_len][hidden]                  */
} mpv_sigma_weights;

int mpv_sigma_load(mpv_ctx* ctx, const mpv_sigma_desc* desc, const mpv_sigma_weights* host_weights);

/* series: dev [V][L][F] (already sub-sampled).  Window w covers samples
 * [first + w*step, first + w*step + seq_len) of every node (A.2 of SURVEY.md).
 * mean/std: dev [F] for a fused (x-mean)/std, or both NULL for an already scaled input.
 * add_eps is added to the head's output (10e-8 in NsDiff_model_variants, 0 in NsDiff_model
 * and in the gx-only path).  gx: dev [n_windows][V][O][F]. */
int mpv_sigma_head(mpv_ctx* ctx, const float* series, int32_t V, int64_t L, int32_t n_windows, int64_t first,
                   int64_t step, const float* mean, const float* std, float add_eps, float* gx);

/* ------------------------------------------------------------------------------------
 * MPV reduction
 * replaces: summarize_pred_future_list (diffusion_model_uncertainy.py:286-303; with
 *           _feature_inverse_transform :267-283) and the variance part of
 *           summarize_slbp_sensitivity (:529-538).
 * samples: dev [W][R][F][S] (R = B*O).  std/mean: dev [F] inverse scaling or NULL.
 * mpv[w]       = mean over (r,f) of the population variance over s of x*std+mean
 * pred_mean[w] = mean over everything of x*std+mean
 * mpv_feat[w][f] (nullable) = mean over r of that variance, per feature
 * ---------------------------------------------------------------------------------- */
int mpv_reduce(mpv_ctx* ctx, const float* samples, int32_t W, int64_t R, int32_t F, int32_t S, const float* std,
               const float* mean, float* mpv, float* pred_mean, float* mpv_feat);

/* Average device time in milliseconds of the chain-sampler launches since the last call
 * (measured with HIP events on the context's stream); also returns their count.  Used by
 * bench.py for the roofline line.  Synchronises the stream. */
int mpv_chain_timing(mpv_ctx* ctx, double* avg_ms, int64_t* launches);
int mpv_chain_timing_enable(mpv_ctx* ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* MPV_HIP_H */
