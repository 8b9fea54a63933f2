// The graph convolutions `gnn_conv` can put into a SpatialBlock besides ResGatedGraphConv (reference models/layer/gnn_conv.py:4-28), shared by the
// DiffSTG / NsDiff_spatial engine (stg.hip) and Model_spatial's graph stage (nst.hip).  Round 4; restated from torch_geometric's published
// definitions (the package is absent: parity unpinned).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <vector>

namespace {

// ---- the other graph convolutions of models/layer/gnn_conv.py:4-28 (round 4; restated from torch_geometric's published definitions, unpinned) ------
// The Linear maps commute with the aggregation, so each layer is ONE product ar [N][NG] = z W^T (gemm(), like the gated layer's four maps) and one
// gather over the incoming edges in edge order.  MODE 1 SAGEConv(aggr = "mean"): relu(lin_l(mean_j x_j) + lin_r(x_i)) -- ar = (W_l x | W_r x), bias =
// lin_l.bias; a node without incoming edges aggregates 0.  MODE 2 GraphConv / SAGEConv(aggr = "add"): the sum instead of the mean.  MODE 3 GCNConv:
// relu(sum_j (d_i d_j)^-1/2 W x_j + b) over the incoming edges and ONE self loop per node (self loops of the input are replaced by it), d = in-degree
// without self loops + 1 (gcn_norm with add_remaining_self_loops, unit edge weights).
template <int MODE>
__global__ void graph_aggr_kernel(const float* __restrict__ ar, int NG, const float* __restrict__ bias, const int* __restrict__ row_ptr,
                                const int* __restrict__ col_idx, const int* __restrict__ deg /* in-degree without self loops */, int V, int C,
                                long long n, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // node*C + ch
    if (i >= n) return;
    const int ch = (int)(i % C);
    const long long node = i / C;
    const int v = (int)(node % V);
    const long long base = node - v;
    const int e0 = row_ptr[v], e1 = row_ptr[v + 1];
    float a = 0.0f;
    if (MODE == 3) {
        const float di = 1.0f / sqrtf((float)(deg[v] + 1));
        for (int e = e0; e < e1; ++e) {
            const int j = col_idx[e];
            if (j == v) continue;                                                // (a self loop of the input is replaced by the one added here)
            const float dj = 1.0f / sqrtf((float)(deg[j] + 1));
            a += (dj * di) * ar[(base + j) * NG + ch];
        }
        a += (di * di) * ar[node * NG + ch];                                     // the self loop (appended behind the edges)
    } else {
        for (int e = e0; e < e1; ++e) a += ar[(base + col_idx[e]) * NG + ch];
        if (MODE == 1) a = a / (float)max(e1 - e0, 1);
        a += ar[node * NG + C + ch];                                             // root weight (zeros when the layer has none)
    }
    out[i] = fmaxf(a + bias[ch], 0.0f);
}

// GATConv(heads = H, concat = True, add_self_loops = True), eval mode (no attention dropout): ar = (x' = W x [C = H D] | s_src [H] | s_dst [H]) with
// s_src[j][h] = <x'_j[h], att_src[h]> (folded into the product as H extra weight rows att_src[h]^T W_h, likewise s_dst);
// alpha_ij = softmax_j leaky_relu(s_src[j] + s_dst[i]) over the incoming edges (self loops in the input dropped) and one self loop;
// out_i = relu(sum_j alpha_ij x'_j + bias).  Softmax as torch_geometric.utils.softmax: exp(e - max) / (sum + 1e-16).
__global__ void graph_gat_kernel(const float* __restrict__ ar, int NG, const float* __restrict__ bias, const int* __restrict__ row_ptr,
                               const int* __restrict__ col_idx, int V, int C, int H, float slope, long long n, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // node*C + ch
    if (i >= n) return;
    const int ch = (int)(i % C), h = ch / (C / H);
    const long long node = i / C;
    const int v = (int)(node % V);
    const long long base = node - v;
    const int e0 = row_ptr[v], e1 = row_ptr[v + 1];
    const float sd = ar[node * NG + C + H + h];
    auto score = [&](long long j) { const float t = ar[j * NG + C + h] + sd; return t > 0.0f ? t : slope * t; };
    const float self = score(node);
    float m = self;
    for (int e = e0; e < e1; ++e)
        if (col_idx[e] != v) m = fmaxf(m, score(base + col_idx[e]));
    float den = 0.0f, a = 0.0f;
    for (int e = e0; e < e1; ++e) {
        if (col_idx[e] == v) continue;
        const long long j = base + col_idx[e];
        const float w = expf(score(j) - m);
        den += w;
        a += w * ar[j * NG + ch];
    }
    const float ws = expf(self - m);
    den += ws;
    a += ws * ar[node * NG + ch];
    out[i] = fmaxf(a / (den + 1e-16f) + bias[ch], 0.0f);
}

// gnn: 0 ResGatedGraphConv, 1 SAGEConv mean, 2 sum (GraphConv, SAGEConv add), 3 GCNConv, 4 GATConv (= step kind - MPV_STG_RES_SAGE_MEAN + 1).
// Columns of the layer's ONE product ar = x wg^T + bg:
inline int graph_conv_ng(int gnn, int C, int H) { return gnn == 0 ? 4 * C : (gnn == 3 ? C : (gnn == 4 ? (C + 2 * H + 3) & ~3 : 2 * C)); }

// The layer's Linear maps as one weight matrix wg [ng][C], a bias per product column bg [ng] and the bias added behind the aggregation gb [C], from the
// 8 tensor slots of the C ABI (include/mpv_hip.h: mpv_stg_step.w[12..19]; NULL where a kind has none).
inline void graph_conv_pack(int gnn, int C, int H, const float* const* w, std::vector<float>& wg, std::vector<float>& bg, std::vector<float>& gb) {
    const int ng = graph_conv_ng(gnn, C, H), D = C / std::max(H, 1);
    wg.assign((size_t)ng * C, 0.0f); bg.assign((size_t)ng, 0.0f); gb.assign((size_t)C, 0.0f);
    if (gnn == 0) {
        for (int part = 0; part < 4; ++part) {
            std::memcpy(&wg[(size_t)part * C * C], w[part < 3 ? 2 * part : 6], sizeof(float) * C * C);      // key.w, query.w, value.w at 0, 2, 4; skip.w at 6
            if (part < 3) std::memcpy(&bg[(size_t)part * C], w[2 * part + 1], sizeof(float) * C);
        }
        std::memcpy(gb.data(), w[7], sizeof(float) * C);
        return;
    }
    std::memcpy(wg.data(), w[0], sizeof(float) * C * C);                                               // lin_l / lin_rel / lin .weight
    if (w[1]) std::memcpy(gb.data(), w[1], sizeof(float) * C);                                          // lin_l.bias / lin_rel.bias / bias: added behind the aggregation
    if ((gnn == 1 || gnn == 2) && w[2]) std::memcpy(&wg[(size_t)C * C], w[2], sizeof(float) * C * C);   // lin_r / lin_root .weight
    if (gnn == 4)
        for (int side = 0; side < 2; ++side)                                                            // rows C + side H + h: att[h]^T W_h (float64 sums)
            for (int h = 0; h < H; ++h)
                for (int k = 0; k < C; ++k) {
                    double acc = 0.0;
                    for (int d = 0; d < D; ++d) acc += (double)w[2 + side][(size_t)h * D + d] * (double)w[0][((size_t)h * D + d) * C + k];
                    wg[((size_t)C + (size_t)side * H + h) * C + k] = (float)acc;
                }
}

// Aggregation + ReLU of kinds 1..4 over `rows` = a whole number of V-node graphs: ar [rows][ng] -> out [rows][C].
inline void graph_aggr_launch(int gnn, hipStream_t st, const float* ar, int ng, const float* gb, const int* row_ptr, const int* col_idx, const int* deg,
                              int V, int C, int H, float slope, long long rows, float* out) {
    const long long n = rows * C;
    const dim3 grid((unsigned)((n + 255) / 256));
    switch (gnn) {
        case 1: hipLaunchKernelGGL(graph_aggr_kernel<1>, grid, dim3(256), 0, st, ar, ng, gb, row_ptr, col_idx, deg, V, C, n, out); break;
        case 2: hipLaunchKernelGGL(graph_aggr_kernel<2>, grid, dim3(256), 0, st, ar, ng, gb, row_ptr, col_idx, deg, V, C, n, out); break;
        case 3: hipLaunchKernelGGL(graph_aggr_kernel<3>, grid, dim3(256), 0, st, ar, ng, gb, row_ptr, col_idx, deg, V, C, n, out); break;
        default: hipLaunchKernelGGL(graph_gat_kernel, grid, dim3(256), 0, st, ar, ng, gb, row_ptr, col_idx, V, C, H, slope, n, out); break;
    }
}

}  // namespace
