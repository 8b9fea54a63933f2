// Stand-alone timing of the f(x) Linear-layer kernels (csrc/nst_gemm.h) on the shapes of one 2168-row pass of BASELINE configs[1]:
// where does the time of a 128x128 output tile go -- main loop, global loads, epilogue?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I <pkg>/csrc tools/ubench/gemm_h3.hip -o /tmp/gemm_h3 && /tmp/gemm_h3
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include "nst_gemm.h"
#include "stg_down_mfma.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int PROBE>
float run(const NstGemmHArgs& a, int reps) {
    const long long tiles = ((a.M + G_BM - 1) / G_BM) * ((a.N + G_BN - 1) / G_BN);
    const dim3 grid((unsigned)(8 * ((tiles + 7) / 8)));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(nst_gemm_h3_kernel<PROBE>, grid, dim3(256), 0, 0, a);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(nst_gemm_h3_kernel<PROBE>, grid, dim3(256), 0, 0, a);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int PROBE>
float run256(const NstGemmHArgs& a, int reps) {
    const long long tiles = ((a.M + 255) / 256) * ((a.N + G_BN - 1) / G_BN);
    const dim3 grid((unsigned)(8 * ((tiles + 7) / 8)));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((nst_gemm_h3_kernel<PROBE, 256>), grid, dim3(256), 0, 0, a);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((nst_gemm_h3_kernel<PROBE, 256>), grid, dim3(256), 0, 0, a);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int PROBE>
float runlite(const NstGemmHArgs& a, int reps) {
    const long long tiles = ((a.M + 127) / 128) * ((a.N + G_BN - 1) / G_BN);
    const dim3 grid((unsigned)(8 * ((tiles + 7) / 8)));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((nst_gemm_h3_kernel<PROBE, 128, false, true>), grid, dim3(256), 0, 0, a);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((nst_gemm_h3_kernel<PROBE, 128, false, true>), grid, dim3(256), 0, 0, a);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

float run64(const NstGemmHArgs& a, int reps) {
    const long long tiles = ((a.M + 63) / 64) * ((a.N + G_BN - 1) / G_BN);
    const dim3 grid((unsigned)(8 * ((tiles + 7) / 8)));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((nst_gemm_h3_kernel<0, 64>), grid, dim3(256), 0, 0, a);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((nst_gemm_h3_kernel<0, 64>), grid, dim3(256), 0, 0, a);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

// operands with realistic bit patterns (O(1) values, all significand bits in play): zero-filled buffers draw less power, the chip clocks higher and every
// variant looks 10-20 % faster than it is inside the sampler (MI355X_MICROARCH.md: +19 % TFLOP/s for zero inputs at the same cycle count)
__global__ void fill_f32(float* p, long long n, unsigned seed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * 1.7f;
}
__global__ void fill_f16(_Float16* p, long long n, unsigned seed, float scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (_Float16)(((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * scale);
}

int main() {
    const long long M = 216800;
    struct Shape { int N, K; int act; bool res; const char* name; } shapes[] = {
        {1536, 512, 0, false, "qkv"}, {512, 512, 0, true, "out+res"}, {256, 512, 2, false, "ffn1 gelu"}, {512, 256, 0, true, "ffn2+res"}};
    float *X, *Y, *R, *bias;
    _Float16 *Wh, *Wl;
    CK(hipMalloc(&X, M * 1536 * 4)); CK(hipMalloc(&Y, M * 1536 * 4)); CK(hipMalloc(&R, M * 512 * 4)); CK(hipMalloc(&bias, 1536 * 4));
    CK(hipMalloc(&Wh, 1536 * 512 * 2)); CK(hipMalloc(&Wl, 1536 * 512 * 2));
    if (getenv("GEMM_ZEROS")) {
        CK(hipMemset(X, 0, M * 1536 * 4)); CK(hipMemset(R, 0, M * 512 * 4)); CK(hipMemset(bias, 0, 1536 * 4)); CK(hipMemset(Wh, 0, 1536 * 512 * 2));
        CK(hipMemset(Wl, 0, 1536 * 512 * 2));
    } else {
        hipLaunchKernelGGL(fill_f32, dim3((unsigned)((M * 1536 + 255) / 256)), dim3(256), 0, 0, X, M * 1536, 1u);
        hipLaunchKernelGGL(fill_f32, dim3((unsigned)((M * 512 + 255) / 256)), dim3(256), 0, 0, R, M * 512, 2u);
        hipLaunchKernelGGL(fill_f32, dim3(6), dim3(256), 0, 0, bias, 1536, 3u);
        hipLaunchKernelGGL(fill_f16, dim3(1536 * 512 / 256), dim3(256), 0, 0, Wh, 1536LL * 512, 4u, 12000.0f);
        hipLaunchKernelGGL(fill_f16, dim3(1536 * 512 / 256), dim3(256), 0, 0, Wl, 1536LL * 512, 5u, 4.0f);
        CK(hipDeviceSynchronize());
    }
    // the "downsampling" products on their own kernel (csrc/stg_down_mfma.h) against the general one: time and the largest difference of the outputs
    if (getenv("GEMM_DOWN")) {
        struct S3 { long long M; int N, K; const char* name; } sh3[] = {
            {25600, 80, 1600, "down T200"}, {25600, 160, 3200, "down cat"}, {25600, 80, 800, "down T100"}, {25600, 80, 3200, "N 80 K 3200"},
            {76800, 80, 1600, "nss 12 win"}, {76800, 160, 3200, "nss 12 win"}, {25500, 72, 1596, "ragged"}, {130, 20, 48, "tiny"}};
        float* Y2;
        CK(hipMalloc(&Y2, 76800LL * 160 * 4));
        if (getenv("GEMM_DOWN_PROBE")) {                  // where does a stage of the K = 3200 product go?  GEMM_DOWN_PROBE=5: N = 80, =10: N = 160
            const bool wide = atoi(getenv("GEMM_DOWN_PROBE")) == 10;
            const int Np = wide ? 160 : 80;
            NstGemmHArgs a{X, 3200, Wh, Wl, 1.0f / 8192.0f, bias, nullptr, Np, Y2, Np, 25600, Np, 3200, 0, 3200};
#define PROBE_RUN_NT(NT_, P_) do { \
            const int lds = (2 * 4 * (16 * NT_ * 32 + DN_PAD) + DN_WAVES * 4 * (16 * DN_MT * 32 + DN_PAD)) * 2; \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stg_down_h3_kernel<NT_, P_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); \
            hipLaunchKernelGGL((stg_down_h3_kernel<NT_, P_>), dim3(200), dim3(64 * DN_WAVES), lds, 0, a); \
            (void)hipEventRecord(e0, 0); \
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stg_down_h3_kernel<NT_, P_>), dim3(200), dim3(64 * DN_WAVES), lds, 0, a); \
            (void)hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); \
            printf("N=%d probe %2d: %.1f us\n", Np, P_, ms * 100); } while (0)
#define PROBE_RUN(P_) do { if (wide) PROBE_RUN_NT(10, P_); else PROBE_RUN_NT(5, P_); } while (0)
            PROBE_RUN(0); PROBE_RUN(1); PROBE_RUN(2); PROBE_RUN(4); PROBE_RUN(8); PROBE_RUN(12); PROBE_RUN(14); PROBE_RUN(13); PROBE_RUN(11);
            return 0;
        }
        for (const S3& s : sh3) {
            const int ldw = (s.K + 31) & ~31;
            CK(hipMemset(Wh, 0, 1536 * 512 * 2)); CK(hipMemset(Wl, 0, 1536 * 512 * 2));                 // columns K..ldw of a weight row are zero
            for (int n = 0; n < s.N; ++n) {
                hipLaunchKernelGGL(fill_f16, dim3((s.K + 255) / 256), dim3(256), 0, 0, Wh + (size_t)n * ldw, (long long)s.K, 40u + n, 12000.0f);
                hipLaunchKernelGGL(fill_f16, dim3((s.K + 255) / 256), dim3(256), 0, 0, Wl + (size_t)n * ldw, (long long)s.K, 900u + n, 4.0f);
            }
            NstGemmHArgs a{X, s.K, Wh, Wl, 1.0f / 8192.0f, bias, nullptr, s.N, Y, s.N, s.M, s.N, s.K, 0, ldw};
            NstGemmHArgs b = a;
            b.Y = Y2;
            const float tl = runlite<0>(a, 10);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            stg_down_h3_launch(b, 0);
            (void)hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) stg_down_h3_launch(b, 0);
            (void)hipEventRecord(e1, 0);
            CK(hipEventSynchronize(e1));
            float td = 0;
            (void)hipEventElapsedTime(&td, e0, e1);
            td /= 10;
            std::vector<float> h1((size_t)s.M * s.N), h2((size_t)s.M * s.N);
            CK(hipMemcpy(h1.data(), Y, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), Y2, h2.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0, big = 0;
            for (size_t i = 0; i < h1.size(); ++i) { worst = std::max(worst, (double)fabsf(h1[i] - h2[i])); big = std::max(big, (double)fabsf(h1[i])); }
            const double mb = (double)s.M * (s.K + s.N) * 4 / 1e6;
            printf("%-12s M=%6lld N=%4d K=%4d: general %.1f us (%.2f TB/s) | down kernel %.1f us (%.2f TB/s) | max |diff| %.3g of max |y| %.3g\n", s.name, s.M, s.N, s.K,
                   tl * 1e3, mb / tl / 1e3, td * 1e3, mb / td / 1e3, worst, big);
        }
        return 0;
    }
    // DiffSTG at BASELINE configs[4] (4 windows x 6400 node rows per launch; C = Td_h c_out, k = T + 1 convolutions as [rows][c_out T] x [c_out T][C])
    if (getenv("GEMM_STG")) {
        const long long Ms = 25600;
        struct S2 { int N, K; bool res; const char* name; } sh2[] = {
            {80, 1600, false, "down T200"}, {160, 3200, false, "down cat"}, {80, 800, false, "down T100"}, {320, 80, false, "gate"},
            {1600, 80, true, "up T200"}, {3200, 80, true, "up x2"}};
        for (const S2& s : sh2) {
            const int ldw = (s.K + 31) & ~31;
            NstGemmHArgs a{X, s.K, Wh, Wl, 1.0f, bias, s.res ? R : nullptr, s.N, Y, s.N, Ms, s.N, s.K, 0, ldw};
            const double mb = (double)Ms * (s.K + s.N * (s.res ? 2 : 1)) * 4 / 1e6;
            const float full = run<0>(a, 5);
            const float t64 = run64(a, 5);
            const float tl = runlite<0>(a, 5);
            printf("%-10s M=%lld N=%4d K=%4d: 128-row tiles %.1f us, 64-row tiles %.1f us, 128-row 3 WG/CU %.1f us  (%.0f MB algorithmic -> %.2f / %.2f / %.2f TB/s)\n", s.name, Ms, s.N, s.K, full * 1e3,
                   t64 * 1e3, tl * 1e3, mb, mb / full / 1e3, mb / t64 / 1e3, mb / tl / 1e3);
        }
        return 0;
    }
    for (const Shape& s : shapes) {
        NstGemmHArgs a{X, s.K, Wh, Wl, 1.0f / 8192.0f, bias, s.res ? R : nullptr, 512, Y, s.N, M, s.N, s.K, s.act, s.K};
        const double gf = 2.0 * M * s.N * s.K / 1e9;
        const float full = run<0>(a, 5), nostore = run<1>(a, 5), onek = run<2>(a, 5), noload = run<4>(a, 5), loop_only = run<5>(a, 5);
        printf("%-10s N=%4d K=%3d: full %.3f ms (%.0f TF/s) | no stores %.3f | one k-tile %.3f | no loads in loop %.3f | no loads, no stores %.3f\n", s.name, s.N, s.K,
               full, gf / full, nostore, onek, noload, loop_only);
        const float flite = runlite<0>(a, 5), llite = runlite<5>(a, 5), olite = runlite<2>(a, 5);
        printf("           3 workgroups / CU: full %.3f ms (%.0f TF/s) | one k-tile %.3f | no loads, no stores %.3f\n", flite, gf / flite, olite, llite);
        const float f256 = run256<0>(a, 5), l256 = run256<5>(a, 5), o256 = run256<2>(a, 5);
        printf("           256-row tiles: full %.3f ms (%.0f TF/s) | one k-tile %.3f | no loads, no stores %.3f\n", f256, gf / f256, o256, l256);
    }
    return 0;
}
