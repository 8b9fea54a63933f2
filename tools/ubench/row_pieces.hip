// What does HBM deliver for the access pattern of a k-outer GEMM over row-major X[M][K] fp32?  Every workgroup owns 128 consecutive rows (4 waves x 32)
// and walks along K in stages; a stage reads P contiguous bytes of each of its rows.  P = 128 is nst_gemm_h3_kernel's k-tile, P = 256 the stage of
// stg_down_h3_kernel.  "linear": the workgroup reads its 128 x K block as one contiguous stream instead (what a k-inner loop nest would do).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/ubench/row_pieces.hip -o build_variants/row_pieces && build_variants/row_pieces
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int P, int DEPTH>
__global__ __launch_bounds__(256) void rd_pieces(const float* X, int K, long long M, float* out) {
    constexpr int L = P / 32;                              // 16-byte loads per lane and stage (a wave: 32 rows x P bytes)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long m0 = (long long)blockIdx.x * 128 + wave * 32;
    const int stages = K * 4 / P;
    const char* base[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int q = i * 64 + lane, row = q / (P / 16), off = (q % (P / 16)) * 16;
        base[i] = reinterpret_cast<const char*>(X + (size_t)(m0 + row) * K) + off;
    }
    v4f buf[DEPTH][L];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < L; ++i) buf[d][i] = *reinterpret_cast<const v4f*>(base[i] + (size_t)(d < stages ? d : stages - 1) * P);
    v4f sum{0, 0, 0, 0};
    for (int s = 0; s < stages; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < L; ++i) sum += buf[d][i];
            const int nx = s + d + DEPTH < stages ? s + d + DEPTH : stages - 1;
#pragma unroll
            for (int i = 0; i < L; ++i) buf[d][i] = *reinterpret_cast<const v4f*>(base[i] + (size_t)nx * P);
        }
    }
    if (sum[0] + sum[1] + sum[2] + sum[3] == 123.456f) out[0] = 1.0f;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void rd_linear(const float* X, int K, long long M, float* out) {
    constexpr int L = 8;                                   // 32 KB per workgroup and stage
    const char* base = reinterpret_cast<const char*>(X + (size_t)blockIdx.x * 128 * K) + threadIdx.x * 16;
    const int stages = 128 * K * 4 / 32768;
    v4f buf[DEPTH][L];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < L; ++i) buf[d][i] = *reinterpret_cast<const v4f*>(base + (size_t)(d < stages ? d : stages - 1) * 32768 + i * 4096);
    v4f sum{0, 0, 0, 0};
    for (int s = 0; s < stages; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < L; ++i) sum += buf[d][i];
            const int nx = s + d + DEPTH < stages ? s + d + DEPTH : stages - 1;
#pragma unroll
            for (int i = 0; i < L; ++i) buf[d][i] = *reinterpret_cast<const v4f*>(base + (size_t)nx * 32768 + i * 4096);
        }
    }
    if (sum[0] + sum[1] + sum[2] + sum[3] == 123.456f) out[0] = 1.0f;
}

template <typename F>
float timed(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const long long Mmax = 76800;
    float *X, *out;
    CK(hipMalloc(&X, Mmax * 3200 * 4)); CK(hipMalloc(&out, 16));
    CK(hipMemset(X, 1, Mmax * 3200 * 4));
    struct { long long M; int K; } cases[] = {{25600, 3200}, {25600, 1600}, {76800, 3200}, {76800, 1600}};
    for (auto c : cases) {
        const dim3 grid((unsigned)(c.M / 128));
        const double mb = (double)c.M * c.K * 4 / 1e6;
        const long long M = c.M; const int K = c.K;
#define RUN(NAME, KERNEL) do { const float ms = timed([&] { hipLaunchKernelGGL((KERNEL), grid, dim3(256), 0, 0, X, K, M, out); }); \
        printf("M=%6lld K=%4d  %-28s %7.1f us  %5.2f TB/s\n", M, K, NAME, ms * 1e3, mb / ms / 1e3); } while (0)
        RUN("128-B pieces, 2 in flight", (rd_pieces<128, 2>));
        RUN("128-B pieces, 4 in flight", (rd_pieces<128, 4>));
        RUN("256-B pieces, 2 in flight", (rd_pieces<256, 2>));
        RUN("256-B pieces, 4 in flight", (rd_pieces<256, 4>));
        RUN("512-B pieces, 2 in flight", (rd_pieces<512, 2>));
        RUN("1024-B pieces, 1 in flight", (rd_pieces<1024, 1>));
        RUN("1024-B pieces, 2 in flight", (rd_pieces<1024, 2>));
        RUN("linear 32 KB, 2 in flight", (rd_linear<2>));
        RUN("linear 32 KB, 4 in flight", (rd_linear<4>));
    }
    CK(hipDeviceSynchronize());
    return 0;
}
