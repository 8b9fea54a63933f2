// Cycle-accurate (s_memtime) issue-cost measurements on gfx950, one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// OP: 0 none 1 v_fma 2 v_pk_fma 3 v_exp 4 v_log 5 v_cndmask 6 v_pk_mul 7 v_mul
template <int OP, int N>
__device__ __forceinline__ void fillers(float (&x)[32]) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int i = j % 16;
        if (OP == 1) x[i] = __builtin_fmaf(x[i], 1.0000001f, 1e-9f);
        if (OP == 2) { f32x2 v = {x[2 * i], x[2 * i + 1]}; f32x2 m = {1.0000001f, 0.9999999f}; f32x2 c = {1e-9f, 2e-9f};
                       asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(m), "v"(c)); x[2 * i] = v[0]; x[2 * i + 1] = v[1]; }
        if (OP == 3) x[i] = __builtin_amdgcn_exp2f(x[i]);
        if (OP == 4) x[i] = __builtin_amdgcn_logf(x[i]);
        if (OP == 5) x[i] = x[i] > 0.5f ? x[(i + 1) % 16] : x[i];
        if (OP == 6) { f32x2 v = {x[2 * i], x[2 * i + 1]}; f32x2 m = {1.0000001f, 0.9999999f};
                       asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(m)); x[2 * i] = v[0]; x[2 * i + 1] = v[1]; }
        if (OP == 7) x[i] = x[i] * 1.0000001f;
    }
}

template <int KIND /*0 none 1 fp32 2 f16*/, int OP, int N, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(unsigned long long* cyc, float* out, int iters) {
    f32x16 acc = {0};
    float a = threadIdx.x, b = a * 0.5f;
    f16x8 ah, bh;
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(a + j); bh[j] = (_Float16)(b); }
    float x[32];
    for (int j = 0; j < 32; ++j) x[j] = 1.0f + a * 1e-4f + j * 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (KIND == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            fillers<OP, N>(x);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc[0] + acc[5];
    for (int j = 0; j < 32; ++j) s += x[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int OP, int N, int WAVES>
void run(const char* name, unsigned long long* dc, float* d) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((k<KIND, OP, N, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, dc, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, OP, N, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, dc, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * WAVES);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / (iters * 4.0);
    printf("%-6s + %2d x %-9s  %d wave/SIMD: %7.1f cyc per gap per wave (%6.1f per SIMD-gap)  wall %.3f ms  eff clock %.2f GHz\n",
           KIND == 1 ? "fp32" : (KIND == 2 ? "f16" : "-"), N, name, WAVES / 4, med, med / (WAVES / 4), ms,
           (double)h[h.size() / 2] / (ms * 1e6));
}

int main() {
    unsigned long long* dc; float* d;
    (void)hipMalloc(&dc, 8 * 256 * 8); (void)hipMalloc(&d, 4096);
#define R(K, OP, N, W, NAME) run<K, OP, N, W>(NAME, dc, d);
    R(0, 1, 16, 4, "v_fma") R(0, 1, 16, 8, "v_fma") R(0, 7, 16, 4, "v_mul") R(0, 2, 8, 4, "v_pk_fma") R(0, 2, 8, 8, "v_pk_fma") R(0, 6, 8, 8, "v_pk_mul")
    R(0, 3, 16, 4, "v_exp") R(0, 3, 16, 8, "v_exp") R(0, 4, 16, 8, "v_log") R(0, 5, 16, 4, "v_cndmask") R(0, 5, 16, 8, "v_cndmask")
    R(1, 0, 0, 4, "-") R(1, 0, 0, 8, "-") R(1, 1, 8, 4, "v_fma") R(1, 1, 8, 8, "v_fma") R(1, 3, 4, 4, "v_exp")
    R(2, 0, 0, 4, "-") R(2, 0, 0, 8, "-") R(2, 1, 4, 4, "v_fma") R(2, 1, 8, 4, "v_fma") R(2, 1, 12, 4, "v_fma") R(2, 1, 4, 8, "v_fma") R(2, 1, 8, 8, "v_fma") R(2, 1, 16, 8, "v_fma")
    R(2, 3, 2, 4, "v_exp") R(2, 3, 4, 4, "v_exp") R(2, 3, 2, 8, "v_exp") R(2, 3, 4, 8, "v_exp") R(2, 2, 4, 4, "v_pk_fma") R(2, 2, 8, 8, "v_pk_fma")
    return 0;
}
