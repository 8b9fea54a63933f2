// Pinned-order (inline asm) issue-cost measurements on gfx950: MFMA gap + N fillers, in s_memtime cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int OP, int N, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(unsigned long long* cyc, float* out, int iters) {
    f32x16 acc = {0};
    float a = threadIdx.x, b = a * 0.5f;
    f16x8 ah, bh;
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(a + j); bh[j] = (_Float16)(b); }
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = 1.0f + a * 1e-4f + j * 1e-3f;
    float m = 1.0000001f, c = 1e-9f;
    asm volatile("" : "+v"(m), "+v"(c));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (KIND == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah), "v"(bh));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int i2 = (kk * N + j) % 16;
                if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i2]) : "v"(m), "v"(c));
                if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i2]));
                if (OP == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i2]) : "v"(c));
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc[0] + acc[5];
    for (int j = 0; j < 16; ++j) s += x[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int OP, int N, int WAVES>
void run(const char* name, unsigned long long* dc, float* d) {
    const int iters = 4000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, OP, N, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, dc, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * WAVES);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / (iters * 4.0);
    printf("%-5s MFMA + %2d x %-7s %d wave/SIMD: %7.1f cyc per gap per wave = %6.1f per SIMD-gap\n",
           KIND == 1 ? "fp32" : (KIND == 2 ? "f16" : "no"), N, name, WAVES / 4, med, med / (WAVES / 4));
}

int main() {
    unsigned long long* dc; float* d;
    (void)hipMalloc(&dc, 8 * 256 * 8); (void)hipMalloc(&d, 4096);
#define R(K, OP, N, W, NAME) run<K, OP, N, W>(NAME, dc, d);
    R(0, 1, 8, 4, "v_fma") R(0, 1, 8, 8, "v_fma") R(0, 3, 4, 4, "v_exp") R(0, 3, 4, 8, "v_exp") R(0, 8, 8, 8, "v_max")
    R(1, 0, 0, 4, "-") R(1, 1, 4, 4, "v_fma") R(1, 1, 8, 4, "v_fma") R(1, 1, 8, 8, "v_fma") R(1, 3, 2, 4, "v_exp")
    R(2, 0, 0, 4, "-") R(2, 0, 0, 8, "-")
    R(2, 1, 2, 4, "v_fma") R(2, 1, 4, 4, "v_fma") R(2, 1, 6, 4, "v_fma") R(2, 1, 8, 4, "v_fma") R(2, 1, 12, 4, "v_fma")
    R(2, 1, 2, 8, "v_fma") R(2, 1, 4, 8, "v_fma") R(2, 1, 6, 8, "v_fma") R(2, 1, 8, 8, "v_fma") R(2, 1, 12, 8, "v_fma")
    R(2, 3, 1, 4, "v_exp") R(2, 3, 2, 4, "v_exp") R(2, 3, 3, 4, "v_exp") R(2, 3, 1, 8, "v_exp") R(2, 3, 2, 8, "v_exp")
    return 0;
}
