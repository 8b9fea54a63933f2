// Issue rate of the matrix instructions the DiffusionTS kernels could use (gfx950, one wave per SIMD, cycles via s_memtime):
// v_mfma_f32_16x16x4_f32 (today), v_mfma_f32_16x16x16_f16 (K = head width in one instruction), v_mfma_f32_16x16x32_f16, 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, float* out, int iters) {
    f32x4 acc[4] = {{0}, {0}, {0}, {0}};
    f16x4 a4, b4; f16x8 a8, b8; float af = threadIdx.x * 1e-3f, bf = 0.5f;
    for (int j = 0; j < 4; ++j) { a4[j] = (_Float16)(af + j); b4[j] = (_Float16)0.25f; }
    for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(af + j); b8[j] = (_Float16)0.25f; }
    asm volatile("" : "+v"(af), "+v"(bf));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[q & 3]) : "v"(af), "v"(bf));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[q & 3]) : "v"(a4), "v"(b4));
            if (KIND == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q & 3]) : "v"(a8), "v"(b8));
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, double flop, unsigned long long* dc, float* d) {
    const int iters = 2000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, dc, d, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / (iters * 16.0);
    printf("%-28s %6.1f cycles per instruction per SIMD, %7.1f flop/clk/SIMD\n", name, med, flop / med);
}

int main() {
    unsigned long long* dc; float* d;
    (void)hipMalloc(&dc, 8 * 256 * 4); (void)hipMalloc(&d, 4096);
    run<0>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, dc, d);
    run<1>("v_mfma_f32_16x16x16_f16", 2.0 * 16 * 16 * 16, dc, d);
    run<2>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, dc, d);
    return 0;
}
