// Clustered vs interleaved MFMA/VALU issue, two waves per SIMD (gfx950, cycles via s_memtime).
// Each iteration issues G groups; a group is C dependent f16 MFMAs back to back followed by C*N independent v_fma.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int C, int N, int WAVES, bool INDEP>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(unsigned long long* cyc, float* out, int iters) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    f16x8 ah, bh;
    float a = threadIdx.x;
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(a + j); bh[j] = (_Float16)(a * 0.5f); }
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = 1.0f + a * 1e-4f + j * 1e-3f;
    float m = 1.0000001f, c = 1e-9f;
    asm volatile("" : "+v"(m), "+v"(c));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 12 / C; ++g) {
#pragma unroll
            for (int q = 0; q < C; ++q) {
                if (!INDEP || (q & 3) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(ah), "v"(bh));
                else if ((q & 3) == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(ah), "v"(bh));
                else if ((q & 3) == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(ah), "v"(bh));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc3) : "v"(ah), "v"(bh));
            }
#pragma unroll
            for (int j = 0; j < C * N; ++j)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j % 16]) : "v"(m), "v"(c));
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc0[0] + acc1[5] + acc2[1] + acc3[2];
    for (int j = 0; j < 16; ++j) s += x[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int C, int N, int WAVES, bool INDEP>
void run(unsigned long long* dc, float* d) {
    const int iters = 2000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<C, N, WAVES, INDEP>), dim3(blocks), dim3(WAVES * 64), 0, 0, dc, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * WAVES);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / (iters * 12.0);
    printf("cluster of %2d %s MFMA then %3d v_fma, %d wave/SIMD: %6.1f cyc per MFMA per wave = %6.1f per SIMD (ideal max(32, %d*2.3=%.0f))\n",
           C, INDEP ? "indep" : "dep  ", C * N, WAVES / 4, med, med / (WAVES / 4), N, N * 2.3);
}

int main() {
    unsigned long long* dc; float* d;
    (void)hipMalloc(&dc, 8 * 256 * 8); (void)hipMalloc(&d, 4096);
    run<1, 6, 8, false>(dc, d); run<3, 6, 8, false>(dc, d); run<12, 6, 8, false>(dc, d);
    run<1, 12, 8, false>(dc, d); run<3, 12, 8, false>(dc, d); run<12, 12, 8, false>(dc, d);
    run<1, 20, 8, false>(dc, d); run<3, 20, 8, false>(dc, d); run<12, 20, 8, false>(dc, d);
    run<12, 12, 8, true>(dc, d); run<12, 20, 8, true>(dc, d);
    run<1, 12, 4, false>(dc, d); run<12, 12, 4, false>(dc, d); run<12, 20, 4, false>(dc, d);
    return 0;
}
