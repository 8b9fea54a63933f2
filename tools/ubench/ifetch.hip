// Is a long straight-line VALU stream instruction-fetch bound on gfx950?  N unrolled v_fma per loop iteration,
// 8 waves per CU (2 per SIMD), every CU busy.  Cycles per instruction per SIMD from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int N, bool E64>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* cyc, float* out, int iters) {
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = 1.0f + threadIdx.x * 1e-4f + j * 1e-3f;
    float m = 1.0000001f, c = 1e-9f;
    asm volatile("" : "+v"(m), "+v"(c));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (E64) x[j % 16] = __builtin_fmaf(x[j % 16], m, c);          // v_fma_f32: 8-byte encoding
            else x[j % 16] = x[j % 16] * m;                                 // v_mul_f32_e32: 4-byte encoding
        }
        asm volatile("" : "+v"(x[0]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += x[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int N, bool E64>
void run(unsigned long long* dc, float* d) {
    const int total = 1 << 20;
    const int iters = total / N, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<N, E64>), dim3(blocks), dim3(512), 0, 0, dc, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%5d x %s per iteration (%6.1f KB of code): %5.2f cycles per instruction per wave, %5.2f per SIMD\n", N,
           E64 ? "v_fma_f32 (8 B)" : "v_mul_f32 (4 B)", N * (E64 ? 8 : 4) / 1024.0, (double)h[h.size() / 2] / ((double)iters * N),
           (double)h[h.size() / 2] / ((double)iters * N) / 2);
}

int main() {
    unsigned long long* dc; float* d;
    (void)hipMalloc(&dc, 8 * 256 * 8); (void)hipMalloc(&d, 4096);
    run<1024, true>(dc, d); run<1152, true>(dc, d); run<1280, true>(dc, d); run<1408, true>(dc, d); run<1536, true>(dc, d);
    run<1664, true>(dc, d); run<1792, true>(dc, d); run<1920, true>(dc, d); run<2048, true>(dc, d); run<4096, true>(dc, d);
    run<2048, false>(dc, d); run<3072, false>(dc, d); run<4096, false>(dc, d);
    return 0;
}
