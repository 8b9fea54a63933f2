// In-wave overlap: dependent MFMA chain with N independent v_fma / v_exp fillers per MFMA gap,
// one wave per SIMD (256-thread workgroups), and cross-wave with the VALU wave as the OLDER wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND /*1 fp32, 2 f16*/, int NF /*v_fma fillers per MFMA*/, int NE /*v_exp fillers per MFMA*/>
__global__ __launch_bounds__(256, 1) void inwave(float* out, int iters) {
    f32x16 acc = {0};
    float a = threadIdx.x, b = a * 0.5f;
    f16x8 ah, bh;
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(a + j); bh[j] = (_Float16)(b); }
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = a * 1e-3f + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (KIND == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NF; ++j) x[j] = __builtin_fmaf(x[j], 1.0000001f, 1e-9f);
#pragma unroll
            for (int j = 0; j < NE; ++j) x[15 - j] = __builtin_amdgcn_exp2f(x[15 - j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = acc[0] + acc[5];
    for (int j = 0; j < 16; ++j) s += x[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND, int NF, int NE>
float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((inwave<KIND, NF, NE>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((inwave<KIND, NF, NE>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

#define ROW(K, NF, NE, IT) { float ms = run<K, NF, NE>(d, IT); \
    printf("%s MFMA + %2d v_fma + %d v_exp per gap: %7.3f ms  = %6.1f cycles/MFMA @2.4GHz\n", K == 1 ? "fp32" : (K == 2 ? "f16 " : "none"), NF, NE, ms, ms * 1e-3 * 2.4e9 / (IT * 8.0)); }

int main() {
    float* d; (void)hipMalloc(&d, 4096);
    const int IT = 20000;
    ROW(1, 0, 0, IT) ROW(1, 4, 0, IT) ROW(1, 8, 0, IT) ROW(1, 12, 0, IT) ROW(1, 16, 0, IT) ROW(1, 0, 2, IT) ROW(1, 8, 2, IT)
    ROW(2, 0, 0, IT) ROW(2, 2, 0, IT) ROW(2, 4, 0, IT) ROW(2, 6, 0, IT) ROW(2, 8, 0, IT) ROW(2, 12, 0, IT) ROW(2, 0, 1, IT) ROW(2, 0, 2, IT) ROW(2, 4, 1, IT)
    ROW(0, 8, 0, IT) ROW(0, 16, 0, IT) ROW(0, 0, 2, IT)
    return 0;
}
