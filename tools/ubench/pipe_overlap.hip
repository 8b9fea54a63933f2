// Do MFMA and VALU issued by two different waves of one SIMD overlap on gfx950?
// Workgroup = 512 threads = 2 waves per SIMD.  Waves 0-3 run role A, waves 4-7 role B.
// Roles: 0 = idle, 1 = fp32 MFMA (32x32x2), 2 = f16 MFMA (32x32x16), 3 = VALU v_fma_f32, 4 = v_exp_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int ROLE>
__device__ __forceinline__ float work(int iters, float seed) {
    if (ROLE == 1) {
        f32x16 acc = {0};
        float a = seed, b = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        return acc[0] + acc[7];
    } else if (ROLE == 2) {
        f32x16 acc = {0};
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + j); b[j] = (_Float16)(seed * 0.5f); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        return acc[0] + acc[7];
    } else if (ROLE == 3) {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = seed + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 32; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], 1.0000001f, 1e-9f);
        }
        float s = 0; for (int j = 0; j < 8; ++j) s += x[j];
        return s;
    } else if (ROLE == 4) {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = seed * 1e-3f + j * 1e-4f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_exp2f(x[j]) * 1e-3f;
        }
        float s = 0; for (int j = 0; j < 8; ++j) s += x[j];
        return s;
    }
    return 0.f;
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void k(float* out, int itA, int itB) {
    const int wave = threadIdx.x >> 6;
    float r;
    if (__builtin_amdgcn_readfirstlane(wave) < 4) r = work<RA>(itA, (float)threadIdx.x);
    else r = work<RB>(itB, (float)threadIdx.x);
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int RA, int RB>
float run(float* d, int itA, int itB) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, itA, itB);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, itA, itB);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* d; hipMalloc(&d, 4096);
    // iteration counts chosen so each role alone takes roughly the same time
    const int M32 = 4000;     // 16 fp32 MFMA x 64 cyc = 1024 cyc/iter
    const int M16 = 8000;     // 16 f16 MFMA x 32 cyc = 512 cyc/iter
    const int VF = 4000;      // 256 v_fma/iter
    const int VE = 8000;      // 64 v_exp + 64 mul /iter
    printf("fp32 MFMA alone (waves 0-3)         %8.3f ms\n", run<1, 0>(d, M32, 0));
    printf("f16 MFMA alone                      %8.3f ms\n", run<2, 0>(d, M16, 0));
    printf("v_fma alone                         %8.3f ms\n", run<3, 0>(d, VF, 0));
    printf("v_exp alone                         %8.3f ms\n", run<4, 0>(d, VE, 0));
    printf("fp32 MFMA || fp32 MFMA              %8.3f ms\n", run<1, 1>(d, M32, M32));
    printf("f16 MFMA  || f16 MFMA               %8.3f ms\n", run<2, 2>(d, M16, M16));
    printf("v_fma     || v_fma                  %8.3f ms\n", run<3, 3>(d, VF, VF));
    printf("v_exp     || v_exp                  %8.3f ms\n", run<4, 4>(d, VE, VE));
    printf("fp32 MFMA || v_fma                  %8.3f ms\n", run<1, 3>(d, M32, VF));
    printf("fp32 MFMA || v_exp                  %8.3f ms\n", run<1, 4>(d, M32, VE));
    printf("f16 MFMA  || v_fma                  %8.3f ms\n", run<2, 3>(d, M16, VF));
    printf("f16 MFMA  || v_exp                  %8.3f ms\n", run<2, 4>(d, M16, VE));
    return 0;
}
