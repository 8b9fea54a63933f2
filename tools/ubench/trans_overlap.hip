// Does the transcendental unit (v_exp/v_log) run beside plain VALU issued by the OTHER wave of the SIMD,
// and beside plain VALU of the same wave?  Cycle counts via s_memtime.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// role 0: idle; 1: 16 x v_exp per iter; 2: 16 x v_fma per iter; 3: 8 v_exp interleaved with 8... (in-wave mix: exp, fma, fma, fma)
template <int ROLE>
__device__ __forceinline__ float body(int iters, float seed) {
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = 0.5f + seed * 1e-4f + j * 1e-3f;
    float m = 0.99999f, c = 1e-6f;
    asm volatile("" : "+v"(m), "+v"(c));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (ROLE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
            if (ROLE == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
            if (ROLE == 3) {   // 1 exp : 3 fma
                if ((j & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
            }
            if (ROLE == 4) {   // 1 exp : 1 fma
                if ((j & 1) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += x[j];
    return s;
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* cyc, float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r;
    if (__builtin_amdgcn_readfirstlane(wave) < 4) r = body<RA>(iters, (float)threadIdx.x);
    else r = body<RB>(iters, (float)threadIdx.x);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r == 12345.678f) out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RA, int RB>
void run(const char* name, unsigned long long* dc, float* d) {
    const int iters = 4000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<RA, RB>), dim3(blocks), dim3(512), 0, 0, dc, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int i = 0; i < blocks * 8; ++i) ((i & 7) < 4 ? a : b).push_back((double)h[i] / (iters * 16.0));
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-34s waves0-3: %6.2f cyc/instr   waves4-7: %6.2f cyc/instr\n", name, a[a.size() / 2], b[b.size() / 2]);
}

int main() {
    unsigned long long* dc; float* d;
    (void)hipMalloc(&dc, 8 * 256 * 8); (void)hipMalloc(&d, 4096);
    run<1, 0>("v_exp alone", dc, d);
    run<2, 0>("v_fma alone", dc, d);
    run<1, 1>("v_exp || v_exp", dc, d);
    run<2, 2>("v_fma || v_fma", dc, d);
    run<1, 2>("v_exp || v_fma", dc, d);
    run<3, 0>("(1 exp : 3 fma) alone", dc, d);
    run<3, 3>("(1 exp : 3 fma) || same", dc, d);
    run<4, 0>("(1 exp : 1 fma) alone", dc, d);
    run<4, 4>("(1 exp : 1 fma) || same", dc, d);
    return 0;
}
