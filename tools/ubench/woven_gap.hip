// What breaks MFMA || VALU co-issue in a woven stream?  One dependent f16 MFMA per gap plus a filler mix, optional
// LDS reads of the next A operands, VALU working on a second 16-register tile (gfx950, s_memtime cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MIX: 0 = 6 v_mul on tile regs, 1 = 4 v_mul + v_exp + v_log, 2 = 6 v_fma on 3 private regs (as pipe_overlap4)
// LDS: 0 none, 1 = one ds_read_b128 per gap into the A ring (consumed 3 gaps later), 2 = two per gap
// ACCK: 0 = one accumulator (dependent chain), 1 = MFMA accumulates into tile t0 while VALU works on tile t1, swapping
template <int MIX, int LDS, int WAVES, int NV>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(unsigned long long* cyc, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = 1.0f + i * 1e-6f;
    f32x16 acc = {0}, tile;
    float a = threadIdx.x, b = a * 0.5f;
    f16x8 ring[3], bh;
    for (int j = 0; j < 8; ++j) { ring[0][j] = ring[1][j] = ring[2][j] = (_Float16)(a + j); bh[j] = (_Float16)(b); }
    for (int j = 0; j < 16; ++j) tile[j] = 1.0f + a * 1e-4f + j * 1e-3f;
    float m = 1.0000001f, c = 1e-9f, x0 = 1.0f, x1 = 1.1f, x2 = 1.2f;
    asm volatile("" : "+v"(m), "+v"(c));
    const unsigned addr = (threadIdx.x & 63) * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (LDS >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[(g + 2) % 3]) : "v"(addr), "n"(g * 1024));
            if (LDS >= 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ring[g % 3]), "v"(bh));
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int e = (g * NV + j) % 16;
                if (MIX == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(tile[e]) : "v"(m));
                if (MIX == 1) {
                    if (j == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(tile[e]));
                    else if (j == 4) asm volatile("v_log_f32 %0, %0" : "+v"(tile[e]));
                    else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(tile[e]) : "v"(m));
                }
                if (MIX == 2) {
                    if (j % 3 == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(m), "v"(c));
                    if (j % 3 == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(m), "v"(c));
                    if (j % 3 == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(m), "v"(c));
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc[0] + acc[5] + x0 + x1 + x2 + (float)ring[0][0] + (float)ring[1][1] + (float)ring[2][2];
    for (int j = 0; j < 16; ++j) s += tile[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MIX, int LDS, int WAVES, int NV>
void run(unsigned long long* dc, float* d) {
    const int iters = 3000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MIX, LDS, WAVES, NV>), dim3(blocks), dim3(WAVES * 64), 0, 0, dc, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * WAVES);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / (iters * 6.0);
    static const char* mix[] = {"v_mul on a 16-reg tile", "4 v_mul + exp + log (tile)", "v_fma on 3 private regs"};
    printf("MFMA + %d x %-28s lds reads/gap %d  %d wave/SIMD: %7.1f cyc per gap per wave = %6.1f per SIMD-gap\n", NV, mix[MIX], LDS,
           WAVES / 4, med, med / (WAVES / 4));
}

int main() {
    unsigned long long* dc; float* d;
    (void)hipMalloc(&dc, 8 * 256 * 8); (void)hipMalloc(&d, 4096);
    run<2, 0, 8, 6>(dc, d); run<0, 0, 8, 6>(dc, d); run<1, 0, 8, 6>(dc, d);
    run<2, 1, 8, 6>(dc, d); run<0, 1, 8, 6>(dc, d); run<1, 1, 8, 6>(dc, d);
    run<2, 0, 4, 6>(dc, d); run<0, 0, 4, 6>(dc, d); run<1, 0, 4, 6>(dc, d); run<1, 1, 4, 6>(dc, d);
    run<0, 0, 8, 4>(dc, d); run<0, 0, 8, 8>(dc, d); run<1, 0, 8, 8>(dc, d);
    return 0;
}
