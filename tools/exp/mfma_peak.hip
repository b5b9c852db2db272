// Experiment (not part of the product): what does v_mfma_f32_32x32x2_f32 sustain on this part when nothing else is in
// the way?  Register-only MFMA loop on every SIMD; reports TFLOP/s from HIP events and the shader clock seen by the
// kernel (s_memtime ticks / s_memrealtime ticks at 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_peak.hip -o tools/exp/_bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("CUs %d  clockRate %d kHz\n", cus, p.clockRate);
    for (int wpb = 1; wpb <= 3; ++wpb) {               // resident workgroups (of 4 waves) per CU
        const int blocks = cus * wpb;
        float* out; unsigned long long* clk;
        hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double flops = (double)blocks * 4 * iters * 8 * 4 * (2.0 * 32 * 32 * 2);
            printf("waves/SIMD %d  %.2f ms  %.1f TFLOP/s  | kernel-side: %llu shader ticks / %llu ref ticks -> %.0f MHz (if ref = 100 MHz)\n",
                   wpb, ms, flops / ms / 1e9, h[0], h[1], (double)h[0] / (double)h[1] * 100.0);
        }
        hipFree(out); hipFree(clk);
    }
    return 0;
}
