// Experiment (not part of the product): how fast can 64 channel planes of an NCHW fp32 activation be WRITTEN on gfx950
// as a function of the store pattern?  Sizes are the stem's: n x 64 x 800 x 1333.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/store_pattern.hip -o gpurun_out/store_pattern && gpurun_out/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// mode 0: block (4 waves) owns 256 px of a row; wave w writes planes 16w..16w+15, 16 B per lane   (the stem's pattern)
// mode 1: same, 4 B per lane stores (4 per plane)
// mode 2: wave owns 256 px x ROWS rows of 16 planes (more contiguous work per wave, same pattern per store)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ y, int H, int W, float val)
{
    const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int px0 = (blockIdx.x * 64 + lane) * 4, py = blockIdx.y, n = blockIdx.z;
    const size_t HW = (size_t)H * W;
    if (px0 >= W) return;
    float* yn = y + (size_t)n * 64 * HW + (size_t)py * W + px0 + (size_t)cg * 16 * HW;
    const bool full = px0 + 4 <= W;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float* dst = yn + (size_t)c * HW;
        if (MODE == 0 && full) { f4u q = {val, val, val, val}; *reinterpret_cast<f4u*>(dst) = q; }
        else { for (int p = 0; p < 4; ++p) if (px0 + p < W) dst[p] = val; }
    }
}

// mode 3: one pixel per lane (old stem): block = 64 px x 4 rows, each thread writes 64 planes x 4 B
__global__ __launch_bounds__(256) void k_old(float* __restrict__ y, int H, int W, float val)
{
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
    const size_t HW = (size_t)H * W;
    if (px >= W || py >= H) return;
    float* yn = y + (size_t)n * 64 * HW + (size_t)py * W + px;
#pragma unroll
    for (int c = 0; c < 64; ++c) yn[(size_t)c * HW] = val;
}

// mode 4: plane-major sweep: block writes 4 KB contiguous of ONE plane (upper bound for NCHW)
__global__ __launch_bounds__(256) void k_seq(float* __restrict__ y, size_t total, float val)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= total) { f4u q = {val, val, val, val}; *reinterpret_cast<f4u*>(y + i) = q; }
}

// mode 5: wave writes 4 consecutive rows-chunks?  block owns 256 px of a row for ALL 64 planes but loops planes in the
// outer loop with all 4 waves on the SAME plane (4 x 256 B = 1 KB... ) -- variant: each wave 64 px x 4 B, 4 waves cover
// 4 different rows of the same plane.
__global__ __launch_bounds__(256) void k_rows(float* __restrict__ y, int H, int W, float val)
{
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int px0 = (blockIdx.x * 64 + lane) * 4, py = blockIdx.y * 4 + r, n = blockIdx.z;
    const size_t HW = (size_t)H * W;
    if (px0 >= W || py >= H) return;
    float* yn = y + (size_t)n * 64 * HW + (size_t)py * W + px0;
    const bool full = px0 + 4 <= W;
#pragma unroll 16
    for (int c = 0; c < 64; ++c) {
        float* dst = yn + (size_t)c * HW;
        if (full) { f4u q = {val, val, val, val}; *reinterpret_cast<f4u*>(dst) = q; }
        else { for (int p = 0; p < 4; ++p) if (px0 + p < W) dst[p] = val; }
    }
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 48, H = 800, W = argc > 2 ? atoi(argv[2]) : 1333;
    const size_t total = (size_t)n * 64 * H * W;
    float* y; hipMalloc(&y, total * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 6; ++mode) {
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((W + 255) / 256, H, n), dim3(256), 0, 0, y, H, W, 1.f);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3((W + 255) / 256, H, n), dim3(256), 0, 0, y, H, W, 1.f);
            if (mode == 2) hipMemsetAsync(y, 0, total * 4, 0);
            if (mode == 3) hipLaunchKernelGGL(k_old, dim3((W + 63) / 64, (H + 3) / 4, n), dim3(256), 0, 0, y, H, W, 1.f);
            if (mode == 4) hipLaunchKernelGGL(k_seq, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, 0, y, total, 1.f);
            if (mode == 5) hipLaunchKernelGGL(k_rows, dim3((W + 255) / 256, (H + 3) / 4, n), dim3(256), 0, 0, y, H, W, 1.f);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        printf("mode %d  n=%d W=%d  %.3f ms  %.2f TB/s\n", mode, n, W, best, total * 4 / best / 1e9);
    }
    return 0;
}
