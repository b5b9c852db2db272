// Microbenchmark (experiment, not product): throughput of global_load_lds dword vs dwordx4 per CU, aligned and
// dword-misaligned sources, as a function of waves per CU.  Prints cycles per DMA instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;

template <int SIZE, int BATCH>
__global__ void k(const float* __restrict__ src, long long* out, int iters, int misalign, size_t span_floats)
{
    __shared__ __attribute__((aligned(16))) float lds[16 * 1024];   // 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    float* base = lds + (size_t)wave * (16 * 1024 / 16) ;            // 1024 floats per wave (up to 16 waves)
    size_t off = ((size_t)blockIdx.x * 7919 * 64 + (size_t)wave * 4096) & (span_floats - 1);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const float* g = src + ((off + (size_t)b * 256 * (SIZE / 4) + (size_t)lane * (SIZE / 4) + misalign) & (span_floats - 1));
            if constexpr (SIZE == 16) __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)(base + (b & 3) * 256), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)(base + (b & 3) * 64), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        off = (off + 65536 + 64) & (span_floats - 1);
    }
    const long long t1 = clock64();
    if (lane == 0) out[(size_t)blockIdx.x * nw + wave] = t1 - t0;
    if (tid == 12345) out[0] = (long long)lds[5];
}

template <int SIZE, int BATCH>
void run(const float* src, long long* dout, int threads, int blocks_per_cu, int misalign, size_t span, const char* tag)
{
    const int blocks = 256 * blocks_per_cu, iters = 200, nw = threads / 64;
    hipLaunchKernelGGL((k<SIZE, BATCH>), dim3(blocks), dim3(threads), 0, 0, src, dout, iters, misalign, span);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SIZE, BATCH>), dim3(blocks), dim3(threads), 0, 0, src, dout, iters, misalign, span);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)blocks * nw);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double instr_per_cu = (double)blocks_per_cu * nw * iters * BATCH;
    printf("%-28s size %2d batch %2d waves/CU %2d: %8.0f cyc/wave-loop, %6.1f cyc per DMA instr per CU, %6.2f B/clk/CU, %.3f ms (%.0f GB/s chip)\n",
           tag, SIZE, BATCH, blocks_per_cu * nw, avg, avg / instr_per_cu, 64.0 * SIZE * instr_per_cu / avg, ms,
           256.0 * instr_per_cu * 64 * SIZE / ms / 1e6);
}

int main()
{
    const size_t span_l2 = 1u << 20;        // 4 MB: L2/MALL resident
    const size_t span_hbm = 1u << 28;       // 1 GB: streams from HBM
    float* src; hipMalloc(&src, (span_hbm + 4096) * 4); hipMemset(src, 0, (span_hbm + 4096) * 4);
    long long* dout; hipMalloc(&dout, 8 * 256 * 16 * 16);
    for (int bpc : {1, 2, 4}) {
        run<4, 8>(src, dout, 256, bpc, 0, span_l2, "L2-resident dword");
        run<16, 8>(src, dout, 256, bpc, 0, span_l2, "L2-resident x4 aligned");
        run<16, 8>(src, dout, 256, bpc, 1, span_l2, "L2-resident x4 misaligned+1");
        run<4, 16>(src, dout, 256, bpc, 0, span_l2, "L2-resident dword b16");
        run<16, 2>(src, dout, 256, bpc, 0, span_l2, "L2-resident x4 b2");
    }
    for (int bpc : {2, 4}) {
        run<4, 8>(src, dout, 256, bpc, 0, span_hbm, "HBM dword");
        run<16, 8>(src, dout, 256, bpc, 0, span_hbm, "HBM x4 aligned");
        run<16, 8>(src, dout, 256, bpc, 3, span_hbm, "HBM x4 misaligned+3");
    }
    // correctness of a misaligned x4 DMA
    return 0;
}
