// Experiment (not part of the product): issue rate of fp32 VALU instructions of ONE wave on a SIMD of gfx950 as a function of the
// distance between dependent instructions -- the situation of the F(4x4,3x3) kernels' epilogues (one wave per SIMD, VALU only).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/valu_ilp.hip -o gpurun_out/valu_ilp && gpurun_out/valu_ilp
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP, int KIND>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float a, float b)
{
    float x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 48 / ILP; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));            // 4-byte VOP2
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));            // 8-byte VOP3
                if (KIND == 2) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
                if (KIND == 3) asm volatile("v_fmac_f32_e32 %0, 0x3f400000, %1" : "+v"(x[i]) : "v"(b));           // literal
                if (KIND == 4) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));            // SGPR operand
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// accumulator reads between arithmetic
template <int ILP>
__global__ __launch_bounds__(64) void kacc(float* out, unsigned long long* cyc, float a)
{
    float x[ILP];
    float acc = threadIdx.x;
    asm volatile("v_accvgpr_write_b32 a0, %0" ::"v"(acc));
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 48 / ILP; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                float r;
                asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(r));
                asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(r));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ILP, int KIND>
void run(const char* name, float* out, unsigned long long* cyc, int blocks)
{
    unsigned long long h[4096];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const double n = 64.0 * (48 / ILP) * ILP;
    printf("%-28s ILP %2d  blocks %4d: %6.2f cycles per instruction (block 0)\n", name, ILP, blocks, h[0] / n);
}
template <int ILP>
void runacc(float* out, unsigned long long* cyc)
{
    unsigned long long h;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((kacc<ILP>), dim3(1), dim3(64), 0, 0, out, cyc, 1.f); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-28s ILP %2d: %6.2f cycles per (read + add) pair\n", "accvgpr_read + v_add", ILP, h / (64.0 * (48 / ILP) * ILP));
}

int main()
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 64 * 4); hipMalloc(&cyc, 4096 * 8);
#define ALL(KIND, name) run<1, KIND>(name, out, cyc, 1); run<2, KIND>(name, out, cyc, 1); run<3, KIND>(name, out, cyc, 1); run<4, KIND>(name, out, cyc, 1); \
    run<6, KIND>(name, out, cyc, 1); run<8, KIND>(name, out, cyc, 1); run<12, KIND>(name, out, cyc, 1);
    ALL(0, "v_fmac_f32_e32 (VOP2)")
    ALL(1, "v_fma_f32 (VOP3)")
    ALL(2, "v_add_f32_e32")
    ALL(3, "v_fmac_f32 literal")
    ALL(4, "v_fmac_f32 SGPR")
    run<8, 0>("v_fmac, 1024 blocks (4/CU)", out, cyc, 1024);
    run<8, 0>("v_fmac, 2048 blocks (8/CU)", out, cyc, 2048);
    runacc<1>(out, cyc); runacc<4>(out, cyc); runacc<8>(out, cyc);
    return 0;
}
