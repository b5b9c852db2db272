// experiment: raw-buffer range check with SGPR offset, and 32-bit wrap of voffset + soffset (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* dst, int so, unsigned sentinel)
{
    const int lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 100, 0x00020000);
    const unsigned voff = (lane & 1) ? sentinel : (unsigned)lane * 4u;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 1.f + lane), r, (int)voff, so, 0);
}
int main()
{
    float h[128]; float* d; hipMalloc(&d, sizeof h);
    for (unsigned sentinel : {0xFFFFFFFFu, 0xFFFFFFC0u, 0x80000000u}) {
        for (int so : {0, 64}) {
            hipMemset(d, 0, sizeof h);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, so, sentinel);
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("sentinel %08x soffset %2d: written:", sentinel, so);
            for (int i = 0; i < 128; ++i) if (h[i] != 0.f) printf(" [%d]=%g", i, h[i]);
            printf("\n");
        }
    }
    return 0;
}
