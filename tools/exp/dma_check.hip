// does global_load_lds dwordx4 accept dword-aligned (not 16-B aligned) global addresses?  prints OK/BAD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;
__global__ void k(const float* src, float* dst, int mis)
{
    __shared__ __attribute__((aligned(16))) float lds[256];
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + lane * 4 + mis), (lds_void_t*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) dst[i] = lds[i];
}
int main()
{
    float h[512], o[256]; for (int i = 0; i < 512; ++i) h[i] = (float)i;
    float *s, *d; hipMalloc(&s, sizeof h); hipMalloc(&d, sizeof o); hipMemcpy(s, h, sizeof h, hipMemcpyHostToDevice);
    for (int mis = 0; mis < 4; ++mis) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, mis);
        hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 256; ++i) bad += (o[i] != (float)(i + mis));
        printf("misalign %d floats: %s (o[0..4] = %g %g %g %g %g)\n", mis, bad ? "BAD" : "OK", o[0], o[1], o[2], o[3], o[4]);
    }
    return 0;
}
