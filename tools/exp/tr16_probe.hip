// tools/exp/tr16_probe.hip -- what ds_read_b64_tr_b16 returns (gfx950): every lane points at 4 consecutive 16-bit values of an LDS
// image holding its own index; prints, per lane, the four values it receives.      hipcc --offload-arch=gfx950 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4 lds_s4;
__global__ void k(s4* o, int mode)
{
    __shared__ __attribute__((aligned(16))) short l[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) l[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // mode 0: lane a -> element 4a (lane-linear);  mode 1: the c8 gather of the P8 weight-gradient operand:
    //         lane a of a 16-lane group -> pixel a/4, channel quad a%4 (2 c8 planes of 16 pixels each at 0 and 1024 elements)
    int e;
    if (mode == 0) e = 4 * lane;
    else {
        const int g = lane >> 4, a = lane & 15, pix = g * 4 + a / 4, q = a % 4;
        e = (q >> 1) * 1024 + pix * 8 + (q & 1) * 4;
    }
    o[lane] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(l + e));
}
int main()
{
    s4* d;
    hipMalloc(&d, 64 * sizeof(s4));
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        s4 h[64];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int i = 0; i < 64; ++i) printf("lane %2d: %5d %5d %5d %5d\n", i, h[i][0], h[i][1], h[i][2], h[i][3]);
    }
    return 0;
}
