// xcc_probe: which XCC_ID does workgroup b of a 256 / 1024-workgroup launch see?  (hipcc --offload-arch=gfx950 -o _bin/xcc_probe xcc_probe.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { o[2 * blockIdx.x] = v; o[2 * blockIdx.x + 1] = hw; }
}
int main() {
    unsigned* d; hipMalloc(&d, 8 * 2048); unsigned h[4096];
    for (int grid : {256, 1024}) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d); hipMemcpy(h, d, 8 * grid, hipMemcpyDeviceToHost);
        printf("grid %d: raw XCC_ID of blocks 0..31:", grid);
        for (int i = 0; i < 32; ++i) printf(" %x", h[2 * i]);
        int same = 0; for (int i = 0; i < grid; ++i) same += (h[2 * i] & 7u) == (unsigned)(i & 7);
        int hist[16] = {0}; for (int i = 0; i < grid; ++i) hist[h[2 * i] & 15u]++;
        printf("\n  (XCC_ID & 7) == (block & 7) for %d of %d blocks; histogram of XCC_ID & 15:", same, grid);
        for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
        printf("\n");
    }
    return 0;
}
