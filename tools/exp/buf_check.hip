// experiment: semantics of buffer_load ... lds (raw buffer -> LDS DMA) on gfx950: OOB lanes, x4, imm offset
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ void k(const float* src, float* dst, int mode)
{
    __shared__ __attribute__((aligned(16))) float lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -7.f;
    __syncthreads();
    // base points 8 floats BEFORE src+16
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + 8), 0, 0xFFFFFFFF, 0x00020000);
    if (mode == 0) {            // dword, odd lanes invalid (-1)
        int voff = (lane & 1) ? -1 : lane * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds, 4, voff, 0, 0, 0);
    } else if (mode == 1) {     // x4, lanes >= 32 invalid, misaligned base (+1 float)
        __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(src + 9), 0, 0xFFFFFFFF, 0x00020000);
        int voff = (lane >= 32) ? -1 : lane * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_void_t*)lds, 16, voff, 0, 0, 0);
    } else if (mode == 2) {     // dword with imm offset 256: where does it land?
        int voff = lane * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds, 4, voff, 0, 256, 0);
    } else if (mode == 3) {     // soffset = 1024 bytes
        int voff = lane * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds, 4, voff, 1024, 0, 0);
    } else {                    // small num_records = 100 bytes: lanes >= 25 OOB
        __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc((void*)(src + 8), 0, 100, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r3, (lds_void_t*)lds, 4, lane * 4, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) dst[i] = lds[i];
}
int main()
{
    static float h[4096], o[1024];
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *s, *d; hipMalloc(&s, sizeof h); hipMalloc(&d, sizeof o); hipMemcpy(s, h, sizeof h, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, mode);
        hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
        printf("mode %d:", mode);
        for (int i = 0; i < 10; ++i) printf(" %g", o[i]);
        printf(" | [24..27] %g %g %g %g | [62..66] %g %g %g %g %g | [126..130] %g %g %g %g %g | [254..257] %g %g %g %g\n", o[24], o[25], o[26], o[27],
               o[62], o[63], o[64], o[65], o[66], o[126], o[127], o[128], o[129], o[130], o[254], o[255], o[256], o[257]);
    }
    return 0;
}
