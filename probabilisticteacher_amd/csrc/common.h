// common.h -- shared helpers for libptmi355 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ptmi355.h"

void ptmi_set_error(const char* fmt, ...);

#define PTMI_CHECK_ARG(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            ptmi_set_error(__VA_ARGS__);     \
            return -1;                       \
        }                                    \
    } while (0)

#define PTMI_LAUNCH_CHECK(name)                                                  \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            ptmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return -2;                                                           \
        }                                                                        \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ptmi_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ptmi_bf16x4 __attribute__((ext_vector_type(4)));

// ---- buffer -> LDS DMA helpers (buffer_load_dword[x4] ... lds) ----------------------------------------------
// Raw buffer resource over [base, base + bytes): lanes whose offset is >= bytes (e.g. 0xFFFFFFFF) are zero-filled
// by the range check without touching memory.  0x00020000 = DATA_FORMAT 32 (gfx9 raw buffer).
typedef __attribute__((address_space(3))) void ptmi_lds_void_t;
typedef __attribute__((address_space(3))) f32x4 ptmi_lds_f32x4_t;
typedef __attribute__((address_space(3))) float ptmi_lds_f32_t;

// A wave-uniform pointer the compiler may have parked in VGPRs: pull it back into SGPRs (a descriptor built from
// VGPRs turns every DMA into a waterfall loop).
__device__ __forceinline__ void* ptmi_uniform_ptr(const void* p)
{
    const unsigned long long a = (unsigned long long)p;
    return (void*)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ptmi_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(ptmi_uniform_ptr(base), 0, __builtin_amdgcn_readfirstlane((int)bytes),
                                             0x00020000);
}
// 16 B per lane: LDS destination = lds_wave_base + lane * 16 B; global source = base + voff + soff
__device__ __forceinline__ void ptmi_bdma16(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff, float* lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (ptmi_lds_void_t*)lds_wave_base, 16, (int)voff, soff, 0, 0);
}

// eight fp32 values -> the bf16x8 operand of v_mfma_f32_32x32x16_bf16 (v_cvt_pk_bf16_f32: round to nearest even)
__device__ __forceinline__ ptmi_bf16x8 ptmi_pack_bf16x8(const float* f)
{
    ptmi_bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
    return v;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// wave64 sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Block-wide sum for 256-thread blocks; result valid in thread 0.  smem: >= 4 floats.
__device__ __forceinline__ float block_sum_256(float v, float* smem) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) smem[wv] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    __syncthreads();
    return r;
}
