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
