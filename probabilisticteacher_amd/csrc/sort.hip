// sort.hip -- segmented stable descending sort of fp32 keys with int32 payload (gfx950).
//
// Replaces torch.sort(descending=True) at pt/modeling/proposal_generator/proposal_utils.py:87 and the
// score sort inside torchvision nms (proposal_utils.py:140, fast_rcnn.py:104).  rocPRIM's segmented
// radix sort is stable, so equal keys keep ascending original index (the build's tie policy, shared
// with the CPU oracle's descending_order()).  The payload written is the index WITHIN the segment.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace {

__global__ void iota_segments_kernel(int32_t* __restrict__ idx, const int32_t* __restrict__ seg, int nseg)
{
    const int sgm = blockIdx.y;
    if (sgm >= nseg) return;
    const int beg = seg[sgm], end = seg[sgm + 1];
    for (int i = beg + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) idx[i] = i - beg;
}

size_t rocprim_temp_bytes(int64_t total, int nseg)
{
    size_t bytes = 0;
    (void)rocprim::segmented_radix_sort_pairs_desc(nullptr, bytes, (const float*)nullptr, (float*)nullptr,
                                                   (const int32_t*)nullptr, (int32_t*)nullptr, (unsigned)total,
                                                   (unsigned)nseg, (const int32_t*)nullptr, (const int32_t*)nullptr, 0,
                                                   32, (hipStream_t)0);
    return bytes;
}

}  // namespace

extern "C" {

int64_t ptmi_segsort_ws_bytes(int64_t total, int nseg)
{
    if (total <= 0 || nseg <= 0) return 256;
    const size_t t = rocprim_temp_bytes(total, nseg);
    return (int64_t)(((t + 255) / 256) * 256 + (size_t)total * 4 + 256);
}

int ptmi_segsort_desc(const float* keys_in, float* keys_out, int32_t* idx_out, int64_t total, int nseg,
                      const int32_t* seg_offsets, void* ws, int64_t ws_bytes, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(total >= 0 && nseg > 0 && seg_offsets, "segsort_desc: bad args");
    if (total == 0) return 0;
    PTMI_CHECK_ARG(keys_in && keys_out && idx_out && ws, "segsort_desc: null buffer");
    PTMI_CHECK_ARG(total < (1ll << 31), "segsort_desc: too many keys");
    hipStream_t st = (hipStream_t)s;
    size_t temp = rocprim_temp_bytes(total, nseg);
    const size_t temp_al = ((temp + 255) / 256) * 256;
    PTMI_CHECK_ARG((int64_t)(temp_al + (size_t)total * 4) <= ws_bytes, "segsort_desc: workspace too small");
    int32_t* idx_in = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + temp_al);
    int64_t per = total / nseg + 1;
    unsigned bx = (unsigned)((per + 255) / 256);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(iota_segments_kernel, dim3(bx, nseg), dim3(256), 0, st, idx_in, seg_offsets, nseg);
    PTMI_LAUNCH_CHECK("segsort_iota");
    hipError_t e = rocprim::segmented_radix_sort_pairs_desc(ws, temp, keys_in, keys_out, (const int32_t*)idx_in,
                                                            idx_out, (unsigned)total, (unsigned)nseg, seg_offsets,
                                                            seg_offsets + 1, 0, 32, st);
    if (e != hipSuccess) {
        ptmi_set_error("segsort_desc: rocprim failed: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}

}  // extern "C"
