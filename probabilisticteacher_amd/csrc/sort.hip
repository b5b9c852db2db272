// sort.hip -- segmented stable descending sort of fp32 keys with int32 payload (gfx950).
//
// Replaces torch.sort(descending=True) at pt/modeling/proposal_generator/proposal_utils.py:87 and the
// score sort inside torchvision nms (proposal_utils.py:140, fast_rcnn.py:104).  Equal keys keep ascending original index
// (the build's tie policy, shared with the CPU oracle's descending_order()); -0.0 ties with +0.0 and NaN sorts first, as in
// torch.sort.  The payload written is the index WITHIN the segment.
//   * segments of up to 16 384 keys (the re-scored 12 000 proposals, the 8 x 2 000 (roi, class) candidates of an image), and the
//     first `topk` <= 16 384 entries of longer ones (12 000 of 37 350 anchor scores): ONE workgroup per segment keeps
//     (ordering key, index) in LDS -- longer segments first pass a 4 x 8-bit radix select + ordered compaction of their topk
//     smallest ordering keys -- and runs a bitonic network on the pairs (the index makes the order total, so the unstable
//     network yields the stable result); exchanges with a stride below 128 stay inside a wave's 128-element block and need
//     no workgroup barrier (77 of the 105 steps at 16 384 elements);
//   * anything larger: rocPRIM's segmented radix sort (stable; SURVEY N10 allows it), which occupies only one workgroup per
//     segment as well and needs ~0.55 ms for 32 x 37 350 keys where the kernel above needs ~0.1 ms.
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

constexpr int LS_CAP = 16384, LS_THREADS = 1024;

// ordering key: ascending unsigned order of ord(f) == descending order of f, NaN first, -0.0 == +0.0
__device__ __forceinline__ unsigned ord_of(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0u;                       // NaN
    if (u == 0x80000000u) u = 0u;                                        // -0.0
    const unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // ascending order of f
    return ~asc;                                                         // (>= 1 for every non-NaN: asc <= 0xFF800000 | ...)
}

__global__ __launch_bounds__(LS_THREADS) void segsort_lds_kernel(const float* __restrict__ keys_in, float* __restrict__ keys_out,
                                                                 int32_t* __restrict__ idx_out, const int32_t* __restrict__ seg,
                                                                 int topk)
{
    __shared__ unsigned sk[LS_CAP];                  // ordering keys
    __shared__ unsigned short si[LS_CAP];            // index within the segment (< 65 536 where it is stored: see the launcher)
    __shared__ int hist[256];
    __shared__ unsigned sel_prefix;
    __shared__ int sel_want, run_less, run_tie;
    __shared__ int wcnt[2][LS_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int beg = seg[blockIdx.x], n = seg[blockIdx.x + 1] - beg;
    if (n <= 0) return;
    if (n > LS_CAP && (topk <= 0 || topk > LS_CAP || n >= 65536)) {
        // the caller's max_len / topk did not describe this segment: no GPU fault (a trap aborts the process without a message --
        // ADVICE r4) -- the segment's output is NaN keys over index 0.  The host wrapper (ops.segsort_desc) checks the callers'
        // host-side segment lengths against max_len before the launch, so the product path cannot get here; a raw
        // C-ABI caller sees the NaNs
        for (int i = tid; i < n; i += LS_THREADS) { keys_out[beg + i] = __builtin_nanf(""); idx_out[beg + i] = 0; }
        return;
    }
    const float* kin = keys_in + beg;
    int cnt;                                          // entries in LDS
    if (n <= LS_CAP) {
        cnt = n;
        for (int i = tid; i < n; i += LS_THREADS) { sk[i] = ord_of(kin[i]); si[i] = (unsigned short)i; }
    } else {
        // the topk smallest ordering keys: radix select of the threshold (4 x 8 bits), then ordered compaction
        cnt = topk;
        if (tid == 0) { sel_prefix = 0u; sel_want = topk; run_less = 0; run_tie = 0; }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned pf = sel_prefix;
            for (int i = tid; i < n; i += LS_THREADS) {
                const unsigned d = ord_of(kin[i]);
                if ((d & himask) == pf) atomicAdd(&hist[(d >> shift) & 255], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int w = sel_want, b = 0;
                for (; b < 255; ++b) {
                    if (w <= hist[b]) break;
                    w -= hist[b];
                }
                sel_want = w;
                sel_prefix = pf | ((unsigned)b << shift);
            }
            __syncthreads();
        }
        const unsigned T = sel_prefix;
        const int tie_take = sel_want;                // entries equal to T that belong to the selection (lowest indices first)
        // n_less = topk - tie_take entries are below T; they keep their index order in LDS, the ties follow them
        const int n_less = topk - tie_take;
        for (int base = 0; base < n; base += LS_THREADS) {
            const int i = base + tid;
            unsigned d = 0xFFFFFFFFu;
            bool less = false, tied = false;
            if (i < n) { d = ord_of(kin[i]); less = d < T; tied = d == T; }
            const unsigned long long m0 = __ballot(less), m1 = __ballot(tied);
            if (lane == 0) { wcnt[0][wv] = __popcll(m0); wcnt[1][wv] = __popcll(m1); }
            __syncthreads();
            int b0 = run_less, b1 = run_tie;
            for (int w2 = 0; w2 < wv; ++w2) { b0 += wcnt[0][w2]; b1 += wcnt[1][w2]; }
            const unsigned long long below = (1ull << lane) - 1ull;
            if (less) {
                const int pos = b0 + __popcll(m0 & below);
                sk[pos] = d; si[pos] = (unsigned short)i;
            } else if (tied) {
                const int r = b1 + __popcll(m1 & below);
                if (r < tie_take) { sk[n_less + r] = d; si[n_less + r] = (unsigned short)i; }
            }
            __syncthreads();
            if (tid == 0) {
                int a0 = 0, a1 = 0;
                for (int w2 = 0; w2 < LS_THREADS / 64; ++w2) { a0 += wcnt[0][w2]; a1 += wcnt[1][w2]; }
                run_less += a0; run_tie += a1;
            }
            __syncthreads();
        }
    }
    // pad to a power of two (>= 128: a wave's block) with entries that sort last
    int m = 128;
    while (m < cnt) m <<= 1;
    for (int i = cnt + tid; i < m; i += LS_THREADS) { sk[i] = 0xFFFFFFFFu; si[i] = 0xFFFFu; }
    __syncthreads();
    // bitonic network on (key, index) pairs, ascending.  Pair p of a step exchanges elements i = 2p - (p mod stride) and
    // i + stride; thread t handles pairs t, t + 1024, ...: a wave's 64 pairs lie in ONE aligned block of 128 elements when
    // stride <= 64, and it is the same block in every such step -- a wave barrier orders those steps.
    const int half = m >> 1;
    for (int size = 2; size <= m; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = tid; p < half; p += LS_THREADS) {
                const int i = ((p & ~(stride - 1)) << 1) | (p & (stride - 1)), j = i + stride;
                const unsigned ki = sk[i], kj = sk[j];
                const unsigned short ii = si[i], ij = si[j];
                const bool up = (i & size) == 0;
                const bool gt = ki > kj || (ki == kj && ii > ij);
                if (gt == up) { sk[i] = kj; sk[j] = ki; si[i] = ij; si[j] = ii; }
            }
            if (stride > 64) __syncthreads();
            else if (stride == 1 && (size << 1) > 128 && size < m) __syncthreads();     // the next step's first stride is >= 128
            else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"), __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    for (int i = tid; i < cnt; i += LS_THREADS) {
        const unsigned short ix = si[i];
        keys_out[beg + i] = kin[ix];
        idx_out[beg + i] = (int)ix;
    }
    // entries past topk of a long segment: defined, never to be used (key -inf, index 0)
    for (int i = cnt + tid; i < n; i += LS_THREADS) { keys_out[beg + i] = -INFINITY; idx_out[beg + i] = 0; }
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

int ptmi_segsort_topk_fits(int64_t max_len, int64_t topk)
{
    if (max_len <= LS_CAP) return 1;
    return topk > 0 && topk <= LS_CAP && max_len < 65536 ? 1 : 0;
}

int ptmi_segsort_topk_desc(const float* keys_in, float* keys_out, int32_t* idx_out, int nseg, const int32_t* seg_offsets,
                           int64_t max_len, int64_t topk, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(nseg > 0 && seg_offsets && max_len >= 0, "segsort_topk_desc: bad args");
    if (max_len == 0) return 0;
    PTMI_CHECK_ARG(keys_in && keys_out && idx_out, "segsort_topk_desc: null buffer");
    PTMI_CHECK_ARG(ptmi_segsort_topk_fits(max_len, topk), "segsort_topk_desc: segments of %lld keys (topk %lld) do not fit the LDS sort",
                   (long long)max_len, (long long)topk);
    hipLaunchKernelGGL(segsort_lds_kernel, dim3(nseg), dim3(LS_THREADS), 0, (hipStream_t)s, keys_in, keys_out, idx_out, seg_offsets,
                       (int)(max_len > LS_CAP ? topk : 0));
    PTMI_LAUNCH_CHECK("segsort_topk_desc");
    return 0;
}

}  // extern "C"
