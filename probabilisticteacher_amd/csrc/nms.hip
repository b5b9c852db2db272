// nms.hip -- batched greedy NMS for gfx950: 64-bit suppression bitmask (one wave64 ballot word per
// 64-box column block) + a chunked in-order scan, all images of a batch in one launch each.
//
// Replaces torchvision.ops.nms (via detectron2 batched_nms) at
// pt/modeling/proposal_generator/proposal_utils.py:140 and pt/modeling/roi_heads/fast_rcnn.py:104.
// Bit-exact against the CPU oracle: IoU = inter / (area_i + area_j - inter), fp32, no FMA contraction
// (-ffp-contract=off), suppression iff IoU > thr (strict), candidates visited in the given
// (descending-score, stable) order.
#include "common.h"

namespace {

typedef unsigned long long u64;

// mask[img][i][w] bit b set  <=>  box j = 64*w + b (j > i) is suppressed by box i.
// grid = (colBlocks, rowBlocks, nimg), block = 64 threads (one wave): thread t owns row box 64*rb + t.
// `cnt` (may be null): only the first cnt[img] boxes of segment img exist (fixed-capacity segments whose fill level is
// known on the device only).
__device__ __forceinline__ int seg_count(const int32_t* __restrict__ seg, const int32_t* __restrict__ cnt, int img)
{
    const int cap = seg[img + 1] - seg[img];
    return cnt ? min(cap, cnt[img]) : cap;
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes,
                                                      const int32_t* __restrict__ seg,
                                                      const int32_t* __restrict__ cnt, float thr, int words,
                                                      int64_t max_count, u64* __restrict__ mask)
{
    const int cb = blockIdx.x, rb = blockIdx.y, img = blockIdx.z;
    if (cb < rb) return;
    const int beg = seg[img], n = seg_count(seg, cnt, img);
    if (rb * 64 >= n || cb * 64 >= n) return;
    __shared__ float4 cbox[64];
    __shared__ float carea[64];
    const int t = threadIdx.x;
    const int cj = cb * 64 + t;
    if (cj < n) {
        const float4 b = reinterpret_cast<const float4*>(boxes)[beg + cj];
        cbox[t] = b;
        carea[t] = (b.z - b.x) * (b.w - b.y);
    }
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= n) return;
    const float4 a = reinterpret_cast<const float4*>(boxes)[beg + i];
    const float aarea = (a.z - a.x) * (a.w - a.y);
    const int lim = min(64, n - cb * 64);
    u64 bits = 0;
    const int start = (cb == rb) ? t + 1 : 0;
    // `inter / uni > thr` decided without the division wherever the quotient is not within 2^-20 of the threshold: the IEEE
    // quotient q = fl(inter / uni) is within 2^-24 (relative) of inter / uni and p = fl(thr uni) within 2^-24 of thr uni, so
    // inter > p (1 + 2^-20) implies q > thr and inter < p (1 - 2^-20) implies q < thr; everything else (and uni <= 0 / NaN) takes
    // the division -- the same decisions bit for bit, a tenth of the instructions for almost every pair
    constexpr float HI = 1.f + 0x1p-20f, LO = 1.f - 0x1p-20f;
    for (int k = start; k < lim; ++k) {
        const float4 b = cbox[k];
        const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
        const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
        float w = xx2 - xx1, h = yy2 - yy1;
        w = w < 0.f ? 0.f : w;
        h = h < 0.f ? 0.f : h;
        const float inter = w * h;
        const float uni = aarea + carea[k] - inter;
        const float p = thr * uni;
        bool sup;
        // (p >= FLT_MIN: the 2^-24 bounds hold for NORMAL products only -- a denormal thr uni, or thr = 0 with an underflowing
        // quotient, takes the division)
        if (uni > 0.f && p >= 1.17549435e-38f && inter > p * HI) sup = true;
        else if (uni > 0.f && p >= 1.17549435e-38f && inter < p * LO) sup = false;
        else sup = inter / uni > thr;
        if (sup) bits |= 1ull << k;
    }
    mask[((size_t)img * max_count + i) * words + cb] = bits;
}

// One workgroup per image walks the boxes in order, 64 at a time: wave 0 resolves the chunk's 64 x 64 diagonal block (only the
// candidates that survive are visited: a bit scan, one readlane pair each), then all four waves fold the kept rows into the
// remaining words of the `removed` vector (a thread per word, the loads of up to sixteen kept rows in flight together).
constexpr int SCAN_THREADS = 256;

__global__ __launch_bounds__(SCAN_THREADS) void nms_scan_kernel(const u64* __restrict__ mask, const int32_t* __restrict__ seg,
                                                                const int32_t* __restrict__ cnt, int words, int64_t max_count,
                                                                int max_keep, int32_t* __restrict__ keep,
                                                                int32_t* __restrict__ keep_count)
{
    extern __shared__ u64 removed[];   // words entries
    __shared__ u64 kept_s;
    __shared__ int count_s;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = seg_count(seg, cnt, img);
    const u64* M = mask + (size_t)img * max_count * words;
    for (int w = tid; w < words; w += SCAN_THREADS) removed[w] = 0;
    if (tid == 0) count_s = 0;
    __syncthreads();
    int count = 0;
    const int chunks = (n + 63) / 64;
    u64 diag_next = (wave == 0 && lane < n) ? M[(size_t)lane * words] : 0;          // diagonal block of chunk 0
    for (int c = 0; c < chunks; ++c) {
        if (wave == 0) {
            const int row = c * 64 + lane;
            const u64 diag = diag_next;
            if (c + 1 < chunks) {                                           // prefetch the next diagonal block: its load
                const int rn = row + 64;                                    // latency hides behind this chunk's resolve
                diag_next = (rn < n) ? M[(size_t)rn * words + c + 1] : 0;
            }
            u64 cur = removed[c];
            if (c == chunks - 1 && (n & 63)) cur |= ~0ull << (n & 63);   // rows past n do not exist
            // candidates still alive, as a wave-uniform scalar pair
            u64 alive = ~(((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur >> 32)) << 32) |
                          (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur));
            u64 kept = 0;
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            while (alive) {
                const int i = __builtin_ctzll(alive);
                kept |= 1ull << i;
                const u64 di = ((u64)(unsigned)__builtin_amdgcn_readlane((int)dhi, i) << 32) |
                               (u64)(unsigned)__builtin_amdgcn_readlane((int)dlo, i);
                alive &= ~(di | (1ull << i));                               // (row i's block only has bits above i)
            }
            // emit kept indices in order, capped at max_keep
            if ((kept >> lane) & 1ull) {
                const int pos = count + __popcll(kept & ((1ull << lane) - 1ull));
                if (pos < max_keep) keep[(size_t)img * max_keep + pos] = row;
            }
            if (lane == 0) {
                kept_s = kept;
                count_s = count + __popcll(kept);
            }
        }
        __syncthreads();
        count = count_s;
        if (count >= max_keep) break;
        const unsigned klo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kept_s);
        const unsigned khi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kept_s >> 32));
        const u64 kept = ((u64)khi << 32) | klo;
        for (int w = c + 1 + tid; w < chunks; w += SCAN_THREADS) {
            u64 acc = removed[w];
            const u64* Mw = M + (size_t)c * 64 * words + w;
            u64 k = kept;
            while (k) {
                u64 v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (k) {                                                 // (scalar: k is wave-uniform)
                        const int i = __builtin_ctzll(k);
                        k &= k - 1;
                        v[j] = Mw[(size_t)i * words];
                    } else v[j] = 0ull;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) acc |= v[j];
            }
            removed[w] = acc;
        }
        __syncthreads();
    }
    if (tid == 0) keep_count[img] = count < max_keep ? count : max_keep;
}

}  // namespace

extern "C" {

int64_t ptmi_nms_ws_bytes(int64_t max_count, int nimg)
{
    const int64_t words = (max_count + 63) / 64;
    return (int64_t)nimg * max_count * words * 8;
}

int ptmi_nms_batched(const float* boxes, const int32_t* seg_offsets, const int32_t* seg_counts, int nimg,
                     int64_t max_count, float thr, int max_keep, int32_t* keep_out, int32_t* keep_count, void* ws,
                     ptmi_stream_t s)
{
    PTMI_CHECK_ARG(seg_offsets && keep_out && keep_count && nimg > 0 && max_count >= 0 && max_keep > 0,
                   "nms_batched: bad args");
    PTMI_CHECK_ARG(max_count == 0 || (boxes && ws), "nms_batched: boxes / workspace missing");
    hipStream_t st = (hipStream_t)s;
    if (max_count == 0) {        // no candidate in any image (boxes and ws may be null)
        hipError_t e = hipMemsetAsync(keep_count, 0, sizeof(int32_t) * (size_t)nimg, st);
        if (e != hipSuccess) { ptmi_set_error("nms_batched: memset failed"); return -2; }
        return 0;
    }
    const int words = (int)((max_count + 63) / 64);
    PTMI_CHECK_ARG((size_t)words * 8 <= 64 * 1024, "nms_batched: max_count %lld too large", (long long)max_count);
    dim3 grid(words, words, nimg);
    hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(64), 0, st, boxes, seg_offsets, seg_counts, thr, words, max_count,
                       reinterpret_cast<u64*>(ws));
    PTMI_LAUNCH_CHECK("nms_mask");
    hipLaunchKernelGGL(nms_scan_kernel, dim3(nimg), dim3(SCAN_THREADS), (size_t)words * 8, st, reinterpret_cast<const u64*>(ws),
                       seg_offsets, seg_counts, words, max_count, max_keep, keep_out, keep_count);
    PTMI_LAUNCH_CHECK("nms_scan");
    return 0;
}

}  // extern "C"
