// boxes.hip -- anchor grid, box codec, IoU matcher and RPN proposal preparation for gfx950.
// Compiled with -ffp-contract=off: the integer outputs (match indices, labels, valid masks) must be
// bit-exact against the CPU oracle, so the fp32 expression trees below mirror the reference's
// (pt/modeling/box_regression.py, D2 pairwise_iou / Matcher -- SURVEY.md A.2/A.3) op for op.
#include "common.h"
#include <algorithm>

namespace {

__global__ void grid_anchors_kernel(const float* __restrict__ cell, float* __restrict__ out, int h, int w, int A,
                                    float stride, float offset)
{
    const int64_t total = (int64_t)h * w * A;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int a = i % A;
        const int64_t p = i / A;
        const int x = p % w, y = p / w;
        // torch.arange(offset*stride, W*stride, step=stride): start + k*step
        const float sx = offset * stride + (float)x * stride;
        const float sy = offset * stride + (float)y * stride;
        const float4 c = reinterpret_cast<const float4*>(cell)[a];
        reinterpret_cast<float4*>(out)[i] = make_float4(sx + c.x, sy + c.y, sx + c.z, sy + c.w);
    }
}

// box_regression.py:101-139: decode one (dx, dy, dw, dh) quadruple on box b.
__device__ __forceinline__ float4 decode_box(const float4 b, const float* __restrict__ d, float wx, float wy, float ww,
                                             float wh, float clampv)
{
    const float widths = b.z - b.x, heights = b.w - b.y;
    const float cx = b.x + 0.5f * widths, cy = b.y + 0.5f * heights;
    const float dx = d[0] / wx, dy = d[1] / wy;
    float dw = d[2] / ww, dh = d[3] / wh;
    dw = fminf(dw, clampv);
    dh = fminf(dh, clampv);
    const float pcx = dx * widths + cx, pcy = dy * heights + cy;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    return make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
}

// One thread per (row, k) decoded box.
__global__ void apply_deltas_kernel(const float* __restrict__ deltas, const float* __restrict__ boxes,
                                    float* __restrict__ out, int64_t rows, int k, int dstride, int64_t nb,
                                    float wx, float wy, float ww, float wh, float clampv)
{
    const int64_t total = rows * k;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / k;
        const int j = i % k;
        const float4 o4 = decode_box(reinterpret_cast<const float4*>(boxes)[row % nb], deltas + row * dstride + 4 * j,
                                     wx, wy, ww, wh, clampv);
        float* o = out + row * (int64_t)(4 * k) + 4 * j;
        o[0] = o4.x; o[1] = o4.y; o[2] = o4.z; o[3] = o4.w;
    }
}

// box_regression.py:66-99
__device__ __forceinline__ void get_deltas_row(const float4 s, const float4 t, float wx, float wy, float ww,
                                               float wh, float* o)
{
    const float sw = s.z - s.x, sh = s.w - s.y;
    const float sx = s.x + 0.5f * sw, sy = s.y + 0.5f * sh;
    const float tw = t.z - t.x, th = t.w - t.y;
    const float tx = t.x + 0.5f * tw, ty = t.y + 0.5f * th;
    o[0] = wx * (tx - sx) / sw;
    o[1] = wy * (ty - sy) / sh;
    o[2] = ww * logf(tw / sw + 1e-9f);
    o[3] = wh * logf(th / sh + 1e-9f);
}

__global__ void get_deltas_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                  float* __restrict__ out, int64_t rows, float wx, float wy, float ww, float wh)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows;
         i += (int64_t)gridDim.x * blockDim.x) {
        float o[4];
        get_deltas_row(reinterpret_cast<const float4*>(src)[i], reinterpret_cast<const float4*>(tgt)[i], wx, wy,
                       ww, wh, o);
        reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// d(loss)/d(src box) of get_deltas, scattered (atomic) into dsrc[src_index[i]].
__global__ void get_deltas_bwd_src_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                          const float* __restrict__ dd, const int64_t* __restrict__ sidx,
                                          int64_t rows, float wx, float wy, float ww, float wh,
                                          float* __restrict__ dsrc)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 s = reinterpret_cast<const float4*>(src)[i];
        const float4 t = reinterpret_cast<const float4*>(tgt)[i];
        const float4 g = reinterpret_cast<const float4*>(dd)[i];
        const float sw = s.z - s.x, sh = s.w - s.y;
        const float sx = s.x + 0.5f * sw, sy = s.y + 0.5f * sh;
        const float tw = t.z - t.x, th = t.w - t.y;
        const float tx = t.x + 0.5f * tw, ty = t.y + 0.5f * th;
        // dx = wx*(tx-sx)/sw ; dw = ww*log(tw/sw + 1e-9)
        const float ddx_dsx = -wx / sw, ddx_dsw = -wx * (tx - sx) / (sw * sw);
        const float ddw_dsw = ww * (-tw / (sw * sw)) / (tw / sw + 1e-9f);
        const float ddy_dsy = -wy / sh, ddy_dsh = -wy * (ty - sy) / (sh * sh);
        const float ddh_dsh = wh * (-th / (sh * sh)) / (th / sh + 1e-9f);
        const float g_sx = g.x * ddx_dsx, g_sw = g.x * ddx_dsw + g.z * ddw_dsw;
        const float g_sy = g.y * ddy_dsy, g_sh = g.y * ddy_dsh + g.w * ddh_dsh;
        // sx = x1 + 0.5*(x2-x1), sw = x2 - x1
        float* o = dsrc + 4 * sidx[i];
        atomicAdd(o + 0, 0.5f * g_sx - g_sw);
        atomicAdd(o + 2, 0.5f * g_sx + g_sw);
        atomicAdd(o + 1, 0.5f * g_sy - g_sh);
        atomicAdd(o + 3, 0.5f * g_sy + g_sh);
    }
}

// ---------------------------------------------------------------------------------- IoU + Matcher
__device__ __forceinline__ float iou_pair(const float4 g, float garea, const float4 b, float barea)
{
    // D2 pairwise_iou: wh = min(rb) - max(lt), clamp(min=0), inter = w*h,
    // iou = inter > 0 ? inter / (area1 + area2 - inter) : 0
    float w = fminf(g.z, b.z) - fmaxf(g.x, b.x);
    float h = fminf(g.w, b.w) - fmaxf(g.y, b.y);
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float inter = w * h;
    return inter > 0.f ? inter / (garea + barea - inter) : 0.f;
}

constexpr int MAX_GT_LDS = 2048;

__device__ __forceinline__ void iou_match_pass1_body(const float* __restrict__ gt, const float* __restrict__ boxes,
                                                     int m, int64_t nb, int64_t* __restrict__ midx,
                                                     float* __restrict__ miou, int* __restrict__ best_bits)
{
    __shared__ float4 sg[MAX_GT_LDS / 8];
    __shared__ float sa[MAX_GT_LDS / 8];
    __shared__ float swmax[4][256];
    const int64_t j = blockIdx.x * 256ll + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    float4 b = make_float4(0, 0, 0, 0);
    float barea = 0.f;
    if (j < nb) {
        b = reinterpret_cast<const float4*>(boxes)[j];
        barea = (b.z - b.x) * (b.w - b.y);
    }
    float best = -1.f;
    int64_t bi = 0;
    for (int g0 = 0; g0 < m; g0 += 256) {
        __syncthreads();
        const int gi = g0 + threadIdx.x;
        if (gi < m) {
            const float4 g = reinterpret_cast<const float4*>(gt)[gi];
            sg[threadIdx.x] = g;
            sa[threadIdx.x] = (g.z - g.x) * (g.w - g.y);
        }
        __syncthreads();
        const int cnt = min(256, m - g0);
        for (int i = 0; i < cnt; ++i) {
            const float v = iou_pair(sg[i], sa[i], b, barea);
            if (v > best) { best = v; bi = g0 + i; }      // first maximum (torch.max dim=0)
            // per-gt best over all boxes: lanes past nb hold a zero box (IoU 0), so the unconditional wave
            // reduction is safe.  Wave maxima go to LDS; ONE global atomic per workgroup per gt afterwards (every
            // wave hitting the same address from inside this loop serialised in L2: 86 us per call).
            float wmax = v;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
            if ((threadIdx.x & 63) == 0) swmax[wave][i] = wmax;
        }
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            // IoU >= 0, so the int bit pattern is order preserving
            const float w4 = fmaxf(fmaxf(swmax[0][threadIdx.x], swmax[1][threadIdx.x]),
                                   fmaxf(swmax[2][threadIdx.x], swmax[3][threadIdx.x]));
            atomicMax(best_bits + g0 + threadIdx.x, __float_as_int(w4));
        }
    }
    if (j < nb) {
        midx[j] = bi;
        miou[j] = best;
    }
}

__global__ __launch_bounds__(256) void iou_match_pass1(const float* __restrict__ gt, const float* __restrict__ boxes,
                                                       int m, int64_t nb, int64_t* __restrict__ midx,
                                                       float* __restrict__ miou, int* __restrict__ best_bits)
{
    iou_match_pass1_body(gt, boxes, m, nb, midx, miou, best_bits);
}

__device__ __forceinline__ void iou_match_pass2_body(const float* __restrict__ gt, const float* __restrict__ boxes,
                                                     int m, int64_t nb, const float* __restrict__ miou,
                                                     const int* __restrict__ best_bits, float t0, float t1, int l0,
                                                     int l1, int l2, int n_thr, int lowq,
                                                     int8_t* __restrict__ mlabel)
{
    __shared__ float4 sg[256];
    __shared__ float sa[256];
    __shared__ float sb[256];
    const int64_t j = blockIdx.x * 256ll + threadIdx.x;
    float4 b = make_float4(0, 0, 0, 0);
    float barea = 0.f, val = 0.f;
    if (j < nb) {
        b = reinterpret_cast<const float4*>(boxes)[j];
        barea = (b.z - b.x) * (b.w - b.y);
        val = miou[j];
    }
    // Matcher: labels over [-inf,t0), [t0,t1), [t1,inf)   (n_thr == 1: [-inf,t0), [t0,inf))
    int lab;
    if (n_thr == 2) lab = val < t0 ? l0 : (val < t1 ? l1 : l2);
    else lab = val < t0 ? l0 : l1;
    if (lowq) {
        bool hit = false;
        for (int g0 = 0; g0 < m; g0 += 256) {
            __syncthreads();
            const int gi = g0 + threadIdx.x;
            if (gi < m) {
                const float4 g = reinterpret_cast<const float4*>(gt)[gi];
                sg[threadIdx.x] = g;
                sa[threadIdx.x] = (g.z - g.x) * (g.w - g.y);
                sb[threadIdx.x] = __int_as_float(best_bits[gi]);
            }
            __syncthreads();
            const int cnt = min(256, m - g0);
            for (int i = 0; i < cnt; ++i) hit |= (iou_pair(sg[i], sa[i], b, barea) == sb[i]);
        }
        if (hit) lab = 1;
    }
    if (j < nb) mlabel[j] = (int8_t)lab;
}

__global__ __launch_bounds__(256) void iou_match_pass2(const float* __restrict__ gt, const float* __restrict__ boxes,
                                                       int m, int64_t nb, const float* __restrict__ miou,
                                                       const int* __restrict__ best_bits, float t0, float t1, int l0,
                                                       int l1, int l2, int n_thr, int lowq,
                                                       int8_t* __restrict__ mlabel)
{
    iou_match_pass2_body(gt, boxes, m, nb, miou, best_bits, t0, t1, l0, l1, l2, n_thr, lowq, mlabel);
}

// Batched over images (blockIdx.y): gt rows gt_off[i] .. gt_off[i+1] against box rows box_off[i] .. box_off[i+1], or
// against the same `nb_shared` boxes for every image when box_off is null (RPN anchors; outputs at i * nb_shared).
// One launch per pass for the whole batch instead of two per image.
__global__ __launch_bounds__(256) void iou_match_pass1_batched(const float* __restrict__ gt, const int32_t* __restrict__ gt_off,
                                                               const float* __restrict__ boxes,
                                                               const int32_t* __restrict__ box_off, int64_t nb_shared,
                                                               int64_t* __restrict__ midx, float* __restrict__ miou,
                                                               int8_t* __restrict__ mlabel, int* __restrict__ best_bits,
                                                               int label_nomatch)
{
    const int img = blockIdx.y;
    const int g0 = gt_off[img], m = gt_off[img + 1] - g0;
    const int64_t b0 = box_off ? box_off[img] : 0, nb = box_off ? box_off[img + 1] - b0 : nb_shared;
    const int64_t o0 = box_off ? b0 : (int64_t)img * nb_shared;
    if ((int64_t)blockIdx.x * 256 >= nb) return;
    if (m == 0) {        // Matcher on an empty (0,N) matrix: matches 0, labels = labels[0] (A.3)
        const int64_t j = blockIdx.x * 256ll + threadIdx.x;
        if (j < nb) { midx[o0 + j] = 0; miou[o0 + j] = 0.f; mlabel[o0 + j] = (int8_t)label_nomatch; }
        return;
    }
    iou_match_pass1_body(gt + 4 * (size_t)g0, boxes + 4 * (size_t)b0, m, nb, midx + o0, miou + o0, best_bits + g0);
}

__global__ __launch_bounds__(256) void iou_match_pass2_batched(const float* __restrict__ gt, const int32_t* __restrict__ gt_off,
                                                               const float* __restrict__ boxes,
                                                               const int32_t* __restrict__ box_off, int64_t nb_shared,
                                                               const float* __restrict__ miou, const int* __restrict__ best_bits,
                                                               float t0, float t1, int l0, int l1, int l2, int n_thr, int lowq,
                                                               int8_t* __restrict__ mlabel)
{
    const int img = blockIdx.y;
    const int g0 = gt_off[img], m = gt_off[img + 1] - g0;
    const int64_t b0 = box_off ? box_off[img] : 0, nb = box_off ? box_off[img + 1] - b0 : nb_shared;
    const int64_t o0 = box_off ? b0 : (int64_t)img * nb_shared;
    if ((int64_t)blockIdx.x * 256 >= nb || m == 0) return;
    iou_match_pass2_body(gt + 4 * (size_t)g0, boxes + 4 * (size_t)b0, m, nb, miou + o0, best_bits + g0, t0, t1, l0, l1, l2,
                         n_thr, lowq, mlabel + o0);
}

// D2 subsample_labels for a batch of label vectors without host round trips.  Element j of image i
// carries a random key; the sample is the (at most) n_pos positives and n_neg negatives with the smallest keys --
// the first entries of the permutation argsort(keys[candidates]) -- written in ascending key order.  An image's
// candidates are ranked with an O(P^2) count over LDS (P <= 12 288: a few thousand proposals) by gridDim.y workgroups:
// each stages all P keys and ranks every gridDim.y-th block of 256 candidates (one workgroup per image was 1.7 ms of
// pure latency per call at 2 000 proposals).
constexpr int SAMPLE_MAXP = 12288;

__global__ __launch_bounds__(256) void sample_by_keys_kernel(const int64_t* __restrict__ cls, const float* __restrict__ keys,
                                                             const int32_t* __restrict__ off, int num_samples,
                                                             int num_pos_max, int bg_label, int kf_stride, int kb_stride,
                                                             int64_t* __restrict__ out_fg, int64_t* __restrict__ out_bg,
                                                             int32_t* __restrict__ counts)
{
    extern __shared__ float skey[];                        // P keys, then P type bytes
    __shared__ int scnt[2];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int part = blockIdx.y, parts = gridDim.y;
    const int p0 = off[img], P = off[img + 1] - p0;
    unsigned char* stype = reinterpret_cast<unsigned char*>(skey + P);
    if (tid < 2) scnt[tid] = 0;
    __syncthreads();
    int cf = 0, cb = 0;
    for (int i = tid; i < P; i += 256) {
        const int64_t c = cls[p0 + i];
        const unsigned char t = (c == bg_label) ? 2 : ((c != -1) ? 1 : 0);
        stype[i] = t;
        skey[i] = keys[p0 + i];
        cf += t == 1;
        cb += t == 2;
    }
    atomicAdd(&scnt[0], cf);
    atomicAdd(&scnt[1], cb);
    __syncthreads();
    const int n_f = min(scnt[0], num_pos_max);
    const int n_b = min(scnt[1], num_samples - n_f);
    if (tid == 0 && part == 0) { counts[2 * img] = n_f; counts[2 * img + 1] = n_b; }
    for (int i = part * 256 + tid; i < P; i += 256 * parts) {
        const unsigned char t = stype[i];
        if (t == 0) continue;
        const float k = skey[i];
        int rank = 0;
        for (int j = 0; j < P; ++j) rank += (stype[j] == t) && (skey[j] < k || (skey[j] == k && j < i));
        if (t == 1) { if (rank < n_f) out_fg[(size_t)img * kf_stride + rank] = i; }
        else if (rank < n_b) out_bg[(size_t)img * kb_stride + rank] = i;
    }
}

// RPN._subsample_labels for a batch (rpn.py:433 -> D2 subsample_labels): one workgroup per image selects the n_f positives
// and n_b negatives with the smallest keys out of R ~ 37 000 anchors and writes the relabelled vector (1 / 0 / -1).  The
// threshold key of each class is found by a 4 x 8-bit radix select on the key bits (keys are >= 0, so the bit pattern orders
// like the value): per pass one histogram of 2 x 256 bins in LDS over the candidates that still share the prefix; ties on the
// threshold key are resolved towards the lowest index by an ordered scan of the few tied candidates.
constexpr int RELABEL_THREADS = 1024;

__global__ __launch_bounds__(RELABEL_THREADS) void rpn_subsample_relabel_kernel(const int8_t* __restrict__ labels,
                                                                                const float* __restrict__ keys,
                                                                                int8_t* __restrict__ out, int64_t R,
                                                                                int num_samples, int num_pos_max, int bg_label)
{
    __shared__ int hist[2][256];
    __shared__ unsigned prefix[2];      // key bits fixed so far (high bits)
    __shared__ int want[2];             // how many candidates with the current prefix are still to be taken (1-based rank)
    __shared__ int total[2];
    __shared__ int tie_take[2];         // candidates equal to the threshold that are taken (lowest indices first)
    __shared__ int tie_seen[2];
    const int tid = threadIdx.x;
    const int8_t* lab = labels + (int64_t)blockIdx.x * R;
    const float* key = keys + (int64_t)blockIdx.x * R;
    int8_t* o = out + (int64_t)blockIdx.x * R;
    if (tid < 2) { total[tid] = 0; prefix[tid] = 0; tie_seen[tid] = 0; }
    __syncthreads();
    int c0 = 0, c1 = 0;
    for (int64_t i = tid; i < R; i += RELABEL_THREADS) {
        const int l = lab[i];
        c0 += (l != -1 && l != bg_label);
        c1 += (l == bg_label);
    }
    for (int d = 32; d; d >>= 1) { c0 += __shfl_xor(c0, d); c1 += __shfl_xor(c1, d); }
    if ((tid & 63) == 0) { atomicAdd(&total[0], c0); atomicAdd(&total[1], c1); }
    __syncthreads();
    const int n_f = min(total[0], num_pos_max);
    const int n_b = min(total[1], num_samples - n_f);
    const int n_sel[2] = {n_f, n_b};
    if (tid < 2) want[tid] = n_sel[tid];
    // radix select: after the four passes prefix[t] is the n_sel[t]-th smallest key of class t and want[t] the number of
    // candidates EQUAL to it that belong to the selection
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < 512; i += RELABEL_THREADS) (&hist[0][0])[i] = 0;
        __syncthreads();
        const unsigned p0 = prefix[0], p1 = prefix[1];
        for (int64_t i = tid; i < R; i += RELABEL_THREADS) {
            const int l = lab[i];
            if (l == -1) continue;
            const int t = l == bg_label;
            if (n_sel[t] == 0) continue;
            const unsigned kb = __float_as_uint(key[i]);
            if ((kb & himask) == (t ? p1 : p0)) atomicAdd(&hist[t][(kb >> shift) & 255], 1);
        }
        __syncthreads();
        if (tid < 2 && n_sel[tid] > 0) {
            int w = want[tid], b = 0;
            for (; b < 255; ++b) {
                const int h = hist[tid][b];
                if (w <= h) break;
                w -= h;
            }
            want[tid] = w;
            prefix[tid] |= (unsigned)b << shift;
        }
        __syncthreads();
    }
    if (tid < 2) tie_take[tid] = want[tid];
    __syncthreads();
    const unsigned th[2] = {prefix[0], prefix[1]};
    // ties: candidates whose key equals the threshold are taken in index order.  One pass in index order over blocks of
    // RELABEL_THREADS elements; the running count of tied candidates is carried in LDS.
    for (int64_t base = 0; base < R; base += RELABEL_THREADS) {
        const int64_t i = base + tid;
        int l = -1;
        unsigned kb = 0;
        if (i < R) { l = lab[i]; kb = __float_as_uint(key[i]); }
        const int t = l == bg_label;
        const bool cand = l != -1 && n_sel[t] > 0;
        const bool less = cand && kb < th[t];
        const bool tied = cand && kb == th[t];
        int8_t res = -1;
        if (less) res = t ? 0 : 1;
        // rank of a tied candidate among the tied ones of its class with a lower index
        const unsigned long long m0 = __ballot(tied && t == 0), m1 = __ballot(tied && t == 1);
        __shared__ int wave_cnt[2][RELABEL_THREADS / 64];
        const int wv = tid >> 6, ln = tid & 63;
        if (ln == 0) { wave_cnt[0][wv] = __popcll(m0); wave_cnt[1][wv] = __popcll(m1); }
        __syncthreads();
        if (tied) {
            int before = tie_seen[t] + __popcll((t ? m1 : m0) & ((1ull << ln) - 1));
            for (int w2 = 0; w2 < wv; ++w2) before += wave_cnt[t][w2];
            if (before < tie_take[t]) res = t ? 0 : 1;
        }
        if (i < R) o[i] = res;
        __syncthreads();
        if (tid < 2) {
            int a = 0;
            for (int w2 = 0; w2 < RELABEL_THREADS / 64; ++w2) a += wave_cnt[tid][w2];
            tie_seen[tid] += a;
        }
        __syncthreads();
    }
}

__global__ void fill_nomatch_kernel(int64_t* __restrict__ midx, int8_t* __restrict__ mlabel, float* __restrict__ miou,
                                    int64_t nb, int lab)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nb; i += (int64_t)gridDim.x * blockDim.x) {
        midx[i] = 0;
        mlabel[i] = (int8_t)lab;
        if (miou) miou[i] = 0.f;
    }
}

// ---------------------------------------------------------------------------------- RPN prepare
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void rpn_prepare_kernel(const float* __restrict__ decoded, const float* __restrict__ slog,
                                   const int32_t* __restrict__ sidx, const float* __restrict__ sigma,
                                   const float* __restrict__ sizes, float* __restrict__ boxes_out,
                                   float* __restrict__ keys_out, int32_t* __restrict__ counts,
                                   int32_t* __restrict__ nonfinite, int n, int64_t R, int k, float min_size)
{
    const int64_t total = (int64_t)n * k;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int img = i / k;
        const int j = i % k;
        const int32_t a = sidx[(int64_t)img * R + j];
        const float4 b = reinterpret_cast<const float4*>(decoded)[(int64_t)img * R + a];
        const float sc = slog[(int64_t)img * R + j];
        const bool fin = isfinite(b.x) && isfinite(b.y) && isfinite(b.z) && isfinite(b.w) && isfinite(sc);
        if (!fin) atomicOr(nonfinite + img, 1);
        const float H = sizes[2 * img], W = sizes[2 * img + 1];
        float4 c;
        c.x = fminf(fmaxf(b.x, 0.f), W);
        c.y = fminf(fmaxf(b.y, 0.f), H);
        c.z = fminf(fmaxf(b.z, 0.f), W);
        c.w = fminf(fmaxf(b.w, 0.f), H);
        const bool ne = ((c.z - c.x) > min_size) && ((c.w - c.y) > min_size);
        // proposal_utils.py:94: sigma rows are the first-k anchors in raster order (row j), not idx
        const float4 s4 = reinterpret_cast<const float4*>(sigma)[(int64_t)img * R + j];
        const float ssum = ((sigmoidf_(s4.x) + sigmoidf_(s4.y)) + sigmoidf_(s4.z)) + sigmoidf_(s4.w);
        const float rescore = sc * (1.f - ssum / 4.0f);
        const bool valid = fin && ne;
        reinterpret_cast<float4*>(boxes_out)[i] = c;
        // dropped entries (non-finite / empty after clipping) sort behind every kept one: the stable descending sort
        // of the keys then yields exactly the order of the reference's filtered list
        keys_out[i] = valid ? rescore : -INFINITY;
        // one atomic per wave and image: peel off the lanes of one image at a time (wave-uniform loop)
        const unsigned long long m = __ballot(valid);
        unsigned long long rem = __ballot(true);
        const int lane = threadIdx.x & 63;
        while (rem) {
            const int leader = __ffsll(rem) - 1;
            const int limg = __shfl(img, leader, 64);
            const unsigned long long grp = __ballot(img == limg);
            if (lane == leader) {
                const int c0 = __popcll(m & grp);
                if (c0) atomicAdd(counts + limg, c0);
            }
            rem &= ~grp;
        }
    }
}

// ---------------------------------------------------------------------------------- ROI inference prepare
// fast_rcnn.py:34-101 for all ROIs of a batch in one pass (one thread per ROI): decode the K mean quadruples
// (:63 drops the sigma quadruples), finite filter over the ROI's K boxes and K+1 probabilities (:67), clip to the
// ROI's image (:78), score threshold on the foreground probabilities (:85), sigma rescoring (:101).  Outputs are dense
// over (roi, class): clipped boxes, and a sort key = rescored score for candidates / -1 otherwise (scores are > 0, so
// a stable descending sort of the keys lists an image's candidates in exactly the order batched_nms visits them);
// per image the candidate count and the largest candidate coordinate (torchvision's batched_nms class offset is
// boxes + cls * (max + 1)).  Coordinates are >= 0 after clipping, so the float max is an integer max on the bits.
// roi_valid / img_invalid: the finite filter's verdict per ROI and the number of dropped ROIs per image (the reference
// indexes its outputs by position in the FILTERED list, fast_rcnn.py:96,126 -- the caller remaps when any were dropped).
__global__ void roi_infer_prepare_kernel(const float* __restrict__ deltas, const float* __restrict__ pboxes,
                                         const float* __restrict__ probs, const int32_t* __restrict__ roi_img,
                                         const float* __restrict__ sizes, float* __restrict__ boxes_out,
                                         float* __restrict__ keys_out, uint8_t* __restrict__ roi_valid,
                                         unsigned* __restrict__ img_max, int32_t* __restrict__ img_cnt,
                                         int32_t* __restrict__ img_invalid, int64_t R, int K, float wx, float wy,
                                         float ww, float wh, float clampv, float thresh)
{
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        const float4 b = reinterpret_cast<const float4*>(pboxes)[r];
        const int img = roi_img[r];
        const float H = sizes[2 * img], W = sizes[2 * img + 1];
        const float* d = deltas + r * (int64_t)(8 * K);
        const float* p = probs + r * (int64_t)(K + 1);
        bool fin = isfinite(p[K]);
        for (int j = 0; j < K; ++j) {
            const float4 o = decode_box(b, d + 8 * j, wx, wy, ww, wh, clampv);
            fin = fin && isfinite(o.x) && isfinite(o.y) && isfinite(o.z) && isfinite(o.w) && isfinite(p[j]);
            float4 c;
            c.x = fminf(fmaxf(o.x, 0.f), W);
            c.y = fminf(fmaxf(o.y, 0.f), H);
            c.z = fminf(fmaxf(o.z, 0.f), W);
            c.w = fminf(fmaxf(o.w, 0.f), H);
            reinterpret_cast<float4*>(boxes_out)[r * K + j] = c;
        }
        roi_valid[r] = (uint8_t)fin;
        if (!fin) atomicAdd(img_invalid + img, 1);
        int cnt = 0;
        float mx = 0.f;
        for (int j = 0; j < K; ++j) {
            const bool cand = fin && p[j] > thresh;
            float key = -1.f;
            if (cand) {
                const float* sg = d + 8 * j + 4;
                const float ssum = ((sigmoidf_(sg[0]) + sigmoidf_(sg[1])) + sigmoidf_(sg[2])) + sigmoidf_(sg[3]);
                key = p[j] * (1.f - ssum / 4.0f);
                const float4 c = reinterpret_cast<const float4*>(boxes_out)[r * K + j];
                mx = fmaxf(mx, fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)));
                ++cnt;
            }
            keys_out[r * K + j] = key;
        }
        if (cnt) {
            atomicAdd(img_cnt + img, cnt);
            atomicMax(img_max + img, __float_as_uint(mx));
        }
    }
}

// boxes handed to NMS, in sorted order: box + cls * (max coordinate of the image's candidates + 1), fp32 (torchvision
// batched_nms).  grid.y = image; `order` = position within the image's dense (roi, class) segment.
__global__ void roi_infer_nms_boxes_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ order,
                                           const int32_t* __restrict__ seg, const unsigned* __restrict__ img_max,
                                           int K, float* __restrict__ out)
{
    const int img = blockIdx.y;
    const int beg = seg[img], n = seg[img + 1] - beg;
    const float off1 = __uint_as_float(img_max[img]) + 1.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int dpos = order[beg + i];
        const float off = (float)(dpos % K) * off1;
        const float4 b = reinterpret_cast<const float4*>(boxes)[beg + dpos];
        reinterpret_cast<float4*>(out)[beg + i] = make_float4(b.x + off, b.y + off, b.z + off, b.w + off);
    }
}

inline unsigned grid_for(int64_t n)
{
    int64_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int ptmi_grid_anchors(const float* cell, float* out, int h, int w, int a, float stride, float offset,
                      ptmi_stream_t s)
{
    PTMI_CHECK_ARG(cell && out && h > 0 && w > 0 && a > 0, "grid_anchors: bad args");
    hipLaunchKernelGGL(grid_anchors_kernel, dim3(grid_for((int64_t)h * w * a)), dim3(256), 0, (hipStream_t)s, cell,
                       out, h, w, a, stride, offset);
    PTMI_LAUNCH_CHECK("grid_anchors");
    return 0;
}

int ptmi_apply_deltas(const float* deltas, const float* boxes, float* out, int64_t rows, int k, int dstride,
                      int64_t nb, float wx, float wy, float ww, float wh, float scale_clamp, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(deltas && boxes && out && rows >= 0 && k > 0 && nb > 0 && dstride >= 4 * k, "apply_deltas: bad args");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(apply_deltas_kernel, dim3(grid_for(rows * k)), dim3(256), 0, (hipStream_t)s, deltas, boxes, out,
                       rows, k, dstride, nb, wx, wy, ww, wh, scale_clamp);
    PTMI_LAUNCH_CHECK("apply_deltas");
    return 0;
}

int ptmi_get_deltas(const float* src, const float* tgt, float* out, int64_t rows, float wx, float wy, float ww,
                    float wh, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(src && tgt && out && rows >= 0, "get_deltas: bad args");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(get_deltas_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)s, src, tgt, out, rows, wx,
                       wy, ww, wh);
    PTMI_LAUNCH_CHECK("get_deltas");
    return 0;
}

int ptmi_get_deltas_bwd_src(const float* src, const float* tgt, const float* ddeltas, const int64_t* src_index,
                            int64_t rows, float wx, float wy, float ww, float wh, float* dsrc, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(src && tgt && ddeltas && src_index && dsrc && rows >= 0, "get_deltas_bwd_src: bad args");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(get_deltas_bwd_src_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)s, src, tgt,
                       ddeltas, src_index, rows, wx, wy, ww, wh, dsrc);
    PTMI_LAUNCH_CHECK("get_deltas_bwd_src");
    return 0;
}

int ptmi_iou_match(const float* gt, const float* boxes, int m, int64_t nb, const float* thresholds_host,
                   const int* labels_host, int n_thr, int allow_low_quality, int64_t* matched_idx,
                   int8_t* matched_label, float* matched_iou, float* ws, ptmi_stream_t s)
{
    const float* thresholds = thresholds_host;
    const int* labels = labels_host;
    PTMI_CHECK_ARG(boxes && thresholds && labels && matched_idx && matched_label && matched_iou && nb >= 0 && m >= 0,
                   "iou_match: bad args");
    PTMI_CHECK_ARG(n_thr == 1 || n_thr == 2, "iou_match: n_thr must be 1 or 2");
    if (nb == 0) return 0;
    hipStream_t st = (hipStream_t)s;
    if (m == 0) {   // Matcher on an empty (0,N) matrix: matches 0, labels = labels[0] (A.3)
        hipLaunchKernelGGL(fill_nomatch_kernel, dim3(grid_for(nb)), dim3(256), 0, st, matched_idx, matched_label,
                           matched_iou, nb, labels[0]);
        PTMI_LAUNCH_CHECK("iou_match_fill");
        return 0;
    }
    PTMI_CHECK_ARG(gt && ws, "iou_match: gt/ws missing");
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * (size_t)m, st);   // bits of +0.0f
    if (e != hipSuccess) { ptmi_set_error("iou_match: memset failed"); return -2; }
    const unsigned blocks = (unsigned)((nb + 255) / 256);
    hipLaunchKernelGGL(iou_match_pass1, dim3(blocks), dim3(256), 0, st, gt, boxes, m, nb, matched_idx, matched_iou,
                       reinterpret_cast<int*>(ws));
    PTMI_LAUNCH_CHECK("iou_match_pass1");
    const float t0 = thresholds[0], t1 = n_thr == 2 ? thresholds[1] : 0.f;
    hipLaunchKernelGGL(iou_match_pass2, dim3(blocks), dim3(256), 0, st, gt, boxes, m, nb, matched_iou,
                       reinterpret_cast<const int*>(ws), t0, t1, labels[0], labels[1], n_thr == 2 ? labels[2] : 0,
                       n_thr, allow_low_quality, matched_label);
    PTMI_LAUNCH_CHECK("iou_match_pass2");
    return 0;
}

int ptmi_iou_match_batched(const float* gt_all, const int32_t* gt_off, const float* boxes, const int32_t* box_off,
                           int nimg, int64_t max_boxes, int64_t total_gt, const float* thresholds_host,
                           const int* labels_host, int n_thr, int allow_low_quality, int64_t* matched_idx,
                           int8_t* matched_label, float* matched_iou, float* ws, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(gt_off && boxes && thresholds_host && labels_host && matched_idx && matched_label && matched_iou &&
                       nimg > 0 && max_boxes >= 0 && total_gt >= 0,
                   "iou_match_batched: bad args");
    PTMI_CHECK_ARG(n_thr == 1 || n_thr == 2, "iou_match_batched: n_thr must be 1 or 2");
    PTMI_CHECK_ARG(total_gt == 0 || (gt_all && ws), "iou_match_batched: gt/ws missing");
    if (max_boxes == 0) return 0;
    hipStream_t st = (hipStream_t)s;
    if (total_gt > 0) {
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * (size_t)total_gt, st);   // bits of +0.0f
        if (e != hipSuccess) { ptmi_set_error("iou_match_batched: memset failed"); return -2; }
    }
    const dim3 grid((unsigned)((max_boxes + 255) / 256), (unsigned)nimg);
    hipLaunchKernelGGL(iou_match_pass1_batched, grid, dim3(256), 0, st, gt_all, gt_off, boxes, box_off, max_boxes,
                       matched_idx, matched_iou, matched_label, reinterpret_cast<int*>(ws), labels_host[0]);
    PTMI_LAUNCH_CHECK("iou_match_pass1_batched");
    const float t0 = thresholds_host[0], t1 = n_thr == 2 ? thresholds_host[1] : 0.f;
    hipLaunchKernelGGL(iou_match_pass2_batched, grid, dim3(256), 0, st, gt_all, gt_off, boxes, box_off, max_boxes,
                       matched_iou, reinterpret_cast<const int*>(ws), t0, t1, labels_host[0], labels_host[1],
                       n_thr == 2 ? labels_host[2] : 0, n_thr, allow_low_quality, matched_label);
    PTMI_LAUNCH_CHECK("iou_match_pass2_batched");
    return 0;
}

int ptmi_sample_by_keys(const int64_t* cls_all, const float* keys_all, const int32_t* offsets, int nimg,
                        int64_t max_count, int num_samples, int num_pos_max, int bg_label, int64_t* out_fg,
                        int64_t* out_bg, int32_t* counts, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(cls_all && keys_all && offsets && out_fg && out_bg && counts && nimg > 0 && num_samples > 0 &&
                       num_pos_max >= 0 && num_pos_max <= num_samples,
                   "sample_by_keys: bad args");
    PTMI_CHECK_ARG(max_count >= 0 && max_count <= SAMPLE_MAXP, "sample_by_keys: %lld candidates per image exceed %d",
                   (long long)max_count, SAMPLE_MAXP);
    const int kf = num_pos_max > 0 ? num_pos_max : 1;
    const int parts = (int)std::min<int64_t>(16, std::max<int64_t>(1, (max_count + 255) / 256));
    hipLaunchKernelGGL(sample_by_keys_kernel, dim3(nimg, parts), dim3(256), (size_t)max_count * 5 + 16, (hipStream_t)s, cls_all,
                       keys_all, offsets, num_samples, num_pos_max, bg_label, kf, num_samples, out_fg, out_bg, counts);
    PTMI_LAUNCH_CHECK("sample_by_keys");
    return 0;
}

int ptmi_rpn_subsample_relabel(const int8_t* labels, const float* keys, int8_t* out, int nimg, int64_t r, int num_samples,
                               int num_pos_max, int bg_label, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(labels && keys && out && nimg > 0 && r > 0 && num_samples > 0 && num_pos_max >= 0 &&
                       num_pos_max <= num_samples,
                   "rpn_subsample_relabel: bad args");
    hipLaunchKernelGGL(rpn_subsample_relabel_kernel, dim3(nimg), dim3(RELABEL_THREADS), 0, (hipStream_t)s, labels, keys, out,
                       r, num_samples, num_pos_max, bg_label);
    PTMI_LAUNCH_CHECK("rpn_subsample_relabel");
    return 0;
}

int ptmi_rpn_prepare(const float* decoded, const float* sorted_logits, const int32_t* sorted_idx,
                     const float* sigma_logits, const float* image_sizes_hw, float* boxes_out, float* keys_out,
                     int32_t* counts_out, int32_t* nonfinite_out, int n, int64_t r, int k, float min_size,
                     ptmi_stream_t s)
{
    PTMI_CHECK_ARG(decoded && sorted_logits && sorted_idx && sigma_logits && image_sizes_hw && boxes_out &&
                       keys_out && counts_out && nonfinite_out && n > 0 && r > 0 && k > 0 && k <= r,
                   "rpn_prepare: bad args");
    hipStream_t st = (hipStream_t)s;
    hipError_t e = hipMemsetAsync(nonfinite_out, 0, sizeof(int32_t) * (size_t)n, st);
    if (e == hipSuccess) e = hipMemsetAsync(counts_out, 0, sizeof(int32_t) * (size_t)n, st);
    if (e != hipSuccess) { ptmi_set_error("rpn_prepare: memset failed"); return -2; }
    hipLaunchKernelGGL(rpn_prepare_kernel, dim3(grid_for((int64_t)n * k)), dim3(256), 0, st, decoded, sorted_logits,
                       sorted_idx, sigma_logits, image_sizes_hw, boxes_out, keys_out, counts_out, nonfinite_out, n, r,
                       k, min_size);
    PTMI_LAUNCH_CHECK("rpn_prepare");
    return 0;
}

int ptmi_roi_infer_prepare(const float* deltas, const float* proposal_boxes, const float* probs, const int32_t* roi_img,
                           const float* image_sizes_hw, float* boxes_out, float* keys_out, uint8_t* roi_valid_out,
                           float* img_max_out, int32_t* img_count_out, int32_t* img_invalid_out, int64_t r, int k,
                           int nimg, float wx, float wy, float ww, float wh, float scale_clamp, float score_thresh,
                           ptmi_stream_t s)
{
    PTMI_CHECK_ARG(nimg > 0 && k > 0 && r >= 0 && img_max_out && img_count_out && img_invalid_out,
                   "roi_infer_prepare: bad args");
    hipStream_t st = (hipStream_t)s;
    hipError_t e = hipMemsetAsync(img_max_out, 0, sizeof(float) * (size_t)nimg, st);
    if (e == hipSuccess) e = hipMemsetAsync(img_count_out, 0, sizeof(int32_t) * (size_t)nimg, st);
    if (e == hipSuccess) e = hipMemsetAsync(img_invalid_out, 0, sizeof(int32_t) * (size_t)nimg, st);
    if (e != hipSuccess) { ptmi_set_error("roi_infer_prepare: memset failed"); return -2; }
    if (r == 0) return 0;
    PTMI_CHECK_ARG(deltas && proposal_boxes && probs && roi_img && image_sizes_hw && boxes_out && keys_out && roi_valid_out,
                   "roi_infer_prepare: null buffer");
    hipLaunchKernelGGL(roi_infer_prepare_kernel, dim3(grid_for(r)), dim3(256), 0, st, deltas, proposal_boxes, probs,
                       roi_img, image_sizes_hw, boxes_out, keys_out, roi_valid_out,
                       reinterpret_cast<unsigned*>(img_max_out), img_count_out, img_invalid_out, r, k, wx, wy, ww, wh,
                       scale_clamp, score_thresh);
    PTMI_LAUNCH_CHECK("roi_infer_prepare");
    return 0;
}

int ptmi_roi_infer_nms_boxes(const float* boxes, const int32_t* order, const int32_t* seg_offsets, const float* img_max,
                             int nimg, int max_count, int k, float* out, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(seg_offsets && img_max && nimg > 0 && nimg < 65536 && max_count >= 0 && k > 0,
                   "roi_infer_nms_boxes: bad args");
    if (max_count == 0) return 0;
    PTMI_CHECK_ARG(boxes && order && out, "roi_infer_nms_boxes: null buffer");
    hipLaunchKernelGGL(roi_infer_nms_boxes_kernel, dim3(cdiv(max_count, 256), nimg), dim3(256), 0, (hipStream_t)s, boxes,
                       order, seg_offsets, reinterpret_cast<const unsigned*>(img_max), k, out);
    PTMI_LAUNCH_CHECK("roi_infer_nms_boxes");
    return 0;
}

}  // extern "C"
