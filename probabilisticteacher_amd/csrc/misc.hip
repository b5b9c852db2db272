// misc.hip -- HBM-bound elementwise / reduction operators of the train step (gfx950).
//   * 2x2 max pool fwd/bwd      (ATen MaxPool2d at pt/modeling/backbone/vgg.py:59,71)
//   * image normalise + pad     (D2 preprocess_image reached at pt/modeling/meta_arch/rcnn.py:40)
//   * shrink-and-paste resize   (pt/engine/trainer.py:557-590)
//   * EMA / grad-norm / clip+SGD on flat parameter buffers (pt/engine/trainer.py:431-449,592-603,386)
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------- max pool
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int nc, int h, int w,
                                   int oh, int ow)
{
    const int64_t total = (int64_t)nc * oh * ow;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = i % ow;
        const int64_t t = i / ow;
        const int oy = t % oh;
        const int64_t c = t / oh;
        const float* p = x + (c * h + 2 * oy) * (int64_t)w + 2 * ox;
        const float a = p[0], b = p[1], cc = p[w], d = p[w + 1];
        // ATen semantics: running max with (val > max) || isnan(val)
        float m = a;
        if (b > m || b != b) m = b;
        if (cc > m || cc != cc) m = cc;
        if (d > m || d != d) m = d;
        y[i] = m;
    }
}

// dx[pos] = dy[window] iff pos is the FIRST maximum of its window (scan order (0,0),(0,1),(1,0),(1,1)).
// One thread per 2x2 WINDOW: every x element is read once and every dx element written once (the first version ran one
// thread per input element, i.e. four reads of each window); the odd last row / column of a floor-mode pool gets zeros from
// the threads of the neighbouring window.  grid.y = plane (n * c), threads walk the (oh + 1) x (ow + 1) window grid.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int nc, int h, int w, int oh,
                                                          int ow, int relu_mask)
{
  for (int64_t plane = blockIdx.y; plane < nc; plane += gridDim.y) {
    const float* xp = x + plane * (int64_t)h * w;
    float* dp = dx + plane * (int64_t)h * w;
    const float* gp = dy + plane * (int64_t)oh * ow;
    const int gw = ow + (w & 1), gh = oh + (h & 1);          // window grid incl. the unpooled tail column / row
    const int total = gw * gh;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int oy = i / gw, ox = i - oy * gw;
        const int y0 = 2 * oy, x0 = 2 * ox;
        if (oy < oh && ox < ow) {
            const float* p = xp + (int64_t)y0 * w + x0;
            const float v0 = p[0], v1 = p[1], v2 = p[w], v3 = p[w + 1];
            int am = 0;
            float m = v0;
            if (v1 > m || v1 != v1) { m = v1; am = 1; }
            if (v2 > m || v2 != v2) { m = v2; am = 2; }
            if (v3 > m || v3 != v3) { m = v3; am = 3; }
            // relu_mask: x is a post-ReLU activation; also apply d/d(pre-activation) = (x > 0) in the same pass
            const float g = (!relu_mask || m > 0.f) ? gp[(int64_t)oy * ow + ox] : 0.f;
            float* q = dp + (int64_t)y0 * w + x0;
            q[0] = am == 0 ? g : 0.f;
            q[1] = am == 1 ? g : 0.f;
            q[w] = am == 2 ? g : 0.f;
            q[w + 1] = am == 3 ? g : 0.f;
        } else {
            // tail: elements that belong to no window
            for (int dyy = 0; dyy < 2; ++dyy)
                for (int dxx = 0; dxx < 2; ++dxx)
                    if (y0 + dyy < h && x0 + dxx < w) dp[(int64_t)(y0 + dyy) * w + x0 + dxx] = 0.f;
        }
    }
  }
}

// ---------------------------------------------------------------------------------- image prep
__device__ __forceinline__ float prep_pixel(const uint8_t* __restrict__ img, int h, int w, int c, int y, int x, float m0,
                                            float m1, float m2, float s0, float s1, float s2)
{
    if (y >= h || x >= w) return 0.f;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    return ((float)img[((int64_t)c * h + y) * w + x] - mean) / sd;
}

__global__ void preprocess_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int h, int w,
                                  int hmax, int wmax, float m0, float m1, float m2, float s0, float s1, float s2)
{
    const int64_t total = 3ll * hmax * wmax;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int x = i % wmax;
        const int64_t t = i / wmax;
        out[i] = prep_pixel(img, h, w, (int)(t / hmax), (int)(t % hmax), x, m0, m1, m2, s0, s1, s2);
    }
}

// whole batch in one launch: blockIdx.y = image; desc = 8 int64 words per image (include/ptmi355.h)
__global__ void preprocess_batched_kernel(const int64_t* __restrict__ desc, float* __restrict__ out, int hmax, int wmax,
                                          float m0, float m1, float m2, float s0, float s1, float s2)
{
    const int64_t* d = desc + 8 * (int64_t)blockIdx.y;
    const uint8_t* img = (const uint8_t*)d[0];
    const int h = (int)d[2], w = (int)d[3];
    const int64_t total = 3ll * hmax * wmax;
    float* o = out + (int64_t)blockIdx.y * total;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int x = i % wmax;
        const int64_t t = i / wmax;
        o[i] = prep_pixel(img, h, w, (int)(t / hmax), (int)(t % hmax), x, m0, m1, m2, s0, s1, s2);
    }
}

// F.interpolate(bilinear, align_corners=False) from (h,w) to (dh,dw), truncated to uint8, pasted at
// (y1,x1) on a canvas of int(pixel_mean)   (trainer.py:563-575).
//
// Byte-exact with ATen's CPU kernel as the reference runs it (float NCHW input, more than one intra-op thread:
// upsample_generic_Nd_kernel_impl, built with FMA contraction -- verified against torch 2.10 on AVX-512 for 1e6s of
// pixels, tools/exp/aten_bilinear_order.py):
//     src   = max(fma(scale, dst + 0.5, -0.5), 0)            scale = float(in) / float(out)
//     i0    = min(floor(src), in - 1);  l1 = clamp(src - i0, 0, 1);  l0 = 1 - l1;  i1 = i0 + (i0 < in - 1)
//     t_r   = fma(p[r][x0], lx0, p[r][x1] * lx1)             (row r = y0, y1)
//     out   = fma(t_y0, ly0, t_y1 * ly1)
// (-ffp-contract=off for this library: every fused operation is spelled out.)  With a single intra-op thread and
// C == 3 ATen takes a different (channels-last) kernel whose rounding differs in ~1e-3 of the bytes; the training
// process of the reference runs with the default thread count (> 1).
__device__ __forceinline__ void aten_src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1)
{
    float src = fmaf(scale, (float)dst + 0.5f, -0.5f);
    src = src < 0.f ? 0.f : src;
    i0 = (int)floorf(src);
    if (i0 > in_size - 1) i0 = in_size - 1;
    l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    l0 = 1.f - l1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
}

__device__ __forceinline__ uint8_t shrink_pixel(const uint8_t* __restrict__ img, int h, int w, int dh, int dw, int y1,
                                                int x1, int c, int y, int x, int mean_c, float sy, float sx)
{
    const int oy = y - y1, ox = x - x1;
    if (oy < 0 || oy >= dh || ox < 0 || ox >= dw) return (uint8_t)mean_c;
    int iy0, iy1, ix0, ix1;
    float ly0, ly1, lx0, lx1;
    if (dh == h) { iy0 = iy1 = oy; ly0 = 1.f; ly1 = 0.f; } else aten_src_index(sy, oy, h, iy0, iy1, ly0, ly1);
    if (dw == w) { ix0 = ix1 = ox; lx0 = 1.f; lx1 = 0.f; } else aten_src_index(sx, ox, w, ix0, ix1, lx0, lx1);
    const uint8_t* p = img + (int64_t)c * h * w;
    const float t0 = fmaf((float)p[(int64_t)iy0 * w + ix0], lx0, (float)p[(int64_t)iy0 * w + ix1] * lx1);
    const float t1 = fmaf((float)p[(int64_t)iy1 * w + ix0], lx0, (float)p[(int64_t)iy1 * w + ix1] * lx1);
    const float r = fmaf(t0, ly0, t1 * ly1);
    int v = (int)r;                        // float -> uint8 truncation on assignment into the uint8 canvas
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    return (uint8_t)v;
}

__global__ void shrink_paste_kernel(const uint8_t* __restrict__ img, uint8_t* __restrict__ out, int h, int w,
                                    int dh, int dw, int y1, int x1, int m0, int m1, int m2)
{
    const int64_t total = 3ll * h * w;
    const float sy = (float)h / (float)dh, sx = (float)w / (float)dw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int x = i % w;
        const int64_t t = i / w;
        const int c = (int)(t / h);
        out[i] = shrink_pixel(img, h, w, dh, dw, y1, x1, c, (int)(t % h), x, c == 0 ? m0 : (c == 1 ? m1 : m2), sy, sx);
    }
}

__global__ void shrink_paste_batched_kernel(const int64_t* __restrict__ desc, int m0, int m1, int m2)
{
    const int64_t* d = desc + 8 * (int64_t)blockIdx.y;
    const uint8_t* img = (const uint8_t*)d[0];
    uint8_t* out = (uint8_t*)d[1];
    const int h = (int)d[2], w = (int)d[3], dh = (int)d[4], dw = (int)d[5], y1 = (int)d[6], x1 = (int)d[7];
    const int64_t total = 3ll * h * w;
    const float sy = (float)h / (float)dh, sx = (float)w / (float)dw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int x = i % w;
        const int64_t t = i / w;
        const int c = (int)(t / h);
        out[i] = shrink_pixel(img, h, w, dh, dw, y1, x1, c, (int)(t % h), x, c == 0 ? m0 : (c == 1 ? m1 : m2), sy, sx);
    }
}

// ---------------------------------------------------------------------------------- optimiser
__global__ void ema_kernel(const float* __restrict__ s, float* __restrict__ t, int64_t n, float k, float omk)
{
    const int64_t n4 = n >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(s);
    float4* t4 = reinterpret_cast<float4*>(t);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = s4[i];
        float4 b = t4[i];
        b.x = a.x * omk + b.x * k;
        b.y = a.y * omk + b.y * k;
        b.z = a.z * omk + b.z * k;
        b.w = a.w * omk + b.w * k;
        t4[i] = b;
    }
    for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        t[i] = s[i] * omk + t[i] * k;
}

constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n,
                                                            float* __restrict__ partial)
{
    __shared__ float sm[4];
    float acc = 0.f;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = g[i];
        acc += v * v;
    }
    const float t = block_sum_256(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void sum_final_kernel(const float* __restrict__ partial, int np,
                                                        float* __restrict__ out)
{
    __shared__ float sm[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) acc += partial[i];
    const float t = block_sum_256(acc, sm);
    if (threadIdx.x == 0) out[0] = t;
}

__global__ void clip_sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                int64_t n, const float* __restrict__ sumsq, float clip, float lr, float mu,
                                float wd, int first)
{
    const float total = sqrtf(sumsq[0]);
    const float sc = clip / fmaxf(total, clip);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float gv = g[i] * sc + wd * pv;
        const float b = first ? gv : mu * buf[i] + gv;
        buf[i] = b;
        p[i] = pv - lr * b;
    }
}

__global__ void scale_by_clip_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ sumsq, float clip)
{
    const float total = sqrtf(sumsq[0]);
    const float sc = clip / fmaxf(total, clip);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        g[i] *= sc;
}

inline unsigned grid_for(int64_t n, int per = 256, int cap = 8192)
{
    int64_t b = (n + per - 1) / per;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// hold_cus: `n` workgroups that each take 64 KB of (dynamic, so the optimiser cannot drop it) LDS and spin on the 100 MHz real-time
// counter for `ticks` -- a stand-in for a co-running collective kernel: no >= 100-KB workgroup of this library (every MFMA kernel)
// fits beside one, and the 512-register persistent convolution waves fit beside nothing at all
__global__ __launch_bounds__(256) void hold_cus_kernel(int* __restrict__ out, unsigned long long ticks)
{
    extern __shared__ int hold_pad[];
    hold_pad[threadIdx.x] = (int)threadIdx.x;
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    do {
        __builtin_amdgcn_s_sleep(32);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    } while (t - t0 < ticks);
    if (ticks == 0xFFFFFFFFFFFFFFFFull) out[0] = hold_pad[(threadIdx.x + 1) & 255];     // (never: keeps the LDS accesses alive)
}

}  // namespace

extern "C" {

int ptmi_hold_cus(int32_t* scratch, int n_cus, int microseconds, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(scratch && n_cus > 0 && n_cus <= 4096 && microseconds > 0 && microseconds <= 10000000, "hold_cus: bad args");
    hipLaunchKernelGGL(hold_cus_kernel, dim3((unsigned)n_cus), dim3(256), 65536, (hipStream_t)s, (int*)scratch,
                       (unsigned long long)microseconds * 100ull);
    PTMI_LAUNCH_CHECK("hold_cus");
    return 0;
}

int ptmi_maxpool2x2_fwd(const float* x, float* y, int nc, int h, int w, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && y && nc > 0 && h >= 2 && w >= 2, "maxpool2x2_fwd: bad args");
    const int oh = h / 2, ow = w / 2;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((int64_t)nc * oh * ow)), dim3(256), 0, (hipStream_t)s, x,
                       y, nc, h, w, oh, ow);
    PTMI_LAUNCH_CHECK("maxpool2x2_fwd");
    return 0;
}

int ptmi_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int nc, int h, int w, int relu_mask,
                        ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && dy && dx && nc > 0 && h >= 2 && w >= 2, "maxpool2x2_bwd: bad args");
    const int oh = h / 2, ow = w / 2;
    const int windows = (oh + (h & 1)) * (ow + (w & 1));
    int bx = (windows + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)bx, (unsigned)(nc < 65535 ? nc : 65535)), dim3(256), 0,
                       (hipStream_t)s, x, dy, dx, nc, h, w, oh, ow, relu_mask);
    PTMI_LAUNCH_CHECK("maxpool2x2_bwd");
    return 0;
}

int ptmi_preprocess_image(const uint8_t* img, float* out, int h, int w, int hmax, int wmax, float m0, float m1,
                          float m2, float s0, float s1, float s2, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(img && out && h > 0 && w > 0 && hmax >= h && wmax >= w, "preprocess_image: bad args");
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(3ll * hmax * wmax)), dim3(256), 0, (hipStream_t)s, img, out,
                       h, w, hmax, wmax, m0, m1, m2, s0, s1, s2);
    PTMI_LAUNCH_CHECK("preprocess_image");
    return 0;
}

int ptmi_shrink_paste(const uint8_t* img, uint8_t* out, int h, int w, int dh, int dw, int y1, int x1, int m0,
                      int m1, int m2, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(img && out && h > 0 && w > 0 && dh > 0 && dw > 0 && dh <= h && dw <= w && y1 >= 0 && x1 >= 0,
                   "shrink_paste: bad args");
    hipLaunchKernelGGL(shrink_paste_kernel, dim3(grid_for(3ll * h * w)), dim3(256), 0, (hipStream_t)s, img, out, h,
                       w, dh, dw, y1, x1, m0, m1, m2);
    PTMI_LAUNCH_CHECK("shrink_paste");
    return 0;
}

int ptmi_preprocess_batched(const int64_t* desc, float* out, int n, int hmax, int wmax, float m0, float m1, float m2,
                            float s0, float s1, float s2, ptmi_stream_t s)
{
    if (n == 0) return 0;
    PTMI_CHECK_ARG(desc && out && n > 0 && n < 65536 && hmax > 0 && wmax > 0, "preprocess_batched: bad args");
    const int64_t per = 3ll * hmax * wmax;
    int bx = (int)((per + 1023) / 1024);                      // ~4 elements per thread
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(preprocess_batched_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)s, desc, out, hmax, wmax, m0,
                       m1, m2, s0, s1, s2);
    PTMI_LAUNCH_CHECK("preprocess_batched");
    return 0;
}

int ptmi_shrink_paste_batched(const int64_t* desc, int n, int64_t max_elems, int m0, int m1, int m2, ptmi_stream_t s)
{
    if (n == 0) return 0;
    PTMI_CHECK_ARG(desc && n > 0 && n < 65536 && max_elems > 0, "shrink_paste_batched: bad args");
    int bx = (int)((max_elems + 1023) / 1024);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(shrink_paste_batched_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)s, desc, m0, m1, m2);
    PTMI_LAUNCH_CHECK("shrink_paste_batched");
    return 0;
}

int ptmi_ema_update(const float* student, float* teacher, int64_t n, float keep_rate, float one_minus_keep_rate,
                    ptmi_stream_t s)
{
    if (n == 0) return 0;
    PTMI_CHECK_ARG(student && teacher && n > 0, "ema_update: bad args");
    // (1 - k) is formed by the caller in double and rounded once, exactly as python's `1 - keep_rate`
    // reaches torch at trainer.py:443; deriving it from the fp32 k here would differ in the last bits.
    const float omk = one_minus_keep_rate;
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)s, student, teacher, n,
                       keep_rate, omk);
    PTMI_LAUNCH_CHECK("ema_update");
    return 0;
}

int ptmi_sumsq(const float* g, int64_t n, float* sumsq_out, float* ws, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(g && sumsq_out && ws && n >= 0, "sumsq: bad args");
    int np = (int)((n + 255) / 256);
    if (np > SUMSQ_BLOCKS) np = SUMSQ_BLOCKS;
    if (np < 1) np = 1;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(np), dim3(256), 0, (hipStream_t)s, g, n, ws);
    PTMI_LAUNCH_CHECK("sumsq_partial");
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, ws, np, sumsq_out);
    PTMI_LAUNCH_CHECK("sumsq_final");
    return 0;
}

int ptmi_clip_sgd_step(float* p, const float* g, float* buf, int64_t n, const float* sumsq, float clip_norm,
                       float lr, float momentum, float weight_decay, int first, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(p && g && buf && sumsq && n >= 0, "clip_sgd_step: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(clip_sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, p, g, buf, n, sumsq,
                       clip_norm, lr, momentum, weight_decay, first);
    PTMI_LAUNCH_CHECK("clip_sgd_step");
    return 0;
}

int ptmi_scale_by_clip(float* g, int64_t n, const float* sumsq, float clip_norm, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(g && sumsq && n >= 0, "scale_by_clip: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_by_clip_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, g, n, sumsq, clip_norm);
    PTMI_LAUNCH_CHECK("scale_by_clip");
    return 0;
}

}  // extern "C"
