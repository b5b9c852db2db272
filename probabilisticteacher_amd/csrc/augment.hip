// augment.hip -- strong augmentation of the two-crop mapper on the device (gfx950), byte-exact with Pillow.
//
// Replaces the PIL / torchvision CPU work the reference does per image in its DataLoader workers:
// pt/data/detection_utils.py:38-60 build_strong_augmentation [RandomApply(ColorJitter(0.4, 0.4, 0.4, 0.1), p=0.8),
// RandomGrayscale(p=0.2), RandomApply(GaussianBlur([0.1, 2.0]), p=0.5), RandomApply(Solarize(0.5), p=0.2)], applied at
// pt/data/dataset_mapper.py:151-159; GaussianBlur / Solarize are pt/data/transforms/augmentation_impl.py:21-53.
// The pixel arithmetic is Pillow's (libImaging Blend.c, Convert.c L24 / rgb2hsv / hsv2rgb, BoxBlur.c), restated in
// oracle/csrc/ref_aug.c, which is pinned against the real Pillow exhaustively over all 2^24 colours.
//
// All kernels are HBM-bound byte kernels over planar uint8 (3, H, W) images (the record format of SURVEY.md 8-a0), batched
// over images through a device descriptor table: blockIdx.y = image, 8 int64 words per image (include/ptmi355.h).
// Compiled with -ffp-contract=off; float / double expression trees mirror the C sources operation for operation.
#include "common.h"

namespace {

__device__ __forceinline__ uint8_t clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v); }

__device__ __forceinline__ uint8_t luma(uint8_t r, uint8_t g, uint8_t b)
{
    return (uint8_t)(((uint32_t)r * 19595u + (uint32_t)g * 38470u + (uint32_t)b * 7471u + 0x8000u) >> 16);
}

__device__ __forceinline__ uint8_t blend(uint8_t in1, uint8_t in2, float alpha)
{
    if (alpha >= 0.f && alpha <= 1.0f) return (uint8_t)((int)in1 + alpha * ((int)in2 - (int)in1));
    const float t = (float)((int)in1 + alpha * ((int)in2 - (int)in1));
    if (t <= 0.0f) return 0;
    if (t >= 255.0f) return 255;
    return (uint8_t)t;
}

__device__ __forceinline__ void rgb2hsv(uint8_t r, uint8_t g, uint8_t b, uint8_t& oh, uint8_t& os, uint8_t& ov)
{
    const uint8_t maxc = r > g ? (r > b ? r : b) : (g > b ? g : b);
    const uint8_t minc = r < g ? (r < b ? r : b) : (g < b ? g : b);
    ov = maxc;
    if (minc == maxc) { oh = 0; os = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = ((float)(maxc - r)) / cr, gc = ((float)(maxc - g)) / cr, bc = ((float)(maxc - b)) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + rc - bc);
    else h = (float)(4.0 + gc - rc);
    h = (float)fmod((h / 6.0 + 1.0), 1.0);
    oh = clip8((int)(h * 255.0));
    os = clip8((int)(s * 255.0));
}

__device__ __forceinline__ void hsv2rgb(uint8_t h, uint8_t s, uint8_t v, uint8_t& r, uint8_t& g, uint8_t& b)
{
    if (s == 0) { r = g = b = v; return; }
    const int i = (int)floor((float)h * 6.0 / 255.0);
    const float f = (float)((float)h * 6.0 / 255.0 - (float)i);
    const float fs = (float)(((float)s) / 255.0);
    const int p = (int)round((float)v * (1.0 - fs));
    const int q = (int)round((float)v * (1.0 - fs * f));
    const int t = (int)round((float)v * (1.0 - fs * (1.0 - f)));
    const uint8_t up = clip8(p), uq = clip8(q), ut = clip8(t);
    switch (i % 6) {
        case 0: r = v; g = ut; b = up; break;
        case 1: r = uq; g = v; b = up; break;
        case 2: r = up; g = v; b = ut; break;
        case 3: r = up; g = uq; b = v; break;
        case 4: r = ut; g = up; b = v; break;
        default: r = v; g = up; b = uq; break;
    }
}

// desc words: [0] src, [1] dst, [2] h, [3] w, [4] op, [5] float parameter (bits), [6] integer parameter, [7] unused
// sum of the grey levels of every image (ImageStat of convert("L")): one 64-bit atomic per workgroup
__global__ __launch_bounds__(256) void aug_gray_sum_kernel(const int64_t* __restrict__ desc,
                                                           unsigned long long* __restrict__ sums)
{
    const int64_t* d = desc + 8 * (int64_t)blockIdx.y;
    const uint8_t* src = (const uint8_t*)d[0];
    const int64_t hw = d[2] * d[3];
    unsigned long long s = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x)
        s += luma(src[i], src[hw + i], src[2 * hw + i]);
    // integer sums: the order of the additions does not matter
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sums + blockIdx.y, part[0] + part[1] + part[2] + part[3]);
}

enum { OP_COPY = 0, OP_BRIGHTNESS = 1, OP_CONTRAST = 2, OP_SATURATION = 3, OP_HUE = 4, OP_GRAY = 5, OP_SOLARIZE = 6 };

__global__ __launch_bounds__(256) void aug_color_kernel(const int64_t* __restrict__ desc,
                                                        const unsigned long long* __restrict__ sums)
{
    const int64_t* d = desc + 8 * (int64_t)blockIdx.y;
    const uint8_t* src = (const uint8_t*)d[0];
    uint8_t* dst = (uint8_t*)d[1];
    const int64_t hw = d[2] * d[3];
    const int op = (int)d[4];
    const float f = __int_as_float((int)d[5]);
    int ip = (int)d[6];
    if (op == OP_CONTRAST)      // int(ImageStat.Stat(img.convert("L")).mean[0] + 0.5)
        ip = (int)((double)sums[blockIdx.y] / (double)hw + 0.5);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
        uint8_t r = src[i], g = src[hw + i], b = src[2 * hw + i];
        switch (op) {
            case OP_BRIGHTNESS: r = blend(0, r, f); g = blend(0, g, f); b = blend(0, b, f); break;
            case OP_CONTRAST: { const uint8_t m = (uint8_t)ip; r = blend(m, r, f); g = blend(m, g, f); b = blend(m, b, f); break; }
            case OP_SATURATION: { const uint8_t l = luma(r, g, b); r = blend(l, r, f); g = blend(l, g, f); b = blend(l, b, f); break; }
            case OP_HUE: { uint8_t h, s, v; rgb2hsv(r, g, b, h, s, v); h = (uint8_t)(h + (uint8_t)ip); hsv2rgb(h, s, v, r, g, b); break; }
            case OP_GRAY: { const uint8_t l = luma(r, g, b); r = g = b = l; break; }
            case OP_SOLARIZE: r = r < ip ? r : (uint8_t)(255 - r); g = g < ip ? g : (uint8_t)(255 - g); b = b < ip ? b : (uint8_t)(255 - b); break;
            default: break;
        }
        dst[i] = r; dst[hw + i] = g; dst[2 * hw + i] = b;
    }
}

// One pass of Pillow's extended box blur (BoxBlur.c ImagingLineBoxBlur8) along x (vertical = 0) or y (vertical = 1):
//   out[x] = (sum_{|d| <= radius} in[clamp(x + d)] * ww + (in[clamp(x - radius - 1)] + in[clamp(x + radius + 1)]) * fw + 2^23) >> 24
// desc: [4] vertical, [5] radius, [6] ww, [7] fw.  (GaussianBlur sigma in [0.1, 2] gives radius 0 or 1: <= 5 taps.)
__global__ __launch_bounds__(256) void aug_box_blur_kernel(const int64_t* __restrict__ desc)
{
    const int64_t* d = desc + 8 * (int64_t)blockIdx.y;
    const uint8_t* src = (const uint8_t*)d[0];
    uint8_t* dst = (uint8_t*)d[1];
    const int h = (int)d[2], w = (int)d[3], vertical = (int)d[4], radius = (int)d[5];
    const uint32_t ww = (uint32_t)d[6], fw = (uint32_t)d[7];
    const int64_t hw = (int64_t)h * w, total = 3 * hw;
    const int n = vertical ? h : w, st = vertical ? w : 1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        const int y = (int)((i / w) % h);
        const int pos = vertical ? y : x;
        const uint8_t* line = src + (i - (int64_t)pos * st);
        uint32_t acc = 0;
        for (int k = -radius; k <= radius; ++k) {
            int q = pos + k;
            q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
            acc += line[(int64_t)q * st];
        }
        int l = pos - radius - 1, r = pos + radius + 1;
        l = l < 0 ? 0 : l;
        r = r > n - 1 ? n - 1 : r;
        const uint32_t bulk = acc * ww + ((uint32_t)line[(int64_t)l * st] + (uint32_t)line[(int64_t)r * st]) * fw;
        dst[i] = (uint8_t)((bulk + (1u << 23)) >> 24);
    }
}

// horizontal flip (D2 RandomFlip -> HFlipTransform: image[:, ::-1]) of planar images; desc [4] = 1 to flip, 0 to copy
__global__ __launch_bounds__(256) void aug_hflip_kernel(const int64_t* __restrict__ desc)
{
    const int64_t* d = desc + 8 * (int64_t)blockIdx.y;
    const uint8_t* src = (const uint8_t*)d[0];
    uint8_t* dst = (uint8_t*)d[1];
    const int h = (int)d[2], w = (int)d[3], flip = (int)d[4];
    const int64_t total = 3ll * h * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        dst[i] = flip ? src[i - x + (w - 1 - x)] : src[i];
    }
}

// One pass of Pillow's Image.resize(..., BILINEAR) (Resample.c; D2 ResizeTransform of the weak augmentation,
// dataset_mapper.py:107-109) along x (p5 = 0: (3,h,w) -> (3,h,p4)) or y (p5 = 1: (3,h,w) -> (3,p4,w)): triangle filter whose
// support grows with the down-scaling factor, coefficients normalised in double and rounded to 22-bit fixed point, integer
// accumulation from 2^21, >> 22, saturate.  Every thread recomputes the few coefficients of its output column / row with the
// exact double expression tree of precompute_coeffs() (a table kernel would save ALU work nobody is waiting for).
constexpr int RS_PRECISION_BITS = 32 - 8 - 2;
constexpr int RS_MAX_TAPS = 32;             // ksize = 2 * ceil(scale) + 1: down-scaling factors up to 15

__global__ __launch_bounds__(256) void aug_resize_pass_kernel(const int64_t* __restrict__ desc)
{
    const int64_t* d = desc + 8 * (int64_t)blockIdx.y;
    const uint8_t* src = (const uint8_t*)d[0];
    uint8_t* dst = (uint8_t*)d[1];
    const int h = (int)d[2], w = (int)d[3], outSize = (int)d[4], vertical = (int)d[5];
    const int inSize = vertical ? h : w;
    const int oh = vertical ? outSize : h, ow = vertical ? w : outSize;
    const int64_t total = 3ll * oh * ow;
    double scale = (double)inSize / outSize, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % ow);
        const int oy = (int)((i / ow) % oh);
        const int c = (int)(i / ((int64_t)ow * oh));
        const int xx = vertical ? oy : ox;
        const double center = 0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        if (xmax > RS_MAX_TAPS) xmax = RS_MAX_TAPS;        // (rejected on the host)
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            ww += t < 1.0 ? 1.0 - t : 0.0;
        }
        const uint8_t* p = vertical ? src + ((int64_t)c * h + xmin) * w + ox : src + ((int64_t)c * h + oy) * w + xmin;
        const int64_t st = vertical ? w : 1;
        int acc = 1 << (RS_PRECISION_BITS - 1);
        for (int x = 0; x < xmax; ++x) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            double v = t < 1.0 ? 1.0 - t : 0.0;
            if (ww != 0.0) v /= ww;
            const int k = v < 0 ? (int)(-0.5 + v * (1 << RS_PRECISION_BITS)) : (int)(0.5 + v * (1 << RS_PRECISION_BITS));
            acc += (int)p[x * st] * k;
        }
        acc >>= RS_PRECISION_BITS;
        dst[i] = acc < 0 ? 0 : (acc > 255 ? 255 : (uint8_t)acc);
    }
}

inline dim3 grid_for(int n, int64_t max_elems)
{
    int64_t bx = (max_elems + 1023) / 1024;
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)n);
}

}  // namespace

extern "C" {

int ptmi_aug_gray_sum_batched(const int64_t* desc, int n, int64_t max_hw, uint64_t* sums_out, ptmi_stream_t s)
{
    if (n == 0) return 0;
    PTMI_CHECK_ARG(desc && sums_out && n > 0 && n < 65536 && max_hw > 0, "aug_gray_sum_batched: bad args");
    hipStream_t st = (hipStream_t)s;
    hipError_t e = hipMemsetAsync(sums_out, 0, sizeof(uint64_t) * (size_t)n, st);
    if (e != hipSuccess) { ptmi_set_error("aug_gray_sum_batched: memset failed"); return -2; }
    hipLaunchKernelGGL(aug_gray_sum_kernel, grid_for(n, max_hw), dim3(256), 0, st, desc,
                       reinterpret_cast<unsigned long long*>(sums_out));
    PTMI_LAUNCH_CHECK("aug_gray_sum_batched");
    return 0;
}

int ptmi_aug_color_batched(const int64_t* desc, int n, int64_t max_hw, const uint64_t* gray_sums, ptmi_stream_t s)
{
    if (n == 0) return 0;
    PTMI_CHECK_ARG(desc && n > 0 && n < 65536 && max_hw > 0, "aug_color_batched: bad args");
    hipLaunchKernelGGL(aug_color_kernel, grid_for(n, max_hw), dim3(256), 0, (hipStream_t)s, desc,
                       reinterpret_cast<const unsigned long long*>(gray_sums));
    PTMI_LAUNCH_CHECK("aug_color_batched");
    return 0;
}

int ptmi_aug_box_blur_batched(const int64_t* desc, int n, int64_t max_elems, ptmi_stream_t s)
{
    if (n == 0) return 0;
    PTMI_CHECK_ARG(desc && n > 0 && n < 65536 && max_elems > 0, "aug_box_blur_batched: bad args");
    hipLaunchKernelGGL(aug_box_blur_kernel, grid_for(n, max_elems), dim3(256), 0, (hipStream_t)s, desc);
    PTMI_LAUNCH_CHECK("aug_box_blur_batched");
    return 0;
}

int ptmi_aug_hflip_batched(const int64_t* desc, int n, int64_t max_elems, ptmi_stream_t s)
{
    if (n == 0) return 0;
    PTMI_CHECK_ARG(desc && n > 0 && n < 65536 && max_elems > 0, "aug_hflip_batched: bad args");
    hipLaunchKernelGGL(aug_hflip_kernel, grid_for(n, max_elems), dim3(256), 0, (hipStream_t)s, desc);
    PTMI_LAUNCH_CHECK("aug_hflip_batched");
    return 0;
}

int ptmi_aug_resize_pass_batched(const int64_t* desc, int n, int64_t max_out_elems, ptmi_stream_t s)
{
    if (n == 0) return 0;
    PTMI_CHECK_ARG(desc && n > 0 && n < 65536 && max_out_elems > 0, "aug_resize_pass_batched: bad args");
    hipLaunchKernelGGL(aug_resize_pass_kernel, grid_for(n, max_out_elems), dim3(256), 0, (hipStream_t)s, desc);
    PTMI_LAUNCH_CHECK("aug_resize_pass_batched");
    return 0;
}

}  // extern "C"
