/*
 * oracle/csrc/ref_aug.c -- TEST INFRASTRUCTURE (CPU oracle), plain C.
 *
 * Strong augmentation of the reference's two-crop mapper (pt/data/detection_utils.py:38-60 `build_strong_augmentation`,
 * applied at pt/data/dataset_mapper.py:155-159 on a PIL image): torchvision ColorJitter(0.4, 0.4, 0.4, 0.1) /
 * RandomGrayscale -> PIL ImageEnhance / convert("L") / convert("HSV"); the reference's own
 * pt/data/transforms/augmentation_impl.py GaussianBlur -> PIL ImageFilter.GaussianBlur and Solarize -> ImageOps.solarize.
 * Neither torchvision==0.8.2 nor Pillow's C sources are under /root/reference; this file restates Pillow's pixel
 * arithmetic (libImaging Blend.c, Convert.c rgb2hsv/hsv2rgb/L24, BoxBlur.c) and is PINNED against the real Pillow in this
 * image (tests/test_augment_cpu.py: exhaustive over all 2^24 colours for the colour conversions, random images for blur)
 * and against outputs of the reference's own GaussianBlur / Solarize classes (tests/golden/augment.npz).
 * Images are planar uint8 (3, H, W) in the record's channel order, which PIL is told is "RGB" (dataset_mapper.py:155).
 *
 * Compile: gcc -O2 -ffp-contract=off (see oracle/build.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint8_t clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v); }

/* Convert.c: L24(rgb) = r*19595 + g*38470 + b*7471 + 0x8000, >> 16 */
static inline uint8_t luma(uint8_t r, uint8_t g, uint8_t b)
{
    return (uint8_t)(((uint32_t)r * 19595u + (uint32_t)g * 38470u + (uint32_t)b * 7471u + 0x8000u) >> 16);
}

/* Blend.c ImagingBlend(in1 = degenerate, in2 = image, alpha): float arithmetic, truncation */
static inline uint8_t blend(uint8_t in1, uint8_t in2, float alpha)
{
    if (alpha >= 0.f && alpha <= 1.0f) return (uint8_t)((int)in1 + alpha * ((int)in2 - (int)in1));
    const float t = (float)((int)in1 + alpha * ((int)in2 - (int)in1));
    if (t <= 0.0f) return 0;
    if (t >= 255.0f) return 255;
    return (uint8_t)t;
}

void ptaug_to_gray(const uint8_t* in, uint8_t* out, int64_t hw)              /* convert("L") replicated to 3 channels */
{
    for (int64_t i = 0; i < hw; ++i) {
        const uint8_t l = luma(in[i], in[hw + i], in[2 * hw + i]);
        out[i] = out[hw + i] = out[2 * hw + i] = l;
    }
}

/* ImageStat.Stat(img.convert("L")).mean[0]: sum of the grey levels / count, in double; + 0.5, int() */
int ptaug_gray_mean(const uint8_t* in, int64_t hw)
{
    uint64_t s = 0;
    for (int64_t i = 0; i < hw; ++i) s += luma(in[i], in[hw + i], in[2 * hw + i]);
    return (int)((double)s / (double)hw + 0.5);
}

void ptaug_brightness(const uint8_t* in, uint8_t* out, int64_t hw, float f)  /* ImageEnhance.Brightness: degenerate = 0 */
{
    for (int64_t i = 0; i < 3 * hw; ++i) out[i] = blend(0, in[i], f);
}

void ptaug_contrast(const uint8_t* in, uint8_t* out, int64_t hw, float f)    /* ImageEnhance.Contrast: degenerate = mean grey */
{
    const uint8_t m = (uint8_t)ptaug_gray_mean(in, hw);
    for (int64_t i = 0; i < 3 * hw; ++i) out[i] = blend(m, in[i], f);
}

void ptaug_saturation(const uint8_t* in, uint8_t* out, int64_t hw, float f)  /* ImageEnhance.Color: degenerate = grey image */
{
    for (int64_t i = 0; i < hw; ++i) {
        const uint8_t l = luma(in[i], in[hw + i], in[2 * hw + i]);
        for (int c = 0; c < 3; ++c) out[c * hw + i] = blend(l, in[c * hw + i], f);
    }
}

/* Convert.c rgb2hsv_row / hsv2rgb (after colorsys.py); the literals are doubles as in the C source */
static inline void rgb2hsv(uint8_t r, uint8_t g, uint8_t b, uint8_t* oh, uint8_t* os, uint8_t* ov)
{
    const uint8_t maxc = r > g ? (r > b ? r : b) : (g > b ? g : b);
    const uint8_t minc = r < g ? (r < b ? r : b) : (g < b ? g : b);
    *ov = maxc;
    if (minc == maxc) { *oh = 0; *os = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = ((float)(maxc - r)) / cr, gc = ((float)(maxc - g)) / cr, bc = ((float)(maxc - b)) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = 2.0 + rc - bc;
    else h = 4.0 + gc - rc;
    h = fmod((h / 6.0 + 1.0), 1.0);
    *oh = clip8((int)(h * 255.0));
    *os = clip8((int)(s * 255.0));
}

static inline void hsv2rgb(uint8_t h, uint8_t s, uint8_t v, uint8_t* r, uint8_t* g, uint8_t* b)
{
    if (s == 0) { *r = *g = *b = v; return; }
    const int i = (int)floor((float)h * 6.0 / 255.0);
    const float f = (float)h * 6.0 / 255.0 - (float)i;
    const float fs = ((float)s) / 255.0;
    const int p = (int)round((float)v * (1.0 - fs));
    const int q = (int)round((float)v * (1.0 - fs * f));
    const int t = (int)round((float)v * (1.0 - fs * (1.0 - f)));
    const uint8_t up = clip8(p), uq = clip8(q), ut = clip8(t);
    switch (i % 6) {
        case 0: *r = v; *g = ut; *b = up; break;
        case 1: *r = uq; *g = v; *b = up; break;
        case 2: *r = up; *g = v; *b = ut; break;
        case 3: *r = up; *g = uq; *b = v; break;
        case 4: *r = ut; *g = up; *b = v; break;
        default: *r = v; *g = up; *b = uq; break;
    }
}

/* torchvision adjust_hue (functional_pil): h = (h + uint8(hue_factor * 255)) mod 256 on the HSV image */
void ptaug_hue(const uint8_t* in, uint8_t* out, int64_t hw, int shift)
{
    for (int64_t i = 0; i < hw; ++i) {
        uint8_t h, s, v;
        rgb2hsv(in[i], in[hw + i], in[2 * hw + i], &h, &s, &v);
        h = (uint8_t)(h + (uint8_t)shift);
        hsv2rgb(h, s, v, &out[i], &out[hw + i], &out[2 * hw + i]);
    }
}

void ptaug_solarize(const uint8_t* in, uint8_t* out, int64_t hw, int thr)    /* ImageOps.solarize */
{
    for (int64_t i = 0; i < 3 * hw; ++i) out[i] = in[i] < thr ? in[i] : (uint8_t)(255 - in[i]);
}

/* BoxBlur.c: one horizontal pass of the "extended box blur" over a line of n pixels with stride st */
static void box_line(const uint8_t* in, uint8_t* out, int n, int st, int radius, uint32_t ww, uint32_t fw)
{
    const int last = n - 1;
    for (int x = 0; x < n; ++x) {
        uint32_t acc = 0;
        for (int d = -radius; d <= radius; ++d) {
            int k = x + d;
            k = k < 0 ? 0 : (k > last ? last : k);
            acc += in[(int64_t)k * st];
        }
        int l = x - radius - 1, r = x + radius + 1;
        l = l < 0 ? 0 : l;
        r = r > last ? last : r;
        const uint32_t bulk = acc * ww + ((uint32_t)in[(int64_t)l * st] + (uint32_t)in[(int64_t)r * st]) * fw;
        out[(int64_t)x * st] = (uint8_t)((bulk + (1u << 23)) >> 24);
    }
}

/* ImageFilter.GaussianBlur(radius) -> ImagingGaussianBlur(passes = 3) -> ImagingBoxBlur(box radius, n = 3):
 * three horizontal box passes, then three vertical ones (Pillow transposes; same arithmetic). */
void ptaug_box_weights(float radius_in, int* radius_out, uint32_t* ww_out, uint32_t* fw_out)
{
    const int passes = 3;
    const float sigma2 = radius_in * radius_in / passes;
    const float L = sqrt(12.0 * sigma2 + 1.0);
    const float l = floor((L - 1.0) / 2.0);
    const float a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    const float fr = l + a / (6 * (sigma2 - (l + 1) * (l + 1)));        /* fractional box radius */
    const int radius = (int)fr;
    const uint32_t ww = (uint32_t)((1 << 24) / (fr * 2 + 1));
    *radius_out = radius;
    *ww_out = ww;
    *fw_out = (uint32_t)(((1 << 24) - (radius * 2 + 1) * ww) / 2);
}

void ptaug_gaussian_blur(const uint8_t* in, uint8_t* out, int h, int w, float radius_in)
{
    const int passes = 3;
    int radius;
    uint32_t ww, fw;
    ptaug_box_weights(radius_in, &radius, &ww, &fw);
    const int64_t hw = (int64_t)h * w;
    uint8_t* tmp = (uint8_t*)malloc((size_t)hw);
    for (int c = 0; c < 3; ++c) {
        const uint8_t* src = in + c * hw;
        uint8_t* dst = out + c * hw;
        const uint8_t* cur = src;
        for (int p = 0; p < passes; ++p) {                      /* horizontal */
            uint8_t* o = (p % 2 == 0) ? dst : tmp;
            for (int y = 0; y < h; ++y) box_line(cur + (int64_t)y * w, o + (int64_t)y * w, w, 1, radius, ww, fw);
            cur = o;
        }
        for (int p = 0; p < passes; ++p) {                      /* vertical */
            uint8_t* o = (cur == dst) ? tmp : dst;
            for (int x = 0; x < w; ++x) box_line(cur + x, o + x, h, w, radius, ww, fw);
            cur = o;
        }
        if (cur != dst) memcpy(dst, cur, (size_t)hw);
    }
    free(tmp);
}

/* Image.resize(size, Image.BILINEAR) as detectron2's ResizeTransform.apply_image calls it for uint8 images (the
 * ResizeShortestEdge of the weak augmentation, pt/data/dataset_mapper.py:107-109): Pillow's Resample.c -- separable
 * convolution with the triangle filter whose support grows with the down-scaling factor (antialiasing), coefficients
 * normalised in double and rounded to 22-bit fixed point, horizontal pass first, uint8 intermediate, a pass is skipped when
 * its size does not change.  Pinned against the live Pillow in tests/test_augment_cpu.py. */
#define PT_PRECISION_BITS (32 - 8 - 2)

static int resample_coeffs(int inSize, int outSize, int xx, int* xmin_out, int32_t* k /* >= ksize entries */)
{
    double scale = (double)inSize / outSize, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const double center = 0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    double w[64], ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        w[x] = t < 1.0 ? 1.0 - t : 0.0;
        ww += w[x];
    }
    for (int x = 0; x < xmax; ++x) {
        double v = w[x];
        if (ww != 0.0) v /= ww;
        k[x] = v < 0 ? (int32_t)(-0.5 + v * (1 << PT_PRECISION_BITS)) : (int32_t)(0.5 + v * (1 << PT_PRECISION_BITS));
    }
    *xmin_out = xmin;
    return xmax;
}

static inline uint8_t resample_clip8(int32_t v)
{
    v >>= PT_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v);
}

/* one pass over planar (c, h, w) data: along x -> (c, h, out) or along y -> (c, out, w) */
static void resample_pass(const uint8_t* in, uint8_t* out, int c, int h, int w, int outSize, int vertical)
{
    const int inSize = vertical ? h : w;
    int32_t k[64];
    for (int o = 0; o < outSize; ++o) {
        int xmin;
        const int n = resample_coeffs(inSize, outSize, o, &xmin, k);
        for (int ch = 0; ch < c; ++ch) {
            if (!vertical) {
                for (int y = 0; y < h; ++y) {
                    int32_t ss = 1 << (PT_PRECISION_BITS - 1);
                    const uint8_t* row = in + ((int64_t)ch * h + y) * w + xmin;
                    for (int x = 0; x < n; ++x) ss += row[x] * k[x];
                    out[((int64_t)ch * h + y) * outSize + o] = resample_clip8(ss);
                }
            } else {
                for (int x = 0; x < w; ++x) {
                    int32_t ss = 1 << (PT_PRECISION_BITS - 1);
                    for (int y = 0; y < n; ++y) ss += in[((int64_t)ch * h + xmin + y) * w + x] * k[y];
                    out[((int64_t)ch * outSize + o) * w + x] = resample_clip8(ss);
                }
            }
        }
    }
}

void ptaug_resize_bilinear(const uint8_t* in, uint8_t* out, int c, int h, int w, int nh, int nw)
{
    if (nh == h && nw == w) { memcpy(out, in, (size_t)c * h * w); return; }
    if (nw != w && nh != h) {
        uint8_t* tmp = (uint8_t*)malloc((size_t)c * h * nw);
        resample_pass(in, tmp, c, h, w, nw, 0);
        resample_pass(tmp, out, c, h, nw, nh, 1);
        free(tmp);
    } else if (nw != w) {
        resample_pass(in, out, c, h, w, nw, 0);
    } else {
        resample_pass(in, out, c, h, w, nh, 1);
    }
}
