/*
 * oracle/csrc/ref_ops.c -- TEST INFRASTRUCTURE (CPU oracle), plain C, scalar.
 *
 * Restates the two torchvision==0.8.2 C++ operators the reference reaches through
 * detectron2==0.5 (neither source is vendored under /root/reference => parity
 * unpinned for these; spec = SURVEY.md Appendix A.8 / A.9):
 *
 *   - nms            <- detectron2 batched_nms -> torchvision.ops.nms, called at
 *                       /root/reference/pt/modeling/proposal_generator/proposal_utils.py:140
 *                       and /root/reference/pt/modeling/roi_heads/fast_rcnn.py:104
 *   - roi_align fwd/bwd (aligned=True, sampling_ratio=0) <- detectron2 ROIPooler
 *                       built at /root/reference/pt/modeling/roi_heads/roi_heads.py:68-73,
 *                       called at roi_heads.py:126
 *
 * Compile:  gcc -O2 -ffp-contract=off -shared -fPIC ref_ops.c -o libptoracle.so -lm
 * (-ffp-contract=off: IoU comparisons must not be perturbed by FMA contraction.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Greedy NMS.  `order` = indices sorted by descending score (computed by the
 * caller so that the tie-break policy lives in one place).  A later box j is
 * suppressed by a kept box i iff inter/(area_i+area_j-inter) > thr (strict).
 * Returns number kept; keep[] receives original indices in descending-score order. */
int64_t ptref_nms(const float* boxes, const int64_t* order, int64_t n, float thr,
                  int64_t* keep)
{
    if (n <= 0) return 0;
    uint8_t* sup = (uint8_t*)calloc((size_t)n, 1);
    float* area = (float*)malloc(sizeof(float) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const float* b = boxes + 4 * i;
        area[i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    int64_t nk = 0;
    for (int64_t a = 0; a < n; ++a) {
        int64_t i = order[a];
        if (sup[i]) continue;
        keep[nk++] = i;
        const float* bi = boxes + 4 * i;
        for (int64_t c = a + 1; c < n; ++c) {
            int64_t j = order[c];
            if (sup[j]) continue;
            const float* bj = boxes + 4 * j;
            float xx1 = bi[0] > bj[0] ? bi[0] : bj[0];
            float yy1 = bi[1] > bj[1] ? bi[1] : bj[1];
            float xx2 = bi[2] < bj[2] ? bi[2] : bj[2];
            float yy2 = bi[3] < bj[3] ? bi[3] : bj[3];
            float w = xx2 - xx1; if (w < 0.f) w = 0.f;
            float h = yy2 - yy1; if (h < 0.f) h = 0.f;
            float inter = w * h;
            float iou = inter / (area[i] + area[j] - inter);
            if (iou > thr) sup[j] = 1;
        }
    }
    free(sup); free(area);
    return nk;
}

/* bilinear tap set for one sample point; returns 0 if the sample is outside
 * [-1,H] x [-1,W] (contributes zero). */
static int bilin(float y, float x, int H, int W, int* yl, int* xl, int* yh, int* xh,
                 float* w1, float* w2, float* w3, float* w4)
{
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
    *yl = y_low; *xl = x_low; *yh = y_high; *xh = x_high;
    *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx;
    return 1;
}

/* feat (N,C,H,W) fp32; rois (R,5) = [batch_idx,x1,y1,x2,y2]; out (R,C,P,P).
 * aligned=True, sampling_ratio=0 (adaptive grid ceil(roi/P)). */
void ptref_roi_align_fwd(const float* feat, const float* rois, float* out, int N, int C,
                         int H, int W, int R, int P, float scale)
{
    (void)N;
    for (int r = 0; r < R; ++r) {
        const float* roi = rois + 5 * r;
        int b = (int)roi[0];
        float sw = roi[1] * scale - 0.5f, sh = roi[2] * scale - 0.5f;
        float ew = roi[3] * scale - 0.5f, eh = roi[4] * scale - 0.5f;
        float rw = ew - sw, rh = eh - sh;
        float bh = rh / (float)P, bw = rw / (float)P;
        int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
        float count = (float)(gh * gw > 1 ? gh * gw : 1);
        for (int c = 0; c < C; ++c) {
            const float* f = feat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < P; ++ph)
                for (int pw = 0; pw < P; ++pw) {
                    float acc = 0.f;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = sh + ph * bh + ((float)iy + .5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = sw + pw * bw + ((float)ix + .5f) * bw / (float)gw;
                            int yl, xl, yh, xh; float w1, w2, w3, w4;
                            if (!bilin(y, x, H, W, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4)) continue;
                            acc += w1 * f[yl * W + xl] + w2 * f[yl * W + xh] +
                                   w3 * f[yh * W + xl] + w4 * f[yh * W + xh];
                        }
                    }
                    out[(((size_t)r * C + c) * P + ph) * P + pw] = acc / count;
                }
        }
    }
}

/* gfeat must be zero-initialised by the caller. */
void ptref_roi_align_bwd(const float* gout, const float* rois, float* gfeat, int N, int C,
                         int H, int W, int R, int P, float scale)
{
    (void)N;
    for (int r = 0; r < R; ++r) {
        const float* roi = rois + 5 * r;
        int b = (int)roi[0];
        float sw = roi[1] * scale - 0.5f, sh = roi[2] * scale - 0.5f;
        float ew = roi[3] * scale - 0.5f, eh = roi[4] * scale - 0.5f;
        float rw = ew - sw, rh = eh - sh;
        float bh = rh / (float)P, bw = rw / (float)P;
        int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
        float count = (float)(gh * gw > 1 ? gh * gw : 1);
        for (int c = 0; c < C; ++c) {
            float* g = gfeat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < P; ++ph)
                for (int pw = 0; pw < P; ++pw) {
                    float go = gout[(((size_t)r * C + c) * P + ph) * P + pw];
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = sh + ph * bh + ((float)iy + .5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = sw + pw * bw + ((float)ix + .5f) * bw / (float)gw;
                            int yl, xl, yh, xh; float w1, w2, w3, w4;
                            if (!bilin(y, x, H, W, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4)) continue;
                            g[yl * W + xl] += go * w1 / count; g[yl * W + xh] += go * w2 / count;
                            g[yh * W + xl] += go * w3 / count; g[yh * W + xh] += go * w4 / count;
                        }
                    }
                }
        }
    }
}

/* ATen CPU upsample_bilinear2d (align_corners=False), float NCHW input, as the reference reaches it through
 * F.interpolate at /root/reference/pt/engine/trainer.py:573-576, followed by the float -> uint8 truncation of the
 * assignment into the uint8 canvas (:573).  ATen's source is not under /root/reference; this restates the evaluation
 * order of its generic kernel (upsample_generic_Nd_kernel_impl, the path taken with more than one intra-op thread) as
 * built with FMA contraction, established empirically against torch 2.10 (tools/exp/aten_bilinear_order.py) and
 * pinned by tests/golden/trainer_pieces.npz (outputs of the real PTrainer.resize) in tests/test_oracle_golden.py:
 *     src = max(fma(scale, dst + 0.5, -0.5), 0);  i0 = min(floor(src), in-1);  l1 = clamp(src - i0, 0, 1);  l0 = 1 - l1
 *     t_r = fma(p[r][x0], lx0, p[r][x1] * lx1);   out = fma(t_y0, ly0, t_y1 * ly1)
 * fmaf() is explicit (this file is compiled with -ffp-contract=off).  out is (3, dh, dw) uint8. */
static void ptref_src_index(float scale, int dst, int in_size, int* i0, int* i1, float* l0, float* l1)
{
    float src = fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (src < 0.f) src = 0.f;
    int i = (int)floorf(src);
    if (i > in_size - 1) i = in_size - 1;
    float l = src - (float)i;
    l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
    *i0 = i; *i1 = i + (i < in_size - 1 ? 1 : 0); *l1 = l; *l0 = 1.f - l;
}

void ptref_bilinear_shrink_u8(const uint8_t* img, uint8_t* out, int c, int h, int w, int dh, int dw)
{
    const float sy = (float)h / (float)dh, sx = (float)w / (float)dw;
    for (int ch = 0; ch < c; ++ch) {
        const uint8_t* p = img + (int64_t)ch * h * w;
        for (int oy = 0; oy < dh; ++oy) {
            int y0, y1; float ly0, ly1;
            if (dh == h) { y0 = y1 = oy; ly0 = 1.f; ly1 = 0.f; } else ptref_src_index(sy, oy, h, &y0, &y1, &ly0, &ly1);
            for (int ox = 0; ox < dw; ++ox) {
                int x0, x1; float lx0, lx1;
                if (dw == w) { x0 = x1 = ox; lx0 = 1.f; lx1 = 0.f; } else ptref_src_index(sx, ox, w, &x0, &x1, &lx0, &lx1);
                const float t0 = fmaf((float)p[(int64_t)y0 * w + x0], lx0, (float)p[(int64_t)y0 * w + x1] * lx1);
                const float t1 = fmaf((float)p[(int64_t)y1 * w + x0], lx0, (float)p[(int64_t)y1 * w + x1] * lx1);
                const float r = fmaf(t0, ly0, t1 * ly1);
                out[((int64_t)ch * dh + oy) * dw + ox] = (uint8_t)r;
            }
        }
    }
}
