// roi_align.hip -- ROIAlign (aligned=True, sampling_ratio=0 adaptive grid) fwd/bwd for gfx950.
//
// Replaces torchvision roi_align, which the reference reaches through detectron2's ROIPooler built at
// pt/modeling/roi_heads/roi_heads.py:68-73 and called at :126.  Gather-bound: the
// 50x83x512 fp32 feature map of an image (8.5 MB) lives in L2/MALL, one workgroup per ROI walks all
// channels with threads laid out along (c, ph, pw) so that the (R, C, 7, 7) output is written with
// fully coalesced stores.  Backward scatters with hardware fp32 atomics (-munsafe-fp-atomics).
#include "common.h"

namespace {

struct RoiGeom {
    int b;
    float sw, sh, bw, bh;
    int gw, gh;
    float count;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ roi, float scale, int P)
{
    RoiGeom g;
    g.b = (int)roi[0];
    g.sw = roi[1] * scale - 0.5f;
    g.sh = roi[2] * scale - 0.5f;
    const float ew = roi[3] * scale - 0.5f, eh = roi[4] * scale - 0.5f;
    const float rw = ew - g.sw, rh = eh - g.sh;
    g.bw = rw / (float)P;
    g.bh = rh / (float)P;
    g.gw = (int)ceilf(rw / (float)P);
    g.gh = (int)ceilf(rh / (float)P);
    const int c = g.gh * g.gw;
    g.count = (float)(c > 1 ? c : 1);
    return g;
}

__device__ __forceinline__ bool bilin(float y, float x, int H, int W, int& yl, int& xl, int& yh, int& xh, float& w1,
                                      float& w2, float& w3, float& w4)
{
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return false;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    yl = (int)y;
    xl = (int)x;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
    w1 = hy * hx; w2 = hy * lx; w3 = ly * hx; w4 = ly * lx;
    return true;
}

__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const float* __restrict__ feat,
                                                            const float* __restrict__ rois, float* __restrict__ out,
                                                            int C, int H, int W, int P, float scale)
{
    const int r = blockIdx.x;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)r, scale, P);
    const int PP = P * P, total = C * PP;
    const float* fb = feat + (size_t)g.b * C * H * W;
    float* ob = out + (size_t)r * total;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int c = i / PP, rem = i - c * PP;
        const int ph = rem / P, pw = rem - ph * P;
        const float* f = fb + (size_t)c * H * W;
        float acc = 0.f;
        for (int iy = 0; iy < g.gh; ++iy) {
            const float y = g.sh + (float)ph * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
                const float x = g.sw + (float)pw * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
                int yl, xl, yh, xh;
                float w1, w2, w3, w4;
                if (!bilin(y, x, H, W, yl, xl, yh, xh, w1, w2, w3, w4)) continue;
                acc += w1 * f[yl * W + xl] + w2 * f[yl * W + xh] + w3 * f[yh * W + xl] + w4 * f[yh * W + xh];
            }
        }
        ob[i] = acc / g.count;
    }
}

__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const float* __restrict__ dout,
                                                            const float* __restrict__ rois, float* __restrict__ dfeat,
                                                            int C, int H, int W, int P, float scale)
{
    const int r = blockIdx.x;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)r, scale, P);
    const int PP = P * P, total = C * PP;
    float* fb = dfeat + (size_t)g.b * C * H * W;
    const float* ob = dout + (size_t)r * total;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int c = i / PP, rem = i - c * PP;
        const int ph = rem / P, pw = rem - ph * P;
        float* f = fb + (size_t)c * H * W;
        const float go = ob[i];
        for (int iy = 0; iy < g.gh; ++iy) {
            const float y = g.sh + (float)ph * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
                const float x = g.sw + (float)pw * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
                int yl, xl, yh, xh;
                float w1, w2, w3, w4;
                if (!bilin(y, x, H, W, yl, xl, yh, xh, w1, w2, w3, w4)) continue;
                atomicAdd(f + yl * W + xl, go * w1 / g.count);
                atomicAdd(f + yl * W + xh, go * w2 / g.count);
                atomicAdd(f + yh * W + xl, go * w3 / g.count);
                atomicAdd(f + yh * W + xh, go * w4 / g.count);
            }
        }
    }
}

// Forward for rois grouped by image.  The samples of a bin are separable: a bin's value is
//     (1 / count) sum_{y, x} Wy_ph[y] F[y][x] Wx_pw[x]
// over the (at most gh + 1) x (gw + 1) feature cells its gh x gw samples touch, with Wy_ph[y] = the summed y-weights of the bin's
// sample rows on feature row y (same for x) -- about half of the 4 gh gw tap reads of the sample-by-sample form (12 instead of 24
// for the step's average 2 x 3 samples per bin).  roi_bin_tables_kernel writes, once per ROI, the first cell and the weights of
// every bin row / bin column (the x weights carry the 1 / count); the gather kernel keeps CG = 4 channel planes of one image in
// LDS (read from HBM / L2 once, coalesced), runs 20 ROIs at a time on 980 of its 1024 threads (thread = (roi slot, bin); it
// walks the four planes itself, so a ROI's table entries are fetched once per FOUR outputs) and needs no barrier after the
// planes are loaded.  Not the torchvision summation order: equal to the plain gather kernel to fp32 rounding, not bit for bit.
struct RoiBinHeader { int sy, sx, pad0, pad1; int ylo[8]; int xlo[8]; };      // then wy[7][TS], wx[7][TS] (floats)

__global__ __launch_bounds__(64) void roi_bin_tables_kernel(const float* __restrict__ rois, void* __restrict__ ws, int H, int W,
                                                            float scale, int TS)
{
    extern __shared__ float tw[];                    // [14][TS]
    __shared__ int slo[14], sn[14];
    const int r = blockIdx.x, t = threadIdx.x;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)r, scale, 7);
    for (int i = t; i < 14 * TS; i += 64) tw[i] = 0.f;
    __syncthreads();
    if (t < 14) {
        const bool isx = t >= 7;
        const int pb = isx ? t - 7 : t;
        const int gn = isx ? g.gw : g.gh, L = isx ? W : H;
        const float start = isx ? g.sw : g.sh, bsz = isx ? g.bw : g.bh;
        float* wv = tw + t * TS;
        int lo0 = -1, n = 0;
        for (int i = 0; i < gn; ++i) {
            float v = start + (float)pb * bsz + ((float)i + .5f) * bsz / (float)gn;
            if (v < -1.0f || v > (float)L) continue;
            if (v <= 0.f) v = 0.f;
            int l = (int)v, h2;
            if (l >= L - 1) { h2 = l = L - 1; v = (float)l; } else h2 = l + 1;
            const float lw = v - (float)l, hw = 1.f - lw;
            if (lo0 < 0) lo0 = l;                    // samples ascend: the first valid one has the lowest cell
            if (h2 - lo0 < TS) {
                wv[l - lo0] += hw;
                wv[h2 - lo0] += lw;
                n = h2 - lo0 + 1;
            } else n = -(1 << 20);                   // a box many times the map: a bin spans more cells than the table holds
        }
        slo[t] = lo0 < 0 ? 0 : lo0;
        sn[t] = n;
    }
    __syncthreads();
    char* base = (char*)ws + (size_t)r * (sizeof(RoiBinHeader) + 14 * (size_t)TS * sizeof(float));
    if (t == 0) {
        RoiBinHeader hd;
        hd.sy = hd.sx = 0; hd.pad0 = hd.pad1 = 0;
        bool fits = true;
        for (int i = 0; i < 7; ++i) {
            fits = fits && sn[i] >= 0 && sn[7 + i] >= 0;
            hd.sy = max(hd.sy, sn[i]); hd.sx = max(hd.sx, sn[7 + i]); hd.ylo[i] = slo[i]; hd.xlo[i] = slo[7 + i];
        }
        if (!fits) hd.sy = hd.sx = -1;               // the gather kernel walks this ROI sample by sample
        hd.ylo[7] = hd.xlo[7] = 0;
        *reinterpret_cast<RoiBinHeader*>(base) = hd;
    }
    float* wt = reinterpret_cast<float*>(base + sizeof(RoiBinHeader));
    const float inv_count = 1.f / g.count;
    for (int i = t; i < 14 * TS; i += 64) wt[i] = i < 7 * TS ? tw[i] : tw[i] * inv_count;
}

constexpr int RF2_THREADS = 1024, RF2_SLOTS = 20, RF2_CG = 4;
constexpr int RF2_CG8 = 8;   // channel planes per workgroup where they fit (variants: tools/exp/roi_variants.sh edits a COPY of this file)

// PACK: instead of the fp32 (R, C, 7, 7) tensor the kernel writes the flattened ROI features as the bf16 "P8 matrix" operands of the
// box head's first Linear layer under SOLVER.AMP.ENABLED (csrc/p8gemm.hip): xk[k / 8][R][8] (k = c * 49 + bin: the forward GEMM's
// operand) and, if xt != null, xt[r / 8][C * 49][8] (the weight gradient's operand, contraction over ROIs) -- the values a pack of
// the fp32 tensor would hold (round to nearest even), without the fp32 tensor and the two pack passes over it.
template <bool PACK, int NCH>
__global__ __launch_bounds__(RF2_THREADS) void roi_align_fwd_bin_kernel(const float* __restrict__ feat,
                                                                        const void* __restrict__ ws,
                                                                        const int32_t* __restrict__ img_off,
                                                                        float* __restrict__ out, int C, int H, int W,
                                                                        int CG, int TS, const float* __restrict__ rois, float scale,
                                                                        unsigned short* __restrict__ xk, unsigned short* __restrict__ xt,
                                                                        int R)
{
    extern __shared__ float smem[];
    const int HW = H * W;
    float* plane = smem;                                                 // CG * HW
    const int n = blockIdx.y, c0 = blockIdx.x * CG;
    const int cg = min(CG, C - c0);
    const int tid = threadIdx.x;
    const float* src = feat + ((size_t)n * C + c0) * HW;
    for (int i = tid; i < cg * HW; i += RF2_THREADS) plane[i] = src[i];
    __syncthreads();
    const int slot = tid / 49, bin = tid - slot * 49;
    const int ph = bin / 7, pw = bin - ph * 7;
    if (slot >= RF2_SLOTS) return;
    const size_t rstride = sizeof(RoiBinHeader) + 14 * (size_t)TS * sizeof(float);
    const int r1 = img_off[n + 1];
    // planes past the workgroup's last channel alias plane 0 (their sums are not stored)
    int po[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) po[c] = c < cg ? c * HW : 0;
    for (int r = img_off[n] + slot; r < r1; r += RF2_SLOTS) {
        const char* base = (const char*)ws + (size_t)r * rstride;
        const RoiBinHeader* hd = reinterpret_cast<const RoiBinHeader*>(base);
        const int sy = hd->sy, sx = hd->sx, ylo = hd->ylo[ph], xlo = hd->xlo[pw];
        const float* wyp = reinterpret_cast<const float*>(base + sizeof(RoiBinHeader)) + ph * TS;
        const float* wxp = reinterpret_cast<const float*>(base + sizeof(RoiBinHeader)) + (7 + pw) * TS;
        float acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = 0.f;
        auto cell = [&](float w, int y, int x) {
            const float* f = plane + min(y, H - 1) * W + min(x, W - 1);      // (cells past the map carry zero weights)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_fmaf(w, f[po[c]], acc[c]);
        };
        if (sy < 0) {
            // no table (see roi_bin_tables_kernel): the torchvision loop over the bin's samples
            const RoiGeom g = roi_geom(rois + 5 * (size_t)r, scale, 7);
            for (int iy = 0; iy < g.gh; ++iy) {
                const float y = g.sh + (float)ph * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
                for (int ix = 0; ix < g.gw; ++ix) {
                    const float x = g.sw + (float)pw * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
                    int yl, xl, yh, xh;
                    float w1, w2, w3, w4;
                    if (!bilin(y, x, H, W, yl, xl, yh, xh, w1, w2, w3, w4)) continue;
                    cell(w1 / g.count, yl, xl);
                    cell(w2 / g.count, yl, xh);
                    cell(w3 / g.count, yh, xl);
                    cell(w4 / g.count, yh, xh);
                }
            }
        } else if (sy <= 4 && sx <= 4) {
            // common case (bins of up to 3 samples a side): the bin's weights arrive as two 16-byte loads, the 4 x 4 cell grid is
            // unrolled under per-lane predicates
            const f32x4 wy = *reinterpret_cast<const f32x4*>(wyp), wx = *reinterpret_cast<const f32x4*>(wxp);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a >= sy) break;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b >= sx) break;
                    cell(wy[a] * wx[b], ylo + a, xlo + b);
                }
            }
        } else {
            for (int a = 0; a < sy; ++a) {
                const float wya = wyp[a];
                for (int b = 0; b < sx; ++b) cell(wya * wxp[b], ylo + a, xlo + b);
            }
        }
        if constexpr (PACK) {
            const size_t KD = (size_t)C * 49;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c >= cg) break;
                const size_t k = (size_t)(c0 + c) * 49 + bin;
                const unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)acc[c]);      // round to nearest even, as the pack kernels
                xk[((k >> 3) * (size_t)R + r) * 8 + (k & 7)] = b;
                if (xt) xt[((size_t)(r >> 3) * KD + k) * 8 + (r & 7)] = b;
            }
        } else {
            float* dst = out + ((size_t)r * C + c0) * 49 + bin;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (c < cg) dst[c * 49] = acc[c];
        }
    }
}

// Backward without any atomics (7x7 pooling).  ROIAlign is separable: a sample's bilinear weight is
// wy(sample_y, cell_y) * wx(sample_x, cell_x) and its bin is (ph(sample_y), pw(sample_x)), so for one ROI and one channel
//     dF[fy][fx] += (1/count) * sum_{ph,pw} Wy[fy][ph] * dOut[ph][pw] * Wx[fx][pw]
// with Wy[fy][ph] = sum of the y-weights that the gh sample rows of bin-row ph put on feature row fy (same for Wx).  The per-ROI
// 1-D weight tables (8 floats per feature row / column, indexed by ABSOLUTE row / column and zero outside the ROI's range, so that
// the consumer's loads do not depend on the header; the Wx rows carry the 1 / count) are built ONCE per ROI by
// roi_bwd_tables_kernel into a workspace.
struct RoiBwdHeader { int y0, y1, x0, x1; float count; int pad[3]; };
__global__ __launch_bounds__(64) void roi_bwd_tables_kernel(const float* __restrict__ rois, void* __restrict__ ws,
                                                            int H, int W, float scale)
{
    extern __shared__ float tsm[];                   // Wy_abs[H][8] | Wx_abs[W][8]
    __shared__ int rng[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)r, scale, 7);
    for (int i = tid; i < (H + W) * 8; i += 64) tsm[i] = 0.f;
    if (tid < 4) rng[tid] = (tid & 1) ? -1 : (1 << 30);                    // ymin, ymax, xmin, xmax
    __syncthreads();
    if (tid < 14) {
        const bool isx = tid >= 7;
        const int pb = isx ? tid - 7 : tid;
        const int gn = isx ? g.gw : g.gh, L = isx ? W : H;
        const float start = isx ? g.sw : g.sh, bsz = isx ? g.bw : g.bh;
        float* Wt = isx ? tsm + H * 8 : tsm;
        int lo = 1 << 30, hi = -1;
        for (int i = 0; i < gn; ++i) {
            float v = start + (float)pb * bsz + ((float)i + .5f) * bsz / (float)gn;
            if (v < -1.0f || v > (float)L) continue;
            if (v <= 0.f) v = 0.f;
            int l = (int)v, h2;
            if (l >= L - 1) { h2 = l = L - 1; v = (float)l; } else h2 = l + 1;
            const float lw = v - (float)l, hw = 1.f - lw;
            Wt[l * 8 + pb] += hw;
            Wt[h2 * 8 + pb] += lw;
            lo = min(lo, l);
            hi = max(hi, h2);
        }
        if (hi >= 0) {
            atomicMin(&rng[isx ? 2 : 0], lo);
            atomicMax(&rng[isx ? 3 : 1], hi);
        }
    }
    __syncthreads();
    const size_t stride = sizeof(RoiBwdHeader) + (size_t)(H + W) * 8 * sizeof(float);
    char* base = (char*)ws + (size_t)r * stride;
    const int y0 = rng[0], y1 = rng[1], x0 = rng[2], x1 = rng[3];
    if (tid == 0) {
        RoiBwdHeader hd;
        hd.y0 = y0; hd.y1 = (x1 >= 0) ? y1 : -1; hd.x0 = x0; hd.x1 = (y1 >= 0) ? x1 : -1; hd.count = g.count;
        hd.pad[0] = hd.pad[1] = hd.pad[2] = 0;
        *reinterpret_cast<RoiBwdHeader*>(base) = hd;
    }
    // rows are stored by ABSOLUTE feature row / column (zeros outside the ROI's range), so the consumer's loads do
    // not depend on the header and can be prefetched a whole ROI ahead
    // the Wx rows carry the ROI's 1 / count
    float* wt = reinterpret_cast<float*>(base + sizeof(RoiBwdHeader));
    const float inv_count = 1.f / g.count;
    for (int i = tid; i < (H + W) * 8; i += 64) wt[i] = i < H * 8 ? tsm[i] : tsm[i] * inv_count;
}

// The accumulation kernel (round 4; lanes laid out for the ROIs the path actually sees -- median 11 x 12 feature cells: its
// predecessor, a wave with one lane per map column of ONE channel, kept 10-20 of its 64 lanes busy).
//   * a workgroup owns CG <= 4 channel planes of one image in LDS (plane pitch = 16 banks mod 64: the four channels of a wave's
//     16-column window fall on disjoint bank quarters); wave w owns the ROWS [w BR, (w + 1) BR) of all planes -- fixed row
//     ownership, so consecutive ROIs never hand a cell from one wave to another (a wave's LDS operations execute in order): no
//     barrier in the ROI walk, a fixed summation order per cell (ROI order), each plane written to HBM once;
//   * lane = (channel c = lane / 16, column j = lane % 16 of a 16-column window that starts at the ROI's first column and
//     slides in steps of 16): t[ph] = sum_pw dOut[c][ph][pw] Wx[x][pw] / count once per window, then one read-modify-write
//     per owned row.  A wave whose band the ROI does not touch skips it on the (scalar) header alone;
//   * everything a ROI needs is fetched while the previous one is accumulated: headers two ROIs ahead (scalar), the 4 x 49
//     gradients, the band's rows of the Wy table and the first window's Wx rows one ahead.
constexpr int RB3_PLANES = 4;
constexpr int RB3_WAVES = 8;     // row bands = waves per workgroup (6 and 4 measured slower: tools/exp/roi_variants.sh)
constexpr int RB3_UNROLL = 2;    // rows in flight per lane

__global__ __launch_bounds__(64 * RB3_WAVES, RB3_WAVES == 6 ? 3 : 4) void roi_align_bwd_band_kernel(const float* __restrict__ dout, const void* __restrict__ ws,
                                                                 const int32_t* __restrict__ img_off, float* __restrict__ dfeat,
                                                                 int C, int H, int W, int CG, int PITCH, int BR)
{
    extern __shared__ float smem[];
    const int HW = H * W;
    float* plane = smem;                                         // CG * PITCH
    const int n = blockIdx.y, c0 = blockIdx.x * CG;
    const int cg = min(CG, C - c0);
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* wyl = plane + CG * PITCH + wave * (64 * 4 + 4 * 52);  // this wave's strip: Wy rows of its band (64 x 16 B) | 4 x 52 gradients
    float* gl = wyl + 64 * 4;
    for (int i = tid; i < CG * PITCH; i += nthreads) plane[i] = 0.f;
    __syncthreads();
    const int b0 = wave * BR, b1 = min(H, b0 + BR);              // owned rows [b0, b1)
    const int c = lane >> 4, j = lane & 15;
    const size_t stride = sizeof(RoiBwdHeader) + (size_t)(H + W) * 8 * sizeof(float);
    const int r0 = img_off[n], r1 = img_off[n + 1];
    if (b0 < b1 && r0 < r1) {
        struct Pre { float g[4]; f32x4 wy, wxa, wxb; };
        auto header = [&](int r) { return *reinterpret_cast<const RoiBwdHeader*>((const char*)ws + (size_t)r * stride); };
        auto touches = [&](const RoiBwdHeader& hd) { return hd.y1 >= 0 && hd.x1 >= 0 && max(hd.y0, b0) <= min(hd.y1, b1 - 1); };
        auto wxrow = [&](int r, int x, f32x4& a, f32x4& b) {
            const f32x4* wxg = reinterpret_cast<const f32x4*>((const char*)ws + (size_t)r * stride + sizeof(RoiBwdHeader)) + (size_t)H * 2;
            const int xc = x < W ? x : W - 1;                    // lanes past the ROI only read
            a = wxg[xc * 2];
            b = wxg[xc * 2 + 1];
        };
        // range-checked buffer loads: lanes past the 49 cg gradients / past the band's table rows read zeros without a branch
        const unsigned gbytes = 49u * (unsigned)cg * 4u, wybytes = (unsigned)(b1 - b0) * 32u;
        auto fetch = [&](int r, const RoiBwdHeader& hd, Pre& p) {
            const char* base = (const char*)ws + (size_t)r * stride + sizeof(RoiBwdHeader);
            const __amdgpu_buffer_rsrc_t rg = ptmi_rsrc(dout + ((size_t)r * C + c0) * 49, gbytes);
            const __amdgpu_buffer_rsrc_t rwy = ptmi_rsrc(base + (size_t)b0 * 32, wybytes);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                p.g[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, (lane + 64 * q) * 4, 0, 0));
            p.wy = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rwy, lane * 16, 0, 0));
            wxrow(r, hd.x0 + j, p.wxa, p.wxb);
        };
        int gslot[4];                                            // LDS slot of gradient lane + 64 q: [channel][52]; slot 51 of channel 0 is a pad
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = lane + 64 * q;
            gslot[q] = idx < 196 ? (idx / 49) * 52 + idx % 49 : 51;
        }
        RoiBwdHeader hd = header(r0), hdn = hd, hdnn = hd;
        if (r0 + 1 < r1) hdn = header(r0 + 1);
        Pre cur, nxt;
        if (touches(hd)) fetch(r0, hd, cur);
        for (int r = r0; r < r1; ++r) {
            if (r + 2 < r1) hdnn = header(r + 2);
            const bool actn = r + 1 < r1 && touches(hdn);
            if (actn) fetch(r + 1, hdn, nxt);
            if (touches(hd)) {                                   // wave-uniform
#pragma unroll
                for (int q = 0; q < 4; ++q) gl[gslot[q]] = cur.g[q];
                reinterpret_cast<f32x4*>(wyl)[lane] = cur.wy;    // (64 entries per wave: rows past the band are zeros)
                __builtin_amdgcn_wave_barrier();                 // the strip is read by other lanes of this wave
                float gq[52];
#pragma unroll
                for (int q = 0; q < 13; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(gl + c * 52 + 4 * q);
                    gq[4 * q] = v[0]; gq[4 * q + 1] = v[1]; gq[4 * q + 2] = v[2]; gq[4 * q + 3] = v[3];
                }
                const int ya = max(hd.y0, b0), yb = min(hd.y1, b1 - 1);
                f32x4 wxa = cur.wxa, wxb = cur.wxb;
                for (int xs = hd.x0; xs <= hd.x1; xs += 16) {
                    const int x = xs + j;
                    if (xs != hd.x0) wxrow(r, x, wxa, wxb);
                    float t[7];
#pragma unroll
                    for (int ph = 0; ph < 7; ++ph) {
                        float a = gq[ph * 7 + 0] * wxa[0];
                        a = __builtin_fmaf(gq[ph * 7 + 1], wxa[1], a);
                        a = __builtin_fmaf(gq[ph * 7 + 2], wxa[2], a);
                        a = __builtin_fmaf(gq[ph * 7 + 3], wxa[3], a);
                        a = __builtin_fmaf(gq[ph * 7 + 4], wxb[0], a);
                        a = __builtin_fmaf(gq[ph * 7 + 5], wxb[1], a);
                        t[ph] = __builtin_fmaf(gq[ph * 7 + 6], wxb[2], a);
                    }
                    if (x <= hd.x1 && c < cg) {
                        float* col = plane + c * PITCH + x;
                        auto rowsum = [&](const f32x4& wa, const f32x4& wb, float acc) {
                            acc = __builtin_fmaf(wa[0], t[0], acc);
                            acc = __builtin_fmaf(wa[1], t[1], acc);
                            acc = __builtin_fmaf(wa[2], t[2], acc);
                            acc = __builtin_fmaf(wa[3], t[3], acc);
                            acc = __builtin_fmaf(wb[0], t[4], acc);
                            acc = __builtin_fmaf(wb[1], t[5], acc);
                            acc = __builtin_fmaf(wb[2], t[6], acc);
                            return acc;
                        };
                        int fy = ya;
                        for (; fy + RB3_UNROLL - 1 <= yb; fy += RB3_UNROLL) {
                            f32x4 wa[RB3_UNROLL], wb[RB3_UNROLL];
                            float cv[RB3_UNROLL];
#pragma unroll
                            for (int u = 0; u < RB3_UNROLL; ++u) {
                                wa[u] = *reinterpret_cast<const f32x4*>(wyl + (fy + u - b0) * 8);
                                wb[u] = *reinterpret_cast<const f32x4*>(wyl + (fy + u - b0) * 8 + 4);
                                cv[u] = col[(fy + u) * W];
                            }
#pragma unroll
                            for (int u = 0; u < RB3_UNROLL; ++u) col[(fy + u) * W] = rowsum(wa[u], wb[u], cv[u]);
                        }
                        for (; fy <= yb; ++fy) {
                            const f32x4 wa = *reinterpret_cast<const f32x4*>(wyl + (fy - b0) * 8);
                            const f32x4 wb = *reinterpret_cast<const f32x4*>(wyl + (fy - b0) * 8 + 4);
                            col[fy * W] = rowsum(wa, wb, col[fy * W]);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();                 // the next ROI overwrites the strip
            }
            hd = hdn; hdn = hdnn; cur = nxt;
        }
    }
    __syncthreads();
    for (int i = tid; i < cg * HW; i += nthreads) {
        const int cc = i / HW;
        dfeat[((size_t)n * C + c0 + cc) * HW + (i - cc * HW)] = plane[cc * PITCH + (i - cc * HW)];
    }
}

static int roi_tab_stride(int h, int w)
{
    // a bin's samples touch at most g + 1 cells, g = ceil(roi extent / 7) <= ceil((dim + 1) / 7) (+ 1 slack), rounded up to 16 bytes
    const int gmax = ((h > w ? h : w) + 1 + 6) / 7 + 1;
    return (gmax + 1 + 3) & ~3;
}

// channel planes per workgroup: 8 in 150 KB of LDS where they fit (one workgroup of 16 waves per CU: a thread's table loads --
// the kernel's largest cost, 2.7 KB per ROI and workgroup -- serve eight outputs), else up to 4 in 72 KB (two workgroups per CU)
static int roi_fwd_planes(int c, int h, int w, int pooled)
{
    const size_t plane_bytes = (size_t)h * w * sizeof(float);
    if (pooled != 7 || plane_bytes > 72 * 1024) return 0;
    if (c >= RF2_CG8 && RF2_CG8 * plane_bytes <= 150 * 1024) return RF2_CG8;
    int cg = (int)(72 * 1024 / plane_bytes);
    if (cg > RF2_CG) cg = RF2_CG;
    if (cg > c) cg = c;
    return cg;
}

template <bool PACK>
static int roi_fwd_grouped_launch(const float* feat, const float* rois, const int32_t* img_offsets, float* out, void* ws, int n, int c,
                                  int h, int w, int r, float scale, unsigned short* xk, unsigned short* xt, hipStream_t st)
{
    const int cg = roi_fwd_planes(c, h, w, 7);
    const int TS = roi_tab_stride(h, w);
    hipLaunchKernelGGL(roi_bin_tables_kernel, dim3(r), dim3(64), (size_t)14 * TS * sizeof(float), st, rois, ws, h, w, scale, TS);
    PTMI_LAUNCH_CHECK("roi_align_tables");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)roi_align_fwd_bin_kernel<PACK, RF2_CG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)roi_align_fwd_bin_kernel<PACK, RF2_CG8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const size_t lds = (size_t)cg * h * w * sizeof(float);
    if (cg > RF2_CG)
        hipLaunchKernelGGL((roi_align_fwd_bin_kernel<PACK, RF2_CG8>), dim3(cdiv(c, cg), n), dim3(RF2_THREADS), lds, st, feat, ws, img_offsets,
                           out, c, h, w, cg, TS, rois, scale, xk, xt, r);
    else
        hipLaunchKernelGGL((roi_align_fwd_bin_kernel<PACK, RF2_CG>), dim3(cdiv(c, cg), n), dim3(RF2_THREADS), lds, st, feat, ws, img_offsets,
                           out, c, h, w, cg, TS, rois, scale, xk, xt, r);
    PTMI_LAUNCH_CHECK("roi_align_fwd_grouped");
    return 0;
}

}  // namespace

extern "C" {

int ptmi_roi_align_fwd(const float* feat, const float* rois, float* out, int n, int c, int h, int w, int r,
                       int pooled, float scale, ptmi_stream_t s)
{
    if (r == 0) return 0;
    PTMI_CHECK_ARG(feat && out && rois && n > 0 && c > 0 && h > 0 && w > 0 && r > 0 && pooled > 0,
                   "roi_align_fwd: bad args");
    hipLaunchKernelGGL(roi_align_fwd_kernel, dim3(r), dim3(256), 0, (hipStream_t)s, feat, rois, out, c, h, w, pooled,
                       scale);
    PTMI_LAUNCH_CHECK("roi_align_fwd");
    return 0;
}

int ptmi_roi_align_bwd(const float* dout, const float* rois, float* dfeat, int n, int c, int h, int w, int r,
                       int pooled, float scale, ptmi_stream_t s)
{
    if (r == 0) return 0;
    PTMI_CHECK_ARG(dout && dfeat && rois && n > 0 && c > 0 && h > 0 && w > 0 && r > 0 && pooled > 0,
                   "roi_align_bwd: bad args");
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(r), dim3(256), 0, (hipStream_t)s, dout, rois, dfeat, c, h, w, pooled,
                       scale);
    PTMI_LAUNCH_CHECK("roi_align_bwd");
    return 0;
}

int64_t ptmi_roi_align_ws_bytes(int r, int h, int w)
{
    return (int64_t)(r > 0 ? r : 1) * (int64_t)(sizeof(RoiBinHeader) + 14 * (size_t)roi_tab_stride(h, w) * sizeof(float));
}

int ptmi_roi_align_fwd_grouped(const float* feat, const float* rois, const int32_t* img_offsets, float* out, void* ws,
                               int n, int c, int h, int w, int r, int pooled, float scale, ptmi_stream_t s)
{
    if (r == 0) return 0;
    PTMI_CHECK_ARG(feat && rois && img_offsets && out && n > 0 && c > 0 && h > 0 && w > 0 && r > 0 && pooled > 0,
                   "roi_align_fwd_grouped: bad args");
    if (!roi_fwd_planes(c, h, w, pooled) || !ws)
        return ptmi_roi_align_fwd(feat, rois, out, n, c, h, w, r, pooled, scale, s);
    return roi_fwd_grouped_launch<false>(feat, rois, img_offsets, out, ws, n, c, h, w, r, scale, nullptr, nullptr, (hipStream_t)s);
}

int ptmi_roi_align_fwd_p8m_fits(int c, int h, int w, int pooled) { return roi_fwd_planes(c, h, w, pooled) > 0 && (c * pooled * pooled) % 8 == 0; }

int ptmi_roi_align_fwd_p8m(const float* feat, const float* rois, const int32_t* img_offsets, void* xk, void* xt, void* ws, int n,
                           int c, int h, int w, int r, int pooled, float scale, ptmi_stream_t s)
{
    if (r == 0) return 0;
    PTMI_CHECK_ARG(feat && rois && img_offsets && xk && ws && n > 0 && c > 0 && h > 0 && w > 0 && r > 0, "roi_align_fwd_p8m: bad args");
    PTMI_CHECK_ARG(ptmi_roi_align_fwd_p8m_fits(c, h, w, pooled), "roi_align_fwd_p8m: shape (c=%d h=%d w=%d pooled=%d) is not served",
                   c, h, w, pooled);
    hipStream_t st = (hipStream_t)s;
    if (xt && (r & 7)) {                 // the last row octet is partly beyond R: zeros there (the GEMM contracts over whole octets)
        const size_t slab = (size_t)c * 49 * 16;
        hipError_t e = hipMemsetAsync((char*)xt + (size_t)(r >> 3) * slab, 0, slab, st);
        if (e != hipSuccess) { ptmi_set_error("roi_align_fwd_p8m: memset failed"); return -2; }
    }
    return roi_fwd_grouped_launch<true>(feat, rois, img_offsets, nullptr, ws, n, c, h, w, r, scale, (unsigned short*)xk,
                                        (unsigned short*)xt, st);
}

int64_t ptmi_roi_align_bwd_ws_bytes(int r, int h, int w)
{
    return (int64_t)(r > 0 ? r : 1) * (int64_t)(sizeof(RoiBwdHeader) + (size_t)(h + w) * 8 * sizeof(float));
}

int ptmi_roi_align_bwd_grouped(const float* dout, const float* rois, const int32_t* img_offsets, float* dfeat, void* ws,
                               int n, int c, int h, int w, int r, int pooled, float scale, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(dfeat && img_offsets && n > 0 && c > 0 && h > 0 && w > 0 && r >= 0 && pooled > 0,
                   "roi_align_bwd_grouped: bad args");
    hipStream_t st = (hipStream_t)s;
    const size_t plane_bytes = (size_t)h * w * sizeof(float);
    // the band kernel: RB3_WAVES row bands (at most 32 rows each), up to four channel planes per workgroup at a pitch of
    // 16 (mod 64) floats, two workgroups per CU if they fit, one otherwise
    const int nb = RB3_WAVES, br = cdiv(h, nb);
    const int pitch = ((h * w + 47) / 64) * 64 + 16;
    const size_t strips = (size_t)nb * (64 * 4 + 4 * 52) * sizeof(float);
    int cg = (int)((80 * 1024 - strips) / ((size_t)pitch * 4));
    if (cg < 1) cg = (int)((160 * 1024 - strips) / ((size_t)pitch * 4));
    if (cg > RB3_PLANES) cg = RB3_PLANES;
    if (cg > c) cg = c;
    const bool band_ok = pooled == 7 && ws && br <= 32 && cg >= 1;
    if (!band_ok || r == 0) {             // not the hot-path shape (or no ROI at all): zero + atomic scatter kernel
        hipError_t e = hipMemsetAsync(dfeat, 0, (size_t)n * c * plane_bytes, st);
        if (e != hipSuccess) { ptmi_set_error("roi_align_bwd_grouped: memset failed"); return -2; }
        return r == 0 ? 0 : ptmi_roi_align_bwd(dout, rois, dfeat, n, c, h, w, r, pooled, scale, s);
    }
    PTMI_CHECK_ARG(dout && rois, "roi_align_bwd_grouped: null buffer");
    hipLaunchKernelGGL(roi_bwd_tables_kernel, dim3(r), dim3(64), (size_t)(h + w) * 8 * sizeof(float), st, rois, ws, h, w,
                       scale);
    PTMI_LAUNCH_CHECK("roi_align_bwd_tables");
    static bool attr3 = false;
    if (!attr3) {
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_band_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr3 = true;
    }
    hipLaunchKernelGGL(roi_align_bwd_band_kernel, dim3(cdiv(c, cg), n), dim3(64 * nb), (size_t)cg * pitch * 4 + strips, st, dout, ws,
                       img_offsets, dfeat, c, h, w, cg, pitch, br);
    PTMI_LAUNCH_CHECK("roi_align_bwd_grouped(band)");
    return 0;
}

}  // extern "C"
