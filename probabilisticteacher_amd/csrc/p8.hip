// p8.hip -- bf16 STORAGE path of the 3x3 convolution stack (SOLVER.AMP.ENABLED, BASELINE configs[4]; reference AMP flag
// pt/engine/trainer.py:98: under autocast cuDNN reads and writes bf16 activations at pt/modeling/backbone/vgg.py:45-53,66-69 and
// pt/modeling/proposal_generator/rpn.py:96).  gfx950 only.
//
// Round 4.  The bf16-INPUT kernels of conv.hip keep fp32 activations in HBM and LDS and round between LDS and the matrix core:
// twice the bytes at every level, and 0.23 of the bf16 MFMA peak.  Here activations and activation gradients LIVE in bf16:
//
//   "P8" layout (padded, 8-channel blocks)      t[2 ceil(C/16)][ROWS][WS][8]  bf16,   ROWS = N (H + 1) + 1,  WS = W + 1
//     (whole 16-channel chunks: channels beyond C are zeros)
//     * pixel (n, r, c) sits at row n (H + 1) + 1 + r, column 1 + c; row n (H + 1) (one zero row between images, one before the
//       first, one after the last) and column 0 of every row hold ZEROS: the zero padding of a 3x3 convolution is IN the tensor
//       (the right neighbour of a row's last pixel is the next row's column 0), so a tap is the flat pixel offset
//       (ky - 1) WS + (kx - 1) and no kernel tests an image edge;
//     * the 8 channels of a pixel are one 16-byte vector: ONE ds_read_b128 is a lane's whole B operand (8 k values) of
//       v_mfma_f32_32x32x16_bf16 in the forward / dgrad kernel, and the weight gradient -- which contracts over PIXELS -- gets its
//       operands out of the same image with ds_read_b64_tr_b16 (the gfx950 transposing LDS read);
//     * every producer writes the pad positions as zeros (conv epilogue, pooling, conversion): the invariant the consumers rely on.
//
//   p8_conv3x3_kernel<MT>     forward and dgrad (flipped / transposed pack), implicit GEMM M = Cout, N = pixels, K = 9 Cin:
//                             workgroup = 4 waves = (32 MT) output channels x (64 / MT) rows x 32 columns, wave = MT x (16 / MT)
//                             accumulator tiles (256 AGPRs); K in chunks of 16 channels x 9 taps, two LDS stages filled by
//                             buffer_load ... lds (weights: one lane-linear slab per (channel tile, chunk); patch: 16-byte pixel
//                             vectors with per-lane source offsets), one barrier per chunk.
//   p8_wgrad_kernel           dW[co][ci][tap] = sum over flat pixels of dY[co][f] X[ci][f + tap offset]: (see there)
//   conversions, max-pool, ReLU backward, weight packs.
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) ptmi_bf16x8 plds_bf16x8_t;
typedef __attribute__((address_space(3))) void plds_void_t;
typedef unsigned short u16;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int P8T = 256;                     // threads per workgroup (4 waves, one per SIMD)

struct P8Dims {
    int N, H, W, HS, WS, ROWS;
    int64_t PT;                              // pixels per 8-channel plane
};
inline P8Dims p8_dims(int n, int h, int w)
{
    P8Dims d;
    d.N = n; d.H = h; d.W = w; d.HS = h + 1; d.WS = w + 1; d.ROWS = n * d.HS + 1;
    d.PT = (int64_t)d.ROWS * d.WS;
    return d;
}

// round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// ------------------------------------------------------------------------------------------------ conversions
// fp32 NCHW -> P8 (channels beyond C are zero: the 3-channel image becomes a 16-channel P8 tensor for the stem layer)
__global__ __launch_bounds__(256) void p8_from_nchw_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int C, int H, int W,
                                                           int HS, int WS, int64_t PT, int CB)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cb = blockIdx.y;
    if (f >= PT) return;
    const int row = (int)(f / WS), col = (int)(f - (int64_t)row * WS);
    const int n = row / HS, r = row - n * HS - 1, c = col - 1;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r >= 0 && c >= 0) {
        const int64_t hw = (int64_t)H * W;
        const float* p = x + ((int64_t)n * C + cb * 8) * hw + (int64_t)r * W + c;
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (cb * 8 + j < C) ? p[j * hw] : 0.f;
        v = (u32x4){pack_bf16x2(e[0], e[1]), pack_bf16x2(e[2], e[3]), pack_bf16x2(e[4], e[5]), pack_bf16x2(e[6], e[7])};
    }
    y[(int64_t)cb * PT + f] = v;
}

// P8 -> fp32 NCHW (exact: bf16 -> fp32)
__global__ __launch_bounds__(256) void p8_to_nchw_kernel(const u32x4* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W,
                                                         int HS, int WS, int64_t PT)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over N * H * W
    const int cb = blockIdx.y;
    const int64_t hw = (int64_t)H * W;
    const int n = (int)(i / hw);
    const int64_t rem = i - n * hw;
    const int r = (int)(rem / W), c = (int)(rem - (int64_t)r * W);
    if (n >= N) return;
    const u32x4 v = x[(int64_t)cb * PT + ((int64_t)n * HS + 1 + r) * WS + 1 + c];
    float* p = y + ((int64_t)n * C + cb * 8) * hw + rem;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (cb * 8 + j < C) p[j * hw] = __builtin_bit_cast(float, (j & 1) ? (v[j >> 1] & 0xFFFF0000u) : (v[j >> 1] << 16));
}

// ------------------------------------------------------------------------------------------------ max pool 2x2 (floor)
__device__ __forceinline__ float bfe(const u32x4& v, int j) { return __builtin_bit_cast(float, (j & 1) ? (v[j >> 1] & 0xFFFF0000u) : (v[j >> 1] << 16)); }

__global__ __launch_bounds__(256) void p8_maxpool_fwd_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int HSi, int WSi,
                                                             int64_t PTi, int HSo, int WSo, int64_t PTo)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;          // output flat pixel
    const int cb = blockIdx.y;
    if (f >= PTo) return;
    const int row = (int)(f / WSo), col = (int)(f - (int64_t)row * WSo);
    const int n = row / HSo, r = row - n * HSo - 1, c = col - 1;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (r >= 0 && c >= 0) {
        const u32x4* p = x + (int64_t)cb * PTi + ((int64_t)n * HSi + 1 + 2 * r) * WSi + 1 + 2 * c;
        const u32x4 a = p[0], b = p[1], d = p[WSi], e = p[WSi + 1];
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaxf(bfe(a, j), bfe(b, j)), fmaxf(bfe(d, j), bfe(e, j)));
        o = (u32x4){pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7])};
    }
    y[(int64_t)cb * PTo + f] = o;
}

// dx (input geometry) from dy (output geometry): the gradient goes to the FIRST maximum of a window in (0,0),(0,1),(1,0),(1,1)
// order (ATen); relu_mask: additionally times (x > 0) (x is a post-ReLU activation: pool backward + ReLU backward in one pass)
__global__ __launch_bounds__(256) void p8_maxpool_bwd_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ dy,
                                                             u32x4* __restrict__ dx, int H, int W, int HSi, int WSi, int64_t PTi,
                                                             int HSo, int WSo, int64_t PTo, int relu_mask)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;          // input flat pixel
    const int cb = blockIdx.y;
    if (f >= PTi) return;
    const int row = (int)(f / WSi), col = (int)(f - (int64_t)row * WSi);
    const int n = row / HSi, r = row - n * HSi - 1, c = col - 1;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (r >= 0 && c >= 0 && (r >> 1) < (H >> 1) && (c >> 1) < (W >> 1)) {
        const int wr = r >> 1, wc = c >> 1, pos = (r & 1) * 2 + (c & 1);
        const u32x4* p = x + (int64_t)cb * PTi + ((int64_t)n * HSi + 1 + 2 * wr) * WSi + 1 + 2 * wc;
        const u32x4 q[4] = {p[0], p[1], p[WSi], p[WSi + 1]};
        const u32x4 g = dy[(int64_t)cb * PTo + ((int64_t)n * HSo + 1 + wr) * WSo + 1 + wc];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float e0 = bfe(q[0], j), e1 = bfe(q[1], j), e2 = bfe(q[2], j), e3 = bfe(q[3], j);
            const float m = fmaxf(fmaxf(e0, e1), fmaxf(e2, e3));
            const int first = e0 == m ? 0 : (e1 == m ? 1 : (e2 == m ? 2 : 3));
            const float mine = pos == 0 ? e0 : (pos == 1 ? e1 : (pos == 2 ? e2 : e3));
            v[j] = (first == pos && (!relu_mask || mine > 0.f)) ? bfe(g, j) : 0.f;
        }
        o = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
    dx[(int64_t)cb * PTi + f] = o;
}

// dz = dy * (y > 0) on P8 tensors (pads stay zero: dy's pads are zero)
__global__ __launch_bounds__(256) void p8_relu_bwd_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ y,
                                                          u32x4* __restrict__ dz, int64_t n16)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const u32x4 g = dy[i], a = y[i];
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned lo = ((a[k] & 0x8000u) || !(a[k] & 0x7FFFu)) ? 0u : (g[k] & 0xFFFFu);            // y <= 0 (or -0)
        const unsigned hi = ((a[k] & 0x80000000u) || !(a[k] & 0x7FFF0000u)) ? 0u : (g[k] & 0xFFFF0000u);
        o[k] = lo | hi;
    }
    dz[i] = o;
}

// zero pad row / column vectors of a P8 tensor whose interior another kernel writes (the pooling epilogue of the conv kernel)
__global__ __launch_bounds__(256) void p8_zero_pads_kernel(u32x4* __restrict__ y, int HS, int WS, int ROWS, int64_t PT)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nimg1 = (ROWS - 1) / HS + 1;                 // pad rows
    int64_t f;
    if (i < ROWS) f = (int64_t)i * WS;                                       // column 0 of row i
    else if (i - ROWS < nimg1 * WS) f = (int64_t)((i - ROWS) / WS) * HS * WS + (i - ROWS) % WS;
    else return;
    y[(int64_t)blockIdx.y * PT + f] = (u32x4){0u, 0u, 0u, 0u};
}

// ------------------------------------------------------------------------------------------------ weight pack (forward / dgrad)
// wp[coTile][chunk][tap][mt][lane = 32 h + r][e]:  A operand of v_mfma_f32_32x32x16_bf16 -- row r = output channel
// coTile (32 MT) + 32 mt + r, k = 8 h + e = input channel within the 16-channel chunk; 16 B per lane, lane-linear: the kernel
// copies a (coTile, chunk) slab of 9 MT KB to LDS verbatim and every A fragment is one conflict-free ds_read_b128.
// mode 0: forward (rows = w's dim 0);  mode 1: dgrad (rows = w's dim 1, taps flipped: dX = conv(dY, W^T rotated by 180 degrees))
__global__ __launch_bounds__(256) void p8_pack_weights_kernel(const float* __restrict__ w, u16* __restrict__ wp, int w_cout, int w_cin,
                                                              int mode, int MT, int coTiles, int nChunks)
{
    const int convCout = mode ? w_cin : w_cout, convCin = mode ? w_cout : w_cin;
    const int64_t total = (int64_t)coTiles * nChunks * 9 * MT * 64 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int e = (int)(t & 7); t >>= 3;
        const int lane = (int)(t & 63); t >>= 6;
        const int mt = (int)(t % MT); t /= MT;
        const int tap = (int)(t % 9); t /= 9;
        const int chunk = (int)(t % nChunks);
        const int cot = (int)(t / nChunks);
        const int co = (cot * MT + mt) * 32 + (lane & 31), ci = chunk * 16 + 8 * (lane >> 5) + e;
        float v = 0.f;
        if (co < convCout && ci < convCin) {
            const int ky = tap / 3, kx = tap % 3;
            v = mode == 0 ? w[(((int64_t)co * w_cin + ci) * 3 + ky) * 3 + kx]
                          : w[(((int64_t)ci * w_cin + co) * 3 + (2 - ky)) * 3 + (2 - kx)];
        }
        wp[i] = (u16)(pack_bf16x2(v, 0.f) & 0xFFFFu);
    }
}

// ------------------------------------------------------------------------------------------------ forward / dgrad
// FLAT = false: a workgroup's pixels are a 2-D tile of TR rows x 32 columns of the (ROWS x WS) grid, the patch its (TR + 2) x 34 halo
// box.  Column tiles of 32 waste 13 % of the MFMAs on WS = 167 (the 100 x 166 maps) and 12.5 % on WS = 84 (50 x 83).
// FLAT = true (narrow maps, WS <= P8_FLAT_MAX_WS): the workgroup owns TP CONSECUTIVE FLAT pixels f0 .. f0 + TP - 1 of the plane --
// rows, images and pad positions just follow one another (the pads are in the tensor, so nothing special happens at a row or image
// end) -- and the patch is the contiguous flat range [f0 - WS - 1, f0 + TP + WS + 1): no column padding at all, one partly filled tile
// per launch.  Tap offsets ky * WS are then run-time values (three address registers instead of one).
constexpr int P8_FLAT_MAX_WS = 170;
template <int MT, bool FLAT>
struct P8G {
    static constexpr int NTB = 16 / MT;                  // pixel blocks (32 pixels of a row / 32 flat pixels) per wave
    static constexpr int TR = 4 * NTB;                   // tile rows per workgroup (2-D)
    static constexpr int TC = 32;                        // tile columns (2-D)
    static constexpr int TP = 4 * NTB * 32;              // tile pixels
    static constexpr int PR = TR + 2, PC = TC + 2;       // patch rows / columns (2-D: halo of one)
    static constexpr int PPL = FLAT ? TP + 2 * P8_FLAT_MAX_WS + 2 : PR * PC;     // patch pixels per 8-channel plane (FLAT: the most)
    static constexpr int PIN = (2 * PPL + P8T - 1) / P8T;   // patch DMA instructions per lane and chunk (2 planes)
    static constexpr int PBYTES = PIN * P8T * 16;        // patch bytes per stage (padded to whole DMA instructions)
    static constexpr int WBYTES = 9 * MT * 1024;         // weight slab bytes per (channel tile, chunk)
    static constexpr int WPIECES = 9 * MT;               // ... in 1-KB wave pieces
    static constexpr int WIN = (WPIECES + 3) / 4;        // weight DMA instructions per lane and chunk
    static constexpr int STAGE = WBYTES + PBYTES;
    static constexpr int NDMA = PIN + WIN;
};

template <int MT, bool FLAT, bool POOL = false>
__global__ __launch_bounds__(P8T, 1) void p8_conv3x3_kernel(
    const u16* __restrict__ x, const u16* __restrict__ wp, const float* __restrict__ bias, const u16* __restrict__ mref,
    u16* __restrict__ y, int Cout, int ycb, int HS, int WS, int ROWS, long long PT, int nChunks, int epi, int coTiles, int tilesC,
    int nPix, int nWork)
{
    using G = P8G<MT, FLAT>;
    constexpr int NTB = G::NTB;
    __shared__ __attribute__((aligned(16))) char lds[2 * G::STAGE];
    const int PPLF = G::TP + 2 * WS + 2;                 // FLAT: patch pixels per plane actually used (<= G::PPL)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PERSISTENT workgroups: one per CU (the two LDS stages leave no room for a second), each walking work items blockIdx.x,
    // + gridDim.x, ...  The last chunk of a tile treats the first chunk of the workgroup's NEXT tile as its successor -- DMA during
    // its taps, hand-over, first operand reads -- so that only the epilogue separates the MFMA streams of two tiles (one tile per
    // workgroup left the prologue -- 14 DMA issues + their latency -- exposed once per tile: 15 % of a 4-chunk conv1_2 tile).
    // work item -> (channel tile, pixel tile): the channel tiles of a pixel tile run back to back on ONE XCD (item id mod 8 = the
    // workgroup id mod 8: gridDim.x is a multiple of 8), so its patch leaves HBM once
    auto decode = [&](int t, int& cot_, int& R0_, int& C0_) {
        const int slot = t >> 3;
        cot_ = slot % coTiles;
        const int pix = (slot / coTiles) * 8 + (t & 7);
        if constexpr (FLAT) {                          // R0_ carries the tile's first FLAT pixel, C0_ is unused
            R0_ = pix * G::TP;
            C0_ = 0;
        } else if constexpr (POOL) {                   // tiles per IMAGE, origin at its pixel (0, 0): 2x2 windows never straddle tiles
            const int rt = pix / tilesC, bands = (HS - 1 + G::TR - 1) / G::TR;
            R0_ = (rt / bands) * HS + 1 + (rt % bands) * G::TR;
            C0_ = 1 + (pix % tilesC) * G::TC;
        } else {
            R0_ = (pix / tilesC) * G::TR;
            C0_ = (pix % tilesC) * G::TC;
        }
        return t < nWork && pix < nPix;
    };
    auto next_valid = [&](int t, int& cot_, int& R0_, int& C0_) {       // first valid item at or after t on this workgroup's walk, or -1
        while (t < nWork) {
            if (decode(t, cot_, R0_, C0_)) return t;
            t += (int)gridDim.x;
        }
        return -1;
    };
    int cot, R0, C0;
    int t = next_valid((int)blockIdx.x, cot, R0, C0);
    if (t < 0) return;

    // ---- DMA set-up: per-lane source offsets of a tile's patch pieces (the chunk advance moves the descriptor)
    auto patch_offsets = [&](int R0_, int C0_, unsigned (&pv)[G::PIN]) {
#pragma unroll
        for (int i = 0; i < G::PIN; ++i) {
            const int piece = (i * 4 + wave) * 64 + lane;
            if constexpr (FLAT) {
                const int plane = piece >= PPLF ? 1 : 0;
                const long long flat = (long long)R0_ - WS - 1 + (piece - plane * PPLF);
                const bool ok = piece < 2 * PPLF && flat >= 0 && flat < PT;
                pv[i] = ok ? (unsigned)((plane * PT + flat) * 16) : 0xFFFFFFFFu;
            } else {
                const int plane = piece >= G::PPL ? 1 : 0;
                const int q = piece - plane * G::PPL;
                const int prow = q / G::PC, pcol = q - prow * G::PC;
                const long long flat = (long long)(R0_ - 1 + prow) * WS + (C0_ - 1 + pcol);
                const bool ok = piece < 2 * G::PPL && flat >= 0 && flat < PT;
                pv[i] = ok ? (unsigned)((plane * PT + flat) * 16) : 0xFFFFFFFFu;
            }
        }
    };
    unsigned pv_cur[G::PIN], pv_next[G::PIN], pvd[G::PIN];     // this tile's / the next tile's / the ones the DMA slots use
    patch_offsets(R0, C0, pv_cur);
    const unsigned chunk_bytes_lo = (unsigned)((2 * PT * 16) & 0xFFFFFFFFll);      // < 4 GB (checked by the launcher)
    const unsigned wvoff = (unsigned)lane * 16u;

    // descriptors of a chunk's weight slab and of its two input planes; `live` = false gives EMPTY descriptors: the DMA
    // instructions are still issued (every lane out of range: zero fill, no memory traffic) so that the loop has no tail copies
    auto desc_w = [&](int cot_, int chunk, bool live) {
        return ptmi_rsrc(wp + ((size_t)(cot_ * nChunks + chunk) * G::WBYTES) / 2, live ? (unsigned)G::WBYTES : 0u);
    };
    auto desc_x = [&](int chunk, bool live) { return ptmi_rsrc(x + (size_t)chunk * 2 * PT * 8, live ? chunk_bytes_lo : 0u); };
    auto dma = [&](auto d_c, const __amdgpu_buffer_rsrc_t& rw, const __amdgpu_buffer_rsrc_t& rx, char* base, const unsigned (&pv)[G::PIN]) {
        constexpr int d = decltype(d_c)::value;
        if constexpr (d < G::WIN) {
            const int piece = d * 4 + wave;
            if (4 * G::WIN == G::WPIECES || piece < G::WPIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (plds_void_t*)(base + piece * 1024), 16, (int)wvoff, piece * 1024, 0, 0);
        } else if constexpr (d - G::WIN < G::PIN) {
            constexpr int i = d - G::WIN;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (plds_void_t*)(base + G::WBYTES + (i * 4 + wave) * 1024), 16, (int)pv[i], 0, 0, 0);
        }
    };

    // ---- operand addresses
    const int h = lane >> 5, px = lane & 31;
    const int a_off = lane * 16;                                                     // + stage + (tap MT + mt) 1024
    // B operand of pixel block n, tap (ky, kx):  2-D: patch slot (wave NTB + n + ky) PC + px + kx;  FLAT: slot wave NTB 32 + n 32 + px
    // + ky WS + kx (slot 0 = flat pixel f0 - WS - 1)
    const int b_off = FLAT ? G::WBYTES + (h * PPLF + wave * NTB * 32 + px) * 16
                           : G::WBYTES + (h * G::PPL + wave * NTB * G::PC + px) * 16;
    const int b_ky[3] = {b_off, b_off + WS * 16, b_off + 2 * WS * 16};              // (FLAT only)

    {   // the very first chunk of the workgroup: nothing to hide it behind
        const __amdgpu_buffer_rsrc_t rw = desc_w(cot, 0, true), rx = desc_x(0, true);
        dma(std::integral_constant<int, 0>{}, rw, rx, lds, pv_cur);   dma(std::integral_constant<int, 1>{}, rw, rx, lds, pv_cur);
        dma(std::integral_constant<int, 2>{}, rw, rx, lds, pv_cur);   dma(std::integral_constant<int, 3>{}, rw, rx, lds, pv_cur);
        dma(std::integral_constant<int, 4>{}, rw, rx, lds, pv_cur);   dma(std::integral_constant<int, 5>{}, rw, rx, lds, pv_cur);
        dma(std::integral_constant<int, 6>{}, rw, rx, lds, pv_cur);   dma(std::integral_constant<int, 7>{}, rw, rx, lds, pv_cur);
        dma(std::integral_constant<int, 8>{}, rw, rx, lds, pv_cur);   dma(std::integral_constant<int, 9>{}, rw, rx, lds, pv_cur);
        dma(std::integral_constant<int, 10>{}, rw, rx, lds, pv_cur);  dma(std::integral_constant<int, 11>{}, rw, rx, lds, pv_cur);
        dma(std::integral_constant<int, 12>{}, rw, rx, lds, pv_cur);  dma(std::integral_constant<int, 13>{}, rw, rx, lds, pv_cur);
        dma(std::integral_constant<int, 14>{}, rw, rx, lds, pv_cur);  dma(std::integral_constant<int, 15>{}, rw, rx, lds, pv_cur);
        dma(std::integral_constant<int, 16>{}, rw, rx, lds, pv_cur);  dma(std::integral_constant<int, 17>{}, rw, rx, lds, pv_cur);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    f32x16 acc[MT][NTB];
    ptmi_bf16x8 A0[MT], B0[NTB], A1[MT], B1[NTB];
    auto read_a = [&](const char* st, int tap, int m) { return *(const volatile plds_bf16x8_t*)(st + a_off + (tap * MT + m) * 1024); };
    auto read_b = [&](const char* st, int tap, int n) {
        if constexpr (FLAT) return *(const volatile plds_bf16x8_t*)(st + b_ky[tap / 3] + (n * 32 + tap % 3) * 16);
        else return *(const volatile plds_bf16x8_t*)(st + b_off + ((n + tap / 3) * G::PC + tap % 3) * 16);
    };
#pragma unroll
    for (int m = 0; m < MT; ++m) A0[m] = read_a(lds, 0, m);
#pragma unroll
    for (int n = 0; n < NTB; ++n) B0[n] = read_b(lds, 0, n);

    // One k-step = one tap of one 16-channel chunk = 16 MFMAs; the rest of the wave's work sits between them, one item per MFMA
    // (one wave per SIMD: whatever is not issued in the shadow of an MFMA leaves the matrix pipe idle):
    //   slots 0 .. MT + NTB - 1   one operand read each for the NEXT k-step (tap + 1, or tap 0 of the next chunk)
    //   slots 10, 12, 14 (taps 0 .. 5) one DMA instruction each for the chunk after this one (its <= 18 instructions)
    //   tap 8, before slot 0      hand-over: own DMA pieces of the next chunk landed (vmcnt(0): they were issued >= 3 taps ago),
    //                             this k-step's operands in registers (lgkmcnt(0)), workgroup barrier
    // The very first k-step of a tile has a zero C operand (no 256 accumulator writes).
    auto kstep = [&](auto tap_c, auto zc_c, const ptmi_bf16x8 (&A)[MT], const ptmi_bf16x8 (&B)[NTB], ptmi_bf16x8 (&An)[MT],
                     ptmi_bf16x8 (&Bn)[NTB], int cur, const __amdgpu_buffer_rsrc_t& rw, const __amdgpu_buffer_rsrc_t& rx) {
        constexpr int TAP = decltype(tap_c)::value;
        constexpr bool ZC = decltype(zc_c)::value;
        const char* src = lds + (TAP < 8 ? cur : cur ^ 1) * G::STAGE;
        char* dst = lds + (cur ^ 1) * G::STAGE;
        constexpr int NT_ = (TAP + 1) % 9;
        if constexpr (TAP == 8) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        auto slot_fn = [&](auto j_c) {
            constexpr int J = decltype(j_c)::value;
            constexpr int m = J / NTB, n = J % NTB;
            if constexpr (ZC) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], B[n], (f32x16){0}, 0, 0, 0);
            else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], B[n], acc[m][n], 0, 0, 0);
            if constexpr (J < MT) An[J] = read_a(src, NT_, J);
            else if constexpr (J < MT + NTB) Bn[J - MT] = read_b(src, NT_, J - MT);
            if constexpr (TAP < 6 && (J == 10 || J == 12 || J == 14))      // the next chunk into the other stage: everybody left it at the
                dma(std::integral_constant<int, TAP * 3 + (J - 10) / 2>{}, rw, rx, dst, pvd);      // last hand-over
            __builtin_amdgcn_sched_barrier(0);
        };
        slot_fn(std::integral_constant<int, 0>{});  slot_fn(std::integral_constant<int, 1>{});
        slot_fn(std::integral_constant<int, 2>{});  slot_fn(std::integral_constant<int, 3>{});
        slot_fn(std::integral_constant<int, 4>{});  slot_fn(std::integral_constant<int, 5>{});
        slot_fn(std::integral_constant<int, 6>{});  slot_fn(std::integral_constant<int, 7>{});
        slot_fn(std::integral_constant<int, 8>{});  slot_fn(std::integral_constant<int, 9>{});
        slot_fn(std::integral_constant<int, 10>{}); slot_fn(std::integral_constant<int, 11>{});
        slot_fn(std::integral_constant<int, 12>{}); slot_fn(std::integral_constant<int, 13>{});
        slot_fn(std::integral_constant<int, 14>{}); slot_fn(std::integral_constant<int, 15>{});
    };
    static_assert(G::NDMA <= 18, "the chunk's DMA instructions are issued three per tap over taps 0..5");
    static_assert(MT + NTB <= 10, "operand reads occupy the slots before the first DMA slot");

    const std::true_type T{};
    const std::false_type F{};
    // nine k-steps per chunk: the two operand register sets swap roles from chunk to chunk.  A tile's chunks alternate
    // even / odd bodies starting with an even one (three instantiations: even with a zero C operand, odd, even); the LDS stage
    // is a running parity (`st`) of its own, because a tile with an odd chunk count hands the next tile the other stage.
    auto chunk_even = [&](auto zc_c, int st, const __amdgpu_buffer_rsrc_t& rw, const __amdgpu_buffer_rsrc_t& rx) {
        kstep(std::integral_constant<int, 0>{}, zc_c, A0, B0, A1, B1, st, rw, rx);
        kstep(std::integral_constant<int, 1>{}, F, A1, B1, A0, B0, st, rw, rx);
        kstep(std::integral_constant<int, 2>{}, F, A0, B0, A1, B1, st, rw, rx);
        kstep(std::integral_constant<int, 3>{}, F, A1, B1, A0, B0, st, rw, rx);
        kstep(std::integral_constant<int, 4>{}, F, A0, B0, A1, B1, st, rw, rx);
        kstep(std::integral_constant<int, 5>{}, F, A1, B1, A0, B0, st, rw, rx);
        kstep(std::integral_constant<int, 6>{}, F, A0, B0, A1, B1, st, rw, rx);
        kstep(std::integral_constant<int, 7>{}, F, A1, B1, A0, B0, st, rw, rx);
        kstep(std::integral_constant<int, 8>{}, F, A0, B0, A1, B1, st, rw, rx);
    };
    auto chunk_odd = [&](int st, const __amdgpu_buffer_rsrc_t& rw, const __amdgpu_buffer_rsrc_t& rx) {
        kstep(std::integral_constant<int, 0>{}, F, A1, B1, A0, B0, st, rw, rx);
        kstep(std::integral_constant<int, 1>{}, F, A0, B0, A1, B1, st, rw, rx);
        kstep(std::integral_constant<int, 2>{}, F, A1, B1, A0, B0, st, rw, rx);
        kstep(std::integral_constant<int, 3>{}, F, A0, B0, A1, B1, st, rw, rx);
        kstep(std::integral_constant<int, 4>{}, F, A1, B1, A0, B0, st, rw, rx);
        kstep(std::integral_constant<int, 5>{}, F, A0, B0, A1, B1, st, rw, rx);
        kstep(std::integral_constant<int, 6>{}, F, A1, B1, A0, B0, st, rw, rx);
        kstep(std::integral_constant<int, 7>{}, F, A0, B0, A1, B1, st, rw, rx);
        kstep(std::integral_constant<int, 8>{}, F, A1, B1, A0, B0, st, rw, rx);
    };
    // (an accumulator element extracted in C++ makes the compiler copy whole 16-register tiles to VGPRs, all of them at the head of
    // the epilogue -- hundreds of spills; explicit v_accvgpr_read keeps the tiles where they are)
    auto rd = [](float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; };
    const unsigned plane_bytes = (unsigned)(PT * 16);

    int st = 0;                                           // LDS stage of the chunk about to be computed
    for (;;) {
        int cotn, R0n, C0n;
        const int tn = next_valid(t + (int)gridDim.x, cotn, R0n, C0n);
        const bool more_tiles = tn >= 0;
        if (more_tiles) patch_offsets(R0n, C0n, pv_next);
#pragma unroll
        for (int i = 0; i < G::PIN; ++i) pvd[i] = pv_cur[i];
        // biases of the lane's channels: co = (cot MT + m) 32 + 8 q + 4 h + e; used in the epilogue, fetched now
        f32x4 bv[MT][4];
        {
            const __amdgpu_buffer_rsrc_t rb = ptmi_rsrc(bias ? (const void*)bias : (const void*)y, bias ? (unsigned)Cout * 4u : 0u);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bv[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (epi <= 1)
                        bv[m][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, ((cot * MT + m) * 32 + 8 * q + 4 * h) * 4, 0, 0));
                }
        }
        // the successor of chunk c: chunk c + 1 of this tile, or -- after the last one -- chunk 0 of the workgroup's next tile
        auto succ = [&](int chunk, __amdgpu_buffer_rsrc_t& rw, __amdgpu_buffer_rsrc_t& rx) {
            const bool last = chunk + 1 == nChunks;
            if (last) {
#pragma unroll
                for (int i = 0; i < G::PIN; ++i) pvd[i] = pv_next[i];
            }
            rw = desc_w(last ? cotn : cot, last ? 0 : chunk + 1, !last || more_tiles);
            rx = desc_x(last ? 0 : chunk + 1, !last || more_tiles);
        };
        {
            __amdgpu_buffer_rsrc_t rw, rx;
            succ(0, rw, rx);
            chunk_even(T, st, rw, rx);
            st ^= 1;
            if (nChunks > 1) {
                succ(1, rw, rx);
                chunk_odd(st, rw, rx);
                st ^= 1;
            }
            for (int chunk = 2; chunk < nChunks; chunk += 2) {
                succ(chunk, rw, rx);
                chunk_even(F, st, rw, rx);
                st ^= 1;
                if (chunk + 1 < nChunks) {
                    succ(chunk + 1, rw, rx);
                    chunk_odd(st, rw, rx);
                    st ^= 1;
                }
            }
            if (nChunks & 1) {       // an even body ran last: the next tile's first operands sit in the odd register set
#pragma unroll
                for (int m = 0; m < MT; ++m) A0[m] = A1[m];
#pragma unroll
                for (int n = 0; n < NTB; ++n) B0[n] = B1[n];
            }
        }

    if constexpr (POOL) {
        // bias + ReLU + 2x2 max-pool (floor): y has the POOLED geometry.  A wave's NTB rows are NTB / 2 window rows (tile origins are
        // even image rows / columns), the two columns of a window sit in neighbouring lanes (quad_perm 1,0,3,2); lanes with an even
        // column hold the pooled pixel, and permlane32_swap pairs two pooled rows into 16-byte stores as below.
        const int Ho = (HS - 1) >> 1, Wo = (WS - 1) >> 1, HSo = Ho + 1, WSo = Wo + 1;
        const long long PTo = ((long long)((ROWS - 1) / HS) * HSo + 1) * WSo;
        const unsigned oplane_bytes = (unsigned)(PTo * 16);
        const int img = (R0 - 1) / HS;
        const int pr0 = (R0 - 1 - img * HS + wave * NTB) >> 1, pcol = (C0 - 1 + px) >> 1;
        unsigned psv[NTB / 4];
#pragma unroll
        for (int j = 0; j < NTB / 4; ++j) {
            const int prow = pr0 + 2 * j + h;              // lower lanes store pooled row 2 j, upper lanes 2 j + 1
            const bool ok = !(px & 1) && prow < Ho && pcol < Wo;
            psv[j] = ok ? (unsigned)((((long long)img * HSo + 1 + prow) * WSo + 1 + pcol) * 16) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cb = (cot * MT + m) * 4 + q;
                if (cb >= ycb) continue;
                const __amdgpu_buffer_rsrc_t ry = ptmi_rsrc(y + (size_t)cb * PTo * 8, oplane_bytes);
                u32x2 o[NTB / 2];
#pragma unroll
                for (int n2 = 0; n2 < NTB / 2; ++n2) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = rd(acc[m][2 * n2][4 * q + e]) + bv[m][q][e], b = rd(acc[m][2 * n2 + 1][4 * q + e]) + bv[m][q][e];
                        const float vm = fmaxf(fmaxf(a, b), 0.f);
                        const float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vm), 0xB1, 0xF, 0xF, true));
                        v[e] = fmaxf(vm, nb);
                    }
                    o[n2] = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                }
#pragma unroll
                for (int j = 0; j < NTB / 4; ++j) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(o[2 * j][0], o[2 * j + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(o[2 * j][1], o[2 * j + 1][1], false, false);
                    const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(w, ry, (int)psv[j], 0, 0);
                }
            }
        }
    } else {
    unsigned mvoff[NTB];                     // byte offset of the lane's 8-byte half (mask loads; 0xFFFFFFFF: outside the grid)
    bool zero[NTB];                          // pad position: store zeros
    unsigned svoff[NTB / 2];                 // byte offset of the 16-byte vector this lane stores for the block pair (n0, n1)
    if constexpr (FLAT) {
        // flat pixel f = f0 + 32 (wave NTB + n) + px -> (row, column): one division, then +32 columns per block
        const long long f = (long long)R0 + (wave * NTB) * 32 + px;
        int row = (int)(f / WS), col = (int)(f - (long long)row * WS), rmod = row % HS;
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
            const long long fn = f + 32 * n;
            zero[n] = rmod == 0 || col == 0;
            mvoff[n] = fn < PT ? (unsigned)(fn * 16 + h * 8) : 0xFFFFFFFFu;
            col += 32;
            while (col >= WS) {                      // (WS >= 1: a 1-pixel-wide map wraps up to 16 times per block)
                col -= WS;
                rmod = rmod + 1 == HS ? 0 : rmod + 1;
            }
        }
    } else {
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
            const int Rg = R0 + wave * NTB + n, Cg = C0 + px;
            const bool inside = Rg < ROWS && Cg < WS;
            zero[n] = (Rg % HS) == 0 || Cg == 0;
            mvoff[n] = inside ? (unsigned)(((long long)Rg * WS + Cg) * 16 + h * 8) : 0xFFFFFFFFu;
        }
    }
#pragma unroll
    for (int j = 0; j < NTB / 2; ++j) {
        const unsigned own = h ? mvoff[2 * j + 1] : mvoff[2 * j];
        svoff[j] = own == 0xFFFFFFFFu ? own : own - (unsigned)h * 8u;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cb = (cot * MT + m) * 4 + q;                     // 8-channel plane of the output
            if (cb >= ycb) continue;                                   // (wave-uniform) planes beyond the tensor
            const __amdgpu_buffer_rsrc_t ry = ptmi_rsrc(y + (size_t)cb * PT * 8, plane_bytes);
            u32x2 mk[NTB];
            if (epi == 3) {
                const __amdgpu_buffer_rsrc_t rm = ptmi_rsrc(mref + (size_t)cb * PT * 8, plane_bytes);
#pragma unroll
                for (int n = 0; n < NTB; ++n) mk[n] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rm, (int)mvoff[n], 0, 0));
            }
            u32x2 o[NTB];
#pragma unroll
            for (int n = 0; n < NTB; ++n) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = rd(acc[m][n][4 * q + e]) + bv[m][q][e];
                    if (epi == 1) v[e] = fmaxf(v[e], 0.f);
                }
                if (epi == 3) {
                    // mask of the producing layer's ReLU: its stored activation > 0 (bf16 bit patterns: sign clear, magnitude set)
                    if ((mk[n][0] & 0x8000u) || !(mk[n][0] & 0x7FFFu)) v[0] = 0.f;
                    if ((mk[n][0] & 0x80000000u) || !(mk[n][0] & 0x7FFF0000u)) v[1] = 0.f;
                    if ((mk[n][1] & 0x8000u) || !(mk[n][1] & 0x7FFFu)) v[2] = 0.f;
                    if ((mk[n][1] & 0x80000000u) || !(mk[n][1] & 0x7FFF0000u)) v[3] = 0.f;
                }
                o[n] = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                if (zero[n]) o[n] = (u32x2){0u, 0u};
            }
#pragma unroll
            for (int j = 0; j < NTB / 2; ++j) {
                // (vdst, vsrc) -> vdst keeps its lower lanes and takes vsrc's lower lanes into its upper ones; vsrc takes vdst's upper
                // lanes into its lower ones and keeps its upper lanes:  lower lanes end with (own n0 half, partner's n0 half), upper
                // lanes with (partner's n1 half, own n1 half) -- channels 0..3 first in both
                const auto s0 = __builtin_amdgcn_permlane32_swap(o[2 * j][0], o[2 * j + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(o[2 * j][1], o[2 * j + 1][1], false, false);
                const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(w, ry, (int)svoff[j], 0, 0);
            }
        }
    }
    }
        // ---- next tile of this workgroup (its first chunk is already in LDS, its first operands in registers)
        if (!more_tiles) break;
        t = tn; cot = cotn; R0 = R0n; C0 = C0n;
#pragma unroll
        for (int i = 0; i < G::PIN; ++i) pv_cur[i] = pv_next[i];
    }
}

inline int p8_mt(int cout) { return cout <= 64 ? 2 : 4; }

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][ci][tap] = sum over ALL flat pixels f of dY[co][f] X[ci][f + (ky - 1) WS + (kx - 1)]   (pads of dY are zero, pads of X are
// the convolution's zero padding: a plain GEMM M = co, N = ci, K = pixels with nine shifted B operands).  The contraction runs
// over PIXELS while the layout packs CHANNELS: operands come out of LDS through ds_read_b64_tr_b16 -- a 16-lane group hands in
// the addresses of sixteen 8-byte pieces (4 pixels x 4 channel quads) and each lane receives 4 consecutive PIXELS of one channel;
// two of them are a lane's 8 k values.  (Semantics measured with tools/exp/tr16_probe.hip: result lane i, element j =
// element i % 4 of the piece addressed by lane 4 j + i / 4 of the group.)
//   workgroup = 8 waves (two per SIMD, <= 256 registers each) = 128 co x 64 ci x 9 taps; wave (cw, iw, tg): co half, ci half, taps
//   0..4 / 5..8 -- the second tap group's fifth accumulator pair multiplies dY by an all-ones operand: every column of it is the
//   BIAS gradient sum_f dY[co][f], for free and with the two groups issuing 10 MFMAs per k-step each;
//   K tile = 4 rows x 32 columns of the (ROWS x WS) grid = 8 k-steps of 16 pixels; LDS stage = dY [16 planes][128 px] (plane
//   pitch + 64 B: the four planes a 32-lane read touches sit on disjoint bank quarters) + X [8 planes][6 x 34 px] (pitch 3264 B:
//   the same by itself); two stages, the next tile's DMA issued in the tile's first k-steps;
//   split-K over contiguous tile ranges, fp32 partials [split][tap][co][ci] (+ [split][co] for db), fixed-order reduction.
constexpr int GT = 512;                      // threads
constexpr int G_CO = 128, G_CI = 64;
constexpr int G_DYP = 128 * 16 + 64;         // dY plane pitch in LDS (bytes)
constexpr int G_XPL = 6 * 34;                // X patch pixels per plane
constexpr int G_XP = G_XPL * 16;             // X plane pitch (3264 B = 816 dwords = 48 banks mod 64)
constexpr int G_DYB = 16 * G_DYP;            // 33792
constexpr int G_XB = 26 * 1024;              // 8 planes x 204 pieces = 1632 pieces, padded to 26 wave instructions
constexpr int G_STAGE = G_DYB + G_XB;        // 60416


__global__ __launch_bounds__(GT, 2) void p8_wgrad_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, float* __restrict__ ws,
                                                         float* __restrict__ wsb, int Cin, int Cout, int xcb, int dcb, int WS, int ROWS,
                                                         long long PT, int ciTiles, int S, int tilesC, int nTiles)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * G_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, iw = (wave >> 1) & 1, tg = wave >> 2;
    // workgroup -> (channel-tile pair, split): the splits of a pair share nothing but the weights they add up to
    const int pair = blockIdx.x / S, split = blockIdx.x - pair * S;
    const int cot = pair / ciTiles, cit = pair - cot * ciTiles;
    const int t0 = (int)((long long)nTiles * split / S), t1 = (int)((long long)nTiles * (split + 1) / S);

    const unsigned x_bytes = (unsigned)((long long)xcb * PT * 16), dy_bytes = (unsigned)((long long)dcb * PT * 16);
    const __amdgpu_buffer_rsrc_t rx = ptmi_rsrc(x, x_bytes), rdy = ptmi_rsrc(dy, dy_bytes);

    // ---- DMA pieces of this lane: dY 4 (wave w: planes 2 w, 2 w + 1; 128 pixels each), X 4 (pieces 64 (4 w + i) + lane of 1632).
    // Per lane: byte offset relative to the tile's first pixel (loop invariant); per tile: one scalar base + the validity of the
    // piece (rows / columns beyond the grid are ZERO for dY: a column beyond WS would alias the next row's pixels)
    int dy_row[4], dy_col[4];
    unsigned dy_lane[4];
    bool dy_chan[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int plane = 2 * wave + (i >> 1), pxl = (i & 1) * 64 + lane;
        dy_row[i] = pxl >> 5;
        dy_col[i] = pxl & 31;
        const int pg = cot * 16 + plane;
        dy_chan[i] = pg < dcb;
        dy_lane[i] = (unsigned)((long long)pg * PT * 16) + (unsigned)(dy_row[i] * WS + dy_col[i]) * 16u;
    }
    int x_lane[4];
    unsigned x_plane[4];
    bool x_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = (4 * wave + i) * 64 + lane;
        const int plane = q / G_XPL, slot = q - plane * G_XPL;
        x_lane[i] = (slot / 34 - 1) * WS + (slot % 34 - 1);
        const int pg = cit * 8 + plane;
        x_ok[i] = q < 8 * G_XPL && 4 * wave + i < 26 && pg < xcb;
        x_plane[i] = (unsigned)((long long)pg * PT * 16);
    }
    auto issue = [&](auto i_c, int R0, int C0, int st) {
        constexpr int I = decltype(i_c)::value;                 // 0..3 dY, 4..7 X
        const int tb = R0 * WS + C0;                             // (scalar) the tile's first flat pixel; < PT < 2^30
        char* base = lds + st * G_STAGE;
        if constexpr (I < 4) {
            const bool ok = dy_chan[I] && R0 + dy_row[I] < ROWS && C0 + dy_col[I] < WS;
            const unsigned off = ok ? dy_lane[I] + (unsigned)tb * 16u : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (plds_void_t*)(base + (2 * wave + (I >> 1)) * G_DYP + (I & 1) * 1024), 16, (int)off, 0, 0, 0);
        } else {
            constexpr int i = I - 4;
            if (4 * wave + i < 26) {                             // (wave-uniform)
                const unsigned flat = (unsigned)(tb + x_lane[i]);            // negative -> huge: out of range
                const bool ok = x_ok[i] && flat < (unsigned)PT;
                const unsigned off = ok ? x_plane[i] + flat * 16u : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (plds_void_t*)(base + G_DYB + (4 * wave + i) * 1024), 16, (int)off, 0, 0, 0);
            }
        }
    };

    // ---- operand addresses (ds_read_b64_tr_b16): lane a of a 16-lane group addresses pixel (a >> 2), channel quad (a & 3) of the
    // group's 16 channels; group gq = (lane >> 4): rows 16 (gq & 1) .. + 15 of the 32-row operand, k half gq >> 1
    const int a = lane & 15, gq = lane >> 4, kh = gq >> 1;
    const int lane_px = 8 * kh + (a >> 2);                                               // + 16 s + 4 u
    const int lane_pl = 2 * (gq & 1) + ((a & 3) >> 1);                                    // plane within the 32-channel tile
    const int a_base = (4 * (2 * cw) + lane_pl) * G_DYP + lane_px * 16 + (a & 1) * 8;      // + t 4 G_DYP + s 256 + u 64
    int b_base[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int tap = min(tg * 5 + t, 8);
        b_base[t] = G_DYB + (4 * iw + lane_pl) * G_XP + ((tap / 3) * 34 + tap % 3 + lane_px) * 16 + (a & 1) * 8;   // + ((s >> 1) 34 + 16 (s & 1) + 4 u) 16
    }
    const bool ones_slot = tg == 1;                                                       // (wave-uniform) slot 4 = bias gradient

    f32x16 acc[2][5];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][t][e] = 0.f;

    const ptmi_bf16x8 ones = __builtin_bit_cast(ptmi_bf16x8, (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
    const unsigned stage_addr = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds;      // LDS byte address of stage 0

    if (t0 < t1) {
        const int R0 = (t0 / tilesC) * 4, C0 = (t0 % tilesC) * 32;
        issue(std::integral_constant<int, 0>{}, R0, C0, 0); issue(std::integral_constant<int, 1>{}, R0, C0, 0);
        issue(std::integral_constant<int, 2>{}, R0, C0, 0); issue(std::integral_constant<int, 3>{}, R0, C0, 0);
        issue(std::integral_constant<int, 4>{}, R0, C0, 0); issue(std::integral_constant<int, 5>{}, R0, C0, 0);
        issue(std::integral_constant<int, 6>{}, R0, C0, 0); issue(std::integral_constant<int, 7>{}, R0, C0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int tile = t0; tile < t1; ++tile) {
        const int st = (tile - t0) & 1;
        const int R0 = (tile / tilesC) * 4, C0 = (tile % tilesC) * 32;
        const int R1 = ((tile + 1) / tilesC) * 4, C1 = ((tile + 1) % tilesC) * 32;
        const bool more = tile + 1 < t1;
        auto kstep = [&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            // the next tile's pieces go out EARLY (their latency hides behind the rest of the tile; issued at the end of the tile
            // the hand-over waited 2/3 of the kernel's time for them), four per k-step, and staggered between the two waves of a
            // SIMD (tap group 0: k-steps 0, 1; tap group 1: k-steps 2, 3) so that one of them keeps the matrix pipe busy meanwhile;
            // into the other stage: everybody left it at the last barrier
            if constexpr (s < 4) {
                if (more && (s >> 1) == tg) {
                    issue(std::integral_constant<int, (s & 1) * 4 + 0>{}, R1, C1, st ^ 1);
                    issue(std::integral_constant<int, (s & 1) * 4 + 1>{}, R1, C1, st ^ 1);
                    issue(std::integral_constant<int, (s & 1) * 4 + 2>{}, R1, C1, st ^ 1);
                    issue(std::integral_constant<int, (s & 1) * 4 + 3>{}, R1, C1, st ^ 1);
                }
            }
            // k-steps whose 16 pixels lie beyond the grid carry only zeros in dY: skip them (wave-uniform)
            if (R0 + (s >> 1) < ROWS && C0 + 16 * (s & 1) < WS) {
                // operand reads as inline asm: seen as LDS loads by the compiler, each k-step's first one gets an `s_waitcnt vmcnt(0)`
                // in front once LDS-DMA instructions are in flight (it cannot tell the DMA's stage from the one being read) -- the
                // waves then sat out the latency of the pieces they had just issued, four k-steps per tile (conv3_2: 3.76 -> 3.35 ms).
                // (Requesting k-step s + 1's operands before k-step s's MFMAs -- two register sets, counted lgkmcnt -- was tried on
                // top: 250 registers, 3.46 ms; the SIMD's second wave already fills those gaps.)
                const unsigned sa = stage_addr + (unsigned)(st * G_STAGE);
                constexpr int so = ((s >> 1) * 34 + 16 * (s & 1)) * 16;
                u32x2 ra[4], rb[10];
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ra[0]) : "v"(sa + (unsigned)a_base), "n"(s * 256));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ra[1]) : "v"(sa + (unsigned)a_base), "n"(s * 256 + 64));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ra[2]) : "v"(sa + (unsigned)a_base), "n"(4 * G_DYP + s * 256));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ra[3]) : "v"(sa + (unsigned)a_base), "n"(4 * G_DYP + s * 256 + 64));
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(rb[2 * t]) : "v"(sa + (unsigned)b_base[t]), "n"(so));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(rb[2 * t + 1]) : "v"(sa + (unsigned)b_base[t]), "n"(so + 64));
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]),
                               "+v"(rb[5]), "+v"(rb[6]), "+v"(rb[7]), "+v"(rb[8]), "+v"(rb[9]));
                const ptmi_bf16x8 A0 = __builtin_bit_cast(ptmi_bf16x8, (u32x4){ra[0][0], ra[0][1], ra[1][0], ra[1][1]});
                const ptmi_bf16x8 A1 = __builtin_bit_cast(ptmi_bf16x8, (u32x4){ra[2][0], ra[2][1], ra[3][0], ra[3][1]});
                ptmi_bf16x8 B[5];
#pragma unroll
                for (int t = 0; t < 5; ++t) B[t] = __builtin_bit_cast(ptmi_bf16x8, (u32x4){rb[2 * t][0], rb[2 * t][1], rb[2 * t + 1][0], rb[2 * t + 1][1]});
                if (ones_slot) B[4] = ones;
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B[t], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[t], acc[1][t], 0, 0, 0);
                }
            }
        };
        kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{}); kstep(std::integral_constant<int, 3>{});
        kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
        kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- partials: C layout column = lane & 31 = ci, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = co
    const int ci = cit * G_CI + iw * 32 + (lane & 31);
    const int co0 = cot * G_CO + cw * 64 + 4 * (lane >> 5);
    float* wsp = ws + (size_t)split * 9 * Cout * Cin;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int tap = tg * 5 + t;
            if (tap < 9) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2);
                    if (co < Cout && ci < Cin) wsp[((size_t)tap * Cout + co) * Cin + ci] = acc[m][t][r];
                }
            } else if (cit == 0 && iw == 0 && (lane & 31) == 0) {          // bias gradient: any one column of the ones product
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2);
                    if (co < Cout) wsb[(size_t)split * Cout + co] = acc[m][t][r];
                }
            }
        }
}

// dW (Cout, Cin, 3, 3) = sum of the split partials in split order (+ dW if accumulate); db likewise
__global__ __launch_bounds__(256) void p8_wgrad_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ dw,
                                                              float* __restrict__ db, int Cout, int Cin, int S, int accumulate)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t cc = (int64_t)Cout * Cin;
    if (i < cc) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += ws[((int64_t)s * 9 + tap) * cc + i];
            float* o = dw + i * 9 + tap;
            *o = accumulate ? *o + v : v;
        }
    }
    if (db && i < Cout) {
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += wsb[(int64_t)s * Cout + i];
        db[i] = accumulate ? db[i] + v : v;
    }
}

inline int p8_wgrad_splits(int n, int cin, int cout, int h, int w, int waves = 1)
{
    const P8Dims d = p8_dims(n, h, w);
    const int pairs = cdiv(cout, G_CO) * cdiv(cin, G_CI);
    const int64_t tiles = (int64_t)cdiv(d.ROWS, 4) * cdiv(d.WS, 32);
    int S = cdiv(256 * (waves < 1 ? 1 : waves > 16 ? 16 : waves), pairs);
    if (S > tiles) S = (int)tiles;
    return S < 1 ? 1 : S;
}

}  // namespace

extern "C" {

int64_t ptmi_p8_plane_pixels(int n, int h, int w)
{
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return p8_dims(n, h, w).PT;
}

int ptmi_p8_from_nchw(const float* x, void* y, int n, int c, int h, int w, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && y && n > 0 && c > 0 && h > 0 && w > 0, "p8_from_nchw: bad args");
    const P8Dims d = p8_dims(n, h, w);
    const int cb_out = ptmi_p8_planes(c);
    hipLaunchKernelGGL(p8_from_nchw_kernel, dim3((unsigned)cdiv64(d.PT, 256), cb_out), dim3(256), 0, (hipStream_t)s, x, (u32x4*)y, c, h,
                       w, d.HS, d.WS, d.PT, cb_out);
    PTMI_LAUNCH_CHECK("p8_from_nchw");
    return 0;
}

int ptmi_p8_to_nchw(const void* x, float* y, int n, int c, int h, int w, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && y && n > 0 && c > 0 && h > 0 && w > 0, "p8_to_nchw: bad args");
    const P8Dims d = p8_dims(n, h, w);
    hipLaunchKernelGGL(p8_to_nchw_kernel, dim3((unsigned)cdiv64((int64_t)n * h * w, 256), cdiv(c, 8)), dim3(256), 0, (hipStream_t)s,
                       (const u32x4*)x, y, n, c, h, w, d.HS, d.WS, d.PT);
    PTMI_LAUNCH_CHECK("p8_to_nchw");
    return 0;
}

int ptmi_p8_maxpool2x2_fwd(const void* x, void* y, int n, int c, int h, int w, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && y && n > 0 && c > 0 && h > 1 && w > 1, "p8_maxpool2x2_fwd: bad args");
    const P8Dims di = p8_dims(n, h, w), dq = p8_dims(n, h / 2, w / 2);
    hipLaunchKernelGGL(p8_maxpool_fwd_kernel, dim3((unsigned)cdiv64(dq.PT, 256), ptmi_p8_planes(c)), dim3(256), 0, (hipStream_t)s, (const u32x4*)x,
                       (u32x4*)y, di.HS, di.WS, di.PT, dq.HS, dq.WS, dq.PT);
    PTMI_LAUNCH_CHECK("p8_maxpool2x2_fwd");
    return 0;
}

int ptmi_p8_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int n, int c, int h, int w, int relu_mask, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && dy && dx && n > 0 && c > 0 && h > 1 && w > 1, "p8_maxpool2x2_bwd: bad args");
    const P8Dims di = p8_dims(n, h, w), dq = p8_dims(n, h / 2, w / 2);
    hipLaunchKernelGGL(p8_maxpool_bwd_kernel, dim3((unsigned)cdiv64(di.PT, 256), ptmi_p8_planes(c)), dim3(256), 0, (hipStream_t)s, (const u32x4*)x,
                       (const u32x4*)dy, (u32x4*)dx, h, w, di.HS, di.WS, di.PT, dq.HS, dq.WS, dq.PT, relu_mask);
    PTMI_LAUNCH_CHECK("p8_maxpool2x2_bwd");
    return 0;
}

int ptmi_p8_relu_bwd(const void* dy, const void* y, void* dz, int64_t pixels16, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(dy && y && dz && pixels16 >= 0, "p8_relu_bwd: bad args");
    if (pixels16 == 0) return 0;
    hipLaunchKernelGGL(p8_relu_bwd_kernel, dim3((unsigned)cdiv64(pixels16, 256)), dim3(256), 0, (hipStream_t)s, (const u32x4*)dy,
                       (const u32x4*)y, (u32x4*)dz, pixels16);
    PTMI_LAUNCH_CHECK("p8_relu_bwd");
    return 0;
}

int ptmi_p8_planes(int c) { return c > 0 ? 2 * cdiv(c, 16) : 0; }

int64_t ptmi_p8_packed_elems(int cin, int cout)
{
    const int MT = p8_mt(cout);
    return (int64_t)cdiv(cout, 32 * MT) * cdiv(cin, 16) * 9 * MT * 512;
}

int ptmi_p8_pack_weights(const float* w, void* wp, int w_cout, int w_cin, int mode, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(w && wp && w_cout > 0 && w_cin > 0, "p8_pack_weights: bad args");
    const int convCout = mode ? w_cin : w_cout, convCin = mode ? w_cout : w_cin;
    const int MT = p8_mt(convCout), coTiles = cdiv(convCout, 32 * MT), nChunks = cdiv(convCin, 16);
    const int64_t total = (int64_t)coTiles * nChunks * 9 * MT * 512;
    const int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
    hipLaunchKernelGGL(p8_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (u16*)wp, w_cout, w_cin, mode, MT, coTiles,
                       nChunks);
    PTMI_LAUNCH_CHECK("p8_pack_weights");
    return 0;
}

int ptmi_p8_conv3x3(const void* x, const void* wp, const float* bias, const void* mask_ref, void* y, int n, int cin, int cout,
                    int h, int w, int epilogue, ptmi_stream_t s)
{
    return ptmi_p8_conv3x3_waves(x, wp, bias, mask_ref, y, n, cin, cout, h, w, epilogue, 1, s);
}

int ptmi_p8_conv3x3_waves(const void* x, const void* wp, const float* bias, const void* mask_ref, void* y, int n, int cin, int cout,
                          int h, int w, int epilogue, int waves, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && wp && y && n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "p8_conv3x3: bad args");
    PTMI_CHECK_ARG(epilogue >= 0 && epilogue <= 4, "p8_conv3x3: bad epilogue %d", epilogue);
    PTMI_CHECK_ARG((epilogue > 1 && epilogue != 4) || bias, "p8_conv3x3: bias required for epilogue %d", epilogue);
    PTMI_CHECK_ARG(epilogue != 4 || (h > 1 && w > 1), "p8_conv3x3: pooling epilogue on a %d x %d map", h, w);
    PTMI_CHECK_ARG(epilogue != 3 || mask_ref, "p8_conv3x3: mask_ref required for epilogue 3");
    const P8Dims d = p8_dims(n, h, w);
    PTMI_CHECK_ARG(d.PT * 32 < (1ll << 32), "p8_conv3x3: %lld pixels per plane exceed the 32-bit buffer offsets", (long long)d.PT);
    const int MT = p8_mt(cout), coTiles = cdiv(cout, 32 * MT), nChunks = cdiv(cin, 16), ycb = ptmi_p8_planes(cout);
    // narrow maps: flat tile line (no column padding); wide maps: 2-D tiles (a flat tile's halo -- two whole rows -- would not fit)
    const bool pool = epilogue == 4;                                     // 2-D tiles per image (see the kernel's decode)
    const bool flat = !pool && d.WS <= P8_FLAT_MAX_WS;
    const int TR = MT == 4 ? P8G<4, false>::TR : P8G<2, false>::TR, TP = MT == 4 ? P8G<4, true>::TP : P8G<2, true>::TP;
    const int tilesC = pool ? cdiv(w, 32) : cdiv(d.WS, 32);
    const int64_t nPix = flat ? cdiv64(d.PT, TP) : (int64_t)(pool ? n * cdiv(h, TR) : cdiv(d.ROWS, TR)) * tilesC;
    const int64_t nWork = cdiv64(nPix, 8) * 8 * coTiles;                // work items (some beyond nPix: skipped by the kernel)
    PTMI_CHECK_ARG(nWork < (1ll << 31), "p8_conv3x3: too many tiles");
    // persistent workgroups: one per CU (a multiple of 8: a work item stays on the XCD of its id mod 8)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
        cus = 256;
    // waves > 1: that many fills of the one-workgroup-per-CU slots, each workgroup walking a 1 / waves share of the items (still a
    // multiple of 8: the XCD affinity of an item id stays) -- the hardware dispatcher then hands the later workgroups to whichever
    // CU frees up first, so CUs held by another kernel (a collective overlapping backward) cost their share, not a second pass
    const int64_t slots = (int64_t)(cus / 8) * 8 * (waves < 1 ? 1 : waves > 64 ? 64 : waves);
    const int64_t grid = nWork < slots ? nWork : slots;
#define P8_LAUNCH(MT_, FLAT_, POOL_)                                                                                                       \
    hipLaunchKernelGGL((p8_conv3x3_kernel<MT_, FLAT_, POOL_>), dim3((unsigned)grid), dim3(P8T), 0, (hipStream_t)s, (const u16*)x, (const u16*)wp, bias, \
                       (const u16*)mask_ref, (u16*)y, cout, ycb, d.HS, d.WS, d.ROWS, (long long)d.PT, nChunks, epilogue, coTiles, tilesC,   \
                       (int)nPix, (int)nWork)
    if (pool) {
        const P8Dims q = p8_dims(n, h / 2, w / 2);
        hipLaunchKernelGGL(p8_zero_pads_kernel, dim3((unsigned)cdiv(q.ROWS + (n + 1) * q.WS, 256), ycb), dim3(256), 0, (hipStream_t)s, (u32x4*)y,
                           q.HS, q.WS, q.ROWS, q.PT);
        epilogue = 1;
        if (MT == 4) P8_LAUNCH(4, false, true);
        else P8_LAUNCH(2, false, true);
    } else if (MT == 4 && flat) P8_LAUNCH(4, true, false);
    else if (MT == 4) P8_LAUNCH(4, false, false);
    else if (flat) P8_LAUNCH(2, true, false);
    else P8_LAUNCH(2, false, false);
#undef P8_LAUNCH
    PTMI_LAUNCH_CHECK("p8_conv3x3");
    return 0;
}

int64_t ptmi_p8_wgrad_ws_floats_waves(int n, int cin, int cout, int h, int w, int waves)
{
    return (int64_t)p8_wgrad_splits(n, cin, cout, h, w, waves) * (9 * (int64_t)cout * cin + cout);
}

int64_t ptmi_p8_wgrad_ws_floats(int n, int cin, int cout, int h, int w)
{
    return ptmi_p8_wgrad_ws_floats_waves(n, cin, cout, h, w, 1);
}

int ptmi_p8_wgrad_fits(int n, int cin, int cout, int h, int w)
{
    if (n <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return 0;
    const P8Dims d = p8_dims(n, h, w);
    const int xcb = ptmi_p8_planes(cin), dcb = ptmi_p8_planes(cout);
    return (int64_t)xcb * d.PT * 16 < (1ll << 32) && (int64_t)dcb * d.PT * 16 < (1ll << 32) && d.PT < (1ll << 31) / 2;
}

int ptmi_p8_wgrad(const void* x, const void* dy, float* dw, float* db, float* ws, int n, int cin, int cout, int h, int w,
                  int accumulate, ptmi_stream_t s)
{
    return ptmi_p8_wgrad_waves(x, dy, dw, db, ws, n, cin, cout, h, w, accumulate, 1, s);
}

int ptmi_p8_wgrad_waves(const void* x, const void* dy, float* dw, float* db, float* ws, int n, int cin, int cout, int h, int w,
                        int accumulate, int waves, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && dy && dw && ws && n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "p8_wgrad: bad args");
    const P8Dims d = p8_dims(n, h, w);
    const int xcb = ptmi_p8_planes(cin), dcb = ptmi_p8_planes(cout);
    PTMI_CHECK_ARG(ptmi_p8_wgrad_fits(n, cin, cout, h, w),
                   "p8_wgrad: tensors beyond the 32-bit buffer offsets (n=%d cin=%d cout=%d h=%d w=%d; ptmi_p8_wgrad_fits)", n, cin, cout, h, w);
    const int S = p8_wgrad_splits(n, cin, cout, h, w, waves);
    const int coTiles = cdiv(cout, G_CO), ciTiles = cdiv(cin, G_CI), tilesC = cdiv(d.WS, 32);
    const int64_t nTiles = (int64_t)cdiv(d.ROWS, 4) * tilesC;
    PTMI_CHECK_ARG(nTiles < (1ll << 31), "p8_wgrad: too many tiles");
    float* wsb = ws + (size_t)S * 9 * cout * cin;
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(p8_wgrad_kernel, dim3((unsigned)(coTiles * ciTiles * S)), dim3(GT), 0, st, (const u16*)x, (const u16*)dy, ws, wsb, cin,
                       cout, xcb, dcb, d.WS, d.ROWS, (long long)d.PT, ciTiles, S, tilesC, (int)nTiles);
    PTMI_LAUNCH_CHECK("p8_wgrad");
    const int64_t cc = (int64_t)cout * cin;
    hipLaunchKernelGGL(p8_wgrad_reduce_kernel, dim3((unsigned)cdiv64(cc > cout ? cc : cout, 256)), dim3(256), 0, st, ws, wsb, dw, db, cout, cin, S,
                       accumulate);
    PTMI_LAUNCH_CHECK("p8_wgrad_reduce");
    return 0;
}

}  // extern "C"
