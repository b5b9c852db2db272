// wino4w.hip -- weight gradient of the 3x3 s1 p1 convolution through the Winograd F(4x4,3x3) transform domain of wino4.hip
// (round 5; replaces cuDNN's BWD_FILTER of the trainable layers the reference reaches at pt/modeling/backbone/vgg.py:45-53,66-69
// and pt/modeling/proposal_generator/rpn.py:96 -- the F(2x2,3x3)-domain kernel of wino.hip stays for shapes this one does not serve).
//
//   Y = A^T [U (.) V] A   =>   dU_p[co][ci] = sum over (image, tile) of W_p[co][tile] V_p[ci][tile],   W = A dY A^T  (6x6 from the
//   4x4 output-gradient tile),  V = B^T d B  (6x6 from the 6x6 input window),  dg = G^T dU G:  36 multiplies per 4x4 tile and
//   channel pair instead of 64 (F(2x2,3x3)) or 144 (direct).  Same points / constants as wino4.hip (0, +-3/4, +-3/2, inf).
//   Numerics: the tile sum averages the transforms' rounding -- error / max|dW| measured 2e-6, the direct fp32 kernel's own level
//   (tools/exp/wino4_numerics.py).
//
// GEMM per position: M = co, N = ci, K = tiles, on v_mfma_f32_32x32x2_f32 (K = 2 tiles per instruction).
//   * A workgroup (4 waves, one per SIMD) owns 64 co x 32 ci x 36 positions and a contiguous range of "chunks" -- (image, tile row,
//     16-column block) = 4 tiles = two k-steps -- of the split it belongs to.  A wave owns 32 co x 32 ci x 18 positions (three of
//     the six transform rows: waves 0 / 2 rows (0, +a, -a), waves 1 / 3 rows (+b, -b, inf)) = eighteen 32x32 accumulator tiles =
//     288 registers (256 AGPRs + 32 VGPRs: inline-asm MFMAs, as wino4.hip).  Positions are independent OUTPUTS here, so the split
//     costs nothing at the end -- and a wave transforms only ITS rows: vertical transform first (3 of 6 rows), horizontal after.
//   * operands: lane (c = lane & 31, t = lane >> 5) is channel c and tile 2 ks + t of the chunk for BOTH operands: it reads the 4x4
//     dY tile of its output channel (four 16-byte reads) and the 5 window rows its transform rows need of its input channel
//     (b32 + b128 + b32 each), and transforms them in registers: 40 + 72 FMAs per k-step of 18 MFMAs -- no cross-lane traffic, no
//     transform-domain tensor anywhere.  Plane pitches 68 / 148 floats (odd multiples of 4): the 16 lanes of a 16-byte read pass
//     fall on 16 distinct 16-byte bank groups.
//   * three LDS stages of 40 KB (64 dY planes of 4 x 16 pixels, 32 x planes of 6 x 24), `buffer_load_dwordx4 ... lds`, ten DMA
//     instructions per lane and chunk issued five per k-step, one and a half chunks ahead; ONE workgroup barrier per chunk behind
//     a counted vmcnt(5); interior chunks use the lane's loop-invariant offsets, border chunks recompute per-piece validity.
//   * partial dU go to the workspace [split][position 36][co][ci]; the bias gradient (sum of dY) is accumulated by the row-(0,+a,-a)
//     waves on every ciTiles-th chunk (each (co tile, ci tile) pair a different residue: together every chunk once) and appended
//     as [split][ci tile][co]; wino4_wgrad_reduce sums splits in a fixed order (deterministic) and applies G^T . G.

#include "common.h"
#include <type_traits>
#include <utility>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 zlds_f32x4_t;
typedef __attribute__((address_space(3))) float zlds_f32_t;
typedef __attribute__((address_space(3))) void zlds_void_t;

constexpr int ZC = 64, ZI = 32;          // output / input channels per workgroup
constexpr int ZNT = 256;
constexpr int ZDPC = 17, ZXPC = 37;      // 16-B pieces per dY plane (4 rows x 4 + pad) / x plane (6 rows x 6 + pad)
constexpr int ZDP = 4 * ZDPC, ZXP = 4 * ZXPC;   // plane pitches in floats: 68 / 148
constexpr int ZND = (ZC * ZDPC + 255) / 256;    // dY DMA instructions per lane and chunk (5)
constexpr int ZNX = (ZI * ZXPC + 255) / 256;    // x DMA instructions (5)
constexpr int ZNP = ZND + ZNX;                  // 10
constexpr int ZDR = ZND * 1024, ZXR = ZNX * 1024;   // floats of the two regions of a stage (whole instructions)
constexpr int ZSTAGE = ZDR + ZXR;               // 10240 floats = 40 KB; three stages

constexpr float ZA = 0.75f, ZB = 1.5f, ZA2 = 0.5625f, ZB2 = 2.25f, ZA2B2 = 1.265625f, ZS2 = 2.8125f;

__device__ __forceinline__ float zfma(float c, float x, float y) { return __builtin_fmaf(c, x, y); }
__device__ __forceinline__ float zfnma(float c, float x, float y) { return __builtin_fmaf(-c, x, y); }

// 1-D input transform t = B^T d, operation K of 12 (as wino4.hip: 0 .. 5 independent, 6 .. 11 depend only on them)
template <int K>
__device__ __forceinline__ void zin_op(const float (&d)[6], float (&t)[6], float (&E)[4])
{
    if constexpr (K == 0) t[0] = zfnma(ZS2, d[2], d[4]);
    if constexpr (K == 1) t[5] = zfnma(ZS2, d[3], d[5]);
    if constexpr (K == 2) E[0] = zfnma(ZB2, d[2], d[4]);
    if constexpr (K == 3) E[1] = zfnma(ZB2, d[1], d[3]);
    if constexpr (K == 4) E[2] = zfnma(ZA2, d[2], d[4]);
    if constexpr (K == 5) E[3] = zfnma(ZA2, d[1], d[3]);
    if constexpr (K == 6) t[0] = zfma(ZA2B2, d[0], t[0]);
    if constexpr (K == 7) t[5] = zfma(ZA2B2, d[1], t[5]);
    if constexpr (K == 8) t[1] = zfma(ZA, E[1], E[0]);
    if constexpr (K == 9) t[2] = zfnma(ZA, E[1], E[0]);
    if constexpr (K == 10) t[3] = zfma(ZB, E[3], E[2]);
    if constexpr (K == 11) t[4] = zfnma(ZB, E[3], E[2]);
}
// the three of its six outputs a wave needs (PH 0: points 0, +a, -a; PH 1: +b, -b, inf), operation K of 6; the PH-0 wave reads
// d[0 .. 4], the PH-1 wave d[1 .. 5]
template <int PH, int K>
__device__ __forceinline__ void zin_half_op(const float (&d)[6], float (&t)[3], float (&E)[2])
{
    if constexpr (PH == 0) {
        if constexpr (K == 0) t[0] = zfnma(ZS2, d[2], d[4]);
        if constexpr (K == 1) E[0] = zfnma(ZB2, d[2], d[4]);
        if constexpr (K == 2) E[1] = zfnma(ZB2, d[1], d[3]);
        if constexpr (K == 3) t[0] = zfma(ZA2B2, d[0], t[0]);
        if constexpr (K == 4) t[1] = zfma(ZA, E[1], E[0]);
        if constexpr (K == 5) t[2] = zfnma(ZA, E[1], E[0]);
    } else {
        if constexpr (K == 0) t[2] = zfnma(ZS2, d[3], d[5]);
        if constexpr (K == 1) E[0] = zfnma(ZA2, d[2], d[4]);
        if constexpr (K == 2) E[1] = zfnma(ZA2, d[1], d[3]);
        if constexpr (K == 3) t[2] = zfma(ZA2B2, d[1], t[2]);
        if constexpr (K == 4) t[0] = zfma(ZB, E[1], E[0]);
        if constexpr (K == 5) t[1] = zfnma(ZB, E[1], E[0]);
    }
}
// w = A y (6 from 4): w_i = sum_k p_i^k y_k = (y0 + p^2 y2) + p (y1 + p^2 y3); w_0 = y0, w_5 = y3.  The half a wave needs, operation K of 4
template <int PH, int K>
__device__ __forceinline__ void zout_half_op(const float (&y)[4], float (&w)[3], float (&E)[2])
{
    constexpr float P = PH ? ZB : ZA, P2 = PH ? ZB2 : ZA2;
    constexpr int lo = PH ? 0 : 1;                             // PH 0: w = (y0, +a, -a); PH 1: w = (+b, -b, y3)
    if constexpr (K == 0) E[0] = zfma(P2, y[2], y[0]);
    if constexpr (K == 1) E[1] = zfma(P2, y[3], y[1]);
    if constexpr (K == 2) w[lo] = zfma(P, E[1], E[0]);
    if constexpr (K == 3) {
        w[lo + 1] = zfnma(P, E[1], E[0]);
        w[PH ? 2 : 0] = PH ? y[3] : y[0];
    }
}
// all six, operation K of 8
template <int K>
__device__ __forceinline__ void zout_op(const float (&y)[4], float (&w)[6], float (&E)[4])
{
    if constexpr (K == 0) E[0] = zfma(ZA2, y[2], y[0]);
    if constexpr (K == 1) E[1] = zfma(ZA2, y[3], y[1]);
    if constexpr (K == 2) E[2] = zfma(ZB2, y[2], y[0]);
    if constexpr (K == 3) E[3] = zfma(ZB2, y[3], y[1]);
    if constexpr (K == 4) w[1] = zfma(ZA, E[1], E[0]);
    if constexpr (K == 5) w[2] = zfnma(ZA, E[1], E[0]);
    if constexpr (K == 6) w[3] = zfma(ZB, E[3], E[2]);
    if constexpr (K == 7) {
        w[4] = zfnma(ZB, E[3], E[2]);
        w[0] = y[0];
        w[5] = y[3];
    }
}

__device__ __forceinline__ void zmfma_a(f32x16& c, float a, float b) { asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void zmfma_v(f32x16& c, float a, float b) { asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

template <int... I, class F>
__device__ __forceinline__ void zfor(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// ---- the k-step schedule: 18 slots (MFMA q on the current operands + its share of the NEXT k-step's operands)
//   slots 0 .. 6: the 19 raw reads (three per slot: dY rows 0 .. 3, then the five window rows x three parts)
//   slot 7 of a chunk's first k-step: the hand-over for the next chunk
//   slots 8, 10, .., 16: one DMA instruction each
//   slots 2 .. 17: the 112 transform FMAs, seven per slot: 0 .. 39 W (vertical 16, horizontal 24), 40 .. 75 V vertical, 76 .. 111 V horizontal
//   (reads four per slot and the FMAs from slot 6 on -- five slots behind the reads they consume -- measured 2 % SLOWER: the reads' cost
//   is not latency)
constexpr int ZHAND = 7;
__host__ __device__ constexpr int z_dma_at(int s) { return (s >= 8 && s <= 16 && !(s & 1)) ? (s - 8) / 2 : -1; }
__host__ __device__ constexpr int z_valu_before(int s) { return s < 2 ? 0 : ((s - 2) * 7 > 112 ? 112 : (s - 2) * 7); }

template <int PH>
__device__ __forceinline__ void wino4_wgrad_body(
    float* lds, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, float* __restrict__ bpartial,
    int N, int Cin, int Cout, int H, int W, int ciTiles, int S, int tileRows, int colBlocks)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * W;
    int bid = blockIdx.x;
    const int split = bid % S; bid /= S;
    const int cit = bid % ciTiles, cot = bid / ciTiles;
    const int co0 = cot * ZC, ci0 = cit * ZI;
    // this split's contiguous range of the chunk list (image, tile row, column block; column block fastest)
    const int nChunksAll = N * tileRows * colBlocks;
    const int per = (nChunksAll + S - 1) / S;
    const int cBegin = split * per;
    const int nC = max(min(cBegin + per, nChunksAll) - cBegin, 0);

    // ---- per-lane DMA pieces relative to the chunk's origin: dY (y0, x0), x (y0 - 1, x0 - 4); index 0 .. ZND-1 dY, then x
    unsigned voff[ZNP];
    auto piece_coords = [&](int i, bool& chok, int& r, int& q4) __attribute__((always_inline)) {      // plane row / first column (relative) of piece i
        if (i < ZND) {
            const int pd = tid + i * ZNT;
            const int ch = pd / ZDPC, rem = pd - ch * ZDPC;
            r = rem >> 2; q4 = 4 * (rem & 3);
            chok = pd < ZC * ZDPC && rem < 16 && co0 + ch < Cout;
        } else {
            const int px = tid + (i - ZND) * ZNT;
            const int ch = px / ZXPC, rem = px - ch * ZXPC;
            r = rem / 6; q4 = 4 * (rem - r * 6) - 4;
            chok = px < ZI * ZXPC && rem < 36 && ci0 + ch < Cin;
        }
    };
#pragma unroll
    for (int i = 0; i < ZNP; ++i) {
        bool chok; int r, q4;
        piece_coords(i, chok, r, q4);
        const int ch = i < ZND ? (tid + i * ZNT) / ZDPC : (tid + (i - ZND) * ZNT) / ZXPC;
        voff[i] = chok ? (unsigned)(ch * HW + r * W + q4 + (i < ZND ? 0 : 4)) * 4u : 0xFFFFFFFFu;
    }
    const char* x_end = (const char*)(x + (size_t)N * Cin * HW);
    const char* dy_end = (const char*)(dy + (size_t)N * Cout * HW);
    auto clamp_rec = [](long long rem) __attribute__((always_inline)) { return (int)(rem > 0xFFFFFFFEll ? 0xFFFFFFFEll : (rem < 0 ? 0 : rem)); };

    // ---- the fetch in progress: descriptors, effective per-piece offsets, fix-up masks (current and previous set-up)
    int f_cb, f_ty, f_n;                                 // coordinates of the NEXT chunk to set up
    {
        const int g = min(cBegin, max(nChunksAll - 1, 0));
        f_cb = g % colBlocks;
        const int t = g / colBlocks;
        f_ty = t % tileRows;
        f_n = t / tileRows;
    }
    int f_left = nC;                                     // chunks of this split not yet set up
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(ptmi_uniform_ptr(x), 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = rx;
    unsigned eoff[ZNP];                                  // effective offsets of a chunk on the image border (else voff)
    bool f_plain = false;
    unsigned long long fix = 0, fix_prev = 0;            // words past the image edge, bits 4 i + e
    auto fetch_setup = [&]() __attribute__((always_inline)) {
        const bool any = f_left > 0;
        const int y0 = 4 * f_ty, x0 = f_cb * 16;
        const float* xb = x + ((size_t)f_n * Cin + ci0) * HW + ((ptrdiff_t)y0 - 1) * W + (x0 - 4);
        const float* db = dy + ((size_t)f_n * Cout + co0) * HW + (size_t)y0 * W + x0;
        rx = __builtin_amdgcn_make_buffer_rsrc(ptmi_uniform_ptr(xb), 0, clamp_rec(x_end - (const char*)xb), 0x00020000);
        rd = __builtin_amdgcn_make_buffer_rsrc(ptmi_uniform_ptr(db), 0, clamp_rec(dy_end - (const char*)db), 0x00020000);
        fix_prev = fix;
        // interior chunk: every row and column of both patches inside the image -- the loop-invariant offsets serve
        f_plain = any && x0 >= 4 && x0 + 20 <= W && y0 >= 1 && y0 + 5 <= H;
        fix = 0;
        if (!f_plain) {
#pragma unroll
            for (int i = 0; i < ZNP; ++i) {
                bool chok; int r, q4;
                piece_coords(i, chok, r, q4);
                const int gy = y0 + r - (i < ZND ? 0 : 1), gx = x0 + q4;
                const bool ok = any && chok && gy >= 0 && gy < H && gx >= 0 && gx < W;
                eoff[i] = ok ? voff[i] : 0xFFFFFFFFu;
                if (ok) {
#pragma unroll
                    for (int e = 1; e < 4; ++e) fix |= (gx + e >= W) ? (1ull << (4 * i + e)) : 0ull;
                }
            }
        }
        --f_left;
        if (++f_cb == colBlocks) {
            f_cb = 0;
            if (++f_ty == tileRows) { f_ty = 0; ++f_n; }
        }
    };
    auto fetch_piece = [&](int idx, int stage) __attribute__((always_inline)) {
        // (a scalar branch instead of a per-lane select: every VALU instruction in the loop costs MFMA time)
        zlds_void_t* dst = (zlds_void_t*)(lds + stage * ZSTAGE + (idx < ZND ? wave * 256 + idx * ZNT * 4 : ZDR + wave * 256 + (idx - ZND) * ZNT * 4));
        if (f_plain) __builtin_amdgcn_raw_ptr_buffer_load_lds(idx < ZND ? rd : rx, dst, 16, (int)voff[idx], 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(idx < ZND ? rd : rx, dst, 16, (int)eoff[idx], 0, 0, 0);
    };
    auto fixup = [&](int stage, unsigned long long fm) __attribute__((always_inline)) {
        if ((unsigned)fm | (unsigned)(fm >> 32)) {
#pragma unroll
            for (int i = 0; i < ZNP; ++i) {
                float* pc = lds + stage * ZSTAGE + (i < ZND ? (tid + i * ZNT) * 4 : ZDR + (tid + (i - ZND) * ZNT) * 4);
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (fm & (1ull << (4 * i + e))) pc[e] = 0.f;
            }
        }
    };

    // ---- the lane's role in the MFMAs: v_mfma_f32_32x32x2_f32 takes A[m = lane & 31][k = lane >> 5], B[k][n = lane & 31]
    const int chh = wave >> 1;                               // co half of the workgroup's 64
    const int c = lane & 31, t = lane >> 5;
    const int a_off = (chh * 32 + c) * ZDP + 4 * t;                                          // + 8 ks, + 16 row
    const int b_off = ZDR + c * ZXP + PH * 24 + 4 * t + 3;                                   // + 8 ks, + 24 row (of the wave's five), + {0, 1 (b128), 5}

    f32x16 accA[16];                        // positions (of the wave's 18) 0 .. 15: AGPRs
    f32x16 accV[2];                         // 16, 17: VGPRs
    zfor(std::make_integer_sequence<int, 16>{}, [&](auto i_c) __attribute__((always_inline)) {
        accA[decltype(i_c)::value] = (f32x16){0};
        asm volatile("" : "+a"(accA[decltype(i_c)::value]));
    });
    zfor(std::make_integer_sequence<int, 2>{}, [&](auto i_c) __attribute__((always_inline)) {
        accV[decltype(i_c)::value] = (f32x16){0};
        asm volatile("" : "+v"(accV[decltype(i_c)::value]));
    });
    float WA[2][18], VB[2][18];             // the operands of two consecutive k-steps: position q = 6 (row of the wave's three) + column
    float YR[4][4];                         // raw dY tile: [row][column]
    float D[6][6];                          // raw window rows (PH 0: 0 .. 4, PH 1: 1 .. 5), [row][column]
    float U3[3][4];                         // A dY: the wave's three rows x four columns
    float T3[3][6];                         // B^T d: the wave's three rows x six columns
    float E[4];
    float bsum = 0.f;                       // sum of this lane's dY tiles (PH-0 waves, on the chunks of this workgroup's residue)

    auto raw_read = [&](auto k_c, const float* src, int ks8) __attribute__((always_inline)) {    // read k of 19
        constexpr int k = decltype(k_c)::value;
        if constexpr (k < 4) {
            const f32x4 v = *(const volatile zlds_f32x4_t*)(src + a_off + ks8 + 16 * k);
            YR[k][0] = v[0]; YR[k][1] = v[1]; YR[k][2] = v[2]; YR[k][3] = v[3];
        } else {
            constexpr int row = (k - 4) / 3 + PH, part = (k - 4) % 3;
            const float* p = src + b_off + ks8 + 24 * ((k - 4) / 3);
            if constexpr (part == 0) D[row][0] = *(const volatile zlds_f32_t*)p;
            if constexpr (part == 1) {
                const f32x4 v = *(const volatile zlds_f32x4_t*)(p + 1);
                D[row][1] = v[0]; D[row][2] = v[1]; D[row][3] = v[2]; D[row][4] = v[3];
            }
            if constexpr (part == 2) D[row][5] = *(const volatile zlds_f32_t*)(p + 5);
        }
    };
    auto valu_op = [&](auto k_c, float (&wa)[18], float (&vb)[18]) __attribute__((always_inline)) {   // transform FMA k of 112
        constexpr int k = decltype(k_c)::value;
        if constexpr (k < 16) {                               // A dY, column k / 4: rows (y0 .. y3) -> the wave's three
            constexpr int col = k / 4;
            const float y[4] = {YR[0][col], YR[1][col], YR[2][col], YR[3][col]};
            float w[3] = {U3[0][col], U3[1][col], U3[2][col]};
            float e2[2] = {E[0], E[1]};
            zout_half_op<PH, k % 4>(y, w, e2);
            U3[0][col] = w[0]; U3[1][col] = w[1]; U3[2][col] = w[2]; E[0] = e2[0]; E[1] = e2[1];
        } else if constexpr (k < 40) {                        // (A dY) A^T, row (k - 16) / 8: four values -> six
            constexpr int row = (k - 16) / 8;
            float w[6] = {wa[6 * row], wa[6 * row + 1], wa[6 * row + 2], wa[6 * row + 3], wa[6 * row + 4], wa[6 * row + 5]};
            zout_op<(k - 16) % 8>(U3[row], w, E);
            wa[6 * row] = w[0]; wa[6 * row + 1] = w[1]; wa[6 * row + 2] = w[2]; wa[6 * row + 3] = w[3]; wa[6 * row + 4] = w[4]; wa[6 * row + 5] = w[5];
        } else if constexpr (k < 76) {                        // B^T d, column (k - 40) / 6: the wave's three rows
            constexpr int col = (k - 40) / 6;
            const float d[6] = {D[0][col], D[1][col], D[2][col], D[3][col], D[4][col], D[5][col]};
            float tt[3] = {T3[0][col], T3[1][col], T3[2][col]};
            float e2[2] = {E[0], E[1]};
            zin_half_op<PH, (k - 40) % 6>(d, tt, e2);
            T3[0][col] = tt[0]; T3[1][col] = tt[1]; T3[2][col] = tt[2]; E[0] = e2[0]; E[1] = e2[1];
        } else {                                              // (B^T d) B, row (k - 76) / 12
            constexpr int row = (k - 76) / 12;
            float v[6] = {vb[6 * row], vb[6 * row + 1], vb[6 * row + 2], vb[6 * row + 3], vb[6 * row + 4], vb[6 * row + 5]};
            zin_op<(k - 76) % 12>(T3[row], v, E);
            vb[6 * row] = v[0]; vb[6 * row + 1] = v[1]; vb[6 * row + 2] = v[2]; vb[6 * row + 3] = v[3]; vb[6 * row + 4] = v[4]; vb[6 * row + 5] = v[5];
        }
    };

    if (nC > 0) {
        // ---- start: chunks 0 and 1 entirely, the first half of chunk 2
        fetch_setup();
#pragma unroll
        for (int idx = 0; idx < ZNP; ++idx) fetch_piece(idx, 0);
        fetch_setup();
#pragma unroll
        for (int idx = 0; idx < ZNP; ++idx) fetch_piece(idx, 1);
        const unsigned long long fix0 = fix_prev;
        fetch_setup();
#pragma unroll
        for (int idx = 0; idx < 5; ++idx) fetch_piece(idx, 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ZNP + 5) : "memory");
        fixup(0, fix0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // operands of (chunk 0, k-step 0)
        zfor(std::make_integer_sequence<int, 19>{}, [&](auto k_c) __attribute__((always_inline)) { raw_read(k_c, lds, 0); });
        zfor(std::make_integer_sequence<int, 112>{}, [&](auto k_c) __attribute__((always_inline)) { valu_op(k_c, WA[0], VB[0]); });

        int s_cur = 0, s_nxt = 1, s_dma = 2;                  // stages of chunk c / c + 1 / the chunk being fetched
        int chunk_mod = cBegin % ciTiles;                     // (chunk index) mod ciTiles: the bias gradient's residue test
        // k-step KS of the chunk in stage s_cur: MFMAs on operand set KS; raw reads + transforms of the NEXT k-step (KS 0: this chunk's
        // second, from s_cur; KS 1: the next chunk's first, from s_nxt) into set KS ^ 1; DMA: KS 0 pieces 5 .. 9 of chunk c + 2 (set up in
        // the previous k-step), KS 1 set-up + pieces 0 .. 4 of chunk c + 3
        auto kstep = [&](auto ks_c) __attribute__((always_inline)) {
            constexpr int KS = decltype(ks_c)::value;
            const float* src = lds + (KS == 0 ? s_cur : s_nxt) * ZSTAGE;
            constexpr int ks8 = KS == 0 ? 8 : 0;               // the next k-step's tiles: 2, 3 of this chunk / 0, 1 of the next
            if constexpr (KS == 1) {
                { const int tt = s_cur; s_cur = s_nxt; s_nxt = s_dma; s_dma = tt; }       // (after this k-step s_cur is the next chunk)
                fetch_setup();
            }
            const int st_dma = s_dma;                          // KS 0: the stage set up one k-step ago; KS 1: the one just rotated in
            {   // one address register for the k-step's raw reads
                unsigned va = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)src;
                asm volatile("" : "+v"(va));
                src = (const float*)(const __attribute__((address_space(3))) float*)(size_t)va;
            }
            zfor(std::make_integer_sequence<int, 18>{}, [&](auto q_c) __attribute__((always_inline)) {
                constexpr int Q = decltype(q_c)::value;
                if constexpr (Q < 16) zmfma_a(accA[Q], WA[KS][Q], VB[KS][Q]);   // [x4:mf]
                else zmfma_v(accV[Q - 16], WA[KS][Q], VB[KS][Q]);   // [x4:mf]
                if constexpr (Q <= 6) {
                    zfor(std::make_integer_sequence<int, (Q == 6 ? 1 : 3)>{}, [&](auto r_c) __attribute__((always_inline)) {
                        raw_read(std::integral_constant<int, 3 * Q + decltype(r_c)::value>{}, src, ks8);   // [x4:rd]
                    });
                }
                if constexpr (Q == ZHAND && KS == 0) {
                    // hand-over: everything but the five newest DMA instructions (the first half of chunk c + 2) has landed, i.e.
                    // chunk c + 1; fix-ups; barrier -- then stage s_cur is free (its last raw reads were this k-step's)
                    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // [x4:ho]
                    fixup(s_nxt, fix_prev);   // [x4:ho]
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // [x4:ho]
                }
                constexpr int di = z_dma_at(Q);
                if constexpr (di >= 0) fetch_piece(KS == 0 ? 5 + di : di, st_dma);   // [x4:dma]
                constexpr int k0 = z_valu_before(Q), k1 = z_valu_before(Q + 1);
                zfor(std::make_integer_sequence<int, k1 - k0>{}, [&](auto k_c) __attribute__((always_inline)) {
                    valu_op(std::integral_constant<int, k0 + decltype(k_c)::value>{}, WA[KS ^ 1], VB[KS ^ 1]);   // [x4:xf]
                });
                if constexpr (PH == 0 && Q == 8) {             // bias gradient: the raw dY tile of the NEXT k-step (read in slots 0, 1)
                    // of chunk (KS 0: this one, KS 1: the next one); every (co tile, ci tile) pair takes the chunks of one residue
                    const int cm = KS == 0 ? chunk_mod : (chunk_mod + 1 == ciTiles ? 0 : chunk_mod + 1);
                    if (cm == cit) {
                        float s0 = (YR[0][0] + YR[0][1]) + (YR[0][2] + YR[0][3]), s1 = (YR[1][0] + YR[1][1]) + (YR[1][2] + YR[1][3]);
                        float s2 = (YR[2][0] + YR[2][1]) + (YR[2][2] + YR[2][3]), s3 = (YR[3][0] + YR[3][1]) + (YR[3][2] + YR[3][3]);
                        bsum += (s0 + s1) + (s2 + s3);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (KS == 1) chunk_mod = chunk_mod + 1 == ciTiles ? 0 : chunk_mod + 1;
        };
        if (cit == chunk_mod % ciTiles) {                      // the very first k-step's tile (read in the prologue)
            if constexpr (PH == 0) {
                float s0 = (YR[0][0] + YR[0][1]) + (YR[0][2] + YR[0][3]), s1 = (YR[1][0] + YR[1][1]) + (YR[1][2] + YR[1][3]);
                float s2 = (YR[2][0] + YR[2][1]) + (YR[2][2] + YR[2][3]), s3 = (YR[3][0] + YR[3][1]) + (YR[3][2] + YR[3][3]);
                bsum += (s0 + s1) + (s2 + s3);
            }
        }
        for (int chunk = 0; chunk < nC; ++chunk) {
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last (empty) fetches must have landed before the workgroup gives up its LDS
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (inline asm: the compiler pads nothing)
    // ---- partial db: [split][ci tile][co]
    if constexpr (PH == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (t == 0 && co0 + chh * 32 + c < Cout) bpartial[((size_t)split * ciTiles + cit) * Cout + co0 + chh * 32 + c] = bsum;
    }
    // ---- partial dU: [split][position][co][ci]; accumulator element r of a lane: co row (r & 3) + 8 (r >> 2) + 4 t, ci column c
    auto rdacc = [](float a) __attribute__((always_inline)) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; };
    const int ci = ci0 + c;
    zfor(std::make_integer_sequence<int, 18>{}, [&](auto q_c) __attribute__((always_inline)) {
        constexpr int Q = decltype(q_c)::value;
        constexpr int p = 6 * (3 * PH + Q / 6) + Q % 6;
        float* dst = partial + ((size_t)split * 36 + p) * Cout * Cin;
        zfor(std::make_integer_sequence<int, 16>{}, [&](auto r_c) __attribute__((always_inline)) {
            constexpr int r = decltype(r_c)::value;
            const int co = co0 + chh * 32 + (r & 3) + 8 * (r >> 2) + 4 * t;
            float v;
            if constexpr (Q < 16) v = rdacc(accA[Q][r]);
            else v = accV[Q - 16][r];
            if (co < Cout && ci < Cin) dst[(size_t)co * Cin + ci] = v;
        });
    });
}

__global__ __launch_bounds__(ZNT, 1) void conv3x3_wino4_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, float* __restrict__ bpartial,
    int N, int Cin, int Cout, int H, int W, int ciTiles, int S, int tileRows, int colBlocks)
{
    __shared__ __attribute__((aligned(16))) float lds[3 * ZSTAGE];
    // (wave-uniform) the transform rows of this wave: two specialisations of the whole body -- the halves differ in constants and in
    // which raw rows they read, and a per-lane select in the loop would cost MFMA time
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1)
        wino4_wgrad_body<1>(lds, x, dy, partial, bpartial, N, Cin, Cout, H, W, ciTiles, S, tileRows, colBlocks);
    else
        wino4_wgrad_body<0>(lds, x, dy, partial, bpartial, N, Cin, Cout, H, W, ciTiles, S, tileRows, colBlocks);
}

// dW[co][ci][3][3] (+)= G^T ( sum_splits dU[split] ) G;  db[co] (+)= sum over splits and ci tiles of the partial sums
__global__ void wino4_wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial,
                                          float* __restrict__ dw, float* __restrict__ db, int Cout, int Cin, int S, int ciTiles, int accumulate)
{
    const float G[6][3] = {{64.f / 81.f, 0.f, 0.f},
                           {-128.f / 243.f, -32.f / 81.f, -8.f / 27.f},
                           {-128.f / 243.f, 32.f / 81.f, -8.f / 27.f},
                           {32.f / 243.f, 16.f / 81.f, 8.f / 27.f},
                           {32.f / 243.f, -16.f / 81.f, 8.f / 27.f},
                           {0.f, 0.f, 1.f}};
    const int64_t cc = (int64_t)Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cc; i += (int64_t)gridDim.x * blockDim.x) {
        if (db && i < Cout) {
            float sum = 0.f;
            for (int s = 0; s < S * ciTiles; ++s) sum += bpartial[(size_t)s * Cout + i];
            db[i] = accumulate ? db[i] + sum : sum;
        }
        float tq[3][6];                                      // G^T u: [ky][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) tq[0][j] = tq[1][j] = tq[2][j] = 0.f;
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            float sum = 0.f;
            for (int s = 0; s < S; ++s) sum += partial[((size_t)s * 36 + p) * cc + i];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) tq[ky][p % 6] += G[p / 6][ky] * sum;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float g = 0.f;
#pragma unroll
                for (int j = 0; j < 6; ++j) g += tq[ky][j] * G[j][kx];
                float* o = dw + i * 9 + ky * 3 + kx;
                *o = accumulate ? *o + g : g;
            }
        }
    }
}

// splits per (co tile, ci tile): W4W_WAVES_OF_WORKGROUPS x the chip's 256 one-workgroup-per-CU slots.  1 = one long workgroup per
// CU (round 5); more = the hardware dispatcher hands the later workgroups to whichever CU frees up first -- the non-persistent
// form of work stealing: CUs held by another kernel when the launch starts (a collective overlapping backward) then cost their
// share instead of a second pass (tools/exp/contention.py), for one more prologue / partial-sum store per extra workgroup
#ifndef W4W_WAVES_OF_WORKGROUPS
#define W4W_WAVES_OF_WORKGROUPS 1
#endif
static int wino4_wgrad_splits(int n, int cin, int cout, int h, int w, int waves)
{
    const int pairs = cdiv(cout, ZC) * cdiv(cin, ZI);
    const int64_t chunks = (int64_t)n * cdiv(h, 4) * cdiv(w, 16);
    int S = cdiv(256 * W4W_WAVES_OF_WORKGROUPS * (waves < 1 ? 1 : waves > 16 ? 16 : waves), pairs);
    if (S > chunks) S = (int)chunks;
    return S < 1 ? 1 : S;
}

}  // namespace

extern "C" {

int ptmi_conv3x3_wino4_wgrad_fits(int h, int w)
{
    return h > 0 && w > 0 && (int64_t)(ZC + 1) * h * w * 4 < (1ll << 31);
}

int64_t ptmi_conv3x3_wino4_wgrad_ws_floats_waves(int n, int cin, int cout, int h, int w, int waves)
{
    return (int64_t)wino4_wgrad_splits(n, cin, cout, h, w, waves) * (36 * (int64_t)cout * cin + (int64_t)cdiv(cin, ZI) * cout);
}

int64_t ptmi_conv3x3_wino4_wgrad_ws_floats(int n, int cin, int cout, int h, int w)
{
    return ptmi_conv3x3_wino4_wgrad_ws_floats_waves(n, cin, cout, h, w, 1);
}

int ptmi_conv3x3_wino4_wgrad(const float* x, const float* dy, float* dw, float* db, float* ws, int n, int cin, int cout, int h,
                             int w, int accumulate, ptmi_stream_t s)
{
    return ptmi_conv3x3_wino4_wgrad_waves(x, dy, dw, db, ws, n, cin, cout, h, w, accumulate, 1, s);
}

int ptmi_conv3x3_wino4_wgrad_waves(const float* x, const float* dy, float* dw, float* db, float* ws, int n, int cin, int cout, int h,
                                   int w, int accumulate, int waves, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && dy && dw && ws && n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "conv3x3_wino4_wgrad: bad args");
    PTMI_CHECK_ARG(ptmi_conv3x3_wino4_wgrad_fits(h, w), "conv3x3_wino4_wgrad: map %dx%d too large for 32-bit buffer offsets", h, w);
    const int S = wino4_wgrad_splits(n, cin, cout, h, w, waves);
    const int coTiles = cdiv(cout, ZC), ciTiles = cdiv(cin, ZI);
    hipStream_t st = (hipStream_t)s;
    float* bws = ws + (size_t)S * 36 * cout * cin;
    hipLaunchKernelGGL(conv3x3_wino4_wgrad_kernel, dim3((unsigned)(coTiles * ciTiles * S)), dim3(ZNT), 0, st, x, dy, ws, bws, n, cin, cout,
                       h, w, ciTiles, S, cdiv(h, 4), cdiv(w, 16));
    PTMI_LAUNCH_CHECK("conv3x3_wino4_wgrad");
    const int64_t cc = (int64_t)cout * cin;
    hipLaunchKernelGGL(wino4_wgrad_reduce_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, st, ws, bws, dw, db, cout, cin, S, ciTiles,
                       accumulate);
    PTMI_LAUNCH_CHECK("conv3x3_wino4_wgrad_reduce");
    return 0;
}

}  // extern "C"
