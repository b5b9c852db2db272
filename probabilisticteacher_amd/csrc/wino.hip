// wino.hip -- fused Winograd F(2x2, 3x3) convolution for gfx950 (CDNA4): 3x3 s1 p1, fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32: forward and dgrad (conv3x3_wino_kernel) and the weight gradient (conv3x3_wino_wgrad_kernel, second
// half of the file) of the 32..512-channel layers.
//
// Replaces the cuDNN conv2d (+bias, +ReLU) the reference reaches at pt/modeling/backbone/vgg.py:45-53,66-69 (conv1_2 ..
// conv5_3) and the 3x3 conv of D2's StandardRPNHead (pt/modeling/proposal_generator/rpn.py:96) -- cuDNN itself runs these
// fp32 3x3 s1 layers as Winograd.  The direct implicit-GEMM kernel of conv.hip sits at 0.88 of the fp32 MFMA peak; the
// only way past that roof is fewer multiplies: F(2x2,3x3) needs 16 instead of 36 per 2x2 output tile and channel pair.
//
// One kernel, nothing in HBM but x, the pre-transformed weights and y:
//   * transform domain: for each of the 16 positions p = (i, j) of the 4x4 tile, M_p[co][tile] = sum_ci U_p[co][ci] V_p[ci][tile]
//     with U = G g G^T (packed once per call by wino_pack_weights_kernel), V = B^T d B (computed per lane from LDS) and
//     Y = A^T M A (epilogue, in registers);
//   * a workgroup = 4 wave64s owns 64 output channels x (8 rows x 32 columns) = 64 tiles of the FLAT tile line (every (image,
//     8-row band) is a strip of `period` columns; the strips are concatenated; a workgroup may straddle strips); a wave owns
//     32 channels x 32 tiles (16 tile columns x 2 tile rows) x 16 positions = sixteen 32x32 accumulator tiles = 256
//     accumulator registers: ONE wave per SIMD, the whole accumulator file of it;
//   * v_mfma_f32_32x32x2_f32 takes A[co = lane & 31][k = lane >> 5] and B[k = lane >> 5][tile = lane & 31]: a lane IS one
//     (tile, channel) pair, so it reads its own 4x4 input window from the LDS patch (eight 4-byte-aligned pairs: ds_read2_b32),
//     applies B^T d B with 16 packed additions and holds the B operands of all 16 positions' MFMAs -- no cross-lane movement,
//     no transform-domain tensor anywhere.  The A operands of the 16 positions are 64 contiguous bytes per (channel, ci) in
//     the packed slab (four ds_read_b128);
//   * K is walked in 8-channel chunks (four k-steps of 2 channels = 64 MFMAs = 4096 matrix-pipe cycles per wave and chunk)
//     through THREE LDS stages; operands arrive with `buffer_load_dwordx4 ... lds` (per-lane offsets are loop invariants;
//     halo rows / columns and padded channels carry offset 0xFFFFFFFF and are zero-filled by the buffer range check), ONE
//     workgroup barrier per chunk behind a counted vmcnt, placed inside the last k-step's MFMA stream whose operands are
//     already in registers;
//   * the next k-step's LDS reads and input transform are issued between the current k-step's MFMAs (software pipeline in
//     source order, pinned with sched_barrier): with one wave per SIMD nothing else hides them -- and VALU time adds to fp32
//     MFMA time, so every instruction in the loop is paid for;
//   * epilogue: inverse transform in registers (the 16 positions of a (channel, tile) pair live in the same lane and register
//     index of the 16 accumulator tiles; explicit v_accvgpr_read, two channel rows per packed addition), then the epilogues of
//     conv.hip: bias / bias+ReLU / none / ReLU mask of the producer (dgrad) / bias+ReLU+2x2 max pool (a Winograd tile IS a
//     pool window).
// dgrad = the same kernel on dY with the flipped / transposed filter (pack mode 1), as in conv.hip.

#include "common.h"
#include <type_traits>

// channel tiles of a pixel tile run back to back on ONE XCD when there are at most this many of them (their U slabs then fit
// that XCD's 4 MB L2)
#define WINO_COLOCATE_MAX_COTILES 4

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2 wlds_f32x2_t;
typedef __attribute__((address_space(3))) f32x4 wlds_f32x4_t;
typedef __attribute__((address_space(3))) void wlds_void_t;

constexpr int WKC = 8;                 // input channels per chunk
constexpr int WBM = 64;                // output channels per workgroup
constexpr int WTH = 8, WTW = 32;       // output pixels per workgroup: 8 rows x 32 (flat) columns = 4 x 16 Winograd tiles
constexpr int WPP = 48;                // patch row pitch in floats: 10 loaded 16-B pieces (flat columns u0-4 .. u0+35) + 2 pad
                                       // pieces; 2 rows = 96 floats = 32 banks (mod 64): the two tile rows of a wave read
                                       // disjoint bank halves with ds_read_b64
constexpr int WPR = WTH + 2;           // patch rows (image rows y0-1 .. y0+8)
constexpr int WPL = WPR * WPP;         // floats per channel plane (480)
constexpr int WUS = WKC * 4 * WBM * 4; // U floats per chunk: [ci 8][position row i 4][co 64][position column j 4] = 8192
constexpr int WPS = WKC * WPL;         // patch floats per chunk (3840 = 960 pieces = 15 waves' worth)
constexpr int WPSP = 4096;             // ... padded to 16 waves' worth: every wave issues the same four patch DMAs (the counted
                                       // vmcnt waits rely on it; pieces 960 .. 1023 carry offset 0xFFFFFFFF)
constexpr int WSTAGE = WUS + WPSP;     // 12288 floats = 48 KB; three stages
constexpr int WNT = 256;
constexpr int WUI = WUS / 4 / WNT;     // U DMA instructions per lane and chunk (8)
constexpr int WPI = (WPS / 4 + WNT - 1) / WNT;   // patch DMA instructions per lane and chunk (4; the last one waves 0-2 only)

__device__ __forceinline__ void wino_vmwait0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// packed fp32 additions on register pairs (VOP3P): the transforms cost half the VALU issue slots of scalar adds -- and VALU
// time adds to fp32 MFMA time on this chip (measured: removing the transforms shortens the kernel by exactly their issue time)
typedef __attribute__((address_space(3), aligned(4))) f32x2 wlds_f32x2_a4_t;     // 4-byte aligned pair: ds_read2_b32
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// (a.lo + b.hi, a.lo - b.hi)
__device__ __forceinline__ f32x2 pk_lo_pm_hi(f32x2 a, f32x2 b) { f32x2 r; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }

// V = B^T d B for the lane's (tile, channel): window row a arrives as two 4-byte-aligned pairs (ds_read2_b32), D[2a + c] =
// window columns (2c, 2c + 1) = LDS columns 2 ttx + 3 + 2c, + 1.  Output pairs: V[2i] = (v_i0, v_i3), V[2i + 1] = (v_i1, v_i2).
__device__ __forceinline__ void wino_xform_rows(const f32x2 (&D)[8], f32x2 (&T)[4][2], int c)
{
    T[0][c] = pk_sub(D[0 + c], D[4 + c]);
    T[1][c] = pk_add(D[2 + c], D[4 + c]);
    T[2][c] = pk_sub(D[4 + c], D[2 + c]);
    T[3][c] = pk_sub(D[2 + c], D[6 + c]);
}
__device__ __forceinline__ void wino_xform_col(const f32x2 (&T)[4][2], f32x2 (&V)[8], int i)
{
    V[2 * i] = pk_sub(T[i][0], T[i][1]);                  // (t0 - t2, t1 - t3)
    V[2 * i + 1] = pk_lo_pm_hi(T[i][1], T[i][0]);         // (t2 + t1, t2 - t1)
}
__device__ __forceinline__ void wino_xform(const f32x2 (&D)[8], f32x2 (&V)[8])
{
    f32x2 T[4][2];
    wino_xform_rows(D, T, 0);
    wino_xform_rows(D, T, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) wino_xform_col(T, V, i);
}
// operand of position p = 4 i + j from the pairs above
__device__ __forceinline__ float wino_vop(const f32x2 (&V)[8], int p)
{
    const int i = p >> 2, j = p & 3;
    return (j == 0 || j == 3) ? V[2 * i][j == 3] : V[2 * i + 1][j == 2];
}

// RAGGED: Cin is not a multiple of the 8-channel chunk (tests and odd layers only: the production instantiation carries none of
// that bookkeeping -- it cost 1 % of the step in scalar registers and loop instructions)
template <bool RAGGED>
__global__ __launch_bounds__(WNT, 1) void conv3x3_wino_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ mref, float* __restrict__ y, int N, int Cin, int Cout, int H, int W, int nChunks, int epi,
    int coTiles, int bands, int period, int nPix, int colocate)
{
    __shared__ __attribute__((aligned(16))) float lds[3 * WSTAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (channel tile, pixel tile).  Workgroups go to XCDs round robin (linear id mod 8), each XCD has its own
    // 4 MB L2, and a pixel tile's patch is wanted by every channel tile:
    //   colocate = 0 (8 channel tiles, 512 output channels): id = pixel tile * coTiles + channel tile -- every XCD keeps ONE
    //       channel tile's U slabs (2 MB at Cin = 512) in its L2 and each patch is fetched by all eight L2s;
    //   colocate = 1 (<= 4 channel tiles: all their U slabs fit one L2): the channel tiles of a pixel tile run back to back
    //       on ONE XCD (id mod 8 = pixel tile mod 8): the patch leaves HBM / MALL once instead of coTiles times.
    // Pixel tiles are FLAT: every (image, 8-row band) is a strip of `period` columns (W rounded up to a multiple of 4 with at
    // least one zero column: the strips' shared halo), the strips are concatenated, and workgroup `pix` owns tile columns
    // 16 pix .. 16 pix + 15 of that line -- it may straddle strips (bands, images).  Padding is paid once per launch instead
    // of once per row of workgroups: a 32-column grid cost 14 % of the MFMAs on W = 166 / 83 and 5 % on W = 333.
    int cot, pix;
    if (colocate) {
        const int slot = blockIdx.x >> 3;
        cot = slot % coTiles;
        pix = (slot / coTiles) * 8 + (blockIdx.x & 7);
        if (pix >= nPix) return;
    } else {
        cot = blockIdx.x % coTiles;
        pix = blockIdx.x / coTiles;
    }
    const int HW = H * W;
    const int nStrips = N * bands;
    const int u0 = pix * WTW;                                // flat column of the workgroup's first output column
    const int s_first = u0 / period;                         // strip of the first tile column; its image is the address base
    const int n = s_first / bands;
    // (scalar) strip / image / band of the patch's first column: per-lane positions are reached from it by short walks instead of
    // two vector integer divisions per piece and tile -- 300 VALU instructions per workgroup, 2 % of a 64-channel tile
    const int s0 = (u0 > 4 ? u0 - 4 : 0) / period;
    const int n0 = s0 / bands, b0 = s0 - n0 * bands;

    // ---- DMA descriptors: this lane's patch pieces (channel, patch row, 16-B piece) -> byte offset from the chunk's first plane
    // of image n.  LDS column c of the patch <-> flat column u0 - 4 + c; periods are multiples of 4, so a piece lies in ONE strip.
    unsigned pvoff[WPI];
    int fix = 0;                                             // words 1..3 of piece i (bits 4i+1 .. 4i+3) beyond the image edge
#pragma unroll
    for (int i = 0; i < WPI; ++i) {
        const int pidx = tid + i * WNT;
        const int ci = pidx / (WPR * 12), rem = pidx - ci * (WPR * 12);
        const int r = rem / 12, q = rem - r * 12;
        int gx = u0 - 4 + 4 * q - s0 * period, band = b0, sn = n0;       // gx < 0 only in the very first patch (u0 = 0, q = 0)
        while (gx >= period) { gx -= period; ++band; }
        while (band >= bands) { band -= bands; ++sn; }
        const int gy = band * WTH - 1 + r;
        pvoff[i] = 0xFFFFFFFFu;
        if (pidx < WPS / 4 && q < 10 && gx >= 0 && sn < N && gy >= 0 && gy < H && gx < W) {
            pvoff[i] = (unsigned)(((sn - n) * Cin + ci) * HW + gy * W + gx) * 4u;
#pragma unroll
            for (int e = 1; e < 4; ++e) fix |= (gx + e >= W) ? (1 << (4 * i + e)) : 0;
        }
    }
    const unsigned wvoff = (unsigned)tid * 16u;
    // (workgroup-uniform, computed on the scalar unit -- a __syncthreads_or would add a second LDS object, after which the
    // compiler no longer tells the DMA's LDS writes from the operand reads and puts a vmcnt(0) in front of every read: 2x slower)
    bool edge = false;                                       // some loaded piece may straddle the right edge of an image row
    if (W & 3) {
        const int e0 = W & ~3;                               // strip-local first column of the straddling piece
        for (int st = s0; st < nStrips && st * period + e0 < u0 + 36; ++st)
            edge |= st * period + e0 >= u0 - 4;
    }

    const char* xc = (const char*)(x + (size_t)n * Cin * HW);
    const char* wc = (const char*)(wp + (size_t)cot * nChunks * WUS);
    // bytes from xc to the end of the tensor, clamped (offsets reach into the next image: (Cin + 8) HW 4 < 2^32, launcher)
    const long long xtail = (long long)(N - n) * Cin * HW * 4;
    unsigned xleft = (unsigned)(xtail > 0xFFFFFFFEll ? 0xFFFFFFFEll : xtail);

    // one DMA instruction of a chunk: idx 0 .. WPI-1 = patch pieces, WPI .. WPI+WUI-1 = U pieces
    auto dma_piece = [&](int idx, int buf) {
        if (idx < WPI) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
                ptmi_uniform_ptr(xc), 0, __builtin_amdgcn_readfirstlane((int)xleft), 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (wlds_void_t*)(lds + buf * WSTAGE + WUS + wave * 256 + idx * WNT * 4),
                                                     16, (int)pvoff[idx], 0, 0, 0);
        } else {
            const int i = idx - WPI;
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(ptmi_uniform_ptr(wc), 0, WUS * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (wlds_void_t*)(lds + buf * WSTAGE + wave * 256 + i * WNT * 4), 16,
                                                     (int)wvoff, i * WNT * 16, 0, 0);
        }
    };
    // Cin not a multiple of 8: the last chunk's channels >= Cin are the NEXT image's first planes (the descriptor reaches to the
    // end of the tensor) -- their weights are packed as zeros, but 0 x Inf is not 0: those pieces are switched off for the last chunk
    int chunks_left = nChunks;                               // chunks not yet issued (RAGGED only)
    auto mask_ragged_channels = [&]() {
        if constexpr (RAGGED) {
            if (chunks_left == 1) {
#pragma unroll
                for (int i = 0; i < WPI; ++i)
                    if ((tid + i * WNT) / (WPR * 12) >= (Cin & (WKC - 1))) pvoff[i] = 0xFFFFFFFFu;
            }
        }
    };
    mask_ragged_channels();
    auto advance = [&]() {
        wc += WUS * 4;
        xc += (size_t)WKC * HW * 4;
        xleft = xleft == 0xFFFFFFFEu ? xleft : xleft - (unsigned)WKC * (unsigned)HW * 4u;
        if constexpr (RAGGED) {
            --chunks_left;
            mask_ragged_channels();
        }
    };
    auto issue = [&](int buf) {
#pragma unroll
        for (int idx = 0; idx < WPI + WUI; ++idx) dma_piece(idx, buf);
        advance();
    };
    auto fixup = [&](int buf) {
        if (edge && fix) {
            float* pw = lds + buf * WSTAGE + WUS + tid * 4;
#pragma unroll
            for (int i = 0; i < WPI; ++i) {
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (fix & (1 << (4 * i + e))) pw[i * WNT * 4 + e] = 0.f;
            }
        }
    };

    // ---- the lane's role in the MFMAs
    const int wm = wave >> 1, wn = wave & 1;                 // channel half (32) / row half (4 rows = 2 tile rows) of the tile
    const int nl = lane & 31, kh = lane >> 5;
    const int ttx = nl & 15, tty = nl >> 4;                  // tile column / row inside the wave's 16 x 2 tiles
    // the lane's tile: strip (image, band), first output pixel.  Evaluated here for `active` and AGAIN in the epilogue (from an
    // opaque copy of u0: keeping five more values alive across the main loop spills, and with scratch in the kernel the
    // compiler puts a vmcnt(0) in front of every LDS read that follows a DMA -- 2x slower)
    auto tile_geometry = [&](int u0v, int& tsn_o, int& py_o, int& px_o) {
        const int sf = u0v / period;                         // (scalar divisions) strip / image / band of the first tile column
        int px_t = u0v + 2 * ttx - sf * period, sn = sf / bands, band = sf - sn * bands;
        while (px_t >= period) { px_t -= period; ++band; }
        while (band >= bands) { band -= bands; ++sn; }
        tsn_o = sn;
        py_o = band * WTH + wn * 4 + tty * 2;
        px_o = px_t;
        return sn < N && px_o < W && py_o < H;
    };
    bool active;
    {
        int a0, a1, a2;
        active = __any(tile_geometry(u0, a0, a1, a2));        // (wave-uniform) some tile of the wave lies inside an image
    }
    const int a_off = kh * (4 * WBM * 4) + (wm * 32 + nl) * 4;                               // + ks * 2048 + i * 256
    const int b_off = WUS + kh * WPL + (wn * 4 + tty * 2) * WPP + 2 * ttx + 3;              // + ks * 960 + a * 48 + 2 c   (odd: ds_read2_b32)

    // the lane's 16 biases, fetched BEFORE the first DMA (in-order vmcnt: they have landed long before anything the loop waits
    // for): loaded at the head of the epilogue their latency was exposed once per workgroup -- 2.7 k cycles, 4 % at Cin = 64
    const int half4 = 4 * kh;
    const int co_w = cot * WBM + wm * 32;                    // wave-uniform first channel
    f32x4 bv[4];
    {
        const __amdgpu_buffer_rsrc_t rbias = ptmi_rsrc(bias ? bias : y, bias ? (unsigned)Cout * 4u : 0u);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bv[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (epi <= 1 || epi == 4)
                bv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, (half4 + co_w + g * 8) * 4, 0, 0));
        }
    }

    f32x16 acc[16];                        // defined by the first k-step's MFMAs (zero C operand)

    if (active) {
        f32x4 A0[4], A1[4];
        f32x2 D0[8];                       // raw window of the NEXT k-step: read (slots 0-7) and transformed (12-15) within one k-step
        f32x2 V0[8], V1[8];
        f32x2 T[4][2];
        // One k-step = 16 MFMAs on (A, V), one per position, with the rest of the wave's work placed between them (one
        // wave per SIMD: whatever is not issued in the shadow of an MFMA leaves the matrix pipe idle):
        //   P = 0      in the chunk's LAST k-step the hand-over (this k-step's operands are already in registers): own DMA
        //              pieces of the next chunk landed (counted vmcnt), edge fix-ups, workgroup barrier
        //   P = 0..7   the next k-step's eight window reads (one 4-byte-aligned pair each)
        //   P = 4..7   one DMA instruction each for the chunk AFTER the next (k-steps 0..2 carry its 4 + 8 instructions)
        //   P = 8..11  the next k-step's four 16-byte A reads
        //   P = 12, 13 input transform, rows (one column pair each: 4 packed additions);  P = 14, 15  columns (two position
        //              rows each: 4 packed additions) -- 16 VALU instructions per k-step: VALU time ADDS to fp32 MFMA time
        // (s_memtime probes, tools/exp: with the hand-over's reads still in flight, the edge fix-up branches in the main
        // body and two LDS stages the chunk's last k-step took 1.8k cycles against 1.1-1.2k for the others.)
        auto kstep = [&](auto ks_c, auto more_c, auto more2_c, auto edge_c, auto zc_c, const f32x4 (&A)[4], const f32x2 (&V)[8],
                         f32x4 (&An)[4], f32x2 (&Dn)[8], f32x2 (&Vn)[8], int cur, int nxt, int nn) {
            constexpr int KS = decltype(ks_c)::value;
            constexpr bool more = decltype(more_c)::value;        // a chunk follows this one
            constexpr bool more2 = decltype(more2_c)::value;      // ... and another one after it (its DMA is issued here)
            constexpr bool EDGE = decltype(edge_c)::value;        // right-edge workgroup: DMA pieces may need fix-ups
            constexpr bool ZC = decltype(zc_c)::value;            // a tile's very first k-step: zero C operand instead of 256 accumulator writes
            constexpr bool next = KS < 3 || more;
            const float* src = lds + (KS < 3 ? cur : nxt) * WSTAGE;
            const float* ap = src + a_off + ((KS + 1) & 3) * (2 * 4 * WBM * 4);
            const float* bp = src + b_off + ((KS + 1) & 3) * (2 * WPL);
            {   // one address register for the k-step's eight window reads: ds_read2_b32 offsets are 8-bit dword counts
                unsigned bpa = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)bp;
                asm volatile("" : "+v"(bpa));
                bp = (const float*)(const __attribute__((address_space(3))) float*)(size_t)bpa;
            }
            auto dread = [&](int e) { Dn[e] = *(const volatile wlds_f32x2_a4_t*)(bp + (e >> 1) * WPP + 2 * (e & 1)); };
            auto step = [&](auto p_c) {
                constexpr int P = decltype(p_c)::value;
                if constexpr (ZC) acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[P >> 2][P & 3], wino_vop(V, P), (f32x16){0}, 0, 0, 0);
                else acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[P >> 2][P & 3], wino_vop(V, P), acc[P], 0, 0, 0);
                if constexpr (P == 0 && KS == 3 && more) {
                    // chunk + 1 was issued a whole chunk ago; the WPI + WUI DMA instructions of chunk + 2 are newer
                    if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI + WUI) : "memory");
                    else wino_vmwait0();
                    if (EDGE) fixup(nxt);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                if constexpr (P <= 7) {                       // ONE window read per slot
                    if (next) dread(P);
                }
                if constexpr (P >= 8 && P <= 11) {            // ONE A read per slot
                    if (next) An[P - 8] = *(const volatile wlds_f32x4_t*)(ap + (P - 8) * (WBM * 4));
                }
                if constexpr (P >= 4 && P <= 7) {
                    if (KS < 3 && more2) dma_piece(KS * 4 + (P - 4), nn);
                    if (P == 7 && KS == 2 && more2) advance();
                }
                if constexpr (P == 12 || P == 13) {
                    if (next) wino_xform_rows(Dn, T, P - 12);
                }
                if constexpr (P >= 14) {
                    if (next) {
                        wino_xform_col(T, Vn, 2 * (P - 14));
                        wino_xform_col(T, Vn, 2 * (P - 14) + 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{});  step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});  step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{});  step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{});  step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{});  step(std::integral_constant<int, 9>{});
            step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
            step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
        };

        issue(0);
        if (nChunks > 1) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI + WUI) : "memory");
        } else {
            wino_vmwait0();
        }
        fixup(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {   // operands of the first k-step
            const float* ap = lds + a_off;
            const float* bp = lds + b_off;
#pragma unroll
            for (int i = 0; i < 4; ++i) A0[i] = *(const volatile wlds_f32x4_t*)(ap + i * (WBM * 4));
#pragma unroll
            for (int e = 0; e < 8; ++e) D0[e] = *(const volatile wlds_f32x2_a4_t*)(bp + (e >> 1) * WPP + 2 * (e & 1));
            wino_xform(D0, V0);
        }
        // three stages in LDS: chunk c is computed from stage c % 3 while chunk c + 1 sits complete (or landing) in the next
        // one and chunk c + 2 is being fetched into the third: a DMA piece has a whole chunk (~4.5k cycles) to land before
        // the hand-over that waits for it.  Right-edge workgroups (some 16-B piece straddles the image edge) run their own
        // copy of the loop with the fix-up writes in the hand-over; the common copy is branch-free.
        auto main_loop = [&](auto edge_c) {
            auto chunk_body = [&](auto more_c, auto more2_c, auto zc_c, int cur, int nxt, int nn) {
                kstep(std::integral_constant<int, 0>{}, more_c, more2_c, edge_c, zc_c, A0, V0, A1, D0, V1, cur, nxt, nn);
                kstep(std::integral_constant<int, 1>{}, more_c, more2_c, edge_c, std::false_type{}, A1, V1, A0, D0, V0, cur, nxt, nn);
                kstep(std::integral_constant<int, 2>{}, more_c, more2_c, edge_c, std::false_type{}, A0, V0, A1, D0, V1, cur, nxt, nn);
                kstep(std::integral_constant<int, 3>{}, more_c, more2_c, edge_c, std::false_type{}, A1, V1, A0, D0, V0, cur, nxt, nn);
            };
            const std::true_type T{};
            const std::false_type F{};
            int cur = 0, nxt = 1, nn = 2;
            auto rotate = [&]() { const int t = cur; cur = nxt; nxt = nn; nn = t; };
            // the first chunk's first k-step defines the accumulators (zero C operand)
            if (nChunks > 2) {
                chunk_body(T, T, T, cur, nxt, nn);
                rotate();
                for (int chunk = 1; chunk + 2 < nChunks; ++chunk) {
                    chunk_body(T, T, F, cur, nxt, nn);
                    rotate();
                }
                chunk_body(T, F, F, cur, nxt, nn);
                cur = nxt;
                chunk_body(F, F, F, cur, nxt, nn);
            } else if (nChunks == 2) {
                chunk_body(T, F, T, cur, nxt, nn);
                cur = nxt;
                chunk_body(F, F, F, cur, nxt, nn);
            } else {
                chunk_body(F, F, T, cur, nxt, nn);
            }
        };
        if (edge) main_loop(std::true_type{});
        else main_loop(std::false_type{});
    } else {
        // a wave whose rows all lie below the image: same DMA issue / wait / barrier sequence, no MFMAs
        issue(0);
        if (nChunks > 1) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI + WUI) : "memory");
        } else {
            wino_vmwait0();
        }
        fixup(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int nxt = 1, nn = 2;
        for (int chunk = 0; chunk + 1 < nChunks; ++chunk) {
            if (chunk + 2 < nChunks) {
                issue(nn);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI + WUI) : "memory");
            } else {
                wino_vmwait0();
            }
            fixup(nxt);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int t = nxt; nxt = nn; nn = (t + 2) % 3;
        }
        return;
    }

    // ---- epilogue: Y = A^T M A per (channel, tile) in registers, then bias / ReLU / mask / pool and buffer stores.
    // Accumulator element r of a lane: channel (r & 3) + 8 (r >> 2) + 4 kh of the wave's 32, tile nl.
    int tsn, py, px;
    int u0e = u0;
    asm volatile("" : "+s"(u0e));
    const bool tile_ok = tile_geometry(u0e, tsn, py, px);
    // channel row rr of this lane lies inside the tensor (the descriptors below reach to the END of the tensor -- a lane's tile
    // may belong to a later image than the workgroup's first -- so the range check no longer drops channels >= Cout)
    const int cmax = Cout - co_w - half4;
    auto crow = [&](int rr) { return (rr & 3) + 8 * (rr >> 2) < cmax; };
    // Two accumulator rows (channels r, r + 1: adjacent registers of every position tile) at a time on packed additions, all
    // opaque to the optimiser (left to itself the compiler spends ~1 270 instructions per lane here -- 590 accumulator reads for
    // 256 values, 250 register moves around its own packed operations -- and the epilogue is 5-20 % of the kernel).
    auto inverse2 = [&](auto rp_c, f32x2 (&o)[4]) {
        constexpr int r = 2 * decltype(rp_c)::value;
        f32x2 s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (explicit reads: an element extracted in C++ makes the compiler copy the WHOLE 16-register tile to VGPRs, all 16
            // tiles up front -- 256 live VGPRs at the head of the epilogue, spills of the DMA offsets in the main loop)
            auto rd = [](float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; };
            const f32x2 m0 = {rd(acc[j][r]), rd(acc[j][r + 1])}, m1 = {rd(acc[4 + j][r]), rd(acc[4 + j][r + 1])};
            const f32x2 m2 = {rd(acc[8 + j][r]), rd(acc[8 + j][r + 1])}, m3 = {rd(acc[12 + j][r]), rd(acc[12 + j][r + 1])};
            s0[j] = pk_add(pk_add(m0, m1), m2);
            s1[j] = pk_sub(pk_sub(m1, m2), m3);
        }
        o[0] = pk_add(pk_add(s0[0], s0[1]), s0[2]);
        o[1] = pk_sub(pk_sub(s0[1], s0[2]), s0[3]);
        o[2] = pk_add(pk_add(s1[0], s1[1]), s1[2]);
        o[3] = pk_sub(pk_sub(s1[1], s1[2]), s1[3]);
    };
    auto for_row_pairs = [&](auto&& f) {
        f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});
        f(std::integral_constant<int, 3>{}); f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{});
        f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{});
    };
    if (epi == 4) {
        // bias + ReLU + 2x2/2 max pool (floor mode): the tile's four outputs are one pool window
        const int OH = H >> 1, OW = W >> 1, OHW = OH * OW;
        const long long ytail = (long long)(N - n) * Cout * OHW * 4;
        const __amdgpu_buffer_rsrc_t ry = ptmi_rsrc(y + (size_t)n * Cout * OHW, (unsigned)(ytail > 0xFFFFFFFEll ? 0xFFFFFFFEll : ytail));
        const int oy = py >> 1, ox = px >> 1;
        const unsigned pv = (tile_ok && oy < OH && ox < OW) ? (unsigned)(((tsn - n) * Cout + half4) * OHW + oy * OW + ox) * 4u : 0xFFFFFFFFu;
        for_row_pairs([&](auto rp_c) {
            constexpr int r = 2 * decltype(rp_c)::value;
            f32x2 o[4];
            inverse2(rp_c, o);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int soff = (co_w + ((r + h) & 3) + 8 * ((r + h) >> 2)) * OHW * 4;
                const float b = bv[(r + h) >> 2][(r + h) & 3];
                const float m = fmaxf(fmaxf(fmaxf(o[0][h] + b, o[1][h] + b), fmaxf(o[2][h] + b, o[3][h] + b)), 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), ry, crow(r + h) ? (int)pv : -1, soff, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        return;
    }
    const long long ytail = (long long)(N - n) * Cout * HW * 4;
    const unsigned img_bytes = (unsigned)(ytail > 0xFFFFFFFEll ? 0xFFFFFFFEll : ytail);     // to the end of the tensor, clamped
    const __amdgpu_buffer_rsrc_t ry = ptmi_rsrc(y + (size_t)n * Cout * HW, img_bytes);
    const __amdgpu_buffer_rsrc_t rm = ptmi_rsrc(epi == 3 ? mref + (size_t)n * Cout * HW : y, epi == 3 ? img_bytes : 0u);
    // per-lane byte offsets of the tile's two rows: pair = both columns inside the image (8-byte access), single = only
    // the first one (odd W, last column)
    unsigned pv2[2], pv1[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const bool rok = tile_ok && py + a < H;
        const unsigned o = (unsigned)(((tsn - n) * Cout + half4) * HW + (py + a) * W + px) * 4u;
        pv2[a] = (rok && px + 1 < W) ? o : 0xFFFFFFFFu;
        pv1[a] = (rok && px + 1 == W) ? o : 0xFFFFFFFFu;
    }
    const bool odd_edge = (W & 1) && __any(pv1[0] != 0xFFFFFFFFu);    // (wave-uniform) a lane holds a single column
    auto store_rows = [&](auto epi_c) {
        constexpr int EPI = decltype(epi_c)::value;
        for_row_pairs([&](auto rp_c) {
            constexpr int r = 2 * decltype(rp_c)::value;
            f32x2 mk[2][2];
            float ms[2][2];
            if constexpr (EPI == 3) {                                  // the producer's activations first: their latency hides
#pragma unroll                                                         // behind the inverse transform
                for (int h = 0; h < 2; ++h) {
                    const int soff = (co_w + ((r + h) & 3) + 8 * ((r + h) >> 2)) * HW * 4;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        mk[h][a] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rm, crow(r + h) ? (int)pv2[a] : -1, soff, 0));
                        if (odd_edge) ms[h][a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, crow(r + h) ? (int)pv1[a] : -1, soff, 0));
                    }
                }
            }
            f32x2 o[4];
            inverse2(rp_c, o);
            if constexpr (EPI <= 1) {
                const f32x2 b2 = {bv[r >> 2][r & 3], bv[r >> 2][(r & 3) + 1]};
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = pk_add(o[k], b2);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int soff = (co_w + ((r + h) & 3) + 8 * ((r + h) >> 2)) * HW * 4;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v0 = o[2 * a][h], v1 = o[2 * a + 1][h];
                    if constexpr (EPI == 1) {
                        v0 = fmaxf(v0, 0.f);
                        v1 = fmaxf(v1, 0.f);
                    } else if constexpr (EPI == 3) {
                        float m0 = mk[h][a][0];
                        if (odd_edge) m0 = (pv1[a] != 0xFFFFFFFFu) ? ms[h][a] : m0;
                        v0 = (m0 > 0.f) ? v0 : 0.f;
                        v1 = (mk[h][a][1] > 0.f) ? v1 : 0.f;
                    }
                    const f32x2 st = {v0, v1};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, st), ry, crow(r + h) ? (int)pv2[a] : -1, soff, 0);
                    if (odd_edge) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), ry, crow(r + h) ? (int)pv1[a] : -1, soff, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // one row pair at a time: hoisted accumulator reads of later pairs spill
        });
    };
    if (epi == 0) store_rows(std::integral_constant<int, 0>{});
    else if (epi == 1) store_rows(std::integral_constant<int, 1>{});
    else if (epi == 2) store_rows(std::integral_constant<int, 2>{});
    else store_rows(std::integral_constant<int, 3>{});
}

// U = G g G^T with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], laid out as the kernel's LDS image:
// [channel tile (64)][chunk (8 ci)][ci][position row i][co][position column j].  mode as ptmi_conv3x3_pack_weights.
__global__ void wino_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int wCout, int wCin, int mode,
                                         int coTiles, int nChunks)
{
    const int64_t total = (int64_t)coTiles * nChunks * WKC * WBM;
    const int convCout = mode ? wCin : wCout, convCin = mode ? wCout : wCin;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = idx;
        const int col = t % WBM; t /= WBM;
        const int cil = t % WKC; t /= WKC;
        const int chunk = t % nChunks;
        const int cot = t / nChunks;
        const int co = cot * WBM + col, ci = chunk * WKC + cil;
        float g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float v = 0.f;
                if (co < convCout && ci < convCin)
                    v = mode == 0 ? w[((size_t)co * wCin + ci) * 9 + ky * 3 + kx]
                                  : w[((size_t)ci * wCin + co) * 9 + (2 - ky) * 3 + (2 - kx)];
                g[ky][kx] = v;
            }
        }
        float r[4][3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            r[0][kx] = g[0][kx];
            r[1][kx] = 0.5f * ((g[0][kx] + g[1][kx]) + g[2][kx]);
            r[2][kx] = 0.5f * ((g[0][kx] - g[1][kx]) + g[2][kx]);
            r[3][kx] = g[2][kx];
        }
        float* dst = wp + ((size_t)(cot * nChunks + chunk) * WKC + cil) * (4 * WBM * 4) + col * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 u;
            u[0] = r[i][0];
            u[1] = 0.5f * ((r[i][0] + r[i][1]) + r[i][2]);
            u[2] = 0.5f * ((r[i][0] - r[i][1]) + r[i][2]);
            u[3] = r[i][2];
            *reinterpret_cast<f32x4*>(dst + i * (WBM * 4)) = u;
        }
    }
}


// ================================================================================================ weight gradient
// dW through the same transform domain:  Y = A^T [U (.) V] A  =>  dU_p[co][ci] = sum over (image, tile) of W_p[co][tile] V_p[ci][tile]
// with W = A dY A^T (the 2x2 output-gradient tile spread over the 16 positions) and V = B^T d B as in forward, then
// dg = G^T dU G.  16 multiplies per tile and channel pair instead of 36 -- the same 2.25x as forward.
//
// GEMM per position: M = co, N = ci, K = tiles.  A workgroup owns 64 co x 64 ci (a wave 32 x 32 x 16 positions = 256
// accumulator registers, one wave per SIMD) and a contiguous range of "chunks" -- (image, tile row, 32-column block) = 16
// tiles = 8 k-steps of 2 tiles -- of the split it belongs to; partial dU go to the workspace [split][position][co][ci] and a
// second kernel sums the splits in a fixed order (deterministic) and applies G^T . G.
//   * MFMA operands: lane (co = lane & 31, kh = lane >> 5) transforms the 2x2 dY tile of ITS channel and tile 2 ks + kh
//     (two 8-byte LDS reads, 12 additions), lane (ci, kh) the 4x4 input window of its channel and the same tile (twelve reads,
//     32 additions): again no cross-lane traffic.  W is computed without its minus signs (row / column 3 of A dY A^T are
//     negated sums); the reduction kernel multiplies position (i, j) by s_i s_j, s = (1, 1, 1, -1).
//   * lanes of one MFMA operand differ in the CHANNEL, so the LDS plane pitches are odd multiples of 4 floats (68 / 164): the
//     32 lanes of an 8-byte read fall on 16 distinct bank pairs (2-way; 16-byte DMA pieces allow no better).
//   * two LDS stages; a chunk's operands are fetched (5 + 11 DMA instructions per lane: 64 channels x 2 x 32 pixels of dY,
//     64 x 4 x 40 of x) four per k-step during the last two k-steps of chunk c - 2 and the first two of chunk c - 1; LDS reads
//     run two k-steps ahead of the MFMAs and the transforms one, so the chunk hand-over (vmcnt(0), edge fix-ups, barrier) sits at
//     k-step 6 of 8 with every own read four k-steps old.
constexpr int GWC = 64;                  // channels per operand tile
// geometry of a chunk of KSN k-steps (= 2 KSN tiles = 4 KSN columns of one tile row).  KSN = 8: 32 columns; KSN = 7: 28 columns,
// which tiles W = 83 / 166 / 333 (21 / 42 / 84 tile pairs per row) without the 12.5 / 12.5 / 4.5 % of padded k-steps of KSN = 8
template <int KSN> struct WG {
    static constexpr int COLS = 4 * KSN;
    static constexpr int XPR = KSN + 2;              // 16-B pieces per x row: columns x0 - 4 .. x0 + 4 KSN + 3
    static constexpr int XPC = 4 * XPR + 1;          // pieces per x channel plane: 4 rows + one pad piece
    static constexpr int DPC = 2 * KSN + 1;          // pieces per dY channel plane: 2 rows + one pad piece
    static constexpr int GXP = 4 * XPC;              // x plane pitch in floats (164 / 148: an odd multiple of 4)
    static constexpr int GDYP = 4 * DPC;             // dY plane pitch (68 / 60)
    static constexpr int GNX = (GWC * XPC + 255) / 256;      // x DMA instructions per lane and chunk (11 / 10)
    static constexpr int GND = (GWC * DPC + 255) / 256;      // dY DMA instructions (5 / 4)
    static constexpr int NP = GNX + GND;
    static constexpr int GDYR = GND * 1024;          // floats of the dY region of a stage (regions padded to whole instructions)
    static constexpr int GXR = GNX * 1024;
    static constexpr int GSTAGE = GDYR + GXR;        // 64 KB / 56 KB; two stages
};

template <int KSN>
__global__ __launch_bounds__(WNT, 1) void conv3x3_wino_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, float* __restrict__ bpartial,
    int N, int Cin, int Cout, int H, int W, int ciTiles, int S, int tilesY2, int colBlocks)
{
    using G = WG<KSN>;
    constexpr int GNX = G::GNX, GND = G::GND, GDYR = G::GDYR, GSTAGE = G::GSTAGE, GXP = G::GXP, GDYP = G::GDYP;
    __shared__ __attribute__((aligned(16))) float lds[2 * GSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * W;
    int bid = blockIdx.x;
    const int split = bid % S; bid /= S;
    const int cit = bid % ciTiles, cot = bid / ciTiles;
    const int co0 = cot * GWC, ci0 = cit * GWC;
    // this split's contiguous range of the chunk list (image, tile row, column block; column block fastest)
    const int nChunksAll = N * tilesY2 * colBlocks;
    const int per = (nChunksAll + S - 1) / S;
    const int cBegin = split * per;
    const int nC = max(min(cBegin + per, nChunksAll) - cBegin, 0);

    // ---- per-lane DMA pieces (fetch order: 11 of x, then 5 of dY), relative to the chunk's origin: x (y0 - 1, x0 - 4), dY (y0, x0).
    // Validity is (channel) & (column, by the class of the column block: first / interior / last) & (row, by the tile row):
    // bit masks over the lane's 16 pieces, built once; a chunk combines them with a handful of wave-uniform selects.
    unsigned voff[GNX + GND];
    unsigned m_ch = 0, m_first = 0, m_last = 0;          // channel inside the tensor; column valid in the first / last column block
    unsigned m_r0 = 0, m_r2 = 0, m_r3 = 0, m_d1 = 0;     // x pieces of window row 0 / 2 / 3, dY pieces of row 1
    unsigned long long fix_last = 0;                     // last column block: words past the image edge, bits 4 i + e
    const int wvl = W - (colBlocks - 1) * G::COLS;       // width of the last column block (1 .. 4 KSN)
#pragma unroll
    for (int i = 0; i < GNX; ++i) {
        const int px = tid + i * WNT;
        const int ch = px / G::XPC, rem = px - ch * G::XPC;
        const int r = rem / G::XPR, q4 = 4 * (rem - r * G::XPR) - 4;
        const bool ok = px < GWC * G::XPC && rem < 4 * G::XPR && ci0 + ch < Cin;
        voff[i] = ok ? (unsigned)(ch * HW + r * W + q4 + 4) * 4u : 0xFFFFFFFFu;
        m_ch |= (unsigned)ok << i;
        m_first |= (unsigned)(q4 >= 0) << i;
        m_last |= (unsigned)(q4 < wvl) << i;
        m_r0 |= (unsigned)(r == 0) << i;
        m_r2 |= (unsigned)(r == 2) << i;
        m_r3 |= (unsigned)(r == 3) << i;
        if (ok && q4 < wvl && q4 + 4 > wvl)
            for (int e = 1; e < 4; ++e) fix_last |= (q4 + e >= wvl) ? (1ull << (4 * i + e)) : 0ull;
    }
#pragma unroll
    for (int i = 0; i < GND; ++i) {
        const int pd = tid + i * WNT;
        const int ch = pd / G::DPC, rem = pd - ch * G::DPC;
        const int r = rem / KSN, q4 = 4 * (rem - r * KSN);
        const bool ok = pd < GWC * G::DPC && rem < 2 * KSN && co0 + ch < Cout;
        voff[GNX + i] = ok ? (unsigned)(ch * HW + r * W + q4) * 4u : 0xFFFFFFFFu;
        m_ch |= (unsigned)ok << (GNX + i);
        m_first |= 1u << (GNX + i);
        m_last |= (unsigned)(q4 < wvl) << (GNX + i);
        m_d1 |= (unsigned)(r == 1) << (GNX + i);
        if (ok && q4 < wvl && q4 + 4 > wvl)
            for (int e = 1; e < 4; ++e) fix_last |= (q4 + e >= wvl) ? (1ull << (4 * (GNX + i) + e)) : 0ull;
    }
    const char* x_end = (const char*)(x + (size_t)N * Cin * HW);
    const char* dy_end = (const char*)(dy + (size_t)N * Cout * HW);
    auto clamp_rec = [](long long rem) { return (int)(rem > 0xFFFFFFFEll ? 0xFFFFFFFEll : (rem < 0 ? 0 : rem)); };

    // ---- the fetch in progress: descriptors, effective per-piece offsets, fix-up mask
    int f_cb, f_ty, f_n;                                 // coordinates of the NEXT chunk to set up
    {
        const int g = min(cBegin, max(nChunksAll - 1, 0));
        f_cb = g % colBlocks;
        const int t = g / colBlocks;
        f_ty = t % tilesY2;
        f_n = t / tilesY2;
    }
    int f_left = nC;                                     // chunks of this split not yet set up
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(ptmi_uniform_ptr(x), 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = rx;
    unsigned eoff[GNX + GND];                            // effective offsets of a chunk on the image border (else voff)
    bool f_plain = false;
    unsigned long long fix = 0;
    auto fetch_setup = [&]() {
        const bool any = f_left > 0;
        const int y0 = 2 * f_ty, x0 = f_cb * G::COLS;
        const float* xb = x + ((size_t)f_n * Cin + ci0) * HW + ((ptrdiff_t)y0 - 1) * W + (x0 - 4);
        const float* db = dy + ((size_t)f_n * Cout + co0) * HW + (size_t)y0 * W + x0;
        rx = __builtin_amdgcn_make_buffer_rsrc(ptmi_uniform_ptr(xb), 0, clamp_rec(x_end - (const char*)xb), 0x00020000);
        rd = __builtin_amdgcn_make_buffer_rsrc(ptmi_uniform_ptr(db), 0, clamp_rec(dy_end - (const char*)db), 0x00020000);
        const bool first = f_cb == 0, last = f_cb == colBlocks - 1;
        const bool top = y0 == 0, bot2 = y0 + 1 >= H, bot3 = y0 + 2 >= H;
        f_plain = any && !first && !last && !top && !bot3;
        if (f_plain) {                                   // interior chunk: only the channel mask (already in voff)
            fix = 0;
        } else {
            unsigned m = any ? m_ch : 0u;
            m &= first ? m_first : ~0u;
            m &= last ? m_last : ~0u;
            m &= top ? ~m_r0 : ~0u;
            m &= bot2 ? ~(m_r2 | m_d1) : ~0u;
            m &= bot3 ? ~m_r3 : ~0u;
            fix = (any && last) ? fix_last : 0ull;
            // (a piece whose row or channel is masked needs no fix-up: zeroing zeros is harmless)
#pragma unroll
            for (int i = 0; i < GNX + GND; ++i) eoff[i] = (m >> i) & 1 ? voff[i] : 0xFFFFFFFFu;
        }
        --f_left;
        if (++f_cb == colBlocks) {
            f_cb = 0;
            if (++f_ty == tilesY2) { f_ty = 0; ++f_n; }
        }
    };
    auto fetch_piece = [&](int idx, int stage) {
        // (a scalar branch instead of a per-lane select: every VALU instruction in the loop costs MFMA time)
        wlds_void_t* dst = (wlds_void_t*)(lds + stage * GSTAGE + (idx < GNX ? GDYR + wave * 256 + idx * WNT * 4 : wave * 256 + (idx - GNX) * WNT * 4));
        if (f_plain) __builtin_amdgcn_raw_ptr_buffer_load_lds(idx < GNX ? rx : rd, dst, 16, (int)voff[idx], 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(idx < GNX ? rx : rd, dst, 16, (int)eoff[idx], 0, 0, 0);
    };
    auto fixup = [&](int stage, unsigned long long fm) {
        if ((unsigned)fm | (unsigned)(fm >> 32)) {
#pragma unroll
            for (int i = 0; i < GNX + GND; ++i) {
                float* pc = lds + stage * GSTAGE + (i < GNX ? GDYR + (tid + i * WNT) * 4 : (tid + (i - GNX) * WNT) * 4);
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (fm & (1ull << (4 * i + e))) pc[e] = 0.f;
            }
        }
    };

    // ---- the lane's role in the MFMAs
    const int wm = wave >> 1, wn = wave & 1;                 // co half / ci half of the 64 x 64 tile
    const int nl = lane & 31, kh = lane >> 5;
    const int a_off = (wm * 32 + nl) * GDYP + 2 * kh;                                    // + 4 ks  (tile 2 ks + kh), + 32 r
    const int b_off = GDYR + (wn * 32 + nl) * GXP + 2 * kh + 3;                          // + 4 ks, + 40 a + 2 t   (odd: ds_read2_b32)

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = (f32x16){0};
    f32x2 RA[2][2], RB[2][8];              // raw operands of two k-steps in flight: the 2x2 dY tile, the 4x4 window (row a: 2 a, 2 a + 1)
    // transformed operands of two consecutive k-steps.  Position p = 4 i + j:
    //   W' = A dY A^T (signs dropped): WR[i] = (w_i0, w_i3), WZ[i] = (w_i1, w_i2);   V = B^T d B: VX[i] = (v_i0, v_i3), VY[i] = (v_i1, v_i2)
    f32x2 WR[2][4], WZ[2][4], VX[2][4], VY[2][4];
    f32x2 T[4][2];                         // B^T d, rows i, column pairs (0, 1), (2, 3)
    float bsum = 0.f;                      // sum of this lane's dY tiles: position (1, 1) of W' is d00 + d01 + d10 + d11

    // one address register per k-step for the eight window reads (ds_read2_b32 offsets are 8-bit dword counts)
    auto window_base = [&](const float* stage, int ks) {
        unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(stage + b_off + 4 * ks);
        asm volatile("" : "+v"(a));
        return (const float*)(const __attribute__((address_space(3))) float*)(size_t)a;
    };
    auto raw_read = [&](const float* stage, const float* wb, int ks, f32x2 (&ra)[2], f32x2 (&rb)[8], int P) {      // one read per MFMA slot P = 0..9
        if (P < 2) ra[P] = *(const volatile wlds_f32x2_t*)(stage + a_off + 4 * ks + G::COLS * P);
        else if (P < 10) rb[P - 2] = *(const volatile wlds_f32x2_a4_t*)(wb + ((P - 2) >> 1) * (4 * G::XPR) + 2 * ((P - 2) & 1));
    };
    auto xform_a_rows = [&](const f32x2 (&ra)[2], f32x2 (&wr)[4]) {
        wr[0] = ra[0];
        wr[1] = pk_add(ra[0], ra[1]);
        wr[2] = pk_sub(ra[0], ra[1]);
        wr[3] = ra[1];
    };
    auto xform_a_col = [&](const f32x2 (&wr)[4], f32x2 (&wz)[4], int i) { wz[i] = pk_lo_pm_hi(wr[i], wr[i]); };
    auto xform_b_rows = [&](const f32x2 (&rb)[8], int c) {      // column pair c of all four rows
        T[0][c] = pk_sub(rb[0 + c], rb[4 + c]);
        T[1][c] = pk_add(rb[2 + c], rb[4 + c]);
        T[2][c] = pk_sub(rb[4 + c], rb[2 + c]);
        T[3][c] = pk_sub(rb[2 + c], rb[6 + c]);
    };
    auto xform_b_col = [&](f32x2 (&vx)[4], f32x2 (&vy)[4], int i) {
        vx[i] = pk_sub(T[i][0], T[i][1]);                 // (t0 - t2, t1 - t3)
        vy[i] = pk_lo_pm_hi(T[i][1], T[i][0]);            // (t2 + t1, t2 - t1)
    };
    auto opa = [&](int M, int p) -> float { const int i = p >> 2, j = p & 3; return (j == 0 || j == 3) ? WR[M][i][j == 3] : WZ[M][i][j == 2]; };
    auto opb = [&](int M, int p) -> float { const int i = p >> 2, j = p & 3; return (j == 0 || j == 3) ? VX[M][i][j == 3] : VY[M][i][j == 2]; };

    if (nC > 0) {
        // ---- start: chunk 0 entirely, chunk 1's first two quarters
        fetch_setup();
#pragma unroll
        for (int idx = 0; idx < GNX + GND; ++idx) fetch_piece(idx, 0);
        const unsigned long long fix0 = fix;
        fetch_setup();
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) fetch_piece(idx, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        fixup(0, fix0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        unsigned long long fix_h = fix;          // fix-ups of the next chunk to be handed over (chunk 1)
        {   // operands of k-step 0 (transformed) and raw operands of k-step 1
#pragma unroll
            for (int P = 0; P < 10; ++P) raw_read(lds, window_base(lds, 0), 0, RA[0], RB[0], P);
            xform_a_rows(RA[0], WR[0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) xform_a_col(WR[0], WZ[0], i);
            xform_b_rows(RB[0], 0);
            xform_b_rows(RB[0], 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) xform_b_col(VX[0], VY[0], i);
#pragma unroll
            for (int P = 0; P < 10; ++P) raw_read(lds, window_base(lds, 1), 1, RA[1], RB[1], P);
        }
        // k-step KS of the chunk in stage `cur`, operand parity PAR: MFMAs on the operands [M], M = (KS + PAR) & 1; raw reads of k-step
        // KS + 2 into R[M]; transforms of k-step KS + 1 (raw R[M ^ 1], read one k-step ago) into the operands [M ^ 1].  With an odd
        // number of k-steps per chunk the parity alternates from chunk to chunk: the loop body is then two chunks.
        auto kstep = [&](auto ks_c, auto par_c, int cur) {
            constexpr int KS = decltype(ks_c)::value;
            constexpr int M = (KS + decltype(par_c)::value) & 1, O = M ^ 1;
            constexpr bool late = KS >= KSN - 2;                 // after the hand-over: reads come from the next chunk's stage
            const float* src = lds + (late ? cur ^ 1 : cur) * GSTAGE;
            constexpr int KR = (KS + 2) % KSN;
            const float* wb = window_base(src, KR);
            auto step = [&](auto p_c) {
                constexpr int P = decltype(p_c)::value;
                acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa(M, P), opb(M, P), acc[P], 0, 0, 0);
                if constexpr (P == 15) bsum += WZ[M][1][0];
                if constexpr (P == 0 && KS == KSN - 2) {
                    // hand-over: the next chunk's pieces (all issued by k-step 1) have landed; edge fix-ups; barrier; then the
                    // stage this chunk occupied is free (its last raw reads were k-step KSN - 3's) and the fetch of chunk + 2 starts
                    wino_vmwait0();
                    fixup(cur ^ 1, fix_h);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    fetch_setup();
                    fix_h = fix;
                }
                if constexpr (P < 10) raw_read(src, wb, KR, RA[M], RB[M], P);
                // fetch of chunk + 2 (last two k-steps) / of chunk + 1 (k-steps 0, 1): four DMA instructions per k-step
                if constexpr ((late || KS == 0 || KS == 1) && (P == 1 || P == 8 || P == 9 || P == 14)) {
                    constexpr int part = KS == KSN - 2 ? 0 : (KS == KSN - 1 ? 1 : (KS == 0 ? 2 : 3));
                    constexpr int sub = P == 1 ? 0 : (P == 8 ? 1 : (P == 9 ? 2 : 3));
                    if constexpr (part * 4 + sub < G::NP) {
                        fetch_piece(part * 4 + sub, late ? cur : cur ^ 1);
                    }
                }
                // transforms of k-step KS + 1
                if constexpr (P == 2) xform_a_rows(RA[O], WR[O]);
                if constexpr (P == 3) { xform_a_col(WR[O], WZ[O], 0); xform_a_col(WR[O], WZ[O], 1); xform_a_col(WR[O], WZ[O], 2); xform_a_col(WR[O], WZ[O], 3); }
                if constexpr (P == 4 || P == 5) xform_b_rows(RB[O], 0 + (P - 4));
                if constexpr (P >= 10 && P <= 13) xform_b_col(VX[O], VY[O], P - 10);
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{});  step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});  step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{});  step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{});  step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{});  step(std::integral_constant<int, 9>{});
            step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
            step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
        };
        auto chunk_body = [&](auto par_c, int cur) {
            kstep(std::integral_constant<int, 0>{}, par_c, cur);
            kstep(std::integral_constant<int, 1>{}, par_c, cur);
            kstep(std::integral_constant<int, 2>{}, par_c, cur);
            kstep(std::integral_constant<int, 3>{}, par_c, cur);
            kstep(std::integral_constant<int, 4>{}, par_c, cur);
            kstep(std::integral_constant<int, 5>{}, par_c, cur);
            kstep(std::integral_constant<int, 6>{}, par_c, cur);
            if constexpr (KSN == 8) kstep(std::integral_constant<int, 7>{}, par_c, cur);
        };
        if constexpr (KSN & 1) {
            // (an odd chunk count runs one empty chunk: its stage was filled by an all-invalid fetch, i.e. with zeros)
            for (int chunk = 0; chunk < nC; chunk += 2) {
                chunk_body(std::integral_constant<int, 0>{}, 0);
                chunk_body(std::integral_constant<int, 1>{}, 1);
            }
        } else {
            for (int chunk = 0; chunk < nC; ++chunk) chunk_body(std::integral_constant<int, 0>{}, chunk & 1);
        }
        wino_vmwait0();           // the last (empty) fetches must have landed before the workgroup gives up its LDS
    }
    // ---- partial db: [split][co], from the workgroups of ci tile 0 (every ci tile sees the same dY)
    bsum += __shfl_xor(bsum, 32);
    if (cit == 0 && wn == 0 && kh == 0 && co0 + wm * 32 + nl < Cout) bpartial[(size_t)split * Cout + co0 + wm * 32 + nl] = bsum;
    // ---- partial dU' (signs applied by the reduction): [split][position][co][ci]
    const int ci = ci0 + wn * 32 + nl;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        float* dst = partial + ((size_t)split * 16 + p) * Cout * Cin;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (co < Cout && ci < Cin) dst[(size_t)co * Cin + ci] = acc[p][r];
        }
    }
}

// dW[co][ci][3][3] (+)= G^T ( sum_splits dU'[split] (.) S ) G,  S_ij = s_i s_j, s = (1, 1, 1, -1)
__global__ void wino_wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial,
                                         float* __restrict__ dw, float* __restrict__ db, int Cout, int Cin, int S, int accumulate)
{
    const int64_t cc = (int64_t)Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cc; i += (int64_t)gridDim.x * blockDim.x) {
        if (db && i < Cout) {
            float sum = 0.f;
            for (int s = 0; s < S; ++s) sum += bpartial[(size_t)s * Cout + i];
            db[i] = accumulate ? db[i] + sum : sum;
        }
        float u[4][4];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            float sum = 0.f;
            for (int s = 0; s < S; ++s) sum += partial[((size_t)s * 16 + p) * cc + i];
            const bool neg = ((p >> 2) == 3) != ((p & 3) == 3);
            u[p >> 2][p & 3] = neg ? -sum : sum;
        }
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
            t[1][j] = 0.5f * (u[1][j] - u[2][j]);
            t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float g0 = t[k][0] + 0.5f * (t[k][1] + t[k][2]);
            const float g1 = 0.5f * (t[k][1] - t[k][2]);
            const float g2 = 0.5f * (t[k][1] + t[k][2]) + t[k][3];
            float* o = dw + i * 9 + k * 3;
            o[0] = accumulate ? o[0] + g0 : g0;
            o[1] = accumulate ? o[1] + g1 : g1;
            o[2] = accumulate ? o[2] + g2 : g2;
        }
    }
}

// k-steps per chunk: the choice with fewer (padded) k-steps per tile row
static int wino_wgrad_ksn(int w)
{
    const int pairs = cdiv(cdiv(w, 2), 2);
    return cdiv(pairs, 7) * 7 < cdiv(pairs, 8) * 8 ? 7 : 8;
}

// splits per (co tile, ci tile): fill the chip's 256 one-workgroup-per-CU slots a whole number of times
// (waves: how many such fills -- see ptmi_conv3x3_wino_wgrad_waves)
static int wino_wgrad_splits(int n, int cin, int cout, int h, int w, int waves)
{
    const int pairs = cdiv(cout, GWC) * cdiv(cin, GWC);
    const int64_t chunks = (int64_t)n * cdiv(h, 2) * cdiv(w, 4 * wino_wgrad_ksn(w));
    int S = cdiv(256 * (waves < 1 ? 1 : waves > 16 ? 16 : waves), pairs);
    if (S > chunks) S = (int)chunks;
    return S < 1 ? 1 : S;
}

}  // namespace

extern "C" {

int64_t ptmi_conv3x3_wino_packed_floats(int cin, int cout)
{
    return (int64_t)cdiv(cout, WBM) * cdiv(cin, WKC) * WUS;
}

int ptmi_conv3x3_wino_pack_weights(const float* w, float* wp, int w_cout, int w_cin, int mode, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(w && wp && w_cout > 0 && w_cin > 0, "conv3x3_wino_pack_weights: bad args");
    const int convCout = mode ? w_cin : w_cout, convCin = mode ? w_cout : w_cin;
    const int coTiles = cdiv(convCout, WBM), nChunks = cdiv(convCin, WKC);
    const int64_t total = (int64_t)coTiles * nChunks * WKC * WBM;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(wino_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, wp, w_cout, w_cin, mode,
                       coTiles, nChunks);
    PTMI_LAUNCH_CHECK("conv3x3_wino_pack_weights");
    return 0;
}

int ptmi_conv3x3_wino_fwd_fits(int cin, int cout, int h, int w)
{
    if (cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return 0;
    // a workgroup's 32 flat columns may reach into the strips of later images: per-lane offsets are relative to the first one
    const int64_t img_span = 32 / ((w + 4) & ~3) + 2;
    return (img_span * cin + WKC) * h * w * 4 < (1ll << 32) && (img_span * cout + WBM) * h * w * 4 < (1ll << 32);
}

int ptmi_conv3x3_wino_wgrad_fits(int h, int w)
{
    return h > 0 && w > 0 && (int64_t)(GWC + 1) * h * w * 4 < (1ll << 31);
}

int ptmi_conv3x3_wino_fwd(const float* x, const float* wp, const float* bias, const float* mask_ref, float* y, int n,
                          int cin, int cout, int h, int w, int epilogue, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && wp && y && n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "conv3x3_wino_fwd: bad args");
    PTMI_CHECK_ARG(epilogue >= 0 && epilogue <= 4, "conv3x3_wino_fwd: bad epilogue %d", epilogue);
    PTMI_CHECK_ARG(ptmi_conv3x3_wino_fwd_fits(cin, cout, h, w),
                   "conv3x3_wino_fwd: image too large for 32-bit buffer offsets (n=%d cin=%d cout=%d h=%d w=%d)", n, cin,
                   cout, h, w);
    PTMI_CHECK_ARG(epilogue > 1 || bias, "conv3x3_wino_fwd: bias required for epilogue %d", epilogue);
    PTMI_CHECK_ARG(epilogue != 4 || bias, "conv3x3_wino_fwd: bias required for epilogue 4");
    PTMI_CHECK_ARG(epilogue != 3 || mask_ref, "conv3x3_wino_fwd: mask_ref required for epilogue 3");
    const int bands = cdiv(h, WTH), coTiles = cdiv(cout, WBM), nChunks = cdiv(cin, WKC);
    const int period = (w + 1 + 3) & ~3;                     // strip length: W + at least one zero column, a multiple of 4
    const int64_t nPix = cdiv64((int64_t)n * bands * period, WTW);
    PTMI_CHECK_ARG(nPix * WTW < (1ll << 31), "conv3x3_wino_fwd: too many tiles");
    const int colocate = WINO_COLOCATE_MAX_COTILES >= coTiles;
    const int64_t nWg = colocate ? cdiv64(nPix, 8) * 8 * coTiles : nPix * coTiles;
    PTMI_CHECK_ARG(nWg < (1ll << 31), "conv3x3_wino_fwd: too many tiles");
    if (cin & (WKC - 1))
        hipLaunchKernelGGL(conv3x3_wino_kernel<true>, dim3((unsigned)nWg), dim3(WNT), 0, (hipStream_t)s, x, wp, bias, mask_ref, y, n,
                           cin, cout, h, w, nChunks, epilogue, coTiles, bands, period, (int)nPix, colocate);
    else
        hipLaunchKernelGGL(conv3x3_wino_kernel<false>, dim3((unsigned)nWg), dim3(WNT), 0, (hipStream_t)s, x, wp, bias, mask_ref, y, n,
                           cin, cout, h, w, nChunks, epilogue, coTiles, bands, period, (int)nPix, colocate);
    PTMI_LAUNCH_CHECK("conv3x3_wino_fwd");
    return 0;
}


int64_t ptmi_conv3x3_wino_wgrad_ws_floats_waves(int n, int cin, int cout, int h, int w, int waves)
{
    return (int64_t)wino_wgrad_splits(n, cin, cout, h, w, waves) * (16 * (int64_t)cout * cin + cout);
}

int64_t ptmi_conv3x3_wino_wgrad_ws_floats(int n, int cin, int cout, int h, int w)
{
    return ptmi_conv3x3_wino_wgrad_ws_floats_waves(n, cin, cout, h, w, 1);
}

int ptmi_conv3x3_wino_wgrad(const float* x, const float* dy, float* dw, float* db, float* ws, int n, int cin, int cout, int h,
                            int w, int accumulate, ptmi_stream_t s)
{
    return ptmi_conv3x3_wino_wgrad_waves(x, dy, dw, db, ws, n, cin, cout, h, w, accumulate, 1, s);
}

int ptmi_conv3x3_wino_wgrad_waves(const float* x, const float* dy, float* dw, float* db, float* ws, int n, int cin, int cout, int h,
                                  int w, int accumulate, int waves, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && dy && dw && ws && n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "conv3x3_wino_wgrad: bad args");
    PTMI_CHECK_ARG(ptmi_conv3x3_wino_wgrad_fits(h, w), "conv3x3_wino_wgrad: map %dx%d too large for 32-bit buffer offsets", h, w);
    const int S = wino_wgrad_splits(n, cin, cout, h, w, waves);
    const int coTiles = cdiv(cout, GWC), ciTiles = cdiv(cin, GWC);
    hipStream_t st = (hipStream_t)s;
    float* bws = ws + (size_t)S * 16 * cout * cin;
    if (wino_wgrad_ksn(w) == 7)
        hipLaunchKernelGGL(conv3x3_wino_wgrad_kernel<7>, dim3((unsigned)(coTiles * ciTiles * S)), dim3(WNT), 0, st, x, dy, ws, bws, n, cin, cout,
                           h, w, ciTiles, S, cdiv(h, 2), cdiv(w, 28));
    else
        hipLaunchKernelGGL(conv3x3_wino_wgrad_kernel<8>, dim3((unsigned)(coTiles * ciTiles * S)), dim3(WNT), 0, st, x, dy, ws, bws, n, cin, cout,
                           h, w, ciTiles, S, cdiv(h, 2), cdiv(w, 32));
    PTMI_LAUNCH_CHECK("conv3x3_wino_wgrad");
    const int64_t cc = (int64_t)cout * cin;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, st, ws, bws, dw, db, cout, cin, S, accumulate);
    PTMI_LAUNCH_CHECK("conv3x3_wino_wgrad_reduce");
    return 0;
}

}  // extern "C"
