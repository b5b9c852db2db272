// wino4p.hip -- the fused Winograd F(4x4, 3x3) convolution of wino4.hip with the 36 transform-domain POSITIONS SPLIT over the two
// waves of a tile row (round 6): forward and dgrad of the fp32 3x3 layers with >= 64 input channels.
//
// Replaces the cuDNN conv2d (+bias, +ReLU) the reference reaches at pt/modeling/backbone/vgg.py:45-53,66-69 and the 3x3 conv of
// D2's StandardRPNHead (pt/modeling/proposal_generator/rpn.py:96) -- same ABI contract, epilogues and dgrad convention as
// ptmi_conv3x3_wino4p_* (include/ptmi355.h); same points (0, +-3/4, +-3/2, inf), same patch / slab stages, DMA pattern, persistent
// workgroups and tile schedules (static walk / XCD work queues).  What changes is who computes what inside a workgroup:
//   * wino4.hip: wave (wm, wn) = 32 output channels x tile row wn x ALL 36 positions: the two channel halves of a tile row
//     transform the SAME windows -- 144 transform FMAs per lane and chunk beside 72 MFMAs, and an fp32 MFMA shares the VALU's FMA
//     lanes: every FMA is issue time added to the MFMA time;
//   * here: wave (wp, wn) = ALL 64 output channels x tile row wn x 18 positions -- transform rows (0, +a, -a) for wp = 0, rows
//     (inf, +b, -b) for wp = 1.  A lane transforms only its three rows: vertical first (5 of the 6 window rows are read: rows
//     0..4 or 1..5; 6 operations per column = 36), horizontal after (3 rows x 12 = 36): 72 FMAs instead of 144, 15 window reads
//     instead of 18, each B operand feeds FOUR MFMAs (the four 16-channel tiles).  Accumulators: 18 positions x 4 channel tiles =
//     72 tiles of 16x16 = 288 registers, as before.
//   * the price: Y = A^T M A needs all six rows of M.  Each wave reduces ITS three rows to a partial 4x4 output per channel
//     (Y = sum_i A^T[:, i] (M[i, :] A): the row sum splits), keeps the partials of 32 channels, hands the other 32 to its partner
//     wave through LDS (the slab stage the tile's last chunk has left free: 16 floats per lane and channel, double buffered,
//     one workgroup barrier per channel) and adds the partner's partials to the ones it kept -- 8 barriers and 64 16-byte LDS
//     accesses per lane and tile.
//   * the two halves run ONE code path: the 1-D transforms of rows (+-a) and (+-b) differ by two constants (wave-uniform SGPR
//     operands); the rows 0 / inf are the same expression on window rows shifted by one (a2b2 d0 - s2 d2 + d4 and
//     a2b2 d1 - s2 d3 + d5); only those 12 FMAs sit behind a scalar branch.  The weight slab is packed per half:
//     [ci 4][half 2][4 groups of 4 positions + 1 group of 2][co 64] -- 16-byte A reads for the first 16 positions, 8-byte for the
//     last 2; wave wp = 1 addresses the channel tiles in the order (2, 3, 0, 1), so that "keep tiles 0, 1 / send tiles 2, 3" is the
//     same code in both waves.

#include "common.h"
#include <type_traits>
#include <utility>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 xlds_f32x4_t;
typedef __attribute__((address_space(3))) float xlds_f32_t;
typedef __attribute__((address_space(3))) void xlds_void_t;

constexpr int XKC = 4;                  // input channels per chunk = K of one MFMA
constexpr int XBM = 64;                 // output channels per workgroup
constexpr int XTH = 8, XTW = 64;        // output pixels per workgroup: 8 rows x 64 flat columns = 2 x 16 tiles of 4x4
constexpr int XPP = 72;                 // patch row pitch in floats: 18 pieces, LDS column c <-> flat column u0 - 4 + c
constexpr int XPR = XTH + 2;            // patch rows (image rows y0-1 .. y0+8)
constexpr int XPL = XPR * XPP;          // floats per channel plane (720)
constexpr int XUS = XKC * 9 * XBM * 4;  // U floats per chunk: [ci 4][half 2][group 5: 4 x (co 64 x 4 positions) + (co 64 x 2)] = 9216 (36 KB)
constexpr int XUC = 9 * XBM * 4;        // ... per input channel (2304), XUH per half (1152); the 2-position group starts at 1024
constexpr int XUH = XUC / 2;
constexpr int XPS = XKC * XPL;          // patch floats per chunk: 2880 = 720 pieces
constexpr int XPSP = 3072;              // ... padded to 3 DMA instructions per lane (pieces 720 .. 767 carry offset 0xFFFFFFFF)
constexpr int XNT = 256;
constexpr int XUI = XUS / 4 / XNT;      // 9 U DMA instructions per lane and chunk
constexpr int XPI = XPSP / 4 / XNT;     // 3 patch DMA instructions per lane and chunk
constexpr int XDI = XUI + XPI;          // 12
constexpr int XNU = 3, XNP = 4;         // stages
constexpr int XRUN = 32;                // dynamic schedule: pixel tiles per channel-tile run of a queue (= the workgroups of one XCD)
constexpr int XLDS = XNU * XUS + XNP * XPSP;   // 39936 floats = 159744 B

// transform constants (a = 3/4, b = 3/2)
constexpr float XA = 0.75f, XB = 1.5f, XA2 = 0.5625f, XB2 = 2.25f, XA3 = 0.421875f, XB3 = 3.375f;
constexpr float XA2B2 = 1.265625f, XS2 = 2.8125f;      // a^2 b^2, a^2 + b^2

// r = c * x + y / r = -c * x + y: explicit FMAs (the file is built with -fno-slp-vectorize: the SLP vectoriser otherwise builds
// v_pk_fma_f32 out of register shuffles -- slower than two scalar FMAs next to MFMAs on this part; inline-asm FMAs cost a
// compiler-inserted s_nop after every dependent pair)
__device__ __forceinline__ float xfma(float c, float x, float y) { return __builtin_fmaf(c, x, y); }
__device__ __forceinline__ float xfnma(float c, float x, float y) { return __builtin_fmaf(-c, x, y); }
__device__ __forceinline__ float xadd(float x, float y) { return x + y; }
__device__ __forceinline__ float xsub(float x, float y) { return x - y; }
__device__ __forceinline__ float xmul(float c, float x) { return c * x; }

// 1-D input transform t = B^T d: operation k of 12 (so that a slot can carry any sub-range of them).  E[] are the four
// intermediates (even / odd parts at +-a and +-b).
// operations 0 .. 5 are independent of each other, 6 .. 11 depend only on 0 .. 5: no back-to-back dependent FMAs
template <int K>
__device__ __forceinline__ void xin_op(const float (&d)[6], float (&t)[6], float (&E)[4])
{
    if constexpr (K == 0) t[0] = xfnma(XS2, d[2], d[4]);
    if constexpr (K == 1) t[5] = xfnma(XS2, d[3], d[5]);
    if constexpr (K == 2) E[0] = xfnma(XB2, d[2], d[4]);        // even part at +-a
    if constexpr (K == 3) E[1] = xfnma(XB2, d[1], d[3]);        // odd part at +-a (before the factor a)
    if constexpr (K == 4) E[2] = xfnma(XA2, d[2], d[4]);
    if constexpr (K == 5) E[3] = xfnma(XA2, d[1], d[3]);
    if constexpr (K == 6) t[0] = xfma(XA2B2, d[0], t[0]);
    if constexpr (K == 7) t[5] = xfma(XA2B2, d[1], t[5]);
    if constexpr (K == 8) t[1] = xfma(XA, E[1], E[0]);
    if constexpr (K == 9) t[2] = xfnma(XA, E[1], E[0]);
    if constexpr (K == 10) t[3] = xfma(XB, E[3], E[2]);
    if constexpr (K == 11) t[4] = xfnma(XB, E[3], E[2]);
}

// 1-D output transform y = A^T m (12 operations): y_k = sum_i p_i^k m_i (+ m_5 for k = 3)
__device__ __forceinline__ void xout(const float (&m)[6], float (&y)[4])
{
    const float s1 = xadd(m[1], m[2]), d1 = xsub(m[1], m[2]), s2 = xadd(m[3], m[4]), d2 = xsub(m[3], m[4]);
    y[0] = xadd(xadd(m[0], s1), s2);
    y[1] = xfma(XB, d2, xmul(XA, d1));
    y[2] = xfma(XB2, s2, xmul(XA2, s1));
    y[3] = xfma(XB3, d2, xfma(XA3, d1, m[5]));
}

// the MFMAs: accumulator tile in AGPRs ("a") or VGPRs ("v")
__device__ __forceinline__ void xmfma_a(f32x4& c, float a, float b) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void xmfma_v(f32x4& c, float a, float b) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
// the same with C = 0 (a tile's FIRST chunk: every accumulator tile is written exactly once per chunk, so the tile needs no zeroing pass)
__device__ __forceinline__ void xmfma_a0(f32x4& c, float a, float b) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void xmfma_v0(f32x4& c, float a, float b) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b)); }

template <int... I, class F>
__device__ __forceinline__ void xfor(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// ---- the chunk schedule (compile-time tables).  72 MFMAs = 20 A-operand groups: group hg < 16 = (position group g = hg >> 2,
// channel tile ct = hg & 3), four MFMAs (positions 4 g + q); groups 16 .. 19 = (g = 4: positions 16, 17; ct = hg - 16), two MFMAs
__host__ __device__ constexpr int y_hg(int s) { return s < 64 ? s >> 2 : 16 + ((s - 64) >> 1); }
__host__ __device__ constexpr int y_q(int s) { return s < 64 ? s & 3 : (s - 64) & 1; }
__host__ __device__ constexpr bool x_is_aread(int s) { return y_q(s) == 1; }           // group hg + 2 is read during group hg
constexpr int XHAND = 68;                                   // hand-over slot: after the chunk's last A read (slot 67: group 19), before the
                                                             // next chunk's first (slot 69: its group 0)
__host__ __device__ constexpr bool x_is_dma(int s) { return s % 6 == 5; }              // 12 DMA instructions spread evenly (as wino4.hip)
__host__ __device__ constexpr int x_dma_at(int s) { return x_is_dma(s) ? s / 6 : -1; }
__host__ __device__ constexpr int x_dma_before(int s) { return (s + 0) / 6; }          // DMA instructions of this chunk issued before slot s
// the k-th slot (k = 0 ..) that carries neither an A read nor a DMA instruction nor the hand-over, from slot 2 on
__host__ __device__ constexpr int x_free_slot(int k)
{
    int s = 2;
    for (;; ++s) {
        if (x_is_aread(s) || x_is_dma(s) || s == XHAND) continue;
        if (k-- == 0) return s;
    }
}
// window read r (0 .. 14: row r / 3 of the FIVE rows a half needs, part r % 3) sits in the r-th free slot (slots 2 .. 22)
__host__ __device__ constexpr int x_wread_at(int s)
{
    for (int r = 0; r < 15; ++r)
        if (x_free_slot(r) == s) return r;
    return -1;
}
// the 72 transform FMAs: 0 .. 11 the rows 0 / inf (one burst behind the scalar branch on the half, slot XFMA0), then three per slot in
// the following slots that carry neither a DMA instruction nor the hand-over: 12 .. 35 the rows +-a / +-b (vertical), 36 .. 71 horizontal
constexpr int XFMA0 = 26;                                   // (the last window read sits in slot 22)
__host__ __device__ constexpr int x_valu_before(int s)
{
    if (s <= XFMA0) return 0;
    int n = 12;
    for (int t = XFMA0 + 1; t < s && t < 72; ++t)
        if (!x_is_dma(t) && t != XHAND) n += 3;
    return n > 72 ? 72 : n;
}
static_assert(x_free_slot(14) + 3 < XFMA0 && x_valu_before(XHAND) == 72, "wino4p chunk schedule");

// DYN: the dynamic tile schedule (sched != nullptr).  Two instantiations: the schedule's scalar state (queue, pending draw, ids)
// costs the chunk loop 2.6 % through SGPR pressure (48 more scalar instructions per chunk pair between the MFMAs: measured on the
// same box against the round-5 kernel, profiles/r06_wino4_static_vs_dynamic.txt), which a single-GPU run need not pay
template <bool DYN>
__global__ __launch_bounds__(XNT, 1) void conv3x3_wino4p_kernel(
    const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias,
    const float* __restrict__ mref, float* __restrict__ y, int N, int Cin, int Cout, int H, int W, int nChunks, int epi,
    int coTiles, int bands, int period, int nPix, int colocate, int nTiles, int* __restrict__ sched)
{
    // ONE LDS object (a second __shared__ variable makes the LDS lowering attach alias scopes to every access, and the waitcnt pass
    // then protects each window / A read against the LDS-DMA instructions with s_waitcnt vmcnt(0): measured as 40 extra waits per
    // chunk pair in the ISA); the dynamic schedule's three words sit behind the stages
    __shared__ __attribute__((aligned(16))) float lds[XLDS + 8];
    typedef __attribute__((address_space(3))) int xlds_int_t;
    volatile xlds_int_t* const sched_ids = (volatile xlds_int_t*)(lds + XLDS);   // [0], [1] the workgroup's first two tile ids, [2] the id after the next,
                                                             // [3] the id of the tile the slab cursor is in, [4] / [5] wave 0's queue state

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * W;
    const int nStrips = N * bands;
    const int grid = gridDim.x;
    // PERSISTENT workgroups (one per CU: the LDS admits no second): workgroup b walks the tile ids b, b + grid, ...; the chunk
    // stream does not stop at a tile boundary -- "the next chunk" of a tile's last chunk is the first chunk of the workgroup's
    // next tile (its DMA, hand-over, window reads, transform, first A reads), so only the epilogue separates the MFMA streams of
    // two tiles, and its stores drain while the next tile computes.  (One tile per workgroup: 32 k cycles per tile outside the
    // chunk loop -- launch, 27 DMA issues + their latency, the first transform, the store burst of all CUs in lock step; 13 % of
    // conv3_2, 35 % of conv1_2 -- tools/exp/wino4_bench.py --custom, intercept of time over chunk count.)
    // tile id -> (channel tile, pixel tile): as wino.hip (colocate: the channel tiles of a pixel tile back to back on one XCD --
    // grid is a multiple of 8, so id mod 8 = the workgroup's XCD for all of its tiles); ids beyond the last pixel tile: the end
    auto decode = [&](int vid, int& cot, int& pix) __attribute__((always_inline)) {
        cot = 0; pix = 0;
        if (vid >= nTiles) return false;
        if (DYN && colocate) {
            // dynamic schedule, <= 4 channel tiles: queue q = vid & 7 owns the pixel tiles = q mod 8 (Gq of them); its index k = vid >> 3
            // walks them in RUNS of XRUN pixel tiles per channel tile -- the ~32 workgroups of an XCD then work on ONE weight slab
            // at a time (0.6 - 2.4 MB: L2-resident) instead of all of them (the static walk keeps 4 slabs = up to 9.4 MB live per 4-MB
            // L2).  The last, shorter cycle uses runs of the remaining length, so the valid indices of a queue are a PREFIX of it
            const int q = vid & 7, k = vid >> 3;
            const int Gq = (nPix - q + 7) >> 3;                      // pixel tiles of this queue
            if (k >= Gq * coTiles) return false;
            const int full = Gq / XRUN;                              // whole cycles of XRUN pixel tiles x coTiles
            const int c = k / (XRUN * coTiles);
            int g;
            if (c < full) {
                const int rem = k - c * (XRUN * coTiles);
                cot = rem / XRUN;
                g = c * XRUN + (rem - cot * XRUN);
            } else {
                const int rt = Gq - full * XRUN, kk = k - full * (XRUN * coTiles);
                cot = kk / rt;
                g = full * XRUN + (kk - cot * rt);
            }
            pix = g * 8 + q;
            return true;
        }
        if (colocate) {
            const int slot = vid >> 3;
            cot = slot % coTiles;
            pix = (slot / coTiles) * 8 + (vid & 7);
            return pix < nPix;
        }
        cot = vid % coTiles;
        pix = vid / coTiles;
        return true;
    };

    // ---- tile schedule.  sched == nullptr: STATIC -- workgroup b walks b, b + grid, ... (a workgroup that starts late -- its CU
    // was held by another kernel, e.g. a collective's -- still runs its whole share after the others have finished: up to 2x on the
    // launch).  sched != nullptr: DYNAMIC -- eight queues over the SAME enumeration (queue q = the ids = q mod 8, so a queue keeps
    // the XCD affinity of the static walk: one weight slab per XCD with 8 channel tiles, the channel tiles of a pixel tile back to
    // back with <= 4); a workgroup draws from the queue of the XCD it runs on (HW_REG_XCC_ID) and, once that is exhausted, from the
    // following ones.  sched[q] = next index of queue q, sched[8] = workgroups finished; the last one out zeroes all nine, so the
    // buffer is zero again when the launch ends (one buffer per stream: launches on one stream do not overlap).
    // Only wave 0 draws (lane 0 issues the atomic); ids reach the other waves through sched_ids[] behind a barrier.  ids are
    // drawn one tile ahead of make_next(), i.e. two tiles ahead of the MFMAs: no wave ever waits for an atomic in the steady state.
    constexpr bool dyn = DYN;
    // The schedule's state lives in LDS, not in SGPRs (a first version kept queue, flags and ids in scalar registers: 48 more scalar
    // instructions per chunk pair between the MFMAs -- spill reloads -- and a 4 % slower chunk loop): sched_ids[4] = the queue
    // wave 0 draws from, [5] = queues not yet seen exhausted; every tile draws exactly once (where its slab cursor wraps) and resolves
    // the draw at its end, so no flags are needed
    int pend_k = 0;                                          // (wave 0, lane 0) the unresolved draw's queue index
    auto id_ok = [&](int v) __attribute__((always_inline)) { int c_, p_; return decode(v, c_, p_); };
    auto draw_add = [&](int q, int cnt) __attribute__((always_inline)) {   // -> queue q's index before the add (wave-uniform)
        int k = 0;
        if (lane == 0) k = __hip_atomic_fetch_add(sched + q, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_amdgcn_readfirstlane(k);
    };
    // a valid id for index k of queue sq -- or, the queue being exhausted there, the first valid id of the following queues;
    // nTiles = everything has been handed out (a queue's valid ids are a prefix of it: once exhausted, always exhausted)
    auto resolve = [&](int k, int& sq, int& sleft) __attribute__((always_inline)) {
        for (;;) {
            if (sleft == 0) return nTiles;
            const long long v = 8ll * k + sq;
            if (v < nTiles && id_ok((int)v)) return (int)v;
            sq = (sq + 1) & 7;
            --sleft;
            if (sleft) k = draw_add(sq, 1);
        }
    };

    // ---- DMA-side state of a tile: the lane's patch pieces (channel, patch row, 16-B piece) -> byte offset from the chunk's first
    // plane of the tile's first image; periods are multiples of 4, so a piece lies in ONE strip
    unsigned n_pvoff[XPI];                                   // the NEXT tile's (the DMA cursors switch to it when they wrap)
    int n_fix;
    bool n_edge;
    const char* n_xc;
    unsigned n_xleft;
    const float* n_slab;
    int n_urange;
    int n_vid;
    auto make_next = [&](int vid) __attribute__((always_inline)) {
        int cot, pix;
        const bool valid = decode(vid, cot, pix);
        n_vid = vid;
        const int u0 = pix * XTW;                            // flat column of the tile's first output column
        const int n = (u0 / period) / bands;                 // image of the first tile column: the address base
        const int s0 = (u0 > 4 ? u0 - 4 : 0) / period;       // strip / image / band of the patch's first column
        const int n0 = s0 / bands, b0 = s0 - n0 * bands;
        n_fix = 0;                                           // words 1..3 of piece i (bits 4i+1 .. 4i+3) beyond the image edge
#pragma unroll
        for (int i = 0; i < XPI; ++i) {
            const int pidx = tid + i * XNT;
            const int ci = pidx / (XPR * 18), rem = pidx - ci * (XPR * 18);
            const int r = rem / 18, q = rem - r * 18;
            int gx = u0 - 4 + 4 * q - s0 * period, band = b0, sn = n0;   // gx < 0 only in the very first patch (u0 = 0, q = 0)
            while (gx >= period) { gx -= period; ++band; }
            while (band >= bands) { band -= bands; ++sn; }
            const int gy = band * XTH - 1 + r;
            n_pvoff[i] = 0xFFFFFFFFu;
            if (valid && pidx < XPS / 4 && gx >= 0 && sn < N && gy >= 0 && gy < H && gx < W) {
                n_pvoff[i] = (unsigned)(((sn - n) * Cin + ci) * HW + gy * W + gx) * 4u;
#pragma unroll
                for (int e = 1; e < 4; ++e) n_fix |= (gx + e >= W) ? (1 << (4 * i + e)) : 0;
            }
        }
        n_edge = false;                                      // some loaded piece may straddle the right edge of an image row
        if (valid && (W & 3)) {
            const int e0 = W & ~3;
            for (int st = s0; st < nStrips && st * period + e0 < u0 + XPP - 4; ++st)
                n_edge |= st * period + e0 >= u0 - 4;
        }
        n_xc = (const char*)(x + (size_t)(valid ? n : 0) * Cin * HW);
        // bytes from xc to the end of the tensor, clamped (offsets reach into the next image: (span Cin + 4) HW 4 < 2^32, launcher)
        const long long xtail = valid ? (long long)(N - n) * Cin * HW * 4 : 0ll;
        n_xleft = (unsigned)(xtail > 0xFFFFFFFEll ? 0xFFFFFFFEll : xtail);
        n_slab = wpk + (size_t)cot * nChunks * XUS;
        n_urange = valid ? nChunks * XUS * 4 : 0;            // no tile: every piece is zero fill
    };
    // the cursors: patch pieces of chunk pcur / slab pieces of chunk ucur of the tile they are in
    unsigned pvoff[XPI];
    int fix;
    bool edge;
    const char* xc;
    unsigned xleft;
    const float* slab;
    int urange;
    unsigned wv;                                             // byte offset of the lane's piece of slab chunk ucur (the range check covers
    int pcur, ucur;                                          // voffset only: beyond the slab = zero fill)
    auto patch_wrap = [&]() __attribute__((always_inline)) {
        if (pcur == nChunks) {
#pragma unroll
            for (int i = 0; i < XPI; ++i) pvoff[i] = n_pvoff[i];
            fix = n_fix; edge = n_edge; xc = n_xc; xleft = n_xleft;
            pcur = 0;
        }
    };
    auto slab_wrap = [&]() __attribute__((always_inline)) {   // (always one chunk after the patch cursor wrapped: then the NEXT state is free)
        if (ucur == nChunks) {
            slab = n_slab; urange = n_urange; wv = (unsigned)tid * 16u;
            ucur = 0;
            int nid = n_vid + grid;
            if constexpr (DYN) {                             // (the slot: written by wave 0 at least one barrier ago)
                nid = __builtin_amdgcn_readfirstlane(sched_ids[2]);
                if (wave == 0) {
                    // the tile the cursor enters (= the next tile when this one ends), and the draw for the tile after `nid`: resolved
                    // and published at this tile's end.  The atomic is older than this chunk's DMA instructions, so the counted
                    // wait of this chunk's hand-over covers it (vector-memory results return in order) -- a chunk later
                    const int q = __builtin_amdgcn_readfirstlane(sched_ids[4]);
                    const int left = __builtin_amdgcn_readfirstlane(sched_ids[5]);
                    if (lane == 0) {
                        sched_ids[3] = n_vid;
                        if (left) pend_k = __hip_atomic_fetch_add(sched + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            make_next(nid);
        }
    };

    float* const ldsU = lds;
    float* const ldsP = lds + XNU * XUS;
    auto dma_patch = [&](int i, int stage /* float offset of the stage */) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            ptmi_uniform_ptr(xc), 0, __builtin_amdgcn_readfirstlane((int)xleft), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (xlds_void_t*)(ldsP + stage + wave * 256 + i * XNT * 4), 16,
                                                 (int)pvoff[i], 0, 0, 0);
    };
    auto dma_u = [&](int i, int stage) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            ptmi_uniform_ptr(slab), 0, __builtin_amdgcn_readfirstlane(urange), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (xlds_void_t*)(ldsU + stage + wave * 256 + i * XNT * 4), 16,
                                                 (int)wv, i * XNT * 16, 0, 0);
    };
    auto advance_patch = [&]() __attribute__((always_inline)) {
        xc += (size_t)XKC * HW * 4;
        xleft = xleft == 0xFFFFFFFEu ? xleft : (xleft > (unsigned)XKC * (unsigned)HW * 4u ? xleft - (unsigned)XKC * (unsigned)HW * 4u : 0u);
        ++pcur;
    };
    auto advance_u = [&]() __attribute__((always_inline)) { wv += XUS * 4; ++ucur; };
    int fix_ho;                                              // fix-up mask / edge flag of the patch the NEXT hand-over completes (the
    bool edge_ho;                                            // one issued during the previous chunk)
    auto fixup = [&](int stage, int fx, bool ed) __attribute__((always_inline)) {
        if (ed && fx) {
            float* pw = ldsP + stage + tid * 4;
#pragma unroll
            for (int i = 0; i < XPI; ++i) {
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (fx & (1 << (4 * i + e))) pw[i * XNT * 4 + e] = 0.f;
            }
        }
    };

    // ---- the lane's role in the MFMAs (the same for every tile)
    const int wp = wave >> 1, wn = wave & 1;                 // position half (also: the 32 channels the wave finalises) / tile row
    const int ttx = lane & 15, kq = lane >> 4;               // tile column / input channel of the chunk
    auto tile_geometry = [&](int u0v, int& tsn_o, int& py_o, int& px_o) __attribute__((always_inline)) {
        const int sf = u0v / period;
        int px_t = u0v + 4 * ttx - sf * period, sn = sf / bands, band = sf - sn * bands;
        while (px_t >= period) { px_t -= period; ++band; }
        while (band >= bands) { band -= bands; ++sn; }
        tsn_o = sn;
        py_o = band * XTH + wn * 4;
        px_o = px_t;
        return sn < N && px_o < W && py_o < H;
    };
    // LDS float offsets of the lane.  A = U[kq][half wp][group g][co][..]; the wave's LOCAL channel tile c' is the real tile c' ^ 2 wp
    // (wave 1 walks the tiles in the order 2, 3, 0, 1): local tiles 0, 1 = the 32 channels the wave keeps, through the "lo" bases,
    // local tiles 2, 3 = the 32 it hands to its partner, through the "hi" bases.  window = patch[kq][4 wn + row][4 ttx + 3 ..]
    const int a_lo = kq * XUC + wp * XUH + (wp * 32 + ttx) * 4;                            // + g * 256 + c' * 64        (16-byte groups)
    const int a_hi = kq * XUC + wp * XUH + ((1 - wp) * 32 + ttx) * 4;                      // + g * 256 + (c' - 2) * 64
    const int a2_lo = kq * XUC + wp * XUH + 1024 + (wp * 32 + ttx) * 2;                    // + c' * 32                  (the 8-byte group)
    const int a2_hi = kq * XUC + wp * XUH + 1024 + ((1 - wp) * 32 + ttx) * 2;              // + (c' - 2) * 32
    const int b_off = kq * XPL + (wn * 4) * XPP + 4 * ttx + 3;                             // + row * 72 + {0, 1 (b128), 5}
    const int e_row = wp ? 5 : 0;                                                         // the window row only this half reads
    // the half's constants: rows (0, +a, -a) or (inf, +b, -b)
    const float c_sq = wp ? XA2 : XB2, c_p = wp ? XB : XA;                                 // vertical input transform: E = d_hi - c_sq d_lo, t = E_e +- c_p E_o
    const float o_p1 = wp ? XB : XA, o_p2 = wp ? XB2 : XA2, o_p3 = wp ? XB3 : XA3;         // vertical output transform: p, p^2, p^3
    const float o_e0 = wp ? 0.f : 1.f, o_e3 = wp ? 1.f : 0.f;                              // ... the row 0 / inf term goes to output row 0 / 3

    int vid = blockIdx.x;
    int vid1 = vid + grid;                                   // the workgroup's second tile
    // the last workgroup out re-arms the schedule for the next launch on this stream
    auto sched_exit = [&]() __attribute__((always_inline)) {
        if (dyn && tid == 0) {
            const int done = __hip_atomic_fetch_add(sched + 8, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (done == grid - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) __hip_atomic_store(sched + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (dyn) {
        if (wave == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            int sq = (int)(xcc & 7u), sleft = 8;
            const int k = draw_add(sq, 3);                   // three ids in one round trip: this tile, the next, the one after
            // (sleft == 8 <=> still on the own queue, whose indices k .. k + 2 this workgroup owns; a later queue: a fresh draw each)
            const int a = resolve(k, sq, sleft);
            const int b = resolve(sleft == 8 ? k + 1 : (sleft ? draw_add(sq, 1) : 0), sq, sleft);
            const int c = resolve(sleft == 8 ? k + 2 : (sleft ? draw_add(sq, 1) : 0), sq, sleft);
            if (lane == 0) { sched_ids[4] = sq; sched_ids[5] = sleft; }
            if (lane == 0) { sched_ids[0] = a; sched_ids[1] = b; sched_ids[2] = c; }
        }
        __syncthreads();
        vid = __builtin_amdgcn_readfirstlane(sched_ids[0]);
        vid1 = __builtin_amdgcn_readfirstlane(sched_ids[1]);
    }
    {
        int c0, p0;
        if (!decode(vid, c0, p0)) {                          // (workgroup-uniform)
            sched_exit();
            return;
        }
    }
    // the lane's 8 biases of a tile (channels co_w + 16 ct + 4 kq + r), fetched at the head of the tile
    f32x4 bv[2];
    const __amdgpu_buffer_rsrc_t rbias = ptmi_rsrc(bias ? bias : y, bias ? (unsigned)Cout * 4u : 0u);
    auto load_bias = [&](int v) __attribute__((always_inline)) {
        int cot, pix;
        decode(v, cot, pix);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            bv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (epi <= 1 || epi == 4)
                bv[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, (cot * XBM + wp * 32 + ct * 16 + 4 * kq) * 4, 0, 0));
        }
    };
    load_bias(vid);

    // prologue DMA of the workgroup's first tile, in the order the counted waits assume: patch 0, U 0, patch 1 | U 1, patch 2
    make_next(vid);
    pcur = ucur = nChunks;                                   // "wrapped": the first issue of either kind switches to the state just made
    patch_wrap();
    {                                                        // slab_wrap() by hand: the second tile's id is vid1, the slot stays untouched
        slab = n_slab; urange = n_urange; wv = (unsigned)tid * 16u;
        ucur = 0;
        make_next(vid1);
    }
    auto issue_patch = [&](int stage) __attribute__((always_inline)) {
        patch_wrap();
#pragma unroll
        for (int i = 0; i < XPI; ++i) dma_patch(i, stage);
        advance_patch();
    };
    auto issue_u = [&](int stage) __attribute__((always_inline)) {
        slab_wrap();
#pragma unroll
        for (int i = 0; i < XUI; ++i) dma_u(i, stage);
        advance_u();
    };
    issue_patch(0);
    const int fix_p0 = fix; const bool edge_p0 = edge;
    issue_u(0);
    issue_patch(XPSP);
    const int fix_p1 = fix; const bool edge_p1 = edge;
    issue_u(XUS);
    issue_patch(2 * XPSP);
    fix_ho = fix; edge_ho = edge;

    f32x4 accA[64];                        // local positions 0 .. 15 (x 4 channel tiles): AGPRs; tile (p, c') = accA[4 p + c']
    f32x4 accV[8];                         // local positions 16, 17: VGPRs; accV[4 (p - 16) + c']
    float V0[3][6], V1[3][6];              // B operands of the current / next chunk: [local row][column]
    float RE[6], RW[4][6];                 // raw window rows of the next chunk: the half's own edge row (0 or 5) and rows 1 .. 4
    float T[3][6];                         // vertically transformed: local rows (edge, +p, -p)
    float Ev[6][2];                        // even / odd parts per column
    float E[4];
    f32x4 A[4];                            // A operands: twenty groups per chunk rotate through FOUR register quads (20 = 0 mod 4: the
                                           // rotation carries across chunks and tiles); group hg + 2 is read during group hg

    auto wread = [&](auto r_c, const float* pb, const float* pe) __attribute__((always_inline)) {   // window read r: row r / 3 (0: edge), part r % 3
        constexpr int a = decltype(r_c)::value / 3, part = decltype(r_c)::value % 3;
        const float* p = a == 0 ? pe : pb + a * XPP;
        float(&dst)[6] = *(a == 0 ? &RE : &RW[a == 0 ? 0 : a - 1]);
        if constexpr (part == 0) dst[0] = *(const volatile xlds_f32_t*)p;
        if constexpr (part == 1) {
            const f32x4 v = *(const volatile xlds_f32x4_t*)(p + 1);
            dst[1] = v[0]; dst[2] = v[1]; dst[3] = v[2]; dst[4] = v[3];
        }
        if constexpr (part == 2) dst[5] = *(const volatile xlds_f32_t*)(p + 5);
    };
    // transform operation k of 72 (x_valu_before): RW[0 .. 3] = window rows d1 .. d4, RE = d0 (half 0) / d5 (half 1)
    auto edge_ops = [&]() __attribute__((always_inline)) {                               // k = 0 .. 11, one burst
        if (wp == 0) {                                                                   // t0 = a2b2 d0 + (d4 - s2 d2)
            xfor(std::make_integer_sequence<int, 6>{}, [&](auto j_c) __attribute__((always_inline)) { constexpr int j = decltype(j_c)::value; T[0][j] = xfnma(XS2, RW[1][j], RW[3][j]); });
            xfor(std::make_integer_sequence<int, 6>{}, [&](auto j_c) __attribute__((always_inline)) { constexpr int j = decltype(j_c)::value; T[0][j] = xfma(XA2B2, RE[j], T[0][j]); });
        } else {                                                                         // t5 = a2b2 d1 + (d5 - s2 d3)
            xfor(std::make_integer_sequence<int, 6>{}, [&](auto j_c) __attribute__((always_inline)) { constexpr int j = decltype(j_c)::value; T[0][j] = xfnma(XS2, RW[2][j], RE[j]); });
            xfor(std::make_integer_sequence<int, 6>{}, [&](auto j_c) __attribute__((always_inline)) { constexpr int j = decltype(j_c)::value; T[0][j] = xfma(XA2B2, RW[0][j], T[0][j]); });
        }
    };
    auto mid_op = [&](auto k_c) __attribute__((always_inline)) {                         // k = 12 .. 35: 12 even / odd parts, then 12 results
        constexpr int k = decltype(k_c)::value - 12;
        if constexpr (k < 12) {
            constexpr int j = k >> 1;
            if constexpr ((k & 1) == 0) Ev[j][0] = xfnma(c_sq, RW[1][j], RW[3][j]);      // d4 - c d2
            else Ev[j][1] = xfnma(c_sq, RW[0][j], RW[2][j]);                             // d3 - c d1
        } else {
            constexpr int j = (k - 12) >> 1;
            if constexpr ((k & 1) == 0) T[1][j] = xfma(c_p, Ev[j][1], Ev[j][0]);
            else T[2][j] = xfnma(c_p, Ev[j][1], Ev[j][0]);
        }
    };
    auto hop = [&](auto k_c, float (&Vn)[3][6]) __attribute__((always_inline)) {         // k = 36 .. 71: local row (k - 36) / 12 -> Vn[row][0 .. 5]
        constexpr int h = decltype(k_c)::value - 36;
        xin_op<h % 12>(T[h / 12], Vn[h / 12], E);
    };

    // float offsets of the slab stage of chunk c / c + 1 / c + 2 and of the patch stage of chunk c + 1 / c + 2 / c + 3 of the chunk
    // STREAM (it runs on across tiles), rotated by compare-and-select (a modulo costs a dozen scalar instructions each)
    int uo0 = 0, uo1 = XUS, uo2 = 2 * XUS, po1 = XPSP, po2 = 2 * XPSP, po3 = 3 * XPSP;
    bool first = true;                                       // the workgroup's first tile: its operands do not come out of a previous tile's last chunk
    bool skipwait = false;                                   // a tile's first hand-over: nothing older than the previous tile's vmcnt(0) is needed,
                                                             // and the epilogue's stores must not be waited for
    // One chunk = 72 slots; EVERY chunk runs the same body (no joins of differently specialised copies: at every join the register
    // allocator moved accumulator tiles around).  PAR: parity of the chunk (which of V0 / V1 is current).
    typedef __attribute__((address_space(3))) f32x2 xlds_f32x2_t;
    auto ld2 = [](const float* p) __attribute__((always_inline)) {                       // the 8-byte group: two positions of one channel
        const f32x2 t2 = *(const volatile xlds_f32x2_t*)p;
        return (f32x4){t2[0], t2[1], 0.f, 0.f};
    };
    auto chunk = [&](auto par_c, auto first_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;       // the tile's first chunk: MFMAs with C = 0
        float(&Vc)[3][6] = PAR ? V1 : V0;
        float(&Vn)[3][6] = PAR ? V0 : V1;
        const float* ap_lo = ldsU + uo0 + a_lo;
        const float* ap_hi = ldsU + uo0 + a_hi;
        const float* ap2_lo = ldsU + uo0 + a2_lo;
        const float* ap2_hi = ldsU + uo0 + a2_hi;
        const float* apn = ldsU + uo1 + a_lo;                // (the next chunk's groups 0, 1: local tiles 0, 1)
        const float* pb = ldsP + po1 + b_off;
        const float* pe = pb + e_row * XPP;
        const int ud = uo2, pf = po2, pd = po3;
        {
            const int t = uo0; uo0 = uo1; uo1 = uo2; uo2 = t;
            const int q = po3 + XPSP == XNP * XPSP ? 0 : po3 + XPSP;
            po1 = po2; po2 = po3; po3 = q;
        }
        slab_wrap();                                         // this chunk's DMA: slab chunk ucur, patch chunk pcur -- of this tile or the next
        patch_wrap();
        {   // one address register per base (immediate offsets behind it)
            auto pin = [](const float*& p) __attribute__((always_inline)) {
                unsigned v = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
                asm volatile("" : "+v"(v));
                p = (const float*)(const __attribute__((address_space(3))) float*)(size_t)v;
            };
            pin(ap_lo); pin(ap_hi); pin(ap2_lo); pin(ap2_hi); pin(apn); pin(pb); pin(pe);
        }
        xfor(std::make_integer_sequence<int, 72>{}, [&](auto s_c) __attribute__((always_inline)) {
            constexpr int S = decltype(s_c)::value;
            constexpr int hg = y_hg(S), q = y_q(S), g = hg < 16 ? hg >> 2 : 4, ct = hg < 16 ? hg & 3 : hg - 16, p = 4 * g + q;
            const float av = A[hg & 3][q], bvv = Vc[p / 6][p % 6];
            if constexpr (p < 16) { if constexpr (FIRST) xmfma_a0(accA[4 * p + ct], av, bvv); else xmfma_a(accA[4 * p + ct], av, bvv); }   // [x4:mf]
            else { if constexpr (FIRST) xmfma_v0(accV[4 * (p - 16) + ct], av, bvv); else xmfma_v(accV[4 * (p - 16) + ct], av, bvv); }   // [x4:mf]
            if constexpr (q == 1) {                          // group hg + 2 (of this chunk, or 0 / 1 of the next one)
                constexpr int h2 = hg + 2, hh = h2 % 20;
                constexpr int g2 = hh < 16 ? hh >> 2 : 4, c2 = hh < 16 ? hh & 3 : hh - 16;
                if constexpr (h2 >= 20) A[h2 & 3] = *(const volatile xlds_f32x4_t*)(apn + c2 * 64);   // [x4:ar]
                else if constexpr (g2 < 4) A[h2 & 3] = *(const volatile xlds_f32x4_t*)((c2 < 2 ? ap_lo : ap_hi) + g2 * 256 + (c2 & 1) * 64);   // [x4:ar]
                else A[h2 & 3] = ld2((c2 < 2 ? ap2_lo : ap2_hi) + (c2 & 1) * 32);   // [x4:ar]
            }
            if constexpr (S == XHAND) {
                // everything but this chunk's DMA instructions so far (11 of its 12) has landed: slab c + 1, patch c + 2
                if (!skipwait) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(x_dma_before(XHAND)) : "memory");   // [x4:ho]
                skipwait = false;
                fixup(pf, fix_ho, edge_ho);   // [x4:ho]
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // [x4:ho]
            }
            constexpr int wr = x_wread_at(S);
            if constexpr (wr >= 0) wread(std::integral_constant<int, (wr >= 0 ? wr : 0)>{}, pb, pe);   // [x4:wr]
            constexpr int di = x_dma_at(S);
            if constexpr (di >= 0) {
                if constexpr (di < XUI) dma_u(di, ud);   // [x4:dma]
                else dma_patch(di - XUI, pd);   // [x4:dma]
                if constexpr (di == XUI - 1) advance_u();
                if constexpr (di == XDI - 1) advance_patch();
            }
            if constexpr (S == XFMA0) edge_ops();   // [x4:xf]
            constexpr int k0 = x_valu_before(S), k1 = x_valu_before(S + 1);
            if constexpr (S != XFMA0) {
                xfor(std::make_integer_sequence<int, k1 - k0>{}, [&](auto k_c) __attribute__((always_inline)) {
                    constexpr int k = k0 + decltype(k_c)::value;
                    if constexpr (k < 36) mid_op(std::integral_constant<int, (k < 36 ? k : 12)>{});   // [x4:xf]
                    else hop(std::integral_constant<int, (k >= 36 ? k : 36)>{}, Vn);   // [x4:xf]
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        fix_ho = fix; edge_ho = edge;                        // (of the patch this chunk issued: the next hand-over completes it)
    };

    auto epilogue = [&](int tvid) __attribute__((always_inline)) {
        int cot, pix;
        decode(tvid, cot, pix);
        const int u0 = pix * XTW;
        const int n = (u0 / period) / bands;
        const int co_w = cot * XBM + wp * 32;
        // ---- epilogue: Y = A^T M A per (channel, tile) in registers, then bias / ReLU / mask / pool and buffer stores.
        // Accumulator element r of tile (p, ct): channel co_w + 16 ct + 4 kq + r, tile ttx, position p.
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last MFMAs' results (inline asm: the compiler pads nothing)
        int tsn, py, px;
        const bool tile_ok = tile_geometry(u0, tsn, py, px);
        const int cmax = Cout - co_w - 4 * kq;                   // channel 16 ct + r of this lane exists iff 16 ct + r < cmax
        auto rd = [](float a) __attribute__((always_inline)) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; };
        // the wave's PARTIAL 4x4 outputs of local channel (ct, r) from its three rows of M: o[k][l], k = output row, l = output column.
        // Local rows (edge, +p, -p): z_i = M[i, :] A (horizontal), then column-wise with s = z_1 + z_2, d = z_1 - z_2:
        // o_0 = s + e0 z_e, o_1 = p d, o_2 = p^2 s, o_3 = p^3 d + e3 z_e  (half 0: edge = point 0 -> row 0; half 1: edge = inf -> row 3)
        auto half_inverse = [&](auto ct_c, auto r_c, float (&o)[4][4]) __attribute__((always_inline)) {
            constexpr int ct = decltype(ct_c)::value, r = decltype(r_c)::value;
            float z[3][4];
            xfor(std::make_integer_sequence<int, 3>{}, [&](auto i_c) __attribute__((always_inline)) {
                constexpr int i = decltype(i_c)::value;
                float m[6];
                xfor(std::make_integer_sequence<int, 6>{}, [&](auto j_c) __attribute__((always_inline)) {
                    constexpr int p = 6 * i + decltype(j_c)::value;
                    if constexpr (p < 16) m[decltype(j_c)::value] = rd(accA[4 * p + ct][r]);
                    else m[decltype(j_c)::value] = accV[4 * (p - 16) + ct][r];
                });
                xout(m, z[i]);
            });
    #pragma unroll
            for (int l = 0; l < 4; ++l) {
                const float sm = xadd(z[1][l], z[2][l]), df = xsub(z[1][l], z[2][l]);
                o[0][l] = xfma(o_e0, z[0][l], sm);
                o[1][l] = xmul(o_p1, df);
                o[2][l] = xmul(o_p2, sm);
                o[3][l] = xfma(o_e3, z[0][l], xmul(o_p3, df));
            }
        };
        // the exchange area: the slab stage the tile's last chunk has left free (uo2: the DMA target of the NEXT chunk, which no wave
        // issues before the barrier that ends this epilogue) -- [buffer 2][wave 4][output row 4][lane 64] x 16 bytes = 32 KB
        float* const xbuf = ldsU + uo2;
        const int xw = wave * 1024 + lane * 4, xr = (wave ^ 2) * 1024 + lane * 4;      // this wave's / its partner's (same tile row, other half) slots
        // the full 4x4 outputs of the lane's j-th finalised channel (local tile c = j >> 2, element r = j & 3: real channel
        // co_w + 16 c + 4 kq + r): the wave's own partial + the partner's, which arrives through LDS while this wave sends the partial of
        // local channel (2 + c, r) the other way.  One workgroup barrier per channel, double-buffered slots, software-pipelined: the
        // partial for channel j + 1 is computed and written while the partner's partial for channel j is on its way back from LDS
        // (the first version -- write, barrier, read, add, per channel -- cost 4.8 k cycles per tile more than the unsplit epilogue)
        auto send = [&](auto j_c) __attribute__((always_inline)) {
            constexpr int j = decltype(j_c)::value, ct = 2 + (j >> 2), r = j & 3;
            // (as half_inverse, emitted row by row: twelve live values instead of twenty-eight)
            float z[3][4];
            xfor(std::make_integer_sequence<int, 3>{}, [&](auto i_c) __attribute__((always_inline)) {
                constexpr int i = decltype(i_c)::value;
                float m[6];
                xfor(std::make_integer_sequence<int, 6>{}, [&](auto jj_c) __attribute__((always_inline)) {
                    constexpr int p = 6 * i + decltype(jj_c)::value;
                    if constexpr (p < 16) m[decltype(jj_c)::value] = rd(accA[4 * p + ct][r]);
                    else m[decltype(jj_c)::value] = accV[4 * (p - 16) + ct][r];
                });
                xout(m, z[i]);
            });
            float* const wb = xbuf + (j & 1) * 4096 + xw;
            f32x4 sm, df;
    #pragma unroll
            for (int l = 0; l < 4; ++l) { sm[l] = xadd(z[1][l], z[2][l]); df[l] = xsub(z[1][l], z[2][l]); }
            *(volatile xlds_f32x4_t*)(wb) = (f32x4){xfma(o_e0, z[0][0], sm[0]), xfma(o_e0, z[0][1], sm[1]), xfma(o_e0, z[0][2], sm[2]), xfma(o_e0, z[0][3], sm[3])};
            *(volatile xlds_f32x4_t*)(wb + 256) = (f32x4){xmul(o_p1, df[0]), xmul(o_p1, df[1]), xmul(o_p1, df[2]), xmul(o_p1, df[3])};
            *(volatile xlds_f32x4_t*)(wb + 512) = (f32x4){xmul(o_p2, sm[0]), xmul(o_p2, sm[1]), xmul(o_p2, sm[2]), xmul(o_p2, sm[3])};
            *(volatile xlds_f32x4_t*)(wb + 768) = (f32x4){xfma(o_e3, z[0][0], xmul(o_p3, df[0])), xfma(o_e3, z[0][1], xmul(o_p3, df[1])),
                                                           xfma(o_e3, z[0][2], xmul(o_p3, df[2])), xfma(o_e3, z[0][3], xmul(o_p3, df[3]))};
        };
        auto inverse = [&](auto ct_c, auto r_c, float (&o)[4][4], auto pipe_c) __attribute__((always_inline)) {
            constexpr int ct = decltype(ct_c)::value, r = decltype(r_c)::value, j = 4 * ct + r;
            constexpr bool PIPE = decltype(pipe_c)::value;           // (false: the mask epilogue, whose producer activations are in flight too --
                                                                      // the pipelined order spills there; its next partial goes out after the adds)
            half_inverse(ct_c, r_c, o);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // (the partner's partial j is in its slot)
            f32x4 pv[4];
            const float* const rb = xbuf + (j & 1) * 4096 + xr;
    #pragma unroll
            for (int a = 0; a < 4; ++a) pv[a] = *(const volatile xlds_f32x4_t*)(rb + a * 256);
            // slot (j + 1) & 1 was last read for channel j - 1: consumed by every wave before it reached the barrier above
            if constexpr (PIPE && j < 7) send(std::integral_constant<int, (j < 7 ? j + 1 : 0)>{});
    #pragma unroll
            for (int a = 0; a < 4; ++a) {
                o[a][0] = xadd(o[a][0], pv[a][0]); o[a][1] = xadd(o[a][1], pv[a][1]); o[a][2] = xadd(o[a][2], pv[a][2]); o[a][3] = xadd(o[a][3], pv[a][3]);
            }
            if constexpr (!PIPE && j < 7) send(std::integral_constant<int, (j < 7 ? j + 1 : 0)>{});
        };
        send(std::integral_constant<int, 0>{});
        auto for_channels = [&](auto&& f) __attribute__((always_inline)) {       // the eight channels the wave finalises (local tiles 0, 1)
            xfor(std::make_integer_sequence<int, 8>{}, [&](auto c_c) __attribute__((always_inline)) {
                f(std::integral_constant<int, (decltype(c_c)::value >> 2)>{}, std::integral_constant<int, (decltype(c_c)::value & 3)>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if (epi == 4) {
            // bias + ReLU + 2x2/2 max pool (floor mode): the tile is four pool windows
            const int OH = H >> 1, OW = W >> 1, OHW = OH * OW;
            const long long ytail = (long long)(N - n) * Cout * OHW * 4;
            const __amdgpu_buffer_rsrc_t ry = ptmi_rsrc(y + (size_t)n * Cout * OHW, (unsigned)(ytail > 0xFFFFFFFEll ? 0xFFFFFFFEll : ytail));
            const int oy = py >> 1, ox = px >> 1;
            unsigned pv2[2], pv1[2];
    #pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bool rok = tile_ok && oy + a < OH;
                const unsigned o = (unsigned)(((tsn - n) * Cout + 4 * kq) * OHW + (oy + a) * OW + ox) * 4u;
                pv2[a] = (rok && ox + 1 < OW) ? o : 0xFFFFFFFFu;
                pv1[a] = (rok && ox + 1 == OW) ? o : 0xFFFFFFFFu;
            }
            for_channels([&](auto ct_c, auto r_c) __attribute__((always_inline)) {
                constexpr int ct = decltype(ct_c)::value, r = decltype(r_c)::value;
                float o[4][4];
                inverse(ct_c, r_c, o, std::true_type{});
                const float b = bv[ct][r];
                const bool cok = 16 * ct + r < cmax;
                const int soff = (co_w + 16 * ct + r) * OHW * 4;
    #pragma unroll
                for (int a = 0; a < 2; ++a) {
                    f32x2 m;
    #pragma unroll
                    for (int l = 0; l < 2; ++l)
                        m[l] = fmaxf(fmaxf(fmaxf(o[2 * a][2 * l] + b, o[2 * a][2 * l + 1] + b),
                                           fmaxf(o[2 * a + 1][2 * l] + b, o[2 * a + 1][2 * l + 1] + b)), 0.f);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, m), ry, cok ? (int)pv2[a] : -1, soff, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m[0]), ry, cok ? (int)pv1[a] : -1, soff, 0);
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (the exchange area is a DMA target again from here on)
            return;
        }
        const long long ytail = (long long)(N - n) * Cout * HW * 4;
        const unsigned img_bytes = (unsigned)(ytail > 0xFFFFFFFEll ? 0xFFFFFFFEll : ytail);     // to the end of the tensor, clamped
        const __amdgpu_buffer_rsrc_t ry = ptmi_rsrc(y + (size_t)n * Cout * HW, img_bytes);
        const __amdgpu_buffer_rsrc_t rm = ptmi_rsrc(epi == 3 ? mref + (size_t)n * Cout * HW : y, epi == 3 ? img_bytes : 0u);
        // per-lane byte offsets of the tile's four rows: pv4 = all four columns inside the image (16-byte access), pve[e] = column
        // e alone (tiles cut by the right edge)
        unsigned pv4[4];
        unsigned pve[4][3];
        bool partial = false;
    #pragma unroll
        for (int a = 0; a < 4; ++a) {
            const bool rok = tile_ok && py + a < H;
            const unsigned o = (unsigned)(((tsn - n) * Cout + 4 * kq) * HW + (py + a) * W + px) * 4u;
            pv4[a] = (rok && px + 3 < W) ? o : 0xFFFFFFFFu;
    #pragma unroll
            for (int e = 0; e < 3; ++e) {
                pve[a][e] = (rok && px + 3 >= W && px + e < W) ? o + 4u * e : 0xFFFFFFFFu;
                partial |= pve[a][e] != 0xFFFFFFFFu;
            }
        }
        const bool cut = __any(partial);                         // (wave-uniform) some lane's tile is cut by the right edge
        auto store_rows = [&](auto epi_c) __attribute__((always_inline)) {
            constexpr int EPI = decltype(epi_c)::value;
            for_channels([&](auto ct_c, auto r_c) __attribute__((always_inline)) {
                constexpr int ct = decltype(ct_c)::value, r = decltype(r_c)::value;
                const bool cok = 16 * ct + r < cmax;
                const int soff = (co_w + 16 * ct + r) * HW * 4;
                f32x4 mk[4];
                float ms[4][3];
                if constexpr (EPI == 3) {                                  // the producer's activations first: their latency hides
    #pragma unroll                                                         // behind the inverse transform
                    for (int a = 0; a < 4; ++a) {
                        mk[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, cok ? (int)pv4[a] : -1, soff, 0));
                        if (cut) {
    #pragma unroll
                            for (int e = 0; e < 3; ++e)
                                ms[a][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, cok ? (int)pve[a][e] : -1, soff, 0));
                        }
                    }
                }
                float o[4][4];
                inverse(ct_c, r_c, o, std::integral_constant<bool, (EPI != 3)>{});
                const float b = bv[ct][r];
    #pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f32x4 st;
    #pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        float v = o[a][l];
                        if constexpr (EPI <= 1) v += b;
                        if constexpr (EPI == 1) v = fmaxf(v, 0.f);
                        if constexpr (EPI == 3) {
                            float m = mk[a][l];
                            if (cut && l < 3) m = (pve[a][l] != 0xFFFFFFFFu) ? ms[a][l] : m;
                            v = (m > 0.f) ? v : 0.f;
                        }
                        st[l] = v;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, st), ry, cok ? (int)pv4[a] : -1, soff, 0);
                    // a 16-byte store reads its data registers for two more states: hipcc pads that against ITS next VALU, not against
                    // an inline-asm one (found the hard way: output (1, 1) of tile columns 12 .. 15 wrong, only with a bias)
                    asm volatile("s_nop 1" ::: "memory");
                    if (cut) {
    #pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            const float sv = e == 0 ? st[0] : e == 1 ? st[1] : st[2];   // (bit_cast of a vector ELEMENT lvalue reads element 0)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sv), ry, cok ? (int)pve[a][e] : -1, soff, 0);
                        }
                    }
                }
            });
        };
        if (epi == 0) store_rows(std::integral_constant<int, 0>{});
        else if (epi == 1) store_rows(std::integral_constant<int, 1>{});
        else if (epi == 2) store_rows(std::integral_constant<int, 2>{});
        else store_rows(std::integral_constant<int, 3>{});
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           // (the exchange area is a DMA target again from here on)
    };

    for (;;) {
        // [x4@t0]
        // (no zeroing pass: the tile's first chunk issues its MFMAs with C = 0)
        if (first) {
            first = false;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XDI) : "memory");
            fixup(0, fix_p0, edge_p0);
            fixup(XPSP, fix_p1, edge_p1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // B operands of chunk 0: window from patch stage 0; A groups 0, 1 (position group 0, local tiles 0, 1) from slab stage 0
            const float* pb = ldsP + b_off;
            xfor(std::make_integer_sequence<int, 15>{}, [&](auto r_c) __attribute__((always_inline)) { wread(r_c, pb, pb + e_row * XPP); });
            edge_ops();
            xfor(std::make_integer_sequence<int, 24>{}, [&](auto k_c) __attribute__((always_inline)) { mid_op(std::integral_constant<int, 12 + decltype(k_c)::value>{}); });
            xfor(std::make_integer_sequence<int, 36>{}, [&](auto k_c) __attribute__((always_inline)) { hop(std::integral_constant<int, 36 + decltype(k_c)::value>{}, V0); });
            A[0] = *(const volatile xlds_f32x4_t*)(ldsU + a_lo);
            A[1] = *(const volatile xlds_f32x4_t*)(ldsU + a_lo + 64);
        }
        // [x4@t1]
        chunk(std::integral_constant<int, 0>{}, std::true_type{});         // (writes every accumulator tile: C = 0)
        chunk(std::integral_constant<int, 1>{}, std::false_type{});
        for (int c = 2; c < nChunks; c += 2) {              // nChunks is even and >= 2 (launcher: Cin % 8 == 0, Cin >= 8)
            chunk(std::integral_constant<int, 0>{}, std::false_type{});
            chunk(std::integral_constant<int, 1>{}, std::false_type{});
        }
        // [x4@t2]
        // the last chunk's own DMA instructions (the next tile's chunk 1 slab / chunk 2 patch): with them done, nothing the next
        // tile's counted waits rely on is older than the epilogue's stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (DYN) {
            if (wave == 0) {                                 // this tile's draw: valid, or the next queue's, or "none left"
                int sq = __builtin_amdgcn_readfirstlane(sched_ids[4]), sleft = __builtin_amdgcn_readfirstlane(sched_ids[5]);
                const int v = resolve(__builtin_amdgcn_readfirstlane(pend_k), sq, sleft);
                if (lane == 0) { sched_ids[2] = v; sched_ids[4] = sq; sched_ids[5] = sleft; }
            }
        }
        // [x4@t3]
        epilogue(vid);
        // [x4@t4]
        if constexpr (DYN) vid = __builtin_amdgcn_readfirstlane(sched_ids[3]);   // (written where this tile's slab cursor wrapped: barriers ago)
        else vid += grid;
        {
            int c0, p0;
            if (!decode(vid, c0, p0)) break;
        }
        load_bias(vid);
        skipwait = true;
    }
    sched_exit();
}

// U = G g G^T (6x6 per filter) laid out as the kernel's LDS image: [channel tile (64)][chunk (4 ci)][ci][half 2][group][co 64][position
// in group]; a half's local position p = 6 i' + j stands for transform row i = (0, 1, 2)[i'] (half 0) / (5, 3, 4)[i'] (half 1) and
// column j; groups 0 .. 3 hold four positions each (16 bytes per channel), group 4 the last two (8 bytes).
// mode as ptmi_conv3x3_pack_weights (1: dgrad -- transposed channels, flipped taps).
__global__ void wino4p_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int wCout, int wCin, int mode,
                                          int coTiles, int nChunks)
{
    const int64_t total = (int64_t)coTiles * nChunks * XKC * XBM;
    const int convCout = mode ? wCin : wCout, convCin = mode ? wCout : wCin;
    const double G[6][3] = {{64.0 / 81.0, 0.0, 0.0},
                            {-128.0 / 243.0, -32.0 / 81.0, -8.0 / 27.0},
                            {-128.0 / 243.0, 32.0 / 81.0, -8.0 / 27.0},
                            {32.0 / 243.0, 16.0 / 81.0, 8.0 / 27.0},
                            {32.0 / 243.0, -16.0 / 81.0, 8.0 / 27.0},
                            {0.0, 0.0, 1.0}};
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = idx;
        const int col = t % XBM; t /= XBM;
        const int cil = t % XKC; t /= XKC;
        const int chunk = t % nChunks;
        const int cot = t / nChunks;
        const int co = cot * XBM + col, ci = chunk * XKC + cil;
        double g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float v = 0.f;
                if (co < convCout && ci < convCin)
                    v = mode == 0 ? w[((size_t)co * wCin + ci) * 9 + ky * 3 + kx]
                                  : w[((size_t)ci * wCin + co) * 9 + (2 - ky) * 3 + (2 - kx)];
                g[ky][kx] = (double)v;
            }
        }
        double rr[6][3];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) rr[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
        }
        float* dst = wp + ((size_t)(cot * nChunks + chunk) * XKC + cil) * XUC;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int half = i >= 3, il = half ? (i == 5 ? 0 : i - 2) : i;      // rows (0, 1, 2) -> local (0, 1, 2); rows (5, 3, 4) -> local (0, 1, 2)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int p = 6 * il + j;
                const float u = (float)(rr[i][0] * G[j][0] + rr[i][1] * G[j][1] + rr[i][2] * G[j][2]);
                if (p < 16) dst[half * XUH + (p >> 2) * (XBM * 4) + col * 4 + (p & 3)] = u;
                else dst[half * XUH + 1024 + col * 2 + (p - 16)] = u;
            }
        }
    }
}

}  // namespace

extern "C" {

int64_t ptmi_conv3x3_wino4p_packed_floats(int cin, int cout)
{
    return (int64_t)cdiv(cout, XBM) * cdiv(cin, XKC) * XUS;
}

int ptmi_conv3x3_wino4p_pack_weights(const float* w, float* wp, int w_cout, int w_cin, int mode, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(w && wp && w_cout > 0 && w_cin > 0, "conv3x3_wino4p_pack_weights: bad args");
    const int convCout = mode ? w_cin : w_cout, convCin = mode ? w_cout : w_cin;
    const int coTiles = cdiv(convCout, XBM), nChunks = cdiv(convCin, XKC);
    const int64_t total = (int64_t)coTiles * nChunks * XKC * XBM;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(wino4p_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, wp, w_cout, w_cin, mode,
                       coTiles, nChunks);
    PTMI_LAUNCH_CHECK("conv3x3_wino4p_pack_weights");
    return 0;
}

int ptmi_conv3x3_wino4p_fwd_fits(int cin, int cout, int h, int w)
{
    if (cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (cin & 7)) return 0;       // chunks of 4 channels, walked in pairs
    // a workgroup's 64 flat columns may reach into the strips of later images: per-lane offsets are relative to the first one
    const int64_t img_span = XTW / ((w + 4) & ~3) + 2;
    return (img_span * cin + XKC) * h * w * 4 < (1ll << 32) && (img_span * cout + XBM) * h * w * 4 < (1ll << 32);
}

// CUs of the current device: a hardware constant, looked up once per device id (35 launches per step); a benign race at worst
// writes the same value twice
static int wino4p_device_cus()
{
    static int cus_by_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64 && cus_by_dev[dev] > 0) return cus_by_dev[dev];
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    if (dev >= 0 && dev < 64) cus_by_dev[dev] = cus;
    return cus;
}

int ptmi_conv3x3_wino4p_fwd_sched(const float* x, const float* wp, const float* bias, const float* mask_ref, float* y, int n,
                                 int cin, int cout, int h, int w, int epilogue, int32_t* sched, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(x && wp && y && n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "conv3x3_wino4p_fwd: bad args");
    PTMI_CHECK_ARG(epilogue >= 0 && epilogue <= 4, "conv3x3_wino4p_fwd: bad epilogue %d", epilogue);
    PTMI_CHECK_ARG(!(cin & 7), "conv3x3_wino4p_fwd: cin %d is not a multiple of 8 (use ptmi_conv3x3_wino_fwd)", cin);
    PTMI_CHECK_ARG(ptmi_conv3x3_wino4p_fwd_fits(cin, cout, h, w),
                   "conv3x3_wino4p_fwd: image too large for 32-bit buffer offsets (n=%d cin=%d cout=%d h=%d w=%d)", n, cin,
                   cout, h, w);
    PTMI_CHECK_ARG(epilogue > 1 || bias, "conv3x3_wino4p_fwd: bias required for epilogue %d", epilogue);
    PTMI_CHECK_ARG(epilogue != 4 || bias, "conv3x3_wino4p_fwd: bias required for epilogue 4");
    PTMI_CHECK_ARG(epilogue != 3 || mask_ref, "conv3x3_wino4p_fwd: mask_ref required for epilogue 3");
    const int bands = cdiv(h, XTH), coTiles = cdiv(cout, XBM), nChunks = cin / XKC;
    const int period = (w + 1 + 3) & ~3;                     // strip length: W + at least one zero column, a multiple of 4
    const int64_t nPix = cdiv64((int64_t)n * bands * period, XTW);
    PTMI_CHECK_ARG(nPix * XTW < (1ll << 31), "conv3x3_wino4p_fwd: too many tiles");
    const int colocate = coTiles <= 4;
    const int64_t nWg = colocate ? cdiv64(nPix, 8) * 8 * coTiles : nPix * coTiles;      // tile ids (colocate: some beyond nPix -- the end)
    PTMI_CHECK_ARG(nWg < (1ll << 31) - 4096, "conv3x3_wino4p_fwd: too many tiles");
    // persistent workgroups: one per CU (a multiple of 8: a tile stays on the XCD of its id mod 8)
    const int cus = wino4p_device_cus();
    const int64_t grid = nWg < (cus / 8) * 8 ? nWg : (cus / 8) * 8;
    if (sched && nChunks >= 4)       // (fewer chunks per tile than the schedule's LDS hand-offs assume: the static walk)
        hipLaunchKernelGGL(conv3x3_wino4p_kernel<true>, dim3((unsigned)grid), dim3(XNT), 0, (hipStream_t)s, x, wp, bias, mask_ref, y, n,
                           cin, cout, h, w, nChunks, epilogue, coTiles, bands, period, (int)nPix, colocate, (int)nWg, (int*)sched);
    else
        hipLaunchKernelGGL(conv3x3_wino4p_kernel<false>, dim3((unsigned)grid), dim3(XNT), 0, (hipStream_t)s, x, wp, bias, mask_ref, y, n,
                           cin, cout, h, w, nChunks, epilogue, coTiles, bands, period, (int)nPix, colocate, (int)nWg, (int*)nullptr);
    PTMI_LAUNCH_CHECK("conv3x3_wino4p_fwd");
    return 0;
}

int ptmi_conv3x3_wino4p_fwd(const float* x, const float* wp, const float* bias, const float* mask_ref, float* y, int n,
                           int cin, int cout, int h, int w, int epilogue, ptmi_stream_t s)
{
    return ptmi_conv3x3_wino4p_fwd_sched(x, wp, bias, mask_ref, y, n, cin, cout, h, w, epilogue, nullptr, s);
}

}  // extern "C"
