// gemm.hip -- fp32 GEMM on v_mfma_f32_32x32x2_f32 for gfx950, all four transpose combinations,
// bounds-checked, batched, with bias / ReLU / accumulate epilogues.
//
// Replaces the cuBLAS Linear kernels behind D2's FastRCNNConvFCHead + the box predictor
// (pt/modeling/roi_heads/roi_heads.py:127-128, pt/modeling/roi_heads/fast_rcnn.py:164) and the
// 1x1 convolutions of StandardRPNHead (pt/modeling/proposal_generator/rpn.py:96), fwd + bwd.
//
// 128x128 workgroup tile, 4 wave64s in 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles, BK = 32; operand tiles go
// global -> LDS with `buffer_load_dwordx4 ... lds` (see gemm_buf_kernel below).
// Accumulation order is fixed inside a tile chain => bitwise reproducible run to run.
#include "common.h"

namespace {

constexpr int BMN = 128;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// bf16-input / fp32-accumulate variant (SOLVER.AMP.ENABLED, BASELINE configs[4]): the operands stay fp32 in HBM and
// LDS; a lane rounds its eight k values to bf16 (v_cvt_pk_bf16_f32, round-to-nearest-even) on the way from LDS to the
// MFMA: v_mfma_f32_32x32x16_bf16, lane half h element j <-> the same k for A and B.
__device__ __forceinline__ f32x16 mfma16bf(ptmi_bf16x8 a, ptmi_bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------ buffer-DMA pipeline
// Default kernel.  Same 128x128x32 tiling and wave layout, but the operand tiles go global -> LDS directly with
// `buffer_load_dwordx4 ... lds` (no VGPR staging, no ds_write), double-buffered with one barrier per K step:
//   * a k-fastest operand (row-major A, or B stored (N,K)) lands as T[mn][36]: 8 data pieces + 1 pad piece per row;
//     the MFMA k index is permuted (lane half h, step j <-> k = 16h + j) so that a lane's 16 k values are
//     contiguous: four ds_read_b128 per 32-row sub-tile, conflict-free at pitch 36 (= 4 * odd);
//   * an mn-fastest operand lands as T[k][128] and is read with ds_read_b32 (32 consecutive mn per lane group);
//   * rows beyond M / N and K-tail rows / pieces carry offset 0xFFFFFFFF (zero-filled by the range check); a 16-B
//     piece that straddles K (K % 4 != 0) is loaded whole and its tail words are zeroed in LDS by the loading lane.
// The k permutation is the same for both operands, so the sum is unchanged up to fp32 ordering, and the order is
// fixed => bitwise reproducible run to run.
// BKT = K extent of a stage: 32 (two workgroups per CU, 64 MFMAs per wave per barrier) or 16 (three per CU, 32).
template <int BKT>
struct TileCfg {
    static constexpr int HK = BKT / 2;                    // k values per lane half
    static constexpr int KF_PITCH = BKT + 4;              // 36 / 20 floats: odd multiple of 16 B
    static constexpr int KF_PR = BKT / 4 + 1;             // 16-B slots per row (data pieces + 1 pad)
    static constexpr int KF_SLOTS = 128 * KF_PR;          // 1152 / 640
    static constexpr int KF_NI = (KF_SLOTS + 255) / 256;  // DMA instructions per lane: 5 / 3 (the last one: waves 0-1)
    static constexpr int MF_NI = BKT / 8;                 // 4 / 2
    static constexpr int OP_FLOATS = KF_SLOTS * 4 + 256;  // per operand per stage (+ DMA overhang)
};

template <bool KF, int BKT>
struct OpTile {
    using Cfg = TileCfg<BKT>;
    unsigned voff[KF ? Cfg::KF_NI : 1];   // loop-invariant byte offsets of this lane's pieces (0xFFFFFFFF = never loaded)
    unsigned qpack;                 // KF: first k of piece i in bits 6i..6i+5;  !KF: k row of piece 0 (rows 8i + that)
    const char* base;               // this tile at the current K step (wave-uniform)
    long long left;                 // bytes from base to the end of the operand
    int step_bytes, row8_bytes;

    __device__ __forceinline__ void init(const float* p, int ld, int mn0, int MN, int K, int tid, int64_t total_floats)
    {
        if constexpr (KF) {
            qpack = 0;
#pragma unroll
            for (int i = 0; i < Cfg::KF_NI; ++i) {
                const int sl = tid + i * 256;
                const int row = sl / Cfg::KF_PR, q = sl - row * Cfg::KF_PR;
                voff[i] = (sl < Cfg::KF_SLOTS && q < Cfg::KF_PR - 1 && mn0 + row < MN) ? (unsigned)(row * ld + 4 * q) * 4u
                                                                                       : 0xFFFFFFFFu;
                qpack |= (unsigned)(4 * q) << (6 * i);
            }
            base = (const char*)(p + (size_t)mn0 * ld);
            left = (total_floats - (int64_t)mn0 * ld) * 4;
            step_bytes = BKT * 4;
            row8_bytes = 0;
        } else {
            const int kr = tid >> 5, q = tid & 31;
            voff[0] = (mn0 + 4 * q < MN) ? (unsigned)(kr * ld + 4 * q) * 4u : 0xFFFFFFFFu;
            qpack = kr;
            base = (const char*)(p + mn0);
            left = (total_floats - mn0) * 4;
            step_bytes = BKT * ld * 4;
            row8_bytes = 8 * ld * 4;
        }
    }
    // krem = K - k0 (>= BKT except in the tail step)
    __device__ __forceinline__ void issue(float* dst_wave, int wave, int krem)
    {
        const __amdgpu_buffer_rsrc_t r = ptmi_rsrc(base, (unsigned)(left > 0xFFFFFFFEll ? 0xFFFFFFFEll : (left < 0 ? 0 : left)));
        // the K-tail masks belong to the LAST step only: as per-lane selects on every piece of every step (what the compiler makes of
        // `if (krem < BKT) v = ...`) they were three VALU instructions per DMA instruction -- and VALU time adds to fp32 MFMA time
        if (krem >= BKT) {                                   // (wave-uniform)
            if constexpr (KF) {
#pragma unroll
                for (int i = 0; i < Cfg::KF_NI; ++i)
                    if (i < Cfg::KF_NI - 1 || wave < 2) ptmi_bdma16(r, voff[i], 0, dst_wave + i * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < Cfg::MF_NI; ++i) ptmi_bdma16(r, voff[0], i * row8_bytes, dst_wave + i * 1024);
            }
        } else {
            asm volatile("" ::: "memory");                   // (keeps the two forms apart: merged, they are the selects again)
            if constexpr (KF) {
#pragma unroll
                for (int i = 0; i < Cfg::KF_NI; ++i) {
                    if (i < Cfg::KF_NI - 1 || wave < 2) {
                        const unsigned v = ((int)((qpack >> (6 * i)) & 63) < krem) ? voff[i] : 0xFFFFFFFFu;
                        ptmi_bdma16(r, v, 0, dst_wave + i * 1024);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < Cfg::MF_NI; ++i) {
                    const unsigned v = ((int)qpack + 8 * i < krem) ? voff[0] : 0xFFFFFFFFu;
                    ptmi_bdma16(r, v, i * row8_bytes, dst_wave + i * 1024);
                }
            }
        }
        base += step_bytes;
        left -= step_bytes;
    }
    // zero the words of own pieces that lie beyond K (KF only, tail step with K % 4 != 0)
    __device__ __forceinline__ void fixup(float* dst_lane, int wave, int krem)
    {
        if constexpr (KF) {
#pragma unroll
            for (int i = 0; i < Cfg::KF_NI; ++i) {
                const int k4 = (qpack >> (6 * i)) & 63;
                if ((i < Cfg::KF_NI - 1 || wave < 2) && voff[i] != 0xFFFFFFFFu && k4 < krem && k4 + 4 > krem) {
#pragma unroll
                    for (int e = 1; e < 4; ++e)
                        if (k4 + e >= krem) dst_lane[i * 1024 + e] = 0.f;
                }
            }
        }
    }
    // fragment for the 32-row sub-tile starting at row r0: f[j] = T(row r0 + (lane & 31), k = HK * (lane >> 5) + j)
    __device__ __forceinline__ static void frag(const float* T, int r0, int lane, float (&f)[Cfg::HK])
    {
        if constexpr (KF) {
            const float* p = T + (r0 + (lane & 31)) * Cfg::KF_PITCH + (lane >> 5) * Cfg::HK;
#pragma unroll
            for (int t = 0; t < Cfg::HK / 4; ++t) {
                const f32x4 v = *(const ptmi_lds_f32x4_t*)(p + 4 * t);
                f[4 * t] = v[0]; f[4 * t + 1] = v[1]; f[4 * t + 2] = v[2]; f[4 * t + 3] = v[3];
            }
        } else {
            const float* p = T + (lane >> 5) * Cfg::HK * 128 + r0 + (lane & 31);
#pragma unroll
            for (int j = 0; j < Cfg::HK; ++j) f[j] = p[j * 128];
        }
    }
};

template <bool AK, bool BKF, int BKT, bool BF>
__global__ __launch_bounds__(256, (BKT == 16 ? 3 : 2)) void gemm_buf_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
    const float* __restrict__ bias, int M, int N, int K, int lda, int ldb, int ldc, int bias_mode, int relu,
    int accumulate, int64_t sa, int64_t sb, int64_t sc, int tilesM, int tilesN, int kChunk, float* __restrict__ partial)
{
    using Cfg = TileCfg<BKT>;
    constexpr int OPF = Cfg::OP_FLOATS, HK = Cfg::HK;
    __shared__ __attribute__((aligned(16))) float lds[4 * OPF];      // [buf][A | B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order.  Workgroup ids go round-robin over the 8 XCDs (id % 8), each with an L2 of its own, so with
    // plain row-major numbering the tilesN workgroups that share a 128-row A panel sit on 8 different L2s and the panel is
    // fetched from HBM / the Infinity Cache 8 times.  Here XCD x walks the contiguous range [x * per, (x + 1) * per) of the
    // row-major tile sequence: consecutive workgroups of an XCD share their A panel, and at any moment the 8 XCDs sit at
    // the same position of their ranges -- when per is a multiple of tilesN (fc1 forward) or tilesM == 8 (fc1 dW) that is
    // the same B panel, which then comes out of HBM once.  A/B on one box: fc1 dW +5 % (R = 16 384: 6.79 -> 6.46 ms),
    // forward and dX unchanged.
    const int per = (tilesM * tilesN + 7) >> 3;
    const int lin = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (lin >= tilesM * tilesN) return;                      // (grid = 8 * per >= tiles)
    const int bm = lin / tilesN, bn = lin - bm * tilesN;
    const int m0 = bm * BMN, n0 = bn * BMN;
    const int b = blockIdx.y;
    A += (size_t)b * sa;
    B += (size_t)b * sb;
    C += (size_t)b * sc;
    const int wm = wave >> 1, wn = wave & 1;
    // split-K (blockIdx.z = slice, kChunk = its K extent, a multiple of the stage depth): the slice's partial product goes,
    // without bias / ReLU / accumulate, to partial[z][M][N]; gemm_splitk_reduce_kernel sums the slices in a fixed order
    if (partial) {
        const int kbeg = blockIdx.z * kChunk;
        A += AK ? (size_t)kbeg : (size_t)kbeg * lda;
        B += BKF ? (size_t)kbeg : (size_t)kbeg * ldb;
        K = (K - kbeg) < kChunk ? (K - kbeg) : kChunk;
        C = partial + (size_t)blockIdx.z * M * N;
        ldc = N;
        bias_mode = relu = accumulate = 0;
    }

    OpTile<AK, BKT> ta;
    OpTile<BKF, BKT> tb;
    ta.init(A, lda, m0, M, K, tid, AK ? ((int64_t)(M - 1) * lda + K) : ((int64_t)(K - 1) * lda + M));
    tb.init(B, ldb, n0, N, K, tid, BKF ? ((int64_t)(N - 1) * ldb + K) : ((int64_t)(K - 1) * ldb + N));

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int nSteps = (K + BKT - 1) / BKT;
    ta.issue(lds + wave * 256, wave, K);
    tb.issue(lds + OPF + wave * 256, wave, K);
    for (int st = 0; st < nSteps; ++st) {
        const int buf = st & 1;
        float* As = lds + buf * 2 * OPF;
        float* Bs = As + OPF;
        const int krem = K - st * BKT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (krem < BKT && (krem & 3)) {
            ta.fixup(As + tid * 4, wave, krem);
            tb.fixup(Bs + tid * 4, wave, krem);
        }
        __syncthreads();
        if (st + 1 < nSteps) {
            float* An = lds + (buf ^ 1) * 2 * OPF + wave * 256;
            ta.issue(An, wave, krem - BKT);
            tb.issue(An + OPF, wave, krem - BKT);
        }
        float a0[HK], a1[HK], b0[HK], b1[HK];
        OpTile<AK, BKT>::frag(As, wm * 64, lane, a0);
        OpTile<AK, BKT>::frag(As, wm * 64 + 32, lane, a1);
        OpTile<BKF, BKT>::frag(Bs, wn * 64, lane, b0);
        OpTile<BKF, BKT>::frag(Bs, wn * 64 + 32, lane, b1);
        if constexpr (BF) {
#pragma unroll
            for (int t = 0; t < HK / 8; ++t) {
                const ptmi_bf16x8 A0 = ptmi_pack_bf16x8(a0 + 8 * t), A1 = ptmi_pack_bf16x8(a1 + 8 * t);
                const ptmi_bf16x8 B0 = ptmi_pack_bf16x8(b0 + 8 * t), B1 = ptmi_pack_bf16x8(b1 + 8 * t);
                acc00 = mfma16bf(A0, B0, acc00);
                acc01 = mfma16bf(A0, B1, acc01);
                acc10 = mfma16bf(A1, B0, acc10);
                acc11 = mfma16bf(A1, B1, acc11);
            }
        } else {
#pragma unroll
            for (int j = 0; j < HK; ++j) {
                acc00 = mfma32(a0[j], b0[j], acc00);
                acc01 = mfma32(a0[j], b1[j], acc01);
                acc10 = mfma32(a1[j], b0[j], acc10);
                acc11 = mfma32(a1[j], b1[j], acc11);
            }
        }
    }
    // Epilogue.  C/D layout: col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m).  Every access is a raw-buffer
    // instruction on a descriptor over this tile's rows [m0, min(m0 + 128, M)): the per-lane byte offset (column, + 4 rows
    // for the upper lane half) is computed once, the row advance is a scalar offset, rows >= M fall outside the descriptor
    // and columns >= N carry offset 0xFFFFFFFF -- both are dropped by the range check.  (The straightforward version, 64-bit
    // address arithmetic and two bounds tests per element, cost as much as a third of a K = 1024 tile: fc1's dX.)
    const int rows_here = (M - m0) < BMN ? (M - m0) : BMN;
    const __amdgpu_buffer_rsrc_t rc = ptmi_rsrc(C + (size_t)m0 * ldc, (unsigned)rows_here * (unsigned)ldc * 4u);
    const __amdgpu_buffer_rsrc_t rbias = ptmi_rsrc(bias_mode ? bias : C, bias_mode == 1 ? (unsigned)M * 4u
                                                                          : (bias_mode == 2 ? (unsigned)N * 4u : 0u));
    const int half4 = 4 * (lane >> 5);
    unsigned pv[2];
    float bn_[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = n0 + wn * 64 + q * 32 + (lane & 31);
        pv[q] = n < N ? (unsigned)(half4 * ldc + n) * 4u : 0xFFFFFFFFu;
        bn_[q] = (bias_mode == 2) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, n < N ? n * 4 : -1, 0, 0)) : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 64 + s * 32 + (r & 3) + 8 * (r >> 2);          // wave-uniform row inside the tile
            const int soff = row * ldc * 4;
            float bm = 0.f;
            if (bias_mode == 1)
                bm = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, half4 * 4, (m0 + row) * 4, 0));
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v = (s == 0) ? (q == 0 ? acc00[r] : acc01[r]) : (q == 0 ? acc10[r] : acc11[r]);
                if (bias_mode == 1) v += bm;
                else if (bias_mode == 2) v += bn_[q];
                if (accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)pv[q], soff, 0));
                if (relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, (int)pv[q], soff, 0);
            }
        }
    }
}

// C = (accumulate ? C : 0) + sum_z partial[z] (z ascending: deterministic) + bias, ReLU -- the epilogue of a split-K GEMM
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ C,
                                                                 const float* __restrict__ bias, int M, int N, int ldc,
                                                                 int S, int bias_mode, int relu, int accumulate)
{
    const int64_t total = (int64_t)M * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
        float v = partial[i];
        for (int z = 1; z < S; ++z) v += partial[(int64_t)z * total + i];
        if (bias_mode == 1) v += bias[m];
        else if (bias_mode == 2) v += bias[n];
        float* dst = C + (int64_t)m * ldc + n;
        if (accumulate) v += *dst;
        if (relu) v = fmaxf(v, 0.f);
        *dst = v;
    }
}

// Split-K factor for an (M, N, K) GEMM.  A 128 x 128 tile is one workgroup and two workgroups share a CU (512 slots), so
//   * few tiles and a long K -- the weight gradients of the box head: fc2 dW = 64 tiles, cls/bbox dW = 8 tiles, K = 16 384
//     ROIs -- run as ONE round whose length is set by K alone (0.9 ms for 34 GFLOP), and
//   * a tile count just above a multiple of 512 -- fc1 dW: 1568 tiles = 3.06 rounds -- idles the chip for most of its last
//     round (77 % utilisation).
// Cost model: rounds(S) x slice length + the traffic of writing / re-reading S partial outputs; S = 1 keeps the plain path.
int pick_splitk(int m, int n, int k, int batch)
{
    if (batch > 1 || k < 1024) return 1;
    const int64_t tiles = (int64_t)cdiv(m, BMN) * cdiv(n, BMN);
    const double P = 512.0;
    const double step_us = 3.4;                          // one 32-deep K step of a workgroup sharing its CU with another
    const double bw = 4.0e6;                             // bytes per microsecond for the partial buffers (~4 TB/s)
    static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
    int best = 1;
    double best_t = 1e30;
    for (int S : cand) {
        const int chunk = cdiv(cdiv(k, S), 32) * 32;
        if (S > 1 && (chunk < 256 || (int64_t)(S - 1) * chunk >= k)) continue;
        const double rounds = (double)((tiles * S + (int64_t)P - 1) / (int64_t)P);
        double t = rounds * (chunk / 32.0) * step_us;
        if (S > 1) t += 5.0 + (2.0 * S + 1.0) * (double)m * n * 4.0 / bw;
        if (t < best_t * 0.97) { best_t = t; best = S; }     // prefer the smaller factor unless the gain is real
    }
    return best;
}

// out[j] (+)= sum_i a[i][j]: a workgroup owns 16 columns; 16 row-partitions are reduced through LDS in a
// fixed order (deterministic).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, float* __restrict__ out,
                                                      int rows, int cols, int accumulate)
{
    __shared__ float sm[16][17];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (c < cols) {
        int r = part;
        for (; r + 48 < rows; r += 64) {
            acc0 += a[(size_t)r * cols + c];
            acc1 += a[(size_t)(r + 16) * cols + c];
            acc2 += a[(size_t)(r + 32) * cols + c];
            acc3 += a[(size_t)(r + 48) * cols + c];
        }
        for (; r < rows; r += 16) acc0 += a[(size_t)r * cols + c];
    }
    sm[part][cl] = (acc0 + acc1) + (acc2 + acc3);
    __syncthreads();
    if (part == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += sm[p][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// tall matrices (bias gradients of the 1x1 RPN heads: 200 000 rows x 15 / 60 columns): S row ranges -> partial[S][cols],
// then colsum_kernel over the partials (fixed order: deterministic).  One workgroup per 16 columns took ~1 ms per call.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, float* __restrict__ partial,
                                                              int rows, int cols, int per)
{
    __shared__ float sm[16][17];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int r0 = blockIdx.y * per, r1 = min(r0 + per, rows);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (c < cols) {
        int r = r0 + part;
        for (; r + 48 < r1; r += 64) {
            acc0 += a[(size_t)r * cols + c];
            acc1 += a[(size_t)(r + 16) * cols + c];
            acc2 += a[(size_t)(r + 32) * cols + c];
            acc3 += a[(size_t)(r + 48) * cols + c];
        }
        for (; r < r1; r += 16) acc0 += a[(size_t)r * cols + c];
    }
    sm[part][cl] = (acc0 + acc1) + (acc2 + acc3);
    __syncthreads();
    if (part == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += sm[p][cl];
        partial[(size_t)blockIdx.y * cols + c] = t;
    }
}

static int colsum_splits(int rows, int cols)
{
    const int groups = cdiv(cols, 16);
    if (groups >= 128 || rows < 4096) return 1;
    int s = cdiv(512, groups);
    if (s > cdiv(rows, 1024)) s = cdiv(rows, 1024);
    return s < 1 ? 1 : s;
}

__global__ __launch_bounds__(256) void rowsum_batched_kernel(const float* __restrict__ a, float* __restrict__ out,
                                                             int batch, int rows, int cols, int accumulate)
{
    __shared__ float sm[4];
    const int r = blockIdx.x;
    float acc = 0.f;
    for (int b = 0; b < batch; ++b) {
        const float* p = a + ((size_t)b * rows + r) * cols;
        for (int i = threadIdx.x; i < cols; i += 256) acc += p[i];
    }
    const float t = block_sum_256(acc, sm);
    if (threadIdx.x == 0) out[r] = accumulate ? out[r] + t : t;
}

int gemm_impl(bool bf, const float* a, const float* b, float* c, const float* bias, int m, int n, int k, int lda,
              int ldb, int ldc, int ta, int tb, int bias_mode, int relu, int accumulate, int batch,
              int64_t stride_a, int64_t stride_b, int64_t stride_c, float* ws, int64_t ws_floats, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(a && b && c && m >= 0 && n >= 0 && k > 0 && batch > 0, "gemm_f32: bad args");
    PTMI_CHECK_ARG(bias_mode == 0 || bias, "gemm_f32: bias missing");
    if (m == 0 || n == 0) return 0;
    const int tilesM = cdiv(m, BMN), tilesN = cdiv(n, BMN);
    hipStream_t st = (hipStream_t)s;
    // A k-fast  <=> stored (M,K) row-major (ta == 0);  B k-fast <=> stored (N,K) (tb == 1)
    const bool ak = (ta == 0), bk = (tb != 0);
    PTMI_CHECK_ARG((int64_t)128 * lda * 4 < (1ll << 31) && (int64_t)128 * ldb * 4 < (1ll << 31) &&
                       (int64_t)128 * ldc * 4 < (1ll << 31),
                   "gemm_f32: leading dimension too large for 32-bit buffer offsets (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    // split-K only with a workspace of the size ptmi_gemm_ws_floats() asks for (callers that pass none get the plain path)
    int S = pick_splitk(m, n, k, batch);
    if (S > 1 && (!ws || ws_floats < (int64_t)S * m * n)) S = 1;
    PTMI_CHECK_ARG(S == 1 || (int64_t)128 * n * 4 < (1ll << 31), "gemm_f32: n too large for the split-K partials");
    const int chunk = S > 1 ? cdiv(cdiv(k, S), 32) * 32 : k;
    float* partial = S > 1 ? ws : nullptr;
    dim3 grid((unsigned)(8 * ((tilesM * tilesN + 7) / 8)), (unsigned)batch, (unsigned)S), block(256);
#define L(AK_, BK_)                                                                                                    \
    do {                                                                                                               \
        if (bf)                                                                                                        \
            hipLaunchKernelGGL((gemm_buf_kernel<AK_, BK_, 32, true>), grid, block, 0, st, a, b, c, bias, m, n, k, lda,  \
                               ldb, ldc, bias_mode, relu, accumulate, stride_a, stride_b, stride_c, tilesM, tilesN,    \
                               chunk, partial);                                                                        \
        else                                                                                                           \
            hipLaunchKernelGGL((gemm_buf_kernel<AK_, BK_, 32, false>), grid, block, 0, st, a, b, c, bias, m, n, k, lda, \
                               ldb, ldc, bias_mode, relu, accumulate, stride_a, stride_b, stride_c, tilesM, tilesN,    \
                               chunk, partial);                                                                        \
    } while (0)
    if (ak && bk) L(true, true);
    else if (ak && !bk) L(true, false);
    else if (!ak && bk) L(false, true);
    else L(false, false);
#undef L
    PTMI_LAUNCH_CHECK("gemm_f32");
    if (S > 1) {
        int64_t blocks = ((int64_t)m * n + 1023) / 1024;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, partial, c, bias, m, n, ldc,
                           S, bias_mode, relu, accumulate);
        PTMI_LAUNCH_CHECK("gemm_splitk_reduce");
    }
    return 0;
}

}  // namespace

extern "C" {

int64_t ptmi_gemm_ws_floats(int m, int n, int k, int batch)
{
    const int S = pick_splitk(m, n, k, batch);
    return S > 1 ? (int64_t)S * m * n : 0;
}

int ptmi_gemm_f32(const float* a, const float* b, float* c, const float* bias, int m, int n, int k, int lda,
                  int ldb, int ldc, int ta, int tb, int bias_mode, int relu, int accumulate, int batch,
                  int64_t stride_a, int64_t stride_b, int64_t stride_c, float* ws, int64_t ws_floats, ptmi_stream_t s)
{
    return gemm_impl(false, a, b, c, bias, m, n, k, lda, ldb, ldc, ta, tb, bias_mode, relu, accumulate, batch, stride_a,
                     stride_b, stride_c, ws, ws_floats, s);
}

int ptmi_gemm_bf16(const float* a, const float* b, float* c, const float* bias, int m, int n, int k, int lda,
                   int ldb, int ldc, int ta, int tb, int bias_mode, int relu, int accumulate, int batch,
                   int64_t stride_a, int64_t stride_b, int64_t stride_c, float* ws, int64_t ws_floats, ptmi_stream_t s)
{
    return gemm_impl(true, a, b, c, bias, m, n, k, lda, ldb, ldc, ta, tb, bias_mode, relu, accumulate, batch, stride_a,
                     stride_b, stride_c, ws, ws_floats, s);
}

int ptmi_colsum(const float* a, float* out, int rows, int cols, int accumulate, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(a && out && rows >= 0 && cols > 0, "colsum: bad args");
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 16)), dim3(256), 0, (hipStream_t)s, a, out, rows, cols,
                       accumulate);
    PTMI_LAUNCH_CHECK("colsum");
    return 0;
}

int64_t ptmi_colsum_ws_floats(int rows, int cols)
{
    const int S = colsum_splits(rows, cols);
    return S > 1 ? (int64_t)S * cols : 0;
}

int ptmi_colsum_ws(const float* a, float* out, float* ws, int rows, int cols, int accumulate, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(a && out && rows >= 0 && cols > 0, "colsum_ws: bad args");
    const int S = colsum_splits(rows, cols);
    if (S <= 1 || !ws) return ptmi_colsum(a, out, rows, cols, accumulate, s);
    const int per = cdiv(rows, S);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(cols, 16), S), dim3(256), 0, (hipStream_t)s, a, ws, rows, cols, per);
    PTMI_LAUNCH_CHECK("colsum_partial");
    return ptmi_colsum(ws, out, S, cols, accumulate, s);
}

int ptmi_rowsum_batched(const float* a, float* out, int batch, int rows, int cols, int accumulate,
                        ptmi_stream_t s)
{
    PTMI_CHECK_ARG(a && out && batch > 0 && rows > 0 && cols > 0, "rowsum_batched: bad args");
    hipLaunchKernelGGL(rowsum_batched_kernel, dim3(rows), dim3(256), 0, (hipStream_t)s, a, out, batch, rows, cols,
                       accumulate);
    PTMI_LAUNCH_CHECK("rowsum_batched");
    return 0;
}

}  // extern "C"
