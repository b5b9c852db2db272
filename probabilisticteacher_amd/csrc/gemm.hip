// gemm.hip -- fp32 GEMM on v_mfma_f32_32x32x2_f32 for gfx950, all four transpose combinations,
// bounds-checked, batched, with bias / ReLU / accumulate epilogues.
//
// Replaces the cuBLAS Linear kernels behind D2's FastRCNNConvFCHead + the box predictor
// (pt/modeling/roi_heads/roi_heads.py:127-128, pt/modeling/roi_heads/fast_rcnn.py:164) and the
// 1x1 convolutions of StandardRPNHead (pt/modeling/proposal_generator/rpn.py:96), fwd + bwd.
//
// 128x128 workgroup tile, 4 wave64s in 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles, BK = 32.
// Both operands are staged into LDS as T[k][m] with pitch 129 (odd) so that
//   * the MFMA fragment read (lane l -> T[k0 + (l>>5)][m0 + (l&31)]) walks consecutive banks, and
//   * a k-fastest global operand (row-major A, or B stored (N,K)) can be written by lanes that walk
//     k (coalesced 128-B global rows) without bank conflicts (stride 129 = 1 mod 32).
// Accumulation is strictly k-ordered inside a tile chain => bitwise reproducible run to run.
#include "common.h"

namespace {

constexpr int BMN = 128, BK = 32, PITCH = 129;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Stage a 128(mn) x 32(k) operand tile into T[k][mn].
//   KFAST = true : element (mn,k) at p[mn*ld + k]   (k contiguous)
//   KFAST = false: element (mn,k) at p[k*ld + mn]   (mn contiguous)
template <bool KFAST>
__device__ __forceinline__ void gload_tile(const float* __restrict__ p, int ld, int mn0, int k0, int MN,
                                           int K, float (&reg)[16], int tid)
{
    if constexpr (KFAST) {
        const int k = k0 + (tid & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int mn = mn0 + (tid >> 5) + 8 * i;
            reg[i] = (mn < MN && k < K) ? p[(size_t)mn * ld + k] : 0.f;
        }
    } else {
        const int mn = mn0 + (tid & 127);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = k0 + (tid >> 7) + 2 * i;
            reg[i] = (mn < MN && k < K) ? p[(size_t)k * ld + mn] : 0.f;
        }
    }
}

template <bool KFAST>
__device__ __forceinline__ void lstore_tile(float* T, const float (&reg)[16], int tid)
{
    if constexpr (KFAST) {
        const int k = tid & 31;
#pragma unroll
        for (int i = 0; i < 16; ++i) T[k * PITCH + (tid >> 5) + 8 * i] = reg[i];
    } else {
        const int mn = tid & 127;
#pragma unroll
        for (int i = 0; i < 16; ++i) T[((tid >> 7) + 2 * i) * PITCH + mn] = reg[i];
    }
}

template <bool AK, bool BKF>
__global__ __launch_bounds__(256, 3) void gemm_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
    const float* __restrict__ bias, int M, int N, int K, int lda, int ldb, int ldc, int bias_mode, int relu,
    int accumulate, int64_t sa, int64_t sb, int64_t sc, int tilesN)
{
    __shared__ float lds[2 * BK * PITCH];
    float* As = lds;
    float* Bs = lds + BK * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm = blockIdx.x / tilesN, bn = blockIdx.x % tilesN;
    const int m0 = bm * BMN, n0 = bn * BMN;
    const int b = blockIdx.y;
    A += (size_t)b * sa;
    B += (size_t)b * sb;
    C += (size_t)b * sc;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float ra[16], rb[16];
    gload_tile<AK>(A, lda, m0, 0, M, K, ra, tid);
    gload_tile<BKF>(B, ldb, n0, 0, N, K, rb, tid);
    const float* al = As + (lane >> 5) * PITCH + wm * 64 + (lane & 31);
    const float* bl = Bs + (lane >> 5) * PITCH + wn * 64 + (lane & 31);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
        lstore_tile<AK>(As, ra, tid);
        lstore_tile<BKF>(Bs, rb, tid);
        __syncthreads();
        if (k0 + BK < K) {
            gload_tile<AK>(A, lda, m0, k0 + BK, M, K, ra, tid);
            gload_tile<BKF>(B, ldb, n0, k0 + BK, N, K, rb, tid);
        }
#pragma unroll 4
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a0 = al[2 * kk * PITCH], a1 = al[2 * kk * PITCH + 32];
            const float b0 = bl[2 * kk * PITCH], b1 = bl[2 * kk * PITCH + 32];
            acc00 = mfma32(a0, b0, acc00);
            acc01 = mfma32(a0, b1, acc01);
            acc10 = mfma32(a1, b0, acc10);
            acc11 = mfma32(a1, b1, acc11);
        }
    }
    // C/D layout: col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = n0 + wn * 64 + q * 32 + (lane & 31);
            if (n >= N) continue;
            const float bn_ = (bias_mode == 2) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= M) continue;
                float v = (s == 0) ? (q == 0 ? acc00[r] : acc01[r]) : (q == 0 ? acc10[r] : acc11[r]);
                if (bias_mode == 1) v += bias[m];
                else if (bias_mode == 2) v += bn_;
                float* dst = C + (size_t)m * ldc + n;
                if (accumulate) v += *dst;
                if (relu) v = fmaxf(v, 0.f);
                *dst = v;
            }
        }
    }
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

}  // namespace

extern "C" {

int ptmi_gemm_f32(const float* a, const float* b, float* c, const float* bias, int m, int n, int k, int lda,
                  int ldb, int ldc, int ta, int tb, int bias_mode, int relu, int accumulate, int batch,
                  int64_t stride_a, int64_t stride_b, int64_t stride_c, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(a && b && c && m >= 0 && n >= 0 && k > 0 && batch > 0, "gemm_f32: bad args");
    PTMI_CHECK_ARG(bias_mode == 0 || bias, "gemm_f32: bias missing");
    if (m == 0 || n == 0) return 0;
    const int tilesM = cdiv(m, BMN), tilesN = cdiv(n, BMN);
    dim3 grid((unsigned)(tilesM * tilesN), (unsigned)batch), block(256);
    hipStream_t st = (hipStream_t)s;
    // A k-fast  <=> stored (M,K) row-major (ta == 0);  B k-fast <=> stored (N,K) (tb == 1)
    const bool ak = (ta == 0), bk = (tb != 0);
#define L(AK_, BK_)                                                                                     \
    hipLaunchKernelGGL((gemm_f32_kernel<AK_, BK_>), grid, block, 0, st, a, b, c, bias, m, n, k, lda, ldb, ldc, \
                       bias_mode, relu, accumulate, stride_a, stride_b, stride_c, tilesN)
    if (ak && bk) L(true, true);
    else if (ak && !bk) L(true, false);
    else if (!ak && bk) L(false, true);
    else L(false, false);
#undef L
    PTMI_LAUNCH_CHECK("gemm_f32");
    return 0;
}

int ptmi_colsum(const float* a, float* out, int rows, int cols, int accumulate, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(a && out && rows >= 0 && cols > 0, "colsum: bad args");
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 16)), dim3(256), 0, (hipStream_t)s, a, out, rows, cols,
                       accumulate);
    PTMI_LAUNCH_CHECK("colsum");
    return 0;
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
