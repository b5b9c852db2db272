// p8gemm.hip -- bf16-STORAGE GEMM for the box head's large Linear layer under SOLVER.AMP.ENABLED (reference AMP flag
// pt/engine/trainer.py:98; FastRCNNConvFCHead fc1 25088 -> 1024, D2 box head reached at pt/modeling/roi_heads/roi_heads.py:126-128:
// cuBLAS bf16 GEMMs under autocast).  gfx950 only.
//
// ptmi_gemm_bf16 (gemm.hip) keeps fp32 operands in HBM / LDS and rounds between LDS and the MFMA: 0.13 of the bf16 MFMA peak.
// Here both operands are bf16 in the "P8 matrix" layout  t[ceil(K/8)][ROWS][8]  -- the k index in OCTETS, one 16-byte vector per
// (row, k octet), rows contiguous inside an octet plane (the pixel-major layout of p8.hip with rows for pixels):
//   * a DMA instruction copies 64 consecutive rows of one octet plane (1 KB contiguous in HBM) to LDS lane-linearly, and
//   * a lane's MFMA operand (row r, k = 8 h .. 8 h + 7) is ONE ds_read_b128; consecutive lanes read consecutive rows: no conflicts,
// for BOTH operands of  C[M][N] = A[M][K] . B[N][K]^T  (fp32 out, + bias[N], + ReLU).  All three products of a Linear layer are
// of that form once each operand has been laid out with its contraction index as k:
//   forward  Y = X W^T        A = X  (rows r, k = input feature)      B = W   (rows n, k = input feature)
//   dX = dZ W                 A = dZ (rows r, k = output feature)     B = W^T (rows = input feature, k = output feature)
//   dW = dZ^T X               A = dZ^T (rows n, k = r)                B = X^T (rows = input feature, k = r)
// ptmi_p8m_pack builds such an operand from an fp32 row-major matrix, either way round (k along the source's rows or columns).
//
//   p8_gemm_nt_kernel: workgroup = 8 waves (two per SIMD, <= 256 registers each) = 256 x 256 outputs; wave = 128 (M) x 64 (N) =
//   8 accumulator tiles; K in chunks of 64 (8 octets): LDS stage = A [8][256 rows] + B [8][256 rows] vectors = 64 KB, two stages,
//   the next chunk's 8 DMA instructions per wave issued at the head of the chunk (the partner wave of the SIMD keeps the matrix
//   pipe busy meanwhile), one barrier per chunk; split-K over chunk ranges for long-K / few-tile shapes (fp32 partials, fixed-order
//   reduction: bitwise reproducible).
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) ptmi_bf16x8 glds_bf16x8_t;
typedef __attribute__((address_space(3))) void glds_void_t;
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// ------------------------------------------------------------------------------------------------ operand packs
// dst[ko][row][e] = src element (row, k = 8 ko + e), zero beyond K.
// k_major = 1: src[row * ld + k]  (k contiguous in the source: a 64 row x 64 k tile goes through LDS so that reads AND writes coalesce)
// k_major = 0: src[k * ld + row]  (rows contiguous in the source: eight strided coalesced reads per vector)
__global__ __launch_bounds__(256) void p8m_pack_kmajor_kernel(const float* __restrict__ src, u32x4* __restrict__ dst, int rows, int K,
                                                              long long ld)
{
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, k = i & 63;
        tile[r][k] = (r0 + r < rows && k0 + k < K) ? src[(long long)(r0 + r) * ld + k0 + k] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * 64; i += 256) {
        const int ko = i >> 6, r = i & 63;
        if (r0 + r < rows && k0 + 8 * ko < K) {
            const float* t = &tile[r][8 * ko];
            dst[(long long)(k0 / 8 + ko) * rows + r0 + r] = (u32x4){pk_bf16(t[0], t[1]), pk_bf16(t[2], t[3]), pk_bf16(t[4], t[5]), pk_bf16(t[6], t[7])};
        }
    }
}

__global__ __launch_bounds__(256) void p8m_pack_rowmajor_kernel(const float* __restrict__ src, u32x4* __restrict__ dst, int rows, int K,
                                                                long long ld)
{
    const int row = blockIdx.x * 256 + threadIdx.x, ko = blockIdx.y;
    if (row >= rows) return;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (8 * ko + j < K) ? src[(long long)(8 * ko + j) * ld + row] : 0.f;
    dst[(long long)ko * rows + row] = (u32x4){pk_bf16(e[0], e[1]), pk_bf16(e[2], e[3]), pk_bf16(e[4], e[5]), pk_bf16(e[6], e[7])};
}

// ------------------------------------------------------------------------------------------------ GEMM
constexpr int GT_ = 512;
constexpr int G_BM = 256, G_BN = 256, G_BK = 64;
constexpr int G_ABYTES = (G_BK / 8) * G_BM * 16;          // 32 KB
constexpr int G_STAGE_ = 2 * G_ABYTES;                    // 64 KB

// TOUT: the result is stored TRANSPOSED, C[col][row] with row pitch ldc -- the caller swaps the operands (C^T = B . A^T), so that a
// lane's four consecutive accumulator rows are four consecutive elements of one output row: 16-byte stores instead of 4-byte ones
// (fc1's dX writes 1.6 GB of fp32 and is store-bound; no bias / ReLU / split-K in this form)
template <bool TOUT>
__global__ __launch_bounds__(GT_, 2) void p8_gemm_nt_kernel(const u16* __restrict__ A, const u16* __restrict__ B, float* __restrict__ C,
                                                            const float* __restrict__ bias, float* __restrict__ ws, int M, int N, int K, int ldc,
                                                            int relu, int tilesN, int S, int nChunks)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * G_STAGE_];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;                       // wave tile: rows wm * 128 .., columns wn * 64 ..
    const int tile = blockIdx.x / S, split = blockIdx.x - tile * S;
    const int m0 = (tile / tilesN) * G_BM, n0 = (tile % tilesN) * G_BN;
    const int c0 = (int)((long long)nChunks * split / S), c1 = (int)((long long)nChunks * (split + 1) / S);
    const int KO = (K + 7) / 8;

    const __amdgpu_buffer_rsrc_t ra = ptmi_rsrc(A, (unsigned)((long long)KO * M * 16)), rb = ptmi_rsrc(B, (unsigned)((long long)KO * N * 16));
    // DMA pieces: a chunk = 8 octet planes x 256 rows per operand = 32 wave instructions each; wave w issues pieces 4 w .. 4 w + 3 of A
    // and of B: piece p -> octet p / 4, rows 64 (p % 4) + lane
    unsigned a_lane[4], b_lane[4];
    bool a_ok[4], b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = 4 * wave + i, ko = p >> 2, row = 64 * (p & 3) + lane;
        a_ok[i] = m0 + row < M;
        b_ok[i] = n0 + row < N;
        a_lane[i] = (unsigned)(((long long)ko * M + m0 + row) * 16);
        b_lane[i] = (unsigned)(((long long)ko * N + n0 + row) * 16);
    }
    auto issue = [&](int chunk, int st) {
        char* base = lds + st * G_STAGE_;
        const int ko0 = chunk * (G_BK / 8);
        const unsigned a_adv = (unsigned)((long long)ko0 * M * 16), b_adv = (unsigned)((long long)ko0 * N * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = 4 * wave + i;
            const bool live = ko0 + (p >> 2) < KO;                // (wave-uniform) octets beyond K: zero
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (glds_void_t*)(base + p * 1024), 16, (int)(a_ok[i] && live ? a_lane[i] + a_adv : 0xFFFFFFFFu), 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = 4 * wave + i;
            const bool live = ko0 + (p >> 2) < KO;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (glds_void_t*)(base + G_ABYTES + p * 1024), 16, (int)(b_ok[i] && live ? b_lane[i] + b_adv : 0xFFFFFFFFu), 0, 0, 0);
        }
    };

    const int h = lane >> 5, r = lane & 31;
    const int a_off = (h * G_BM + wm * 128 + r) * 16;               // + (2 s) octets * BM * 16 + mt * 512
    const int b_off = G_ABYTES + (h * G_BN + wn * 64 + r) * 16;     // + (2 s) octets * BN * 16 + nt * 512

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    if (c0 < c1) issue(c0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int chunk = c0; chunk < c1; ++chunk) {
        const int st = (chunk - c0) & 1;
        const char* base = lds + st * G_STAGE_;
        if (chunk + 1 < c1) issue(chunk + 1, st ^ 1);              // into the other stage: everybody left it at the last barrier
#pragma unroll
        for (int s = 0; s < G_BK / 16; ++s) {
            ptmi_bf16x8 a[4], b[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = *(const volatile glds_bf16x8_t*)(base + a_off + (2 * s) * G_BM * 16 + m * 512);
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = *(const volatile glds_bf16x8_t*)(base + b_off + (2 * s) * G_BN * 16 + n * 512);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // C layout: column = lane & 31 = n, row = (reg & 3) + 8 (reg >> 2) + 4 h = m: a half wave writes 128 contiguous bytes per register
    const int nn0 = n0 + wn * 64 + r;
    if constexpr (TOUT) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = nn0 + n * 32;
                if (col >= N) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = m0 + wm * 128 + m * 32 + 8 * q + 4 * h;          // rows row .. row + 3 = registers 4 q .. 4 q + 3
                    float* o = C + (size_t)col * ldc + row;
                    if (row + 3 < M && (ldc & 3) == 0) *reinterpret_cast<f32x4*>(o) = (f32x4){acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
                    else
                        for (int e = 0; e < 4; ++e)
                            if (row + e < M) o[e] = acc[m][n][4 * q + e];
                }
            }
        return;
    }
    float* out = S > 1 ? ws + (size_t)split * M * N : C;
    const int ld = S > 1 ? N : ldc;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = nn0 + n * 32;
            if (col >= N) continue;
            const float bv = (S == 1 && bias) ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 128 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < M) {
                    float v = acc[m][n][e] + bv;
                    if (S == 1 && relu) v = fmaxf(v, 0.f);
                    out[(size_t)row * ld + col] = v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void p8_gemm_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, const float* __restrict__ bias,
                                                             int M, int N, int ldc, int S, int relu)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * N) return;
    const int row = (int)(i / N), col = (int)(i - (long long)row * N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += ws[(size_t)s * M * N + i];
    if (bias) v += bias[col];
    if (relu) v = fmaxf(v, 0.f);
    C[(size_t)row * ldc + col] = v;
}

inline int p8_gemm_splits(int m, int n, int k)
{
    const long long tiles = (long long)cdiv(m, G_BM) * cdiv(n, G_BN);
    const int chunks = cdiv(k, G_BK);
    if (tiles >= 192 || chunks < 16) return 1;                    // enough workgroups for the 256 CUs, or nothing to split
    int S = (int)(256 / tiles);
    if (S > chunks / 8) S = chunks / 8;                           // >= 8 chunks (512 k) per split
    return S < 1 ? 1 : S;
}

}  // namespace

extern "C" {

int64_t ptmi_p8m_elems(int rows, int k) { return rows > 0 && k > 0 ? (int64_t)cdiv(k, 8) * rows * 8 : 0; }

int ptmi_p8m_pack(const float* src, void* dst, int rows, int k, int64_t ld, int k_major, ptmi_stream_t s)
{
    PTMI_CHECK_ARG(src && dst && rows > 0 && k > 0 && ld > 0, "p8m_pack: bad args");
    if (k_major)
        hipLaunchKernelGGL(p8m_pack_kmajor_kernel, dim3(cdiv(rows, 64), cdiv(k, 64)), dim3(256), 0, (hipStream_t)s, src, (u32x4*)dst, rows, k,
                           (long long)ld);
    else
        hipLaunchKernelGGL(p8m_pack_rowmajor_kernel, dim3(cdiv(rows, 256), cdiv(k, 8)), dim3(256), 0, (hipStream_t)s, src, (u32x4*)dst, rows, k,
                           (long long)ld);
    PTMI_LAUNCH_CHECK("p8m_pack");
    return 0;
}

int64_t ptmi_p8_gemm_nt_ws_floats(int m, int n, int k)
{
    const int S = p8_gemm_splits(m, n, k);
    return S > 1 ? (int64_t)S * m * n : 0;
}

int ptmi_p8_gemm_nt_fits(int m, int n, int k)
{
    return m > 0 && n > 0 && k > 0 && (int64_t)cdiv(k, 8) * m * 16 < (1ll << 32) && (int64_t)cdiv(k, 8) * n * 16 < (1ll << 32);
}

int ptmi_p8_gemm_nt(const void* a, const void* b, float* c, const float* bias, float* ws, int m, int n, int k, int ldc, int relu,
                    ptmi_stream_t s)
{
    PTMI_CHECK_ARG(a && b && c && m > 0 && n > 0 && k > 0 && (ldc >= n || (relu & 2)), "p8_gemm_nt: bad args");
    if (relu & 2) {         // flag bit 1: transposed store C[n][m] (row pitch ldc >= m); operands as given; no bias / ReLU / split-K
        PTMI_CHECK_ARG(!bias && !(relu & 1) && ldc >= m, "p8_gemm_nt: the transposed-store form takes no bias / ReLU and needs ldc >= m");
        PTMI_CHECK_ARG((int64_t)cdiv(k, 8) * m * 16 < (1ll << 32) && (int64_t)cdiv(k, 8) * n * 16 < (1ll << 32),
                       "p8_gemm_nt: operands beyond the 32-bit buffer offsets (m=%d n=%d k=%d)", m, n, k);
        const int tilesN_ = cdiv(n, G_BN), tiles_ = cdiv(m, G_BM) * tilesN_;
        hipLaunchKernelGGL(p8_gemm_nt_kernel<true>, dim3((unsigned)tiles_), dim3(GT_), 0, (hipStream_t)s, (const u16*)a, (const u16*)b, c, bias, ws, m,
                           n, k, ldc, 0, tilesN_, 1, cdiv(k, G_BK));
        PTMI_LAUNCH_CHECK("p8_gemm_nt(transposed store)");
        return 0;
    }
    PTMI_CHECK_ARG((int64_t)cdiv(k, 8) * m * 16 < (1ll << 32) && (int64_t)cdiv(k, 8) * n * 16 < (1ll << 32),
                   "p8_gemm_nt: operands beyond the 32-bit buffer offsets (m=%d n=%d k=%d)", m, n, k);
    const int S = p8_gemm_splits(m, n, k);
    PTMI_CHECK_ARG(S == 1 || ws, "p8_gemm_nt: this shape runs split-K and needs the workspace of ptmi_p8_gemm_nt_ws_floats");
    const int tilesN = cdiv(n, G_BN), tiles = cdiv(m, G_BM) * tilesN;
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(p8_gemm_nt_kernel<false>, dim3((unsigned)(tiles * S)), dim3(GT_), 0, st, (const u16*)a, (const u16*)b, c, bias, ws, m, n, k, ldc,
                       relu, tilesN, S, cdiv(k, G_BK));
    PTMI_LAUNCH_CHECK("p8_gemm_nt");
    if (S > 1) {
        hipLaunchKernelGGL(p8_gemm_reduce_kernel, dim3((unsigned)cdiv64((int64_t)m * n, 256)), dim3(256), 0, st, ws, c, bias, m, n, ldc, S, relu);
        PTMI_LAUNCH_CHECK("p8_gemm_reduce");
    }
    return 0;
}

}  // extern "C"
