/*
 * ptmi355.h -- C ABI of libptmi355.so: the MI355X (gfx950 / CDNA4) operators behind the
 * ProbabilisticTeacher teacher+student train step (PTrainer.run_step, reference
 * /root/reference/pt/engine/trainer.py:263-392).
 *
 * The reference has NO native boundary of its own: it is pure Python and reaches native code
 * only through torch/ATen, cuDNN, torchvision C++/CUDA ops and NCCL.  Each entry point below
 * replaces one of those implicitly-invoked native operators; the comment on each names the
 * reference call site (file:line relative to /root/reference) whose operator it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no torch types.  All pointers are DEVICE pointers
 *     unless the name ends in _host.  fp32 tensors are dense row-major NCHW / (R,C) unless noted.
 *   - The caller allocates every output and workspace; the library never allocates, frees or
 *     synchronises; every kernel is enqueued on the passed hipStream_t (void* here).
 *   - Return 0 on success, negative on error; ptmi_last_error() gives a thread-local message.
 *   - int64 indices, uint8 masks, fp32 floats, matching the reference's observable dtypes.
 */
#ifndef PTMI355_H
#define PTMI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ptmi_stream_t; /* hipStream_t */

const char* ptmi_last_error(void);
int ptmi_abi_version(void);

/* ------------------------------------------------------------------ conv stack (N1, N3)
 * replaces cuDNN conv2d 3x3 s1 p1 (+bias) + F.relu_ at pt/modeling/backbone/vgg.py:45-53,66-69
 * and the D2 StandardRPNHead 3x3 conv at pt/modeling/proposal_generator/rpn.py:96.
 * fp32 implicit GEMM on v_mfma_f32_32x32x2_f32.
 *
 * Packed weight layout (built by ptmi_conv3x3_pack_weights): for BM = ptmi_conv3x3_bm(Cout),
 * CK = ptmi_conv3x3_ck(Cin):  [ceil(Cout/BM)][ceil(Cin/CK)][9 taps][CK][BM] fp32, zero padded.
 * mode 0 = forward weights  Wp[co][ci][ky][kx]            = W[co][ci][ky][kx]
 * mode 1 = dgrad weights    (roles of co/ci swapped)      = W[ci][co][2-ky][2-kx]
 *          i.e. dX = conv3x3(dY, pack(W, mode 1)) with "Cin" := Cout(W), "Cout" := Cin(W).
 */
int ptmi_conv3x3_bm(int cout);
int ptmi_conv3x3_ck(int cin);
int64_t ptmi_conv3x3_packed_floats(int cin, int cout);
int ptmi_conv3x3_pack_weights(const float* w, float* wp, int w_cout, int w_cin, int mode,
                              ptmi_stream_t s);
/* epilogue: 0 = y = acc + bias;  1 = y = relu(acc + bias);  2 = y = acc (bias may be NULL);
 *           3 = y = (mask_ref[idx] > 0) ? acc : 0   (dgrad through the producer's ReLU;
 *               mask_ref has the shape of y);
 *           4 = y = maxpool2x2(relu(acc + bias)), y is (n, cout, h/2, w/2): conv + F.relu_ + MaxPool2d of a
 *               VGG block (vgg.py:66-71) in one pass, for blocks whose activations are not kept. */
int ptmi_conv3x3_fwd(const float* x, const float* wp, const float* bias, const float* mask_ref,
                     float* y, int n, int cin, int cout, int h, int w, int epilogue,
                     ptmi_stream_t s);
/* wgrad: dw[co][ci][3][3] (+)= sum_{n,y,x} dy[n][co][y][x] * x[n][ci][y+ky-1][x+kx-1].
 * workspace: ptmi_conv3x3_wgrad_ws_floats(...) fp32; split-K partials reduced in a fixed order
 * (deterministic).  db (may be NULL) = sum dy over n,y,x.  accumulate != 0 adds into dw AND db. */
int64_t ptmi_conv3x3_wgrad_ws_floats(int n, int cin, int cout, int h, int w);
int ptmi_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, float* ws,
                       int n, int cin, int cout, int h, int w, int accumulate, ptmi_stream_t s);
/* Fused Winograd F(2x2,3x3) variant of ptmi_conv3x3_fwd for the layers with >= 32 input channels (conv1_2 .. conv5_3 at
 * pt/modeling/backbone/vgg.py:45-53,66-69 and the RPN 3x3 conv at pt/modeling/proposal_generator/rpn.py:96; cuDNN, which
 * the reference runs there, uses Winograd for these fp32 3x3 s1 layers too).  Same arguments, epilogues and dgrad
 * convention (mode 1 pack) as ptmi_conv3x3_fwd; fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32, 16 instead of 36
 * multiplies per 2x2 output tile and channel pair.  Input transform (B^T d B), the 16 transform-domain GEMMs and the
 * output transform (A^T M A) run in ONE kernel; only x, the packed weights and y touch HBM.
 * Packed weights (ptmi_conv3x3_wino_pack_weights): U = G g G^T as
 * [ceil(Cout/64)][ceil(Cin/8)][8 ci][4 (position row)][64 co][4 (position column)] fp32, zero padded. */
int64_t ptmi_conv3x3_wino_packed_floats(int cin, int cout);
int ptmi_conv3x3_wino_pack_weights(const float* w, float* wp, int w_cout, int w_cin, int mode,
                                   ptmi_stream_t s);
int ptmi_conv3x3_wino_fwd(const float* x, const float* wp, const float* bias, const float* mask_ref,
                          float* y, int n, int cin, int cout, int h, int w, int epilogue,
                          ptmi_stream_t s);
/* 1 if the shape fits the 32-bit buffer offsets of ptmi_conv3x3_wino_fwd / ptmi_conv3x3_wino_wgrad (per-lane byte offsets
 * relative to a workgroup's first image; roughly h*w < 8 M pixels at 64 channels), else 0: the caller then routes the layer
 * to ptmi_conv3x3_fwd / ptmi_conv3x3_wgrad (the Winograd entry points themselves reject such shapes with an error). */
int ptmi_conv3x3_wino_fwd_fits(int cin, int cout, int h, int w);
int ptmi_conv3x3_wino_wgrad_fits(int h, int w);
/* Fused Winograd F(4x4,3x3) variant (round 5, csrc/wino4.hip) of ptmi_conv3x3_wino_fwd for the same layers
 * (pt/modeling/backbone/vgg.py:45-53,66-69, pt/modeling/proposal_generator/rpn.py:96) when the input channel count is a multiple
 * of 8: 36 transform-domain products per 4x4 outputs instead of 64, on v_mfma_f32_16x16x4_f32; interpolation points
 * 0, +-3/4, +-3/2, inf (every constant of B^T and A^T exact in fp32; measured error against an fp64 convolution 1.2e-5 of a
 * unit-scale output at 512 channels -- inside the 1e-4 bar of the parity tests, which are unchanged).  Same arguments,
 * epilogues and dgrad convention (mode 1 pack) as ptmi_conv3x3_fwd.  Packed weights: U = G g G^T as
 * [ceil(Cout/64)][Cin/4][4 ci][9 (groups of four positions p = 6 i + j)][64 co][4] fp32, zero padded.
 * ptmi_conv3x3_wino4_fwd_fits: 1 if the shape is served (cin % 8 == 0 and the 32-bit buffer offsets suffice). */
int64_t ptmi_conv3x3_wino4_packed_floats(int cin, int cout);
int ptmi_conv3x3_wino4_pack_weights(const float* w, float* wp, int w_cout, int w_cin, int mode,
                                    ptmi_stream_t s);
int ptmi_conv3x3_wino4_fwd(const float* x, const float* wp, const float* bias, const float* mask_ref,
                           float* y, int n, int cin, int cout, int h, int w, int epilogue,
                           ptmi_stream_t s);
int ptmi_conv3x3_wino4_fwd_fits(int cin, int cout, int h, int w);
/* ptmi_conv3x3_wino4_fwd with a DYNAMIC tile schedule (round 6).  The kernel runs one persistent workgroup per CU; with the
 * static schedule of ptmi_conv3x3_wino4_fwd workgroup b walks tiles b, b + grid, ..., so a workgroup whose CU is held by another
 * kernel when the launch starts (an RCCL collective overlapping backward: pt/engine/trainer.py:92-95,384 under DDP) still runs
 * its whole share after everyone else has finished -- up to 2x on the launch.  Here the workgroups draw tile ids from eight
 * queues (queue q = the ids = q mod 8 of the same enumeration: the XCD affinity of the static walk is kept; a workgroup starts
 * on the queue of the XCD it runs on and moves on when that is exhausted).  sched: 16 int32 in device memory, ZERO before the
 * first launch; the last workgroup of a launch zeroes it again, so one buffer serves every launch of ONE stream (launches that
 * may overlap -- different streams -- need different buffers).  sched == NULL: the static schedule.  Results are bit-identical
 * either way (a tile's arithmetic does not depend on who computes it). */
int ptmi_conv3x3_wino4_fwd_sched(const float* x, const float* wp, const float* bias, const float* mask_ref,
                                 float* y, int n, int cin, int cout, int h, int w, int epilogue,
                                 int32_t* sched, ptmi_stream_t s);
/* Winograd-domain weight gradient (same contract as ptmi_conv3x3_wgrad; replaces cuDNN's Winograd-nonfused BWD_FILTER
 * for the trainable 3x3 layers with >= 64 input and output channels): dU_p[co][ci] = sum over tiles of (A dY A^T)_p V_p on v_mfma_f32_32x32x2_f32 (16 instead
 * of 36 multiplies per tile and channel pair), split over contiguous tile ranges whose partials (workspace
 * [split][16][Cout][Cin] fp32, ptmi_conv3x3_wino_wgrad_ws_floats) are summed in a fixed order and mapped back by
 * dW = G^T dU G; db = sum of dy per channel (position (1,1) of A dY A^T is the tile's sum), accumulated by the same
 * workgroups and appended to the workspace as [split][Cout].  Deterministic. */
int64_t ptmi_conv3x3_wino_wgrad_ws_floats(int n, int cin, int cout, int h, int w);
int ptmi_conv3x3_wino_wgrad(const float* x, const float* dy, float* dw, float* db, float* ws, int n,
                            int cin, int cout, int h, int w, int accumulate, ptmi_stream_t s);
/* F(4x4,3x3)-domain weight gradient (round 5, csrc/wino4w.hip; same contract as ptmi_conv3x3_wgrad / ptmi_conv3x3_wino_wgrad;
 * replaces cuDNN's BWD_FILTER at pt/modeling/backbone/vgg.py:45-53,66-69, pt/modeling/proposal_generator/rpn.py:96):
 * dU_p[co][ci] = sum over tiles of (A dY A^T)_p V_p for the 36 positions of the 6x6 transform domain on v_mfma_f32_32x32x2_f32
 * (36 multiplies per 4x4 tile and channel pair instead of 64 / 144), split over contiguous tile ranges whose partials (workspace
 * [split][36][Cout][Cin] fp32 + bias partials [split][ceil(Cin/32)][Cout], ptmi_conv3x3_wino4_wgrad_ws_floats) are summed in a fixed
 * order and mapped back by dW = G^T dU G.  Deterministic.  ptmi_conv3x3_wino4_wgrad_fits: the 32-bit buffer offsets suffice. */
int ptmi_conv3x3_wino4_wgrad_fits(int h, int w);
int64_t ptmi_conv3x3_wino4_wgrad_ws_floats(int n, int cin, int cout, int h, int w);
int ptmi_conv3x3_wino4_wgrad(const float* x, const float* dy, float* dw, float* db, float* ws, int n,
                             int cin, int cout, int h, int w, int accumulate, ptmi_stream_t s);
/* ptmi_conv3x3_wino4_* with the 36 transform-domain positions SPLIT over the two waves of a tile row (round 6, csrc/wino4p.hip):
 * same layers (pt/modeling/backbone/vgg.py:45-53,66-69, pt/modeling/proposal_generator/rpn.py:96), same contract, epilogues, dgrad
 * convention (mode 1 pack), shape limits (ptmi_conv3x3_wino4p_fwd_fits == ptmi_conv3x3_wino4_fwd_fits) and tile schedules
 * (sched as ptmi_conv3x3_wino4_fwd_sched; NULL = static).  A wave owns all 64 output channels x 18 positions instead of 32 channels
 * x 36 positions: it transforms only its three rows of every window (72 instead of 144 FMAs per lane and 4-channel chunk -- beside
 * fp32 MFMAs every FMA is issue time), and the two waves exchange partial 4x4 outputs through LDS in the epilogue.  The packed
 * weights have their own layout (ptmi_conv3x3_wino4p_pack_weights); results agree with ptmi_conv3x3_wino4_fwd to fp32 rounding
 * (the inverse transform's row sum is associated differently), not bit for bit. */
int64_t ptmi_conv3x3_wino4p_packed_floats(int cin, int cout);
int ptmi_conv3x3_wino4p_pack_weights(const float* w, float* wp, int w_cout, int w_cin, int mode,
                                     ptmi_stream_t s);
int ptmi_conv3x3_wino4p_fwd(const float* x, const float* wp, const float* bias, const float* mask_ref,
                            float* y, int n, int cin, int cout, int h, int w, int epilogue,
                            ptmi_stream_t s);
int ptmi_conv3x3_wino4p_fwd_sched(const float* x, const float* wp, const float* bias,
                                  const float* mask_ref, float* y, int n, int cin, int cout, int h,
                                  int w, int epilogue, int32_t* sched, ptmi_stream_t s);
int ptmi_conv3x3_wino4p_fwd_fits(int cin, int cout, int h, int w);
/* The two Winograd-domain weight-gradient kernels with `waves` fills of the chip's one-workgroup-per-CU slots instead of one
 * (round 6): their workgroups each own a contiguous share of the (image, tile) list, so with one fill a workgroup whose CU is held by
 * another kernel when the launch starts -- an RCCL collective overlapping backward, pt/engine/trainer.py:92-95,384 under DDP --
 * runs its whole share after everyone else has finished: 2x on the launch (tools/exp/contention.py: 33 -> 65 ms per 8 + 8 step
 * with 8 of 256 CUs held).  With `waves` > 1 (clamped to 16) the later workgroups go to whichever CU frees up first: 45 ms at
 * waves = 4, for 8 % more time when nothing else runs (one more prologue and partial-sum store per workgroup).  Same contract and
 * the same bit-exact repeatability as the one-wave entry points (= waves 1); the workspace grows with the split count
 * (ptmi_conv3x3_wino*_wgrad_ws_floats_waves with the same `waves`). */
int64_t ptmi_conv3x3_wino_wgrad_ws_floats_waves(int n, int cin, int cout, int h, int w, int waves);
int ptmi_conv3x3_wino_wgrad_waves(const float* x, const float* dy, float* dw, float* db, float* ws,
                                  int n, int cin, int cout, int h, int w, int accumulate, int waves,
                                  ptmi_stream_t s);
int64_t ptmi_conv3x3_wino4_wgrad_ws_floats_waves(int n, int cin, int cout, int h, int w, int waves);
int ptmi_conv3x3_wino4_wgrad_waves(const float* x, const float* dy, float* dw, float* db, float* ws,
                                   int n, int cin, int cout, int h, int w, int accumulate, int waves,
                                   ptmi_stream_t s);
/* ------------------------------------------------------------------ bf16 STORAGE path of the conv stack ("P8", round 4)
 * SOLVER.AMP.ENABLED (reference pt/engine/trainer.py:98; BASELINE configs[4]): under autocast the reference's cuDNN convolutions
 * (pt/modeling/backbone/vgg.py:45-53,66-69, pt/modeling/proposal_generator/rpn.py:96) read and write bf16 activations.  The
 * entry points below keep activations and activation gradients in bf16 in HBM and LDS, in the P8 layout
 *     t[2 ceil(C/16)][ROWS = N (H + 1) + 1][WS = W + 1][8]  bf16
 * pixel (n, r, c) at row n (H + 1) + 1 + r, column 1 + c; rows n (H + 1) and column 0 hold zeros (the convolution's zero padding
 * is part of the tensor: a tap is a flat pixel offset, no kernel tests an image edge); every entry point writes them as zeros.
 * Accumulation, bias, losses and weight gradients are fp32; a value is rounded to bf16 (nearest even) when it is stored.
 * ptmi_p8_plane_pixels = ROWS * WS (a tensor holds ptmi_p8_planes(C) * that many 16-byte pixel vectors). */
int64_t ptmi_p8_plane_pixels(int n, int h, int w);
/* A tensor of c channels holds ptmi_p8_planes(c) = 2 ceil(c/16) planes (whole 16-channel chunks; channels >= c are zeros: the
 * 3-channel image is the 16-channel input of the first layer). */
int ptmi_p8_planes(int c);
/* fp32 NCHW <-> P8 */
int ptmi_p8_from_nchw(const float* x, void* y, int n, int c, int h, int w, ptmi_stream_t s);
int ptmi_p8_to_nchw(const void* x, float* y, int n, int c, int h, int w, ptmi_stream_t s);
/* MaxPool2d(2,2) forward / backward (vgg.py:59,71) on P8 tensors; backward as ptmi_maxpool2x2_bwd (first maximum, optional
 * ReLU mask of the pooled activation). */
int ptmi_p8_maxpool2x2_fwd(const void* x, void* y, int n, int c, int h, int w, ptmi_stream_t s);
int ptmi_p8_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int n, int c, int h, int w, int relu_mask,
                           ptmi_stream_t s);
/* dz = dy * (y > 0) over `pixels16` 16-byte pixel vectors of P8 tensors */
int ptmi_p8_relu_bwd(const void* dy, const void* y, void* dz, int64_t pixels16, ptmi_stream_t s);
/* conv3x3 s1 p1 on P8 tensors: v_mfma_f32_32x32x16_bf16, fp32 accumulate, bf16 out; any channel counts (planes as above).
 * Weights packed by ptmi_p8_pack_weights (bf16, MFMA A-operand order [coTile][cin/16][tap][mt][64 lanes][8]; mode 0 forward,
 * mode 1 dgrad = flipped taps, transposed channels; ptmi_p8_packed_elems bf16 elements).
 * epilogue 0: + bias; 1: + bias, ReLU; 2: none (dgrad); 3: dgrad times (mask_ref > 0), mask_ref = the producing layer's stored
 * activation (P8, cout channels); 4: + bias, ReLU, 2x2 max-pool (floor) -- y is the P8 tensor of the POOLED map (n, cout, h/2,
 * w/2), pads included, and the full-resolution activation never reaches HBM (layers without a backward pass). */
int64_t ptmi_p8_packed_elems(int cin, int cout);
int ptmi_p8_pack_weights(const float* w, void* wp, int w_cout, int w_cin, int mode, ptmi_stream_t s);
int ptmi_p8_conv3x3(const void* x, const void* wp, const float* bias, const void* mask_ref, void* y, int n,
                    int cin, int cout, int h, int w, int epilogue, ptmi_stream_t s);
/* Weight + bias gradient of the conv above from its P8 input x (cin channels) and P8 output gradient dy (cout channels; pads
 * zero): dW (Cout, Cin, 3, 3) and db (Cout) in fp32 (same contract as ptmi_conv3x3_wgrad: split partials in a caller-allocated
 * workspace of ptmi_p8_wgrad_ws_floats floats, summed in a fixed order; accumulate != 0 adds to dW / db).  The contraction runs
 * over pixels: operands reach the MFMA through ds_read_b64_tr_b16; db is the product of dy with an all-ones operand. */
/* 1 if ptmi_p8_wgrad / ptmi_p8_gemm_nt serve the shape (their per-launch 32-bit buffer offsets: a P8 tensor of at most 4 GiB,
 * e.g. conv1_2 (64 channels, 1333x800) up to 31 images; fc1 up to ~85 k ROIs); the host side routes larger shapes elsewhere
 * (p8.wgrad: the fp32 direct kernel on the widened operands -- the same products, fp32 accumulation). */
int ptmi_p8_wgrad_fits(int n, int cin, int cout, int h, int w);
int ptmi_p8_gemm_nt_fits(int m, int n, int k);
int64_t ptmi_p8_wgrad_ws_floats(int n, int cin, int cout, int h, int w);
int ptmi_p8_wgrad(const void* x, const void* dy, float* dw, float* db, float* ws, int n, int cin, int cout,
                  int h, int w, int accumulate, ptmi_stream_t s);
/* The bf16-storage convolution / weight gradient with `waves` fills of the one-workgroup-per-CU slots (round 6; same contract as the
 * one-fill entry points = waves 1).  ptmi_p8_conv3x3's persistent workgroups walk items b, b + grid, ...; ptmi_p8_wgrad's own one
 * contiguous share each: a CU held by another kernel when the launch starts -- an RCCL collective overlapping backward under DDP,
 * pt/engine/trainer.py:92-95,384, the configuration BASELINE configs[4] names -- makes its workgroup run after all others (2x on the
 * launch, tools/exp/contention.py).  With more, shorter workgroups the dispatcher rebalances: what PTrainer selects when its gradient
 * exchange is active (convolution 16 waves: one more prologue per workgroup; weight gradient 4 waves: one more partial-sum store). */
int ptmi_p8_conv3x3_waves(const void* x, const void* wp, const float* bias, const void* mask_ref, void* y,
                          int n, int cin, int cout, int h, int w, int epilogue, int waves,
                          ptmi_stream_t s);
int64_t ptmi_p8_wgrad_ws_floats_waves(int n, int cin, int cout, int h, int w, int waves);
int ptmi_p8_wgrad_waves(const void* x, const void* dy, float* dw, float* db, float* ws, int n, int cin,
                        int cout, int h, int w, int accumulate, int waves, ptmi_stream_t s);
/* bf16-STORAGE GEMM for the box head's large Linear layer under SOLVER.AMP.ENABLED (FastRCNNConvFCHead fc1 25088 -> 1024 behind
 * pt/modeling/roi_heads/roi_heads.py:126-128; cuBLAS bf16 under autocast): C[m][n] (fp32, row pitch ldc) = A . B^T (+ bias[n]) (+ ReLU)
 * with BOTH operands bf16 in the "P8 matrix" layout t[ceil(k/8)][rows][8] (k in octets, one 16-byte vector per (row, octet)).
 * ptmi_p8m_pack builds an operand from an fp32 matrix: element (row, k) = src[row * ld + k] (k_major = 1) or src[k * ld + row]
 * (k_major = 0), rounded to bf16 (nearest even), zero beyond k; ptmi_p8m_elems bf16 elements.  Forward, dX and dW of a Linear layer
 * are all of this form (operands packed with their contraction index as k).  Shapes with few tiles and long k run split-K through a
 * caller-allocated workspace (ptmi_p8_gemm_nt_ws_floats floats; 0 = none), reduced in a fixed order.
 * relu: bit 0 = ReLU; bit 1 = store the result TRANSPOSED, C[n][m] with row pitch ldc >= m (no bias / ReLU / split-K): with the
 * operands swapped by the caller this writes the same matrix with 16-byte stores (fc1's dX, 1.6 GB of fp32, is store-bound). */
int64_t ptmi_p8m_elems(int rows, int k);
int ptmi_p8m_pack(const float* src, void* dst, int rows, int k, int64_t ld, int k_major, ptmi_stream_t s);
int64_t ptmi_p8_gemm_nt_ws_floats(int m, int n, int k);
int ptmi_p8_gemm_nt(const void* a, const void* b, float* c, const float* bias, float* ws, int m, int n, int k,
                    int ldc, int relu, ptmi_stream_t s);
/* dz = dy * (y > 0), elementwise (ReLU backward; F.relu_ at vgg.py:67). In-place allowed. */
int ptmi_relu_bwd(const float* dy, const float* y, float* dz, int64_t numel, ptmi_stream_t s);

/* ------------------------------------------------------------------ max pool (N2)
 * replaces ATen MaxPool2d(2,2) fwd/bwd at vgg.py:59,71 (floor mode).  bwd routes the gradient to
 * the first maximum of each window in (0,0),(0,1),(1,0),(1,1) order (ATen CPU semantics). */
int ptmi_maxpool2x2_fwd(const float* x, float* y, int nc, int h, int w, ptmi_stream_t s);
/* relu_mask != 0: x is a post-ReLU activation and the result is additionally multiplied by (x > 0), i.e. the
 * gradient w.r.t. the conv pre-activation (MaxPool2d backward + F.relu_ backward of vgg.py:66-71 in one pass). */
int ptmi_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int nc, int h, int w,
                        int relu_mask, ptmi_stream_t s);

/* ------------------------------------------------------------------ GEMM (N3 1x1 convs, N13 FC)
 * replaces cuBLAS Linear (D2 FastRCNNConvFCHead; fast_rcnn.py:164) and the 1x1 RPN convs.
 * C[b] (M,N) = op(A[b]) (M,K) * op(B[b]) (K,N)  [+ bias] [relu] [+ C if accumulate]
 * ta: 0 -> A stored (M,K) row-major with leading dim lda; 1 -> stored (K,M).
 * tb: 0 -> B stored (K,N) row-major with leading dim ldb; 1 -> stored (N,K).
 * bias_mode: 0 none, 1 per-row (M), 2 per-column (N).  batch strides in elements.
 * fp32 on v_mfma_f32_32x32x2_f32; k-ordered accumulation (deterministic).
 * ws / ws_floats: optional workspace of ptmi_gemm_ws_floats(m, n, k, batch) floats.  With it, shapes with few 128x128
 * tiles and a long K (the box head's weight gradients) or a tile count just above a multiple of the chip's workgroup
 * slots run split-K: K slices into ws, summed in a fixed order (still deterministic).  NULL / 0 = never split. */
int64_t ptmi_gemm_ws_floats(int m, int n, int k, int batch);
int ptmi_gemm_f32(const float* a, const float* b, float* c, const float* bias, int m, int n,
                  int k, int lda, int ldb, int ldc, int ta, int tb, int bias_mode, int relu,
                  int accumulate, int batch, int64_t stride_a, int64_t stride_b,
                  int64_t stride_c, float* ws, int64_t ws_floats, ptmi_stream_t s);
/* Same contract, bf16-input / fp32-accumulate (SOLVER.AMP.ENABLED, pt/engine/trainer.py:98; BASELINE configs[4]): A and B
 * are fp32 in memory, every operand element is rounded to bf16 (round-to-nearest-even) on its way into
 * v_mfma_f32_32x32x16_bf16; products are exact in fp32 and accumulated in fp32; bias / ReLU / C stay fp32. */
int ptmi_gemm_bf16(const float* a, const float* b, float* c, const float* bias, int m, int n,
                   int k, int lda, int ldb, int ldc, int ta, int tb, int bias_mode, int relu,
                   int accumulate, int batch, int64_t stride_a, int64_t stride_b,
                   int64_t stride_c, float* ws, int64_t ws_floats, ptmi_stream_t s);
/* column sums: out[j] (+)= sum_i a[i][j]  (bias gradients), a is (rows, cols) row-major. */
int ptmi_colsum(const float* a, float* out, int rows, int cols, int accumulate, ptmi_stream_t s);
/* the same for tall matrices (rows >> cols: the 1x1 RPN heads' bias gradients, rpn.py:96 backward): row ranges are summed by
 * separate workgroups into ws (ptmi_colsum_ws_floats(rows, cols) floats; 0 = not worth splitting), then reduced in a fixed
 * order.  ws == NULL falls back to ptmi_colsum. */
int64_t ptmi_colsum_ws_floats(int rows, int cols);
int ptmi_colsum_ws(const float* a, float* out, float* ws, int rows, int cols, int accumulate, ptmi_stream_t s);
/* row sums over the inner dim: out[i] (+)= sum_j a[i][j] for each of `batch` slabs summed. */
int ptmi_rowsum_batched(const float* a, float* out, int batch, int rows, int cols, int accumulate,
                        ptmi_stream_t s);

/* ------------------------------------------------------------------ ROIAlign (N12)
 * replaces torchvision roi_align(aligned=True, sampling_ratio=0) built at
 * pt/modeling/roi_heads/roi_heads.py:68-73 and called at :126.  rois (R,5)=[img,x1,y1,x2,y2]. */
int ptmi_roi_align_fwd(const float* feat, const float* rois, float* out, int n, int c, int h,
                       int w, int r, int pooled, float scale, ptmi_stream_t s);
/* Same result to fp32 rounding (a bin's cells are summed with separable row / column weights, not sample by sample) for rois
 * GROUPED BY IMAGE (img_offsets int32 (n+1) device array): every bin row's / column's first cell and weights are tabulated
 * once into `ws` (ptmi_roi_align_ws_bytes(r, h, w) bytes of device scratch), channel planes are staged in LDS once per image.
 * ws == NULL falls back to ptmi_roi_align_fwd. */
int64_t ptmi_roi_align_ws_bytes(int r, int h, int w);
int ptmi_roi_align_fwd_grouped(const float* feat, const float* rois, const int32_t* img_offsets,
                               float* out, void* ws, int n, int c, int h, int w, int r, int pooled,
                               float scale, ptmi_stream_t s);
/* The grouped forward writing the box head's first Linear layer's bf16 operands directly (SOLVER.AMP.ENABLED; replaces the
 * ROIPooler -> flatten -> autocast Linear hand-over of roi_heads.py:126-128): xk = "P8 matrix" [c*49/8][r][8] of the flattened
 * ROI features (what ptmi_p8m_pack(k_major) of the fp32 result would hold, round to nearest even), xt (may be NULL) =
 * [ceil(r/8)][c*49][8], the weight gradient's operand (contraction over ROIs; rows beyond r are zeros).  pooled = 7 and maps
 * whose planes fit the LDS budget only: ptmi_roi_align_fwd_p8m_fits. */
int ptmi_roi_align_fwd_p8m_fits(int c, int h, int w, int pooled);
int ptmi_roi_align_fwd_p8m(const float* feat, const float* rois, const int32_t* img_offsets, void* xk, void* xt, void* ws,
                           int n, int c, int h, int w, int r, int pooled, float scale, ptmi_stream_t s);
/* dfeat must be zeroed by the caller (atomic scatter-add). */
int ptmi_roi_align_bwd(const float* dout, const float* rois, float* dfeat, int n, int c, int h,
                       int w, int r, int pooled, float scale, ptmi_stream_t s);
/* Same result (to summation order) for rois GROUPED BY IMAGE (rows img_offsets[i]..img_offsets[i+1] belong to image i; int32
 * device array of n+1): each workgroup accumulates up to four channel planes of one image in LDS, every plane row owned
 * by one wave, and writes them once -- no global atomics, a fixed summation order (bitwise reproducible), dfeat need not be
 * zeroed (every element is written).  `ws`: ptmi_roi_align_bwd_ws_bytes(r, h, w) bytes of device scratch for the per-ROI
 * weight tables (NULL, or planes that do not fit the LDS: the atomic kernel on a zeroed dfeat). */
int64_t ptmi_roi_align_bwd_ws_bytes(int r, int h, int w);
int ptmi_roi_align_bwd_grouped(const float* dout, const float* rois, const int32_t* img_offsets,
                               float* dfeat, void* ws, int n, int c, int h, int w, int r, int pooled,
                               float scale, ptmi_stream_t s);

/* ------------------------------------------------------------------ boxes (N4, N5, N9)
 * anchors: D2 DefaultAnchorGenerator / pt/modeling/anchor_generator.py:108-122: out (h*w*A,4),
 * anchor n=(y*w+x)*A+a = [x*stride,y*stride,x*stride,y*stride] + cell[a] (offset 0 folded in). */
int ptmi_grid_anchors(const float* cell, float* out, int h, int w, int a, float stride,
                      float offset, ptmi_stream_t s);
/* Box2BoxTransform.apply_deltas, pt/modeling/box_regression.py:101-139.
 * deltas (rows, 4*k) with row stride `dstride` floats, boxes (nb,4) with row i using box
 * i % nb (nb = rows, or the anchor count when the same anchors repeat per image). */
int ptmi_apply_deltas(const float* deltas, const float* boxes, float* out, int64_t rows, int k,
                      int dstride, int64_t nb, float wx, float wy, float ww, float wh,
                      float scale_clamp, ptmi_stream_t s);
/* Box2BoxTransform.get_deltas, box_regression.py:66-99 (with the +1e-9 inside the logs). */
int ptmi_get_deltas(const float* src, const float* tgt, float* out, int64_t rows, float wx,
                    float wy, float ww, float wh, ptmi_stream_t s);
/* pairwise_iou + Matcher fused (rpn.py:414-415; roi_heads.py:207-214).
 * gt (m,4), boxes (nb,4); thresholds_host / labels_host are HOST arrays.  Outputs per box: matched gt index (int64, argmax over gt, lowest
 * index on ties), label (int8).  thresholds: n_thr in {1,2}; labels[n_thr+1].
 * allow_low_quality: every box whose IoU with gt i equals gt i's best IoU gets label 1.
 * ws: m floats of workspace (per-gt best IoU).  m == 0 -> idx 0, label labels[0]. */
int ptmi_iou_match(const float* gt, const float* boxes, int m, int64_t nb, const float* thresholds_host,
                   const int* labels_host, int n_thr, int allow_low_quality, int64_t* matched_idx,
                   int8_t* matched_label, float* matched_iou, float* ws, ptmi_stream_t s);

/* The same for a whole batch in one launch per pass.  gt_all: the images' gt boxes concatenated, gt_off (nimg+1) int32
 * DEVICE row offsets; boxes: either concatenated per image with box_off (nimg+1) int32 DEVICE offsets (outputs in the
 * same row order), or box_off == NULL: the same max_boxes boxes for every image (RPN anchors; outputs (nimg, max_boxes)).
 * max_boxes = largest per-image box count; matched_idx is the gt index WITHIN the image; ws: total_gt floats. */
int ptmi_iou_match_batched(const float* gt_all, const int32_t* gt_off, const float* boxes, const int32_t* box_off,
                           int nimg, int64_t max_boxes, int64_t total_gt, const float* thresholds_host,
                           const int* labels_host, int n_thr, int allow_low_quality, int64_t* matched_idx,
                           int8_t* matched_label, float* matched_iou, float* ws, ptmi_stream_t s);
/* D2 subsample_labels (rpn.py:433 via RPN._subsample_labels; D2 StandardROIHeads._sample_proposals) for
 * a batch of label vectors without host syncs: cls_all int64 (concatenated; -1 ignore, bg_label background, anything
 * else foreground), keys_all i.i.d. uniform keys, offsets (nimg+1) int32 DEVICE.  Per image: n_fg = min(#fg,
 * num_pos_max), n_bg = min(#bg, num_samples - n_fg); out_fg (nimg, num_pos_max) / out_bg (nimg, num_samples) get the
 * local indices of the n_fg / n_bg candidates with the smallest keys in ascending key order (== the prefix of the
 * permutation argsort(keys[candidates])); counts (nimg, 2) int32 = (n_fg, n_bg).  max_count <= 12288. */
int ptmi_sample_by_keys(const int64_t* cls_all, const float* keys_all, const int32_t* offsets, int nimg,
                        int64_t max_count, int num_samples, int num_pos_max, int bg_label, int64_t* out_fg,
                        int64_t* out_bg, int32_t* counts, ptmi_stream_t s);

/* RPN._subsample_labels for a whole batch (pt/modeling/proposal_generator/rpn.py:433 -> D2 subsample_labels; SURVEY A.4,
 * N6): labels (nimg, r) int8 in {-1 ignore, bg_label, anything else = positive}, keys (nimg, r) float32 >= 0 (i.i.d.
 * uniform in production).  Per image n_f = min(#positives, num_pos_max), n_b = min(#negatives, num_samples - n_f); out
 * (nimg, r) int8 = 1 for the n_f positives and 0 for the n_b negatives with the smallest keys (ties: lowest index), -1
 * everywhere else.  Radix select on the key bits, one workgroup per image, no host sync. */
int ptmi_rpn_subsample_relabel(const int8_t* labels, const float* keys, int8_t* out, int nimg, int64_t r, int num_samples,
                               int num_pos_max, int bg_label, ptmi_stream_t s);

/* ------------------------------------------------------------------ sort + proposals (N10)
 * replaces torch.sort(descending) at pt/modeling/proposal_generator/proposal_utils.py:87 and the
 * sort inside torchvision nms.  Stable: ties keep ascending original index.
 * seg_offsets (nseg+1) int32 device array.  ws sized by ptmi_segsort_ws_bytes. */
int64_t ptmi_segsort_ws_bytes(int64_t total, int nseg);
int ptmi_segsort_desc(const float* keys_in, float* keys_out, int32_t* idx_out, int64_t total,
                      int nseg, const int32_t* seg_offsets, void* ws, int64_t ws_bytes,
                      ptmi_stream_t s);
/* The same order from ONE workgroup per segment sorting in LDS (bitonic network on (key, index) pairs): whole segments of up to
 * 16 384 keys; of longer segments (< 65 536 keys) only the first `topk` <= 16 384 entries of the sorted order (radix select +
 * ordered compaction first) -- entries past topk come back as (-inf, 0).  max_len = the longest segment (the caller knows it;
 * no device read).  ptmi_segsort_topk_fits tells whether a (max_len, topk) pair is served; otherwise use ptmi_segsort_desc. */
int ptmi_segsort_topk_fits(int64_t max_len, int64_t topk);
int ptmi_segsort_topk_desc(const float* keys_in, float* keys_out, int32_t* idx_out, int nseg, const int32_t* seg_offsets,
                           int64_t max_len, int64_t topk, ptmi_stream_t s);
/* proposal_utils.py:92-138 for one batch: for image i and rank j < k:
 *   box = clip(decoded[i][sorted_idx[i][j]], image_size[i]); kept = finite & w>min & h>min;
 *   score = sorted_logit[i][j] * (1 - mean4(sigmoid(sigma_logits[i][j])))   <- row j, NOT idx (:94)
 * Outputs dense (n,k,*): boxes, keys = score for kept entries / -inf for dropped ones (a stable descending sort of the
 * keys lists the kept entries in the order batched_nms visits them), the number of kept entries and a nonfinite flag
 * per image -- no host synchronisation is needed before NMS. */
int ptmi_rpn_prepare(const float* decoded, const float* sorted_logits, const int32_t* sorted_idx,
                     const float* sigma_logits, const float* image_sizes_hw, float* boxes_out,
                     float* keys_out, int32_t* counts_out, int32_t* nonfinite_out, int n,
                     int64_t r, int k, float min_size, ptmi_stream_t s);

/* ------------------------------------------------------------------ NMS (N11)
 * replaces torchvision nms (detectron2 batched_nms) at proposal_utils.py:140, fast_rcnn.py:104.
 * Batched over `nimg` images: boxes (sum counts,4) ALREADY sorted by descending score per image,
 * seg_offsets (nimg+1) int32; seg_counts (nimg) int32 or NULL: only the first seg_counts[i] boxes of segment i exist
 * (fixed-capacity segments whose fill level is known on the device only).  keep_out (nimg, max_keep) int32 =
 * positions within the image's sorted list, keep_count (nimg) int32.  Suppress iff IoU > thr (strict), IoU evaluated
 * as inter/(area_i+area_j-inter) in fp32 without FMA contraction => bit-exact vs the CPU oracle.
 * ws: ptmi_nms_ws_bytes(max_count, nimg), max_count = largest segment capacity. */
int64_t ptmi_nms_ws_bytes(int64_t max_count, int nimg);
int ptmi_nms_batched(const float* boxes, const int32_t* seg_offsets, const int32_t* seg_counts, int nimg,
                     int64_t max_count, float thr, int max_keep, int32_t* keep_out, int32_t* keep_count,
                     void* ws, ptmi_stream_t s);

/* ------------------------------------------------------------------ teacher ROI inference (N12)
 * fast_rcnn.py:34-101 (fast_rcnn_inference_single_image up to the NMS call) for all R ROIs of a batch: decode the K
 * mean quadruples of deltas (R, 8K) on proposal_boxes (R,4) with weights (wx,wy,ww,wh) (:63 drops the sigma
 * quadruples), finite filter over a ROI's K boxes and K+1 probabilities (:67), clip to the image roi_img[r] (:78),
 * probs[r][j] > score_thresh (:85), score *= 1 - mean4(sigmoid(sigma logits)) (:101).
 * Outputs dense over (roi, class): boxes_out (R,K,4) clipped; keys_out (R,K) = rescored score of a candidate, -1
 * otherwise (scores are > 0: a stable descending sort of an image's keys lists its candidates in batched_nms order);
 * img_max_out (nimg) = largest candidate coordinate, img_count_out (nimg) = number of candidates;
 * roi_valid_out (R) u8 = the finite filter's verdict, img_invalid_out (nimg) = ROIs it dropped (the reference indexes
 * `scores_logists` and the returned ROI indices by position in the FILTERED list, :96,:126). */
int ptmi_roi_infer_prepare(const float* deltas, const float* proposal_boxes, const float* probs,
                           const int32_t* roi_img, const float* image_sizes_hw, float* boxes_out,
                           float* keys_out, uint8_t* roi_valid_out, float* img_max_out,
                           int32_t* img_count_out, int32_t* img_invalid_out, int64_t r, int k, int nimg,
                           float wx, float wy, float ww, float wh, float scale_clamp, float score_thresh,
                           ptmi_stream_t s);
/* torchvision batched_nms class offset (fast_rcnn.py:104): out[beg_i + p] = boxes[beg_i + order[beg_i + p]] +
 * (order[..] % k) * (img_max[i] + 1) in fp32, for the dense per-image segments seg_offsets (nimg+1) of (roi, class)
 * entries; max_count = largest segment. */
int ptmi_roi_infer_nms_boxes(const float* boxes, const int32_t* order, const int32_t* seg_offsets,
                             const float* img_max, int nimg, int max_count, int k, float* out,
                             ptmi_stream_t s);

/* ------------------------------------------------------------------ losses (N7, N8, N14)
 * Every loss kernel writes the scalar loss (already normalised) to loss_out[0] and the gradient
 * w.r.t. its differentiable inputs (already multiplied by 1/normaliser) in the same launch;
 * reductions are block-tree + fixed-order final sum (deterministic).  ws: 4096 floats. */
/* rpn.py:242-246: sum BCE-with-logits over label>=0, / norm.  labels int8 (-1,0,1). */
int ptmi_bce_logits_sum(const float* logits, const int8_t* labels, int64_t n, float inv_norm,
                        float* loss_out, float* dlogits, float* ws, ptmi_stream_t s);
/* box_regression.py:165-176 / fast_rcnn.py:286-296: sum over rows of -log(pdf(mu;t,var)+1e-9),
 * var = sigmoid(slog); d (rows,8) = [mu(4), slog(4)] gathered rows; t (rows,4) targets.
 * Gradients: dd (rows,8); dt (rows,4) optional (NULL to skip; needed for d(loss)/d(anchors)). */
int ptmi_gaussian_nll_sum(const float* d, const float* t, int64_t rows, float inv_norm,
                          float* loss_out, float* dd, float* dt, float* ws, ptmi_stream_t s);
/* fast_rcnn.py:408 + D2 FastRCNNOutputLayers.losses: mean cross entropy; dlogits (r,c). */
int ptmi_softmax_ce_mean(const float* logits, const int64_t* target, int64_t r, int c,
                         float* loss_out, float* dlogits, float* ws, ptmi_stream_t s);
/* softmax over the last dim (fast_rcnn.py:408 predict_probs). */
int ptmi_softmax_rows(const float* logits, float* probs, int64_t r, int c, ptmi_stream_t s);
/* fast_rcnn.py:179-213 cls_loss_unsupervised: teacher logits T (r,c), student logits S (r,c):
 * sum_r w_r * softmax(T/tau)·(-log_softmax(S)) * inv_norm, w_r=(1-H(softmax T)/log c)^lambda
 * (w_r = 1 if !efl).  dS (r,c). */
int ptmi_soft_ce_efl(const float* t, const float* s_logits, int64_t r, int c, float tau,
                     float lambda, int efl, float inv_norm, float* loss_out, float* ds, float* ws,
                     ptmi_stream_t s);
/* rpn.py:285-304 RPN soft objectness loss (incl. the sigmoid(1-x) quirk at :299).
 * T (k,c) teacher logits of the matched pseudo box per positive anchor, x (k) student objectness.
 * Also emits fg[k] (uint8) = argmax T != c-1 (rpn.py:292-293). */
int ptmi_rpn_soft_obj_loss(const float* t, const float* x, int64_t k, int c, float tau, float lambda,
                           int efl, float inv_norm, float* loss_out, float* dx, uint8_t* fg,
                           float* ws, ptmi_stream_t s);
/* rpn.py:321-355 / fast_rcnn.py:215-263: KL(N(mu_p,var_p*tau) || N(mu_q,var_q)) with the
 * per-coordinate entropy-focal weight; rows selected by fg (NULL = all rows).
 * q (rows,8)=[mu_q, slog_q]; mu_p (rows,4); slog_p (rows,4) teacher sigma logits.
 * reduction: 0 = sum*inv_norm (RPN), 1 = mean over selected rows*4 (ROI, inv_norm ignored).
 * Gradients dq (rows,8), dmu_p (rows,4; NULL when mu_p is detached as in fast_rcnn.py:235). */
int ptmi_kl_efl_loss(const float* q, const float* mu_p, const float* slog_p, const uint8_t* fg,
                     int64_t rows, float tau, float lambda, int efl, int reduction, float inv_norm,
                     float* loss_out, float* dq, float* dmu_p, float* ws, ptmi_stream_t s);
/* backward of get_deltas w.r.t. the SOURCE boxes (anchors), accumulated into danchors (na,4):
 * rows index anchors via anchor_index[i] (int64).  Used only for the differentiable anchors
 * (rpn.py:311 with danchor=True, anchor_generator.py:147). */
int ptmi_get_deltas_bwd_src(const float* src, const float* tgt, const float* ddeltas,
                            const int64_t* src_index, int64_t rows, float wx, float wy, float ww,
                            float wh, float* dsrc, ptmi_stream_t s);

/* ------------------------------------------------------------------ optimiser / EMA (N15-N17)
 * flat fp32 buffers.  EMA: trainer.py:431-449  t = s*(1-k) + t*k  (that evaluation order). */
int ptmi_ema_update(const float* student, float* teacher, int64_t n, float keep_rate,
                    float one_minus_keep_rate, ptmi_stream_t s);
/* trainer.py:592-603 clip_gradient: sumsq_out[0] = sum g^2 (two-stage, deterministic). */
int ptmi_sumsq(const float* g, int64_t n, float* sumsq_out, float* ws, ptmi_stream_t s);
/* fused clip + SGD(momentum, weight decay) step (trainer.py:385-386; torch.optim.SGD semantics):
 * s = clip/max(sqrt(sumsq[0]),clip); g' = g*s + wd*p; buf = first ? g' : mu*buf+g'; p -= lr*buf. */
int ptmi_clip_sgd_step(float* p, const float* g, float* buf, int64_t n, const float* sumsq,
                       float clip_norm, float lr, float momentum, float weight_decay, int first,
                       ptmi_stream_t s);
int ptmi_scale_by_clip(float* g, int64_t n, const float* sumsq, float clip_norm, ptmi_stream_t s);

/* ------------------------------------------------------------------ image prep (N18)
 * rcnn.py:40 -> D2 preprocess_image: out[i] (3,hmax,wmax) = (u8 - mean)/std, zero padded.
 * One launch per image (images have individual sizes). */
int ptmi_preprocess_image(const uint8_t* img, float* out, int h, int w, int hmax, int wmax,
                          float m0, float m1, float m2, float s0, float s1, float s2,
                          ptmi_stream_t s);
/* trainer.py:557-590 resize: bilinear (align_corners=False) shrink to (dh,dw), truncated to u8,
 * pasted at (y1,x1) on a canvas filled with int(pixel_mean).  Byte-exact with ATen's CPU kernel as the reference
 * runs it (generic NCHW path, FMA-contracted; see csrc/misc.hip). */
int ptmi_shrink_paste(const uint8_t* img, uint8_t* out, int h, int w, int dh, int dw, int y1,
                      int x1, int m0, int m1, int m2, ptmi_stream_t s);

/* Whole-batch variants (one launch instead of one per image; the step prepares 64 + 32 images).  `desc` is a DEVICE
 * array of 8 int64 words per image: [src u8 pointer, dst u8 pointer, h, w, dh, dw, y1, x1] (dst and dh..x1 are ignored
 * by preprocess; every pointed-to buffer is owned by the caller and must stay alive until the launch has run).
 * preprocess: out is (n, 3, hmax, wmax) fp32.  shrink_paste: max_elems = max_i 3*h_i*w_i (sizes the grid). */
int ptmi_preprocess_batched(const int64_t* desc, float* out, int n, int hmax, int wmax, float m0, float m1,
                            float m2, float s0, float s1, float s2, ptmi_stream_t s);
int ptmi_shrink_paste_batched(const int64_t* desc, int n, int64_t max_elems, int m0, int m1, int m2,
                              ptmi_stream_t s);

/* ------------------------------------------------------------------ strong augmentation on the device
 * replaces the per-image PIL / torchvision work of the two-crop mapper: pt/data/detection_utils.py:38-60
 * (ColorJitter(0.4,0.4,0.4,0.1), RandomGrayscale, GaussianBlur, Solarize), pt/data/transforms/augmentation_impl.py:21-53,
 * applied at pt/data/dataset_mapper.py:151-159.  Byte-exact with Pillow (oracle/csrc/ref_aug.c).  Planar uint8 (3,H,W)
 * images, batched through a DEVICE table of 8 int64 words per image: [src, dst, h, w, p4, p5, p6, p7].
 *   gray_sum : sums_out[i] = sum of convert("L") grey levels of image i (ImageStat, for ImageEnhance.Contrast)
 *   color    : p4 = op (0 copy, 1 brightness, 2 contrast, 3 saturation, 4 hue, 5 grayscale, 6 solarize); p5 = the float
 *              enhancement factor (its bit pattern); p6 = hue shift in 1/256 turns / solarize threshold; contrast reads
 *              gray_sums[i].  src == dst is allowed.
 *   box_blur : one pass of Pillow's extended box blur (GaussianBlur = 3 horizontal + 3 vertical passes):
 *              p4 = 0 along x / 1 along y, p5 = integer radius, p6 = ww, p7 = fw (24-bit fixed-point weights).  src != dst.
 *   hflip    : p4 = 1 flips the image left-right (D2 RandomFlip), 0 copies.  src != dst. */
int ptmi_aug_gray_sum_batched(const int64_t* desc, int n, int64_t max_hw, uint64_t* sums_out, ptmi_stream_t s);
int ptmi_aug_color_batched(const int64_t* desc, int n, int64_t max_hw, const uint64_t* gray_sums, ptmi_stream_t s);
int ptmi_aug_box_blur_batched(const int64_t* desc, int n, int64_t max_elems, ptmi_stream_t s);
int ptmi_aug_hflip_batched(const int64_t* desc, int n, int64_t max_elems, ptmi_stream_t s);
/*   resize   : one pass of Pillow's Image.resize(..., BILINEAR) (D2 ResizeShortestEdge / ResizeTransform of the weak
 *              augmentation, dataset_mapper.py:107-109; antialiasing triangle filter, 22-bit fixed-point coefficients):
 *              p4 = output size along the pass, p5 = 0: (3,h,w) -> (3,h,p4), 1: (3,h,w) -> (3,p4,w).  A resize is the x pass
 *              followed by the y pass (a pass whose size does not change is skipped).  Down-scaling factors up to 15. */
int ptmi_aug_resize_pass_batched(const int64_t* desc, int n, int64_t max_out_elems, ptmi_stream_t s);

/* ------------------------------------------------------------------ diagnostics
 * CU-contention probe (round 6; no counterpart in the reference): n_cus workgroups that each hold one CU (64 KB of LDS: no
 * MFMA workgroup of this library fits beside one) for `microseconds` on stream s -- stands in for a collective's kernels
 * (RCCL all-reduce overlapping backward under DDP, pt/engine/trainer.py:92-95,384) when only one GPU is at hand.
 * tools/exp/contention.py, tests/test_wino4_gpu.py.  scratch: one int32 in device memory (never written). */
int ptmi_hold_cus(int32_t* scratch, int n_cus, int microseconds, ptmi_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* PTMI355_H */
