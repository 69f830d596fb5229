/*
 * mridc_amd.h -- C ABI of libmridc_amd.so: the MI355X (gfx950) implementation of the mridc
 * unrolled-reconstruction hot path.
 *
 * The reference (wdika/mridc) has no FFI: its boundary is Python call signatures over torch tensors.
 * Each entry point below replaces the PyTorch op sequence of the cited reference function and is
 * what a ctypes / cffi binding on the reference side would call.  Reference
 * paths are relative to the reference root.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 data unless stated; complex tensors use the
 *     reference's real view [..., 2] (re, im interleaved);
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library allocates device memory
 *     only for small read-only twiddle tables, once per FFT length, outside any launch sequence
 *     (mrx_fft_prepare does it eagerly, e.g. before hipGraph capture);
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered and asynchronous;
 *   - return 0 on success, a negative MRX_E* code otherwise; mrx_last_error() returns the message
 *     (thread-local).  Nothing throws across the ABI.
 */
#ifndef MRIDC_AMD_H
#define MRIDC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRX_OK 0
#define MRX_EINVAL (-1)   /* bad argument (shape, enum, null pointer) */
#define MRX_EUNSUP (-2)   /* valid request the library does not implement (size limits) */
#define MRX_EHIP (-3)     /* a HIP runtime call failed */
#define MRX_EBOUND (-4)   /* CHECK builds only (-DMRX_CHECK_BOUNDS): an operand bound handed to a two-term fp16 kernel does not bound its tensor */

/* normalization, as the reference's `fft_normalization` strings (fft.py:13-18): */
#define MRX_NORM_BACKWARD 0
#define MRX_NORM_ORTHO 1
#define MRX_NORM_FORWARD 2
#define MRX_NORM_NONE 3 /* "none" -> torch default = backward (fft.py:80) */

/* mask element kinds */
#define MRX_MASK_U8 0  /* torch.bool or torch.uint8, one byte per element */
#define MRX_MASK_F32 1

/* activations */
#define MRX_ACT_NONE 0
#define MRX_ACT_RELU 1
#define MRX_ACT_LEAKY 2 /* slope passed separately */

/* padding modes for mrx_conv2d */
#define MRX_PAD_ZERO 0
#define MRX_PAD_REPLICATE 1

int mrx_version(void);
const char* mrx_last_error(void);
/* 0 when `stream` is not being captured into a hipGraph, otherwise the non-zero id of the capture (hipStreamGetCaptureInfo).  The host side
 * keys its per-slice operand caches on it: an operand prepared eagerly must not be baked into a graph (a replay on refilled inputs would
 * read stale operands) and one prepared inside a capture must not be used outside it. */
int64_t mrx_stream_capture_id(void* stream);
/* The library's one arithmetic switch, environment MRIDC_AMD_ARITH (read at every call): 0 = "f16x2" (default: two-term fp16 operands, three term
 * products per multiply, where a kernel has that form), 1 = "bf16x3" (three-term bf16 operands, six exact term products), 2 = "fp32" (the
 * fp32-input MFMA kernels).  All three give fp32 results; tests cross-check them, bench.py prices them (exact_fp32_route). */
int mrx_arith(void);
/* 1 in a CHECK build of the library (python -m mridc_amd._build --check-bounds -> mridc_amd/lib_chk/, selected with MRIDC_AMD_LIB), else 0.  A check
 * build verifies, at every entry point that takes the bound of an operand (mrx_conv3x3_sb_chain, mrx_rim_layer2_f16[_cb8], mrx_unet_conv3x3_h,
 * mrx_conv3x3_h), that max |x| <= bound <= 2^16 max |x| over the tensor the call is about to read: one reduction launch + a stream synchronisation per
 * call (skipped while the stream is being captured), MRX_EBOUND and a message naming the entry point otherwise.  The product build has none of it. */
int mrx_checks_enabled(void);

/* Create (and cache) the twiddle tables for lengths h and w.  Optional; every FFT entry point does
 * it lazily.  Call it before capturing a hipGraph. */
int mrx_fft_prepare(int h, int w);
/* Largest supported 1-D FFT length (LDS-resident transform). */
int mrx_fft_max_len(void);

/* A1/A2  fft2 / ifft2 (common/parts/fft.py:13-88, :91-166) over the last two dims of [batch,H,W,2].
 * centered: ifftshift before / fftshift after folded into index math (fft.py:74-75,83-84, :279,:320).
 * in == out is allowed. */
int mrx_fft2(const float* in, float* out, int64_t batch, int H, int W, int inverse, int norm, int centered,
             void* stream);

/* A3  roll / fftshift / ifftshift (fft.py:169-322): out[i] = in[(i - shift) mod n] along every listed dim.
 * Bit-exact copy of `elem_bytes`-sized elements (1,2,4,8,16).  ndim <= 8.  shifts[d] = 0 for untouched dims. */
int mrx_roll(const void* in, void* out, int elem_bytes, int ndim, const int64_t* shape, const int64_t* shifts,
             void* stream);

/* A4  complex_mul with broadcasting (common/parts/utils.py:96-118).  Shapes/strides are in COMPLEX
 * elements over ndim <= 6 leading dims (the trailing [2] is implicit); stride 0 broadcasts. */
int mrx_complex_mul(const float* x, const float* y, float* out, int ndim, const int64_t* shape,
                    const int64_t* xstride, const int64_t* ystride, int conj_y, void* stream);
/* A5  complex_conj / complex_abs / complex_abs_sq (utils.py:121-175) over n complex elements. */
int mrx_complex_conj(const float* x, float* out, int64_t n, void* stream);
int mrx_complex_abs(const float* x, float* out, int64_t n, int squared, void* stream);

/* A6  coil combination (utils.py:194-272) over a middle dim: input viewed as [outer, R, inner].
 *   mrx_rss:          out[o,i] = sqrt(sum_r x[o,r,i]^2), x is the REAL view (inner counts floats)   (utils.py:209)
 *   mrx_rss_complex:  out[o,i] = sqrt(sum_r |x[o,r,i]|^2), inner counts complex elements            (utils.py:227)
 *   mrx_sense:        out[o,i] = sum_r x[o,r,i] * conj(s[o,r,i]), complex                           (utils.py:248) */
int mrx_rss(const float* x, float* out, int64_t outer, int64_t R, int64_t inner, void* stream);
int mrx_rss_complex(const float* x, float* out, int64_t outer, int64_t R, int64_t inner, void* stream);
int mrx_sense(const float* x, const float* s, float* out, int64_t outer, int64_t R, int64_t inner, void* stream);

/* A8  apply_mask's arithmetic (utils.py:341): out = data * mask + 0.0, mask fp32 broadcast over
 * [B,C,H,W] with element strides mstride[4] (0 = broadcast). */
int mrx_apply_mask(const float* data, const float* mask, float* out, int B, int C, int H, int W,
                   const int64_t* mstride, void* stream);

/* K1  sens_expand: fft2(complex_mul(x[B,1,H,W], S[B,C,H,W]))  (vn_block.py:51-69, rim_utils.py:44-51,
 * rim_block.py:260-265).  x is [B,H,W,2]; out is [B,C,H,W,2]. */
int mrx_sens_expand(const float* x, const float* S, float* out, int B, int C, int H, int W, int norm,
                    int centered, void* stream);
/* K3  sens_reduce: sum_c ifft2(k)[c] * conj(S[c])  (vn_block.py:71-87, rim_block.py:199-210, utils.py:230-248
 * after ifft2).  k,S [B,C,H,W,2]; out [B,H,W,2]; work: caller scratch of B*C*H*W*2 floats (may alias k:
 * then k is destroyed). */
int mrx_sens_reduce(const float* k, const float* S, float* out, float* work, int B, int C, int H, int W,
                    int norm, int centered, void* stream);
/* Row-transform-only forms for row-invariant (1-D column) masks, on k-space kept in hybrid space kh = IFFT_H(k) (mrx_fft_cols once
 * per slice): masked data consistency commutes with the H transform, so a cascade (vn_block.py:89-119, ccnn_block.py:101-139) runs
 * as mrx_sens_reduce_rows -> regulariser -> mrx_sens_expand_rows -> mrx_dc_combine on kh, and the final SENSE combination of
 * ifft2(k) is mrx_sens_reduce_rows(kh). */
int mrx_sens_expand_rows(const float* x, const float* S, float* out, int B, int C, int H, int W, int norm, int centered,
                         void* stream);
int mrx_sens_reduce_rows(const float* kh, const float* S, float* out, int B, int C, int H, int W, int norm, int centered,
                         void* stream);
/* mrx_sens_expand_rows with the data-consistency combination as its epilogue (the same operations as the two separate launches):
 * out = pred - where(mask, pred - ref, 0) * dc_weight[0] - FFT_W(x * S); out may alias pred. */
int mrx_sens_expand_rows_dc(const float* x, const float* S, const float* pred, const float* ref, const void* mask,
                            int mask_kind, const int64_t* mstride, const float* dc_weight, float* out, int B, int C, int H,
                            int W, int norm, int centered, void* stream);

/* A9  log_likelihood_gradient (models/rim/rim_utils.py:11-67), three launches:
 *     rows: eta*S -> FFT_W ; cols: FFT_H -> mask*(k - y) -> IFFT_H ; rows: IFFT_W -> sum_c conj(S) -> /sigma^2.
 * eta [B,H,W,2]; y,S [B,C,H,W,2]; mask element (b,c,h,w) at mask[b*ms[0]+c*ms[1]+h*ms[2]+w*ms[3]];
 * out4 [B,4,H,W] = (eta_re, eta_im, grad_re, grad_im); work: B*C*H*W*2 floats. */
int mrx_llg(const float* eta, const float* y, const float* S, const void* mask, int mask_kind,
            const int64_t* mstride, float* out4, float* work, int B, int C, int H, int W, float inv_sigma2,
            int norm, int centered, void* stream);

/* A9, row-invariant masks (1-D column masks, mstride[2] == 0): IFFT_H(m(w)(FFT_H(.) - y)) = m(w)(. - IFFT_H y), so with
 * yt = IFFT_H(y) (mrx_fft_cols, inverse, once per cascade) every step is ONE launch of row transforms that moves exactly
 * the algorithmic (25+16C)HW bytes.  Same result as mrx_llg up to fp32 rounding.
 *   mrx_fft_cols   1-D transform along H of [nimg,H,W,2] (the column pass of mrx_fft2); in == out allowed
 *   mrx_llg_hinv   eta [B,H,W,2], yt/S [B,C,H,W,2] -> out4 [B,4,H,W]; `work` (mrx_llg_hinv_work_floats floats, may be NULL)
 *                  lets small grids split the coil sum over several workgroups per row (partial sums + combine) */
int64_t mrx_llg_hinv_work_floats(int B, int C, int H, int W);
int mrx_fft_cols(const float* in, float* out, int64_t nimg, int H, int W, int inverse, int norm, int centered,
                 void* stream);
int mrx_llg_hinv(const float* eta, const float* yt, const float* S, const void* mask, int mask_kind,
                 const int64_t* mstride, float* out4, float* work, int B, int C, int H, int W, float inv_sigma2, int norm,
                 int centered, void* stream);
/* mrx_llg_hinv with the sum over the coil-chunk partials left to the consumer: *nparts (host) = number of partial planes
 * work[k][B][H][W][2] still to be added and scaled by inv_sigma2 (then out4 is NOT written), or 0 (out4 complete).
 * mrx_rim_layer_indrnn_packed_llg is that consumer: the fused first RIM layer reading (eta, partials) in its tile loader. */
int mrx_llg_hinv_parts(const float* eta, const float* yt, const float* S, const void* mask, int mask_kind,
                       const int64_t* mstride, float* out4, float* work, int* nparts, int B, int C, int H, int W,
                       float inv_sigma2, int norm, int centered, void* stream);

/* A9 at the fastMRI knee width (W = 372), row-invariant masks: the same gradient as mrx_llg_hinv from ONE launch of wave-private
 * prime-factor (12 x 31) row transforms (csrc/llg372.hip, pfa372.h).  The loop-invariant operands are laid out once per slice in the
 * order the lanes consume them:
 *   mrx_llg372_prepare  yt (= IFFT_H(y), mrx_fft_cols), S [B,C,H,372,2], mask (column / batch dependent only) ->
 *                       ytp, Sp (mrx_llg372_operand_floats(B,C,H) floats each), maskp (B*372 floats)
 *   mrx_llg372          eta [B,H,372,2] -> nparts != NULL: *nparts = ceil(C/5) partial planes work[k][B][H][372][2] left for the
 *                       consumer (mrx_rim_layer_indrnn_packed_llg), out4 untouched; nparts == NULL: out4 [B,4,H,372] complete.
 *                       work: mrx_llg372_work_floats(B,C,H) floats.
 *                       ytp == NULL: the data is not read (the gradient is affine in eta: A^H M A eta - A^H M y); `work` has one plane more,
 *                       plane T = ceil(C/5) holds the constant term, and T + 1 partial planes are reported / combined.  34.5 MB instead of
 *                       63.1 MB per launch at 15 coils.  With a zero constant plane this is the linear part, which is its own adjoint
 *                       (training's backward through rim_utils.py:53-62).
 *   mrx_llg372_const_plane  -A^H M y -> plane T of `work` (mrx_llg372_work_floats + B*H*372*2 floats; planes 0 .. T-1 are scratch): the full
 *                       kernel on eta = 0 and the sum of its partial planes, once per slice and normalization */
int mrx_llg372_supported(int W);
int64_t mrx_llg372_operand_floats(int B, int C, int H);
int64_t mrx_llg372_work_floats(int B, int C, int H);
int mrx_llg372_prepare(const float* yt, const float* S, const void* mask, int mask_kind, const int64_t* mstride, float* ytp,
                       float* Sp, float* maskp, int B, int C, int H, int centered, void* stream);
/*   mrx_llg372_gather   the ytp = NULL form on eta_out = eta + nine-tap gather of the final convolution's tap products taps [B,18,H,372] (+ b_final,
 *                       may be NULL): mrx_rim_final_gather of the previous RIM step (rim_block.py:239-248) folded into this step's gradient; eta_out
 *                       is written, bit-identical to mrx_rim_final_gather's */
int mrx_llg372_gather(const float* eta, const float* taps, const float* b_final, float* eta_out, const float* Sp, const float* maskp, int mask_batched,
                      float* out4, float* work, int* nparts, int B, int C, int H, float inv_sigma2, int norm, int centered, void* stream);
/* mrx_llg372_gather on the row-pre-summed tap planes of mrx_rim_layer2_f16_cb8_q (taps_q [B][3][H][372][2], edges: mrx_rim_taps_q_edge_floats(B, H, 372) floats);
 * eta_out is bit-identical to mrx_rim_final_gather_q's. */
int mrx_llg372_gather_q(const float* eta, const float* taps_q, const float* edges, const float* b_final, float* eta_out, const float* Sp, const float* maskp,
                        int mask_batched, float* out4, float* work, int* nparts, int B, int C, int H, float inv_sigma2, int norm, int centered, void* stream);
int mrx_llg372_const_plane(const float* ytp, const float* Sp, const float* maskp, int mask_batched, float* work, int B, int C, int H, int norm,
                           int centered, void* stream);
int mrx_llg372(const float* eta, const float* ytp, const float* Sp, const float* maskp, int mask_batched, float* out4, float* work,
               int* nparts, int B, int C, int H, float inv_sigma2, int norm, int centered, void* stream);
/* The two halves of that pipeline as row operators at W = 372 (natural k-space layout on the far side; Sp from mrx_llg372_prepare or
 * mrx_pfa372_prepare_maps): the W transforms of sens_expand / sens_reduce (vn_block.py:51-87) in the hybrid space, and the first / last pass
 * of the general log_likelihood_gradient around mrx_llg_cols_dc (the column pass of mrx_llg on its own, in place).
 *   mrx_pfa372_expand   out [B,C,H,372,2] = FFT_W(x * S); with pred != NULL: out = pred - where(mask, pred - ref, 0) * dc_weight[0] - that
 *   mrx_pfa372_reduce   sum_c conj(S) IFFT_W(k): out [B,H,372,2], or out4 [B,4,H,372] = (eta, post * sum); work: mrx_llg372_work_floats */
int mrx_pfa372_prepare_maps(const float* S, float* Sp, int B, int C, int H, int centered, void* stream);
int mrx_pfa372_expand(const float* x, const float* Sp, float* out, const float* pred, const float* ref, const void* mask, int mask_kind,
                      const int64_t* mstride, const float* dc_weight, int B, int C, int H, int norm, int centered, void* stream);
int mrx_pfa372_reduce(const float* k, const float* Sp, const float* eta, float* out, float* out4, float* work, int B, int C, int H,
                      float post, int norm, int centered, void* stream);
/* mrx_pfa372_expand that also returns red [B,H,372,2] = sum_c conj(S) IFFT_W(out) -- the sens_reduce the NEXT cascade begins with
 * (vn_block.py:71-87 applied to the result of :109-119) -- from the rows while they are in the wave's buffer: the coil stack and the maps
 * are not read again.  work: mrx_llg372_work_floats(B,C,H) floats. */
int mrx_pfa372_expand_reduce(const float* x, const float* Sp, float* out, const float* pred, const float* ref, const void* mask,
                             int mask_kind, const int64_t* mstride, const float* dc_weight, float* red, float* work, int B, int C, int H,
                             int norm, int centered, void* stream);
int mrx_llg_cols_dc(float* work, const float* y, const void* mask, int mask_kind, const int64_t* mstride, int B, int C, int H, int W,
                    int norm, int centered, void* stream);
/* The same three passes with the coil stack between them column-tiled, [B*C][W/4][H][4] complex (every 4-column tile of an image is one
 * contiguous block of H * 32 bytes), so that the column pass moves contiguous blocks (rim_utils.py:44-62 for masks that depend on the row):
 *   mrx_tile4_cols          x [nimg,H,W,2] -> [nimg][W/4][H][4] complex (the measured data, once per slice; W % 4 == 0)
 *   mrx_pfa372_expand_t4    out_t4 = FFT_W(x * S) in the tiled layout
 *   mrx_llg_cols_dc_t4      FFT_H -> mask * (k - y) -> IFFT_H in place on work_t4 (y_t4 tiled; mask indexed [b,c,h,w] as everywhere);
 *                           y_t4 == NULL: mask * k only -- the gradient is linear in (k - y), the caller adds the constant -A^H M y of the
 *                           slice as one more partial plane; mrx_llg_cols_dc_t4_supported(H, W): H <= 2048 and W % 4 == 0
 *   mrx_pfa372_reduce_t4    post * sum_c conj(S) IFFT_W(k_t4): out4 [B,4,H,372] = (eta, that), or with nparts != NULL the coil-group partial
 *                           sums left in work ([*nparts][B,H,372,2]) for mrx_rim_layer_indrnn_packed_llg; work: mrx_llg372_work_floats */
int mrx_tile4_cols(const float* x, float* out, int64_t nimg, int H, int W, void* stream);
int mrx_pfa372_expand_t4(const float* x, const float* Sp, float* out_t4, int B, int C, int H, int norm, int centered, void* stream);
/* mrx_pfa372_expand_t4 on eta_out = eta + the nine-tap gather of `taps` [B,18,H,372] (+ b_final [2] or NULL): mrx_rim_final_gather folded into the first pass
 * of the NEXT step's general-mask gradient (rim_block.py:239-248 then rim_utils.py:47-52).  eta_out [B,H,372,2] is written, bit-identical to
 * mrx_rim_final_gather's result; eta_out != eta. */
int mrx_pfa372_expand_t4_gather(const float* eta, const float* taps, const float* b_final, float* eta_out, const float* Sp, float* out_t4, int B, int C,
                                int H, int norm, int centered, void* stream);
int mrx_llg_cols_dc_t4(float* work_t4, const float* y_t4, const void* mask, int mask_kind, const int64_t* mstride, int B, int C, int H,
                       int W, int norm, int centered, void* stream);
int mrx_llg_cols_dc_t4_supported(int H, int W);
int mrx_pfa372_reduce_t4(const float* k_t4, const float* Sp, const float* eta, float* out4, float* work, int* nparts, int B, int C, int H,
                         float post, int norm, int centered, void* stream);
int mrx_rim_layer_indrnn_packed_llg(const float* eta, const float* part, int nparts, float inv_sigma2, const float* packed,
                                    const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                                    float* h_new, int B, int F, int H, int W, int k, int dil, void* stream);

/* K2  soft data consistency: out = where(mask, pred - ref, 0) * dc_weight[0]   (vn_block.py:109-110,
 * rim_block.py:256).  dc_weight is a device pointer (it is an nn.Parameter). */
int mrx_soft_dc(const float* pred, const float* ref, const void* mask, int mask_kind, const int64_t* mstride,
                const float* dc_weight, float* out, int B, int C, int H, int W, void* stream);
/* out = base - where(mask, pred - ref, 0) * dc_weight[0] - eta_k.  VarNet: base = pred (vn_block.py:109-117);
 * RIM DC tail: base = ref = masked_kspace (rim_block.py:256-267). */
int mrx_dc_combine(const float* base, const float* pred, const float* ref, const void* mask, int mask_kind,
                   const int64_t* mstride, const float* dc_weight, const float* eta_k, float* out, int B, int C,
                   int H, int W, void* stream);

/* N4  VSNet (models/variablesplittingnet/vsnet_block.py):
 *   mrx_hard_dc     out = ((1 - mask) * pred + mask * ref) * dc_weight[0]                               (:23-25)
 *   mrx_vs_average  out[b,c] = param[0] * (kspace[b,c] + pred[b,c]) + (1 - param[0]) * sx[b]            (:35-36 as called at :145;
 *                   sx [B,H,W,2] is broadcast over the coil axis) */
int mrx_hard_dc(const float* pred, const float* ref, const void* mask, int mask_kind, const int64_t* mstride,
                const float* dc_weight, float* out, int B, int C, int H, int W, void* stream);
int mrx_vs_average(const float* kspace, const float* pred, const float* sx, const float* param, float* out, int B, int C,
                   int H, int W, void* stream);

/* N4  sigmanet data-consistency layers (models/sigmanet/dc_layers.py), the pointwise pieces between the FFT / sensitivity kernels:
 *   mrx_coil_sum   out[b,h,w] = sum_c k[b,c,h,w] * mask (mask NULL = no mask): the layers sum k-space over axis -4  (:69-81,:368-378)
 *   mrx_dc_bcast   mode 0: out[b,c] = (a - y[b,c]) * mask                                                            (:82-87)
 *                  mode 1: out[b,c] = (1 - mask) * a + mask * (alpha[0] * a + (1 - alpha[0]) * y[b,c])              (:381, :463)
 *                  a: [B,H,W,2] broadcast over the coils (a_coils = 0) or [B,C,H,W,2] (a_coils = 1)
 *   mrx_lincomb    mode 0: out[i] = x[i % nx] - p[0] * g[i % ng]; mode 1: p[0] * x[i % nx] + (1 - p[0]) * g[i % ng]  (:96, :402) */
int mrx_coil_sum(const float* k, const void* mask, int mask_kind, const int64_t* mstride, float* out, int B, int C, int H,
                 int W, void* stream);
int mrx_dc_bcast(const float* a, int a_coils, const float* y, const void* mask, int mask_kind, const int64_t* mstride,
                 const float* alpha, int mode, float* out, int B, int C, int H, int W, void* stream);
int mrx_lincomb(const float* x, int64_t nx, const float* g, int64_t ng, const float* p, int mode, float* out, int64_t n,
                void* stream);
/* Conjugate-gradient pieces of the proximal layer (dc_layers.py:156-196), n complex elements per batch element:
 *   mrx_cdot     out[b] = (re, im) of sum a * conj(b); work: mrx_cdot_work_floats(B) floats; fixed-order reduction
 *   mrx_cg_step  alpha = rr * conj(pq) / |pq|^2;  x += alpha * p;  r -= alpha * q      (rr, pq: device [B][2])
 *   mrx_cg_dir   p = r + (rr_new / rr) * p
 * mrx_lincomb mode 2: out[i] = p[0] * g[i % ng] + x[i % nx]                            (:250, :254) */
int64_t mrx_cdot_work_floats(int B);
int mrx_cdot(const float* a, const float* b, float* out, float* work, int B, int64_t n, void* stream);
int mrx_cg_step(float* x, float* r, const float* p, const float* q, const float* rr, const float* pq, int B, int64_t n,
                void* stream);
int mrx_cg_dir(float* p, const float* r, const float* rr_new, const float* rr, int B, int64_t n, void* stream);

/* A10 ConvNonlinear / nn.Conv2d (models/rim/conv_layers.py:72-85,121-123; rnn_cells.py:23-38;
 * unet_block.py:251,255,185): NCHW fp32, stride 1, "same" output size, square kernel k, dilation dil,
 * padding dil*(k-1)/2 in `pad_mode`.  bias may be NULL.  fp32-input MFMA (exact fp32 fma chains). */
int mrx_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H,
               int W, int k, int dil, int pad_mode, int act, float slope, void* stream);

/* A10+A11 fused RIM layer (conv_layers.py:121-123 + rnn_cells.py:384-391):
 *     h_new = ReLU( Wih (1x1) * ReLU(conv_reppad(x) + b_conv) + b_ih + hh * h_prev )
 * x [B,Cin,H,W]; h_prev,h_new [B,F,H,W] (F = conv out = hidden, multiple of 32, <= 64); h_prev may be NULL
 * (zeros: rim_block.py:188-193).  b_conv / b_ih may be NULL. */
int mrx_rim_layer_indrnn(const float* x, const float* w_conv, const float* b_conv, const float* w_ih,
                         const float* b_ih, const float* hh, const float* h_prev, float* h_new, int B, int Cin,
                         int F, int H, int W, int k, int dil, void* stream);
/* Tuned variant of the fused layer for F = 64 and (k, dil) in {(5,1), (3,2), (3,1), (1,1)}: weights are packed once
 * (per weight update) into the MFMA operand order, then the packed buffer is passed to every step.
 *   mrx_rim_layer_pack_floats  size of the packed buffer in floats (-1 if unsupported)
 *   mrx_rim_layer_pack         w_conv [F,Cin,k,k] + w_ih [F,F,1,1] -> packed
 *   mrx_rim_layer_supported    1 if a tuned kernel exists for the shape */
int64_t mrx_rim_layer_pack_floats(int Cin, int F, int k);
int mrx_rim_layer_pack(const float* w_conv, const float* w_ih, float* packed, int Cin, int F, int k, void* stream);
int mrx_rim_layer_supported(int Cin, int F, int k, int dil);
int mrx_rim_layer_indrnn_packed(const float* x, const float* packed, const float* b_conv, const float* b_ih,
                                const float* hh, const float* h_prev, float* h_new, int B, int Cin, int F, int H,
                                int W, int k, int dil, void* stream);

/* Winograd F(2x2,3x3) variant of the tuned fused layer for the dilated middle layer of the RIM (k = 3, dil = 2, F = 64;
 * rim_block.py:70-121 with conv_kernels[1] = 3, conv_dilations[1] = 2): the dilation-2 convolution is four plain 3x3
 * convolutions on the (row, column) parity sub-lattices, each evaluated with 16 instead of 36 multiplies per 2x2 outputs.
 * Same contract as mrx_rim_layer_indrnn_packed; results differ from the direct form by fp32 round-off only (the transforms
 * use +-1 and 1/2 coefficients).  The packed buffer holds G g G^T per (cout, cin) and the 1x1 weights. */
/* The same layer (3x3, dilation 2, 64 -> 64 + IndRNN 1x1) as a DIRECT convolution on the bf16 matrix pipe with fp32 results: every fp32
 * operand is the exact sum of three bf16 terms, six term products per multiply (error O(2^-24), as an fp32 FMA chain).  Same contract as
 * mrx_rim_layer_indrnn_wino (rim_block.py:233-238); packed = mrx_rim_layer2_sb_pack(w_conv [64,64,3,3], w_ih [64,64,1,1]). */
int64_t mrx_rim_layer2_sb_pack_floats(void);
int mrx_rim_layer2_sb_pack(const float* w_conv, const float* w_ih /* or NULL */, const float* w_final /* [2,64,3,3] or NULL */, float* packed,
                           void* stream);
int mrx_rim_layer2_sb(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                      float* h_new, int B, int H, int W, void* stream);
/* The second layer AND the final convolution of a RIM step (rim_block.py:233-246; conv_layers.py:121-123 with kernel 3, dilation 1, no
 * activation) in two calls: mrx_rim_layer2_sb_taps is mrx_rim_layer2_sb that also runs the 64 -> 2 convolution's channel contraction on
 * h_new while the kernel still holds it in registers (taps [B][18][H][W]: taps[b][tap * 2 + co] = sum_c w_final[co][c][tap] h_new[b][c];
 * packed must hold w_final); mrx_rim_final_gather adds the nine shifted taps (replicate padding = clamped coordinates), the bias and eta:
 * eta_out [B,H,W,2] = eta + permute(conv(h_new) + b_final).  eta may be NULL (nothing added).
 * In mrx_rim_layer2_sb and mrx_rim_layer2_sb_taps h_new may be h_prev itself (every element is read by the lane that writes it). */
/* The convolution stage of that kernel on its own: y = act(conv3x3(x, dilation 1 | 2, zero | replicate padding) + bias), 64 -> 64 channels, on the
 * bf16 matrix pipe with fp32 results (conv_layers.py:121-123; the 64-channel layers of CascadeNet / VSNet / the Recurrent VarNet).
 * packed = mrx_rim_layer2_sb_pack(w, NULL, NULL). */
int mrx_conv3x3_sb_supported(int Cin, int Cout, int k, int dil);
/* The same operand split for convolutions of FEW input channels (Cin <= 8; k = 3 | 5, dilation 1) into Cout <= 128: the first layers of the
 * cascades (conv_layers.py:121-123; qRIM 5x5 8 -> 128, CascadeNet / VSNet 3x3 2 -> 64).  y = act(conv(x) + bias), zero or replicate padding;
 * packed: mrx_conv_sbs_pack_floats(Cout, k) floats from mrx_conv_sbs_pack. */
int mrx_conv_sbs_supported(int Cin, int Cout, int k, int dil);
int64_t mrx_conv_sbs_pack_floats(int Cout, int k);
int mrx_conv_sbs_pack(const float* w, float* packed, int Cin, int Cout, int k, void* stream);
int mrx_conv_sbs(const float* x, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int k, int pad_mode,
                 int act, float slope, void* stream);
/* mrx_conv_sbs_p16: mrx_conv_sbs in the reference's `precision: 16` inference arithmetic (torch.autocast(float16) around forward; conv_layers.py:121-123 under it): x
 * and W rounded to fp16 once (the first term of the same pack, behind the same exact power-of-two scales), fp32 sums.  MRIDC_AMD_ARITH = f16x2. */
int mrx_conv_sbs_p16(const float* x, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int k, int pad_mode,
                 int act, float slope, void* stream);
int mrx_conv3x3_sb(const float* x, const float* packed, const float* bias, float* y, int B, int H, int W, int dil, int pad_mode, int act,
                   float slope, void* stream);
/* mrx_conv3x3_sb for CHAINS of 64-channel convolutions (CascadeNet, VSNet, the Recurrent VarNet, RIM-GRU / MGU): every call folds max |y| into
 * *xmax_out (device scalar the caller zeroes; NULL: not kept); a call that gets the bound of its input (xmax_in = the previous call's xmax_out)
 * multiplies two-term fp16 operands (packed_f16 = mrx_rim_layer2_f16_pack(w, NULL, NULL): three term products) instead of three-term bf16
 * ones (packed_bf16 = mrx_rim_layer2_sb_pack(w, NULL, NULL): six).  Either pack may be NULL when its form is not the one selected. */
int mrx_conv3x3_sb_chain(const float* x, const float* packed_bf16, const float* packed_f16, const float* bias, float* y, const float* xmax_in,
                         float* xmax_out, int B, int H, int W, int dil, int pad_mode, int act, float slope, void* stream);
/* The same layer with the convolution's operands as TWO fp16 terms (11 + 11 significand bits) and three term products per multiply on
 * v_mfma_f32_32x32x16_f16 (half the MFMAs of the three-term bf16 form; error per product <= ~3 x 2^-22).  fp16's exponent range is narrow:
 * the weights are scaled by a power of two at pack time, x per launch from `xmax`, a device float holding an upper bound of max |x| that the
 * producer of x maintains (mrx_rim_layer_indrnn_packed_xmax / _llg_xmax).  taps may be NULL.  The 1x1 / tap stages keep the bf16 form. */
int mrx_rim_layer1_xmax_supported(int Cin, int F, int k, int dil);
/* mrx_rim_layer_indrnn_packed / _packed_llg that also fold the maximum of their (non-negative) outputs into *xmax (atomic max, never reset
 * by the library: a running upper bound; the caller zeroes it, e.g. once per cascade). */
int mrx_rim_layer_indrnn_packed_xmax(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh,
                                     const float* h_prev, float* h_new, float* xmax, int B, int Cin, int F, int H, int W, int k, int dil,
                                     void* stream);
int mrx_rim_layer_indrnn_packed_llg_xmax(const float* eta, const float* part, int nparts, float inv_sigma2, const float* packed,
                                         const float* b_conv, const float* b_ih, const float* hh, const float* h_prev, float* h_new,
                                         float* xmax, int B, int F, int H, int W, int k, int dil, void* stream);
int64_t mrx_rim_layer2_f16_pack_floats(void);
int mrx_rim_layer2_f16_pack(const float* w_conv, const float* w_ih /* or NULL */, const float* w_final /* [2,64,3,3] or NULL */, float* packed,
                            void* stream);
int mrx_rim_layer2_f16(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                       float* h_new, float* taps, const float* xmax, int B, int H, int W, void* stream);
int mrx_rim_layer2_sb_taps(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh,
                           const float* h_prev, float* h_new, float* taps, int B, int H, int W, void* stream);
int mrx_rim_final_gather(const float* taps, const float* b_final, const float* eta, float* eta_out, int B, int H, int W, void* stream);
/* HOST routine (no GPU): variable-density Poisson-disc sampling for Poisson2DMaskFunc (reconstruction/data/subsample.py:549-633: Bridson's dart
 * throwing with a per-pixel elliptical exclusion radius; a calibration rectangle of calib_y x calib_x pixels around the centre is set first).
 * mask: ny x nx bytes (0 / 1, overwritten); radius_x / radius_y: ny x nx floats >= 1.  The reference runs the loop on Numba's private generator;
 * this one draws from its own xoshiro256** stream seeded with `seed`: reproducible.  Returns the number of samples (or a negative MRX_E* code). */
int64_t mrx_poisson_disc_mask(int nx, int ny, int max_attempts, const float* radius_x, const float* radius_y, double calib_x, double calib_y,
                              uint64_t seed, unsigned char* mask);
/* The two RIM layers on CHANNEL-BLOCKED hidden states h[b][c / 8][y][x][c % 8] ("CB8", fp32) instead of [B,64,H,W] (rim_block.py:230-246 keeps its
 * states between time-steps: the layout between the kernels of a step is free).  A lane of the matrix-core accumulator layout owns four consecutive
 * channels of a block, so state accesses are 16-byte instructions (8 + 8 per image row of a wave instead of 32 + 32) and the second layer's loader
 * reads a pixel's eight channels of a chunk as 2 x 16 bytes instead of 8 x 4 from eight planes.  Results are bit-identical to the NCHW entry points.
 *   mrx_cb8_convert      : [B,C,H,W] -> [B][C/8][H][W][8] (to_cb8 = 1) or back (0); C % 8 == 0, x != y;
 *   mrx_rim_layer1_cb8   : mrx_rim_layer_indrnn_packed[_llg]_xmax (x [B,Cin,H,W] with eta NULL, or eta + coil-group partial sums) with h_prev / h_new CB8;
 *   mrx_rim_layer2_f16_cb8: mrx_rim_layer2_f16 with x, h_prev, h_new CB8 (taps stays [B][18][H][W]). */
int mrx_cb8_convert(const float* x, float* y, int B, int C, int H, int W, int to_cb8, void* stream);
int mrx_rim_layer1_cb8(const float* x, int Cin, const float* eta, const float* part, int nparts, float inv_sigma2, const float* packed,
                       const float* b_conv, const float* b_ih, const float* hh, const float* h_prev, float* h_new, float* xmax, int B, int H, int W,
                       void* stream);
int mrx_rim_layer2_f16_cb8(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                           float* h_new, float* taps, const float* xmax, int B, int H, int W, void* stream);
/* The tap products of the final convolution pre-summed along x inside the kernel (round 5): mrx_rim_layer2_f16_cb8_q leaves taps_q [B][3][H][W][2] (kernel row dy, pair (co 0, co 1):
 * P[dy,0](x - 1) + P[dy,1](x) + P[dy,2](x + 1) with replicate borders) and, for the two columns of every 32-pixel tile whose neighbour lives in another tile, the missing
 * products in `edges` (mrx_rim_taps_q_edge_floats(B, H, W) floats).  mrx_rim_final_gather_q (and mrx_llg372_gather_q, below) finish the 3x3: same result as the
 * 18-plane route up to the order of nine additions.  6 + 2 instead of 18 + 2 values per pixel in the gather, 3 instead of 10 stores per lane in the layer. */
int64_t mrx_rim_taps_q_edge_floats(int B, int H, int W);
int mrx_rim_layer2_f16_cb8_q(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                             float* h_new, float* taps_q, float* edges, const float* xmax, int B, int H, int W, void* stream);
int mrx_rim_final_gather_q(const float* taps_q, const float* edges, const float* b_final, const float* eta, float* eta_out, int B, int H, int W,
                           void* stream);
/* The reduced-precision INFERENCE route (csrc/rim_amp16.hip; round 6): the two RIM layers of a time-step in the arithmetic the reference's own inference
 * configuration runs -- `precision: 16` (projects/reconstruction/model_zoo/conf/base_cirim_run.yaml:132) = torch.autocast: every convolution multiplies fp16
 * operands (ONE term; fp32 accumulation on v_mfma_f32_32x32x16_f16) and returns fp16, while the FFT, the complex products of log_likelihood_gradient and eta stay
 * fp32 (rim_utils.py:11-67, rim_block.py:217-249).  Hidden states are fp16, channel-blocked h[b][c / 16][y][x][c % 16] (32 bytes per pixel and channel block).
 * Never the default: selected per model (RIMBlock.precision / MRIDC_AMD_PRECISION=16), checked against the autocast oracle (oracle/amp.py) at a stated tolerance.
 *   mrx_amp16_pack_floats(layer)  : size of the operand pack of layer 1 / 2 in floats;
 *   mrx_amp16_layer1_pack         : w_conv [64,Cin<=4,5,5], w_ih [64,64,1,1] -> packed;
 *   mrx_amp16_layer2_pack         : w_conv [64,64,3,3], w_ih [64,64,1,1], w_final [2,64,3,3] (or NULL) -> packed;
 *   mrx_amp16_layer1              : h_new = ReLU(W_ih ReLU(conv5x5_reppad(in) + b_conv) + b_ih + hh * h_prev); in = x [B,Cin,H,W] (eta NULL) or
 *                                   (eta [B,H,W,2], inv_sigma2 * sum of nparts <= 4 partial planes part [nparts][B][H][W][2]) as mrx_rim_layer1_cb8;
 *                                   h_prev (NULL = the zero state) / h_new: fp16 [B][4][H][W][16];
 *   mrx_amp16_layer2              : the dilation-2 3x3 layer + IndRNN cell on fp16 x / h_prev / h_new [B][4][H][W][16]; with taps_q / edges (both or neither) also
 *                                   the final convolution's tap products in the layout of mrx_rim_layer2_f16_cb8_q (fp32; finished by mrx_rim_final_gather_q /
 *                                   mrx_llg372_gather_q). */
int64_t mrx_amp16_pack_floats(int layer);
int mrx_amp16_layer1_pack(const float* w_conv, const float* w_ih, float* packed, int Cin, void* stream);
int mrx_amp16_layer2_pack(const float* w_conv, const float* w_ih, const float* w_final /* or NULL */, float* packed, void* stream);
int mrx_amp16_layer1(const float* x, int Cin, const float* eta, const float* part, int nparts, float inv_sigma2, const float* packed, const float* b_conv,
                     const float* b_ih, const float* hh, const void* h_prev, void* h_new, int B, int H, int W, void* stream);
int mrx_amp16_layer2(const void* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const void* h_prev, void* h_new,
                     float* taps_q, float* edges, int B, int H, int W, void* stream);
/* Complex instance normalisation around a regulariser (models/sigmanet/sensitivity_net.py:16-139): m = mean of every real and imaginary entry,
 * C = 2x2 covariance of (re - m, im - m) per batch element (sums over per_b complex values, divided by `divisor` -- the reference's
 * shape[2] * shape[3] - 1); coef[b] = {m, C^(1/2) row-major, C^(-1/2) row-major} (9 floats).  center = 0 takes the data as mean-free.
 * apply: x [B][C][plane] complex -> out [B C][2][plane] = clamp(C^(-1/2)(x - m), -6, 6) (the regulariser's channel-first input);
 * unapply: y [B C][2][plane] -> out [B][C][plane] complex = C^(1/2) y + m.  work: mrx_cnorm_work_doubles(B) doubles. */
int64_t mrx_cnorm_work_doubles(int B);
int mrx_cnorm_stats(const float* x, int B, int64_t per_b, double divisor, int center, float* coef, double* work, void* stream);
int mrx_cnorm_apply(const float* x, const float* coef, float* out, int B, int C, int64_t plane, void* stream);
int mrx_cnorm_unapply(const float* y, const float* coef, float* out, int B, int C, int64_t plane, void* stream);
/* The gather half of a 3x3 convolution into Cout <= 4 channels done as a 1x1 channel contraction Cin -> 9 Cout (mrx_conv2d with the weights
 * re-ordered to [tap * Cout + co][c]) + nine shifted adds: out[b][co] = bias[co] + sum_tap shift_tap(taps[b][tap * Cout + co]); replicate padding =
 * clamped coordinates, zero padding = taps outside the image dropped (conv_layers.py:121-123 for thin final layers, e.g. qrim_block.py:226-236). */
int mrx_taps_gather(const float* taps /* [B][Ct >= 9 Cout][H][W] */, const float* bias, float* out, int B, int Ct, int Cout, int H, int W,
                    int pad_mode, void* stream);
int64_t mrx_rim_layer_wino_pack_floats(int Cin, int F);
int mrx_rim_layer_wino_pack(const float* w_conv, const float* w_ih, float* packed, int Cin, int F, void* stream);
int mrx_rim_layer_indrnn_wino(const float* x, const float* packed, const float* b_conv, const float* b_ih,
                              const float* hh, const float* h_prev, float* h_new, int B, int Cin, int F, int H, int W,
                              void* stream);
/* Plain 3x3 convolution into 64 channels (or a multiple: one launch per block of 64, `packed` = the blocks' packs back to back) on the
 * Winograd kernel (no 1x1 stage): dilation 1 or 2, MRX_PAD_ZERO or
 * MRX_PAD_REPLICATE, bias (or NULL) and MRX_ACT_* fused.  packed = mrx_rim_layer_wino_pack(w, NULL, ...).  Replaces nn.Conv2d
 * (3x3, 64 out channels) in conv/conv2d.py:36-43, recurrentvarnet/conv2gru.py:71-79, recurrentvarnet.py:68-71. */
int mrx_conv3x3_wino_supported(int Cin, int Cout, int k, int dil);
int mrx_conv3x3_wino(const float* x, const float* packed, const float* bias, float* out, int B, int Cin, int Cout, int H,
                     int W, int dil, int pad_mode, int act, float slope, void* stream);

/* A11 stand-alone IndRNN cell (rnn_cells.py:295-312,384-391): h_new = ReLU(conv_zero_pad(x; w_ih, b_ih) + hh*h_prev). */
int mrx_indrnn_cell(const float* x, const float* w_ih, const float* b_ih, const float* hh, const float* h_prev,
                    float* h_new, int B, int Cin, int F, int H, int W, int k, int dil, void* stream);

/* K8  final conv + eta update (rim_block.py:239-248): eta_out[B,H,W,2] = eta + permute(conv_reppad(h)),
 * conv F -> 2 channels, kernel k, dilation dil, bias optional. */
int mrx_rim_final(const float* h, const float* w, const float* bias, const float* eta, float* eta_out, int B,
                  int F, int H, int W, int k, int dil, void* stream);

/* out[B,H,W,2] = permute(conv3x3(h; w [2,F,3,3], bias), (0,2,3,1)): a convolution into one complex image, zero or replicate padding
 * (the last layer of the CascadeNet / VSNet / Recurrent VarNet regularisers: conv/conv2d.py:36-43 + ccnn_block.py:133,
 * conv2gru.py:158-162 + recurrentvarnet.py:221).  3x3, dilation 1, W % 4 == 0, F % 4 == 0, 16-byte aligned pointers; other
 * shapes return MRX_EUNSUP (use mrx_conv2d and permute). */
int mrx_conv_to_complex(const float* h, const float* w, const float* bias, float* out, int B, int F, int H, int W, int k,
                        int dil, int pad_mode, void* stream);

/* A12 GRU / MGU gate math (rnn_cells.py:118-127, :255-261) on precomputed ih / hh conv outputs
 * ([B,3F,H,W] / [B,2F,H,W]); out may alias h. */
int mrx_gru_gates(const float* ih, const float* hh, const float* h, float* out, int B, int F, int64_t HW,
                  void* stream);
int mrx_mgu_gates(const float* ih, const float* hh, const float* h, float* out, int B, int F, int64_t HW,
                  void* stream);
/* Their backward (training RIMs with gated cells: the derivative of the lines above, which the reference leaves to autograd): given dy = dL/d out, the
 * gradients w.r.t. ih, hh (same shapes) and h in ONE pass; the gates are recomputed from ih / hh.  No aliasing between outputs and inputs. */
int mrx_gru_gates_bwd(const float* dy, const float* ih, const float* hh, const float* h, float* dih, float* dhh, float* dh, int B, int F, int64_t HW,
                      void* stream);
int mrx_mgu_gates_bwd(const float* dy, const float* ih, const float* hh, const float* h, float* dih, float* dhh, float* dh, int B, int F, int64_t HW,
                      void* stream);

/* A12 whole ConvGRUCell (gates = 3, rnn_cells.py:112-127) / ConvMGUCell (gates = 2, rnn_cells.py:249-261) with 1x1 `ih` and `hh`
 * kernels in ONE launch: both per-pixel GEMMs on the matrix cores and the gate math on the accumulators; the gates*F-channel conv
 * outputs are never materialised.  x [B,Cin,HW], h [B,F,HW] or NULL (zero state), out [B,F,HW] (must not alias x or h).
 * Cin = F = 64 only (mrx_gated_cell_supported); other shapes: mrx_conv2d x2 + mrx_gru_gates / mrx_mgu_gates.
 *   mrx_gated_cell_pack   w_ih [gates*F,Cin,1,1] and w_hh [gates*F,F,1,1] -> packed, mrx_gated_cell_pack_floats floats */
int mrx_gated_cell_supported(int Cin, int F, int k, int gates);
int64_t mrx_gated_cell_pack_floats(int Cin, int F, int gates);
int mrx_gated_cell_pack(const float* w_ih, const float* w_hh, float* packed, int Cin, int F, int gates, void* stream);
int mrx_gated_cell_1x1(const float* x, const float* h, const float* packed, const float* b_ih, float* out, int B, int Cin,
                       int F, int64_t HW, int gates, void* stream);
/* ... that also folds max |out| into the device scalar *xmax (atomic max; the caller zeroes it): the bound a following 64-channel convolution
 * scales its two-term fp16 operands by (mrx_conv3x3_sb_chain's xmax_in).  MRX_EUNSUP with MRIDC_AMD_ARITH=fp32. */
int mrx_gated_cell_1x1_xmax(const float* x, const float* h, const float* packed, const float* b_ih, float* out, float* xmax, int B, int Cin,
                            int F, int64_t HW, int gates, void* stream);

/* N4  Conv2dGRU layer of the Recurrent Variational Network (models/recurrentvarnet/conv2gru.py:139-157):
 *   update = sigmoid(Wu [x;h] + bu), reset = sigmoid(Wr [x;h] + br), delta = tanh(Wo [x; h*reset] + bo),
 *   h_new = h * (1 - update) + delta * update; out_relu (optional) = ReLU(h_new), the next layer's input.
 * mrx_conv2dgru_cell_1x1: one launch for 1x1 gates on 64 features (x, h [B,64,HW]; h NULL = zero state; bias [3][64] in the order
 *   update, reset, out, or NULL); mrx_conv2dgru_pack packs the three [64,128,1,1] gate weights (mrx_conv2dgru_pack_floats floats).
 * Other shapes: mrx_conv2d on the concatenated inputs + mrx_mul_sigmoid (h * sigmoid(pre), h NULL = zeros) + mrx_gru_blend. */
int mrx_conv2dgru_supported(int Cin, int F, int k);
int64_t mrx_conv2dgru_pack_floats(int F);
int mrx_conv2dgru_pack(const float* w_update, const float* w_reset, const float* w_out, float* packed, int F, void* stream);
int mrx_conv2dgru_cell_1x1(const float* x, const float* h, const float* packed, const float* bias, float* out,
                           float* out_relu, int B, int F, int64_t HW, void* stream);
/* ... that also folds max(out_relu) into the device scalar *xmax (atomic max; the caller zeroes it): mrx_conv3x3_sb_chain's xmax_in for the next layer. */
int mrx_conv2dgru_cell_1x1_xmax(const float* x, const float* h, const float* packed, const float* bias, float* out, float* out_relu, float* xmax,
                                int B, int F, int64_t HW, void* stream);
int mrx_mul_sigmoid(const float* h, const float* pre, float* out, int64_t n, void* stream);
int mrx_gru_blend(const float* h, const float* pre_update, const float* pre_out, float* out, float* out_relu, int64_t n,
                  void* stream);

/* Training path: backward kernels of the convolutional regulariser.  The reference differentiates
 * rim_block.py:217-249 through torch autograd (backward of Conv2d with ReplicationPad2d, ReLU, the IndRNN cell).
 *   mrx_conv_wgrad     dw[Cout,Cin,k,k] (= or +=) sum_{b,pixel} dy[b,co,pixel] * pad(x)[b,ci,pixel + tap*dil]; Cout = 64 on the matrix
 *                      cores, Cout <= 4 (k = 1, 3, 5) on the vector ALUs; work: mrx_conv_wgrad_work_floats floats; fixed-order reduction
 *   mrx_reppad_fold    adjoint of ReplicationPad2d(pad): g [planes,H+2pad,W+2pad] -> out [planes,H,W].  The data gradient of a
 *                      replicate-padded conv is mrx_conv2d (zero 'same' padding, flipped + transposed weights) of dy zero-extended by
 *                      pad on every side, folded by this kernel
 *   mrx_relu_bwd       dpre = dy * (y > 0); sums[c] = (sum dpre, sum dpre * h_prev); with h_prev also dh_prev = dpre * hh[c]
 *                      (IndRNN cell, rnn_cells.py:384-391); work: mrx_relu_bwd_work_floats(C) floats */
int64_t mrx_conv_wgrad_work_floats(int B, int Cin, int Cout, int H, int W, int k);
int mrx_conv_wgrad(const float* x, const float* dy, float* dw, float* work, int B, int Cin, int Cout, int H, int W, int k,
                   int dil, int pad_mode, int accumulate, void* stream);
int mrx_reppad_fold(const float* g, float* out, int64_t planes, int H, int W, int pad, void* stream);
/* the same fold when the interior of the gradient is in `out` already and only the frame of width pad is in g [planes, H + 2 pad, W + 2 pad]
 * (mrx_conv2d_bf16_dgrad_rep): adds the frame onto the 2 (H + W) - 4 edge pixels of every plane */
int mrx_reppad_fold_edges(const float* g, float* out, int64_t planes, int H, int W, int pad, void* stream);
int64_t mrx_relu_bwd_work_floats(int C);
int mrx_relu_bwd(const float* dy, const float* y, const float* h_prev, const float* hh, float* dpre, float* dh_prev,
                 float* sums, float* work, int B, int C, int64_t HW, void* stream);
/* The explicit training tape (mridc_amd/training.py; the reference gets these steps from torch autograd over rim_block.py:217-249):
 *   mrx_relu_bwd_acc   mrx_relu_bwd with the upstream gradient given as two addends (dy + dy2, dy2 may be null) and the per-channel sums ADDED
 *                      into the gradient buffers: acc_bias[c] += sum dpre, acc_hh[c] += sum dpre * h_prev (either may be null)
 *   mrx_eta_grad_in    tot [B,H,W,2] = carry (or 0) + gl;  d2 [B,2,H,W] = tot channel-first (the gradient entering the final convolution)
 *   mrx_g4_to_complex  dz [B,H,W,2] = channels 2, 3 of g4 [B,4,H,W]
 *   mrx_eta_grad_out   out [B,H,W,2] = tot + channels 0, 1 of g4 + channels 2, 3 of t4 (identity path, eta channels, adjoint log-likelihood gradient) */
int mrx_relu_bwd_acc(const float* dy, const float* dy2, const float* y, const float* h_prev, const float* hh, float* dpre, float* dh_prev,
                     float* acc_bias, float* acc_hh, float* work, int B, int C, int64_t HW, void* stream);
int mrx_eta_grad_in(const float* carry, const float* gl, float* tot, float* d2, int B, int64_t plane, void* stream);
int mrx_g4_to_complex(const float* g4, float* dz, int B, int64_t plane, void* stream);
int mrx_eta_grad_out(const float* tot, const float* g4, const float* t4, float* out, int B, int64_t plane, void* stream);

/* Mixed precision for the training path (BASELINE config 4; the reference trains under AMP, base_cirim_train.yaml:180): convolutions with
 * bf16 operands and fp32 accumulation (v_mfma_f32_32x32x16_bf16) on the same fp32 NCHW tensors -- the tile loader rounds activations to
 * bf16, the weights are packed to bf16 once per version.
 *   mrx_conv_bf16_pack   w [Cout,Cin,k,k] fp32 -> packed (mrx_conv_bf16_pack_bytes bytes); transposed = 1 packs the flipped, channel-
 *                        transposed weights of the data gradient (then Cin / Cout are those of the GRADIENT convolution: w is [Cin,Cout,k,k])
 *   mrx_conv2d_bf16      out = act(conv(x) + bias [+ hh * hprev]), 'same' size, stride 1 (conv_layers.py:121-123; with hprev the
 *                        IndRNN cell of rnn_cells.py:384-391); shapes: mrx_conv_bf16_supported */
int mrx_conv_bf16_supported(int Cin, int Cout, int k, int dil);
int64_t mrx_conv_bf16_pack_bytes(int Cin, int Cout, int k);
int mrx_conv_bf16_pack(const float* w, void* packed, int Cin, int Cout, int k, int transposed, void* stream);
int mrx_conv2d_bf16(const float* x, const void* packed, const float* bias, const float* hh, const float* hprev, float* out, int B,
                    int Cin, int Cout, int H, int W, int k, int dil, int pad_mode, int act, float slope, void* stream);
/*   mrx_conv2d_bf16_ext  the zero-padded convolution of x [B,Cin,H,W] read as if zero-extended by `ext` pixels on every side ->
 *                        out [B,Cout,H + 2 ext,W + 2 ext]: with a transposed pack, the data gradient on the padded domain (then mrx_reppad_fold) */
int mrx_conv2d_bf16_ext(const float* x, const void* packed, float* out, int B, int Cin, int Cout, int H, int W, int k, int dil, int ext,
                        void* stream);
/*   mrx_conv2d_bf16_dgrad_rep  data gradient of a replicate-padded 'same' convolution: interior written straight into dx [B,Cout,H,W], the frame
 *                        of width dil (k - 1) / 2 into frame [B,Cout,H + 2 p,W + 2 p]; follow with mrx_reppad_fold_edges(frame, dx, ...) */
int mrx_conv2d_bf16_dgrad_rep(const float* dy, const void* packed, float* dx, float* frame, int B, int Cin, int Cout, int H, int W, int k,
                              int dil, void* stream);
/*   mrx_conv_wgrad_bf16  dw [64,64,k,k] (= or +=) sum over (b, pixel) of dy * padded x (the weight gradient of a 64 -> 64 'same' convolution),
 *                        bf16 operands, fp32 accumulation per workgroup, fixed-order double sum of the workgroup partials; k = 1, or k = 3 with
 *                        dilation 2 (mrx_conv_wgrad_bf16_supported); work: mrx_conv_wgrad_bf16_work_floats floats */
int mrx_conv_wgrad_bf16_supported(int Cin, int Cout, int k, int dil);
int64_t mrx_conv_wgrad_bf16_work_floats(int B, int H, int W, int k);
int mrx_conv_wgrad_bf16(const float* x, const float* dy, float* dw, float* work, int B, int H, int W, int k, int dil, int pad_mode,
                        int accumulate, void* stream);
/* ... and for every shape of mrx_conv_wgrad_bf16_supported: the 64 -> 64 layers above plus the two thin dilation-1 layers of the RIM
 * (3x3 64 -> Cout <= 32: the final conv; 5x5 Cin <= 32 -> 64: the first conv); dw [Cout,Cin,k,k] */
int64_t mrx_conv_wgrad_bf16_any_work_floats(int B, int Cin, int Cout, int H, int W, int k);
int mrx_conv_wgrad_bf16_any(const float* x, const float* dy, float* dw, float* work, int B, int Cin, int Cout, int H, int W, int k, int dil,
                            int pad_mode, int accumulate, void* stream);

/* ---- Pointwise / per-plane backward steps of the U-Net training path (csrc/diff_bwd.hip; reference: torch autograd through
 * unet_block.py:189-299 -- LeakyReLU, InstanceNorm2d, avg_pool2d, ConvTranspose2d):
 *   mrx_act_bwd            dx = dy * act'(y), y the activation's output
 *   mrx_inorm_act_bwd      backward of act(InstanceNorm2d(x)) from the activation's output y and the forward's partial sums (`work` of
 *                          mrx_instance_norm_act): dx = rstd (g - mean g - z mean(g z)), z the normalised value, g = dy act'(y); act NONE | LEAKY;
 *                          work: mrx_inorm_act_bwd_work_floats floats
 *   mrx_avgpool2x2_bwd     adjoint of mrx_avg_pool2x2 (an odd last row / column receives 0)
 *   mrx_pixel_unshuffle2   [BC,2H,2W] -> [BC,4,H,W]: the layout in which ConvTranspose2d(k 2, s 2)'s gradients are 1x1 GEMMs */
int mrx_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act, float slope, void* stream);
int64_t mrx_inorm_act_bwd_work_floats(int64_t planes, int64_t n);
int mrx_inorm_act_bwd(const float* dy, const float* y, const float* fwd_work, float* dx, float* work, int64_t planes, int64_t HW, float eps, int act,
                      float slope, void* stream);
int mrx_avgpool2x2_bwd(const float* dy, float* dx, int64_t planes, int H, int W, void* stream);
int mrx_pixel_unshuffle2(const float* x, float* out, int64_t BC, int H, int W, void* stream);
/* The pointwise halves of the backward of the VarNet block's coil operators and data consistency (vn_block.py:51-119; the FFT halves are the
 * opposite transforms of this library):
 *   mrx_cmul_bcast          out[b,c,n] = a[b,c,n] * v[b,n] (conj_v: * conj(v[b,n])) * scale; complex, a [B,C,N], v [B,N]
 *                           (sens_reduce: dS_c = conj(dy) * ifft2(k)_c)
 *   mrx_sens_expand_bwd_pw  from G_c = adjoint-fft2(dy_c): dx[b,n] = scale * sum_c conj(S_c) G_c, dS_c = scale * conj(x) G_c (dx or dS may be NULL;
 *                           x may be NULL when dS is)
 *   mrx_dc_combine_bwd      backward of out = base - where(mask, pred - ref, 0) * w - eta_k: dpred = -where(mask, dy, 0) * w (+ dy when add_dy: base
 *                           and pred are one tensor), deta = -dy, dw[0] = -sum where(mask, (pred - ref) . dy); any of the three outputs may be NULL;
 *                           work: mrx_dc_combine_bwd_work_doubles doubles (needed with dw) */
int mrx_cmul_bcast(const float* a, const float* v, float* out, int64_t B, int64_t C, int64_t N, int conj_v, float scale, void* stream);
int mrx_sens_expand_bwd_pw(const float* G, const float* S, const float* x, float* dx, float* dS, int64_t B, int64_t C, int64_t N, float scale,
                           void* stream);
int64_t mrx_dc_combine_bwd_work_doubles(void);
int mrx_dc_combine_bwd(const float* dy, const float* pred, const float* ref, const void* mask, int mask_kind, const int64_t* mstride,
                       const float* dc_weight, float* dpred, float* deta, float* dw, double* work, int add_dy, int B, int C, int H, int W, void* stream);

/* ---- Mixed-precision training with bf16 STORAGE (BASELINE config 4; csrc/train_bf16.hip, csrc/conv_bf16.hip) -------------------------------------
 * The reference trains under pytorch-lightning AMP (projects/reconstruction/model_zoo/conf/base_cirim_train.yaml:180 `precision: 16`): torch.autocast
 * runs every Conv2d of rim_block.py:230-246 on half-precision operands and RETURNS half-precision tensors (so the gradients flowing into them are
 * half precision too); hidden states, `hh * hx` (rnn_cells.py:390), FFTs, eta and the loss stay fp32.  Here half = bf16, and such tensors are
 * "pair tensors" in HBM: uint32 [B][C/2][H][W] = (bf16 of channel 2p) | (bf16 of channel 2p + 1) << 16.
 *   mrx_tl_pack          W_ih [64,64] (+ the final convolution's weights [2,64,3,3], may be null) in the operand orders of the three GEMMs below
 *   mrx_tl_layer_fwd     one RIM layer (ConvNonlinear + IndRNNCell, conv_layers.py:121-123, rnn_cells.py:384-391): a = ReLU(bf16(conv(x) + b)) ->
 *                        a_pairs, h = ReLU(bf16(W_ih a + b_ih) + hh * h_prev) -> h (fp32).  HIDDEN STATES ARE CHANNEL-BLOCKED in this tape:
 *                        h, h_prev (and x when Cin == 64) are [B,8,H,W,8], channel c = 8 q + j at [b][q][y][x][j] (32 contiguous bytes per pixel
 *                        and block: 16-byte accesses everywhere instead of dwords of 64 planes).  taps != null: also the final convolution's (tap, cout)
 *                        products with bf16(h) [B,18,H,W], summed by mrx_tl_final_gather: eta_out = eta + bf16(sum of the 9 shifted planes)
 *                        hmask (may be NULL): uint32 [B,H,W,2] receives (h > 0) as 64 bits per pixel (word w, bit 16 c2 + 4 k + m = channel
 *                        32 c2 + 8 k + 4 w + m): all that mrx_tl_cell_bwd needs of h -- 8 bytes per pixel instead of 256
 *   mrx_tl_cell_bwd      backward of the cell and of the convolution's ReLU in one pass (see train_bf16.hip; the state as h or, if hmask != NULL,
 *                        as the forward's mask bits -- h is then not read and may be NULL); parameter-gradient partials accumulate
 *                        in `part` (mrx_tl_cell_part_floats floats; first != 0 overwrites) until mrx_tl_cell_reduce adds them to the gradients
 *   mrx_tl_dgrad         data gradient of a replicate-padded convolution with bf16 results: interior -> dx, frame -> `frame` for mrx_tl_fold_edges
 *   mrx_conv_wgrad_bf16_pairs   the weight gradients of mrx_conv_wgrad_bf16_any with dy given as a pair tensor */
int64_t mrx_tl_pack_bytes(void);
int mrx_tl_pack(const float* w_ih, const float* w_fin, void* packed, void* stream);
int mrx_tl_layer_fwd(const float* x, const void* conv_packed, const float* conv_bias, const void* tl_packed, const float* ih_bias, const float* hh,
                     const float* hprev, void* a_pairs, float* h, void* hmask, float* taps, int B, int Cin, int H, int W, int k, int dil, void* stream);
int mrx_tl_final_gather(const float* taps, const float* eta, float* eta_out, int B, int H, int W, void* stream);
/*   mrx_tl_final_gather_max   ... and the per-workgroup maxima of |eta_out| (complex modulus, formed as mrx_max_abs forms it) in max_partials
 *                        [mrx_tl_final_gather_max_count]: the training loss's maximum (cirim.py:218-237) without a pass of its own -- mrx_absl1_loss_mp
 *                        reduces the partials */
int64_t mrx_tl_final_gather_max_count(int B, int H, int W);
int mrx_tl_final_gather_max(const float* taps, const float* eta, float* eta_out, float* max_partials, int B, int H, int W, void* stream);
int64_t mrx_tl_cell_part_floats(int B, int H, int W);
int mrx_tl_cell_bwd(const void* dh_above, const float* dH, const float* h, const void* hmask, const float* hprev, const void* a_pairs,
                    const void* tl_packed, const float* hh, float* dh_prev, void* ga_pairs, float* part, int first, int B, int H, int W, void* stream);
int mrx_tl_cell_reduce(const float* part, int B, int H, int W, float* dw_ih, float* db_ih, float* dhh, float* db_conv, void* stream);
int mrx_tl_dgrad(const void* dy, int dy_pairs, const void* packed, void* dx, int dx_pairs, float* frame, int B, int Cdy, int Cdx, int H, int W, int k,
                 int dil, void* stream);
/* mrx_tl_dgrad_l2w: mrx_tl_dgrad with every shape on the generic kernel (weights streamed from L2) -- the 64 -> 64 form with the weights resident in
 * LDS that mrx_tl_dgrad runs is pinned against it, bit for bit */
int mrx_tl_dgrad_l2w(const void* dy, int dy_pairs, const void* packed, void* dx, int dx_pairs, float* frame, int B, int Cdy, int Cdx, int H, int W, int k,
                     int dil, void* stream);
int mrx_tl_fold_edges(const float* frame, void* dx, int dx_pairs, int B, int C, int H, int W, int pad, void* stream);
int mrx_tl_pairs_to_f32(const void* pairs, float* out, int64_t pair_planes, int64_t plane, void* stream);
int mrx_tl_f32_to_pairs(const float* x, void* pairs, int64_t pair_planes, int64_t plane, void* stream);
/*   mrx_tl_wgrad_in      the first layer's weight gradient (5x5, Cin <= 5 -> 64, replicate padding) with dy a pair tensor: one column block of
 *                        (ci, tap) per wave; work: mrx_tl_wgrad_in_work_floats floats */
int64_t mrx_tl_wgrad_in_work_floats(int B, int Cin, int H, int W);
int mrx_tl_wgrad_in(const float* x, const void* dy_pairs, float* dw, float* work, int B, int Cin, int H, int W, int accumulate, void* stream);
int mrx_conv_wgrad_bf16_pairs(const float* x, const void* dy_pairs, float* dw, float* work, int B, int Cin, int H, int W, int k, int dil, int pad_mode,
                              int accumulate, int x_blocked, void* stream);   /* x_blocked: x is channel-blocked [B,8,H,W,8] (3x3 dilation 2 only) */
/*   mrx_conv_wgrad_bf16_xcb   weight gradient of the final 3x3 convolution 64 -> Cout <= 32 with x channel-blocked [B,8,H,W,8], dy fp32;
 *                             work: mrx_conv_wgrad_bf16_any_work_floats(B, 64, Cout, H, W, 3) floats */
int mrx_conv_wgrad_bf16_xcb(const float* x_cb8, const float* dy, float* dw, float* work, int B, int Cout, int H, int W, int pad_mode, int accumulate,
                            void* stream);

/*   mrx_absl1_loss      the l1 training loss of one prediction (cirim.py:218-237): out2[0] = mean |target - |p| / max|p||, out2[1] = an
 *                      intermediate the backward needs; p complex [n], target real [n], maxabs = device scalar from mrx_max_abs (mode 1);
 *                      work: mrx_absl1_work_floats floats.  mrx_absl1_loss_bwd: dp = gout * gscale * d(loss)/dp incl. the path through the max
 *   mrx_adam_step      torch.optim.Adam (weight_decay 0, amsgrad off) on flat buffers; `step` counts from 1; grad_scale multiplies the
 *                      gradient first (1 / world size after the all-reduce(sum) of data-parallel training) */
int64_t mrx_absl1_work_floats(void);
int mrx_absl1_loss(const float* p, const float* target, const float* maxabs, float* out2, float* work, int64_t n, void* stream);
int mrx_absl1_loss_bwd(const float* p, const float* target, const float* maxabs, const float* fwd_out2, const float* gout,
                       float gscale, float* dp, int64_t n, void* stream); /* gout: device scalar (upstream gradient) or NULL = 1 */
/*   mrx_absl1_loss_mp        mrx_absl1_loss with the maximum given as np per-workgroup partial maxima; the maximum is left in maxabs_out[0]
 *   mrx_absl1_loss_bwd_eta   mrx_absl1_loss_bwd + mrx_eta_grad_in in one pass: tot [B,plane,2] = carry (may be NULL) + d(loss)/dp, d2 [B,2,plane]
 *   mrx_eta_grad_out_parts   mrx_eta_grad_out with the adjoint gradient still in mrx_llg372's partial planes: out = tot + g4[:, 0:2] + post * sum_k parts_k */
int mrx_absl1_loss_mp(const float* p, const float* target, const float* max_partials, int np, float* maxabs_out, float* out2, float* work, int64_t n,
                      void* stream);
int mrx_absl1_loss_bwd_eta(const float* p, const float* target, const float* maxabs, const float* fwd_out2, const float* gout, float gscale,
                           const float* carry, float* tot, float* d2, int B, int64_t plane, void* stream);
int mrx_eta_grad_out_parts(const float* tot, const float* g4, const float* parts, int nparts, float post, float* out, int B, int64_t plane, void* stream);
int mrx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale, void* stream);

/* 1x1 convolution 64 -> 64 as a per-pixel GEMM on the matrix cores, HBM-bound: out = act(W x + bias [+ hh * h_prev]) with x, out,
 * h_prev [B,64,HW]; the IndRNN cell with a 1x1 `ih` when hh / h_prev are given (rnn_cells.py:384-391), RecurrentInit's heads
 * (recurrentvarnet.py:73-76), data gradients of 1x1 layers.  packed: 4096 floats from mrx_conv1x1_64_pack(w [64,64,1,1]). */
int mrx_conv1x1_64_pack(const float* w, float* packed, void* stream);
/* the same for C = 64 or 128 channels (qCIRIM's 128-feature IndRNN cells): packed = C*C floats */
int mrx_conv1x1_sq_supported(int Cin, int Cout);
/* mrx_conv1x1_sq_head128: out [B,64,P] = the first 64 rows of a 128 x 128 matrix times x (no bias / epilogue): the channel contraction of a
 * thin 3x3 convolution of 128 channels (<= 7 output channels' 9 tap rows) without the unused outputs. */
int mrx_conv1x1_sq_head128(const float* x, const float* packed, float* out, int B, int64_t HW, void* stream);
int64_t mrx_conv1x1_sq_pack_floats(int C);   /* floats of `packed` (the fp32 operands and, at C = 128, the split-bf16 ones) */
int mrx_conv1x1_sq_pack(const float* w, float* packed, int C, void* stream);
int mrx_conv1x1_sq(const float* x, const float* packed, const float* bias, const float* hh, const float* h_prev, float* out,
                   int B, int C, int64_t HW, int act, float slope, void* stream);
/* mrx_conv1x1_sq that also folds max |out| into the device scalar *xmax (atomic max; the caller zeroes it): the operand bound a two-term fp16
 * consumer of `out` needs (mrx_conv3x3_h), without a pass over the tensor.  C = 128 on the matrix-pipe kernel (mrx_conv1x1_sq_xmax_supported). */
int mrx_conv1x1_sq_xmax_supported(int C);
int mrx_conv1x1_sq_xmax(const float* x, const float* packed, const float* bias, const float* hh, const float* h_prev, float* out, float* xmax,
                        int B, int C, int64_t HW, int act, float slope, void* stream);
/* mrx_conv1x1_sq_p16: the same at C = 128 in the reference's `precision: 16` inference arithmetic (base_qcirim_run.yaml:204 = torch.autocast(float16) around forward;
 * rnn_cells.py:384-391 under it): x and W rounded to fp16 once (a plain cast, as autocast's), fp32 sums; hh * h_prev, bias, activation in fp32.  xmax may be null. */
int mrx_conv1x1_sq_p16(const float* x, const float* packed, const float* bias, const float* hh, const float* h_prev, float* out, float* xmax,
                       int B, int C, int64_t HW, int act, float slope, void* stream);
int mrx_conv1x1_64(const float* x, const float* packed, const float* bias, const float* hh, const float* h_prev, float* out,
                   int B, int64_t HW, int act, float slope, void* stream);

/* A17 NormUnet support (models/unet_base/unet_block.py).  Planes are [B*C] images of H*W floats.
 *   mrx_instance_norm_act   InstanceNorm2d (biased var, eps, no affine) + activation, in place allowed   (:252-253,:294-295)
 *   mrx_group_norm_stats    per-group mean and UNBIASED std over n contiguous floats                       (:78-79)
 *   mrx_group_norm_apply    inverse=0: (x-mean)/std ; inverse=1: x*std+mean                               (:81,:91)
 *   mrx_pad2d               out[y][x] = in[y-top][x-left]; mode 0 zeros (negative pads crop = unpad), 1 reflect (:93-111,:215-222)
 *   mrx_avg_pool2x2         avg_pool2d(kernel 2, stride 2)                                                 (:206)
 *   mrx_conv_transpose2x2   ConvTranspose2d(k=2, s=2, bias=False), weight [Cin,Cout,2,2]                   (:293)
 *   mrx_copy_channels       dst[:, c0:c0+C] = src for the skip concat                                      (:224) */
/* work: caller scratch of mrx_norm_work_floats(planes|groups, HW|n) floats (statistics are reduced by many workgroups per plane) */
int64_t mrx_norm_work_floats(int64_t planes, int64_t n);
/* Conv3x3 + InstanceNorm statistics in one pass over the conv accumulators (unet_block.py:251-253: Conv2d -> InstanceNorm2d):
 *   mrx_conv2d_stats          y = conv(x) (no activation) and stats[b][co] = (mean, sum of squared deviations) of every output plane;
 *                             3x3, dilation 1, Cout <= 64 only (mrx_conv2d_stats_supported); work = mrx_conv2d_stats_work_floats floats
 *   mrx_instance_norm_apply   out = act((x - mean) / sqrt(M2 / HW + eps)) from those statistics, in place allowed */
int64_t mrx_conv2d_stats_work_floats(int B, int Cout, int H, int W);
int mrx_conv2d_stats_supported(int B, int Cout, int H, int W, int k, int dil);
int mrx_conv2d_stats(const float* x, const float* w, const float* bias, float* y, float* stats, float* work, int B, int Cin, int Cout,
                     int H, int W, int k, int dil, int pad_mode, void* stream);
int mrx_instance_norm_apply(const float* x, float* out, const float* stats, int64_t planes, int64_t HW, float eps, int act, float slope,
                            void* stream);
/*   mrx_instance_norm_apply_tiles   the same from the per-tile statistics mrx_conv2d_stats leaves in `work` when called with
 *                                   stats = NULL (every workgroup merges the tiles of its plane itself: one launch fewer) */
int mrx_instance_norm_apply_tiles(const float* x, float* out, const float* tile_stats, int B, int C, int H, int W, float eps, int act,
                                  float slope, void* stream);
int mrx_instance_norm_act(const float* x, float* out, float* work, int64_t planes, int64_t HW, float eps, int act,
                          float slope, void* stream);
int mrx_group_norm_stats(const float* x, float* mean, float* std_, float* work, int64_t groups, int64_t n, void* stream);
int mrx_group_norm_apply(const float* x, const float* mean, const float* std_, float* out, int64_t groups, int64_t n,
                         int inverse, void* stream);
/*   mrx_group_norm_bwd      the derivative of the two lines above (training; the reference leaves it to autograd), two launches:
 *                           inverse = 0: dx from dy (gradient of the normalised tensor), v = the normalised tensor, std and the OPTIONAL gradients dmean, dstd
 *                                        [groups] of the statistics (NormUnet un-normalises with them at its end: unet_block.py:86-91);
 *                           inverse = 1: dx = dy std and dmean_o = sum dy, dstd_o = sum dy v per group, v = the un-normalisation's input.
 *                           work: mrx_norm_work_floats(groups, n) floats; n >= 2 (unbiased std). */
int mrx_group_norm_bwd(const float* dy, const float* v, const float* std_, const float* dmean, const float* dstd, float* dx, float* dmean_o,
                       float* dstd_o, float* work, int64_t groups, int64_t n, int inverse, void* stream);
int mrx_pad2d(const float* in, float* out, int64_t planes, int H, int W, int top, int bottom, int left, int right,
              int mode, void* stream);
int mrx_avg_pool2x2(const float* in, float* out, int64_t planes, int H, int W, void* stream);
int mrx_conv_transpose2x2(const float* x, const float* w, float* out, int B, int Cin, int Cout, int H, int W,
                          void* stream);
/* the same + the InstanceNorm statistics (mean, sum of squared deviations) [B,Cout,2] of its output out of the same accumulators
 * (unet_block.py:296-299: ConvTranspose2d -> InstanceNorm2d -> LeakyReLU becomes this + mrx_instance_norm_apply); even Cout, tuned
 * shapes only (MRX_EUNSUP otherwise).  work: mrx_conv_transpose2x2_stats_work_floats() floats. */
int64_t mrx_conv_transpose2x2_stats_work_floats(int B, int Cout, int H, int W);
int mrx_conv_transpose2x2_stats(const float* x, const float* w, float* out, float* stats, float* work, int B, int Cin, int Cout, int H, int W,
                                void* stream);
int mrx_copy_channels(const float* src, float* dst, int B, int C, int64_t HW, int Ctot, int c0, void* stream);
/* torch.cat([a, b], dim=1) of [B,Ca,HW] and [B,Cb,HW] in one launch (the skip concat, unet_block.py:224) */
int mrx_concat_channels(const float* a, const float* b, float* out, int B, int Ca, int Cb, int64_t HW, void* stream);

/* A20 SSIMLoss.forward (mridc/collections/common/losses/ssim.py:28-61): X,Y [B,1,h,w], data_range [B] -> out[0] = 1 - mean(S).
 * work: mrx_ssim_work_floats(B,h,w) floats. */
int64_t mrx_ssim_work_floats(int B, int h, int w);
int mrx_ssim_loss(const float* X, const float* Y, const float* data_range, float* out, float* work, int B, int h, int w, int win,
                  float k1, float k2, void* stream);

/* A19 quantitative MRI (mridc/collections/quantitative/models/qrim/utils.py, qrim_block.py).
 *   mrx_dc_residual  out[b] = sum_c conj(S[b/sdiv,c]) ifft2(mask (fft2(x[b] S[b/sdiv,c]) - y[b,c]));  b = batch x echoes,
 *                    sdiv = echoes sharing one set of maps (utils.py:235-248).  x,out [B,H,W,2]; y,work [B,C,H,W,2]; S [B/sdiv,C,H,W,2]
 *   mrx_qmri_signal  MEGRE signal model (utils.py:71-121): maps [N,HW] x4 -> [N,E,HW,2]; `tes` is a HOST array of E echo times
 *   mrx_qmri_grad    analytic gradient (utils.py:250-295) from the coil-combined residual: -> [N,4,HW] = mean over echoes of
 *                    (R2*_re, S0_re, R2*_im, S0_im) * post, NaN -> 0 (qrim_block.py:223-224)
 *   mrx_scale        mode bit0: |x| first; bit1: divide instead of multiply   (qrim_block.py:198-201, qcirim.py:248-251,:334)
 *   mrx_qrim_update  out = eta + delta, channel 0 clamped at 0   (qrim_block.py:233-236) */
int mrx_dc_residual(const float* x, const float* y, const float* S, const void* mask, int mask_kind, const int64_t* mstride,
                    float* out, float* work, int B, int C, int H, int W, int sdiv, int norm, int centered, void* stream);
int mrx_qmri_signal(const float* r2, const float* s0, const float* b0, const float* phi, const float* tes, int E, float* out,
                    int64_t N, int64_t HW, float scaling, void* stream);
int mrx_qmri_grad(const float* dinv, const float* r2, const float* s0, const float* b0, const float* phi, const float* tes,
                  int E, float* out, int64_t N, int64_t HW, float scaling, float post, void* stream);
int mrx_scale(const float* x, float* out, int64_t n, float s, int mode, void* stream);
int mrx_qrim_update(const float* eta, const float* delta, float* out, int B, int Cc, int64_t HW, void* stream);

/* N1 per-sample preprocessing on the device (reconstruction/parts/transforms.py:286-288 target scaling, :527-617 max
 * normalisation).  The reductions keep their result on the device so the chain needs no host synchronisation.
 *   mrx_max_abs               out[0] = max |x_i| over n floats (mode 0: torch.max(torch.abs(x)) of a real view) or max complex
 *                             modulus over n complex values (mode 1); NaN propagates; work = mrx_max_abs_work_floats() floats
 *   mrx_div_by_device_scalar  out = x / d[0] (mode 0, n floats) or out[i] = |x_c[i] / d[0]| (mode 1: n complex -> n floats) */
/* N3 BaseSensitivityModel.divide_root_sum_of_squares (models/base.py:824-840): out[o,r,i] = x[o,r,i] / sqrt(sum_r |x[o,r,i]|^2),
 * x, out [outer, R, inner] complex. */
int mrx_div_rss_complex(const float* x, float* out, int64_t outer, int64_t R, int64_t inner, void* stream);
int64_t mrx_max_abs_work_floats(void);
int mrx_max_abs(const float* x, int64_t n, int mode, float* out, float* work, void* stream);
int mrx_div_by_device_scalar(const float* x, const float* d, float* out, int64_t n, int mode, void* stream);

/* Row H, the per-slice metrics of the reference's test harness (models/base.py:415-436 with
 * common/metrics/reconstruction_metrics.py:11-25): target, output = the `abs / max` images [n] ->
 * out5 = { MSE, NMSE, maxval = max(output) - min(output), PSNR = 10 log10(maxval^2 / MSE), sum target^2 }, all on the device
 * (out5[2] is the data_range mrx_ssim_loss takes for the SSIM value).  work: mrx_recon_metrics_work_floats() floats, 8-byte aligned. */
int64_t mrx_recon_metrics_work_floats(void);
int mrx_recon_metrics(const float* target, const float* output, float* out5, float* work, int64_t n, void* stream);

/* ---- U-Net without normalisation passes (csrc/unet_fused.hip; reference unet_base/unet_block.py:139-308) -----------------------------
 * Every 3x3 / transposed convolution of the reference U-Net is followed by InstanceNorm2d + LeakyReLU (unet_block.py:251-258, 296-299).
 * These entry points keep such a tensor as (raw convolution output, norm [B,C,2] = per-plane (mean, 1 / sqrt(var + eps))): the producer
 * writes both (statistics from its accumulators), the consumer normalises + activates while it loads.  A NULL norm = a plain tensor.
 *   mrx_unet_conv3x3   y = conv3x3(zero pad, no bias) over the channels of source A then source B (the skip concatenation,
 *                      unet_block.py:224, is never materialised; Cb = 0: one source); w [Cout, Ca + Cb, 3, 3];
 *                      work: mrx_unet_conv3x3_work_floats() floats
 *   mrx_unet_conv_transpose2x2  ConvTranspose2d(k 2, s 2, no bias), w [Cin, Cout, 2, 2], even Cout; out [B,Cout,2H,2W]
 *   mrx_unet_avgpool   avg_pool2d(2) -> plain [planes, H/2, W/2]
 *   mrx_unet_conv1x1   1x1 convolution + bias, Cout <= 4 -> plain
 *   mrx_unet_apply     leaky((x - mean) / std) written out */
int64_t mrx_unet_conv3x3_work_floats(int B, int Cout, int H, int W);
int mrx_unet_conv3x3(const float* xa, const float* na, int Ca, const float* xb, const float* nb, int Cb, const float* w, float* y,
                     float* norm, float* work, int B, int Cout, int H, int W, float eps, float slope, void* stream);
/* mrx_unet_conv3x3 on the fp16 matrix pipe (csrc/unet_f16.hip): every fp32 operand as two fp16 terms scaled by a power of two, three term products
 * per multiply, fp32 accumulation -- fp32-level results (the arithmetic of mrx_rim_layer2_f16) at 1/5 of the matrix cycles of the fp32-input MFMA form.
 *   mrx_unet_conv3x3_pack : w [Cout, Ca + Cb, 3, 3] -> operand pack (mrx_unet_conv3x3_pack_floats(Cout, Ca + Cb) floats, 16-byte aligned);
 *   bound_a / bound_b     : device scalars >= max |x| of a PLAIN source (NULL allowed for a (raw, norm) source: an instance-normalised plane of
 *                           n values is bounded by sqrt(n), no pass over the data needed).
 * MRX_EUNSUP unless MRIDC_AMD_ARITH is f16x2 (the default). */
int64_t mrx_unet_conv3x3_pack_floats(int Cout, int Ctot);
int mrx_unet_conv3x3_pack(const float* w, int Cout, int Ctot, float* packed, void* stream);
int mrx_unet_conv3x3_h(const float* xa, const float* na, const float* bound_a, int Ca, const float* xb, const float* nb, const float* bound_b, int Cb,
                       const float* packed, float* y, float* norm, float* work, int B, int Cout, int H, int W, float eps, float slope, void* stream);
/*   mrx_unet_conv3x3_p16   the same convolution in the reference's `precision: 16` inference arithmetic (base_vn_run.yaml:98, base_unet_run.yaml:96: native AMP =
 *                          torch.autocast(float16) around the forward pass, unet_block.py:250-259 under it): operands rounded to fp16 once (the first term of
 *                          the same pack), exact products, fp32 sums; raw output and statistics stay fp32.  Same arguments and work buffer. */
int mrx_unet_conv3x3_p16(const float* xa, const float* na, const float* bound_a, int Ca, const float* xb, const float* nb, const float* bound_b, int Cb,
                       const float* packed, float* y, float* norm, float* work, int B, int Cout, int H, int W, float eps, float slope, void* stream);
/* ... with the merge of the tile statistics INSIDE the convolution launch (no k_unorm_finalize launch behind it): the last tile of every plane, found by
 * a ticket per plane, writes `norm`.  counters: mrx_unet_conv3x3_hc_ticket_ints(B, Cout) ints (one 128-byte line per plane), ZERO on entry, zero again on exit (one buffer serves every call on a stream; calls
 * that may overlap -- two streams -- need their own).  `norm` equals mrx_unet_conv3x3_h's up to the order of three double-precision sums. */
int64_t mrx_unet_conv3x3_hc_ticket_ints(int B, int Cout);
int mrx_unet_conv3x3_hc(const float* xa, const float* na, const float* bound_a, int Ca, const float* xb, const float* nb, const float* bound_b, int Cb,
                        const float* packed, float* y, float* norm, float* work, int* counters, int B, int Cout, int H, int W, float eps, float slope,
                        void* stream);
/* y = act(conv3x3(x, dilation 1 | 2, zero | replicate padding) + bias), any channel counts, on the same two-term fp16 kernel (ConvNonlinear,
 * rim/conv_layers.py:121-123, for layers the 64-channel kernels do not cover: qRIM's 128 -> 128, DIDN ...).  bound: device scalar >= max |x|;
 * packed: mrx_unet_conv3x3_pack(w [Cout, Cin, 3, 3]); x != y. */
int mrx_conv3x3_h_supported(int Cin, int Cout, int k, int dil);
int mrx_conv3x3_h(const float* x, const float* bound, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int dil,
                  int pad_mode, int act, float slope, void* stream);
/*   mrx_conv3x3_p16   mrx_conv3x3_h in the reference's `precision: 16` inference arithmetic (base_qcirim_run.yaml:204 ...: torch.autocast(float16) around forward;
 *                     conv_layers.py:121-123 under it): operands rounded to fp16 once (the first term of the same pack), fp32 sums, fp32 result */
int mrx_conv3x3_p16(const float* x, const float* bound, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int dil,
                  int pad_mode, int act, float slope, void* stream);
int64_t mrx_unet_conv_transpose2x2_work_floats(int B, int Cout, int H, int W);
int mrx_unet_conv_transpose2x2(const float* x, const float* nrm, const float* w, float* out, float* norm, float* work, int B, int Cin, int Cout,
                      int H, int W, float eps, float slope, void* stream);
int mrx_unet_avgpool(const float* x, const float* nrm, float* out, int64_t planes, int H, int W, float slope, void* stream);
int mrx_unet_apply(const float* x, const float* nrm, float* out, int64_t planes, int64_t HW, float slope, void* stream);
int mrx_unet_conv1x1(const float* x, const float* nrm, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int64_t HW,
                     float slope, void* stream);
/* NormUnet head and tail on complex-last tensors (unet_block.py:46-136, norm_groups = 2):
 *   mrx_unet_cnorm_pad        x [B,c,H,W,2] -> out [B,2c,H+top+bottom,W+left+right] = pad(norm(complex_to_chan_dim(x))) in three launches (two
 *                             statistics passes, one normalise + permute + pad pass); mean, std [B,2]; work: mrx_unet_cnorm_work_floats(B)
 *   mrx_unet_conv1x1_cunnorm  the closing 1x1 convolution of a (raw, norm) tensor [B,Cin,OH,OW] into 2c <= 4 channels, written as
 *                             chan_complex_to_last_dim(unnorm(unpad(.))): out [B,c,H,W,2] */
int64_t mrx_unet_cnorm_work_floats(int B);
int mrx_unet_cnorm_pad(const float* x, float* out, float* mean, float* std_, float* work, int B, int c, int H, int W, int top, int bottom,
                       int left, int right, void* stream);
int mrx_unet_conv1x1_cunnorm(const float* x, const float* nrm, const float* w, const float* bias, const float* mean, const float* std_,
                             float* out, int B, int Cin, int c, int OH, int OW, int top, int left, int H, int W, float slope, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRIDC_AMD_H */
