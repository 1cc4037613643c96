/*
 * fsc_hip.h -- C ABI of libfsc_hip.so, the MI355X (gfx950) kernels behind the
 * freesound-classification audio-tagging hot path.
 *
 * The reference (ex4sperans/freesound-classification) is pure Python on stock ATen ops and
 * has no FFI of its own; each entry point below cites the reference call site whose device
 * work it replaces (paths relative to the reference root).  INTEGRATION.md shows the ctypes
 * binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless the name ends in _host;
 *   - activations are NCHW (the reference's layout); the 1-d model uses H == 1;
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*);
 *   - the caller owns every buffer, including workspaces sized by the *_workspace_bytes /
 *     *_floats queries; no entry point allocates device memory, and the library keeps no device-side state between calls
 *     (the one piece of state a call can leave behind -- the arrival counters of FSC_BN_TICKETS -- lives in the caller's
 *     workspace and is the caller's to vouch for);
 *   - return value 0 = success; anything else is an error whose text
 *     fsc_last_error_string() returns.  No entry point synchronises the device.
 */
#ifndef FSC_HIP_H
#define FSC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fsc_stream_t; /* hipStream_t */

int fsc_version(void);
const char* fsc_last_error_string(void);

/* ------------------------------------------------------------------ front-end (K1-K5)
 * ops/utils.py:110-127 (torch.stft + magnitude), networks/classifiers.py:565-582
 * (mel conv1d, log(x+1e-4), frequency-encoding channel). */

/* floats needed for the window + twiddle tables of one n_fft (power of two, 64..4096) */
size_t fsc_frontend_table_floats(int n_fft);
/* fills tables with the periodic Hann window and exp(-2*pi*i*m/n_fft), computed in fp64 */
int fsc_frontend_tables_init(float* tables, int n_fft, fsc_stream_t stream);

/* waveform (N, T) [row stride wave_stride] -> log-mel.
 * out + n*out_n_stride holds clip n's (n_mel, frames) plane; if freq_channel != 0 a second
 * plane with linspace(-1, 1, n_mel) broadcast over frames follows it (classifiers.py:553-561).
 * Banded filterbank: row m uses weights mel_w[j*n_mel + m], j < mel_len[m], against
 * magnitude bins mel_start[m] + j.  frames = 1 + T / hop. */
int fsc_frontend_logmel_fwd(const float* wave, int n, int t, long wave_stride,
                            int n_fft, int hop, const float* tables,
                            const int* mel_start, const int* mel_len, const float* mel_w,
                            int n_mel, int max_band, float log_eps,
                            float* out, long out_n_stride, int freq_channel,
                            fsc_stream_t stream);

/* waveform -> |STFT| (apply_log == 0, compute_torch_stft itself) or log(|STFT| + eps)
 * (apply_log != 0, classifiers.py:184-185 / :571-572).  out is (N, n_fft/2+1, frames). */
int fsc_frontend_stft_fwd(const float* wave, int n, int t, long wave_stride,
                          int n_fft, int hop, const float* tables, int apply_log,
                          float log_eps, float* out, long out_n_stride, int freq_channel,
                          fsc_stream_t stream);

/* ------------------------------------------------------------------ convolution (K7, K10)
 * nn.Conv2d 3x3 pad 1 / 1x1 (classifiers.py:526-531, 77-81) and nn.Conv1d k3 / k1
 * (classifiers.py:149-154, 42-46; H == 1, kh == 1).  fp32 in, fp32 out, fp32 accumulation; the products run on
 * v_mfma_f32_16x16x32_f16 through a two-limb fp16 split with exact power-of-two operand scaling (default), on
 * v_mfma_f32_16x16x32_bf16 through an exact three-limb bf16 split, or on v_mfma_f32_16x16x4_f32
 * (fsc_conv_desc.arith), stem layers (c_in <= 4) on the vector ALUs.  Weights are re-packed per use by
 * fsc_conv_pack_weights; the packed format is private to the library and depends on shape and arithmetic mode. */

#define FSC_ARITH_DEFAULT (-1)
typedef struct {
    int n, c_in, c_out, h, w; /* output spatial size == input spatial size (stride 1, same pad) */
    int kh, kw;               /* (3,3), (1,1), (1,3) */
    int arith;                /* arithmetic of THIS call: 0, 1, 3, 6, 9 (below) or FSC_ARITH_DEFAULT */
} fsc_conv_desc;

/* floats of packed-weight workspace for one direction (fwd or dgrad) */
size_t fsc_conv_packed_floats(const fsc_conv_desc* d, int dgrad);
/* weight (c_out, c_in, kh, kw) -> packed [tap][k][m]; dgrad != 0 packs the flipped transpose */
int fsc_conv_pack_weights(const fsc_conv_desc* d, const float* weight, int dgrad,
                          float* packed, fsc_stream_t stream);
/* `count` fsc_conv_pack_weights calls in ceil(count / 48) launches -- the weights of a whole training step up front (job i:
 * descs[i], weights[i], dgrad[i] -> packed[i] of fsc_conv_packed_floats(&descs[i], dgrad[i]) floats; all four are HOST arrays).
 * For the tilings whose fragments need no operand scale (arith 1, 6, 9 on the matrix-core path):
 * fsc_conv_pack_weights_multi_supported says which; any other job fails the call. */
int fsc_conv_pack_weights_multi_supported(const fsc_conv_desc* d, int dgrad);
int fsc_conv_pack_weights_multi(int count, const fsc_conv_desc* descs, const float* const* weights,
                                const int* dgrad, float* const* packed, fsc_stream_t stream);
/* out = conv(in) + bias (accumulate == 0) or out += conv(in) + bias (accumulate != 0).
 * With dgrad != 0: in has c_out channels, out has c_in channels, bias must be NULL, and
 * `packed` must come from fsc_conv_pack_weights(dgrad=1). */
int fsc_conv_fwd(const fsc_conv_desc* d, const float* in, const float* packed,
                 const float* bias, int dgrad, int accumulate, float* out,
                 const float* in_amax, fsc_stream_t stream);
/* Largest magnitude of a tensor.  The scaled split-fp16 arithmetic (arith 3) takes max |x| of each
 * activation / gradient operand as an `*_amax` device buffer of FSC_AMAX_FLOATS non-negative floats whose
 * maximum is the value (many slots so that the thousands of workgroups of a producer do not serialise on one
 * address): from this function, or from the producers that report it for free while they write the tensor
 * (fsc_bn_act_fwd / fsc_bn_act_bwd `*_amax` outputs).  A value larger than the true maximum is safe (it costs
 * low-order bits); a smaller one may overflow fp16.  `*_amax` may be NULL in the other arithmetic modes. */
#define FSC_AMAX_FLOATS 512
int fsc_amax(const float* x, long n, float* out, fsc_stream_t stream);
/* Stem layer (3x3, c_in <= 2) fused with the 2x2 max-pool that follows it (classifiers.py:526-532): writes the
 * pooled tensor (N, c_out, H/2, W/2) and the uint8 window indices of fsc_maxpool_fwd; the full-resolution conv
 * output is never materialised.  `packed` from fsc_conv_pack_weights(dgrad = 0).  fsc_conv_pool_supported
 * returns 1 for the shapes this entry point accepts. */
int fsc_conv_pool_supported(const fsc_conv_desc* d);
int fsc_conv_pool_fwd(const fsc_conv_desc* d, const float* in, const float* packed, const float* bias,
                      float* pooled, uint8_t* idx, fsc_stream_t stream);
/* Weight gradient of the stem layer (3x3, c_in <= 2) straight from the gradient at the POOLED resolution `dpooled`
 * (N, c_out, H/2, W/2) and the window indices of fsc_conv_pool_fwd / fsc_maxpool_fwd -- the un-pooled gradient (one non-zero
 * per 2x2 window) is never materialised.  partial: fsc_conv_stem_wgrad_pooled_blocks(d) x c_out x 32 floats; per channel
 * [c_in * 9 weight-gradient sums (ci, ty, tx) | at 18: 8 border sums of the un-pooled gradient in the order of
 * fsc_plane_border_sums | pad]; fsc_conv_stem_grads_finish sums the blocks.
 * in_mean / in_invstd (per input channel, both or neither): `in` is then the INPUT of the BatchNorm in front of the stem
 * (classifiers.py:524) and the sums are taken against xhat = (in - mean) * invstd (zero outside the image) instead of `in`. */
size_t fsc_conv_stem_wgrad_pooled_blocks(const fsc_conv_desc* d);
int fsc_conv_stem_wgrad_pooled(const fsc_conv_desc* d, const float* in, const float* in_mean, const float* in_invstd,
                               const float* dpooled, const uint8_t* pool_idx, float* partial, fsc_stream_t stream);
/* Finishes fsc_conv_stem_wgrad_pooled: dweight (c_out, c_in, 3, 3) from the partial rows (which it uses as scratch), optionally
 * the 8 border sums per channel (c_out, 8), and optionally the parameter gradients (c_in each) of the BatchNorm in front of the
 * stem WITHOUT the stem's input gradient (classifiers.py:524-531).  chan_sum[c_out] = per-channel total of the
 * un-pooled gradient (= of dpooled).  xhat = 1: the partials were taken against xhat; dweight = gamma dW' + beta T,
 * dgamma = sum w dW', dbeta = sum w T (T[co][tap]: total minus the border sums the tap excludes) -- no division.  xhat = 0:
 * the partials were taken against the conv input a = gamma xhat + beta; dgamma = (sum w dW - beta dbeta) / gamma, singular at
 * gamma = 0 (callers check fsc_absmin(gamma) first). */
int fsc_conv_stem_grads_finish(const fsc_conv_desc* d, float* partial, const float* weight, const float* chan_sum,
                               const float* gamma, const float* beta, int xhat, float* dweight, float* borders, float* dgamma,
                               float* dbeta, fsc_stream_t stream);
/* human-readable tiling chosen for this shape (mode 0 fwd, 1 dgrad, 2 wgrad): kernel
 * instantiation, pixel box, grid, LDS bytes.  For logs, DESIGN.md tables and profiles. */
int fsc_conv_plan_describe(const fsc_conv_desc* d, int mode, char* buf, size_t buf_len);
/* Arithmetic of the convolution kernels: a per-call property, `fsc_conv_desc.arith` (the library keeps no
 * mutable mode).  0: native fp32 MFMA (v_mfma_f32_16x16x4_f32).
 * 3: every fp32 operand x is scaled by a power of two s (from the tensor's largest magnitude, see fsc_amax) and
 * split into two fp16 limbs h = rne(x*s), l = rne(x*s - h) (|x*s - h - l| <= 2^-24 |x*s|); the product is formed
 * from the 3 limb products hh + hl + lh on v_mfma_f32_16x16x32_f16 with fp32 accumulation and unscaled exactly
 * (dropped term ll <= 2^-22 |a*b|).  An operand holding +-Inf (declared maximum Inf) makes every output of the
 * call non-finite; NaN elements propagate to the outputs they touch.
 * 1: plain bf16 arithmetic (BASELINE.json configs[2]): every operand is rounded to ONE bf16 value (round to nearest
 * even), products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- the usual mixed-precision recipe with fp32
 * master weights; relative operand error 2^-9, NOT fp32-accurate.
 * 6 / 9: every fp32 operand is split exactly into three bf16 limbs and the product is formed from 6 / 9
 * limb products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (9 = all terms: the product is exact
 * before accumulation; 6 drops terms below 2^-23 |a*b|).
 * 10 ("f16x6", pre-split L16 tensors only -- fsc_conv_l16_*; the fp32-input entry points serve it as 9): every operand is scaled
 * as in mode 3 and split into THREE fp16 limbs, x*s = h + m + l (11 + 11 + 11 significand bits: EXACT for every element down to
 * 2^-16 of the tensor's declared maximum, an absolute error <= 2^-39 of that maximum below), and the product is formed from the
 * six limb products hh + hm + mh + mm + hl + lh on v_mfma_f32_16x16x32_f16 (dropped: ml + lm + ll <= 2^-32 |a*b|), fp32
 * accumulation, unscaled exactly.  For an fp32 accumulator -- which rounds every partial sum at 2^-24 -- that is the same
 * arithmetic as mode 9 at two thirds of the matrix work; what it gives up is mode 9's independence of the operands' range.
 * Inputs, outputs and accumulators are fp32
 * in every mode.  FSC_ARITH_DEFAULT selects the process default: 10 (the reference's fp32 nn.Conv2d precision,
 * classifiers.py:526-531, 77-81; mode 3 is the opt-in fast mode with 22-bit products), or the environment variable
 * FSC_CONV_ARITH = f32 | bf16 | f16x3 | bf16x6 | bf16x9 | f16x6, read once.  The packed-weight format depends on the mode:
 * pack with the descriptor (same `arith`) the weights are used with. */
int fsc_conv_default_arith(void);
/* bytes of split-K workspace for the weight gradient */
size_t fsc_conv_wgrad_workspace_bytes(const fsc_conv_desc* d);
/* dweight (c_out, c_in, kh, kw) = sum_pixels dout x in  (overwrites dweight) */
int fsc_conv_wgrad(const fsc_conv_desc* d, const float* in, const float* dout,
                   float* dweight, void* workspace, const float* in_amax, const float* dout_amax,
                   fsc_stream_t stream);
/* fsc_conv_wgrad in two halves, for callers that compute several weight gradients before anyone reads them (the four
 * convolutions of a resnet block, classifiers.py:37-69 / 72-104): fsc_conv_wgrad_partial leaves the split-K slices in `workspace`
 * (fsc_conv_wgrad_workspace_bytes, one workspace per pending gradient), fsc_conv_wgrad_reduce_multi sums the slices of `count`
 * such gradients into their dweight tensors in ceil(count / 16) launches.  descs / workspaces / dweights are HOST arrays.  Same
 * additions in the same order as fsc_conv_wgrad: bit-identical results. */
int fsc_conv_wgrad_partial(const fsc_conv_desc* d, const float* in, const float* dout, void* workspace,
                           const float* in_amax, const float* dout_amax, fsc_stream_t stream);
int fsc_conv_wgrad_reduce_multi(int count, const fsc_conv_desc* descs, const void* const* workspaces,
                                float* const* dweights, fsc_stream_t stream);

/* ---- pre-split activations ("L16" tensors) for the split-fp16 arithmetic (arith 3).
 * The kernels behind fsc_conv_fwd split every fp32 activation into its two fp16 limbs beside the MFMAs, once per
 * tap.  An L16 tensor holds the limbs ready-made, in the layout of the MFMA operand: for a logical (N, C, HW) fp32
 * tensor, half[N][ceil(C/8)][2 limbs][HW][8 channels] (4 bytes per element; pad channels are zero), scaled by the
 * power of two the kernels derive from the tensor's declared maximum `amax` (an FSC_AMAX_FLOATS buffer; the SAME buffer
 * must accompany the tensor to its consumers).  Producers: fsc_l16_pack, and the fused BN / PReLU kernels
 * (fsc_bn_act_fwd_l16 ...).  Consumers: fsc_conv_l16_fwd (forward and input gradient of nn.Conv2d 3x3 / 1x1,
 * classifiers.py:526-531, 77-81).  Results are bit-identical to fsc_conv_fwd with arith 3 on the fp32 tensor. */
size_t fsc_l16_bytes(int n, int c, long hw);
int fsc_l16_pack(const float* x, int n, int c, long hw, const float* amax, void* out_l16, fsc_stream_t stream);
/* (h + l) / scale back to fp32 NCHW (tests, debugging) */
int fsc_l16_unpack(const void* in_l16, int n, int c, long hw, const float* amax, float* x, fsc_stream_t stream);
/* Limb formats.  limbs = 2: the scaled fp16 pairs above (arith 3).  limbs = 3: three EXACT bf16 limbs, x = h + m + l
 * (8 + 8 + 8 significand bits, fp32 exponent range: no scale, `amax` is neither read nor written and may be NULL), layout
 * bf16[N][ceil(C / 8)][3 limbs][HW][8 channels], 6 bytes per element -- the operands of arith 9 (all nine limb products on
 * v_mfma_f32_16x16x32_bf16: each product is the exact product of the fp32 operands, fp32 accumulation; 8 / 6 drop the terms
 * below 2^-32 / 2^-23 |a*b|).  limbs = 4 (FSC_L16_F16X3): three SCALED fp16 limbs, x * s = h + m + l with the scale of the
 * two-limb format (`amax` as there), same layout and size as limbs = 3 -- the operands of arith 10.  Every fsc_conv_l16_* entry
 * point takes the format from fsc_conv_desc.arith (3 -> two fp16 limbs, 9 / 8 / 6 -> three bf16 limbs, 10 -> three fp16 limbs)
 * and the producers from their `limbs` argument (fsc_bn_act_*_limbs) or phase flag. */
#define FSC_L16_F16X3 4
size_t fsc_l16_bytes_limbs(int n, int c, long hw, int limbs);
int fsc_l16_pack_limbs(const float* x, int n, int c, long hw, const float* amax, int limbs, void* out_l16, fsc_stream_t stream);
int fsc_l16_unpack_limbs(const void* in_l16, int n, int c, long hw, const float* amax, int limbs, float* x, fsc_stream_t stream);
/* 1 when fsc_conv_l16_fwd has a tiling for this shape and direction (3x3 / 1x1, >= 32 input and >= 48 output
 * channels, enough work items to fill the chip without split-K) in the descriptor's arithmetic: 3 (two limbs), 9 / 10 (three limbs);
 * FSC_ARITH_DEFAULT resolves to the process default (fsc_conv_default_arith) at every entry point */
int fsc_conv_l16_supported(const fsc_conv_desc* d, int dgrad);
size_t fsc_conv_l16_packed_floats(const fsc_conv_desc* d, int dgrad);
int fsc_conv_l16_pack_weights(const fsc_conv_desc* d, const float* weight, int dgrad, float* packed,
                              fsc_stream_t stream);
/* both directions of one weight in one call (either pointer may be NULL): shares the max |w| pass and the launch */
int fsc_conv_l16_pack_weights_pair(const fsc_conv_desc* d, const float* weight, float* packed_fwd,
                                   float* packed_dgrad, fsc_stream_t stream);
/* The same for `count` weights at once (ceil(count / 8) pairs of launches instead of one pair per weight: the 20 L16 convolutions of
 * a cfg-2 training step re-pack their weights every step).  descs[i] / weights[i] / packed_fwd[i] / packed_dgrad[i] as for the pair
 * call; host arrays, read before the call returns. */
int fsc_conv_l16_pack_weights_multi(int count, const fsc_conv_desc* descs, const float* const* weights, float* const* packed_fwd,
                                    float* const* packed_dgrad, fsc_stream_t stream);
/* out (fp32 NCHW) = conv(in) + bias, or += with accumulate; dgrad as in fsc_conv_fwd */
int fsc_conv_l16_fwd(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed,
                     const float* bias, int dgrad, int accumulate, float* out, fsc_stream_t stream);
int fsc_conv_l16_plan_describe(const fsc_conv_desc* d, int dgrad, char* buf, size_t buf_len);
/* Inference (the L16 arithmetics: 3, 9, 10): convolution whose epilogue applies an eval-mode BatchNorm and PReLU and writes
 * the result as the L16 operand of the NEXT convolution -- nn.Conv2d -> nn.BatchNorm2d (eval) -> nn.PReLU of a residual unit
 * (reference classifiers.py:77-101) in one launch, 6 bytes per element leaving the kernel instead of 4 + 4 + 6 through the separate
 * pass.  y = prelu(fma(conv(x) + bias, scale[c], shift[c]), alpha[c]); scale / shift = fsc_bn_eval_prepare's (both NULL: identity),
 * alpha NULL: no activation.  Same expressions as fsc_conv_l16_fwd followed by fsc_bn_act_fwd_limbs: bit-identical limbs.
 * out_l16: fsc_l16_bytes_limbs(n, c_out, h * w, 2 | 3 | FSC_L16_F16X3) bytes.  arith 3 / 10 (scaled fp16 limbs): `out_amax` (FSC_AMAX_FLOATS floats) is the
 * DECLARED maximum of |y| the limbs are scaled by -- a calibrated bound the caller brings (e.g. the bound the two-pass route derived
 * on an earlier batch, with headroom); elements beyond it saturate, so the kernel max-es the largest |y| it wrote into
 * `seen_max[0]` (atomic; the caller zeroes it before the launch and compares afterwards; may be NULL).  arith 9: no scale, out_amax
 * unused.  _supported: 1 when the shape has an L16 forward tiling in the descriptor's arithmetic. */
int fsc_conv_l16_fwd_act_supported(const fsc_conv_desc* d);
int fsc_conv_l16_fwd_act(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias,
                         const float* scale, const float* shift, const float* alpha, void* out_l16, const float* out_amax,
                         float* seen_max, fsc_stream_t stream);
/* Inference, the entry convolution of a block (classifiers.py:526-534): 3x3 convolution -> MaxPool2d(2) -> eval-mode BatchNorm
 * (scale / shift of fsc_bn_eval_prepare) -> PReLU in ONE launch.  The result (N, c_out, H/2, W/2) leaves as fp32 in `out` (the
 * residual unit adds it back; may be NULL) AND as the three-limb L16 tensor `out_l16` the unit's first convolution reads; the
 * convolution output and the pooled pre-activation are never written.  Same expressions as fsc_conv_l16_pool_fwd followed by
 * fsc_bn_act_fwd_limbs: bit-identical fp32 and limbs.  Arith 9 / 10 only; out_amax / seen_max as fsc_conv_l16_fwd_act. */
int fsc_conv_l16_pool_fwd_act_supported(const fsc_conv_desc* d);
int fsc_conv_l16_pool_fwd_act(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias,
                              const float* scale, const float* shift, const float* alpha, float* out, void* out_l16,
                              const float* out_amax, float* seen_max, fsc_stream_t stream);
/* Forward 3x3 convolution fused with the MaxPool2d(2) behind it (classifiers.py:526-532, the blocks after the stem): writes
 * the pooled tensor (N, c_out, H/2, W/2) and the uint8 window indices of fsc_maxpool_fwd (same first-maximum / NaN rule);
 * the full-resolution output is never materialised.  `packed` from fsc_conv_l16_pack_weights(dgrad = 0). */
int fsc_conv_l16_pool_supported(const fsc_conv_desc* d);
int fsc_conv_l16_pool_fwd(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed,
                          const float* bias, float* pooled, uint8_t* idx, fsc_stream_t stream);
/* Weight gradient from L16 operands: dweight (c_out, c_in, kh, kw) = sum over pixels dout x in (overwrites dweight), 3x3 and
 * 1x1, same arithmetic as fsc_conv_wgrad with arith 3.  in_l16 is the (N, c_in, H, W) input of the convolution, dout_l16 the
 * (N, c_out, H, W) gradient of its output, each with the amax buffer its scale derives from. */
int fsc_conv_l16_wgrad_supported(const fsc_conv_desc* d);
size_t fsc_conv_l16_wgrad_workspace_bytes(const fsc_conv_desc* d);
int fsc_conv_l16_wgrad(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const void* dout_l16,
                       const float* dout_amax, float* dweight, void* workspace, fsc_stream_t stream);
int fsc_conv_l16_wgrad_plan_describe(const fsc_conv_desc* d, char* buf, size_t buf_len);

/* ------------------------------------------------------------------ batch norm + PReLU (K6, K9, K11)
 * nn.BatchNorm2d/1d (train and eval) fused with the following per-channel PReLU and the
 * residual add of ResnetBlock(2d) (classifiers.py:524,533-534; 37-104; 543-546).
 * x is (N, C, HW); HW == 1 covers BatchNorm1d on (N, C). */

size_t fsc_bn_workspace_bytes(int c);
/* phase | FSC_BN_TICKETS (fsc_bn_train_stats, fsc_bn_act_bwd, fsc_bn_act_bwd_unpool; single replica, i.e. (phase & 3) == 0): the
 * reduce pass also finalises -- the workgroup that finishes a channel LAST folds the channel's partial sums -- so the call is one
 * launch fewer.  It counts arrivals in the `c` 32-bit words at byte fsc_bn_workspace_ticket_offset(c) of `workspace`.  Contract:
 *   - those words are ZERO when the call is enqueued; the call leaves them zero again (the last workgroup of a channel resets its
 *     word), so a workspace that was zeroed once after allocation (hipMemsetAsync, or fsc_bn_workspace_reset) can be passed call
 *     after call;
 *   - a workspace belongs to ONE call at a time: calls that may overlap (different streams) need different workspaces -- nothing
 *     is shared between calls except what the caller passes;
 *   - if a launch of the call fails (a sticky device error), zero the words again before the next use.
 * Without the flag the workspace needs no initialisation and a separate finalisation kernel runs (same arithmetic, same order of
 * additions: bit-identical results). */
#define FSC_BN_TICKETS 32
/* fsc_bn_act_bwd / fsc_bn_act_bwd_unpool, OR-ed into `phase`: the L16 output (dx_l16 / dc_l16) is written as three exact bf16
 * limbs (fsc_l16_bytes_limbs(.., 3) bytes; the operands of arith 9) instead of two scaled fp16 limbs; the `*_amax` bound is then
 * optional (written when given). */
#define FSC_BN_L16_LIMBS3 64
/* the same for three SCALED fp16 limbs (fsc_l16_bytes_limbs(.., FSC_L16_F16X3); the operands of arith 10): `*_amax` is required
 * as for two limbs.  Excludes FSC_BN_L16_LIMBS3. */
#define FSC_BN_L16_F16X3 128
size_t fsc_bn_workspace_ticket_offset(int c);
int fsc_bn_workspace_reset(void* workspace, int c, fsc_stream_t stream);
/* Cross-replica batch statistics (SyncBN for data parallelism; the reference is single-process so its
 * BatchNorm at classifiers.py:524,533,78-82,543,545 always sees the whole batch).  The three training entry points
 * below take `sync` -- FSC_BN_SYNC_DOUBLES(c) doubles, per channel [sum a, sum b, count, 0] -- and `phase`:
 *   0  single replica, `sync` unused (may be NULL);
 *   1  reduce the local batch, write the local sums to `sync`, return (no outputs besides the parameter gradients: a backward
 *      call may pass dx = NULL -- also the way to get ONLY the parameter gradients when the input needs no gradient);
 *      the CALLER then sum-all-reduces `sync` over the replicas (RCCL; one small message per layer);
 *   2  finish from the reduced `sync` (statistics / input gradient use the global sums and count).  Pass the SAME `workspace` as in
 *      phase 1: the backward finalisation re-reads the replica's own partial sums from it (the closed-form per-channel sum of dx).
 * forward: [sum x, sum x^2, count] about zero; backward: [sum dz, sum dz*xhat, count].  dgamma / dbeta / dalpha are
 * always LOCAL sums (the gradient all-reduce adds the replicas). */
#define FSC_BN_SYNC_DOUBLES(c) (4 * (size_t)(c))
/* train: batch statistics -> scale/shift (scale = gamma*invstd, shift = beta - mean*scale),
 * save_mean / save_invstd for backward, running stats updated with `momentum`
 * (running_var gets the unbiased estimate), exactly one update per call. */
int fsc_bn_train_stats(const float* x, int n, int c, long hw, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, float* save_mean, float* save_invstd,
                       float* scale, float* shift, void* workspace, double* sync, int phase,
                       float* x_minmax, fsc_stream_t stream);
/* Statistics AND apply pass of one training-mode unit in ONE launch (classifiers.py:78-82, 92-93, 96-97, 100-104: BatchNorm in
 * training mode, [+ residual], PReLU) where a channel's n x hw values fit the registers of one workgroup (single replica, planes
 * of 2 ... 1024 values, <= 8 16-byte quads per thread of 256 or 1024 threads: the 1-d model's blocks from 215 frames down at batch
 * 128 -- tensors of <= 14 MB on which a launch costs more than its bytes): all loads in flight at once, every reduction of the
 * unit through one LDS exchange, every thread finalises, the apply pass is arithmetic on registers.  Writes everything
 * fsc_bn_train_stats (phase 0) writes, then y = act(x * scale + shift [+ residual]) with fsc_bn_act_fwd's element arithmetic;
 * against the two-launch route the statistics agree to an ulp of invstd (same additions in the same order where that route runs
 * one split; the products are contracted into other fused multiply-adds).  `_supported` says whether (n, c, hw) is such a shape;
 * the call fails otherwise.  fsc_bn_act_bwd / fsc_bn_act_bwd_unpool (single row, (1, 2) windows) make the same decision on their
 * own -- no L16 output, no amax, single replica, FSC_BN_TICKETS: reduce, finalise and apply in one launch -- nothing to ask for.
 * gmax / gmax_idx (may be NULL; `_supported` bit 1: planes of <= 64 lanes, hw <= ~250): the (n, c) global max of y and its
 * position under fsc_global_maxpool_fwd's rule -- the block output's second reader (classifiers.py:586-590) served by the launch
 * that writes it.  `_supported`: 0 no; 1 yes; 3 yes, and the global max can be asked for. */
int fsc_bn_train_act_fwd_supported(int n, int c, long hw);
int fsc_bn_train_act_fwd(const float* x, const float* residual, int n, int c, long hw, const float* gamma, const float* beta,
                         float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                         float* save_invstd, float* scale, float* shift, float* x_minmax, const float* alpha, float* y,
                         float* gmax, int* gmax_idx, fsc_stream_t stream);
/* x_minmax (2*C floats, may be NULL): per channel [min x, max x] of the local batch -- what fsc_bn_act_fwd needs to
 * bound its output before it writes an L16 tensor.
 * phase | FSC_BN_STATS_FOLDED: the reduction over x was done by the kernel that WROTE x (fsc_bn_act_fwd_rec +
 * fsc_bn_records_fold into this `workspace`): no statistics pass, only the finalisation (x is still read for the pivot). */
#define FSC_BN_STATS_FOLDED 4
/* phase | FSC_BN_STATS_PIVOT_RM (with FSC_BN_STATS_FOLDED): the folded sums are about `running_mean` as it is BEFORE this call
 * (0 when running_mean is NULL) instead of the channel's first element: the convention of the STATS convolutions below.
 * Those sums come from fp32 lane accumulators: when the estimated batch mean lies more than 4 standard deviations from the
 * pivot (or the variance estimate is not positive) the finalisation re-reduces that channel of x about the mean estimate in
 * fp64 (one workgroup per channel; never in steady-state training, where the running mean tracks the batch mean). */
#define FSC_BN_STATS_PIVOT_RM 8
/* phase | FSC_BN_STATS_MINMAX_ONLY (phase 0): only x_minmax is written (everything else may be NULL) -- inference, where the
 * BatchNorm runs on its running statistics but fsc_bn_act_fwd still needs the range of x to write an L16 tensor. */
#define FSC_BN_STATS_MINMAX_ONLY 16
/* The last unit of a block (classifiers.py:102-104: bn3 + residual + PReLU) writes a tensor that the next block's input
 * BatchNorm (classifiers.py:524) and the hierarchical head's global max-pool (classifiers.py:586-590) read again at once.
 * fsc_bn_act_fwd_rec = fsc_bn_act_fwd (fp32 y, hw > 1, no amax) that also leaves one record per (plane, slice) --
 * fsc_bn_records_bytes(n, c, hw) bytes -- with the pivot-shifted sums, min / max and global-max key of what it wrote;
 * fsc_bn_records_fold turns them into split 0 of a BatchNorm workspace (`stats_workspace`, fsc_bn_workspace_bytes(c);
 * pass it to fsc_bn_train_stats with FSC_BN_STATS_FOLDED; may be NULL) and / or the (n, c) global max + argmax of y
 * (same first-maximum / NaN rule as fsc_global_maxpool_fwd; may be NULL).  Bit-identical statistics inputs are NOT
 * promised (the summation order differs from the statistics pass); the pooled values and indices are identical. */
size_t fsc_bn_records_bytes(int n, int c, long hw);
int fsc_bn_act_fwd_rec(const float* x, const float* residual, const float* scale, const float* shift,
                       const float* alpha, float* y, int n, int c, long hw, void* records, fsc_stream_t stream);
int fsc_bn_records_fold(const void* records, const float* y, int n, int c, long hw, void* stats_workspace,
                        float* gmax, int* gmax_idx, fsc_stream_t stream);
/* Statistics from the convolution that PRODUCES the BatchNorm input (classifiers.py:78-101, 524-533: every convolution of a
 * block feeds a BatchNorm): fsc_conv_l16_fwd_stats / fsc_conv_l16_pool_fwd_stats are fsc_conv_l16_fwd (forward) /
 * fsc_conv_l16_pool_fwd whose epilogue also accumulates sum (y - pivot), sum (y - pivot)^2, min y, max y per output channel of what
 * it stores (the pooled values for the pool variant); pivot = stat_pivot[channel] (pass the BatchNorm's running_mean; NULL = 0).
 * fsc_conv_l16_stats_layout -> 1 and out3[4] = {workers, channel blocks, channels per block, worker order} when the shape has such a kernel;
 * `stat_rec` holds workers * 8 * channels-per-block records of 16 bytes.  fsc_bn_records_fold_conv folds them into split 0 of a
 * BatchNorm workspace; then fsc_bn_train_stats(..., phase | FSC_BN_STATS_FOLDED | FSC_BN_STATS_PIVOT_RM, ...) with the SAME
 * running_mean pointer only finalises.  min / max are exact; mean / variance agree with the separate pass to rounding. */
/* The same statistics epilogue on the fp32-input ring kernels in plain bf16 arithmetic (arith 1) on 1-d rows -- the 1-d model's
 * early blocks (classifiers.py:147-163, 78-101: every Conv1d of a ResnetBlock feeds a BatchNorm): fsc_conv_fwd (forward, no
 * accumulation) that also leaves the record format of fsc_conv_l16_stats_layout for fsc_bn_train_stats_conv (sums about `stat_pivot`
 * = that BatchNorm's running mean, or NULL = 0).  `_layout` returns 0 where the layer has no such kernel (other arithmetics, 2-d
 * planes, the small-layer kernels of conv_s1d.hip, split-K plans).  out4[3] = 2: these records are TRANSPOSED -- [channel][slot],
 * out4[0] * 8 * out4[1] * out4[2] float4 in all -- so that the fold reads a channel's slots contiguously. */
int fsc_conv_fwd_stats_layout(const fsc_conv_desc* d, int* out4);
/* ... and of a k3 Conv1d followed by MaxPool1d(2) (the entry convolution of a block of the 1-d model, classifiers.py:149-155) in one
 * launch: `pooled` (n, c_out, 1, w / 2) and `pool_idx` (bytes, same shape) are what fsc_maxpool_fwd gives on the convolution's
 * output -- which is never written --, the records are the statistics of the POOLED tensor (same layout query). */
int fsc_conv_fwd_pool_stats_supported(const fsc_conv_desc* d);
int fsc_conv_fwd_pool_stats(const fsc_conv_desc* d, const float* in, const float* packed, const float* bias, float* pooled,
                            uint8_t* pool_idx, const float* stat_pivot, void* stat_rec, fsc_stream_t stream);
int fsc_conv_fwd_stats(const fsc_conv_desc* d, const float* in, const float* packed, const float* bias, float* out,
                       const float* stat_pivot, void* stat_rec, fsc_stream_t stream);
/* Shader clock (MHz) the chip ran the LAST L16 convolution launch at (which = 0: fsc_conv_l16_fwd family on two-limb
 * operands, 1: fsc_conv_l16_wgrad, 2: fsc_conv_l16_fwd family on three-limb operands):
 * workgroup 0 stamps the shader-cycle counter and the constant 100 MHz reference counter at both ends of the kernel.  The
 * MFMA-bound launches run well below the 2.4 GHz the peak figures assume (power limit); bench.py reports it next to `roofline`.
 * Synchronises the device: a measurement aid, not for the training path. */
int fsc_conv_l16_last_clock(int which, double* shader_mhz);
int fsc_conv_l16_stats_layout(const fsc_conv_desc* d, int pool, int* out3);
int fsc_conv_l16_fwd_stats(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed,
                           const float* bias, float* out, const float* stat_pivot, void* stat_rec, fsc_stream_t stream);
int fsc_conv_l16_pool_fwd_stats(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed,
                                const float* bias, float* pooled, uint8_t* idx, const float* stat_pivot, void* stat_rec,
                                fsc_stream_t stream);
int fsc_bn_records_fold_conv(const void* records, int workers, int blocks, int co_blk, int order, int c,
                             void* stats_workspace, fsc_stream_t stream);
/* fsc_bn_records_fold_conv + fsc_bn_train_stats(phase 0 | FSC_BN_STATS_FOLDED | FSC_BN_STATS_PIVOT_RM) in ONE launch (single replica,
 * training): the workgroup of a channel folds the records of the STATS convolution that wrote x and finalises (pivot = running_mean
 * as it is before this call, 0 when NULL; the far-pivot re-reduction included).  No workspace.  save_mean == NULL: only x_minmax is written (inference: the
 * BatchNorm runs on its running statistics, the L16 producer still needs the range of x). */
int fsc_bn_train_stats_conv(const void* records, int workers, int blocks, int co_blk, int order, const float* x, int n, int c, long hw,
                            const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, float* save_mean, float* save_invstd, float* scale, float* shift, float* x_minmax,
                            fsc_stream_t stream);
/* eval: scale/shift from the running statistics */
int fsc_bn_eval_prepare(int c, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* scale, float* shift,
                        fsc_stream_t stream);
/* y = act(x*scale + shift [+ residual]); act = PReLU(alpha[c]) if alpha != NULL else identity.
 * y_amax (FSC_AMAX_FLOATS floats, may be NULL) receives max |y| -- the operand scale of the split-fp16 conv kernels.
 * y_l16 != NULL: y is (also) written as an L16 tensor (fsc_l16_bytes(n, c, hw) bytes) scaled by y_amax, which is
 * then computed up front from x_minmax (fsc_bn_train_stats; exact: affine + PReLU take their extremes at the ends of
 * each channel's range); needs x_minmax and y_amax, no residual, hw > 1; the fp32 `y` may be NULL then. */
int fsc_bn_act_fwd(const float* x, const float* residual, const float* scale,
                   const float* shift, const float* alpha, float* y, int n, int c, long hw,
                   float* y_amax, const float* x_minmax, void* y_l16, fsc_stream_t stream);
/* fsc_bn_act_fwd with the limb format of the L16 output spelled out: limbs = 2 (as above), 3 (exact bf16 triples: no scale, so
 * x_minmax and y_amax are not needed and may be NULL) or 4 (FSC_L16_F16X3, scaled fp16 triples: x_minmax and y_amax as for 2). */
int fsc_bn_act_fwd_limbs(const float* x, const float* residual, const float* scale,
                         const float* shift, const float* alpha, float* y, int n, int c, long hw,
                         float* y_amax, const float* x_minmax, void* y_l16, int limbs, fsc_stream_t stream);
/* backward of the fused unit.  Upstream gradient = dy (may be NULL) plus, when the output also
 * feeds a global max-pool head, gmax_dy[n*c] scattered at position gmax_idx[n*c] of each plane
 * (both NULL otherwise).  Outputs: dx; dresidual (may be NULL; equals the gradient at the
 * pre-activation); dgamma, dbeta, dalpha (C each; may be NULL); dx_chan_sum (C, may be NULL) =
 * per-channel sum of THIS call's dx = bias gradient of the convolution that produced x, evaluated in closed form from
 * the reduce pass's fp64 sums (k (sum dz - count c1 - c2 sum xhat) for dx = k (dz - c1 - xhat c2)): the sum of the stored
 * dx up to their own rounding, without one atomic per plane; dx_amax (FSC_AMAX_FLOATS floats, may be NULL) = max |dx|.
 * dx_l16 != NULL: dx is (also) written as an L16 tensor scaled by dx_amax, which then receives an upper bound of
 * max |dx| derived in the reduce pass (|k| (max |dz| + |mean dz| + max |xhat| |mean dz xhat|) per channel; an
 * over-estimate is safe); needs dx_amax and hw > 1; the fp32 `dx` may be NULL then. */
int fsc_bn_act_bwd(const float* dy, const float* gmax_dy, const int* gmax_idx, const float* x,
                   const float* residual, const float* save_mean, const float* save_invstd,
                   const float* gamma, const float* beta, const float* alpha, float* dx,
                   float* dresidual, float* dgamma, float* dbeta, float* dalpha,
                   float* dx_chan_sum, int n, int c, long hw, void* workspace,
                   float* dx_amax, double* sync, int phase, void* dx_l16, fsc_stream_t stream);

/* Same backward for the unit that directly follows a max-pool (BN -> PReLU on the pooled tensor x,
 * classifiers.py:532-534), fused with the pool's backward: writes dc (N, C, h, w), the gradient of
 * the UN-pooled tensor -- each window's gradient at its arg-max (pool_idx from fsc_maxpool_fwd),
 * zeros elsewhere -- instead of dx at the pooled resolution.  ph as in fsc_maxpool_fwd.
 * dc_l16 != NULL: dc is (also) written as an L16 tensor of the un-pooled shape, scaled by the bound dc_amax receives
 * (as in fsc_bn_act_bwd); the fp32 `dc` may be NULL then. */
int fsc_bn_act_bwd_unpool(const float* dy, const float* x, const float* save_mean,
                          const float* save_invstd, const float* gamma, const float* beta,
                          const float* alpha, const uint8_t* pool_idx, float* dc, float* dgamma,
                          float* dbeta, float* dalpha, float* dx_chan_sum, int n, int c, int h,
                          int w, int ph, void* workspace, float* dc_amax, double* sync, int phase,
                          void* dc_l16, fsc_stream_t stream);

/* ------------------------------------------------------------------ pooling (K8, K12)
 * nn.MaxPool2d(2,2) / nn.MaxPool1d(2,2), floor mode (classifiers.py:532, 155);
 * nn.AdaptiveMaxPool2d(1) / 1d(1) (classifiers.py:540, 163). ph is 2 (2-d) or 1 (1-d). */
int fsc_maxpool_fwd(const float* x, float* y, uint8_t* idx, int nc, int h, int w, int ph,
                    fsc_stream_t stream);
int fsc_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int nc, int h, int w,
                    int ph, fsc_stream_t stream);
int fsc_global_maxpool_fwd(const float* x, float* y, int* idx, int nc, long hw,
                           fsc_stream_t stream);
/* dx = (dx_in ? dx_in : 0) + scatter(dy at idx) */
int fsc_global_maxpool_bwd(const float* dy, const int* idx, const float* dx_in, float* dx,
                           int nc, long hw, fsc_stream_t stream);

/* ------------------------------------------------------------------ classifier head (K13)
 * nn.Linear (classifiers.py:544,548), nn.Dropout (:547). */
/* y (m, n_out) = x (m, k) . w (n_out, k)^T + bias */
int fsc_linear_fwd(const float* x, const float* w, const float* bias, float* y, int m, int k,
                   int n_out, fsc_stream_t stream);
/* dx (m,k) = dy . w ; dw (n_out,k) = dy^T . x ; dbias (n_out) = column sums of dy.
 * Any of dx / dw / dbias may be NULL. */
int fsc_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw,
                   float* dbias, int m, int k, int n_out, fsc_stream_t stream);
/* inverted dropout with a counter-based generator; mask (uint8) is written for backward */
int fsc_dropout_fwd(const float* x, float* y, uint8_t* mask, long count, float p,
                    uint64_t seed, uint64_t offset, fsc_stream_t stream);
int fsc_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, long count, float p,
                    fsc_stream_t stream);

/* ------------------------------------------------------------------ losses (K14, K15)
 * networks/losses.py:47-58 (lsep_loss), :19-22 (binary_cross_entropy),
 * classifiers.py:687 (sigmoid). */
/* loss[n] = log(1 + sum_{i,j: t[n,j] < t[n,i]} exp(s[n,j] - s[n,i])) */
int fsc_lsep_fwd(const float* logits, const float* targets, float* loss, int n, int c,
                 fsc_stream_t stream);
/* dlogits[n,k] = dloss[n] * d loss[n] / d s[n,k] */
int fsc_lsep_bwd(const float* logits, const float* targets, const float* dloss,
                 float* dlogits, int n, int c, fsc_stream_t stream);
/* mean over all n*c elements of BCE(sigmoid(x), t), log terms clamped at -100 like torch */
int fsc_bce_fwd(const float* logits, const float* targets, float* loss_scalar,
                double* workspace /* 1 double */, long count, fsc_stream_t stream);
int fsc_bce_bwd(const float* logits, const float* targets, const float* dloss_scalar,
                float* dlogits, long count, fsc_stream_t stream);
int fsc_sigmoid(const float* x, float* y, long count, fsc_stream_t stream);
/* y = scale * mean(x) (count elements) and its backward dx = dy * scale / count */
int fsc_mean_fwd(const float* x, float* y, long count, float scale, fsc_stream_t stream);
int fsc_mean_bwd(const float* dy, float* dx, long count, float scale, fsc_stream_t stream);

/* ------------------------------------------------------------------ MixUp (a-8)
 * ops/audio.py:32-52 applied to a batch resident on the device.  For row n:
 * equal lengths -> (a + b) / 2; otherwise out = longer * f32(alpha) except
 * [start, start + shorter) which is REPLACED by shorter * f32(1 - alpha) (`=+` at audio.py:50);
 * samples past the longer length are zero.  labels = clip(la + lb, 0, 1). */
int fsc_mixup_batch(const float* a, const float* b, const int* len_a, const int* len_b,
                    const int* start, const float* alpha, const float* one_minus_alpha,
                    float* out, int n, long t_a, long t_b, long t_out,
                    const float* labels_a, const float* labels_b, float* labels_out, int c,
                    fsc_stream_t stream);

/* MixUp with a partner table: row n of `a` is mixed with row partner[n] of `b` (len_b[n] = that partner's length), or
 * passes through unchanged when partner[n] < 0 (MixUp.p not drawn, ops/transforms.py:57).  labels likewise. */
int fsc_mixup_rows(const float* a, const float* b, const int* partner, const int* len_a,
                   const int* len_b, const int* start, const float* alpha,
                   const float* one_minus_alpha, float* out, int n, long t_a, long t_b, long t_out,
                   const float* labels_a, const float* labels_b, float* labels_out, int c,
                   fsc_stream_t stream);

/* ------------------------------------------------------------------ device-side input pipeline (f-2)
 * The waveform transforms that are pure index work, batched on the device: SampleLongAudio's random crop
 * (ops/transforms.py:292-309) and ShuffleAudio's chunk permutation (:256-271 -> ops/audio.py:55-67).  The random
 * draws stay on the host (same generators, same order as the reference's per-sample Compose); the device gets, per
 * output row n, its source row src_row[n] in `src` (rows of src_stride floats) and seg_count[n] <= max_seg <= 256
 * segments: segment k copies src[seg_src[n*max_seg + k] ...] to out[seg_dst[n*(max_seg+1) + k] ... seg_dst[.. k+1]).
 * Output past the last segment is 0 (the collate padding, ops/padding.py:26-28).  Bit-exact copies. */
int fsc_segments_gather(const float* src, long src_stride, const int* src_row, const int* seg_count,
                        const int* seg_src, const int* seg_dst, int max_seg, float* out, int n,
                        long t_out, fsc_stream_t stream);

/* ------------------------------------------------------------------ RNN aggregation head (f-3)
 * aggregation_type == "rnn" (classifiers.py:514-522, 592-597): rnn_input = mean(h, dim=2).permute(0, 2, 1), then
 * LayerNorm((C,)) and a bidirectional GRU(C, 128, batch_first=True); the two final hidden states are the block's
 * features.  The input projections X W_ih^T + b_ih of all time steps and the weight gradients are GEMMs
 * (fsc_linear_fwd / fsc_linear_bwd); these entry points are the remaining pieces.  Gate order r, z, n; h' = (1 - z) n + z h
 * (torch.nn.GRU).  All fp32. */
/* y (N, W, C) = mean over H of x (N, C, H, W); backward writes dx (N, C, H, W) = dy / H broadcast over H */
int fsc_freq_mean_fwd(const float* x, float* y, int n, int c, int h, int w, fsc_stream_t stream);
int fsc_freq_mean_bwd(const float* dy, float* dx, int n, int c, int h, int w, fsc_stream_t stream);
/* LayerNorm over the last dimension of (rows, C): biased variance, eps inside the root; saves mean / rstd per row */
int fsc_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y,
                      float* mean, float* rstd, long rows, int c, fsc_stream_t stream);
int fsc_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                      const float* gamma, float* dx, float* dgamma, float* dbeta, long rows, int c,
                      fsc_stream_t stream);
/* One GRU time step for a batch.  gx: rows of 3*hidden input projections [r | z | n] (b_ih included), row stride
 * gx_stride floats (a time slice of a (N, T, 3*hidden) tensor); w_hh (3*hidden, hidden), b_hh (3*hidden).
 * r/z/n/ghn_save (batch x hidden each, all NULL in inference) keep what the backward step needs. */
int fsc_gru_step_fwd(const float* gx, long gx_stride, const float* h_prev, const float* w_hh,
                     const float* b_hh, float* h_out, float* r_save, float* z_save, float* n_save,
                     float* ghn_save, int batch, int hidden, fsc_stream_t stream);
/* Backward of one step: from dh (gradient of h_out) writes dgx (row stride dgx_stride), dgh (batch x 3*hidden: the
 * gradient of W_hh h + b_hh, input of the W_hh / b_hh gradient GEMM) and dh_prev = dh * z + dgh W_hh. */
int fsc_gru_step_bwd(const float* dh, const float* r_save, const float* z_save, const float* n_save,
                     const float* ghn_save, const float* h_prev, const float* w_hh, float* dgx,
                     long dgx_stride, float* dgh, float* dh_prev, int batch, int hidden,
                     fsc_stream_t stream);

/* ------------------------------------------------------------------ optimizers (K16)
 * ops/training.py:9-12: Adam(amsgrad=True) and SGD(momentum=0.9, nesterov=True), both with
 * L2 weight decay folded into the gradient (torch semantics). */
typedef struct {
    float* param;
    const float* grad;
    float* state0; /* adam: exp_avg        | sgd: momentum buffer */
    float* state1; /* adam: exp_avg_sq     | sgd: unused */
    float* state2; /* adam: max_exp_avg_sq | sgd: unused */
    long count;
} fsc_opt_tensor;

/* tensors_host is a HOST array (copied into kernel arguments, 64 tensors per launch).
 * step = 1-based step count used for bias correction; grad_scale multiplies every gradient
 * first (1/world_size after a sum all-reduce). */
int fsc_adam_amsgrad_step(const fsc_opt_tensor* tensors_host, int n_tensors, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int step,
                          float grad_scale, fsc_stream_t stream);
/* The same step with its two step-dependent factors read from DEVICE memory -- factors_dev[0] = lr / (1 - beta1^step),
 * factors_dev[1] = 1 / sqrt(1 - beta2^step), as fsc_adam_step_factors computes them on the host -- so that a training step recorded
 * once in a HIP graph (hipStreamBeginCapture around the step's calls; every entry point of this library is capture-safe: no
 * allocation, no synchronisation, no host read-back) can be replayed with a moving learning rate and step count: the caller
 * updates the two floats on the replay stream in front of each hipGraphLaunch. */
void fsc_adam_step_factors(float lr, float beta1, float beta2, int step, float* out2_host);
int fsc_adam_amsgrad_step_dev(const fsc_opt_tensor* tensors_host, int n_tensors, const float* factors_dev,
                              float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                              fsc_stream_t stream);
/* first_step != 0: momentum buffer := gradient (torch's first-step rule) */
int fsc_sgd_nesterov_step(const fsc_opt_tensor* tensors_host, int n_tensors, float lr,
                          float momentum, float weight_decay, int first_step,
                          float grad_scale, fsc_stream_t stream);

/* ------------------------------------------------------------------ misc device helpers */
/* out[c][8] (pre-zeroed) += per channel, over all images of x (N, C, H, W): sums of the first row, last row, first column,
 * last column and the four corner elements.  Used to obtain the first block's input-BN parameter gradients from the stem
 * convolution's weight gradient instead of its input gradient (classifiers.py:524-531). */
int fsc_plane_border_sums(const float* x, int n, int c, int h, int w, float* out, fsc_stream_t stream);
/* First block of the 1-d model (classifiers.py:147-154: BatchNorm1d -> Conv1d(k = 3) on a spectrogram that needs no gradient): the
 * Conv1d weight gradient and the BatchNorm's dgamma / dbeta from the weight-gradient pass over the BatchNorm's RAW input, in one
 * launch: dwx (c_out, c_in, 3) = fsc_conv_wgrad(x, dc), dc (n, c_out, 1, len) the convolution's output gradient, dc_chan_sum
 * (c_out) its per-channel sum; with T[co][t] = dc_chan_sum minus dc's first (t = 0) / last (t = 2) column summed over the images
 * and C = invstd (dwx - mean T):  dw = gamma C + beta T, dgamma[ci] = sum w C, dbeta[ci] = sum w T. */
int fsc_first_block_1d_finish(const float* dwx, const float* dc, const float* dc_chan_sum, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, const float* weight, int n, int c_in, int c_out, int len,
                              float* dw, float* dgamma, float* dbeta, fsc_stream_t stream);
/* out[0] = min |x[i]| over n >= 1 floats (one small launch: the guard of a division by a parameter vector) */
int fsc_absmin(const float* x, long n, float* out, fsc_stream_t stream);
/* split = 0: out (rows, sum widths) = the `count` <= 16 column pieces (rows, widths[i]) side by side (torch.cat(feats, -1),
 * classifiers.py:595); split = 1: the pieces from `out` (the gradient's way back).  `pieces` / `widths` are HOST arrays. */
int fsc_cat_cols(const float* const* pieces, const int* widths, int count, int rows, float* out, int split, fsc_stream_t stream);
/* *counters[i] += 1 for `count` int64 device scalars (HOST array of device pointers): BatchNorm.num_batches_tracked */
int fsc_bump_counters(long* const* counters, int count, fsc_stream_t stream);
int fsc_fill(float* x, float value, long count, fsc_stream_t stream);
/* y = a*x + y  (used for gradient accumulation / bucket flattening) */
int fsc_axpy(const float* x, float a, float* y, long count, fsc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FSC_HIP_H */
