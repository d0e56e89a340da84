/* sept.h -- C ABI of libsept_hip.so: the MI355X (gfx950) hot path of
 * usc-sail/speech-emotion-privacy-trust (batched STFT->mel->dB features + the
 * cloak / CNN+GRU / gradient-reversal training step).
 *
 * The reference is pure Python and has no FFI layer; each entry point below names the
 * reference call it stands behind (file:line relative to the reference tree) -- these are
 * the points where a maintainer would bind this library.
 *
 * Conventions
 *   - every function returns 0 on success, a negative sept_status otherwise;
 *     sept_last_error() returns a thread-local human-readable message for the last failure.
 *   - all tensor pointers are caller-owned DEVICE pointers (hipMalloc'd / torch storage),
 *     dense, row-major in the stated shape; no function allocates caller-visible memory.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised,
 *     and no entry point calls hipMalloc/hipFree/hipMemcpy (graph-capture safe), except the
 *     *_plan_create / *_plan_destroy pair which own small device tables.
 *   - no torch types appear in any signature.
 */
#ifndef SEPT_H
#define SEPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sept_status {
  SEPT_OK = 0,
  SEPT_ERR_INVALID = -1,     /* bad argument (null pointer, non-positive size, ...) */
  SEPT_ERR_UNSUPPORTED = -2, /* shape / parameter outside what the kernels implement */
  SEPT_ERR_HIP = -3,         /* a HIP runtime call failed (text in sept_last_error) */
  SEPT_ERR_NO_DEVICE = -4    /* no gfx950 device visible */
} sept_status;

const char* sept_last_error(void);
/* library ABI version; bumped when a signature changes */
int sept_abi_version(void);
/* 0 if a HIP device is visible and is gfx950, else SEPT_ERR_NO_DEVICE */
int sept_device_check(void);

/* ------------------------------------------------------------------------------------
 * Features: framed STFT (reflect-pad centred, periodic Hann, hop) -> |.|^2 -> mel
 * filterbank -> 10 log10(clamp(., 1e-10)).
 * Replaces: feature_extraction/audio_feature_extraction.py:29-46 mel_spectrogram()
 *   = torchaudio MelSpectrogram(16000, n_mels, n_fft, win_length=n_fft, hop_length=160,
 *     window_fn=hann_window) + AmplitudeToDB()  (torch.stft + MelScale matmul + log10).
 * The plan owns the device tables the reference rebuilds on every call
 * (audio_feature_extraction.py:36-43): window, FFT twiddles, sparse filterbank.
 * ------------------------------------------------------------------------------------ */
typedef struct sept_mel_plan sept_mel_plan;

#define SEPT_MEL_LAYOUT_BFT 0 /* out[B][n_mels][T]  -- the reference's (C, n_mels, T)   */
#define SEPT_MEL_LAYOUT_BTF 1 /* out[B][T][n_mels]  -- window-major, what training eats  */

/* window_host[n_fft], fb_host[n_freq = n_fft/2+1][n_mels] are HOST float32 tables (the
 * host side builds them exactly as torchaudio does).  Supported n_fft: 400, 800, 1024,
 * 1600 (every size the reference uses); hop must be even; each filter's non-zeros must
 * be one contiguous run of bins (true for any triangular bank). */
int sept_mel_plan_create(int n_fft, int hop, int n_mels, const float* window_host,
                         const float* fb_host, sept_mel_plan** plan_out);
int sept_mel_plan_destroy(sept_mel_plan* plan);
/* frames produced for a clip of `length` samples: 1 + length / hop */
int sept_mel_num_frames(const sept_mel_plan* plan, int length);
/* wav[B][L] float32 -> out (layout above) float32 dB.  Requires L > n_fft/2. */
int sept_mel_forward(const sept_mel_plan* plan, const float* wav, int B, int L, float* out,
                     int layout, void* stream);
/* name of the device kernel sept_mel_forward launches for this plan (for rocprof filters) */
const char* sept_mel_kernel_name(const sept_mel_plan* plan);

/* ------------------------------------------------------------------------------------
 * Conv stack of two_d_cnn_lstm (model/baseline_models.py:171-189): Conv2d(k=5, pad=2) on
 * bf16 MFMA with fp32 accumulation.  Activations are NHWC bf16 on the device
 * ([B][H][W][C]); the reference's NCHW fp32 view exists only at the module boundary.
 * ------------------------------------------------------------------------------------ */
/* w_oihw: the nn.Conv2d weight (cout, cin, 5, 5) fp32 -> wt_bf16[25][o'][i'] bf16.
 * mode 0: forward operand (o'=cout, i'=cin).  mode 1: data-gradient operand (o'=cin,
 * i'=cout, taps flipped) so that dX = sept_conv5x5_forward(dY, wt_mode1, cin'=cout, cout'=cin). */
int sept_conv5x5_prep_weights(const float* w_oihw, int cout, int cin, int mode, void* wt_bf16,
                              void* stream);
/* y[B][H][W][cout] = conv5x5(x[B][H][W][cin], wt) (+ bias[cout] if non-null), bf16 in/out.
 * Channel pairs: 32->64, 64->128 (forward), 64->32, 128->64 (data gradient), 128->128. */
int sept_conv5x5_forward(const void* x_bf16, const void* wt_bf16, const float* bias, void* y_bf16,
                         int B, int H, int W, int cin, int cout, void* stream);
/* Forward + the BatchNorm statistics partials of its output (training: the BatchNorm that follows needs no
 * statistics pass): stats[2*cout][sept_conv5x5_stats_parts(B,H,W,cin,cout)] floats -- per-workgroup sums, then
 * sums of squares, of the bf16-rounded outputs -- to be finished by sept_bn_stats_from_partials.
 * sept_conv5x5_stats_parts returns 0 when the shape has no statistics form (cin > cout, unsupported pair, or
 * the LDS copy of the output tile would cost a workgroup per CU). */
int sept_conv5x5_stats_parts(int B, int H, int W, int cin, int cout);
int sept_conv5x5_forward_stats(const void* x_bf16, const void* wt_bf16, const float* bias, void* y_bf16, float* stats,
                               int B, int H, int W, int cin, int cout, void* stream);

/* dW[cout][cin][5][5] (fp32, OIHW, overwritten) = sum over batch and pixels of dy x shifted x.
 * bf16 MFMA with transposing LDS reads; deterministic (per-workgroup slabs in `ws`, which
 * holds sept_conv5x5_wgrad_workspace_floats(cin, cout) floats).  32->64, 64->128, 128->128. */
size_t sept_conv5x5_wgrad_workspace_floats(int cin, int cout);
int sept_conv5x5_backward_weight(const void* x_bf16, const void* dy_bf16, float* ws, float* dw, int B, int H,
                                 int W, int cin, int cout, void* stream);

/* First layer Conv2d(1 -> 32, k=5, pad=2) (baseline_models.py:172): x / dx fp32 [B][H][W],
 * y / dy bf16 [B][H][W][32], w fp32 [32][1][5][5].  fp32 direct forward; packed-bf16 dot2
 * data gradient (feeds the cloak parameters, cloak_models.py:45-58); deterministic weight
 * gradient through a workspace of sept_conv1_workspace_floats() floats.
 * `wprep` is a scratch buffer of sept_conv1_prep_floats() floats (re-laid weights, read with
 * scalar loads by the kernels). */
size_t sept_conv1_prep_floats(void);
int sept_conv1_forward(const float* x, const float* w, const float* bias, float* wprep, void* y_bf16, int B,
                       int H, int W, void* stream);
/* The same forward that also leaves the BatchNorm statistics partials of its output (the nn.BatchNorm2d
 * of baseline_models.py:173 then needs no pass of its own over the tensor): stats = [64][sept_conv1_stats_parts(B, H)]
 * floats (32 sums then 32 sums of squares of the bf16-rounded outputs, one column per workgroup), finished by
 * sept_bn_stats_from_partials (partials [2C][nparts]). */
int sept_conv1_stats_parts(int B, int H);
int sept_conv1_forward_stats(const float* x, const float* w, const float* bias, float* wprep, void* y_bf16,
                             float* stats, int B, int H, int W, void* stream);
int sept_conv1_backward_data(const void* dy_bf16, const float* w, float* wprep, float* dx, int B, int H, int W,
                             void* stream);
size_t sept_conv1_workspace_floats(void);
int sept_conv1_backward_weight(const float* x, const void* dy_bf16, float* ws, float* dw, float* db /*nullable*/,
                               int B, int H, int W, void* stream);

/* BatchNorm2d + ReLU + MaxPool2d(pool in {1,2}) + Dropout2d (baseline_models.py:173-176 etc.),
 * NHWC bf16, C in {32, 64, 128}.  `ws` is a float workspace of sept_bn_workspace_floats(C).
 * sept_bn_stats: training-mode batch statistics of x[n_rows][C] (+ running-stat update with
 * `momentum`, unbiased running_var, num_batches_tracked += 1; any of the three may be null).
 * sept_bn_eval_stats: mean/invstd from the running buffers (module in eval mode).
 * forward:  y[B][H/p][W/p][C] = dropscale[B][C] * maxpool_p(relu(gamma*(x-mean)*invstd+beta))
 *           (dropscale null = no Dropout2d; entries are 0 or 1/(1-p))
 * backward: dx[B][H][W][C] from dy[B][H/p][W/p][C] through dropout, pooling (first maximum,
 *           as ATen), ReLU and training-mode BatchNorm; dgamma/dbeta optional. */
size_t sept_bn_workspace_floats(int C);
int sept_bn_stats(const void* x_bf16, long n_rows, int C, float* ws, float* mean, float* invstd,
                  float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                  float eps, void* stream);
int sept_bn_stats_from_partials(const float* partials, int nparts, long n_rows, int C, float* mean,
                                float* invstd, float* running_mean, float* running_var,
                                long long* num_batches_tracked, float momentum, float eps, void* stream);
int sept_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                       float* invstd, void* stream);
int sept_bn_relu_pool_forward(const void* x_bf16, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* dropscale, void* y_bf16, int B, int H, int W,
                              int C, int pool, void* stream);
/* y_bf16: the pooled output sept_bn_relu_pool_forward produced for the same x (dropout applied), or NULL.
 * With it the channel sums (sum dy, sum dy*xhat) are taken from the pooled tensors alone -- xhat at a
 * window's maximum is (y / dropscale - beta) / gamma -- instead of from every window of x. */
int sept_bn_relu_pool_backward(const void* dy_bf16, const void* x_bf16, const void* y_bf16, const float* mean,
                               const float* invstd, const float* gamma, const float* beta, const float* dropscale,
                               float* ws, void* dx_bf16, float* dgamma, float* dbeta, int B, int H, int W, int C,
                               int pool, void* stream);

/* Sync-BN pieces (statistics over the GLOBAL batch of a data-parallel step; off by default).  The two
 * fused entries above split where the per-channel sums exist so the caller can all-reduce them:
 *   forward:  sept_bn_partial_sums -> sums (2C doubles: sum, sum of squares) of this rank's rows;
 *             [all-reduce SUM];  sept_bn_stats_from_sums(sums, n_total rows over all ranks, ...)
 *   backward: sept_bn_relu_pool_backward_reduce -> sums (2C floats: sum dy, sum dy*xhat; dgamma / dbeta are
 *             the LOCAL sums, the data-parallel gradient average does the rest);  [all-reduce SUM];
 *             sept_bn_relu_pool_backward_apply(..., sums, n_total elements per channel over all ranks, dx). */
int sept_bn_partial_sums(const void* x, long n_rows, int C, float* ws, double* sums, void* stream);
int sept_bn_stats_from_sums(const double* sums, double n_total, int C, float* mean, float* invstd,
                            float* running_mean, float* running_var, long long* num_batches_tracked,
                            float momentum, float eps, void* stream);
int sept_bn_relu_pool_backward_reduce(const void* dy, const void* x, const void* y, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta, const float* dropscale,
                                      float* ws, float* sums, float* dgamma, float* dbeta, int B, int H,
                                      int W, int C, int pool, void* stream);
int sept_bn_relu_pool_backward_apply(const void* dy, const void* x, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, const float* dropscale,
                                     const float* sums, double n_total, void* dx, int B, int H, int W,
                                     int C, int pool, void* stream);

/* ------------------------------------------------------------------------------------
 * Linear layers / GRU projections: C[M][N] = alpha * A(M,K) B(K,N) (+ bias[N]) (+ beta * C)
 * on the exact-fp32 MFMA.  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn] (element
 * strides), C row-major with leading dimension ldc.  A, B and C may each be bf16.
 * Stands behind nn.Linear (baseline_models.py:208-210) and nn.GRU's x W_ih^T
 * (baseline_models.py:191-193) and their autograd (dx = dy W, dW = dy^T x).  `ws` (nullable,
 * ws_floats floats) lets weight-gradient shapes (few tiles, long K) split K deterministically.
 * ------------------------------------------------------------------------------------ */
int sept_gemm(const void* A, long sam, long sak, int a_is_bf16, const void* B, long sbk, long sbn,
              int b_is_bf16, void* C, long ldc, int c_is_bf16, const float* bias, int M, int N, int K,
              float alpha, float beta, float* ws, long ws_floats, void* stream);

/* Long-K "NT" product on the bf16 matrix pipe with split operands (GRU layer-0 input projections,
 * baseline_models.py:191-193 and their data gradient): C[m][n] = sum_k A[m][k] B[n][k] (+ bias[n]).
 * A is bf16 (exact) or fp32 (split hi+lo in the kernel), B is fp32 (split hi+lo), both k-contiguous
 * with 16-byte aligned rows (K a multiple of 32; lda a multiple of 8 for bf16 A / 4 for fp32 A; ldb a
 * multiple of 4);
 * products are accumulated in fp32 from the hi*hi, hi*lo (and lo*hi) passes: ~2^-17 relative. */
int sept_gemm_nt_split(const void* A, long lda, int a_is_bf16, const float* B, long ldb, void* C, long ldc,
                       int c_is_bf16, const float* bias, int M, int N, int K, void* stream);

/* Weight-gradient ("TN") product with split operands: C[m][n] = sum_k A[k][m] B[k][n], A fp32
 * (gradient rows, split hi+lo), B bf16 (exact) or fp32 (split hi+lo), both with k as the row index
 * (16-byte aligned rows; M, lda multiples of 4; N, ldb multiples of 8 for bf16 B / 4 for fp32 B).
 * K is split over workgroups into `ws` (sept_gemm_tn_workspace_floats(M, N) floats, may be NULL)
 * and summed in fixed order.  Replaces the dW = dy^T x products of nn.GRU / nn.Linear autograd. */
size_t sept_gemm_tn_workspace_floats(int M, int N);
int sept_gemm_tn_split(const float* A, long lda, const void* B, long ldb, int b_is_bf16, float* C, long ldc,
                       int M, int N, int K, float* ws, long ws_floats, void* stream);
/* the same product that also leaves colsum[m] = sum_k A[k][m] (M floats): a layer's bias gradient beside its weight
 * gradient, from the gradient rows the product stages anyway (autograd of nn.GRU / nn.Linear, baseline_models.py:191-210) */
int sept_gemm_tn_split_colsum(const float* A, long lda, const void* B, long ldb, int b_is_bf16, float* C, long ldc,
                              float* colsum, int M, int N, int K, float* ws, long ws_floats, void* stream);

/* Recurrent part of nn.GRU(.., hidden 64, bidirectional, batch_first) -- one launch per layer
 * for both directions and all T steps (baseline_models.py:191-193; gate order r, z, n).
 *   gi    [B][T][2][3H]  x W_ih^T + b_ih for (forward, reverse)     (from sept_gemm)
 *   whh_fwd / whh_rev [3H][H], bhh_fwd / bhh_rev [3H]  (weight_hh_l*, weight_hh_l*_reverse, ...)
 *   out   [B][T][2H]     hidden states (forward | reverse), h0 = 0
 *   gates [B][T][2][4][H] saved (r, z, n, W_hn h + b_hn) for the backward pass
 * backward: from dout[B][T][2H] produces dgi (gradient of gi), dgh (gradient of h W_hh^T + b_hh)
 * and hprev[B][T][2][H] (the h each step consumed) so that dW_hh = dgh^T hprev, db_hh =
 * colsum(dgh), dW_ih = dgi^T x, db_ih = colsum(dgi), dx = dgi W_ih are plain sept_gemm /
 * sept_colsum calls.  H must be 64. */
int sept_gru_forward(const float* gi, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd,
                     const float* bhh_rev, float* out, float* gates, int B, int T, int H, void* stream);
int sept_gru_backward(const float* dout, const float* out, const float* gates, const float* whh_fwd,
                      const float* whh_rev, float* dgi, float* dgh, float* hprev, int B, int T, int H,
                      void* stream);
/* The dropout between the two recurrent layers (nn.GRU(dropout=0.2), baseline_models.py:191-193) folded into the
 * recurrences: the forward also writes out_masked = out * mask (mask: (B, T, 2H) scale values, 0 or 1/(1-p)), the next
 * layer's input; the backward takes the gradient of that masked output and multiplies it by the mask as it is fetched. */
int sept_gru_forward_masked(const float* gi, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd,
                            const float* bhh_rev, float* out, float* gates, const float* mask, float* out_masked, int B,
                            int T, int H, void* stream);
int sept_gru_backward_masked(const float* dout, const float* dout_mask, const float* out, const float* gates,
                             const float* whh_fwd, const float* whh_rev, float* dgi, float* dgh, float* hprev, int B, int T,
                             int H, void* stream);

/* LSTM recurrence (gate order i, f, g, o; nn.LSTM of deep_two_d_cnn_lstm_tmp, baseline_models.py:388-509, and the
 * rnn_cell='lstm' option of the other classes): gi (B, T, 2, 4H) = x W_ih^T + b_ih for both directions;
 * out (B, T, 2H); gates (B, T, 2, 4, H) and cells (B, T, 2, H) are kept for the backward pass, which returns
 * dgates (B, T, 2, 4H) -- the gradient wrt the gate pre-activations, i.e. of BOTH gi and W_hh h + b_hh -- and
 * hprev (B, T, 2, H), the operand of dW_hh.  H = 64 or 128. */
int sept_lstm_forward(const float* gi, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd,
                      const float* bhh_rev, float* out, float* gates, float* cells, int B, int T, int H,
                      void* stream);
int sept_lstm_backward(const float* dout, const float* out, const float* gates, const float* cells,
                       const float* whh_fwd, const float* whh_rev, float* dgates, float* hprev, int B,
                       int T, int H, void* stream);

/* ------------------------------------------------------------------------------------
 * cloak_noise (model/cloak_models.py:24-58).  n_per = W*F elements of locs/rhos/eps/mask.
 *   scales = (1 + tanh(rhos)) / 2 * (max_scale - min_scale) + min_scale          (:41-43)
 *   forward: xn[b] = x[b] (*mask) + locs + scales * eps (*mask)                   (:45-58)
 *   backward: g = dxa + gscale_b * dxb  (dxb nullable; the second branch's input gradient goes
 *     through GradientReversal, gscale_b = -grl_lambda, reversal_gradient.py:18-23);
 *     dlocs = sum_b g;  drhos = sum_b g * eps (*mask) * dscales/drhos
 *             - scale_lambda * d/drhos log(mean(scales))   (training_cloak_with_grl.py:158-160)
 * sept_cloak_scales writes scales[n] and/or their mean (device scalar). */
int sept_cloak_forward(const float* x, const float* locs, const float* rhos, const float* eps, const float* mask,
                       float min_scale, float max_scale, float* xn, int B, long n_per, void* stream);
/* sept_window_norm followed by sept_cloak_forward in one pass (preprocess_adversary_data.py:30-35,131 +
 * cloak_models.py:45-58 on the windows of a training step): mel (B, T, F) -> xn (B * nwin, win * F); one epsilon
 * (win * F) for the whole batch; mean / stdv (F) both NULL = no normalisation.  Bit-identical to the two calls.
 * n_per = the element count of locs / rhos / eps (/ mask): the entry refuses anything but win * F (a feature plan whose
 * window or mel count differs from the cloak's parameters would index them out of bounds). */
int sept_window_norm_cloak(const float* mel, const float* mean, const float* stdv, const float* locs, const float* rhos,
                           const float* eps, const float* mask, float min_scale, float max_scale, float* xn, int B, int T,
                           int F, int win, int shift, int nwin, long n_per, void* stream);
/* forward with eps[eps_rows][n_per], eps_rows = 1 (as sept_cloak_forward) or B: one epsilon per row -- the test()
 * loops (training_cloak_with_grl.py:72-83, adversary_cloak_evaluation.py:66-96) run ONE window per forward, so
 * cloak_noise.sample_noise draws a fresh epsilon for every window; the batched inference path keeps that. */
int sept_cloak_forward_rows(const float* x, const float* locs, const float* rhos, const float* eps, int eps_rows,
                            const float* mask, float min_scale, float max_scale, float* xn, int B, long n_per,
                            void* stream);
int sept_cloak_scales(const float* rhos, float min_scale, float max_scale, float* scales, float* mean_out, long n,
                      void* stream);
int sept_cloak_backward(const float* dxa, const float* dxb, float gscale_b, const float* rhos, const float* eps,
                        const float* mask, float min_scale, float max_scale, float scale_lambda,
                        const float* scale_mean, float* dlocs, float* drhos, int B, long n_per, void* stream);

/* ------------------------------------------------------------------------------------
 * Layer 1 in ONE pass with given statistics (inference; baseline_models.py:172-176: Conv2d(1, 32, 5, padding 2) ->
 * BatchNorm2d (running statistics) -> ReLU -> MaxPool2d(2) -> Dropout2d scale): conv1's output is 16x its input, so
 * sept_conv1_bn_relu_pool_forward keeps it in registers and writes only y_pooled (B, H/2, W/2, 32) bf16 (H even,
 * W % 16 == 0: sept_conv1_fused_supported; other shapes use the separate conv1 / BatchNorm entry points above).
 * Training uses the pool-first form below.  wprep: sept_conv1_prep_floats() floats of scratch, as for sept_conv1_forward.
 * (sept_bn_bwd_sums_from_partials: partials[2C][nparts] -> sums[2C] (+ dgamma, dbeta); all-reducible for sync-BN.) */
/* BatchNorm backward sums in the epilogue of the data-gradient conv that PRODUCES dy (baseline_models.py:172-188,
 * backward): sept_conv5x5_dgrad_bnsums = sept_conv5x5_forward on data-gradient operands (wt from
 * sept_conv5x5_prep_weights mode 1, cin > cout) that also leaves partials[2*cout][sept_conv5x5_stats_parts(B, H, W,
 * cin, cout)] of (sum g, sum g * xhat) of the BatchNorm whose pooled output is `ypool` (B, H, W, cout) bf16, formed
 * as in sept_bn_relu_pool_backward's pooled path; sept_bn_relu_pool_backward_presummed then finishes that
 * BatchNorm's backward without a reduce pass (chunks with |gamma| < 1e-3 are re-summed from the windows of x). */
int sept_conv5x5_bwsums_parts(int B, int H, int W, int cin, int cout);   /* 0: no such form for this shape */
/* Data gradient of a 5x5 conv whose incoming gradient is that of a BatchNorm + ReLU + MaxPool 2x2 (+ Dropout2d) block's
 * PRE-ACTIVATIONS (baseline_models.py:178-188 + autograd: blocks 2 and 3 of a network whose conv weights need no gradient),
 * with that block's backward apply pass inside the conv's tile loader: pre (B, H, W, cin) bf16 = the block's stored
 * pre-activations, gpool (B, H/2, W/2, cin) bf16 = gradient of its pooled output, sums[2 cin] = (sum g, sum g xhat) as
 * sept_bn_backward_sums_presummed / sept_bn_relu_pool_backward_reduce leave them, mean / invstd / gamma / beta [cin],
 * dropscale [B][cin] or NULL; wt from sept_conv5x5_prep_weights(mode 1); dx_out (B, H, W, cout) bf16.  The (B, H, W, cin)
 * gradient tensor sept_bn_relu_pool_backward would write (and the conv read) does not exist.  H, W even.
 * Epilogue (optional): ep_ypool + ep_gamma + ep_beta (+ ep_dropscale) as sept_conv5x5_dgrad_bnsums; with ep_mean / ep_invstd
 * too, ep_ypool is a pool-first block's ext, as sept_conv5x5_dgrad_bnsums_ext; partials[2 cout][sept_conv5x5_bnapply_parts()].
 * sept_conv5x5_bnapply_parts(..., want_sums): columns of `partials` (want_sums != 0) or 1 (plain epilogue) when this shape
 * has a loader form at this width, else 0 (the caller then runs the apply pass and the plain conv). */
int sept_conv5x5_bnapply_parts(int B, int H, int W, int cin, int cout, int want_sums);
int sept_conv5x5_dgrad_bnapply(const void* pre, const void* gpool, const float* sums, const float* mean,
                               const float* invstd, const float* gamma, const float* beta, const float* dropscale,
                               const void* wt, void* dx_out, const void* ep_ypool, const float* ep_mean,
                               const float* ep_invstd, const float* ep_gamma, const float* ep_beta,
                               const float* ep_dropscale, float* partials, int B, int H, int W, int cin, int cout,
                               void* stream);
/* Forward conv behind a block in pool-first form (sept_conv1_forward_pool) with that block's activation pass in the conv's
 * tile loader (baseline_models.py:173-178: BatchNorm2d + ReLU + MaxPool2d + Dropout2d of block 1 feeding conv.5): ext
 * (B, H, W, cin) bf16 = the block's window extrema; the conv's input is dropscale * relu(bn(ext)), bit for bit what
 * sept_bn_relu_ext_forward stores, formed on the way into the LDS tile -- that tensor is neither written nor read (for a
 * network whose conv.5 needs no weight gradient nothing else consumes it).  stats (nullable): the statistics partials of
 * sept_conv5x5_forward_stats, [2 cout][sept_conv5x5_act_parts(..., 1)].  sept_conv5x5_act_parts: columns of `stats`
 * (want_stats != 0) or 1 when the shape has this form at this width, else 0. */
int sept_conv5x5_act_parts(int B, int H, int W, int cin, int cout, int want_stats);
int sept_conv5x5_forward_act(const void* ext, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             const float* dropscale, const void* wt, const float* bias, void* y, float* stats, int B, int H,
                             int W, int cin, int cout, void* stream);
/* Which tile shape a launch at image width W takes (host-only, no GPU needed): out[6] = pixel blocks per wave, waves over
 * pixels, waves over output channels, taps per barrier (0 / negative: double-buffered), input-channel slices, LDS bytes.
 * Returns 0, or SEPT_ERR_UNSUPPORTED when the channel pair (or its statistics form, want_stats != 0) has no kernel. */
int sept_conv5x5_variant(int W, int cin, int cout, int want_stats, int* out);
int sept_conv5x5_dgrad_bnsums(const void* dy_out, const void* wt, void* dx_out, const void* ypool, const float* bn_gamma,
                              const float* bn_beta, const float* dropscale, float* partials, int B, int H, int W, int cin,
                              int cout, void* stream);
int sept_bn_relu_pool_backward_presummed(const void* dy, const void* x, const float* mean, const float* invstd,
                                         const float* gamma, const float* beta, const float* dropscale,
                                         const float* partials, int nparts, float* ws, void* dx, float* dgamma,
                                         float* dbeta, int B, int H, int W, int C, int pool, void* stream);

/* Finish of a producer's partial BatchNorm backward sums: sums[2*C] from the partials a data-gradient conv's epilogue left
 * (tiny-|gamma| chunks re-summed from the windows; also yields dgamma / dbeta); n_total = elements per channel. */
int sept_bn_backward_sums_presummed(const void* dy, const void* x, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, const float* dropscale, const float* partials,
                                    int nparts, float* ws, float* sums_out, float* dgamma, float* dbeta, int B, int H,
                                    int W, int C, int pool, void* stream);
/* Block 1's data gradient WITHOUT a pre-activation-sized tensor (baseline_models.py:172-176 + autograd).  BatchNorm's
 * input gradient is scd_c g (at the window maxima) + c0_c + c1_c v with v = conv1(x) + bias; conv1 has one input
 * channel, so the data gradient of the dense part is a fixed linear map of x (a 9 x 9 filter away from the border, the
 * exact tap-by-tap form on the two-pixel ring) and only the sparse part goes through the MFMA data-gradient kernel,
 * whose loader expands it from the POOLED gradient and the window positions recorded by
 * sept_bn_relu_pool_forward_argmax (one byte per pooled element; pool * pool = the ReLU cut it).  w_f32 / bias: conv1's
 * weights (32, 1, 5, 5) and bias; sums[64] = (sum g, sum g * xhat) over n_total elements per channel; coef: SEPT_CONV1_COEF_FLOATS floats
 * of scratch.  H even, W a multiple of 4 and <= 128. */
#define SEPT_CONV1_COEF_FLOATS 2800
int sept_bn_relu_pool_forward_argmax(const void* x, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, const float* dropscale, void* y, void* idx_u8, int B, int H, int W,
                                     int C, int pool, void* stream);
int sept_conv1_backward_data_sparse(const void* dy_pooled, const void* idx_u8, const float* x, const float* w_f32,
                                    const float* bias, const float* mean, const float* invstd, const float* gamma,
                                    const float* dropscale, const float* sums, double n_total, const float* w, float* wprep,
                                    float* coef, float* dx, int B, int H, int W, void* stream);

/* Every weight-only operand build of a network in ONE launch: conv1's operand block (sept_conv1_prep), the bf16 operands of
 * the 5x5 convs (sept_conv5x5_prep_weights) and the packed recurrent input matrices (sept_gru_pack), each item with the exact
 * element mapping of its stand-alone entry (bit-identical results).  At most 12 items per call.
 *   SEPT_PREP_CONV1   : src0 = w (32,1,5,5), src1 = bias (32) or NULL, dst0 = wprep (sept_conv1_prep_floats() floats)
 *   SEPT_PREP_CONV5X5 : src0 = w OIHW fp32, dst0 = wt bf16, p0 = cout, p1 = cin, p2 = mode (0 forward, 1 data gradient)
 *   SEPT_PREP_GRU     : src0 / src1 = W_ih forward / reverse (G, K), src2 / src3 = b_ih forward / reverse (G),
 *                       dst0 = wcat (2G, K), dst1 = wcatT (K, 2G), dst2 = bcat (2G), p0 = G, p1 = K, p2 = C, p3 = Wd
 *                       (C > 0: layer-0 columns permuted from (c, w) to (w, c) feature order, C * Wd == K; C == 0: copied) */
enum { SEPT_PREP_CONV1 = 1, SEPT_PREP_CONV5X5 = 2, SEPT_PREP_GRU = 3 };
typedef struct sept_prep_item {
  int kind;
  const void *src0, *src1, *src2, *src3;
  void *dst0, *dst1, *dst2;
  int p0, p1, p2, p3;
} sept_prep_item;
int sept_prepare_operands(const sept_prep_item* items, int n_items, void* stream);

/* conv1's weights in operand form: every sept_conv1_* entry point builds it into `wprep` from (w, bias) unless called
 * with w == NULL ("wprep is current"); sept_conv1_prep builds it explicitly so a caller can keep it across calls. */
int sept_conv1_prep(const float* w, const float* bias, float* wprep, void* stream);
int sept_conv1_fused_supported(int H, int W);
int sept_conv1_bn_relu_pool_forward(const float* x, const float* w, const float* bias, float* wprep, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, const float* dropscale,
                                    void* y_pooled, int B, int H, int W, void* stream);
int sept_bn_bwd_sums_from_partials(const float* partials, int nparts, int C, float* sums_out, float* dgamma,
                                   float* dbeta, void* stream);

/* ------------------------------------------------------------------------------------
 * Block 1 in POOL-FIRST form (round 3; baseline_models.py:172-176 forward + autograd backward, no 64-byte-per-pixel
 * tensor in either direction).  maxpool(relu(bn(v))) over a 2x2 window = relu(bn(max v)) where gamma >= 0 and
 * relu(bn(min v)) where gamma < 0, and the sign of gamma is known when conv1 runs, so ONE pass over the input leaves
 *   stats [64][sept_conv1_stats_parts(B, H)]  the BatchNorm statistics partials of the (bf16-rounded) conv output of
 *                                             every pixel (finish with sept_bn_stats_from_partials),
 *   ext (B, H/2, W/2, 32) bf16                the window's extremum of the rounded conv output,
 *   idx (B, H/2, W/2, 32) u8                  its window position (scan order 0..3, first one wins).
 * sept_bn_relu_ext_forward then forms y = dropscale * relu(sc * ext + sh) (bit-identical to sept_bn_relu_pool_forward
 * on the stored pre-activations); px_per_item = pooled pixels per batch item (the Dropout2d scale is per item); with
 * idx_u8 given it also re-marks idx = 4 where the ReLU is inactive (the convention of sept_bn_relu_pool_forward_argmax)
 * -- the training path passes NULL and keeps the bytes pure positions.
 * Backward: (sum g, sum g * xhat) from (dy, ext) -- the ReLU is active where fma(ext, sc, sh) > 0, xhat = (ext - mean) *
 * invstd exactly, for any gamma -- either in the epilogue of the data-gradient conv that produces dy
 * (sept_conv5x5_dgrad_bnsums_ext, partials finished by sept_bn_bwd_sums_from_partials) or by
 * sept_bn_backward_sums_ext; BOTH leave dy MASKED (zero where the ReLU is inactive), which is what the consumers below
 * rely on: sept_conv1_backward_data_sparse / _sum (data gradient) and sept_conv1_backward_weight_sparse (weight
 * gradient) work from (masked dy, idx, x).  H even, W % 16 == 0 (sept_conv1_pool_supported); gamma NULL = all maxima. */
int sept_conv1_pool_supported(int H, int W);
/* ... and whether the BACKWARD kernels of a pool-first block 1 (sept_conv1_backward_weight_sparse, _data_sparse, _data_sum)
 * take the shape too: H >= 4, 16 <= W <= 128.  A training step that will need them asks this one. */
int sept_conv1_pool_backward_supported(int H, int W);
/* floats of the `coef` scratch those entries take (== SEPT_CONV1_COEF_FLOATS; a query so that a host never hard-codes it) */
size_t sept_conv1_coef_floats(void);
int sept_conv1_forward_pool(const float* x, const float* w, const float* bias, float* wprep, const float* gamma,
                            void* ext_bf16, void* idx_u8, float* stats, int B, int H, int W, void* stream);
int sept_bn_relu_ext_forward(const void* ext, void* idx_u8 /*nullable, in/out*/, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, const float* dropscale, void* y, int B,
                             long px_per_item, int C, void* stream);
int sept_bn_backward_sums_ext(void* dy /* masked in place */, const void* ext, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, const float* dropscale, float* ws, float* sums_out,
                              float* dgamma, float* dbeta, int B, long px_per_item, int C, void* stream);
size_t sept_conv1_wgrad_sparse_workspace_floats(void);
int sept_conv1_backward_weight_sparse(const void* dy_pooled, const void* idx_u8, const float* x, const float* w_f32,
                                      const float* bias, const float* mean, const float* invstd, const float* gamma,
                                      const float* dropscale, const float* sums, double n_total, float* ws, float* dw,
                                      float* db /*nullable*/, int B, int H, int W, void* stream);
/* sum over the batch of block 1's input gradient, (H, W) fp32 -- all the cloak's backward pass needs (its parameters are
 * shared by every sample: dlocs = sum_b g_b, drhos = eps * dscales/drhos * sum_b g_b, cloak_models.py:45-58).  Every stage
 * is linear in its per-sample input, so the pooled gradients are summed over the batch first (at their recorded positions),
 * then ONE single-image transposed conv and ONE single-image 81-tap pass finish the job in fp32.
 * ws: sept_conv1_dsum_workspace_floats(H, W) floats; coef: SEPT_CONV1_COEF_FLOATS floats of scratch. */
size_t sept_conv1_dsum_workspace_floats(int H, int W);
int sept_conv1_backward_data_sum(const void* dy_pooled, const void* idx_u8, const float* x, const float* w_f32,
                                 const float* bias, const float* mean, const float* invstd, const float* gamma,
                                 const float* dropscale, const float* sums, double n_total, float* ws, float* coef,
                                 float* dxsum, int B, int H, int W, void* stream);
int sept_conv5x5_dgrad_bnsums_ext(const void* dy_out, const void* wt, void* dx_out /* stored masked */, const void* ext,
                                  const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                  const float* dropscale, float* partials, int B, int H, int W, int cin, int cout,
                                  void* stream);

/* y = a * x  (GradientReversalFunction.backward with a = -lambda, reversal_gradient.py:18-23) */
int sept_scale(const float* x, float a, float* y, long n, void* stream);
/* y[i] = value (zero gradients of conv biases in front of a train-mode BatchNorm; flat-buffer housekeeping) */
int sept_fill(float* y, float value, long n, void* stream);
/* dst[0 .. nbytes) = src, device to device, any dtype (the host-fed step's swap of a staged batch into the captured
 * graph's static input tensors, training_cloak_with_grl.py:125-132) */
int sept_copy_bytes(const void* src, void* dst, long nbytes, void* stream);
/* y = x * m  (GRU inter-layer dropout with a pre-scaled mask) */
int sept_mul(const float* x, const float* m, float* y, long n, void* stream);
/* diagnostics: *slot = the device's 100 MHz wall clock when `stream` reaches this launch (also inside a graph replay) */
int sept_debug_stamp(long long* slot, void* stream);
/* Measurement aid: arm the NEXT instrumented launch of the calling thread (the sept_conv5x5_* entries: forward /
 * data-gradient forms and the weight gradient's main kernel) with a ZEROED region of SEPT_KCLOCK_WG * SEPT_KCLOCK_STRIDE
 * device int64 slots: workgroup g stores the 100 MHz device wall clock at its start into slots[g * STRIDE] and each of its
 * waves w (< STRIDE - 1) at its end into slots[g * STRIDE + 1 + w].  (max over the end slots - min over the non-zero start
 * slots) * 10 ns is the launch's duration on the device, also when the launch is a node of a replayed HIP graph (the pointer
 * is a kernel argument: arm while capturing).  Workgroups beyond SEPT_KCLOCK_WG are not recorded.  NULL disarms. */
#define SEPT_KCLOCK_WG 4096
#define SEPT_KCLOCK_STRIDE 16
int sept_kclock_next(long long* slots);
/* out = x + y (the two branch losses of the hand-scheduled GRL step: training_cloak_with_grl.py:160) */
int sept_add(const float* x, const float* y, float* out, long n, void* stream);
/* y = x * (*scalar_dev): scale by a value that lives on the device (no host read of a loss gradient). */
int sept_scale_dev(const float* x, const float* scalar_dev, float* y, long n, void* stream);
/* y = relu(x) * dropscale (nullable)  -- dense_relu1 + dropout, baseline_models.py:248-249 */
int sept_relu_dropout_forward(const float* x, const float* dropscale, float* y, long n, void* stream);
int sept_relu_dropout_backward(const float* dy, const float* x, const float* dropscale, float* dx, long n,
                               void* stream);
/* z[B][D] = mean over T of x[B][T][D]  (torch.mean(x, dim=1), baseline_models.py:232) */
int sept_mean_t_forward(const float* x, float* z, int B, int T, int D, void* stream);
int sept_mean_t_backward(const float* dz, float* dx, int B, int T, int D, void* stream);
/* out[N] (+)= column sums of a[M][lda]  (bias gradients); ws holds sept_colsum_workspace_floats(N) */
size_t sept_colsum_workspace_floats(int N);
int sept_colsum(const float* a, long lda, int M, int N, float* ws, float* out, int accumulate, void* stream);
/* loss (+)= scale * sum_i w_i CE(logits_i, labels_i);  dlogits = scale * w_i * (softmax - onehot)
 * (the per-sample loop of train(), training_cloak_with_grl.py:143-154; weights nullable) */
int sept_cross_entropy(const float* logits, const long long* labels, const float* weights, float scale, int B,
                       int C, float* loss, float* dlogits, int accumulate, void* stream);
/* Sliding-window inference (training_cloak_with_grl.py:70-87, adversary_cloak_evaluation.py:40-110):
 * probs[b] = mean over the nwin windows of utterance b of softmax(logits[b*nwin + i]); pred[b] =
 * argmax (first maximum).  pred nullable. */
int sept_softmax_mean(const float* logits, int B, int nwin, int C, float* probs, long long* pred, void* stream);
/* loss -= lambda * log(*mean)  (training_cloak_with_grl.py:158-160) */
int sept_loss_sub_log(float* loss, const float* mean, float lambda, void* stream);
/* dst[n][w*C + c] = src[n][c*Wd + w] (inverse != 0: the other way): GRU weight_ih_l0 between
 * the reference's (c, w) feature order (cloak_models.py:166-168) and the NHWC (w, c) order */
int sept_permute_cols(const float* src, float* dst, int N, int C, int Wd, int inverse, void* stream);
/* Operands of the GRU input projections of one layer in one launch: wcat (2G, K) = [W_ih; W_ih_reverse]
 * (layer 0: columns permuted from the reference's (c, w) feature order to NHWC (w, c), C*Wd == K;
 * C == 0 copies), its transpose wcatT (K, 2G) for dx = dgi W, and bcat (2G) = [b_ih; b_ih_reverse].
 * G = 3*hidden.  baseline_models.py:191-193, cloak_models.py:166-168. */
int sept_gru_pack(const float* w_fwd, const float* w_rev, const float* b_fwd, const float* b_rev, int G, int K,
                  int C, int Wd, float* wcat, float* wcatT, float* bcat, void* stream);
/* one_d_cnn_lstm (baseline_models.py:47-62), channels-last fp32 [B][T][C]:
 * Conv1d(k=5, pad=2) = sept_gemm on the unfolded input col[B][T][5*C] (col[..][k*C+c] = x[t+k-2][c]);
 * its data gradient folds dcol back; ReLU + MaxPool1d(pool) + Dropout forward/backward with the
 * arg-max byte map idx[B][T/pool][C] (first maximum). */
int sept_unfold1d(const float* x, float* col, int B, int T, int C, void* stream);
int sept_fold1d(const float* dcol, float* dx, int B, int T, int C, void* stream);
int sept_relu_pool1d_forward(const float* x, const float* dropscale, float* y, unsigned char* idx, int B, int T,
                             int C, int pool, void* stream);
int sept_relu_pool1d_backward(const float* dy, const float* x, const float* dropscale, const unsigned char* idx,
                              float* dx, int B, int T, int C, int pool, void* stream);
/* Counter-based RNG (Philox4x32-10): element i of a draw is a function of (seed, offset, i) only.
 * The effective stream offset is `offset + *offset_dev` (offset_dev nullable): a device-resident
 * step counter (advanced with sept_counter_add) lets a captured graph draw fresh numbers on every
 * replay, and equal (seed, offset) on every rank gives the ONE epsilon per step the reference
 * broadcasts over the batch (cloak_models.py:37,45-50).
 * sept_dropout_mask: 0 with probability p else 1/(1-p) (nn.Dropout / Dropout2d / GRU dropout
 * scale masks, baseline_models.py:153,176,193);  sept_normal: N(mean, std) (Normal(0, 0.1)). */
int sept_dropout_mask(float* out, long n, float p, unsigned long long seed, const long long* offset_dev,
                      unsigned long long offset, void* stream);
int sept_normal(float* out, long n, float mean, float stdv, unsigned long long seed, const long long* offset_dev,
                unsigned long long offset, void* stream);
int sept_counter_add(long long* counter, long long inc, void* stream);
int sept_counter_add2(long long* counter0, long long* counter1, long long inc, void* stream);   /* both, one launch */
/* Fused classifier head for the common case (att None, mean pooling, no global features; baseline_models.py:231-258):
 * z = mean_t(x (B, T, D)); d1 = z W1^T + b1; d1a = relu(d1) * dropscale (NULL = 1); logits (B, NC) = d1a Wh^T + bh, Wh
 * (NC, D1) being the prediction layer (both layers stacked for pred='multitask').  z, d1, d1a are kept for the
 * backward pass; sept_head_backward returns dd1 = dL/dd1 (operand of the weight gradients, which stay GEMMs) and
 * dx (B, T, D) = dL/dx.  D, D1 <= 256. */
int sept_head_forward(const float* x, const float* W1, const float* b1, const float* dropscale,
                      const float* Wh, const float* bh, float* z, float* d1, float* d1a, float* logits, int B,
                      int T, int D, int D1, int NC, void* stream);
int sept_head_backward(const float* dlogits, const float* Wh, const float* d1, const float* dropscale,
                       const float* W1, float* dd1, float* dx, int B, int T, int D, int D1, int NC,
                       void* stream);
/* sept_head_backward with the incoming gradient formed in the kernel: dlogits = scale * w_b * (softmax(logits_b) - onehot(label_b)),
 * sept_cross_entropy's expression (training_cloak_with_grl.py:143-151: the weighted cross-entropy of one prediction layer,
 * NC <= 8 classes); `dlogits` (B, NC) is written for the head's weight gradients.  The loss VALUE still comes from
 * sept_cross_entropy (which may then run anywhere behind the head's forward pass, off the backward chain). */
int sept_head_backward_ce(const float* logits, const long long* labels, const float* weights, float scale,
                          const float* Wh, const float* d1, const float* dropscale, const float* W1, float* dlogits,
                          float* dd1, float* dx, int B, int T, int D, int D1, int NC, void* stream);

/* Multi-head self-attention pooling of two_d_cnn_lstm with att='self_att' (baseline_models.py:233-242,
 * cloak_models.py:178-186): scores (B, T, NH) = att_linear2(tanh(att_linear1(x))) come from sept_gemm +
 * sept_tanh_forward; sept_att_pool_forward takes the softmax over T per head (probs, kept for the backward
 * pass) and z[b] = mean_h sum_t probs[b][t][h] x[b][t][:];  the backward entry returns the direct term of
 * dx and dscores.  T <= 1024. */
int sept_tanh_forward(const float* x, float* y, long n, void* stream);
int sept_tanh_backward(const float* dy, const float* y, float* dx, long n, void* stream);
int sept_att_pool_forward(const float* scores, const float* x, float* probs, float* z, int B, int T, int NH,
                          int D, void* stream);
int sept_att_pool_backward(const float* dz, const float* x, const float* probs, float* dx, float* dscores,
                           int B, int T, int NH, int D, void* stream);

/* MFCC pieces (audio_feature_extraction.py:15-26: torchaudio MFCC(16000, n_mfcc=40) on the audio and
 * on numpy.gradient(audio) with spacing 1 and 2): the mel front end is sept_mel_forward with
 * (n_fft 400, hop 200, 128 mels); sept_topdb_clamp applies AmplitudeToDB's top_db = 80 per clip
 * (x = max(x, max(x) - top_db)); the ortho DCT-II is a sept_gemm with the (128, 40) DCT matrix;
 * sept_transpose_last2 brings (B, T, 40) to the reference's (B, 40, T);  sept_gradient1d is
 * numpy.gradient along the last axis (central differences, one-sided at the ends). */
int sept_topdb_clamp(float* x, int B, long n_per, float top_db, void* stream);
int sept_gradient1d(const float* x, float* g, int B, long L, float spacing, void* stream);
int sept_transpose_last2(const float* in, float* out, int B, int R, int C, void* stream);
/* Windowing + per-speaker z-normalisation between the two halves of the path: mel (B, T, F)
 * time-major -> out (B*nwin, win, F), window i = frames [shift*i, shift*i + win) (zero padded
 * past T), each value (x - mean[f]) / (std[f] + 1e-5) when mean/std are given
 * (preprocess_adversary_data.py:30-35,131,377-378; training_cloak_with_grl.py:71). */
int sept_window_norm(const float* mel_btf, const float* mean, const float* stdv, float* out, int B, int T, int F,
                     int win, int shift, int nwin, void* stream);
/* torchaudio.transforms.Resample(orig, new) as used for MSP-Improv (audio_feature_extraction.py:139-141):
 * polyphase windowed-sinc FIR.  orig / newf are the two rates divided by their gcd; ker (newf, 2*width + orig)
 * is the host-built table (sept_amd/resample.py restates torchaudio's _get_sinc_resample_kernel);
 * out (B, target), target = ceil(L * newf / orig). */
int sept_resample_forward(const float* x, const float* ker, float* out, int B, long L, int orig, int newf,
                          int width, long target, void* stream);
/* Per-speaker normalisation and class-balance augmentation of the preprocessing step
 * (preprocess_adversary_data.py:356-423).  sept_speaker_stats: stats (S, 4, F) = {mean, std (population),
 * min, max} per mel bin over ALL frames of the clips of each speaker (spk (B) int32, NULL = one speaker),
 * accumulated in float64 like numpy; ws = sept_speaker_stats_workspace_doubles(B, F) doubles.
 * sept_window_norm_spk: windows of clip b normalised with the statistics of spk[b]; mode 0:
 * (x - mean) / (std + 1e-5), mode 1: (x - min) / (max - min) * 2 - 1; frames past T are zero BEFORE the
 * normalisation, as the reference pads.  sept_add_normal: out = x + Normal(0, stdv) (Philox), the
 * augmentation noise of :416-417. */
size_t sept_speaker_stats_workspace_doubles(int B, int F);
int sept_speaker_stats(const float* mel_btf, const int* spk, int B, int T, int F, int S, double* ws,
                       float* stats, void* stream);
/* The reference's OWN statistics population (preprocess_adversary_data.py:26-27 appends every row of every SAVED
 * item): a clip saved as nwin windows counts frame t once per window containing it and never the frames behind the
 * last window; a clip shorter than win, or a test-split speaker's clip (saved whole, once: :55-59, whole_clip[b]
 * != 0), counts every frame once.  lengths[b] <= T = valid frames of clip b (NULL: T); same outputs / workspace as
 * sept_speaker_stats, which weights every frame 1. */
int sept_speaker_stats_windows(const float* mel_btf, const int* spk, const int* lengths,
                               const unsigned char* whole_clip, int B, int T, int F, int S, int win, int shift,
                               double* ws, float* stats, void* stream);
int sept_window_norm_spk(const float* mel_btf, const float* stats, const int* spk, int mode, float* out,
                         int B, int T, int F, int win, int shift, int nwin, void* stream);
int sept_add_normal(const float* x, float* out, long n, float stdv, unsigned long long seed,
                    const long long* offset_dev, unsigned long long offset, void* stream);
/* torch.optim.SGD(momentum, weight_decay) / torch.optim.Adam(betas, eps, weight_decay) on a flat
 * parameter buffer (training_cloak_with_grl.py:416-421); grad_scale multiplies g first (1/world
 * for averaged data-parallel gradients).  Adam `step` counts from 1. */
int sept_sgd_step(float* p, const float* g, float* momentum_buf, long n, float lr, float momentum,
                  float weight_decay, int first_step, float grad_scale, void* stream);
int sept_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, void* stream);
/* The same updates with the learning rate (and Adam's step count, >= 1, kept current by sept_counter_add) read
 * from device memory: the optimiser can then be part of a captured HIP graph, and a scheduler -- StepLR(10, 0.5)
 * for SGD, ReduceLROnPlateau for Adam, training_cloak_with_grl.py:418,421 -- drives it by writing *lr_dev between
 * replays.  momentum_buf must start as zeros (torch's first SGD step, buf = d, then falls out of the recurrence). */
int sept_sgd_step_dev(float* p, const float* g, float* momentum_buf, long n, const float* lr_dev, float momentum,
                      float weight_decay, float grad_scale, void* stream);
int sept_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float beta1,
                       float beta2, float eps, float weight_decay, const long long* step_dev, float grad_scale,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEPT_H */
