/* sed_hip.h — C ABI of libsed_hip.so: the MI355X (gfx950) hot path of
 * qiuqiangkong/sound_event_detection_dcase2017_task4 (log-mel front-end fused into the
 * Cnn_9layers_* / Cnn_9layers_Gru_FrameAtt train / inference step).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + sizes + a HIP stream; no torch / C++ types.
 *   - every function enqueues on `stream` and never synchronises; returns 0, a hipError_t (>0),
 *     or -22 (bad argument).
 *   - activations are NHWC: [B][H=time][W=mel][C] fp32, C contiguous.  Conv weights are exchanged in the
 *     reference's OIHW layout ((Cout,Cin,3,3), state_dict compatible); packed copies are internal.
 *   - "partials" buffers are fp32 scratch written by one kernel and merged (in fp64) by the next.
 *
 * Each entry cites the reference code it replaces (paths relative to the reference repo).
 */
#ifndef SED_HIP_H
#define SED_HIP_H

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
typedef hipStream_t sed_stream_t;
#else
typedef void* sed_stream_t; /* a hipStream_t */
#endif

#ifdef __cplusplus
extern "C" {
#endif

const char* sed_version(void);

/* ---- K1 log-mel front-end -------------------------------------------------------------------------------
 * Replaces torchlibrosa Spectrogram + LogmelFilterBank as constructed at pytorch/models.py:251-258 and called
 * at :284-285 (reflect pad 512, Hann-windowed 1024-point DFT, hop 320, power, 513x64 Slaney mel matrix,
 * 10*log10(clamp(., amin))).  wave [B2][L] -> out [B2][T = L/320 + 1][64].
 * window[1024] = conv_real.weight[0,0,:] (the Hann window).  The 1024-point FFT of a frame PAIR (z = a + i b) is factored
 * 32 x 32 (n = 32 n1 + n2, k = k1 + 32 k2): tw1024t [32][32] float2 = exp(-2*pi*i*n2*k1/1024) laid out [k1][n2].
 * mel_tasks [n_tasks][4] int32 = {first bin of a 12-bin window, taps (<= 12), offset into mel_w, band}: the non-zero run of
 * each melW column is cut into <= 12-bin chunks; task slot s (s = lane + 64 r of the kernel) reads the (Pa, Pb) pairs of
 * its window and the host places the tasks so that the 32 slots of every group s/32 start at bins that differ mod 32 (bank
 * conflict-free LDS reads); empty slots have taps = 0 (n_tasks <= 128).  mel_bands [64][4] int32 = the task slots of each band
 * (-1 = none), max_band_tasks <= 4 = the largest count; mel_w = the task weights (mel_nnz <= 2048 floats).
 * The i16 variant folds utils/utilities.py:66-67 (int16 / 32767) into the window. */
int sed_logmel_f32(const float* wave, int B2, int L, const float* window, const float* tw1024t, const int* mel_tasks,
                   int n_tasks, const int* mel_bands, int max_band_tasks, const float* mel_w, int mel_nnz, float amin,
                   float* out, sed_stream_t stream);
int sed_logmel_i16(const short* wave, int B2, int L, const float* window, const float* tw1024t, const int* mel_tasks,
                   int n_tasks, const int* mel_bands, int max_band_tasks, const float* mel_w, int mel_nnz, float amin,
                   float* out, sed_stream_t stream);

/* ---- BatchNorm statistics (nn.BatchNorm2d, models.py:87-88, :264; eps 1e-5, momentum 0.1) -------------------
 * sed_chan_stats: per-channel (sum, M2) partials of x [N][C] in tiles of sed_stats_rows_per_part() rows;
 * partials [ceil(N/rows)][2][C].  sed_bn_finalize merges partials (from sed_chan_stats, sed_conv1_fwd or
 * sed_conv3x3_igemm epi 1) into mean / invstd / folded scale = gamma*invstd, shift = beta - mean*scale and
 * updates the running statistics (unbiased variance).  ws: >= 2048*C doubles.  rows_per_part = -1: the parts hold
 * varying row counts, given as nparts floats appended after the [nparts][2][C] partials (sed_conv3x3_wino2).
 * guard_dev / guard_host (nullable; the found-non-finite words of the split-f16 path, see sed_adam_amsgrad): batch
 * statistics that are NaN / inf raise them; null = torch semantics.  cand (nullable, [2][C]): the new running statistics
 * are written THERE instead of in place, and sed_bn_commit (n <= 16 BatchNorms per launch) installs them unless
 * *guard_dev != 0: a forward pass that met a non-finite value anywhere leaves every BatchNorm buffer untouched.  Either way
 * `cand` then holds the statistics from BEFORE the step, and sed_bn_restore (same arguments, launched behind the optimiser
 * step) copies them back when *guard_dev != 0 by then: a step refused because of its backward pass, its all-reduced
 * gradient or another rank's flag leaves the BatchNorm buffers as intact as the parameters.
 * sed_bn_eval_affine: eval mode, fold the running statistics instead. */
int sed_chan_stats(const float* x, long N, int C, float* partials, sed_stream_t stream);
int sed_stats_rows_per_part(void);
int sed_bn_finalize(const float* partials, int nparts, int rows_per_part, long N, int C, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                    float* mean_out, float* invstd_out, float* scale_out, float* shift_out, double* ws,
                    int* guard_dev, int* guard_host, float* cand,
                    const float* y_amax /* nullable */, float* act_bound_out /* nullable [64]: what sed_act_bound computes, from
                                                                                the same launch */,
                    const float* minmax /* nullable [nparts][2][C]: per-part (max, min) of y, same parts as `partials` */,
                    float* act_amax_out /* nullable [64]: what sed_act_amax(minmax, nparts, C, scale_out, shift_out) computes -- the
                                           operand amax of the NEXT convolution -- from the same launch up to 512 parts, by a
                                           follow-up launch beyond */,
                    sed_stream_t stream);
int sed_bn_commit(int n, float* const* cand, float* const* running_mean, float* const* running_var, const int* C,
                  const int* guard_dev, sed_stream_t stream);
int sed_bn_restore(int n, float* const* cand, float* const* running_mean, float* const* running_var, const int* C,
                   const int* guard_dev, sed_stream_t stream);
int sed_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* mean_out, float* invstd_out, float* scale_out,
                       float* shift_out, sed_stream_t stream);
/* BN backward, stage 2: partials [nparts][2][C] = (sum dy, sum dy*xhat) -> dgamma, dbeta and (nullable) the
 * coefficients coef[3][C] of g_y = a*dy + b*y + c.  batch_stats = 1: training-mode BN (gradient also flows
 * through the batch mean/variance); 0: eval-mode BN (fixed affine: b = c = 0). */
int sed_bn_bwd_finalize(const float* partials, int nparts, long N, int C, const float* mean, const float* invstd,
                        const float* scale, int batch_stats, float* dgamma, float* dbeta, float* coef, double* ws,
                        const float* y_amax, const float* g_amax, float ginv,
                        float* bound_out /* nullable [64]: what sed_grad_bound computes, from the same launch: with y_amax (and
                                            minmax = NULL) the bound over |y| <= amax; with minmax (and y_amax = NULL) over each
                                            channel's own range -- up to 512 parts inside this launch, by a follow-up launch beyond */,
                        const float* minmax /* nullable [nparts][2][C]: per-part (max, min) of y, same parts as `partials` */,
                        sed_stream_t stream);
/* g_y = a*dy + b*y + c in place on dy [nrows][C].  amax_out (nullable, device): receives max |g_y| (the split-f16
 * convolution that consumes the tensor takes its scale from it; zeroed and accumulated in stream order). */
int sed_bn_bwd_apply(float* dy_inout, const float* y, long nrows, int C, const float* coef, float* amax_out,
                     sed_stream_t stream);

/* ---- bn0 + SpecAugmentation + mixup (models.py:287-296; do_mixup pytorch_utils.py:80-93) ----------------------
 * logmel [B2][T][64] -> out [B2 or B2/2][T][64].  stripes [B2][8] = {t_bgn0,t_len0,t_bgn1,t_len1,f_bgn0,f_len0,
 * f_bgn1,f_len1} (null: no SpecAugment, eval); lam [B2] (null: no mixup).  The backward only produces the
 * (sum dy, sum dy*xhat) partials [ceil(B2*T/256)][2][64] for dgamma0/dbeta0 (the waveform takes no gradient). */
int sed_bn0_aug_mix_fwd(const float* logmel, int B2, int T, const float* scale, const float* shift,
                        const int* stripes, const float* lam, float* out, sed_stream_t stream);
int sed_bn0_aug_mix_bwd(const float* logmel, const float* g_out, int B2, int T, const float* mean,
                        const float* invstd, const int* stripes, const float* lam, float* partials, int* nparts_out,
                        sed_stream_t stream);

/* ---- ConvBlock tail: BN (folded) + ReLU + avg_pool2d (models.py:102-107), and torch.mean(dim=3) (:303) as the
 * (1, W) pool of block 4.  y [B][H][W][C] raw conv output -> out [B][H/ph][W/pw][C] (floor mode).
 * Backward: pass 1 reduces (sum dy, sum dy*xhat) partials [ceil(B*H*W / sed_pool_bwd_rows_per_block(B*H*W))][2][C];
 * pass 2 writes
 * g_y = a*dy + b*y + c with dy = relu-mask * g_out/(ph*pw). */
int sed_bn_relu_pool_fwd(const float* y, int B, int H, int W, int C, int ph, int pw, const float* scale,
                         const float* shift, float* out, float* amax_out /* nullable, device: max of `out` */,
                         sed_stream_t stream);
/* forward that also writes cnt [B][H/ph][W/pw][C] bytes = how many of the ph*pw window inputs passed the ReLU; with
 * it and the pooled output, backward pass 1 runs at POOLED resolution (sum dy = sum g*cnt/n, sum dy*xhat =
 * sum g*(p - beta*cnt/n)/gamma, n = ph*pw) and never touches y.  The 1/gamma amplifies fp32 rounding of p: use the
 * full-resolution sed_bn_relu_pool_bwd_reduce when some |gamma| is small. */
int sed_bn_relu_pool_fwd_cnt(const float* y, int B, int H, int W, int C, int ph, int pw, const float* scale,
                             const float* shift, float* out, unsigned char* cnt, float* amax_out /* nullable */,
                             sed_stream_t stream);
/* The same three stages for ANY pool_type of ConvBlock.forward (models.py:104-111): pool_mode 0 = 'avg' (what every model
 * selects; identical to the entry points above), 1 = 'max' (F.max_pool2d: the gradient goes to the first maximum of a
 * window), 2 = 'avg+max'. */
int sed_bn_relu_pool_fwd_mode(const float* y, int B, int H, int W, int C, int ph, int pw, int pool_mode, const float* scale,
                              const float* shift, float* out, float* amax_out, sed_stream_t stream);
int sed_bn_relu_pool_bwd_reduce_mode(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                                     int pool_mode, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, float* partials, int* nparts_out, sed_stream_t stream);
int sed_bn_relu_pool_bwd_apply_mode(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                                    int pool_mode, const float* scale, const float* shift, const float* coef, float* gy,
                                    float* amax_out, sed_stream_t stream);
/* Gradients as split-f16 operand pairs (round 4; format: see sed_conv1_act_sf16).  sed_grad_bound: upper bound of
 * max |a*dy + b*y + c| (coef [3][C] of sed_bn_bwd_finalize) from the per-part per-channel range of y (minmax [nparts][2][C], as
 * left by the conv epilogues) and the amax vector of the incoming gradient (times ginv = 1 / pool window); the two apply
 * kernels then write their result as pairs scaled by the power of two of that bound (bound = the vector their consumers
 * take as gy_amax / x_amax together with the pairs flag).  models.py:102-107 backward. */
int sed_grad_bound(const float* minmax, int nparts, int C, const float* coef, const float* g_amax, float ginv, float* bound_out,
                   const float* y_amax /* instead of minmax (exactly one of the two): |y| <= its amax, every channel */,
                   sed_stream_t stream);
int sed_bn_bwd_apply_pairs(float* dy_inout, const float* y, long nrows, int C, const float* coef, const float* bound,
                           sed_stream_t stream);
int sed_bn_relu_pool_bwd_apply_pairs(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                                     const float* scale, const float* shift, const float* coef, void* gy_pairs,
                                     const float* bound, sed_stream_t stream);
/* The pooled block output as operand pairs (round 4): sed_act_bound = max_c max(|scale_c| * amax|y| + shift_c, 0) >= every
 * average of relu(scale*y + shift); sed_bn_relu_pool_fwd_cnt_pairs = sed_bn_relu_pool_fwd_cnt writing pairs scaled by it. */
int sed_act_bound(const float* y_amax, const float* scale, const float* shift, int C, float* bound_out, sed_stream_t stream);
int sed_bn_relu_pool_fwd_cnt_pairs(const float* y, int B, int H, int W, int C, int ph, int pw, const float* scale,
                                   const float* shift, void* out_pairs, unsigned char* cnt, const float* bound,
                                   sed_stream_t stream);
/* amax of a = relu(scale*y + shift) -- the operand `conv2` of a ConvBlock (models.py:102-103) consumes without it ever
 * being materialised -- for the split-f16 scale of that convolution.  sed_act_amax: from per-part per-channel (max, min)
 * of y, minmax [nparts][2][C], which sed_conv1_fwd / sed_conv3x3_sf16 leave beside their statistics (the affine + ReLU
 * is monotone, so the range ends give the exact amax; scale = shift = null: max |y|).  sed_act_amax_full: the same by one
 * pass over y [nrows][C], for producers without range partials. */
int sed_act_amax(const float* minmax, int nparts, int C, const float* scale, const float* shift, float* amax_out,
                 sed_stream_t stream);
int sed_act_amax_full(const float* y, long nrows, int C, const float* scale, const float* shift, float* amax_out,
                      sed_stream_t stream);
int sed_bn_relu_pool_bwd_reduce_win(const float* g_out, const float* pooled, const unsigned char* cnt, long Mp, int C,
                                    int window, const float* gamma, const float* beta, float* partials,
                                    int* nparts_out, sed_stream_t stream);
/* pass 1 with the windowed / exact choice made on the device from this step's gamma (windowed when every |gamma[c]| >=
 * gamma_min; the other kernel returns at once).  partials: sed_bn_relu_pool_bwd_reduce_auto_parts(...) * 2*C floats,
 * zeroed by the call; *nparts_out = that part count. */
long sed_bn_relu_pool_bwd_reduce_auto_parts(int B, int H, int W, int ph, int pw);
int sed_bn_relu_pool_bwd_reduce_auto(const float* y, const float* g_out, const float* pooled, const unsigned char* cnt,
                                     int B, int H, int W, int C, int ph, int pw, const float* scale,
                                     const float* shift, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, float gamma_min, float* partials, int* nparts_out,
                                     const float* pooled_bound /* nullable: `pooled` holds operand pairs scaled by it */,
                                     sed_stream_t stream);
int sed_pool_bwd_rows_per_block(long M);
int sed_bn_relu_pool_bwd_reduce(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                                const float* scale, const float* shift, const float* mean, const float* invstd,
                                float* partials, int* nparts_out, sed_stream_t stream);
int sed_bn_relu_pool_bwd_apply(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                               const float* scale, const float* shift, const float* coef, float* gy,
                               float* amax_out /* nullable: max |gy|, as in sed_bn_bwd_apply */, sed_stream_t stream);

/* ---- 3x3 convolution, stride 1, pad 1, no bias (nn.Conv2d at models.py:77-85) on fp32 MFMA ---------------------
 * sed_pack_conv_weights: OIHW -> wf [9][Cout][Cin] (forward operand) and wd [9][Cin][Cout], taps flipped (dgrad
 * operand); either may be null.
 * sed_conv3x3_igemm: y [B*H*W][Cout] = conv(x [B*H*W][Cin], w_packed [9][Cout][Cin]).  Forward: w_packed = wf.
 * Dgrad: x = g_y, Cin/Cout swapped, w_packed = wd.
 *   in_scale/in_shift (nullable): the operand is relu(in_scale*x + in_shift) computed on the fly (the
 *     preceding BN+ReLU is never materialised).
 *   epi 0 plain; epi 1 also writes BN statistic partials [sed_conv_num_parts][2][Cout] of y (rows per part =
 *     sed_conv_rows_per_part(M, Cout)); epi 2 (dgrad) masks y by relu'(p_scale*yprev + p_shift) and writes
 *     (sum dy, sum dy*xhat) partials for the BN backward of the previous layer.
 * sed_conv3x3_wgrad: dw (OIHW) = sum_p gy[p][co] * a[p + tap][ci]; partial: scratch of
 *   sed_wgrad_partial_floats(B*H*W, Cin, Cout, 9, ...) floats.
 * sed_conv1_*: conv_block1.conv1 (Cin = 1), HBM-bound direct kernels; partials [ceil(M/256)][2][64];
 *   scratch dw_partials sed_conv1_bwd_partial_floats(B, H, W) floats, tbuf 9*M floats (only when gx0 != null).  sed_conv1_bwd with
 *   bn_y / bn_coef non-null treats gy as the masked dgrad output dz and applies the BatchNorm backward
 *   g = a*dz + b*bn_y + c (coef [3][64] of sed_bn_bwd_finalize) on load, replacing a sed_bn_bwd_apply pass. */
int sed_pack_conv_weights(const float* w_oihw, int Cout, int Cin, float* wf, float* wd, sed_stream_t stream);
int sed_conv_rows_per_part(long M, int Cout);
int sed_conv_num_parts(long M, int Cout);
int sed_conv3x3_igemm(const float* x, const float* w_packed, float* y, int B, int H, int W, int Cin, int Cout,
                      const float* in_scale, const float* in_shift, int epi, float* partials, const float* yprev,
                      const float* p_scale, const float* p_shift, const float* p_mean, const float* p_invstd,
                      sed_stream_t stream);
/* Fused 2-D Winograd F(2x2,3x3) variant (16 instead of 36 MACs per 2x2 output tile): same fusions and contract, with
 * w_wino2 = the k-step-major pack [Cin/8][16][Cout][8] from sed_pack_conv_weights_wino2 (uf: forward, ud: dgrad with
 * Cin/Cout swapped).  Statistics parts: P = sed_conv_wino2_num_parts(B,H,W), one per wave (<= 64 pixels);
 * partials = [P][2][Cout] floats followed (epi 1) by P per-part pixel counts -> pass rows_per_part = -1 to
 * sed_bn_finalize.  Needs W in {8,16,32,64}, Cin % 8 == 0, Cout % 32 == 0 (sed_conv3x3_wino2_supported). */
int sed_conv3x3_wino2_supported(int H, int W, int Cin, int Cout);
long sed_conv_wino2_num_parts(int B, int H, int W);
int sed_pack_conv_weights_wino2(const float* w_oihw, int Cout, int Cin, float* uf, float* ud, sed_stream_t stream);
int sed_conv3x3_wino2(const float* x, const float* w_wino2, float* y, int B, int H, int W, int Cin, int Cout,
                      const float* in_scale, const float* in_shift, int epi, float* partials, const float* yprev,
                      const float* p_scale, const float* p_shift, const float* p_mean, const float* p_invstd,
                      sed_stream_t stream);
/* 2-D Winograd-domain weight gradient (16 instead of 36 MACs per 2x2 tile); same contract as sed_conv3x3_wgrad.
 * Needs W in {8,16,32,64}, Cin % 32 == 0, Cout % 64 == 0 (else -22 / 0 floats);
 * partial: sed_wgrad_wino2_partial_floats(...) floats. */
long sed_wgrad_wino2_partial_floats(int B, int H, int W, int Cin, int Cout, int* nslices_out, int* units_per_slice_out);
int sed_conv3x3_wgrad_wino2(const float* x, const float* gy, float* dw_oihw, float* partial, int B, int H, int W, int Cin,
                            int Cout, const float* in_scale, const float* in_shift, sed_stream_t stream);
/* 3x3 convolution (forward / dgrad) on the f16 MFMA pipe with SPLIT operands: x = (hi + lo)/s with hi, lo f16 and s a
 * power of two per tensor, three f16 MFMAs (hi*hi + hi*lo + lo*hi, exact products, fp32 accumulation) per product slab --
 * the error of a direct fp32 convolution at 3/16 of its MFMA issue time (csrc/conv_sf16.hip).  Same contract as
 * sed_conv3x3_wino2 (in_scale/in_shift operand transform, epi 0/1/2, partials = sed_conv_sf16_num_parts(...) parts -- one per
 * workgroup tile of 256 pixels x 64 channels -- with the pixel counts appended for epi 1).  wp: sed_conv_sf16_pack_halfs(...) f16 values and wscale[65] (64 amax slots, scale) written by
 * sed_pack_conv_weights_sf16 (dgrad = 1: operand of the transposed convolution).  Operand scale: x_amax = device pointer
 * to the amax of the operand as the MFMAs see it (of relu(in_scale*x + in_shift) when that transform is fused: sed_act_amax;
 * of x otherwise: sed_amax or the producer kernels' amax_out) -- the power of two that brings it to [2^13, 2^14) is
 * taken on the device, so a finite operand neither overflows nor loses its low half to f16 subnormals at ANY magnitude.
 * A non-finite operand stores 1 through err_host (nullable, host-mapped int) and err_dev (nullable, device int: the
 * skip_flag of sed_adam_amsgrad) -- never silent.  minmax (nullable, epi 0 / 1): per-part (max, min) of the outputs per
 * channel, [sed_conv_sf16_num_parts(...)][2][Cout], the input of sed_act_amax for the NEXT convolution.
 * Needs W in {8,16,32,64}, Cin % 16 == 0, Cout % 64 == 0. */
int sed_conv3x3_sf16_supported(int H, int W, int Cin, int Cout);
/* Split-K forms of sed_conv3x3_sf16 (same arguments and results up to summation order): `ksplit` workgroups share one output tile,
 * each over 1/ksplit of the K-steps; the last to arrive adds the others' accumulators (ws) in a fixed order and runs the epilogue.
 * Tiles [0, nfull) of the launch stay un-split (nfull % 8 == 0).  nfull = 0: the small-M form, for launches with fewer workgroups
 * than resident slots -- the reference's `--batch_size 32` spread over 8 GPUs (main.py:138) leaves 4 clips per GPU.  nfull = a
 * multiple of the 768 resident slots: the tail form -- only the workgroups of the last, partial round of the chip are split (the
 * 125 x 8 layers at the metric's own batch, 32 clips: 1024 workgroups = 1.33 rounds).  Measured slower than not splitting on every
 * production shape (profiles/r06/tail_split_ab.txt), so the library's own rule never selects it: tail_ks = 0 asks for that rule,
 * tail_ks = 2 .. 8 forces a tail split that many ways (A/B runs, tests).
 * sed_conv_sf16_split_plan: the split the library recommends (returns ksplit, 1 = none; *nfull); sed_conv_sf16_ksplit: its small-M
 * part alone; ws: sed_conv_sf16_splitk_floats(...) floats; tickets: sed_conv_sf16_splitk_tickets(...) ints, zero before the FIRST
 * use (every launch leaves them zero). */
int sed_conv_sf16_split_plan(int B, int H, int W, int Cin, int Cout, int tail_ks, int* nfull);
int sed_conv_sf16_ksplit(int B, int H, int W, int Cin, int Cout);
long sed_conv_sf16_splitk_floats(int B, int H, int W, int Cout, int ksplit, int nfull);
long sed_conv_sf16_splitk_tickets(int B, int H, int W, int Cout, int nfull);
int sed_conv3x3_sf16_splitk(const float* x, const void* wp, const float* wscale, float* y, int B, int H, int W, int Cin, int Cout,
                            const float* in_scale, const float* in_shift, int epi, float* partials, const float* yprev,
                            const float* p_scale, const float* p_shift, const float* p_mean, const float* p_invstd,
                            const float* x_amax, float* minmax, int* err_host, int* err_dev, int flags, float* out_amax,
                            int ksplit, int nfull, float* ws, int* tickets, sed_stream_t stream);
long sed_conv_sf16_pack_halfs(int Cin, int Cout);
long sed_conv_sf16_num_parts(int B, int H, int W, int Cout);
/* Device-resident amax values are float[sed_amax_slots()] (= 64), NOT one float: producers publish one atomic per block
 * into slot (block id mod 64) -- thousands of atomics on one word serialise in the L2 -- and consumers take the maximum over
 * the slots.  Every `amax_out` / `*_amax` pointer of this header is such a vector.  The entry points zero amax_out
 * themselves, except for pointers inside an address range registered with sed_amax_prezeroed_range(base, nfloats, 1)
 * (slices of a pool the caller zeroes with one fill: saves the per-launch memsets; on = 0 unregisters; <= 64 ranges).
 * wscale of the weight packs = [64 amax slots][1 power-of-two scale]. */
int sed_amax_slots(void);
int sed_amax_prezeroed_range(const float* base, long nfloats, int on);
int sed_amax(const float* x, long n, float* amax_out, sed_stream_t stream);
int sed_pack_conv_weights_sf16(const float* w_oihw, int Cout, int Cin, int dgrad, float* wscale, void* wp,
                               sed_stream_t stream);
/* The same for n <= 16 weights in TWO launches (one amax pass + one pack pass over all of them): HOST arrays of n device
 * pointers / shapes; dgrad[i] = 0 (forward layout), 1 (dgrad layout) or 4 (both, wp[i] = forward pack then dgrad pack).
 * Every ConvBlock weight of reference models.py:77-85 is packed once per optimiser step this way. */
int sed_pack_conv_weights_sf16_multi(int n, const float* const* w_oihw, const int* Cout, const int* Cin, const int* dgrad,
                                     float* const* wscale, void* const* wp, sed_stream_t stream);
/* Inference form of the second convolution of a ConvBlock (reference models.py:99-113 in eval mode):
 * out = avg_pool(relu(o_scale * conv(relu(in_scale * x + in_shift)) + o_shift)) with the eval-mode BatchNorm folded into
 * o_scale / o_shift (sed_bn_eval_affine) -- the full-resolution convolution output is never written.  Pooling (2, 2) for
 * W in {16, 32, 64} or (1, W) for W = 8; out = [B][H/ph][W/pw][Cout]; out_amax (nullable) receives the amax slots of out. */
int sed_conv3x3_sf16_eval_pool_supported(int H, int W, int Cin, int Cout, int ph, int pw);
int sed_conv3x3_sf16_eval_pool(const float* x, const void* wp, const float* wscale, float* out, int B, int H, int W,
                               int Cin, int Cout, const float* in_scale, const float* in_shift, const float* o_scale,
                               const float* o_shift, int ph, int pw, const float* x_amax, float* out_amax,
                               int* err_host, int* err_dev, sed_stream_t stream);
int sed_conv3x3_sf16(const float* x, const void* wp, const float* wscale, float* y, int B, int H, int W, int Cin,
                     int Cout, const float* in_scale, const float* in_shift, int epi, float* partials,
                     const float* yprev, const float* p_scale, const float* p_shift, const float* p_mean,
                     const float* p_invstd, const float* x_amax, float* minmax, int* err_host, int* err_dev,
                     int flags, float* out_amax /* nullable: amax vector of |y| as written */, sed_stream_t stream);
/* Block 1 without a materialised conv1 output (round 4).  models.py:99-103 for conv_block1: relu(bn1(conv1(x0))) feeds conv2.
 * sed_conv1_fwd(y = null) is the statistics / range pass; sed_conv1_act_sf16 then writes a1 = relu(scale*conv1(x0)+shift)
 * ONCE, as split-f16 operand pairs -- per channel pair two dwords {hi0 | hi1 << 16, lo0 | lo1 << 16}, hi = f16(s*a),
 * lo = f16(s*a - hi), s = the power-of-two scale of a_amax; B*H*W*64 dwords = the bytes of the fp32 tensor it replaces.
 * Consumers: sed_conv3x3_sf16(flags = 1: x is such a tensor, x_amax the amax it was written with; epi 0 / 1, no input
 * transform), sed_conv3x3_wgrad_sf16(flags = 1: likewise for its x operand).  The two backward kernels that need the RAW
 * y1 recompute it from the one-channel input with sed_conv1_fwd's fma sequence (bit-identical): sed_conv3x3_sf16_dgrad_b1 =
 * sed_conv3x3_sf16(epi = 2) of conv2 with yprev = conv1(x0) formed in the epilogue from an x0 patch in LDS (p_* = bn1's
 * folded scale / shift / mean / invstd), and sed_conv1_bwd(bn_y = null). */
int sed_conv1_act_sf16(const float* x0, const float* w_oihw, int B, int H, int W, const float* scale, const float* shift,
                       const float* a_amax, void* out_pairs, int* err_host, int* err_dev, sed_stream_t stream);
int sed_conv3x3_sf16_dgrad_b1(const float* gy, const void* wp, const float* wscale, float* gx, int B, int H, int W, int Cin,
                              int Cout, float* partials, const float* p_scale, const float* p_shift, const float* p_mean,
                              const float* p_invstd, const float* x0, const float* w1_oihw, const float* gy_amax,
                              int* err_host, int* err_dev, int flags /* bit 0: gy holds pairs */, float* out_amax,
                              sed_stream_t stream);
/* Weight gradient with split-f16 operands (csrc/conv_sf16.hip): same contract as sed_conv3x3_wgrad; gy_amax = device
 * pointer to max |gy| (sed_amax or the producer kernels), x_amax = device pointer to the amax of the activation operand
 * (as for sed_conv3x3_sf16); err_host / err_dev as there.
 * Needs W in {8,16,32,64}, Cin % 32 == 0, Cout % 64 == 0; partial: sed_wgrad_sf16_partial_floats(...) floats. */
int sed_wgrad_sf16_supported(int H, int W, int Cin, int Cout);
long sed_wgrad_sf16_partial_floats(int B, int H, int W, int Cin, int Cout);
int sed_conv3x3_wgrad_sf16(const float* x, const float* gy, float* dw_oihw, float* partial, int B, int H, int W,
                           int Cin, int Cout, const float* in_scale, const float* in_shift, const float* gy_amax,
                           const float* x_amax, int* err_host, int* err_dev, int flags, sed_stream_t stream);
long sed_wgrad_partial_floats(long M, int Cin, int Cout, int ntaps, int* nslices_out, int* pix_per_slice_out);
int sed_conv3x3_wgrad(const float* x, const float* gy, float* dw_oihw, float* partial, int B, int H, int W, int Cin,
                      int Cout, const float* in_scale, const float* in_shift, sed_stream_t stream);
int sed_conv1_fwd(const float* x0, const float* w_oihw, float* y, int B, int H, int W, float* partials,
                  float* minmax /* nullable: [ceil(M/rows)][2][64] per-part (max, min) per channel, for sed_act_amax */,
                  sed_stream_t stream);
int sed_conv1_rows_per_part(void);
long sed_conv1_bwd_partial_floats(int B, int H, int W);
int sed_conv1_bwd(const float* x0, const float* w_oihw, const float* gy, const float* bn_y, const float* bn_coef, int B,
                  int H, int W, float* dw, float* gx0, float* dw_partials, float* tbuf, sed_stream_t stream);

/* ---- dense fp32-MFMA GEMMs for fc (models.py:271,:308), AttBlock 1x1 convs (:124-125), nn.GRU projections (:529)
 * sed_gemm_nt: y[M][N] = x[M][K] * w[N][K]^T (+ bias[N]);  K % 32 == 0, N % 64 == 0.
 * sed_gemm_tn: dw[N][K] = sum_m gy[m][n] * x[m][k];  N, K % 64 == 0; partial: scratch of
 *   sed_wgrad_partial_floats(M, K, N, 1, ...) floats. */
int sed_gemm_nt(const float* x, const float* w, const float* bias, float* y, long M, int N, int K,
                sed_stream_t stream);
int sed_gemm_tn(const float* x, const float* gy, float* dw, float* partial, long M, int N, int K,
                sed_stream_t stream);
/* The same NT GEMM on the f16 MFMA pipe with split-f16 operands (csrc/gemm_sf16.hip; the arithmetic of sed_conv3x3_sf16: exact
 * products, fp32 accumulation -- the rounding error of an fp32 dot product): the nn.GRU input projections (models.py:529-530) and
 * their input gradient.  wp / wscale = the weight matrix w [N][K] pre-split by sed_gemm_pack_sf16 (sed_gemm_pack_sf16_halfs(N, K)
 * f16 values; wscale: float[sed_amax_slots() + 1] = its amax slots, then its power-of-two scale); x_amax: device amax vector of x
 * (sed_amax); err_host / err_dev: the found-non-finite words (nullable); out_amax (nullable): amax slots of |y| as written.
 * Needs N % 128 == 0, K % 32 == 0, M * K * 4 < 2^31 (sed_gemm_nt_sf16_supported); M may be ragged. */
int sed_gemm_nt_sf16_supported(long M, int N, int K);
long sed_gemm_pack_sf16_halfs(int N, int K);
int sed_gemm_pack_sf16(const float* w, int N, int K, float* wscale, void* wp, sed_stream_t stream);
int sed_gemm_nt_sf16(const float* x, const void* wp, const float* wscale, const float* bias, float* y, long M, int N, int K,
                     const float* x_amax, int* err_host, int* err_dev, float* out_amax, sed_stream_t stream);
/* ... and the TN form for the weight gradients of those layers: dw[N][K] = sum_m gy[m][n] * x[m][k] with BOTH operands converted
 * when staged (LDS transpose reads, as sed_conv3x3_wgrad_sf16); partial: sed_gemm_tn_sf16_partial_floats(M, N, K) floats of scratch
 * (row slices, reduced in fp64).  N % 128 == 0, K % 128 == 0, M * max(N, K) * 4 < 2^31. */
int sed_gemm_tn_sf16_supported(long M, int N, int K);
long sed_gemm_tn_sf16_partial_floats(long M, int N, int K);
int sed_gemm_tn_sf16(const float* x, const float* gy, float* dw, float* partial, long M, int N, int K, const float* x_amax,
                     const float* gy_amax, int* err_host, int* err_dev, sed_stream_t stream);
/* two independent NT GEMMs of one shape in one launch (the two directions of a BiGRU recurrence step) */
int sed_gemm_nt_pair(const float* x0, const float* x1, const float* w0, const float* w1, const float* bias0,
                     const float* bias1, float* y0, float* y1, long M, int N, int K, sed_stream_t stream);
int sed_reduce_rows(const float* parts, long n, int K, long ld, float* out, int accumulate, float* ws,
                    sed_stream_t stream);
int sed_transpose(const float* x, int batch, int rows, int cols, float* out, sed_stream_t stream);
int sed_axpy(float* out, const float* a, long n, sed_stream_t stream);
/* out[i] = x[i] * scalar_dev[0] (the chain rule through the 0-dim loss of losses.py:5-12: d loss / d p times the upstream gradient). */
int sed_scale_by_scalar(const float* x, const float* scalar_dev, long n, float* out, sed_stream_t stream);

/* ---- heads -----------------------------------------------------------------------------------------------
 * FrameAvg (models.py:306-312) / FrameMax (:221-227): logits [B][T][ldn] = feat x Wfc^T by sed_gemm_nt, then
 * frame = sigmoid(logits + bias), clip = mean_t (mode 0) or max_t (mode 1).
 * AttBlock (models.py:118-149): logits columns [0,ncls) = att, [ncls,2ncls) = cla pre-activations; clamp +-10,
 * exp(. / temperature) + 1e-6, normalise over time, cla = activation (0: 'linear', 1: 'sigmoid'; models.py:145-149), weighted
 * sum.  sed_interpolate = models.py:58-69 (x8 repeat). */
int sed_head_pool_fwd(const float* logits, int B, int T, int ldn, int ncls, const float* bias, int mode, float* frame,
                      float* clip, int* amax, sed_stream_t stream);
int sed_head_pool_bwd(const float* g_clip, const float* frame, const int* amax, int B, int T, int ldn, int ncls,
                      int mode, float* g_logits, sed_stream_t stream);
int sed_att_pool_fwd(const float* logits, int B, int T, int ldn, int ncls, const float* b_att, const float* b_cla,
                     float* clip, float* cla, float* norm_att, float* att_sum, int activation, float temperature,
                     sed_stream_t stream);
int sed_att_pool_bwd(const float* g_clip, const float* logits, const float* b_att, const float* clip,
                     const float* cla, const float* norm_att, const float* att_sum, int B, int T, int ldn, int ncls,
                     float* g_logits, int activation, float temperature, sed_stream_t stream);
int sed_interpolate(const float* x, long BT, int ncls, int ratio, float* out, sed_stream_t stream);

/* ---- nn.GRU gate math (models.py:529-530; PyTorch gate order r,z,n, b_hn inside r*(.)) -------------------------
 * Each call handles BOTH directions of one recurrence step (suffix 0 = forward, 1 = reverse; the two directions are at
 * different time indices, hence explicit pointer pairs).  gi = input projection incl. b_ih (row stride ld_gi), gh =
 * hidden projection incl. b_hh [B][3H]; h_out / h_out2: the new hidden state, written contiguously for the next step's
 * GEMM and into the (B,T,2H) output; save [B][4H] = r,z,n,gh_n.  Backward: dh = g_out + dh_direct + dh_gemm (both
 * nullable; the direct z-path and the W_hh path arriving from the later step) -> dgi, dgh, dh_direct_out = dh*z. */
int sed_gru_gate_fwd(const float* gi0, const float* gi1, long ld_gi, const float* gh0, const float* gh1,
                     const float* h_prev0, const float* h_prev1, int B, int Hd, float* h_out0, float* h_out1,
                     long ld_out, float* h_out2_0, float* h_out2_1, long ld_out2, float* save0, float* save1,
                     sed_stream_t stream);
int sed_gru_gate_bwd(const float* g_out0, const float* g_out1, long ld_go, const float* dh_direct0,
                     const float* dh_direct1, const float* dh_gemm0, const float* dh_gemm1, const float* save0,
                     const float* save1, const float* h_prev0, const float* h_prev1, int B, int Hd, float* dgi0,
                     float* dgi1, long ld_dgi, float* dgh0, float* dgh1, float* dh_direct_out0, float* dh_direct_out1,
                     sed_stream_t stream);

/* ---- nn.GRU recurrence, fused (models.py:529-530, :565-567): ONE persistent launch per pass runs all T steps of both
 * directions -- hidden projection h_prev x W_hh^T on the f16 MFMA pipe with split-f16 operands (the weight slice resident in
 * registers; fp32-level error), gate math, and a hand-over THROUGH THE DATA among the 4 workgroups sharing a (direction,
 * sed_gru_seq_row_block()-row block): hs / dgh are pre-filled with a sentinel NaN by the call and every consumer wave polls its operand block until no
 * sentinel is left (bounded spin, no grid-wide barrier).  Built for Hd = 256 and B <= 512 (all workgroups must be co-resident):
 * sed_gru_seq_supported; callers use the per-step GEMM + sed_gru_gate_* launches otherwise.
 * Layouts: gi [B][T][6H] (forward gates r,z,n then reverse gates, incl. b_ih), hs [2][T][B][H] hidden states,
 * saves = r,z,n,gh_n: sed_gru_seq_saves_floats(B, T) floats in a layout private to the two recurrences (tile-major, so that a
 * wave stores 1 KB runs), out [B][T][2H] = concat(forward, reverse).  Backward: g_out [B][T][2H];
 * wt_* = W_hh^T [H][3H]; produces dgi [B][T][6H] and dgh [2][T][B][3H] (gate pre-activation gradients on the input /
 * hidden side; weight and bias gradients are plain GEMMs / column sums over them).
 * ws: sed_gru_seq_ws_floats() floats of scratch (the give-up word; zeroed by the call).  dgi_amax (nullable): amax slots of |dgi|,
 * published by the backward recurrence (the operand scale of sed_gemm_nt_sf16 for the input gradient).
 * Run-time failure: the workgroups of a launch wait for each other, so all of them must be resident at once.
 * sed_gru_seq_supported also asks the CURRENT device (CU count, occupancy of both kernels) and answers 0 when they
 * cannot be; if a launch still cannot make progress (CU mask, co-tenant kernel), its bounded spin gives up and a
 * follow-up kernel overwrites `out` / `dgi` with NaN and stores 1 (forward) / 2 (backward) into *err_host, a
 * DEVICE-VISIBLE HOST int (hipHostMalloc / pinned; may be null) that the host can poll without synchronising.
 * (The give-up path and the agent-scope fallback are exercised through the test hooks of include/sed_hip_test.h.) */
int sed_gru_seq_supported(int B, int Hd);
long sed_gru_seq_ws_floats(void);
int sed_gru_seq_row_block(void);      /* batch rows per workgroup (16): the granularity of dbias_parts */
long sed_gru_seq_saves_floats(int B, int T);
int sed_gru_seq_fwd(const float* gi, const float* w_hh_f, const float* w_hh_b, const float* b_hh_f,
                    const float* b_hh_b, int B, int T, int Hd, float* hs, float* saves, float* out, float* ws,
                    int* err_host, sed_stream_t stream);
int sed_gru_seq_bwd(const float* g_out, const float* wt_f, const float* wt_b, const float* hs, const float* saves,
                    int B, int T, int Hd, float* dgi, float* dgh,
                    float* dbias_parts /* nullable: [2 directions][ceil(B/row_block)][4: dr, dz, dn, dn*r][Hd] sums over time and
                                          the rows of a block: db_ih = (dr, dz, dn), db_hh = (dr, dz, dn*r) summed over blocks */,
                    float* ws, int* err_host, float* dgi_amax, sed_stream_t stream);

/* ---- multi-head self-attention of the Transformer heads (models.py:587-665; 8 heads x 64) ----------------------
 * q, k, v, o, g_*: [B*T][512] fp32, head h in columns 64h..64h+63 (the Linear outputs of w_qs / w_ks / w_vs, no
 * permutes).  keep: attention-dropout KEEP mask, bytes [8*B][T][T] with row index h*B + b (the (n*b) layout of
 * :651-657), or null in eval mode; p_drop = 0.1 (:590).  stats: [B][8][T][4] floats (row max, row sum, D, -) written by
 * the forward and completed / consumed by the backward.
 *   sed_mha_fwd: O = dropout(softmax(Q K^T / 8)) V            sed_mha_bwd: g_q, g_k, g_v from g_o
 * T <= 128 runs on fp32-MFMA kernels (the whole score tile of a (clip, head) at once), longer sequences on vector kernels.
 * keep_bits: scratch of sed_mha_mask_words(B, T) 32-bit words (0 words for T > 128: pass null), required with a non-null keep
 * for T <= 128: sed_mha_fwd packs the mask into it (dropped bits per query over the keys and per key over the queries),
 * sed_mha_bwd of the SAME mask reads it back -- hand in the buffer the forward call filled.
 * sed_drop_relu_*: y = relu(dropout(x)) of the output projection (:664), keep bytes [n] or null, p_drop = 0.2. */
long sed_mha_mask_words(int B, int T);
int sed_mha_fwd(const float* q, const float* k, const float* v, const unsigned char* keep, float p_drop, int B, int T,
                float* o, float* stats, unsigned* keep_bits, sed_stream_t stream);
int sed_mha_bwd(const float* q, const float* k, const float* v, const float* o, const float* g_o, const unsigned char* keep,
                float p_drop, int B, int T, float* stats, float* g_q, float* g_k, float* g_v, const unsigned* keep_bits,
                sed_stream_t stream);
int sed_drop_relu_fwd(const float* x, const unsigned char* keep, float p_drop, long n, float* y, sed_stream_t stream);
int sed_drop_relu_bwd(const float* g_y, const float* y, const unsigned char* keep, float p_drop, long n, float* g_x,
                      sed_stream_t stream);

/* ---- loss / mixup of targets / optimiser ---------------------------------------------------------------------
 * sed_clip_bce: losses.py:5-12 (F.binary_cross_entropy, mean, log clamped at -100) + d loss / d p.
 * sed_mixup_rows: pytorch_utils.py:80-93 on a [B2][D] matrix (the targets, main.py:246).
 * sed_adam_amsgrad: optim.Adam(betas=(0.9,0.999), eps=1e-8, weight_decay=0, amsgrad=True) (main.py:144-145,:258)
 *   over flat buffers; grad_scale is applied to the gradient first (1/world_size after the RCCL all-reduce).  lr / betas / eps
 *   are doubles like torch's Python scalars: lr / (1 - beta1^t), 1 - beta1, 1 - beta2 are formed in double and rounded once.
 *   skip_flag (nullable, device int[2]) = found-non-finite guard: the gradient is first scanned for NaN / inf (which
 *   raises skip_flag[0] and the nullable host-mapped err_host); when skip_flag[0] != 0 -- from that scan or because a
 *   split-f16 kernel of this step met a non-finite operand (their err_dev word) -- parameters and moments are left
 *   untouched and the refused step is counted in *skipped (nullable device int, one per optimiser; null: in skip_flag[1]).
 *   status_host (nullable, host-mapped int): receives the CUMULATIVE refused-step count as of this step (system-scope
 *   store by the kernel): a host that waits for an event behind step i reads whether step i was refused -- the
 *   deterministic, rank-consistent poll of optim.FusedAdamAmsgrad.  rank_flag: see sed_guard_publish.  g must be 16-byte
 *   aligned when the guard is used. */
int sed_clip_bce(const float* p, const float* y, long n, float* loss, float* grad, sed_stream_t stream);
int sed_mixup_rows(const float* x, const float* lam, long B2, long D, float* out, sed_stream_t stream);
int sed_adam_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, long n, int step, double lr,
                     double beta1, double beta2, double eps, float grad_scale, int* skip_flag, int* skipped, int* err_host,
                     int* status_host, const float* rank_flag, sed_stream_t stream);
/* Rank-consistent found-non-finite guard of a data-parallel job (replaces nothing in the reference: main.py:245-258 has no
 * guard).  sed_guard_publish writes NaN (err_dev[0] != 0) or 0 into flag_out[0]; the caller keeps that word adjacent to
 * the gradient bucket it all-reduces last and hands it to sed_adam_amsgrad as rank_flag (nullable): a non-zero / NaN word
 * after the sum means SOME rank met a non-finite operand, and every rank refuses the same step. */
int sed_guard_publish(const int* err_dev, float* flag_out, sed_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SED_HIP_H */
