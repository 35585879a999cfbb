/* nisqa_train.h -- C ABI of the training-step kernels in libnisqa_hip.so (BASELINE config 5).
 *
 * One optimiser step of the CNN-SA-AP model -- what the reference does at nisqa/NISQA_model.py:131-152 / 330-352
 * (model.train(); model(x, n_wins); biasLoss.get_loss; backward; Adam.step) -- is driven from the host
 * (nisqa_amd/train.py) as a sequence of these operators.  Train-mode BatchNorm needs statistics over every valid
 * segment of the batch between a convolution and its activation, so the per-segment fusion of the inference
 * kernels does not apply; convolutions run as implicit GEMMs on fp32 MFMA, activations live in HBM.
 *
 * Conventions: float32 row-major everywhere; activations are pixel-major, channels contiguous: act[S][H*W][C];
 * token matrices are [tokens][features].  All pointers are device memory owned by the caller, all work is enqueued
 * on `stream`; return 0 or NISQA_ERR_* (nisqa_hip.h).
 */
#ifndef NISQA_TRAIN_H
#define NISQA_TRAIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Grouped GEMM on v_mfma_f32_32x32x2_f32: for every group g, C_g (+)= alpha * op(A_g) op(B_g).
 * desc[g][10] (int64): a_off, b_off, c_off (element offsets into A, B, C), M, N, K, lda, ldb, ldc, tile_start
 * (exclusive prefix sum of ceil(M/64)*ceil(N/64) over the groups; total_tiles = the sum).
 * trans_a: A_g is stored [K][M] (lda = row stride of the stored matrix); trans_b: B_g is stored [N][K] (a Linear /
 * conv weight).  ksplit > 1 splits K over blockIdx.y and accumulates with atomicAdd into a C the caller zeroed
 * (weight gradients: K = number of rows of the batch).  Replaces every F.linear / F.conv2d / torch.bmm and their
 * autograd counterparts of the training step. */
int nisqa_gemm_f32(const float* a, const float* b, float* c, const int64_t* desc, int32_t n_groups,
                   int32_t total_tiles, int32_t trans_a, int32_t trans_b, int32_t ksplit, float alpha, void* stream);
/* the single-group case without a descriptor in device memory: C[m][n] (+)= alpha * op(A) op(B), with an optional
 * epilogue (ksplit == 1 only): + bias[n] (may be NULL), then ReLU if relu != 0 */
int nisqa_gemm_f32_one(const float* a, const float* b, float* c, int64_t m, int64_t n, int64_t k, int64_t lda,
                       int64_t ldb, int64_t ldc, int32_t trans_a, int32_t trans_b, int32_t ksplit, float alpha,
                       const float* bias, int32_t relu, void* stream);

/* conv1 patches straight from the dB spectrogram (Framewise + segment_specs, NISQA_lib.py:2239-2282, 487-502):
 * col[s*720 + m*15 + j][dy*3+dx] = max(mel_tm[frame_off[b] + k*seg_hop + j+dx-1][m+dy-1], clip_floor[b]) or 0
 * outside the 48x15 segment; s = seg_off[b] + k runs over the VALID segments of the batch only. */
int nisqa_im2col_mel(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                     int32_t n_clips, int32_t n_segments, int32_t seg_hop, float* col, void* stream);
/* conv1 (1 -> 16 channels) without a patch matrix: z[s*720 + m*15 + j][co] = bias[co] + sum_tap w[co][tap] * patch, and
 * its weight gradient dw[co][tap] += sum dz * patch (dw zeroed by the caller); w is [16][9] with tap = dy*3 + dx */
int nisqa_conv1_fwd(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                    int32_t n_clips, int32_t n_segments, int32_t seg_hop, const float* w, const float* bias, float* z,
                    void* stream);
int nisqa_conv1_wgrad(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                      int32_t n_clips, int32_t n_segments, int32_t seg_hop, const float* dz, float* dw, void* stream);
/* Layer 1 of the training step without its 720-pixel activations (conv1 -> train-mode BatchNorm -> ReLU -> 24 x 7 adaptive
 * max-pool -> per-channel dropout scale; NISQA_lib.py:688-697 in train mode).  z1 is affine in the nine patch values, so
 * the batch statistics, the reductions of the BatchNorm backward and the dense part of the weight gradient follow from the
 * first and second patch moments; pooled values and their gradients are recomputed from the spectrogram.
 *   nisqa_conv1_moments: mom54 [dev, zeroed by the caller] += P1[t] = sum patch[t] (9), then the upper triangle of
 *                        P2[t][u] = sum patch[t] patch[u] (45), over all pixels of all valid segments;
 *   ..._fwd: sums32 [dev] (sum z, sum z^2 per channel, as nisqa_col_dot(z, z) would give), running statistics and
 *            mean_rstd [dev, 32] updated like nisqa_bn_act_pool_fwd, y [S][168][16] pooled activations (x drop[s][c] if
 *            drop != NULL), arg = pixel (band * 15 + frame) of each maximum;
 *   ..._bwd: dy [S][168][16] -> dgamma, dbeta [16], dw [16][9] (overwritten; the conv bias gradient is exactly zero);
 *            acc176 [dev, zeroed by the caller] is scratch for the float64 reductions. */
int nisqa_conv1_moments(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                        int32_t n_clips, int32_t n_segments, int32_t seg_hop, double* mom54, void* stream);
int nisqa_conv1_bn_act_pool_fwd(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                                int32_t n_clips, int32_t n_segments, int32_t seg_hop, const float* w, const float* bias,
                                const double* mom54, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, double* sums32, float* mean_rstd, const float* drop, float* y,
                                int32_t* arg, void* stream);
int nisqa_conv1_bn_act_pool_bwd(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                                int32_t n_clips, int32_t n_segments, int32_t seg_hop, const float* w, const float* bias,
                                const double* mom54, const float* gamma, const float* beta, const float* mean_rstd,
                                const float* drop, const float* dy, const int32_t* arg, double* acc176, float* dgamma,
                                float* dbeta, float* dw, void* stream);
/* 3x3 patches, padding (1, pad_w): x[S][H*W][C] -> col[S*H*Wo][9*C], Wo = W + 2*pad_w - 2, k = (dy*3+dx)*C + c
 * (C % 4 == 0, 16-byte aligned buffers) */
int nisqa_im2col3x3(const float* x, int32_t n_segments, int32_t h, int32_t w, int32_t c, int32_t pad_w, float* col,
                    void* stream);
/* The same three convolution products without a patch matrix (implicit GEMM: the loaders gather the patches).
 *   mode 0  z[S*H*Wo][co]  = conv(x[S][H*W][ci], w[co][9*ci]) (+ bias, may be NULL)      in: x,  w   out: z
 *   mode 1  dx[S*H*W][ci]  = conv^T(dz[S][H*Wo][co], w)                                  in: dz, w   out: dx
 *   mode 2  dw[co][9*ci]  += dz^T * patches(x)   (dw zeroed by the caller, ksplit chunks) in: x,  dz  out: dw
 * ci, co powers of two >= 4; padding (1, pad_w) as in nisqa_im2col3x3. */
int nisqa_conv3x3_gemm(int32_t mode, const float* x_or_dz, const float* w_or_dz, float* out, int32_t n_segments, int32_t h,
                       int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias, int32_t ksplit, void* stream);
/* The same on split-bf16 MFMA (operands as bf16 hi + lo, three products per term, fp32 accumulation: 16 operand bits, the
 * arithmetic of the inference path's default precision; 5.3x the fp32-MFMA rate).  Same arguments. */
int nisqa_conv3x3_gemm_bf16(int32_t mode, const float* x_or_dz, const float* w_or_dz, float* out, int32_t n_segments, int32_t h,
                            int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias, int32_t ksplit, void* stream);
/* Mode 0 with the BatchNorm batch statistics riding along: stats2c [dev, float64, zeroed by the caller] += sum z, sum z^2
 * per output channel over all rows (what nisqa_col_dot(z, z) would add in a second pass over z); split_bf16 selects the
 * arithmetic of nisqa_conv3x3_gemm_bf16. */
int nisqa_conv3x3_fwd_stats(int32_t split_bf16, const float* x, const float* w_, float* z, int32_t n_segments, int32_t h,
                            int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias, double* stats2c, void* stream);
/* Forward and input-gradient convolutions of the AdaptCNN layers 2..6, SEGMENT-RESIDENT (csrc/train_conv.hip): a workgroup
 * stages the activations of a few whole segments in LDS once (split into bf16 hi + lo once) and runs the K loop of the
 * inference kernel over them; arithmetic = nisqa_conv3x3_gemm_bf16 (three products per term, fp32 accumulation).
 * Replaces, for the five layer shapes of config/train_nisqa_cnn_sa_ap.yaml, what PyTorch does inside
 * nisqa/NISQA_lib.py:690-705 (conv forward) and in autograd's convolution backward (NISQA_model.py:142-143).
 *   nisqa_segconv_supported  1 if (h, w, ci, co, pad_w) is one of those shapes ((24,7,16,32,1) (12,5,32,64,1) (12,5,64,64,1)
 *                            (6,3,64,64,1) (6,3,64,64,0)); otherwise callers use nisqa_conv3x3_gemm_bf16
 *   nisqa_segconv_frag_bytes size of the packed weight fragments of one layer and mode (-1: bad arguments)
 *   nisqa_segconv_pack       w[co][9*ci] -> fragments (bf16 hi / lo, MFMA B-operand order); mode 0 forward, mode 1 input
 *                            gradient (taps mirrored, matrix transposed).  Once per optimiser step.
 *   nisqa_segconv_bf16       mode 0: z[S*h*wo][co] = conv(x[S][h*w][ci]) + bias (may be NULL); stats2c (may be NULL; float64,
 *                            zeroed by the caller) += sum z, sum z^2 per channel;
 *                            mode 1: dx[S*h*w][ci] = conv^T(dz[S][h*wo][co]); bias and stats2c must be NULL.
 *                            frags = nisqa_segconv_pack of the same mode. */
int nisqa_segconv_supported(int32_t h, int32_t w, int32_t ci, int32_t co, int32_t pad_w);
int64_t nisqa_segconv_frag_bytes(int32_t mode, int32_t ci, int32_t co);
int nisqa_segconv_pack(int32_t mode, const float* w, int32_t ci, int32_t co, uint16_t* frags, void* stream);
/* nisqa_segconv_pack for n_jobs <= 10 (layer, mode) pairs in one launch; all arrays are HOST arrays of n_jobs entries
 * (w[j], frags[j] device pointers) */
int nisqa_segconv_pack_many(int32_t n_jobs, const int32_t* modes, const float* const* w, const int32_t* ci, const int32_t* co,
                            uint16_t* const* frags, void* stream);
int nisqa_segconv_bf16(int32_t mode, const float* src, const uint16_t* frags, float* out, int32_t n_segments, int32_t h,
                       int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias, double* stats2c, void* stream);
/* The same two products in EXACT fp32 (v_mfma_f32_32x32x2_f32; fp32 planes in LDS, fp32 fragments): the forward convolutions of
 * the precision modes 'f32' and 'mixed' and the input gradients of 'f32'.  The implicit GEMMs (nisqa_conv3x3_gemm) reach the
 * fp32-MFMA peak on none of these shapes but the 64 -> 64 one at 12 x 5 (narrow outputs, gathers per K-tile); here every
 * activation is fetched once.  nisqa_segconv_frag_bytes_f32 / nisqa_segconv_pack_f32_many / nisqa_segconv_f32 mirror
 * nisqa_segconv_frag_bytes / nisqa_segconv_pack_many / nisqa_segconv_bf16 argument for argument (frags are floats). */
int64_t nisqa_segconv_frag_bytes_f32(int32_t mode, int32_t ci, int32_t co);
int nisqa_segconv_pack_f32_many(int32_t n_jobs, const int32_t* modes, const float* const* w, const int32_t* ci, const int32_t* co,
                                float* const* frags, void* stream);
int nisqa_segconv_f32(int32_t mode, const float* src, const float* frags, float* out, int32_t n_segments, int32_t h, int32_t w,
                      int32_t ci, int32_t co, int32_t pad_w, const float* bias, double* stats2c, void* stream);
/* The same two products at fp32 OPERAND precision on the bf16 matrix pipe (precision mode 'bf16x6'): activations and weights
 * as three exact bf16 terms (hi + mid + lo), six MFMA products per term pair, fp32 accumulation -- the accuracy of
 * nisqa_segconv_f32 at 2.7 x its matrix-pipe rate.  nisqa_segconv_frag_bytes_x6 / nisqa_segconv_pack_x6_many /
 * nisqa_segconv_bf16x6 mirror nisqa_segconv_frag_bytes / nisqa_segconv_pack_many / nisqa_segconv_bf16 argument for argument
 * (three-term fragments: 1.5 x the bytes). */
int64_t nisqa_segconv_frag_bytes_x6(int32_t mode, int32_t ci, int32_t co);
/* The same two products on TWO f16 terms per operand, all four term products (precision mode 'f16x4'): the staged tensor of a
 * workgroup's group of segments as f16 hi + lo of x * 2^e, e from the group's own largest magnitude (measured while the values are
 * in registers); the weights as f16 hi + lo of W * 2^kw, kw from the layer's largest |W| of this optimiser step (computed on the
 * device by the packer, stored behind the fragments: the buffer is 16 bytes longer).  11 + 11 significand bits and the low term's
 * sign: the fp32 value itself for ~75 % of the operands, one fp32 ulp off otherwise; held to the bounds of nisqa_segconv_f32 by the
 * same test.  Mirrors nisqa_segconv_frag_bytes / nisqa_segconv_pack_many / nisqa_segconv_bf16 argument for argument. */
int64_t nisqa_segconv_frag_bytes_f16(int32_t mode, int32_t ci, int32_t co);
int nisqa_segconv_pack_f16_many(int32_t n_jobs, const int32_t* modes, const float* const* w, const int32_t* ci, const int32_t* co,
                                uint16_t* const* frags, void* stream);
int nisqa_segconv_f16(int32_t mode, const float* src, const uint16_t* frags, float* out, int32_t n_segments, int32_t h, int32_t w,
                      int32_t ci, int32_t co, int32_t pad_w, const float* bias, double* stats2c, void* stream);
int nisqa_segconv_pack_x6_many(int32_t n_jobs, const int32_t* modes, const float* const* w, const int32_t* ci, const int32_t* co,
                               uint16_t* const* frags, void* stream);
int nisqa_segconv_bf16x6(int32_t mode, const float* src, const uint16_t* frags, float* out, int32_t n_segments, int32_t h, int32_t w,
                         int32_t ci, int32_t co, int32_t pad_w, const float* bias, double* stats2c, void* stream);
/* Weight gradient of the same layers, segment-resident: dw[co][9*ci] += dz^T * patches(x) (dw zeroed by the caller, like
 * nisqa_conv3x3_gemm mode 2); x[S][h*w][ci], dz[S][h*wo][co].  A workgroup keeps its part of dw in registers over all the
 * segments it walks over and adds it to dw once (fp32 atomics). */
int nisqa_segconv_wgrad_bf16(const float* x, const float* dz, float* dw, int32_t n_segments, int32_t h, int32_t w, int32_t ci,
                             int32_t co, int32_t pad_w, void* stream);
/* adjoint of nisqa_im2col3x3 (gather form, no atomics): dx[S][H*W][C] = sum of the patch entries that read it */
int nisqa_col2im3x3(const float* dcol, int32_t n_segments, int32_t h, int32_t w, int32_t c, int32_t pad_w, float* dx,
                    void* stream);

/* out[0..C) += sum_rows a[r][c], out[C..2C) += sum_rows a[r][c]*b[r][c] in float64 (caller zeroes out).
 * BatchNorm statistics (b = a), BatchNorm / LayerNorm / bias gradients. */
int nisqa_col_dot(const float* a, const float* b, int64_t rows, int32_t c, double* out, void* stream);

/* BatchNorm2d (batch statistics) + ReLU + adaptive_max_pool2d + Dropout2d (NISQA_lib.py:690-705, train mode).
 * sums = nisqa_col_dot(z, z) over rows = S*H*W.  Writes mean_rstd[2C], updates running_mean / running_var
 * (momentum 0.1, unbiased variance), y[S][Ho*Wo][C] and the arg-max pixel of every output (int32, for backward; NOT written
 * when Ho == H and Wo == W -- the identity "pooling" of layers 3, 5, 6, whose backward never reads it).
 * drop (may be NULL): [S][C] multipliers (0 or 1/(1-p)). */
int nisqa_bn_act_pool_fwd(const float* z, const double* sums, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float* mean_rstd, int32_t n_segments, int32_t h,
                          int32_t w, int32_t c, int32_t ho, int32_t wo, const float* drop, float* y, int32_t* arg,
                          void* stream);
/* backward of the above in two passes around a nisqa_col_dot(dyb, z):
 * pass 1: dyb[S][H*W][C] = d loss / d (BatchNorm output) from dy[S][Ho*Wo][C] (pool scatter, dropout, ReLU gate);
 * pass 2: in place dyb -> dz = gamma*rstd*(dyb - mean(dyb) - xhat*mean(dyb*xhat)); dgamma, dbeta from sums2.
 * c must be a multiple of 4 (1024 % c == 0 for nisqa_bn_bwd2).  When 1024 % c == 0 the reductions ride along: pass 1 adds sum(dyb), sum(dyb*z) to sums2_opt[2c] (zeroed by the caller;
 * NULL: run nisqa_col_dot yourself), pass 2 adds the column sums of dz (the conv bias gradient) to sum_dz_opt[2c]. */
int nisqa_bn_act_pool_bwd1(const float* dy, const int32_t* arg, const float* drop, const float* z,
                           const float* mean_rstd, const float* gamma, const float* beta, int32_t n_segments,
                           int32_t h, int32_t w, int32_t c, int32_t ho, int32_t wo, float* dyb, double* sums2_opt,
                           void* stream);
/* The two passes above in a form that reads z and writes dz ONCE: the reductions only see pixels that won a pooling
 * window, so they are taken over the pooled values first (sums2 [2c] float64, zeroed by the caller; 1024 % c == 0). */
int nisqa_bn_act_pool_bwd(const float* dy, const int32_t* arg, const float* drop, const float* z, const float* mean_rstd,
                          const float* gamma, const float* beta, int32_t n_segments, int32_t h, int32_t w, int32_t c, int32_t ho,
                          int32_t wo, double* sums2, float* dz, float* dgamma, float* dbeta, void* stream);
/* The first launch of nisqa_bn_act_pool_bwd on its own (sums2 += sum dyb, sum dyb * z over the pooled values), and the weight
 * gradient of the segment-resident convolutions with the REST of that backward folded into its staging: it is handed z and the
 * pooled gradient dy (+ arg, drop) instead of dz, computes dz = gamma rstd (dyb - mean(dyb) - xhat mean(dyb xhat)) per element
 * while it stages a group of segments, writes dz (for the input-gradient kernel that runs next), dgamma and dbeta, and adds dw
 * like nisqa_segconv_wgrad_bf16.  The dense z -> dz pass of layers 2..6 (memory-bound, 0.29 ms of a 3.4 ms step) disappears.
 * Shapes: the five nisqa_segconv_supported layers with their pooling sizes (ho, wo) = (12,5) (12,5) (6,3) (6,3) (6,1); anything
 * else returns NISQA_ERR_ARG (run nisqa_bn_act_pool_bwd + nisqa_segconv_wgrad_bf16 instead). */
int nisqa_bn_pool_bwd_sums(const float* dy, const int32_t* arg, const float* drop, const float* z, const float* mean_rstd,
                           const float* gamma, const float* beta, int32_t n_segments, int32_t h, int32_t w, int32_t c, int32_t ho,
                           int32_t wo, double* sums2, void* stream);
int nisqa_segconv_wgrad_bn_bf16(const float* x, const float* z, const float* dy, const int32_t* arg, const float* drop,
                                const float* mean_rstd, const float* gamma, const float* beta, const double* sums2, float* dz_out,
                                float* dgamma, float* dbeta, float* dw, int32_t n_segments, int32_t h, int32_t w, int32_t ci,
                                int32_t co, int32_t pad_w, int32_t ho, int32_t wo, void* stream);
/* The weight gradient of the same five layers, segment-resident, in EXACT fp32 (v_mfma_f32_32x32x2_f32: precision mode 'f32',
 * the reference's arithmetic), with or without the BatchNorm backward folded in: z == NULL -> dz_out holds dz on entry (as
 * nisqa_segconv_wgrad_bf16's dz; dy .. sums2, dgamma, dbeta unused); z != NULL -> the contract of nisqa_segconv_wgrad_bn_bf16. */
int nisqa_segconv_wgrad_f32(const float* x, const float* z, const float* dy, const int32_t* arg, const float* drop,
                            const float* mean_rstd, const float* gamma, const float* beta, const double* sums2, float* dz_out,
                            float* dgamma, float* dbeta, float* dw, int32_t n_segments, int32_t h, int32_t w, int32_t ci, int32_t co,
                            int32_t pad_w, int32_t ho, int32_t wo, void* stream);
/* ... and at fp32 OPERAND precision on the bf16 matrix pipe (x and dz as three exact bf16 terms, six products: precision mode
 * 'bf16x6'); the contract of nisqa_segconv_wgrad_f32. */
int nisqa_segconv_wgrad_bf16x6(const float* x, const float* z, const float* dy, const int32_t* arg, const float* drop,
                               const float* mean_rstd, const float* gamma, const float* beta, const double* sums2, float* dz_out,
                               float* dgamma, float* dbeta, float* dw, int32_t n_segments, int32_t h, int32_t w, int32_t ci,
                               int32_t co, int32_t pad_w, int32_t ho, int32_t wo, void* stream);
int nisqa_bn_bwd2(float* dyb_to_dz, const float* z, const double* sums2, const float* mean_rstd, const float* gamma,
                  int64_t rows, int32_t c, float* dgamma, float* dbeta, double* sum_dz_opt, void* stream);

/* LayerNorm over rows of 64 (NISQA_lib.py:991, 1033, 1037): y = gamma*xhat + beta; saves xhat and rstd */
int nisqa_layernorm_fwd(const float* x, const float* gamma, const float* beta, int64_t rows, float* y, float* xhat,
                        float* rstd, void* stream);
/* dx = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma (parameter gradients: nisqa_col_dot(dy, xhat)) */
int nisqa_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, int64_t rows,
                        float* dx, void* stream);

/* Row softmax over ragged rows: row r has len[r] entries at x + off[r]; p = softmax(scale * x) (in place allowed).
 * Attention probabilities (one row per query token) and attention pooling (one row per clip and head). */
int nisqa_softmax_rows_fwd(const float* x, const int64_t* off, const int32_t* len, int64_t rows, float scale,
                           float* p, void* stream);
/* ds = scale * p * (dp - sum_j dp_j p_j)  (in place on dp allowed) */
int nisqa_softmax_rows_bwd(const float* p, const float* dp, const int64_t* off, const int32_t* len, int64_t rows,
                           float scale, float* ds, void* stream);

/* Elementwise helpers on [rows][cols] matrices (cols = row length, bias / vectors of length cols):
 *   op 0: y = x + bias                      op 1: y = relu(x + bias)
 *   op 2: y = x * (aux > 0)   (ReLU gate)   op 3: y = x * aux        (dropout mask)
 *   op 4: y = x + aux          (residual)   op 5: y = x * bias[col]  (per-column scale) */
int nisqa_elementwise(int32_t op, const float* x, const float* aux, const float* bias, int64_t rows, int32_t cols,
                      float* y, void* stream);

/* biasLoss.get_loss (NISQA_lib.py:1880-1892, 1946-1950): loss = sum_h mean_{b: y not NaN} (map_b(y_hat) - y)^2,
 * map_b = cubic with coefficients bias[b][4] (NULL: identity).  Writes loss[1 + heads] (total, then one term per head) and
 * dy_hat[B][heads]. */
int nisqa_mse_loss(const float* y_hat, const float* y, const float* bias, int32_t n_clips, int32_t n_heads,
                   float* loss, float* dy_hat, void* stream);

/* Dropout multipliers (nn.Dropout / Dropout2d in train mode): out[i] = u_i >= p ? 1/(1-p) : 0 with u_i the i-th value of the
 * Philox-4x32-10 stream (seed, offset counts groups of four values); the reference draws its masks from torch's global
 * generator inside the modules, so only the distribution, not the bits, can agree. */
int nisqa_dropout_mask(uint64_t seed, uint64_t offset, float p, int64_t n, float* out, void* stream);

/* dst[table[e][1] + t] = (float)src[table[e][0] + t], t < table[e][2], for e < n_entries (table: int32 [n_entries][3]):
 * the float64 column sums of a step that ARE gradients (biases, LayerNorm parameters) move to the flat gradient buffer */
int nisqa_cast_scatter(const double* src, const int32_t* table, int32_t n_entries, float* dst, void* stream);

/* torch.optim.Adam (betas 0.9 / 0.999, eps 1e-8, no weight decay), step counter t >= 1, on flat buffers */
int nisqa_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr, int32_t t, void* stream);

/* ---- The self-attention block + attention-pooling heads + loss of the training step as ONE call (csrc/train_td.hip) ----
 * Replaces, for the CNN-SA-AP model in train mode, SelfAttention.forward / SelfAttentionLayer.forward (NISQA_lib.py:988-996,
 * 1025-1040), PoolAttFF.forward (NISQA_lib.py:1171-1183), biasLoss.get_loss (NISQA_lib.py:1880-1892, 1946-1950) and autograd's
 * backward of all of them (NISQA_model.py:142-143): a dozen launches (token-tile kernels that chain their products through
 * registers, flash-style attention forward and backward with the dropout masks inside, one grouped split-K GEMM for every
 * weight gradient, one column-sum launch for every bias / LayerNorm gradient) instead of ~116 operator launches.  fp32 MFMA.
 *
 * Parameter offsets `poff` (HOST array, int32, offsets in floats into `params` and `grads`), in this order:
 *   [0] linear.weight as [64][384] with columns in the feature tensor's (y, c) order  [1] linear.bias  [2] norm1.weight  [3] norm1.bias
 *   per layer l at 4 + 12 l: in_proj_weight [192][64], in_proj_bias, out_proj.weight, out_proj.bias, norm1.weight, norm1.bias,
 *                            linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm2.weight, norm2.bias
 *   per head h at 4 + 12 L + 6 h: linear1.weight [128][64], linear1.bias, linear2.weight [128], linear2.bias, linear3.weight [64],
 *                            linear3.bias
 * Token spaces: the caller's tokens (segments) are packed clip after clip (seg_off); inside the block every clip is padded to
 * whole 32-token tiles (ptok_off, multiples of 32; tile_clip[t] = clip of tile t).
 *
 * nisqa_tdtrain_plan (host only, no GPU work): out[0] = workspace floats, out[1] = fragment-buffer floats, out[2] = groups and
 *   out[3] = 64 x 64 tiles of the weight-gradient GEMM, out[4] = column-sum jobs, out[5] / out[6] / out[7] = offsets (floats)
 *   inside the workspace of the INPUT features [n_tokens][384] (the CNN writes them there), of their gradient [n_tokens][384]
 *   and of y_hat [n_clips][n_heads]; the loss (total, then one term per head) follows y_hat at out[7] + round_up(n_clips *
 *   n_heads, 4); from out[8]: the GEMM descriptors [groups][10] then the column-sum jobs [jobs][6] (int64), which the caller
 *   uploads and passes back as wgrad_desc / colsum_jobs.  cap = capacity of out in int64s.
 * nisqa_tdtrain_step: forward, loss, backward.  Adds every parameter gradient of the block into `grads` (zeroed by the
 *   caller), writes y_hat, loss and the feature gradient.  labels [n_clips][n_heads] (NaN = unlabelled), bias_map [n_clips][4]
 *   cubic coefficients or NULL, inv_count [n_heads] = 1 / (labelled clips of the whole batch, all ranks) or 0.  mask_* are the
 *   dropout multipliers of each layer in the CALLER's token order (mask_p: [sum L^2] attention probabilities, row-major per
 *   clip at sq_off; mask_1 / mask_f / mask_2: [n_tokens][64]) or NULL. */
typedef struct nisqa_tdtrain_args {
    int32_t n_clips, n_tokens, n_tokens_padded, n_layers, n_heads, n_wgrad_groups, n_wgrad_tiles, n_colsum_jobs;
    const int32_t* seg_off;      /* device [n_clips + 1] */
    const int32_t* ptok_off;     /* device [n_clips + 1] */
    const int32_t* tile_clip;    /* device [n_tokens_padded / 32] */
    const int64_t* sq_off;       /* device [n_clips + 1]: prefix sum of L^2 */
    const float* params;         /* device, flat parameter buffer */
    float* grads;                /* device, flat gradient buffer (same offsets) */
    const int32_t* poff;         /* HOST */
    float* ws;                   /* device, out[0] floats */
    float* frags;                /* device, out[1] floats */
    const float* labels;         /* device */
    const float* bias_map;       /* device or NULL */
    const float* inv_count;      /* device [n_heads] */
    const float* mask_p[4];
    const float* mask_1[4];
    const float* mask_f[4];
    const float* mask_2[4];
    const int64_t* wgrad_desc;   /* device [n_wgrad_groups][10] */
    const int64_t* colsum_jobs;  /* device [n_colsum_jobs][6] */
} nisqa_tdtrain_args;
int nisqa_tdtrain_plan(int32_t n_clips, int32_t n_tokens, int32_t n_tokens_padded, int32_t n_layers, int32_t n_heads,
                       const int32_t* poff, int64_t* out, int64_t cap);
int nisqa_tdtrain_step(const nisqa_tdtrain_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif
