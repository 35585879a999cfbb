/*
 * nisqa_hip.h -- C ABI of libnisqa_hip.so: the MI355X (gfx950) implementation of the NISQA
 * predict hot path.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * The reference (gabrielmittag/NISQA) has no FFI of its own: its hot path is Python
 * (SURVEY.md section 8b).  Each entry point below names the reference code it replaces, so a
 * maintainer can bind it with ctypes from nisqa/NISQA_lib.py.
 *
 * Conventions
 *   - every pointer marked [dev] is caller-owned DEVICE memory; the library never allocates or
 *     frees caller memory and never synchronises the device;
 *   - `stream` is the caller's hipStream_t (passed as void*); all work is enqueued on it;
 *   - return value: 0 = NISQA_OK, otherwise a NISQA_ERR_* code; launch errors are reported as
 *     NISQA_ERR_LAUNCH (query hipGetLastError for detail);
 *   - shapes: B clips; clip b has len[b] samples, T[b] = 1 + len[b]/hop frames and
 *     n[b] = ceil((T[b] - (seg_length-1)) / seg_hop) segments ("tokens");
 *       frame_off[B+1] = exclusive prefix sum of T          (int32)
 *       tok_off[B+1]   = exclusive prefix sum of round_up(n, 64)  (int32)  -- tokens are stored
 *                        PADDED per clip so attention tiles never straddle clips: to 64 (a workgroup of the three-term kernels
 *                        nisqa_td_selfatt_bf16x6 / nisqa_td_pool_bf16x6, i.e. of nisqa_predict_batch* in its default mode, holds
 *                        64 tokens of ONE clip); the two-term and exact-fp32 kernels accept any multiple of 32;
 *     TT = frame_off[B], NP = tok_off[B].
 *   - the mel spectrogram is kept FRAME-MAJOR on the device: mel_tm[TT][n_mels]
 *     (the reference's per-clip array is its transpose, [n_mels][T]).
 */
#ifndef NISQA_HIP_H
#define NISQA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NISQA_OK 0
#define NISQA_ERR_ARG 1        /* bad argument / unsupported configuration */
#define NISQA_ERR_LAUNCH 2     /* a kernel launch failed */
#define NISQA_ERR_WORKSPACE 3  /* workspace too small */

#define NISQA_ABI_VERSION 2
#define NISQA_N_MELS 48
#define NISQA_SEG_LEN 15
#define NISQA_N_FFT 4096
#define NISQA_FEAT 384          /* AdaptCNN fan-out: 64 channels x 6 rows (NISQA_lib.py:681-684) */
#define NISQA_DMODEL 64

int nisqa_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Mel front end: replaces get_librosa_melspec (nisqa/NISQA_lib.py:2284-2331) after lb.load:
 * centre/reflect-padded STFT (n_fft 4096, periodic hann of `win` samples), magnitude,
 * slaney mel filterbank, amplitude_to_db(ref=1, amin=1e-4, top_db=80).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_fft;      /* must be 4096 */
    int32_t hop;        /* int(sr * ms_hop_length), NISQA_lib.py:2308 */
    int32_t win;        /* int(sr * ms_win_length) <= n_fft, NISQA_lib.py:2309 (<= 1024: the fast instantiation) */
    int32_t n_mels;     /* must be 48 */
    int32_t n_bins;     /* number of FFT bins with a non-zero mel weight (k = 0 .. n_bins-1), <= 2049 */
    int32_t w_floats;   /* length of band_w (padded filterbank weights), <= 8192 */
    float   amin_sq;    /* amin^2 = 1e-8 */
    float   top_db;     /* 80 */
} nisqa_mel_cfg;

/* Tables (all [dev]), built by the host from the checkpoint's ms_* arguments:
 *   window[win]                 float  periodic hann
 *   twiddle[4096][2]            float  (cos, -sin)(2*pi*k/4096)
 *   band_start/len/woff[n_mels] int32  sparse rows of the mel filterbank: band m has weights
 *                                      band_w[woff[m] .. woff[m]+len[m]) on bins start[m] .. ;
 *                                      len[m] is zero-padded to a multiple of 16 and equal for
 *                                      the four bands 4p .. 4p+3 of a pass (see melbank.py)
 * pcm[dev] float mono samples of all clips back to back, clip b at [clip_off[b], clip_off[b+1]).
 * Outputs: mel_tm[TT][48] UNCLAMPED dB, clip_max_enc[B] (must be zero-filled by the caller;
 * receives an order-preserving uint32 encoding of the per-clip maximum dB).
 */
int nisqa_mel_db(const float* pcm, const int64_t* clip_off, const int32_t* frame_off,
                 int32_t n_clips, int32_t total_frames, const nisqa_mel_cfg* cfg,
                 const float* window, const float* twiddle,
                 const int32_t* band_start, const int32_t* band_len, const int32_t* band_woff,
                 const float* band_w,
                 float* mel_tm, uint32_t* clip_max_enc, void* stream);

/* The same on int16 PCM as it sits in the WAV data chunk: the x / 32768 of soundfile (lb.load, NISQA_lib.py:2304)
 * is folded into the window taps (a power of two: bit-identical to nisqa_pcm16_to_f32 + nisqa_mel_db), so 2 bytes
 * per sample are read and no float copy of the batch exists. */
int nisqa_mel_db_pcm16(const int16_t* pcm, const int64_t* clip_off, const int32_t* frame_off,
                       int32_t n_clips, int32_t total_frames, const nisqa_mel_cfg* cfg,
                       const float* window, const float* twiddle,
                       const int32_t* band_start, const int32_t* band_len, const int32_t* band_woff,
                       const float* band_w,
                       float* mel_tm, uint32_t* clip_max_enc, void* stream);

/* Per-clip dB floor = max - top_db (the librosa top_db clamp, NISQA_lib.py:2330).  Writes
 * clip_floor[B]; when clamp_in_place != 0 also applies max(x, floor) to mel_tm so that it equals
 * the reference spectrogram (the CNN applies the floor on load, so the fused path passes 0). */
int nisqa_mel_finalize(float* mel_tm, const int32_t* frame_off, int32_t n_clips,
                       int32_t total_frames, const uint32_t* clip_max_enc, float top_db,
                       float* clip_floor, int32_t clamp_in_place, void* stream);

/* ------------------------------------------------------------------------------------------
 * Framewise CNN: replaces segment_specs (NISQA_lib.py:2239-2282) + Framewise.forward /
 * AdaptCNN.forward (NISQA_lib.py:487-502, 688-710).  Segments are never materialised: token
 * k of clip b reads frames [frame_off[b] + k*seg_hop, +15) of mel_tm directly.
 * Weights: `cnn_w` is the packed blob produced by nisqa_amd.weights.pack_adapt_cnn (layout in
 * DESIGN.md); BatchNorm (eval) is folded into the conv weights/biases.
 * Output feat[NP][384] in the reference's flatten order (channel*6 + row, NISQA_lib.py:706);
 * rows of padding tokens are left untouched.  p3_ws: scratch [NP][18][64] floats.
 * ------------------------------------------------------------------------------------------ */
int nisqa_cnn_adapt(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                    const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                    int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w,
                    float* p3_ws, float* feat, void* stream);
/* The two launches of nisqa_cnn_adapt separately (conv1-4 + pools -> p3_ws; conv5-6 -> feat). */
int nisqa_cnn_front(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                    const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                    int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w,
                    float* p3_ws, void* stream);
/* The whole AdaptCNN (conv1-6 + pools) on split-bf16 MFMA in ONE launch: each fp32 operand = bf16 hi + bf16 lo,
 * three products per term, fp32 accumulation; |dMOS| <= 2e-5 vs the fp32 kernels.  Same inputs
 * and feat output as nisqa_cnn_adapt; cnn_wb = bf16 fragment blob from nisqa_amd.weights.pack_adapt_cnn_bf16,
 * biases are read from cnn_w; p3_opt (may be NULL) receives an fp32 copy of the pooled conv4 output. */
int nisqa_cnn_adapt_bf16(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                         const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                         int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w,
                         const uint16_t* cnn_wb, float* p3_opt, float* feat, void* stream);
int nisqa_cnn_adapt_segments_bf16(const float* x, int32_t seg_len_padded, const int32_t* tok_off,
                                  const int32_t* n_wins, int32_t n_clips, int32_t total_tok_padded,
                                  const float* cnn_w, const uint16_t* cnn_wb, float* feat, void* stream);
/* The whole AdaptCNN at fp32 OPERAND precision on the bf16 matrix pipe ("bf16x6"): each fp32 operand = three bf16 terms
 * (hi + mid + lo, an exact split of the 24-bit mantissa), six products per term pair (hh, hm, mh, hl, lh, mm; the dropped
 * ones are <= 2 x 2^-24 of the product, typically 0.5 x 2^-24: an fp32 multiply-add's own rounding step), fp32 accumulation.  Same inputs and feat
 * output as nisqa_cnn_adapt (replaces NISQA_lib.py:688-710 like it); cnn_wx = the three-term fragment blob from
 * nisqa_amd.weights.pack_adapt_cnn_bf16(terms=3), biases are read from cnn_w. */
int nisqa_cnn_adapt_bf16x6(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                           const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                           int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w,
                           const uint16_t* cnn_wx, float* feat, void* stream);
int nisqa_cnn_adapt_segments_bf16x6(const float* x, int32_t seg_len_padded, const int32_t* tok_off,
                                    const int32_t* n_wins, int32_t n_clips, int32_t total_tok_padded,
                                    const float* cnn_w, const uint16_t* cnn_wx, float* feat, void* stream);
/* nisqa_cnn_adapt on the f16 matrix pipe: every fp32 operand as TWO f16 terms of the power-of-two-scaled tensor (11 + 11
 * significand bits and the low term's sign: the fp32 value itself for ~75 % of the values, one fp32 ulp off otherwise; the
 * scale of every activation tensor follows the measured maximum of the layer's input, so no finite input leaves f16's range),
 * products = 4: hi*hi + hi*lo + lo*hi + lo*lo ('f16x4'), products = 3: without lo*lo ('f16x3').  Measured against float64
 * both are as close as the exact-fp32 kernels (fewer accumulator roundings; tools/micro/f16probe.hip).
 * cnn_wh: nisqa_amd.weights.pack_adapt_cnn_f16 (CNNH_U16S uint16, csrc/layout.hpp: fragments of W * 2^kw + per-layer constants).
 * Replaces the same reference lines as nisqa_cnn_adapt (NISQA_lib.py:2239-2282, 487-502, 688-710). */
int nisqa_cnn_adapt_f16(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off, const int32_t* n_wins,
                        const float* clip_floor, int32_t n_clips, int32_t total_tok_padded, int32_t seg_hop,
                        const float* cnn_w, const uint16_t* cnn_wh, int32_t products, float* feat, void* stream);
int nisqa_cnn_adapt_segments_f16(const float* x, int32_t seg_len_padded, const int32_t* tok_off, const int32_t* n_wins,
                                 int32_t n_clips, int32_t total_tok_padded, const float* cnn_w, const uint16_t* cnn_wh,
                                 int32_t products, float* feat, void* stream);
/* Segment-tensor input mode: the reference's inner operator model.forward(x, n_wins)
 * (NISQA_lib.py:137-142, 260-268) hands over x[B][L][1][48][15] (zero-padded to L segments per clip).
 * Same outputs as nisqa_cnn_adapt; no dB floor is applied (x is already clamped). */
int nisqa_cnn_adapt_segments(const float* x, int32_t seg_len_padded, const int32_t* tok_off,
                             const int32_t* n_wins, int32_t n_clips, int32_t total_tok_padded,
                             const float* cnn_w, float* p3_ws, float* feat, void* stream);
int nisqa_cnn_back(const float* p3_ws, const int32_t* tok_off, const int32_t* n_wins,
                   int32_t n_clips, int32_t total_tok_padded, const float* cnn_w,
                   float* feat, void* stream);

/* ------------------------------------------------------------------------------------------
 * Time dependency: replaces SelfAttention.forward + 2 x SelfAttentionLayer.forward
 * (NISQA_lib.py:988-996, 1025-1040): Linear 384->64, LayerNorm, then per layer masked
 * single-head attention (d=64, scale 1/8), out-proj, residual+LN, FFN(ReLU), residual+LN.
 * td_w: packed blob from nisqa_amd.weights.pack_self_att.  ws: scratch, 6*NP*64 floats
 * (q, k, v-transposed; double-buffered across layers).  x_out[NP][64].
 * ------------------------------------------------------------------------------------------ */
int nisqa_td_selfatt(const float* feat, const int32_t* tok_off, const int32_t* n_wins,
                     int32_t n_clips, int32_t total_tok_padded, int32_t n_layers,
                     const float* td_w, float* ws, float* x_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pooling heads: replaces n_heads x PoolAttFF.forward (NISQA_lib.py:1171-1183) and the
 * torch.cat of NISQA_DIM.forward (NISQA_lib.py:265-266).  n_heads = 5 (NISQA_DIM: mos, noi,
 * dis, col, loud) or 1 (NISQA).  pool_w: packed blob from nisqa_amd.weights.pack_pool_att.
 * ws: scratch 2*NP*8 floats.  out[B][n_heads].
 * ------------------------------------------------------------------------------------------ */
int nisqa_pool_att(const float* x, const int32_t* tok_off, const int32_t* n_wins,
                   int32_t n_clips, int32_t total_tok_padded, int32_t n_heads,
                   const float* pool_w, float* ws, float* out, void* stream);

/* Split-bf16 variants of the two calls above (same results to ~1e-5): td_wb / pool_wb are the bf16
 * hi/lo weight fragments from nisqa_amd.weights.pack_self_att_bf16 / pack_pool_att_bf16; biases and LayerNorm
 * parameters are still read from td_w / pool_w.  ws sizes as for the fp32 calls. */
int nisqa_td_selfatt_bf16(const float* feat, const int32_t* tok_off, const int32_t* n_wins,
                          int32_t n_clips, int32_t total_tok_padded, int32_t n_layers,
                          const float* td_w, const uint16_t* td_wb, float* ws, float* x_out, void* stream);
int nisqa_pool_att_bf16(const float* x, const int32_t* tok_off, const int32_t* n_wins,
                        int32_t n_clips, int32_t total_tok_padded, int32_t n_heads,
                        const float* pool_w, const uint16_t* pool_wb, float* ws, float* out, void* stream);
int nisqa_pool_score_bf16(const float* x, const int32_t* tok_off, const int32_t* n_wins,
                          int32_t n_clips, int32_t total_tok_padded, int32_t n_heads,
                          const float* pool_w, const uint16_t* pool_wb, float* ws, void* stream);
/* The same at fp32 OPERAND precision on the bf16 matrix pipe (three exact bf16 terms per operand, six products; DESIGN.md 4.5
 * "bf16x6"): td_wx / pool_wx are the three-term fragments from nisqa_amd.weights.pack_self_att_bf16 / pack_pool_att_bf16 with
 * terms=3; nisqa_td_selfatt_bf16x6 needs ws of 9 * total_tok_padded * 64 floats (the others: 6 *). */
int nisqa_td_selfatt_bf16x6(const float* feat, const int32_t* tok_off, const int32_t* n_wins,
                            int32_t n_clips, int32_t total_tok_padded, int32_t n_layers, const float* td_w,
                            const uint16_t* td_wx, float* ws, float* x_out, void* stream);
int nisqa_pool_att_bf16x6(const float* x, const int32_t* tok_off, const int32_t* n_wins,
                          int32_t n_clips, int32_t total_tok_padded, int32_t n_heads, const float* pool_w,
                          const uint16_t* pool_wx, float* ws, float* out, void* stream);
int nisqa_pool_score_bf16x6(const float* x, const int32_t* tok_off, const int32_t* n_wins,
                            int32_t n_clips, int32_t total_tok_padded, int32_t n_heads, const float* pool_w,
                            const uint16_t* pool_wx, float* ws, void* stream);
/* Self-attention AND attention pooling (SelfAttention.forward + 5 x PoolAttFF.forward, NISQA_lib.py:988-996, 1025-1040, 1171-1183)
 * in n_layers + 1 launches: the last encoder layer's launch scores its own tokens for every pooling head and the clip's last
 * workgroup to arrive does the softmax over the clip's tokens.  pool_wx: pack_pool_att_bf16(terms=3) followed by
 * pack_pool_att_t16 (nisqa_amd/weights.py); ws as for nisqa_td_selfatt_bf16x6; ws_pool: 16 * total_tok_padded floats +
 * n_clips int32; total_tok_padded a multiple of 64 with every tok_off[b] a multiple of 64. */
int nisqa_td_pool_bf16x6(const float* feat, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                         int32_t total_tok_padded, int32_t n_layers, const float* td_w, const uint16_t* td_wx,
                         int32_t n_heads, const uint16_t* pool_wx, float* ws, float* x_out, float* ws_pool, float* out,
                         void* stream);
/* second pass of the pooling (masked softmax over tokens + weighted sum) on scores left in ws */
int nisqa_pool_final(const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                     int32_t total_tok_padded, int32_t n_heads, const float* ws, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * nisqa_tts.tar architecture.
 * nisqa_cnn_standard replaces segment_specs + Framewise.forward + StandardCNN.forward incl. fc_out
 * (NISQA_lib.py:2239-2282, 487-502, 811-836): feat20[NP][20]; p3_ws scratch [NP][12][64] floats; cnn_std_w from
 * nisqa_amd.weights.pack_standard_cnn.  nisqa_cnn_standard_bf16 is the same operator on split-bf16 MFMA (one
 * kernel, no scratch; cnn_wb from nisqa_amd.weights.pack_adapt_cnn_bf16 -- the conv shapes are the AdaptCNN's).
 * nisqa_lstm_laststep replaces LSTM.forward (bidirectional, hidden 128; NISQA_lib.py:925-943) and
 * PoolLastStepBi.forward (NISQA_lib.py:1107-1115): out[B][1]; hfin_ws scratch [B][256] floats; seq_opt (may be
 * NULL) receives the full [NP][256] LSTM output; lstm_w from nisqa_amd.weights.pack_lstm_laststep.
 * ------------------------------------------------------------------------------------------ */
int nisqa_cnn_standard(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                       const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                       int32_t total_tok_padded, int32_t seg_hop, const float* cnn_std_w,
                       float* p3_ws, float* feat20, void* stream);
int nisqa_cnn_standard_bf16(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                            const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                            int32_t total_tok_padded, int32_t seg_hop, const float* cnn_std_w,
                            const uint16_t* cnn_wb, float* feat20, void* stream);
/* ... and at fp32 OPERAND precision on the bf16 matrix pipe (three exact bf16 terms per operand, six products: "bf16x6"); cnn_wx =
 * the three-term fragment blob (nisqa_amd.weights.pack_adapt_cnn_bf16(terms=3): the conv shapes are the AdaptCNN's). */
int nisqa_cnn_standard_bf16x6(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                              const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                              int32_t total_tok_padded, int32_t seg_hop, const float* cnn_std_w,
                              const uint16_t* cnn_wx, float* feat20, void* stream);
/* nisqa_cnn_standard on two-term f16 operands of the power-of-two-scaled tensors (see nisqa_cnn_adapt_f16; products = 3 or 4;
 * cnn_wh = nisqa_amd.weights.pack_adapt_cnn_f16 of the StandardCNN's convolutions, fc_out stays fp32 VALU). */
int nisqa_cnn_standard_f16(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                           const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                           int32_t total_tok_padded, int32_t seg_hop, const float* cnn_std_w,
                           const uint16_t* cnn_wh, int32_t products, float* feat20, void* stream);
int nisqa_lstm_laststep(const float* feat20, const int32_t* tok_off, const int32_t* n_wins,
                        int32_t n_clips, const float* lstm_w, float* hfin_ws, float* seq_opt,
                        float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole forward for one batch: replaces the body of the per-batch step of predict_dim /
 * predict_mos (NISQA_lib.py:1420-1467): PCM in, [B][n_heads] out.  Internally the calls above,
 * carving scratch out of `ws` (size from nisqa_workspace_bytes).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const float* window; const float* twiddle;
    const int32_t* band_start; const int32_t* band_len; const int32_t* band_woff; const float* band_w;
    const float* cnn_w; const float* td_w; const float* pool_w;
    int32_t n_layers; int32_t n_heads; int32_t seg_hop;
    /* optional profiling hook: NULL, or 6 caller-created hipEvent_t recorded on `stream` at the stage
     * boundaries of nisqa_predict_batch: [0] start, [1] after mel, [2] after the conv1-4 kernel,
     * [3] after the conv5-6 kernel, [4] after self-attention, [5] after pooling; a NULL entry is skipped (an event record costs
     * ~5 us of stream time: six of them 29 us of a 1.03 ms batch, profiles/r06_stage_event_cost.txt) */
    void* const* stage_events;
    const uint16_t* cnn_wb;  /* split-bf16 conv fragments, or NULL */
    int32_t cnn_mode;        /* 0 = exact fp32 MFMA kernels, 1 = split-bf16 kernels (needs cnn_wb, td_wb, pool_wb),
                              * 2 = every GEMM on three-term bf16 (arch 0: nisqa_cnn_adapt_bf16x6, nisqa_td_selfatt_bf16x6,
                              * nisqa_pool_att_bf16x6; cnn_wb / td_wb / pool_wb = their three-term fragments; td_wb or
                              * pool_wb NULL: self-attention and pooling on the exact fp32 kernels; arch 1: nisqa_cnn_standard_bf16x6, the
                              * BiLSTM is fp32 in every mode),
                              * 3 / 4 = the CNN on two-term f16 operands with 3 / 4 products (arch 0: nisqa_cnn_adapt_f16, arch 1:
                              * nisqa_cnn_standard_f16; cnn_wb = the CNNH blob); self-attention and pooling as in mode 2 (td_wb / pool_wb =
                              * three-term fragments) */
    const uint16_t* td_wb;   /* split-bf16 self-attention fragments, or NULL */
    const uint16_t* pool_wb; /* split-bf16 pooling fragments, or NULL */
    int32_t arch;            /* 0 = CNN-SA-AP (nisqa.tar, nisqa_mos_only.tar); 1 = StandardCNN + BiLSTM + last-step
                              * pooling (nisqa_tts.tar): cnn_w = cnn_std_w blob, td_w = lstm_w blob, pool_w unused */
    /* Batches may be in flight on SEVERAL streams at once (the predict loop keeps two): every kernel pair of this library
     * is bit-exact under overlap.  (Round 1 kept the mel + CNN sections of different streams apart with two event
     * fields here; the cause -- a gfx950 packed-f32 op_sel form that misreads next to another kernel's bf16 MFMA waves,
     * tools/micro/corun6.hip -- is avoided in the kernels now and tests/test_host.py lints the ISA for it.) */
} nisqa_model_dev;

size_t nisqa_workspace_bytes(int32_t n_clips, int32_t total_frames, int32_t total_tok_padded);

int nisqa_predict_batch(const float* pcm, const int64_t* clip_off, const int32_t* frame_off,
                        const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                        int32_t total_frames, int32_t total_tok_padded,
                        const nisqa_mel_cfg* cfg, const nisqa_model_dev* model,
                        void* ws, size_t ws_bytes, float* out, void* stream);

/* The same with the batch as int16 PCM (the WAV data chunks back to back): what predict_dir / predict_csv hand over,
 * 2 bytes per sample across PCIe and in HBM.  Results are bit-identical to nisqa_pcm16_to_f32 + nisqa_predict_batch. */
int nisqa_predict_batch_pcm16(const int16_t* pcm, const int64_t* clip_off, const int32_t* frame_off,
                              const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                              int32_t total_frames, int32_t total_tok_padded,
                              const nisqa_mel_cfg* cfg, const nisqa_model_dev* model,
                              void* ws, size_t ws_bytes, float* out, void* stream);

/* int16 PCM -> float32 (x / 32768), the soundfile scaling lb.load applies (NISQA_lib.py:2304). */
int nisqa_pcm16_to_f32(const int16_t* pcm16, float* pcm, int64_t n, void* stream);

/* lb.load(path, sr=ms_sr) for clips already on the device (NISQA_lib.py:2300, 2304 -> librosa.resample(y, sr_file, ms_sr,
 * res_type='kaiser_best') -> resampy): clip b = pcm[in_off[b] .. in_off[b + 1]) (float32 samples, or int16 PCM scaled by 1 / 32768
 * when is_pcm16) at the file's rate -> out[out_off[b] .. out_off[b + 1]) at ratio = ms_sr / sr_file.  out_off[b + 1] - out_off[b] =
 * ceil(len_b * ratio) (librosa's fix_length), of which the first out_valid[b] = (int64)(len_b * ratio) samples are interpolated and the
 * rest is zero; max_out = the longest output clip.  table [nwin][2]: the 'kaiser_best' half window (64 zero crossings, num_table = 512
 * entries per crossing, nwin = 32 769; multiplied by ratio when ratio < 1) and its forward differences
 * (nisqa_amd.melbank.kaiser_best_table).  ws: nisqa_resample_workspace_bytes(n_clips, max_out) bytes of device scratch (resampy's
 * read position is ONE float64 addition per output sample; it is accumulated the same way, checkpointed every 256 samples).
 * All device pointers; offsets in samples.  No shipped checkpoint sets ms_sr. */
size_t nisqa_resample_workspace_bytes(int32_t n_clips, int64_t max_out);
int nisqa_resample(const void* pcm, int32_t is_pcm16, const int64_t* in_off, const int64_t* out_off, const int64_t* out_valid,
                   int32_t n_clips, int64_t max_out, double ratio, const float* table, int32_t nwin, int32_t num_table,
                   void* ws, size_t ws_bytes, float* out, void* stream);

/* Self-test of the MFMA fragment maps the kernels rely on: D = A(32xK) * B(Kx32) with
 * v_mfma_f32_32x32x2_f32, a/b/d [dev] row-major.  Used by tests only. */
int nisqa_selftest_mfma(const float* a, const float* b, float* d, int32_t k, void* stream);

/* Measurement probe (not on the predict path): dense v_mfma_f32_32x32x16_bf16 on register operands, two waves per SIMD.
 * operands [dev] 65536 x 16 bytes of bf16 bit patterns; out [dev] blocks * 256 floats; clk [dev] blocks * 4 pairs
 * (shader-clock ticks, 100 MHz ticks) per wave; every wave issues 16 * iters MFMAs (32768 flop each).  bench.py uses it to
 * report the bf16 rate and shader clock THIS GPU sustains on random operands next to the data-sheet peak. */
int nisqa_probe_mfma_sustained(const void* operands, float* out, uint64_t* clk, int32_t blocks, int32_t iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NISQA_HIP_H */
