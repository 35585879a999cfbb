// libnisqa_hip.so: ABI glue -- version, workspace carving, the whole-batch forward and the MFMA
// fragment-map self-test.  See include/nisqa_hip.h for the contract.
#include "common.hpp"
#include "internal.hpp"
#include "../../include/nisqa_hip.h"

extern "C" int nisqa_abi_version(void) { return NISQA_ABI_VERSION; }

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct ws_plan {
    size_t mel, cmax, cfloor, p3, feat, td, x, pool, total;
};

static ws_plan plan_ws(int32_t n_clips, int32_t total_frames, int32_t np) {
    ws_plan p;
    size_t o = 0;
    p.mel = o;    o += align256((size_t)total_frames * NISQA_N_MELS * 4);
    p.cmax = o;   o += align256((size_t)n_clips * 4);
    p.cfloor = o; o += align256((size_t)n_clips * 4);
    p.p3 = o;     o += align256((size_t)np * 18 * 64 * 4);
    p.feat = o;   o += align256((size_t)np * NISQA_FEAT * 4);
    p.td = o;     o += align256((size_t)np * 64 * 9 * 4);          // (nine bf16 planes x two layer buffers in the three-term mode)
    p.x = o;      o += align256((size_t)np * 64 * 4);
    p.pool = o;   o += align256((size_t)np * 8 * 2 * 4 + (size_t)n_clips * 4);     // scores, values, per-clip arrival counters
    p.total = o;
    return p;
}

extern "C" size_t nisqa_workspace_bytes(int32_t n_clips, int32_t total_frames, int32_t total_tok_padded) {
    if (n_clips <= 0 || total_frames <= 0 || total_tok_padded <= 0) return 0;
    return plan_ws(n_clips, total_frames, total_tok_padded).total;
}

static int predict_batch(const void* pcm, bool pcm16, const int64_t* clip_off, const int32_t* frame_off,
                         const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                         int32_t total_frames, int32_t total_tok_padded, const nisqa_mel_cfg* cfg,
                         const nisqa_model_dev* model, void* ws, size_t ws_bytes, float* out, void* stream) {
    if (!cfg || !model || !ws || n_clips <= 0 || total_frames <= 0 || total_tok_padded <= 0) return NISQA_ERR_ARG;
    const ws_plan p = plan_ws(n_clips, total_frames, total_tok_padded);
    if (ws_bytes < p.total) return NISQA_ERR_WORKSPACE;
    char* w = (char*)ws;
    float* mel = (float*)(w + p.mel);
    uint32_t* cmax = (uint32_t*)(w + p.cmax);
    float* cfloor = (float*)(w + p.cfloor);
    float* p3 = (float*)(w + p.p3);
    float* feat = (float*)(w + p.feat);
    float* td = (float*)(w + p.td);
    float* x = (float*)(w + p.x);
    float* pool = (float*)(w + p.pool);
    void* const* ev = model->stage_events;
#define NQ_STAGE(i) do { if (ev && ev[i] && hipEventRecord((hipEvent_t)ev[i], (hipStream_t)stream) != hipSuccess) return NISQA_ERR_LAUNCH; } while (0)
    if (hipMemsetAsync(cmax, 0, (size_t)n_clips * 4, (hipStream_t)stream) != hipSuccess) return NISQA_ERR_LAUNCH;
    NQ_STAGE(0);
    int rc = pcm16 ? nisqa_mel_db_pcm16((const int16_t*)pcm, clip_off, frame_off, n_clips, total_frames, cfg, model->window,
                                        model->twiddle, model->band_start, model->band_len, model->band_woff,
                                        model->band_w, mel, cmax, stream)
                   : nisqa_mel_db((const float*)pcm, clip_off, frame_off, n_clips, total_frames, cfg, model->window,
                                  model->twiddle, model->band_start, model->band_len, model->band_woff, model->band_w,
                                  mel, cmax, stream);
    if (rc) return rc;
    // the split-bf16 AdaptCNN kernel derives the top_db floor from cmax itself; the other CNN kernels take clip_floor
    if (model->cnn_mode < 0 || model->cnn_mode > 4) return NISQA_ERR_ARG;
    const bool floor_in_cnn = model->arch == 0 && model->cnn_mode >= 1;
    if (!floor_in_cnn) {
        rc = nisqa_mel_finalize(mel, frame_off, n_clips, total_frames, cmax, cfg->top_db, cfloor, 0, stream);
        if (rc) return rc;
    }
    NQ_STAGE(1);
    if (model->arch == 1) {
        // StandardCNN + BiLSTM + last-step pooling; scratch: p3 region holds [NP][12][64], feat region [NP][20],
        // td region the [B][256] final LSTM states
        rc = model->cnn_mode == 1
                 ? nisqa_cnn_standard_bf16(mel, frame_off, tok_off, n_wins, cfloor, n_clips, total_tok_padded,
                                           model->seg_hop, model->cnn_w, model->cnn_wb, feat, stream)
             : model->cnn_mode == 2
                 ? nisqa_cnn_standard_bf16x6(mel, frame_off, tok_off, n_wins, cfloor, n_clips, total_tok_padded,
                                             model->seg_hop, model->cnn_w, model->cnn_wb, feat, stream)
             : model->cnn_mode >= 3
                 ? nisqa_cnn_standard_f16(mel, frame_off, tok_off, n_wins, cfloor, n_clips, total_tok_padded,
                                          model->seg_hop, model->cnn_w, model->cnn_wb, model->cnn_mode, feat, stream)
                 : nisqa_cnn_standard(mel, frame_off, tok_off, n_wins, cfloor, n_clips, total_tok_padded,
                                      model->seg_hop, model->cnn_w, p3, feat, stream);
        if (rc) return rc;
        NQ_STAGE(2);
        NQ_STAGE(3);
        rc = nisqa_lstm_laststep(feat, tok_off, n_wins, n_clips, model->td_w, td, nullptr, out, stream);
        if (rc) return rc;
        NQ_STAGE(4);
        NQ_STAGE(5);
        return NISQA_OK;
    }
    if (model->cnn_mode == 1) {
        rc = nq_cnn_adapt_bf16_from_max(mel, frame_off, tok_off, n_wins, cmax, cfg->top_db, n_clips, total_tok_padded,
                                        model->seg_hop, model->cnn_w, model->cnn_wb, feat, stream);
        if (rc) return rc;
        NQ_STAGE(2);
    } else if (model->cnn_mode == 2) {
        // fp32-grade AdaptCNN on three-term bf16 operands (cnn_wb = the three-term fragments)
        rc = nq_cnn_adapt_bf16x6_from_max(mel, frame_off, tok_off, n_wins, cmax, cfg->top_db, n_clips, total_tok_padded,
                                          model->seg_hop, model->cnn_w, model->cnn_wb, feat, stream);
        if (rc) return rc;
        NQ_STAGE(2);
    } else if (model->cnn_mode == 3 || model->cnn_mode == 4) {
        // AdaptCNN on two-term f16 operands of the scaled tensors, three / four products (cnn_wb = the CNNH blob)
        rc = nq_cnn_adapt_f16_from_max(mel, frame_off, tok_off, n_wins, cmax, cfg->top_db, n_clips, total_tok_padded,
                                       model->seg_hop, model->cnn_w, model->cnn_wb, model->cnn_mode, feat, stream);
        if (rc) return rc;
        NQ_STAGE(2);
    } else {
        rc = nisqa_cnn_front(mel, frame_off, tok_off, n_wins, cfloor, n_clips, total_tok_padded, model->seg_hop,
                             model->cnn_w, p3, stream);
        if (rc) return rc;
        NQ_STAGE(2);
        rc = nisqa_cnn_back(p3, tok_off, n_wins, n_clips, total_tok_padded, model->cnn_w, feat, stream);
        if (rc) return rc;
    }
    NQ_STAGE(3);
    const bool bf = model->cnn_mode == 1 && model->td_wb && model->pool_wb;
    const bool x6 = model->cnn_mode >= 2 && model->td_wb && model->pool_wb;      // three-term fragments in td_wb / pool_wb
    if (x6) {                                            // self-attention and pooling in n_layers + 1 launches
        rc = nisqa_td_pool_bf16x6(feat, tok_off, n_wins, n_clips, total_tok_padded, model->n_layers, model->td_w, model->td_wb,
                                  model->n_heads, model->pool_wb, td, x, pool, out, stream);
        if (rc) return rc;
        NQ_STAGE(4);
        NQ_STAGE(5);
        return NISQA_OK;
    }
    rc = bf ? nisqa_td_selfatt_bf16(feat, tok_off, n_wins, n_clips, total_tok_padded, model->n_layers, model->td_w,
                                    model->td_wb, td, x, stream)
         : x6 ? nisqa_td_selfatt_bf16x6(feat, tok_off, n_wins, n_clips, total_tok_padded, model->n_layers, model->td_w,
                                        model->td_wb, td, x, stream)
            : nisqa_td_selfatt(feat, tok_off, n_wins, n_clips, total_tok_padded, model->n_layers, model->td_w, td, x, stream);
    if (rc) return rc;
    NQ_STAGE(4);
    rc = bf ? nisqa_pool_att_bf16(x, tok_off, n_wins, n_clips, total_tok_padded, model->n_heads, model->pool_w,
                                  model->pool_wb, pool, out, stream)
         : x6 ? nisqa_pool_att_bf16x6(x, tok_off, n_wins, n_clips, total_tok_padded, model->n_heads, model->pool_w,
                                      model->pool_wb, pool, out, stream)
            : nisqa_pool_att(x, tok_off, n_wins, n_clips, total_tok_padded, model->n_heads, model->pool_w, pool, out, stream);
    if (rc) return rc;
    NQ_STAGE(5);
    return NISQA_OK;
}

extern "C" int nisqa_predict_batch(const float* pcm, const int64_t* clip_off, const int32_t* frame_off,
                                   const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                   int32_t total_frames, int32_t total_tok_padded, const nisqa_mel_cfg* cfg,
                                   const nisqa_model_dev* model, void* ws, size_t ws_bytes, float* out, void* stream) {
    return predict_batch(pcm, false, clip_off, frame_off, tok_off, n_wins, n_clips, total_frames, total_tok_padded, cfg,
                         model, ws, ws_bytes, out, stream);
}

extern "C" int nisqa_predict_batch_pcm16(const int16_t* pcm, const int64_t* clip_off, const int32_t* frame_off,
                                         const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                         int32_t total_frames, int32_t total_tok_padded, const nisqa_mel_cfg* cfg,
                                         const nisqa_model_dev* model, void* ws, size_t ws_bytes, float* out,
                                         void* stream) {
    return predict_batch(pcm, true, clip_off, frame_off, tok_off, n_wins, n_clips, total_frames, total_tok_padded, cfg,
                         model, ws, ws_bytes, out, stream);
}

// D[32][32] = A[32][k] * B[k][32] with the fragment maps of common.hpp (k even)
__global__ __launch_bounds__(64) void selftest_mfma_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ d, int k) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc = zero16();
    for (int kk = 0; kk < k; kk += 2) acc = mfma32(a[i * k + kk + h], b[(kk + h) * 32 + i], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[NQ_DROW(r, h) * 32 + i] = acc[r];
}

extern "C" int nisqa_selftest_mfma(const float* a, const float* b, float* d, int32_t k, void* stream) {
    if (k <= 0 || (k & 1)) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d, k);
    return NQ_LAUNCH_STATUS();
}
