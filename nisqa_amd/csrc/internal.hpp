// Entry points shared between the translation units of libnisqa_hip.so that are NOT part of the C ABI
// (include/nisqa_hip.h): fused variants the whole-forward path (api.hip) uses.
#pragma once
#include <stdint.h>

// nisqa_cnn_adapt_bf16 with the top_db floor of each clip taken inside the kernel from clip_max_enc (the encoded
// per-clip maximum nisqa_mel_db publishes) instead of a clip_floor array: saves the nisqa_mel_finalize launch.
int nq_cnn_adapt_bf16_from_max(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                               const int32_t* n_wins, const uint32_t* clip_max_enc, float top_db, int32_t n_clips,
                               int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w, const uint16_t* cnn_wb,
                               float* feat, void* stream);

// the same for the two-term f16 kernels (cnn_bf16.hip; products = 3 or 4; cnn_wh = nisqa_amd.weights.pack_adapt_cnn_f16)
int nq_cnn_adapt_f16_from_max(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off, const int32_t* n_wins,
                              const uint32_t* clip_max_enc, float top_db, int32_t n_clips, int32_t total_tok_padded, int32_t seg_hop,
                              const float* cnn_w, const uint16_t* cnn_wh, int32_t products, float* feat, void* stream);

// the same for the three-term kernel (cnn_bf16x6.hip; cnn_wx = three-term fragments, nisqa_amd.weights.pack_adapt_cnn_bf16(terms=3))
int nq_cnn_adapt_bf16x6_from_max(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                                 const int32_t* n_wins, const uint32_t* clip_max_enc, float top_db, int32_t n_clips,
                                 int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w, const uint16_t* cnn_wx,
                                 float* feat, void* stream);
