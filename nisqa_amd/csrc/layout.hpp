// Packed-weight blob layouts shared by the kernels (device side) and mirrored in
// nisqa_amd/weights.py (host side).  All offsets are in floats and multiples of 4.
#pragma once

// ---- AdaptCNN blob ("cnn_w") -------------------------------------------------------------
// conv1: folded weights w1[16][9] (c_out, tap = dy*3+dx) and bias t1[16]  (VALU path)
// conv2..6: MFMA B-fragments wf[tap][step][ntile][lane][4] and bias t[c_out]
//   value(tap, s, nt, lane, kk) = W[n = (lane&31) + 32*nt][c = 8*s + 4*(lane>>5) + kk][tap] * bn_scale[n]
#define CNN_W1 0
#define CNN_T1 (CNN_W1 + 16 * 9)
#define CNN_WF2 (CNN_T1 + 16)
#define CNN_T2 (CNN_WF2 + 9 * 2 * 1 * 256)
#define CNN_WF3 (CNN_T2 + 32)
#define CNN_T3 (CNN_WF3 + 9 * 4 * 2 * 256)
#define CNN_WF4 (CNN_T3 + 64)
#define CNN_T4 (CNN_WF4 + 9 * 8 * 2 * 256)
#define CNN_WF5 (CNN_T4 + 64)
#define CNN_T5 (CNN_WF5 + 9 * 8 * 2 * 256)
#define CNN_WF6 (CNN_T5 + 64)
#define CNN_T6 (CNN_WF6 + 9 * 8 * 2 * 256)
#define CNN_W_FLOATS (CNN_T6 + 64)

// ---- StandardCNN blob ("cnn_std_w"): the AdaptCNN offsets above (identical conv shapes; conv6 is a full
// padding-1 3x3 here) followed by fc_out 768 -> 20 in kernel order:
//   CNNS_FC_W[k' = pixel*64 + c][j]  = fc_out.weight[j][c*12 + pixel]   (pixel = y*2 + x of the 6x2 map)
#define CNNS_FC_W CNN_W_FLOATS
#define CNNS_FC_B (CNNS_FC_W + 768 * 20)
#define CNNS_W_FLOATS (CNNS_FC_B + 32)

// ---- BiLSTM blob ("lstm_w"), per direction d: W_ih [512][20], W_hh [512][128], b_ih + b_hh [512] (gate order i,f,g,o)
#define LSTM_WIH 0
#define LSTM_WHH (LSTM_WIH + 512 * 20)
#define LSTM_B (LSTM_WHH + 512 * 128)
#define LSTM_DIR_FLOATS (LSTM_B + 512)
// then last-step pooling: linear.weight [256], bias, pad
#define LSTM_POOL_W (2 * LSTM_DIR_FLOATS)
#define LSTM_W_FLOATS (LSTM_POOL_W + 256 + 4)

// ---- AdaptCNN split-bf16 weight fragments ("cnn_wb", uint16 units; biases stay in cnn_w) -------------
// conv1: [3 terms hi/mid/lo][64 lanes][8]   B[k = tap (0..15, taps >= 9 are 0)][n (0..31, n >= 16 are 0)]
//        (the AdaptCNN kernel uses the first two terms: 16 mantissa bits like every other layer; the StandardCNN
//        kernel all three with the six lowest-order products -- its BiLSTM head amplifies input error 10x more)
// conv5, conv6 (N split over 4 waves, 16x16x32 MFMA): [wave][step g = 2*tap + s][hl][64 lanes][8]:
//   value(w, g, hl, lane, e) = split_hl( W[n = 16*w + (lane&15)][c = 32*s + 8*(lane>>4) + e][tap] * bn_scale[n] )
// conv2..4: [step g = tap*(CIN/16) + s][ntile][hl (hi, lo)][64 lanes][8]:
//   value(g, nt, hl, lane, e) = split_hl( W[n = (lane&31)+32*nt][c = 16*s + 8*(lane>>5) + e][tap] * bn_scale[n] )
#define CNNB_W1 0
#define CNNB_W2 (CNNB_W1 + 3 * 512)
#define CNNB_W3 (CNNB_W2 + 9 * 1 * 2 * 512)
#define CNNB_W4 (CNNB_W3 + 18 * 2 * 2 * 512)
#define CNNB_W5 (CNNB_W4 + 36 * 2 * 2 * 512)
#define CNNB_W6 (CNNB_W5 + 36 * 2 * 2 * 512)
#define CNNB_U16S (CNNB_W6 + 36 * 2 * 2 * 512)
// three-term fragments of cnn_bf16x6.hip (bf16 hi + mid + lo = the fp32 weight exactly): as above with [3] terms per fragment
#define CNNX_W1 0
#define CNNX_W2 (CNNX_W1 + 3 * 512)
#define CNNX_W3 (CNNX_W2 + 9 * 1 * 3 * 512)
#define CNNX_W4 (CNNX_W3 + 18 * 2 * 3 * 512)
#define CNNX_W5 (CNNX_W4 + 36 * 2 * 3 * 512)
#define CNNX_W6 (CNNX_W5 + 36 * 2 * 3 * 512)
#define CNNX_U16S (CNNX_W6 + 36 * 2 * 3 * 512)
/* two-term f16 fragments (cnn_bf16.hip, formats F16X3 / F16X4): the CNNB_ blocks with f16 bit patterns of W * 2^kw (kw per layer:
   max |W| * 2^kw in [2^14, 2^15)), followed by 32 dwords of per-layer constants: int kw[1..6] at dword 0..5, float G[1..6] =
   max over output channels of sum |W| (BatchNorm folded) at dword 8..13, float T[1..6] = max |shift| at dword 16..21 */
#define CNNH_META CNNB_U16S
#define CNNH_U16S (CNNB_U16S + 64)

// ---- split-bf16 fragments of the self-attention / pooling weights ("td_wb", "pool_wb"; uint16 units) --------
// A-fragment [step][mtile][hl][64 lanes][8]; lane (i = l&31, h = l>>5), element e:
//   natural order (first GEMM, B operand comes from memory):   W[32*mt + i][16*s + 8*h + e]
//   chain order (B operand = previous D fragment):             W[32*mt + i][16*s + (e&3) + 8*(e>>2) + 4*h]
#define TDB_PROJ 0                                   /* natural, 24 steps x 2 mtiles */
#define TDB_LAYER0 (TDB_PROJ + 24 * 2 * 2 * 512)
#define TDBL_QKV 0                                   /* chain, 4 steps x 6 mtiles */
#define TDBL_OUT (TDBL_QKV + 4 * 6 * 2 * 512)
#define TDBL_FF1 (TDBL_OUT + 4 * 2 * 2 * 512)
#define TDBL_FF2 (TDBL_FF1 + 4 * 2 * 2 * 512)
#define TDBL_U16S (TDBL_FF2 + 4 * 2 * 2 * 512)
#define PLB_U16S (4 * 4 * 2 * 512)                   /* per head: linear1 [128][64], chain, 4 steps x 4 mtiles */
/* three-term fragments of td_bf16x6.hip: the blocks above with [3] terms per fragment */
#define TDX_PROJ 0
#define TDX_LAYER0 (TDX_PROJ + 24 * 2 * 3 * 512)
#define TDXL_QKV 0
#define TDXL_OUT (TDXL_QKV + 4 * 6 * 3 * 512)
#define TDXL_FF1 (TDXL_OUT + 4 * 2 * 3 * 512)
#define TDXL_FF2 (TDXL_FF1 + 4 * 2 * 3 * 512)
#define TDXL_U16S (TDXL_FF2 + 4 * 2 * 3 * 512)
#define PLX_U16S (4 * 4 * 3 * 512)

// ---- self-attention blob ("td_w") ----------------------------------------------------------
// A-fragments af[step][mtile][lane][4]:
//   value(s, mt, lane, kk) = W[row = (lane&31) + 32*mt][k = 8*s + 4*(lane>>5) + kk]
// proj: W = linear.weight [64][384] (48 steps, 2 mtiles); then vectors of 64.
#define TD_PROJ_AF 0
#define TD_PROJ_B (TD_PROJ_AF + 48 * 2 * 256)
#define TD_LN0_G (TD_PROJ_B + 64)
#define TD_LN0_B (TD_LN0_G + 64)
#define TD_LAYER0 (TD_LN0_B + 64)
// per layer (offsets relative to the layer base)
#define TDL_QKV_AF 0                              /* in_proj_weight [192][64]: 8 steps, 6 mtiles */
#define TDL_QKV_B (TDL_QKV_AF + 8 * 6 * 256)      /* 192 */
#define TDL_OUT_AF (TDL_QKV_B + 192)              /* out_proj.weight [64][64]: 8 steps, 2 mtiles */
#define TDL_OUT_B (TDL_OUT_AF + 8 * 2 * 256)
#define TDL_LN1_G (TDL_OUT_B + 64)
#define TDL_LN1_B (TDL_LN1_G + 64)
#define TDL_FF1_AF (TDL_LN1_B + 64)               /* linear1.weight [64][64] */
#define TDL_FF1_B (TDL_FF1_AF + 8 * 2 * 256)
#define TDL_FF2_AF (TDL_FF1_B + 64)               /* linear2.weight [64][64] */
#define TDL_FF2_B (TDL_FF2_AF + 8 * 2 * 256)
#define TDL_LN2_G (TDL_FF2_B + 64)
#define TDL_LN2_B (TDL_LN2_G + 64)
#define TDL_FLOATS (TDL_LN2_B + 64)

// ---- pooling blob ("pool_w"), per head ------------------------------------------------------
#define PL_W1_AF 0                                /* linear1.weight [128][64]: 8 steps, 4 mtiles */
#define PL_B1 (PL_W1_AF + 8 * 4 * 256)            /* 128 */
#define PL_W2 (PL_B1 + 128)                       /* linear2.weight [128] */
#define PL_W3 (PL_W2 + 128)                       /* linear3.weight [64] */
#define PL_B2 (PL_W3 + 64)                        /* linear2.bias, linear3.bias, pad, pad */
#define PL_FLOATS (PL_B2 + 4)
