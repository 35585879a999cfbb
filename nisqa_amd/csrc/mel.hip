// Mel front end for gfx950 -- replaces get_librosa_melspec after lb.load
// (reference nisqa/NISQA_lib.py:2308-2331; arithmetic of librosa 0.8.1 stft / filters.mel /
// amplitude_to_db restated in oracle/mel.py).
//
// One wave computes one STFT frame end to end (window -> FFT -> |.| -> mel -> dB):
//   * the hann window has `win` <= 1024 non-zero taps inside the 4096-sample frame, so after a
//     (magnitude-preserving) circular shift the real sequence is supported on [0, 1024).  Packed as
//     z[n] = x[2n] + i x[2n+1] (n < 512) the 2048-point complex FFT collapses to FOUR 512-point
//     FFTs of z[n] * W2048^(r n), r = 0..3, giving Z[4m + r]: the three outer radix-4 stages of a
//     4096-point transform are never executed (pruned input);
//   * each 512-point FFT is radix 8x8x8 with 8 complex values per lane and two wave-private LDS
//     transposes (padded strides 72 / 80 B-rows: conflict free for ds_write_b64 / ds_read_b128);
//   * the real-input recombination, magnitude, sparse (two-triangles-per-bin) slaney filterbank as
//     48 wavefront reductions, 10*log10(max(amin^2, S^2)) and the per-clip running maximum
//     (order-preserving atomicMax) all stay in the same wave; the only HBM traffic is the 960
//     input samples (coalesced, L2-shared between overlapping frames) and 48 output floats.
// The top_db clamp needs the per-clip maximum, i.e. a reduction over every frame of the clip; it
// is applied by the consumer (the CNN loads max(x, floor)) or by nisqa_mel_finalize in place.
#include "common.hpp"
#include "../../include/nisqa_hip.h"

struct c32 { float x, y; };
NQ_DEV c32 cmk(float x, float y) { c32 r; r.x = x; r.y = y; return r; }
NQ_DEV c32 cadd(c32 a, c32 b) { return cmk(a.x + b.x, a.y + b.y); }
NQ_DEV c32 csub(c32 a, c32 b) { return cmk(a.x - b.x, a.y - b.y); }
NQ_DEV c32 cmul(c32 a, c32 b) { return cmk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
NQ_DEV c32 cnegi(c32 a) { return cmk(a.y, -a.x); }     // a * (-i)

NQ_DEV void dft4(c32& a0, c32& a1, c32& a2, c32& a3) {
    const c32 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cnegi(csub(a1, a3));
    a0 = cadd(t0, t2); a1 = cadd(t1, t3); a2 = csub(t0, t2); a3 = csub(t1, t3);
}

// v[p] <- sum_a v[a] * exp(-2 pi i a p / 8), natural order in and out
NQ_DEV void dft8(c32 (&v)[8]) {
    c32 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    c32 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    const float r = 0.70710678118654752440f;
    o1 = cmk(r * (o1.x + o1.y), r * (o1.y - o1.x));      // * W8^1 = (r, -r)
    o2 = cnegi(o2);                                       // * W8^2 = -i
    o3 = cmk(r * (o3.y - o3.x), -r * (o3.x + o3.y));     // * W8^3 = (-r, -r)
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

#define MEL_Z_BYTES (4 * 520 * 8)          /* Z[K] at ((K&3)*520 + (K>>2)) complex */
#define MEL_B1_OFF MEL_Z_BYTES             /* exchange 1: [p][72] complex */
#define MEL_B2_OFF (MEL_B1_OFF + 8 * 72 * 8) /* exchange 2: 64 rows of 80 B */
#define MEL_MAG_OFF MEL_Z_BYTES            /* magnitudes alias the exchange buffers */
#define MEL_LDS_BYTES (MEL_B2_OFF + 64 * 80)

__global__ __launch_bounds__(64) void mel_frame_kernel(
    const float* __restrict__ pcm, const int64_t* __restrict__ clip_off,
    const int32_t* __restrict__ frame_off, int n_clips, nisqa_mel_cfg cfg,
    const float* __restrict__ window, const float2* __restrict__ tw,
    const int32_t* __restrict__ band_start, const int32_t* __restrict__ band_len,
    const int32_t* __restrict__ band_woff, const float* __restrict__ band_w,
    float* __restrict__ mel_tm, uint32_t* __restrict__ clip_max_enc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int f = blockIdx.x;
    const int b = find_segment(frame_off, n_clips, f);
    const int t = f - frame_off[b];
    const int64_t c0 = clip_off[b];
    const int L = (int)(clip_off[b + 1] - c0);
    const float* y = pcm + c0;
    // first windowed sample of frame t in clip coordinates (librosa: centre pad n_fft/2, window
    // centre-padded by (n_fft - win)/2)
    const int start = t * cfg.hop - cfg.n_fft / 2 + (cfg.n_fft - cfg.win) / 2;

    // ---- windowed, reflect-padded samples, packed as complex pairs: lane holds n = lane + 64 a
    c32 z[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int n = lane + 64 * a;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int m = 2 * n + e;
            float s = 0.f;
            if (m < cfg.win) {
                int i = start + m;
                i = i < 0 ? -i : i;
                i = i >= L ? 2 * (L - 1) - i : i;
                i = min(max(i, 0), L - 1);
                s = y[i] * window[m];
            }
            v[e] = s;
        }
        z[a] = cmk(v[0], v[1]);
    }

    c32* zbuf = (c32*)smem;
    c32* b1 = (c32*)(smem + MEL_B1_OFF);
    char* b2 = smem + MEL_B2_OFF;

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c32 u[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (r == 0) {
                u[a] = z[a];
            } else {
                const float2 w = tw[(2 * r * (lane + 64 * a)) & 4095];
                u[a] = cmul(z[a], cmk(w.x, w.y));
            }
        }
        dft8(u);                                            // over a -> p
#pragma unroll
        for (int p = 1; p < 8; ++p) {
            const float2 w = tw[(8 * lane * p) & 4095];     // W512^(j p)
            u[p] = cmul(u[p], cmk(w.x, w.y));
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) b1[p * 72 + lane] = u[p];
        __syncthreads();
        const int pq = lane >> 3, j1 = lane & 7;
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) u[j2] = b1[pq * 72 + j1 + 8 * j2];
        dft8(u);                                            // over j2 -> q1
#pragma unroll
        for (int q1 = 1; q1 < 8; ++q1) {
            const float2 w = tw[(64 * j1 * q1) & 4095];     // W64^(j1 q1)
            u[q1] = cmul(u[q1], cmk(w.x, w.y));
        }
#pragma unroll
        for (int q1 = 0; q1 < 8; ++q1) *(c32*)(b2 + (pq + 8 * q1) * 80 + j1 * 8) = u[q1];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t4 = *(const f32x4*)(b2 + lane * 80 + q * 16);
            u[2 * q] = cmk(t4[0], t4[1]);
            u[2 * q + 1] = cmk(t4[2], t4[3]);
        }
        dft8(u);                                            // over j1 -> q2 ; k = lane + 64 q2
#pragma unroll
        for (int q2 = 0; q2 < 8; ++q2) zbuf[r * 520 + lane + 64 * q2] = u[q2];   // K = 4k + r
        __syncthreads();
    }

    // ---- real-input recombination + magnitude for the bins that carry mel weight
    float* mag = (float*)(smem + MEL_MAG_OFF);
    for (int K = lane; K < cfg.n_bins; K += 64) {
        const int Ka = K & 2047, Kb = (2048 - K) & 2047;
        const c32 za = zbuf[(Ka & 3) * 520 + (Ka >> 2)];
        const c32 zb = zbuf[(Kb & 3) * 520 + (Kb >> 2)];
        const float2 w = tw[K];
        const float ar = za.x + zb.x, ai = za.y - zb.y;     // Za + conj(Zb)
        const float dr = za.x - zb.x, di = za.y + zb.y;     // Za - conj(Zb)
        const float wr = w.x * dr - w.y * di, wi = w.x * di + w.y * dr;
        const float xr = 0.5f * (ar + wi), xi = 0.5f * (ai - wr);
        mag[K] = sqrtf(xr * xr + xi * xi);
    }
    __syncthreads();

    // ---- sparse slaney filterbank: one wavefront reduction per band
    float mine = 0.f;
    for (int m = 0; m < cfg.n_mels; ++m) {
        const int st = band_start[m], ln = band_len[m], wo = band_woff[m];
        float part = 0.f;
        for (int q = lane; q < ln; q += 64) part = fmaf(band_w[wo + q], mag[st + q], part);
        part = wave_sum(part);
        if (lane == m) mine = part;
    }
    // ---- amplitude_to_db(ref=1, amin=1e-4): 10*log10(max(amin^2, S^2)); running per-clip max
    float db = -3.0e38f;
    if (lane < cfg.n_mels) {
        db = 10.0f * log10f(fmaxf(cfg.amin_sq, mine * mine));
        mel_tm[(size_t)f * cfg.n_mels + lane] = db;
    }
    db = wave_max(db);
    if (lane == 0) atomicMax(clip_max_enc + b, enc_ordered(db));
}

__global__ void mel_floor_kernel(const uint32_t* __restrict__ clip_max_enc, float top_db, int n_clips,
                                 float* __restrict__ clip_floor) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_clips) clip_floor[b] = dec_ordered(clip_max_enc[b]) - top_db;
}

__global__ void mel_clamp_kernel(float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off, int n_clips,
                                 int total_frames, const float* __restrict__ clip_floor) {
    const int f = blockIdx.x;
    const int b = find_segment(frame_off, n_clips, f);
    const float fl = clip_floor[b];
    if (threadIdx.x < NISQA_N_MELS) {
        float* p = mel_tm + (size_t)f * NISQA_N_MELS + threadIdx.x;
        *p = fmaxf(*p, fl);
    }
}

__global__ void pcm16_kernel(const int16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (float)in[i] * (1.0f / 32768.0f);
}

extern "C" int nisqa_mel_db(const float* pcm, const int64_t* clip_off, const int32_t* frame_off,
                            int32_t n_clips, int32_t total_frames, const nisqa_mel_cfg* cfg,
                            const float* window, const float* twiddle, const int32_t* band_start,
                            const int32_t* band_len, const int32_t* band_woff, const float* band_w,
                            float* mel_tm, uint32_t* clip_max_enc, void* stream) {
    if (!cfg || cfg->n_fft != NISQA_N_FFT || cfg->n_mels != NISQA_N_MELS || cfg->win < 2 || cfg->win > 1024 ||
        cfg->hop < 1 || cfg->n_bins < 1 || cfg->n_bins > 2049 || n_clips <= 0 || total_frames <= 0)
        return NISQA_ERR_ARG;
    hipLaunchKernelGGL(mel_frame_kernel, dim3(total_frames), dim3(64), MEL_LDS_BYTES, (hipStream_t)stream, pcm,
                       clip_off, frame_off, n_clips, *cfg, window, (const float2*)twiddle, band_start, band_len,
                       band_woff, band_w, mel_tm, clip_max_enc);
    return hipGetLastError() == hipSuccess ? NISQA_OK : NISQA_ERR_LAUNCH;
}

extern "C" int nisqa_mel_finalize(float* mel_tm, const int32_t* frame_off, int32_t n_clips, int32_t total_frames,
                                  const uint32_t* clip_max_enc, float top_db, float* clip_floor,
                                  int32_t clamp_in_place, void* stream) {
    if (n_clips <= 0 || total_frames <= 0) return NISQA_ERR_ARG;
    hipLaunchKernelGGL(mel_floor_kernel, dim3((n_clips + 63) / 64), dim3(64), 0, (hipStream_t)stream, clip_max_enc,
                       top_db, n_clips, clip_floor);
    if (clamp_in_place)
        hipLaunchKernelGGL(mel_clamp_kernel, dim3(total_frames), dim3(64), 0, (hipStream_t)stream, mel_tm, frame_off,
                           n_clips, total_frames, clip_floor);
    return hipGetLastError() == hipSuccess ? NISQA_OK : NISQA_ERR_LAUNCH;
}

extern "C" int nisqa_pcm16_to_f32(const int16_t* pcm16, float* pcm, int64_t n, void* stream) {
    if (n <= 0) return NISQA_ERR_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pcm16, pcm, n);
    return hipGetLastError() == hipSuccess ? NISQA_OK : NISQA_ERR_LAUNCH;
}
