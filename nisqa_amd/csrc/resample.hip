// nisqa_resample: lb.load(path, sr=ms_sr) for a batch of clips already in HBM -- what librosa 0.8.1 does when the reference passes
// sr=ms_sr (nisqa/NISQA_lib.py:2300, 2304): librosa.resample(y, sr_file, ms_sr, res_type='kaiser_best') = resampy's band-limited
// interpolation (Smith): every output sample is the sum of a left and a right wing of a windowed-sinc half window tabulated at 512
// points per zero crossing, linearly interpolated between table entries; then util.fix_length to ceil(len * ratio).  resampy is
// not in /root/reference (a floating dependency of librosa); the algorithm and the 'kaiser_best' constants are restated from its
// publication / by recollection of version 0.2.2 (oracle/mel.py: resample_kaiser_best is the CPU restatement the tests compare with).
//
// One thread per output sample, a (chunk, clip) grid: <= 2 x 64 / min(1, ratio) taps of one FMA each, weights from the 256 KB table
// (win, delta pairs; L2-resident), samples from HBM as float32 or int16 PCM (/ 32768 like soundfile).  No checkpoint the reference
// ships sets ms_sr: this kernel is on nobody's headline path, it is memory-latency bound and makes no attempt to be more.
#include "common.hpp"
#include "../../include/nisqa_hip.h"

NQ_DEV float pcm_at(const float* p, int64_t i) { return p[i]; }
NQ_DEV float pcm_at(const int16_t* p, int64_t i) { return (float)p[i] * (1.0f / 32768.0f); }

template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const T* __restrict__ pcm, const int64_t* __restrict__ in_off,
                                                        const int64_t* __restrict__ out_off, const int64_t* __restrict__ out_valid,
                                                        double inv_ratio, float scale, int index_step, const float2* __restrict__ table,
                                                        int nwin, int num_table, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t o0 = out_off[b], n_out = out_off[b + 1] - o0;
    if (t >= n_out) return;
    float acc = 0.f;
    if (t < out_valid[b]) {                                     // (beyond: the zero util.fix_length appends)
        const T* x = pcm + in_off[b];
        const int64_t n_orig = in_off[b + 1] - in_off[b];
        const double time = (double)t * inv_ratio;
        const int64_t n = (int64_t)time;
        const float fr = (float)(time - (double)n);
        {                                                       // left wing: x[n], x[n - 1], ...
            const float index_frac = scale * fr * (float)num_table;
            const int offset = (int)index_frac;
            const float eta = index_frac - (float)offset;
            int64_t taps = (nwin - offset) / index_step;
            if (taps > n + 1) taps = n + 1;
            for (int i = 0; i < (int)taps; ++i) {
                const float2 w = table[offset + i * index_step];
                acc = fmaf(fmaf(eta, w.y, w.x), pcm_at(x, n - i), acc);
            }
        }
        {                                                       // right wing: x[n + 1], x[n + 2], ...
            const float index_frac = (scale - scale * fr) * (float)num_table;
            const int offset = (int)index_frac;
            const float eta = index_frac - (float)offset;
            int64_t taps = (nwin - offset) / index_step;
            if (taps > n_orig - n - 1) taps = n_orig - n - 1;
            for (int k = 0; k < (int)taps; ++k) {
                const float2 w = table[offset + k * index_step];
                acc = fmaf(fmaf(eta, w.y, w.x), pcm_at(x, n + k + 1), acc);
            }
        }
    }
    out[o0 + t] = acc;
}

extern "C" int nisqa_resample(const void* pcm, int32_t is_pcm16, const int64_t* in_off, const int64_t* out_off, const int64_t* out_valid,
                              int32_t n_clips, int64_t max_out, double ratio, const float* table, int32_t nwin, int32_t num_table,
                              float* out, void* stream) {
    if (!pcm || !in_off || !out_off || !out_valid || !table || !out || n_clips <= 0 || max_out <= 0 || !(ratio > 0.0) || nwin < 2 || num_table < 1)
        return NISQA_ERR_ARG;
    const double sc = ratio < 1.0 ? ratio : 1.0;
    const int index_step = (int)(sc * (double)num_table);
    if (index_step < 1) return NISQA_ERR_ARG;
    const dim3 grid((unsigned)((max_out + 255) / 256), (unsigned)n_clips);
    NQ_LAUNCH_BEGIN();
    if (is_pcm16)
        hipLaunchKernelGGL(resample_kernel<int16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const int16_t*)pcm, in_off, out_off, out_valid,
                           1.0 / ratio, (float)sc, index_step, (const float2*)table, nwin, num_table, out);
    else
        hipLaunchKernelGGL(resample_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)pcm, in_off, out_off, out_valid,
                           1.0 / ratio, (float)sc, index_step, (const float2*)table, nwin, num_table, out);
    return NQ_LAUNCH_STATUS();
}
