// nisqa_resample: lb.load(path, sr=ms_sr) for a batch of clips already in HBM -- what librosa 0.8.1 does when the reference passes
// sr=ms_sr (nisqa/NISQA_lib.py:2300, 2304): librosa.resample(y, sr_file, ms_sr, res_type='kaiser_best') = resampy's band-limited
// interpolation (Smith): every output sample is the sum of a left and a right wing of a windowed-sinc half window tabulated at 512
// points per zero crossing, linearly interpolated between table entries; then util.fix_length to ceil(len * ratio).  resampy is
// not in /root/reference (a floating dependency of librosa); the algorithm and the 'kaiser_best' constants are restated from its
// publication / by recollection of version 0.2.2 (oracle/mel.py: resample_kaiser_best is the CPU restatement the tests compare with).
//
// One thread per output sample, a (chunk of 256, clip) grid: <= 2 x 64 / min(1, ratio) taps of one FMA each, weights from the 256 KB table
// (win, delta pairs; L2-resident), samples from HBM as float32 or int16 PCM (/ 32768 like soundfile).  No checkpoint the reference
// ships sets ms_sr: this kernel is on nobody's headline path, it is memory-latency bound and makes no attempt to be more.
#include "common.hpp"
#include "../../include/nisqa_hip.h"

NQ_DEV float pcm_at(const float* p, int64_t i) { return p[i]; }
NQ_DEV float pcm_at(const int16_t* p, int64_t i) { return (float)p[i] * (1.0f / 32768.0f); }

// resampy advances its read position by ONE float64 addition per output sample (time_register += 1 / ratio) and the position decides,
// through n = int(time_register), which input sample the two wings start from.  Where the exact position is an integer (every 147th
// output sample of 48 -> 44.1 kHz) the accumulated rounding decides between n and n - 1 -- and with a non-integer table step
// (downsampling by 470.4 entries per input sample, walked in steps of 470) the two choices differ by 7e-4 of full scale.  To give
// what the reference's loop gives, the position is accumulated the same way: this kernel (one thread per clip) adds sequentially and
// keeps every 256th value; the interpolation kernel adds its thread's remaining <= 255 steps.
__global__ __launch_bounds__(64) void resample_time_kernel(const int64_t* __restrict__ out_valid, int n_clips, int chunks, double inc,
                                                           double* __restrict__ tstart) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_clips) return;
    const int64_t n = out_valid[b];
    double time = 0.0;
    double* row = tstart + (size_t)b * chunks;
    for (int64_t t = 0; t < n; ++t) {
        if ((t & 255) == 0) row[t >> 8] = time;
        time = __dadd_rn(time, inc);                               // (never fused, never re-associated)
    }
}

template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const T* __restrict__ pcm, const int64_t* __restrict__ in_off,
                                                        const int64_t* __restrict__ out_off, const int64_t* __restrict__ out_valid,
                                                        double inc, double scale, int index_step, const float2* __restrict__ table,
                                                        int nwin, int num_table, const double* __restrict__ tstart, int chunks,
                                                        float* __restrict__ out) {
    const int b = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t o0 = out_off[b], n_out = out_off[b + 1] - o0;
    if (t >= n_out) return;
    float acc = 0.f;
    if (t < out_valid[b]) {                                     // (beyond: the zero util.fix_length appends)
        const T* x = pcm + in_off[b];
        const int64_t n_orig = in_off[b + 1] - in_off[b];
        double time = tstart[(size_t)b * chunks + blockIdx.x];
        for (int k = 0; k < (int)threadIdx.x; ++k) time = __dadd_rn(time, inc);
        const int64_t n = (int64_t)time;
        const double fr = scale * (time - (double)n);
        {                                                       // left wing: x[n], x[n - 1], ...
            const double index_frac = fr * (double)num_table;
            const int offset = (int)index_frac;
            const float eta = (float)(index_frac - (double)offset);
            int64_t taps = (nwin - offset) / index_step;
            if (taps > n + 1) taps = n + 1;
            for (int i = 0; i < (int)taps; ++i) {
                const float2 w = table[offset + i * index_step];
                acc = fmaf(fmaf(eta, w.y, w.x), pcm_at(x, n - i), acc);
            }
        }
        {                                                       // right wing: x[n + 1], x[n + 2], ...
            const double index_frac = (scale - fr) * (double)num_table;
            const int offset = (int)index_frac;
            const float eta = (float)(index_frac - (double)offset);
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

extern "C" size_t nisqa_resample_workspace_bytes(int32_t n_clips, int64_t max_out) {
    if (n_clips <= 0 || max_out <= 0) return 0;
    return (size_t)n_clips * (size_t)((max_out + 255) / 256) * sizeof(double);
}

extern "C" int nisqa_resample(const void* pcm, int32_t is_pcm16, const int64_t* in_off, const int64_t* out_off, const int64_t* out_valid,
                              int32_t n_clips, int64_t max_out, double ratio, const float* table, int32_t nwin, int32_t num_table,
                              void* ws, size_t ws_bytes, float* out, void* stream) {
    if (!pcm || !in_off || !out_off || !out_valid || !table || !out || !ws || n_clips <= 0 || max_out <= 0 || !(ratio > 0.0) || nwin < 2 || num_table < 1)
        return NISQA_ERR_ARG;
    if (ws_bytes < nisqa_resample_workspace_bytes(n_clips, max_out)) return NISQA_ERR_WORKSPACE;
    const double sc = ratio < 1.0 ? ratio : 1.0;
    const int index_step = (int)(sc * (double)num_table);
    if (index_step < 1) return NISQA_ERR_ARG;
    const int chunks = (int)((max_out + 255) / 256);
    const double inc = 1.0 / ratio;
    const dim3 grid((unsigned)chunks, (unsigned)n_clips);
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(resample_time_kernel, dim3((n_clips + 63) / 64), dim3(64), 0, (hipStream_t)stream, out_valid, n_clips, chunks, inc, (double*)ws);
    if (is_pcm16)
        hipLaunchKernelGGL(resample_kernel<int16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const int16_t*)pcm, in_off, out_off, out_valid,
                           inc, sc, index_step, (const float2*)table, nwin, num_table, (const double*)ws, chunks, out);
    else
        hipLaunchKernelGGL(resample_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)pcm, in_off, out_off, out_valid,
                           inc, sc, index_step, (const float2*)table, nwin, num_table, (const double*)ws, chunks, out);
    return NQ_LAUNCH_STATUS();
}
