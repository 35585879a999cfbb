// Measurement scaffolding -- NOT product code.  The default build never sees this file: common.hpp includes it only under
// -DNQ_EXPERIMENTAL (tools/ab_build.sh NAME UNIT "-DNQ_EXPERIMENTAL" builds ONE translation unit with it into ab_libs/NAME.so),
// and defines the same macros empty otherwise.  A library built with it exports nisqa_debug_* readers; nisqa_amd/lib.py refuses to
// load such a library unless NISQA_ALLOW_DEBUG_LIB=1 (tests/test_host.py), so an instrumented build cannot be taken for the product.
//
// Two in-kernel phase clocks (shader clock, s_memtime), both flushed to per-wave SLOTS with plain stores (a first version added
// every wave's numbers to shared counters: 250 k atomics on one cache line per launch made the kernel 4 x slower):
//   stamp clock  NQ_STAMP_BEGIN(); NQ_STAMP(i) at layer boundaries i = 0..11; NQ_STAMP_END(ARR, wave_index)
//                slot = {phase i -> i + 1 for i < 12, [12] = 1, [13] = wall clock, [14] = launch -> stamp 0}; the LAST launch's numbers stay.
//                (cnn_bf16.hip, cnn_bf16x6.hip; read by tools/phase_clock.py)
//   sum clock    NQ_SUM_BEGIN(); NQ_SUM(i) adds the time since the previous mark to phase i (loops: a phase is hit many times);
//                NQ_SUM_COUNT(i, n) adds n to slot i; NQ_SUM_END(ARR, wave_index, cond) adds the wave's sums to its slot.
//                (mel.hip, lstm.hip, train_conv.hip; read by tools/mel_clock.py, tools/lstm_clock.py, tools/bench_segconv.py)
// NQ_CLK_EXPORT(ARR, FN) at file scope declares the slot array and the host reader  int FN(unsigned long long out[16], int reset).
#pragma once
#ifndef NQ_EXPERIMENTAL
#error "experimental.hpp is measurement scaffolding: only -DNQ_EXPERIMENTAL builds may include it"
#endif
#include <stdlib.h>

#define NQ_CLK_SLOTS 32768
static inline int nq_clk_read(const void* sym, unsigned long long* out16, int reset) {
    const size_t bytes = sizeof(unsigned long long) * NQ_CLK_SLOTS * 16;
    if (out16) {
        unsigned long long* h = (unsigned long long*)malloc(bytes);
        if (!h || hipMemcpyFromSymbol(h, sym, bytes) != hipSuccess) { free(h); return -1; }
        for (int q = 0; q < 16; ++q) out16[q] = 0;
        for (int w = 0; w < NQ_CLK_SLOTS; ++w)
            for (int q = 0; q < 16; ++q) out16[q] += h[(size_t)w * 16 + q];
        free(h);
    }
    if (reset) {
        void* d = nullptr;
        if (hipGetSymbolAddress(&d, sym) != hipSuccess || hipMemset(d, 0, bytes) != hipSuccess) return -1;
    }
    return 0;
}
#define NQ_CLK_EXPORT(ARR, FN)                                  \
    __device__ unsigned long long ARR[NQ_CLK_SLOTS * 16];       \
    extern "C" int FN(unsigned long long* out16, int reset) { return nq_clk_read(HIP_SYMBOL(ARR), out16, reset); }

#define NQ_STAMP_BEGIN() const long long nq_clk_top = clock64(), nq_wall_top = wall_clock64(); long long nq_clk[13]
#define NQ_STAMP(i) nq_clk[i] = clock64()
#define NQ_STAMP_END(ARR, WAVE_INDEX)                                                                                    \
    do {                                                                                                                 \
        nq_clk[12] = clock64();                                                                                          \
        if ((threadIdx.x & 63) == 0) {                                                                                   \
            unsigned long long* slot_ = ARR + (size_t)((WAVE_INDEX) & (NQ_CLK_SLOTS - 1)) * 16;                          \
            for (int q_ = 0; q_ < 12; ++q_) slot_[q_] = (unsigned long long)(nq_clk[q_ + 1] - nq_clk[q_]);               \
            slot_[12] = 1ull;                                                                                            \
            slot_[13] = (unsigned long long)(wall_clock64() - nq_wall_top);                                              \
            slot_[14] = (unsigned long long)(nq_clk[0] - nq_clk_top);                                                    \
        }                                                                                                                \
    } while (0)

#define NQ_SUM_BEGIN() long long nq_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, nq_tprev = clock64()
#define NQ_SUM_RESTART() nq_tprev = clock64()
#define NQ_SUM(i) do { const long long t_ = clock64(); nq_sum[i] += t_ - nq_tprev; nq_tprev = t_; } while (0)
#define NQ_SUM_COUNT(i, n) nq_sum[i] += (n)
#define NQ_SUM_END(ARR, WAVE_INDEX, COND)                                                                                \
    do {                                                                                                                 \
        if (COND) {                                                                                                      \
            unsigned long long* slot_ = ARR + (size_t)((WAVE_INDEX) & (NQ_CLK_SLOTS - 1)) * 16;                          \
            for (int q_ = 0; q_ < 16; ++q_) slot_[q_] += (unsigned long long)nq_sum[q_];                                 \
        }                                                                                                                \
    } while (0)
