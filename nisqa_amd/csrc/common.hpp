// Shared device helpers for the NISQA gfx950 kernels.
//
// Matrix work uses v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bitwise an fmaf chain), so the
// network keeps the reference's fp32 arithmetic.  Fragment maps (wave64), lane l:
//   A (32 x 2): A[i = l & 31][k = l >> 5]            one VGPR
//   B (2 x 32): B[k = l >> 5][j = l & 31]            one VGPR
//   D (32 x 32): reg r of lane l = D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l & 31]
// Throughout the kernels a K-step covers 8 consecutive k: lane half h = l>>5 loads the float4
// k = 8s+4h .. 8s+4h+3 for its row/col, and MFMA kk of the step pairs k = 8s+kk (h=0) with
// k = 8s+4+kk (h=1).  Weight fragments are pre-packed on the host in exactly that order
// ([step][tile][lane][4] floats) so every wave-load is one contiguous 1 KiB read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NQ_DEV static __device__ __forceinline__

// In-kernel phase clocks are measurement scaffolding (csrc/experimental.hpp), compiled in ONLY by -DNQ_EXPERIMENTAL builds of one
// unit (tools/ab_build.sh); the default build sees these empty macros, and no product source carries an experiment #ifdef.
#ifdef NQ_EXPERIMENTAL
#include "experimental.hpp"
#else
#define NQ_CLK_EXPORT(ARR, FN)
#define NQ_STAMP_BEGIN()
#define NQ_STAMP(i)
#define NQ_STAMP_END(ARR, WAVE_INDEX)
#define NQ_SUM_BEGIN()
#define NQ_SUM_RESTART()
#define NQ_SUM(i)
#define NQ_SUM_COUNT(i, n)
#define NQ_SUM_END(ARR, WAVE_INDEX, COND)
#endif

NQ_DEV f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

NQ_DEV f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// D-fragment row of register r for lane half hf
#define NQ_DROW(r, hf) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hf))

// Largest b in [0, n) with off[b] <= p   (off is an exclusive prefix sum, off[n] > p).
NQ_DEV int find_segment(const int32_t* __restrict__ off, int n, int p) {
    int lo = 0, hi = n;            // invariant: off[lo] <= p < off[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// The same for a wave-uniform p with all 64 lanes active: one vector load and a ballot per 64 entries instead of a
// chain of log2(n) dependent loads (one-wave workgroups start on cold caches: every link is a trip to L2).
NQ_DEV int find_segment_wave(const int32_t* __restrict__ off, int n, int p, int lane) {
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool le = i < n && off[i] <= p;
        cnt += __popcll(__ballot(le));
    }
    return cnt - 1;                // off[0] = 0 <= p
}

// order-preserving float <-> uint32 (for atomicMax on floats of either sign)
NQ_DEV uint32_t enc_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
NQ_DEV float dec_ordered(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

NQ_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
NQ_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// The two waves that share a SIMD start together and take the same time per work item, so without help
// they stay phase-locked: both in their VALU/LDS/epilogue phases at once (matrix pipe idle), both in their
// MFMA phases at once (pipe contended).  Delaying the odd hardware wave slots of the FIRST round of
// workgroups by ~a phase de-synchronises the pair for the rest of the launch; the sleeping wave costs
// nothing because its partner then has the matrix pipe to itself.
NQ_DEV void stagger_odd_wave_slot(int first_round_blocks, int sleeps) {
    if ((int)blockIdx.x < first_round_blocks) {
        const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);   // HW_REG_HW_ID[3:0] = wave_id
        if (wave_slot & 1u)
            for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);    // 127 * 64 clocks each
    }
}

// A launch with more than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize on that kernel, per device: done once
// per device ordinal (a process may drive several GPUs) behind a flag array the call site owns (one array per kernel).  0 = ok.
static inline int nq_lds_opt_in(const void* kernel, int bytes, std::atomic<bool> (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!done[dev].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 2;
        done[dev].store(true, std::memory_order_relaxed);
    }
    return 0;
}
// hipGetLastError() is sticky per thread and PyTorch routinely leaves benign errors behind (e.g.
// hipPointerGetAttributes on pageable host memory).  Clear it before our launches, read it after.
#define NQ_LAUNCH_BEGIN() (void)hipGetLastError()
#define NQ_LAUNCH_STATUS() (hipGetLastError() == hipSuccess ? 0 : 2)
// Workgroup i runs on XCD i % 8 (round-robin dispatch) and each XCD has its own L2.  Consecutive work items share data (the
// token tiles of ONE clip all stream that clip's K / V; neighbouring segment groups read overlapping spectrogram frames), so
// they are given to ONE XCD: item = (i % 8) * (n / 8) + i / 8.  With the identity mapping every XCD pulls every clip's K / V
// (8.4 MB through a 4 MB L2: 87 MB of HBM reads per self-attention launch).  Needs n % 8 == 0; otherwise identity.
NQ_DEV int xcd_tile(int i, int n_tiles) { return (n_tiles & 7) ? i : (i & 7) * (n_tiles >> 3) + (i >> 3); }

