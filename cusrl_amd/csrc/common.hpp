// Shared device/host helpers for libcusrl_hip.so (gfx950 / CDNA4 only — wave64, no CUDA paths).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cusrl_hip.h"

namespace cusrl {

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kBlock = 256;    // 4 waves = one wave per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;

// Launch-shape / cache-policy overrides (cusrl_set_option, include/cusrl_hip.h): 0 everywhere = every kernel chooses by its
// own measured rule.  Set by the host through the C ABI — never read from the process environment inside a launch entry point.
enum Option {
    kOptGaePolicy,    // gae_policy    1 + {0, 5, 7}: force that cache policy of the scan (0: by footprint)
    kOptGaeBlock,     // gae_block     128 | 256 threads per block (0: by column count)
    kOptLossPolicy,   // loss_policy   1: default cache policy, 2: non-temporal [B, A] streams (0: by footprint)
    kOptPushPolicy,   // push_policy   1: default cache policy, 2: streaming (0: by footprint)
    kOptColsumRows,   // colsum_rows   rows per block of the mask + column-sum pass (0: 64)
    kOptHeadRows,     // head_rows     rows per block of the narrow-head backward (0: 96)
    kOptGruBiasRows,  // gru_bias_rows rows per partial row of the GRU bias gradients, 4 | 8 | 16 | 32 (0: 16)
    kNumOptions
};
int64_t option(Option which);  // api.hip

inline int launch_status() {
    hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : static_cast<int>(err);
}

inline hipStream_t as_stream(void *stream) { return static_cast<hipStream_t>(stream); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline bool aligned(const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

__device__ __forceinline__ bool aligned_ptr16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- wave / block reductions (wave64 shuffles; LDS only across the 4 waves of a block) ----
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;  // valid in lane 0
}

// Sum over the block; result valid in thread 0.  `scratch` holds kWavesPerBlock entries.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T total = T(0);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) total += scratch[w];
    }
    __syncthreads();
    return total;
}

// Exclusive prefix sum of one int per thread over the block; also returns the block total.
__device__ __forceinline__ int block_exclusive_scan(int v, int *scratch, int &total) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    int incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        int up = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += up;
    }
    if (lane == kWave - 1) scratch[wave] = incl;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
        int s = scratch[w];
        if (w < wave) base += s;
        total += s;
    }
    __syncthreads();
    return base + incl - v;
}

}  // namespace cusrl
