// Bootstrap target (a3), GAE(lambda) scan + return + fused advantage statistics (a4), column statistics,
// normalisation (a5) and the cross-rank mean/var merge (a6) for gfx950.
//
// Numerics contract: the GAE recurrence uses separately rounded fp32 multiply and add (__fmul_rn/__fadd_rn,
// file built with -ffp-contract=off) in exactly the reference's association order, which makes advantage and
// return BIT-EXACT with cusrl/hook/on_policy/gae.py:8-20.  Statistics accumulate in fp64 with a fixed
// (launch-shape-determined) summation order, so results are run-to-run deterministic.
#include <stdlib.h>

#include "common.hpp"

namespace cusrl {

// --------------------------------------------------------------------------------------------- next_value
constexpr int kFlagChunk = kBlock * 16;  // must equal buffer.hip's chunk: block b owns slots [b*4096, (b+1)*4096)

// next_value_flat[i] = value_flat[i + N*D] for slots before the last step, last_value otherwise: a shifted copy
// along the flattened [T*N*D] axis, then per-slot overrides from the 1-byte flags.
template <bool kVec4>
__global__ __launch_bounds__(kBlock) void next_value_kernel(const float *__restrict__ value,
                                                            const uint8_t *__restrict__ terminated,
                                                            const uint8_t *__restrict__ truncated,
                                                            const float *__restrict__ last_value,
                                                            float termination_value, int truncated_mode,
                                                            float *__restrict__ next_value,
                                                            int32_t *__restrict__ block_counts, int64_t S, int64_t N,
                                                            int64_t D) {
    __shared__ int scratch[kWavesPerBlock];
    const int64_t chunk0 = int64_t(blockIdx.x) * kFlagChunk;
    const int64_t last0 = S - N;  // first slot of the last time step
    int trunc_count = 0;
    if constexpr (kVec4) {
        // D == 1, N % 4 == 0, all bases 16 B aligned: 4 slots per lane per iteration, fully coalesced.
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int64_t s = chunk0 + int64_t(it) * (kBlock * 4) + int64_t(threadIdx.x) * 4;
            if (s >= S) break;
            float4 v = s < last0 ? *reinterpret_cast<const float4 *>(value + s + N)
                                 : *reinterpret_cast<const float4 *>(last_value + (s - last0));
            const uint32_t te = *reinterpret_cast<const uint32_t *>(terminated + s);
            const uint32_t tr = *reinterpret_cast<const uint32_t *>(truncated + s);
            float out[4] = {v.x, v.y, v.z, v.w};
            float4 own;
            if (truncated_mode == 1 && tr) own = *reinterpret_cast<const float4 *>(value + s);
            const float ownv[4] = {own.x, own.y, own.z, own.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((te >> (8 * j)) & 0xffu) out[j] = termination_value;
                if ((tr >> (8 * j)) & 0xffu) {
                    ++trunc_count;
                    if (truncated_mode == 1) out[j] = ownv[j];
                }
            }
            *reinterpret_cast<float4 *>(next_value + s) = make_float4(out[0], out[1], out[2], out[3]);
        }
    } else {
        for (int it = 0; it < 16; ++it) {
            const int64_t s = chunk0 + int64_t(it) * kBlock + threadIdx.x;
            if (s >= S) break;
            const bool te = terminated[s] != 0, tr = truncated[s] != 0;
            trunc_count += tr;
            for (int64_t d = 0; d < D; ++d) {
                float v = s < last0 ? value[(s + N) * D + d] : last_value[(s - last0) * D + d];
                if (te) v = termination_value;
                if (tr && truncated_mode == 1) v = value[s * D + d];
                next_value[s * D + d] = v;
            }
        }
    }
    const int total = block_sum(trunc_count, scratch);
    if (block_counts && threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// --------------------------------------------------------------------------------------------- GAE
// One lane owns VEC adjacent columns of the [T, C] (C = N*D) matrices and walks time backwards.  Loads along
// the env axis are unit-stride across lanes (16 B per lane when VEC == 4).  The horizon is consumed in chunks of
// TC steps: all 4 input streams of a chunk are issued before the first dependent arithmetic, so each lane keeps
// TC * (3 * VEC dwords + VEC bytes) in flight — the recurrence itself is only 2 flops per element.
template <int VEC>
struct Vec;
template <>
struct Vec<1> {
    using type = float;
};
template <>
struct Vec<4> {
    using type = float4;
};

template <int VEC>
__device__ __forceinline__ void unpack(const typename Vec<VEC>::type &v, float (&out)[VEC]) {
    if constexpr (VEC == 1) {
        out[0] = v;
    } else {
        out[0] = v.x, out[1] = v.y, out[2] = v.z, out[3] = v.w;
    }
}

template <int VEC>
__device__ __forceinline__ typename Vec<VEC>::type pack(const float (&in)[VEC]) {
    if constexpr (VEC == 1) {
        return in[0];
    } else {
        return make_float4(in[0], in[1], in[2], in[3]);
    }
}

// Cache policy of the scan's streams (round 4, profiles/r04/gae_policy.md).  Every input element is read exactly once and
// every output element written exactly once; beyond the 256 MB Infinity Cache a default-policy launch spends its time
// evicting the predecessor's dirty lines to make room for lines nobody will touch again (measured: 0.40 of the roofline
// behind 1 GiB of fresh writes, 0.64 in a hot loop).  Non-temporal loads / stores stream past the caches.
//   kNtLoad   reward / value / next_value / done are loaded non-temporally
//   kNtAdv    advantage is stored non-temporally (NOT set when the normalisation pass reads it right back)
//   kNtRet    return is stored non-temporally (its next reader is a random minibatch gather, many launches later)
constexpr int kNtLoad = 1, kNtAdv = 2, kNtRet = 4;

typedef float native_float4 __attribute__((ext_vector_type(4)));

template <int VEC, bool NT>
__device__ __forceinline__ typename Vec<VEC>::type load_stream(const float *p) {
    if constexpr (!NT) {
        return *reinterpret_cast<const typename Vec<VEC>::type *>(p);
    } else if constexpr (VEC == 4) {
        const native_float4 v = __builtin_nontemporal_load(reinterpret_cast<const native_float4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return __builtin_nontemporal_load(p);
    }
}

template <int VEC, bool NT>
__device__ __forceinline__ void store_stream(float *p, const typename Vec<VEC>::type &v) {
    if constexpr (!NT) {
        *reinterpret_cast<typename Vec<VEC>::type *>(p) = v;
    } else if constexpr (VEC == 4) {
        const native_float4 n = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(n, reinterpret_cast<native_float4 *>(p));
    } else {
        __builtin_nontemporal_store(v, p);
    }
}

// BLK = threads per block: 128 at scale (finer blocks desynchronise the load and store phases of the one resident wave of
// blocks: +8 % at 4 M envs); 64 (one wave) for small rollouts, where the launch is pure latency and spreading the few
// columns over 4x more CUs — with the whole horizon requested in ONE load round (TC >= T) — is what counts.
template <int VEC, int TC, bool kTwoLambdas, int BLK = kBlock, int POLICY = 0>
__global__ __launch_bounds__(BLK) void gae_kernel(const float *__restrict__ reward, const float *__restrict__ value,
                                                     const float *__restrict__ next_value,
                                                     const uint8_t *__restrict__ done, float *__restrict__ advantage,
                                                     float *__restrict__ ret, double *__restrict__ partials, int T,
                                                     int64_t N, int D, float gamma, float c_adv, float c_val) {
    using V = typename Vec<VEC>::type;
    const int64_t C = N * D;
    const int64_t col = (int64_t(blockIdx.x) * BLK + threadIdx.x) * VEC;
    const bool active = col < C;
    // VEC == 4 is only launched with D == 1 (done index == column); otherwise the flag of column j is env j / D.
    const int64_t env = (VEC == 4 || D == 1) ? col : col / D;

    float carry_adv[VEC], carry_val[VEC];
    double sum = 0.0, sumsq = 0.0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) carry_adv[j] = carry_val[j] = 0.0f;

    if (active) {
        for (int t_hi = T; t_hi > 0; t_hi -= TC) {
            V r[TC], v[TC], nv[TC];
            uint32_t dn[TC];
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                const int t = t_hi - 1 - k;
                if (t >= 0) {
                    const int64_t off = int64_t(t) * C + col;
                    constexpr bool nt = (POLICY & kNtLoad) != 0;
                    r[k] = load_stream<VEC, nt>(reward + off);
                    v[k] = load_stream<VEC, nt>(value + off);
                    nv[k] = load_stream<VEC, nt>(next_value + off);
                    if constexpr (VEC == 4 && nt)
                        dn[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(done + int64_t(t) * N + env));
                    else if constexpr (VEC == 4)
                        dn[k] = *reinterpret_cast<const uint32_t *>(done + int64_t(t) * N + env);
                    else
                        dn[k] = done[int64_t(t) * N + env];
                }
            }
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                const int t = t_hi - 1 - k;
                if (t >= 0) {
                    float rr[VEC], vv[VEC], nn[VEC], aa[VEC], rt[VEC];
                    unpack<VEC>(r[k], rr);
                    unpack<VEC>(v[k], vv);
                    unpack<VEC>(nv[k], nn);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        // delta = (r + nv * gamma) - v                                    gae.py:15
                        const float delta = __fsub_rn(__fadd_rn(rr[j], __fmul_rn(nn[j], gamma)), vv[j]);
                        const bool is_done = ((dn[k] >> (8 * j)) & 0xffu) != 0;
                        float a = delta, av = delta;
                        if (t != T - 1) {
                            // A[t] += (not_done * (gamma*lamda)) * A[t+1]               gae.py:17
                            const float coef = is_done ? 0.0f : c_adv;
                            a = __fadd_rn(delta, __fmul_rn(coef, carry_adv[j]));
                            if constexpr (kTwoLambdas) {
                                const float coefv = is_done ? 0.0f : c_val;
                                av = __fadd_rn(delta, __fmul_rn(coefv, carry_val[j]));
                            }
                        }
                        carry_adv[j] = a;
                        if constexpr (kTwoLambdas) carry_val[j] = av;
                        aa[j] = a;
                        rt[j] = __fadd_rn(vv[j], kTwoLambdas ? av : a);  // return = value + A   gae.py:99-110
                        sum += double(a);
                        sumsq += double(a) * double(a);
                    }
                    const int64_t off = int64_t(t) * C + col;
                    store_stream<VEC, (POLICY & kNtAdv) != 0>(advantage + off, pack<VEC>(aa));
                    store_stream<VEC, (POLICY & kNtRet) != 0>(ret + off, pack<VEC>(rt));
                }
            }
        }
    }

    if (partials) {
        // per-block {sum, sumsq} per value channel, written to partials[blockIdx][d][2]
        __shared__ double red[BLK][2];
        if (D == 1) {
            __shared__ double scratch[BLK / kWave][2];
            const double s = wave_sum(sum), q = wave_sum(sumsq);
            if ((threadIdx.x & (kWave - 1)) == 0) scratch[threadIdx.x / kWave][0] = s, scratch[threadIdx.x / kWave][1] = q;
            __syncthreads();
            if (threadIdx.x == 0) {
                double ts = 0.0, tq = 0.0;
#pragma unroll
                for (int w = 0; w < BLK / kWave; ++w) ts += scratch[w][0], tq += scratch[w][1];
                partials[int64_t(blockIdx.x) * 2 + 0] = ts;
                partials[int64_t(blockIdx.x) * 2 + 1] = tq;
            }
        } else {
            red[threadIdx.x][0] = active ? sum : 0.0;
            red[threadIdx.x][1] = active ? sumsq : 0.0;
            __syncthreads();
            if (threadIdx.x < D) {
                // thread k of this block owns column base + k, channel (base + k) % D
                const int64_t base = int64_t(blockIdx.x) * BLK;
                int first = int((int64_t(threadIdx.x) - base % D + D) % D);
                double s = 0.0, q = 0.0;
                for (int k = first; k < BLK; k += D) {
                    s += red[k][0];
                    q += red[k][1];
                }
                partials[(int64_t(blockIdx.x) * D + threadIdx.x) * 2 + 0] = s;
                partials[(int64_t(blockIdx.x) * D + threadIdx.x) * 2 + 1] = q;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------- column statistics
constexpr int kStatsMaxBlocks = 1024;

__global__ __launch_bounds__(kBlock) void col_stats_kernel(const float *__restrict__ x, int64_t E, int D,
                                                           double *__restrict__ partials) {
    // E = rows * D flat elements; channel of element i is i % D.  A lane's stride is a multiple of D, so each
    // lane stays on one channel.
    double sum = 0.0, sumsq = 0.0;
    const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (D == 1 && aligned_ptr16(x)) {
        const int64_t n4 = E / 4;
        const int64_t stride = int64_t(gridDim.x) * kBlock;
        for (int64_t i = tid; i < n4; i += stride) {
            const float4 v = reinterpret_cast<const float4 *>(x)[i];
            sum += double(v.x) + double(v.y) + double(v.z) + double(v.w);
            sumsq += double(v.x) * double(v.x) + double(v.y) * double(v.y) + double(v.z) * double(v.z) +
                     double(v.w) * double(v.w);
        }
        if (tid == 0)
            for (int64_t i = n4 * 4; i < E; ++i) {
                sum += double(x[i]);
                sumsq += double(x[i]) * double(x[i]);
            }
    } else {
        const int64_t threads = int64_t(gridDim.x) * kBlock;
        const int64_t stride = threads / D * D;  // largest multiple of D; lanes >= stride idle (D <= 256)
        for (int64_t i = tid < stride ? tid : E; i < E; i += stride) {
            sum += double(x[i]);
            sumsq += double(x[i]) * double(x[i]);
        }
    }
    __shared__ double red[kBlock][2];
    if (D == 1) {
        __shared__ double scratch[kWavesPerBlock];
        const double s = block_sum(sum, scratch);
        const double q = block_sum(sumsq, scratch);
        if (threadIdx.x == 0) {
            partials[int64_t(blockIdx.x) * 2 + 0] = s;
            partials[int64_t(blockIdx.x) * 2 + 1] = q;
        }
    } else {
        red[threadIdx.x][0] = sum;
        red[threadIdx.x][1] = sumsq;
        __syncthreads();
        if (threadIdx.x < D) {
            const int64_t base = int64_t(blockIdx.x) * kBlock;
            int first = int((int64_t(threadIdx.x) - base % D + D) % D);
            double s = 0.0, q = 0.0;
            for (int k = first; k < kBlock; k += D) {
                s += red[k][0];
                q += red[k][1];
            }
            partials[(int64_t(blockIdx.x) * D + threadIdx.x) * 2 + 0] = s;
            partials[(int64_t(blockIdx.x) * D + threadIdx.x) * 2 + 1] = q;
        }
    }
}

__global__ __launch_bounds__(kBlock) void stats_finalize_kernel(const double *__restrict__ partials, int64_t P, int D,
                                                                int64_t count, float *__restrict__ mean,
                                                                float *__restrict__ var) {
    __shared__ double scratch[kWavesPerBlock];
    for (int d = 0; d < D; ++d) {
        double s = 0.0, q = 0.0;
        for (int64_t p = threadIdx.x; p < P; p += kBlock) {
            s += partials[(p * D + d) * 2 + 0];
            q += partials[(p * D + d) * 2 + 1];
        }
        s = block_sum(s, scratch);
        q = block_sum(q, scratch);
        if (threadIdx.x == 0) {
            const double n = double(count);
            const double m = s / n;
            // unbiased (correction = 1) like torch.var_mean; count == 1 gives 0/0 = nan like torch
            const double v = (q - s * m) / (n - 1.0);
            mean[d] = float(m);
            var[d] = float(v < 0.0 ? 0.0 : v);
        }
    }
}

// --------------------------------------------------------------------------------------------- normalise
__global__ __launch_bounds__(kBlock) void normalize_kernel(float *__restrict__ x, const float *__restrict__ mean,
                                                           const float *__restrict__ var, float eps, int64_t E, int D,
                                                           int vec4) {
    const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    if (D == 1) {
        const float m = mean[0];
        const float sd = sqrtf(__fadd_rn(var[0], eps));  // (var + 1e-8).sqrt()     advantage.py:114
        if (vec4) {
            const int64_t n4 = E / 4;
            for (int64_t i = tid; i < n4; i += stride) {
                float4 v = reinterpret_cast<float4 *>(x)[i];
                v.x = __fdiv_rn(__fsub_rn(v.x, m), sd);  // sub_(mean).div_(std)         advantage.py:115
                v.y = __fdiv_rn(__fsub_rn(v.y, m), sd);
                v.z = __fdiv_rn(__fsub_rn(v.z, m), sd);
                v.w = __fdiv_rn(__fsub_rn(v.w, m), sd);
                reinterpret_cast<float4 *>(x)[i] = v;
            }
            if (tid == 0)
                for (int64_t i = n4 * 4; i < E; ++i) x[i] = __fdiv_rn(__fsub_rn(x[i], m), sd);
        } else {
            for (int64_t i = tid; i < E; i += stride) x[i] = __fdiv_rn(__fsub_rn(x[i], m), sd);
        }
    } else {
        for (int64_t i = tid; i < E; i += stride) {
            const int d = int(i % D);
            const float sd = sqrtf(__fadd_rn(var[d], eps));
            x[i] = __fdiv_rn(__fsub_rn(x[i], mean[d]), sd);
        }
    }
}

// Normalisation straight from the block partials (single-process case: no cross-rank merge sits between the statistics
// and their use): every block re-reduces the few partial rows itself (P x D x 16 bytes, L2-resident) in the same fixed
// order as stats_finalize_kernel, so the separate one-block finalize launch disappears; block 0 also publishes mean / var.
__global__ __launch_bounds__(kBlock) void normalize_from_partials_kernel(float *__restrict__ x,
                                                                         const double *__restrict__ partials, int64_t P,
                                                                         int64_t count, float eps, int64_t E, int D,
                                                                         int vec4, float *__restrict__ mean_out,
                                                                         float *__restrict__ var_out) {
    __shared__ double scratch[kWavesPerBlock];
    __shared__ float s_mean[kBlock], s_sd[kBlock];
    for (int d = 0; d < D; ++d) {
        double s = 0.0, q = 0.0;
        for (int64_t p = threadIdx.x; p < P; p += kBlock) {
            s += partials[(p * D + d) * 2 + 0];
            q += partials[(p * D + d) * 2 + 1];
        }
        s = block_sum(s, scratch);
        q = block_sum(q, scratch);
        if (threadIdx.x == 0) {
            const double n = double(count);
            const double m = s / n;
            const double v = (q - s * m) / (n - 1.0);
            const float mean = float(m), var = float(v < 0.0 ? 0.0 : v);
            s_mean[d] = mean;
            s_sd[d] = sqrtf(__fadd_rn(var, eps));
            if (blockIdx.x == 0) mean_out[d] = mean, var_out[d] = var;
        }
    }
    __syncthreads();
    const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    if (D == 1) {
        const float m = s_mean[0], sd = s_sd[0];
        if (vec4) {
            const int64_t n4 = E / 4;
            for (int64_t i = tid; i < n4; i += stride) {
                float4 v = reinterpret_cast<float4 *>(x)[i];
                v.x = __fdiv_rn(__fsub_rn(v.x, m), sd);
                v.y = __fdiv_rn(__fsub_rn(v.y, m), sd);
                v.z = __fdiv_rn(__fsub_rn(v.z, m), sd);
                v.w = __fdiv_rn(__fsub_rn(v.w, m), sd);
                reinterpret_cast<float4 *>(x)[i] = v;
            }
            if (tid == 0)
                for (int64_t i = n4 * 4; i < E; ++i) x[i] = __fdiv_rn(__fsub_rn(x[i], m), sd);
        } else {
            for (int64_t i = tid; i < E; i += stride) x[i] = __fdiv_rn(__fsub_rn(x[i], m), sd);
        }
    } else {
        for (int64_t i = tid; i < E; i += stride) {
            const int d = int(i % D);
            x[i] = __fdiv_rn(__fsub_rn(x[i], s_mean[d]), s_sd[d]);
        }
    }
}

// --------------------------------------------------------------------------------------------- merge
__global__ void merge_mean_var_kernel(const float *__restrict__ gathered, int W, int D, float *__restrict__ mean,
                                      float *__restrict__ var) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    // torch.mean(all_means, dim=0); torch.mean(all_vars + (all_means - mean).square(), dim=0)
    float s = 0.0f;
    for (int r = 0; r < W; ++r) s = __fadd_rn(s, gathered[int64_t(r) * 2 * D + d]);
    const float m = __fdiv_rn(s, float(W));
    float q = 0.0f;
    for (int r = 0; r < W; ++r) {
        const float e = __fsub_rn(gathered[int64_t(r) * 2 * D + d], m);
        q = __fadd_rn(q, __fadd_rn(gathered[int64_t(r) * 2 * D + D + d], __fmul_rn(e, e)));
    }
    mean[d] = m;
    var[d] = __fdiv_rn(q, float(W));
}

// Merge + normalise in one launch (a job with several ranks: the all-gathered [W, 2D] rows of every rank's mean | var sit
// between the statistics and their use): every block re-derives the merged statistics with merge_mean_var_kernel's operations in
// its order (W x 2D floats, L2-resident), then normalises its share like normalize_kernel; block 0 publishes mean / var.
__global__ __launch_bounds__(kBlock) void normalize_from_gathered_kernel(float *__restrict__ x, const float *__restrict__ gathered,
                                                                         int W, float eps, int64_t E, int D, int vec4,
                                                                         float *__restrict__ mean_out, float *__restrict__ var_out) {
    __shared__ float s_mean[kBlock], s_sd[kBlock];
    if (threadIdx.x < D) {
        const int d = threadIdx.x;
        float s = 0.0f;
        for (int r = 0; r < W; ++r) s = __fadd_rn(s, gathered[int64_t(r) * 2 * D + d]);
        const float m = __fdiv_rn(s, float(W));
        float q = 0.0f;
        for (int r = 0; r < W; ++r) {
            const float e = __fsub_rn(gathered[int64_t(r) * 2 * D + d], m);
            q = __fadd_rn(q, __fadd_rn(gathered[int64_t(r) * 2 * D + D + d], __fmul_rn(e, e)));
        }
        const float var = __fdiv_rn(q, float(W));
        s_mean[d] = m;
        s_sd[d] = sqrtf(__fadd_rn(var, eps));
        if (blockIdx.x == 0) mean_out[d] = m, var_out[d] = var;
    }
    __syncthreads();
    const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    if (D == 1) {
        const float m = s_mean[0], sd = s_sd[0];
        if (vec4) {
            const int64_t n4 = E / 4;
            for (int64_t i = tid; i < n4; i += stride) {
                float4 v = reinterpret_cast<float4 *>(x)[i];
                v.x = __fdiv_rn(__fsub_rn(v.x, m), sd);
                v.y = __fdiv_rn(__fsub_rn(v.y, m), sd);
                v.z = __fdiv_rn(__fsub_rn(v.z, m), sd);
                v.w = __fdiv_rn(__fsub_rn(v.w, m), sd);
                reinterpret_cast<float4 *>(x)[i] = v;
            }
            if (tid == 0)
                for (int64_t i = n4 * 4; i < E; ++i) x[i] = __fdiv_rn(__fsub_rn(x[i], m), sd);
        } else {
            for (int64_t i = tid; i < E; i += stride) x[i] = __fdiv_rn(__fsub_rn(x[i], m), sd);
        }
    } else {
        for (int64_t i = tid; i < E; i += stride) {
            const int d = int(i % D);
            x[i] = __fdiv_rn(__fsub_rn(x[i], s_mean[d]), s_sd[d]);
        }
    }
}

}  // namespace cusrl

using namespace cusrl;

extern "C" int cusrl_next_value(const float *value, const uint8_t *terminated, const uint8_t *truncated,
                                const float *last_value, float termination_value, int truncated_mode,
                                float *next_value, int32_t *block_counts, int64_t T, int64_t N, int64_t D,
                                void *stream) {
    if (T < 0 || N < 0 || D < 0) return CUSRL_E_INVALID;
    const int64_t S = T * N;
    if (S == 0 || D == 0) return 0;
    if (!value || !terminated || !truncated || !last_value || !next_value) return CUSRL_E_INVALID;
    const int64_t blocks = ceil_div(S, kFlagChunk);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const bool vec4 = D == 1 && N % 4 == 0 && aligned(value, 16) && aligned(last_value, 16) &&
                      aligned(next_value, 16) && aligned(terminated, 4) && aligned(truncated, 4);
    if (vec4)
        hipLaunchKernelGGL(next_value_kernel<true>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), value,
                           terminated, truncated, last_value, termination_value, truncated_mode, next_value,
                           block_counts, S, N, D);
    else
        hipLaunchKernelGGL(next_value_kernel<false>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream),
                           value, terminated, truncated, last_value, termination_value, truncated_mode, next_value,
                           block_counts, S, N, D);
    return launch_status();
}

static bool gae_vec4(const float *reward, const float *value, const float *next_value, const uint8_t *done,
                     float *advantage, float *ret, int64_t N, int64_t D) {
    return D == 1 && N % 4 == 0 && aligned(reward, 16) && aligned(value, 16) && aligned(next_value, 16) &&
           aligned(advantage, 16) && aligned(ret, 16) && aligned(done, 4);
}

// ---- launch shape of the at-scale scan (measured: profiles/r04/gae_policy.md)
// Cache policy by footprint.  While the six streams (21 B/slot) fit the 256 MB Infinity Cache the default policy is the
// fastest; beyond it the inputs and `return` stream past the caches, and `advantage` — which the normalisation pass reads
// right back — stays cached while it fits half of the Infinity Cache, else it streams too.
// cusrl_set_option("gae_policy", 1 + {0, 5, 7}) forces one of the three instantiated policies (A/B measurements,
// scripts/pmc_r04_cases.py).
static int gae_policy(int64_t slots) {
    const int forced = int(option(kOptGaePolicy)) - 1;
    if (forced == 0 || forced == (kNtLoad | kNtRet) || forced == (kNtLoad | kNtAdv | kNtRet)) return forced;
    if (slots * 21 < (int64_t(256) << 20)) return 0;
    return slots * 4 <= (int64_t(128) << 20) ? (kNtLoad | kNtRet) : (kNtLoad | kNtAdv | kNtRet);
}
// 256-thread blocks while ONE resident wave of blocks covers the columns (<= 5 blocks per CU at ~100 VGPRs); beyond
// that 128-thread blocks backfill at a finer grain (+8 % at 4 M envs).  cusrl_set_option("gae_block", 128 | 256) forces one.
static int gae_block(int64_t columns) {
    const int forced = int(option(kOptGaeBlock));
    if (forced == 128 || forced == 256) return forced;
    return ceil_div(columns / 4, 256) <= 5 * 256 ? 256 : 128;
}

template <int POLICY, int BLK, bool TWO>
static void launch_gae_one(uint32_t blocks, hipStream_t s, const float *reward, const float *value, const float *next_value,
                           const uint8_t *done, float *advantage, float *ret, double *partials, int T, int64_t N, int D,
                           float g, float c_adv, float c_val) {
    hipLaunchKernelGGL((gae_kernel<4, 6, TWO, BLK, POLICY>), dim3(blocks), dim3(BLK), 0, s, reward, value, next_value, done,
                       advantage, ret, partials, T, N, D, g, c_adv, c_val);
}

template <int POLICY>
static void launch_gae_policy(int blk, bool two, uint32_t blocks, hipStream_t s, const float *reward, const float *value,
                              const float *next_value, const uint8_t *done, float *advantage, float *ret, double *partials,
                              int T, int64_t N, int D, float g, float c_adv, float c_val) {
#define CUSRL_GAE_ARGS blocks, s, reward, value, next_value, done, advantage, ret, partials, T, N, D, g, c_adv, c_val
    if (blk == 256)
        two ? launch_gae_one<POLICY, 256, true>(CUSRL_GAE_ARGS) : launch_gae_one<POLICY, 256, false>(CUSRL_GAE_ARGS);
    else
        two ? launch_gae_one<POLICY, 128, true>(CUSRL_GAE_ARGS) : launch_gae_one<POLICY, 128, false>(CUSRL_GAE_ARGS);
#undef CUSRL_GAE_ARGS
}

static void launch_gae_scaled(int policy, int blk, bool two, uint32_t blocks, hipStream_t s, const float *reward,
                              const float *value, const float *next_value, const uint8_t *done, float *advantage, float *ret,
                              double *partials, int T, int64_t N, int D, float g, float c_adv, float c_val) {
#define CUSRL_GAE_ARGS blk, two, blocks, s, reward, value, next_value, done, advantage, ret, partials, T, N, D, g, c_adv, c_val
    switch (policy) {
        case 0: launch_gae_policy<0>(CUSRL_GAE_ARGS); break;
        case kNtLoad | kNtRet: launch_gae_policy<kNtLoad | kNtRet>(CUSRL_GAE_ARGS); break;
        default: launch_gae_policy<kNtLoad | kNtAdv | kNtRet>(CUSRL_GAE_ARGS); break;
    }
#undef CUSRL_GAE_ARGS
}

// Partial rows no block of the chosen launch shape writes.  A KERNEL, not hipMemsetAsync: a memset becomes a memset NODE when
// the caller is being captured into a hipGraph, and this repo keeps captured regions free of those.
__global__ __launch_bounds__(cusrl::kBlock) void zero_f64_kernel(double *__restrict__ p, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * cusrl::kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * cusrl::kBlock) p[i] = 0.0;
}

static void zero_partial_rows(double *p, int64_t n, hipStream_t s) {
    const int64_t blocks = cusrl::ceil_div(n, cusrl::kBlock);
    hipLaunchKernelGGL(zero_f64_kernel, dim3(uint32_t(blocks > 1024 ? 1024 : blocks)), dim3(cusrl::kBlock), 0, s, p, n);
}

extern "C" int64_t cusrl_gae_num_partials(int64_t T, int64_t N, int64_t D) {
    (void)T;
    if (N <= 0 || D <= 0) return 0;
    // upper bound valid for every launch shape (one-wave blocks of the small-rollout path included)
    return ceil_div(N * D, kWave);
}

extern "C" int cusrl_gae(const float *reward, const float *value, const float *next_value, const uint8_t *done,
                         float *advantage, float *ret, double *stat_partials, int64_t T, int64_t N, int64_t D,
                         double gamma, double lamda, double lamda_value, void *stream) {
    if (T < 0 || N < 0 || D < 0) return CUSRL_E_INVALID;
    if (T == 0 || N == 0 || D == 0) return 0;
    if (!reward || !value || !next_value || !done || !advantage || !ret) return CUSRL_E_INVALID;
    if (T > INT32_MAX || D > kBlock) return CUSRL_E_UNSUPPORTED;
    const float g = float(gamma);
    const float c_adv = float(gamma * lamda);  // fp32(double(gamma) * double(lamda)), as Python evaluates it
    const bool two = lamda_value >= 0.0;
    const float c_val = two ? float(gamma * lamda_value) : c_adv;
    const int64_t C = N * D;
    hipStream_t s = as_stream(stream);
    // 4 columns per lane only when that still fills the chip (>= 256 blocks); small rollouts (config 2: 4096 envs)
    // are latency-bound and want every lane they can get
    if (gae_vec4(reward, value, next_value, done, advantage, ret, N, D) && C >= int64_t(4) * kBlock * 256) {
        const int policy = gae_policy(T * C);
        const int blk = gae_block(C);
        const int64_t blocks = ceil_div(C / 4, blk);
        // the host sizes `stat_partials` with cusrl_gae_num_partials (>= blocks); unused rows are zeroed
        if (stat_partials) {
            const int64_t rows = cusrl_gae_num_partials(T, N, D);
            if (rows > blocks) zero_partial_rows(stat_partials + blocks * D * 2, (rows - blocks) * D * 2, s);
        }
        launch_gae_scaled(policy, blk, two, uint32_t(blocks), s, reward, value, next_value, done, advantage, ret,
                          stat_partials, int(T), N, int(D), g, c_adv, c_val);
    } else if (C <= 65536 && T <= 32) {
        // small rollouts (config 2: 4096 columns x 24 steps): one-wave blocks on 4x more CUs, the whole horizon in ONE
        // load round (32 steps x 4 streams in flight per lane) — the launch is one memory latency + the scan
        const int64_t blocks = ceil_div(C, kWave);
        if (stat_partials) {
            const int64_t rows = cusrl_gae_num_partials(T, N, D);
            if (rows > blocks) zero_partial_rows(stat_partials + blocks * D * 2, (rows - blocks) * D * 2, s);
        }
        if (two)
            hipLaunchKernelGGL((gae_kernel<1, 32, true, kWave>), dim3(uint32_t(blocks)), dim3(kWave), 0, s, reward, value,
                               next_value, done, advantage, ret, stat_partials, int(T), N, int(D), g, c_adv, c_val);
        else
            hipLaunchKernelGGL((gae_kernel<1, 32, false, kWave>), dim3(uint32_t(blocks)), dim3(kWave), 0, s, reward, value,
                               next_value, done, advantage, ret, stat_partials, int(T), N, int(D), g, c_adv, c_val);
    } else {
        const int64_t blocks = ceil_div(C, kBlock);
        if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
        if (stat_partials) {
            const int64_t rows = cusrl_gae_num_partials(T, N, D);
            if (rows > blocks) zero_partial_rows(stat_partials + blocks * D * 2, (rows - blocks) * D * 2, s);
        }
        if (two)
            hipLaunchKernelGGL((gae_kernel<1, 8, true>), dim3(uint32_t(blocks)), dim3(kBlock), 0, s, reward, value,
                               next_value, done, advantage, ret, stat_partials, int(T), N, int(D), g, c_adv, c_val);
        else
            hipLaunchKernelGGL((gae_kernel<1, 8, false>), dim3(uint32_t(blocks)), dim3(kBlock), 0, s, reward, value,
                               next_value, done, advantage, ret, stat_partials, int(T), N, int(D), g, c_adv, c_val);
    }
    return launch_status();
}

extern "C" int64_t cusrl_col_stats_num_partials(int64_t rows, int64_t D) {
    if (rows <= 0 || D <= 0) return 0;
    const int64_t want = ceil_div(rows * D, int64_t(kBlock) * 16);
    return want < 1 ? 1 : (want > kStatsMaxBlocks ? kStatsMaxBlocks : want);
}

extern "C" int cusrl_col_stats(const float *x, int64_t rows, int64_t D, double *stat_partials, void *stream) {
    if (rows < 0 || D < 0) return CUSRL_E_INVALID;
    if (rows == 0 || D == 0) return 0;
    if (!x || !stat_partials) return CUSRL_E_INVALID;
    if (D > kBlock) return CUSRL_E_UNSUPPORTED;
    const int64_t blocks = cusrl_col_stats_num_partials(rows, D);
    hipLaunchKernelGGL(col_stats_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), x, rows * D,
                       int(D), stat_partials);
    return launch_status();
}

extern "C" int cusrl_stats_finalize(const double *stat_partials, int64_t num_partials, int64_t D, int64_t count,
                                    float *mean, float *var, void *stream) {
    if (num_partials < 0 || D < 0 || count < 0) return CUSRL_E_INVALID;
    if (D == 0) return 0;
    if (!stat_partials || !mean || !var) return CUSRL_E_INVALID;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), stat_partials,
                       num_partials, int(D), count, mean, var);
    return launch_status();
}

extern "C" int cusrl_normalize(float *x, const float *mean, const float *var, float eps, int64_t rows, int64_t D,
                               void *stream) {
    if (rows < 0 || D < 0) return CUSRL_E_INVALID;
    if (rows == 0 || D == 0) return 0;
    if (!x || !mean || !var) return CUSRL_E_INVALID;
    const int64_t E = rows * D;
    const int vec4 = D == 1 && aligned(x, 16);
    int64_t blocks = ceil_div(vec4 ? E / 4 + 1 : E, kBlock);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(normalize_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), x, mean, var, eps,
                       E, int(D), vec4);
    return launch_status();
}

extern "C" int cusrl_normalize_from_partials(float *x, const double *stat_partials, int64_t num_partials, int64_t count,
                                             float eps, int64_t rows, int64_t D, float *mean_out, float *var_out,
                                             void *stream) {
    if (rows < 0 || D < 0 || num_partials < 0 || count < 0) return CUSRL_E_INVALID;
    if (rows == 0 || D == 0) return 0;
    if (!x || !stat_partials || !mean_out || !var_out) return CUSRL_E_INVALID;
    if (D > kBlock) return CUSRL_E_UNSUPPORTED;
    const int64_t E = rows * D;
    const int vec4 = D == 1 && aligned(x, 16);
    int64_t blocks = ceil_div(vec4 ? E / 4 + 1 : E, kBlock);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(normalize_from_partials_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), x,
                       stat_partials, num_partials, count, eps, E, int(D), vec4, mean_out, var_out);
    return launch_status();
}

extern "C" int cusrl_merge_mean_var(const float *gathered, int64_t W, int64_t D, float *mean, float *var,
                                    void *stream) {
    if (W <= 0 || D < 0) return CUSRL_E_INVALID;
    if (D == 0) return 0;
    if (!gathered || !mean || !var) return CUSRL_E_INVALID;
    hipLaunchKernelGGL(merge_mean_var_kernel, dim3(uint32_t(ceil_div(D, 64))), dim3(64), 0, as_stream(stream),
                       gathered, int(W), int(D), mean, var);
    return launch_status();
}

extern "C" int cusrl_normalize_from_gathered(float *x, const float *gathered, int64_t W, float eps, int64_t rows, int64_t D,
                                             float *mean_out, float *var_out, void *stream) {
    if (rows < 0 || D < 0 || W <= 0) return CUSRL_E_INVALID;
    if (rows == 0 || D == 0) return 0;
    if (!x || !gathered || !mean_out || !var_out) return CUSRL_E_INVALID;
    if (D > kBlock) return CUSRL_E_UNSUPPORTED;
    const int64_t E = rows * D;
    const int vec4 = D == 1 && aligned(x, 16);
    int64_t blocks = ceil_div(vec4 ? E / 4 + 1 : E, kBlock);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(normalize_from_gathered_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), x, gathered,
                       int(W), eps, E, int(D), vec4, mean_out, var_out);
    return launch_status();
}
