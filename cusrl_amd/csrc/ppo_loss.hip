// Fused PPO objective for gfx950 (a9-a13): Normal log-prob + entropy + probability ratio, clipped surrogate,
// (clipped) value loss and entropy bonus, forward AND backward in one pass over the minibatch.
//
// Reads  advantage 4 + old_logp 4 + action/mean/std 3*4A + return 4D + curr_value 4D (+ old_value 4D) bytes,
// writes d_mean/d_std 2*4A + d_value 4D bytes per sample (264 B at A=12, D=1) — HBM-bound, ~60 flops/sample.
//
// Layout strategy (A % 4 == 0): the [B, A] matrices are streamed as a flat array of 16-byte chunks so that
// every global access is a fully coalesced dwordx4 (lane i <-> chunk i), independent of the row length.  A
// block owns 256 rows = 256*LPR chunks (LPR = A/4 chunks per row); per-chunk partial log-prob / entropy sums
// are exchanged through LDS so that one lane per row finishes the scalar part (ratio, clipping, value loss),
// publishes d(loss)/d(logp) through LDS, and the chunk owners (who still hold action/mean/std in registers)
// emit the gradients.  Scalar statistics use wave64 shuffle reductions, fp64 partials per block, fixed order.
#include <float.h>

#include "common.hpp"

namespace cusrl {

struct LossParams {
    float lo, hi;            // fp32(1 - clip), fp32(1 + clip)                         ppo.py:16
    float value_clip;        // < 0: plain MSE                                          value.py:131-135
    float g_sur, g_ent, g_val;  // d loss / d(min term), d(entropy_b), d(sq err) incl. weights and 1/B
    float w_sur, w_val, w_ent;
};

__device__ __forceinline__ float log_sqrt_2pi() { return 0.918938533204672741780329736406f; }  // log(sqrt(2 pi))
__device__ __forceinline__ float entropy_const() { return 1.418938533204672741780329736406f; }  // 0.5 + 0.5 log(2 pi)

// Per-row scalar part shared by both kernels.  Returns d(loss)/d(logp_row).
__device__ __forceinline__ float row_terms(float logp, float entropy, float old_logp, float adv, const LossParams &p,
                                           double &sur_acc, double &ent_acc, double &abs_lr_acc, float &ratio_out,
                                           float &lr_out) {
    const float lr = logp - old_logp;                 // action_logp_ratio        common.py:35
    abs_lr_acc += double(fabsf(lr));                  // metric `ratio` = |logp ratio|   common.py:47
    const float ratio = expf(lr);                     // action_prob_ratio        common.py:41
    const float s1 = adv * ratio;                     // ppo.py:14
    const float rc = fminf(fmaxf(ratio, p.lo), p.hi); // clamp                    ppo.py:16
    const float s2 = adv * rc;
    sur_acc += double(fminf(s1, s2));
    ent_acc += double(entropy);
    const bool inside = ratio >= p.lo && ratio <= p.hi;
    float d_ratio;  // autograd of min(): ties split evenly, clamp passes on the closed interval
    if (s1 < s2)
        d_ratio = adv;
    else if (s1 > s2)
        d_ratio = inside ? adv : 0.0f;
    else
        d_ratio = 0.5f * adv + (inside ? 0.5f * adv : 0.0f);
    ratio_out = ratio;
    lr_out = lr;
    return p.g_sur * d_ratio * ratio;
}

// One value channel: (clipped) squared error, its gradient, the `value` metric.  `v` (the old value) is only read in
// the clipped form.
__device__ __forceinline__ void value_term(float cv, float R, float v, float *__restrict__ d_value_slot,
                                           const LossParams &p, double &val_acc, double &value_sum_acc) {
    value_sum_acc += double(cv);  // metric `value` = curr_value.sum(-1)        value.py:141
    const float e1 = cv - R, l1 = e1 * e1, g1 = 2.0f * e1;
    float g;
    if (p.value_clip < 0.0f) {
        val_acc += double(l1);  // mse_loss(return, curr_value)             value.py:132
        g = g1;
    } else {
        const float c = p.value_clip;
        const float dv = cv - v;
        const float dvc = fminf(fmaxf(dv, -c), c);
        const float e2 = (v + dvc) - R, l2 = e2 * e2;   // value.py:85-89
        const float g2 = (dv >= -c && dv <= c) ? 2.0f * e2 : 0.0f;
        val_acc += double(fmaxf(l1, l2));
        g = l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * (g1 + g2));
    }
    if (d_value_slot) *d_value_slot = p.g_val * g;
}

__device__ __forceinline__ void value_terms(const float *__restrict__ ret, const float *__restrict__ curr_value,
                                            const float *__restrict__ old_value, float *__restrict__ d_value,
                                            int64_t row, int D, const LossParams &p, double &val_acc,
                                            double &value_sum_acc) {
    for (int d = 0; d < D; ++d) {
        const int64_t i = row * D + d;
        value_term(curr_value[i], ret[i], p.value_clip < 0.0f ? 0.0f : old_value[i], d_value ? d_value + i : nullptr, p,
                   val_acc, value_sum_acc);
    }
}

constexpr int kLossSums = 5;  // value loss, surrogate, entropy, |logp ratio|, value

// All five sums through ONE LDS exchange (one barrier instead of two per sum): wave-shuffle each, lane 0 of every
// wave parks its five totals, the first five threads add the four waves up in fixed order.
template <bool kPublish = false>
__device__ __forceinline__ void write_block_partials(const double (&acc)[kLossSums], double *__restrict__ partials) {
    __shared__ double scratch[kWavesPerBlock][kLossSums];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < kLossSums; ++k) {
        const double total = wave_sum(acc[k]);
        if (lane == 0) scratch[wave][k] = total;
    }
    __syncthreads();
    if (threadIdx.x < kLossSums) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) total += scratch[w][threadIdx.x];
        double *slot = partials + int64_t(blockIdx.x) * kLossSums + threadIdx.x;
        if (kPublish)  // 8-byte agent-scope store: written through, visible to the finishing block without any fence
            __hip_atomic_store(slot, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            *slot = total;
    }
}

// Hand-off of the per-block partial rows to whichever block finishes last, WITHOUT release / acquire fences: an
// agent-scope release writes back every dirty line of the XCD's L2 — here the ~70 KB of gradients each block has just
// stored — and measured slower than the separate finalize launch it was meant to save.  Instead the few partial values
// are published with 8-byte agent-scope stores (write-through) and read back with agent-scope loads (L1 bypass), the
// "8-byte agent atomics on both sides" form of the MI355X guide; the ticket is a relaxed agent-scope counter taken
// after a barrier (which waits for the stores of every wave of the block).
__device__ __forceinline__ double load_published(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_published(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kRowsPerBlock = kBlock;

template <bool kPublished = false>
__device__ __forceinline__ void reduce_std_rows(const float *in, int64_t rows, int A, float *__restrict__ out);
template <bool kPublished = false>
__device__ __forceinline__ void finalize_losses(const double *partials, int64_t P, int64_t B, int D,
                                                const LossParams &p, float *__restrict__ losses_out,
                                                const float *d_std_partials, int A, float *__restrict__ d_std_vector);

// Last-block-done hand-off: every block publishes its partial rows, then takes a ticket; the block that draws the
// last one reduces all rows in the same launch (no separate 1-block finalize launch).  The ticket re-arms itself, so
// replayed hipGraphs need no memset.
__device__ __forceinline__ bool last_block_done(unsigned int *__restrict__ ticket) {
    __shared__ int is_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's published stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int drawn = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = drawn == gridDim.x - 1;
        if (is_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return is_last != 0;
}

// kStdVec: `std` is ONE row [A] shared by every sample (a state-independent std vector, distribution.py:228-247)
// instead of a [B, A] matrix: it is read from L1 instead of streamed, and d_std leaves the kernel as per-block
// column sums [A] (the gradient of the vector) instead of a [B, A] matrix that a sum(0) launch would have to reduce —
// 96 of the 264 bytes per sample disappear.
constexpr int kStdRowGroups = 16;

template <int LPR, bool kStdVec>
__global__ __launch_bounds__(kBlock) void ppo_loss_chunked_kernel(
    const float *__restrict__ advantage, const float *__restrict__ old_logp, const float *__restrict__ action,
    const float *__restrict__ mean, const float *__restrict__ std, const float *__restrict__ ret,
    const float *__restrict__ curr_value, const float *__restrict__ old_value, int64_t B, int D, LossParams p,
    float *__restrict__ logp_out, float *__restrict__ entropy_out, float *__restrict__ lr_out,
    float *__restrict__ ratio_out, float *__restrict__ d_mean, float *__restrict__ d_std,
    float *__restrict__ d_value, double *__restrict__ partials, float *__restrict__ d_std_partials,
    unsigned int *__restrict__ ticket, float *__restrict__ losses_out) {
    __shared__ float lp_part[kRowsPerBlock * LPR];
    __shared__ float en_part[kRowsPerBlock * LPR];
    __shared__ float dlp_row[kRowsPerBlock];
    __shared__ float4 ds_wave[kStdVec ? kWavesPerBlock * LPR : 1];  // per-wave column sums of d_std
    const int64_t row0 = int64_t(blockIdx.x) * kRowsPerBlock;
    const int64_t chunk0 = row0 * LPR;
    const int64_t total_chunks = B * LPR;
    const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(action);
    const float4 *__restrict__ m4 = reinterpret_cast<const float4 *>(mean);
    const float4 *__restrict__ s4 = reinterpret_cast<const float4 *>(std);

    // the per-row scalars of the second phase are requested together with the matrix chunks: one memory round trip
    // per block instead of two dependent ones (the row phase used to start its own loads behind the first barrier)
    const int64_t my_row = row0 + threadIdx.x;
    const bool row_ok = my_row < B;
    float pre_old_logp = 0.f, pre_adv = 0.f, pre_ret = 0.f, pre_cv = 0.f, pre_ov = 0.f;
    if (row_ok) {
        pre_old_logp = old_logp[my_row];
        pre_adv = advantage[my_row];
        if (D == 1) {
            pre_ret = ret[my_row];
            pre_cv = curr_value[my_row];
            if (p.value_clip >= 0.0f) pre_ov = old_value[my_row];
        }
    }
    float4 x[LPR], mu[LPR], sg[LPR];
#pragma unroll
    for (int k = 0; k < LPR; ++k) {
        const int64_t q = chunk0 + k * kBlock + threadIdx.x;
        if (q < total_chunks) {
            x[k] = x4[q];
            mu[k] = m4[q];
            sg[k] = s4[kStdVec ? int64_t((k * kBlock + int(threadIdx.x)) % LPR) : q];  // chunk0 is a multiple of LPR
        }
    }
#pragma unroll
    for (int k = 0; k < LPR; ++k) {
        const int64_t q = chunk0 + k * kBlock + threadIdx.x;
        float lp = 0.0f, en = 0.0f;
        if (q < total_chunks) {
            const float xs[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
            const float ms[4] = {mu[k].x, mu[k].y, mu[k].z, mu[k].w};
            const float ss[4] = {sg[k].x, sg[k].y, sg[k].z, sg[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float diff = xs[j] - ms[j], ls = logf(ss[j]);
                // -((x - mu)^2) / (2 sigma^2) - log(sigma) - log(sqrt(2 pi))     distribution.py:207-209
                lp += -(diff * diff) / (2.0f * (ss[j] * ss[j])) - ls - log_sqrt_2pi();
                en += entropy_const() + ls;  // distribution.py:211-213
            }
        }
        lp_part[k * kBlock + threadIdx.x] = lp;
        en_part[k * kBlock + threadIdx.x] = en;
    }
    __syncthreads();

    double acc[kLossSums] = {0.0, 0.0, 0.0, 0.0, 0.0};
    {
        const int64_t row = row0 + threadIdx.x;
        float dlp = 0.0f;
        if (row < B) {
            float logp = 0.0f, entropy = 0.0f;
#pragma unroll
            for (int j = 0; j < LPR; ++j) {
                logp += lp_part[threadIdx.x * LPR + j];
                entropy += en_part[threadIdx.x * LPR + j];
            }
            float ratio, lr;
            dlp = row_terms(logp, entropy, pre_old_logp, pre_adv, p, acc[1], acc[2], acc[3], ratio, lr);
            if (logp_out) logp_out[row] = logp;
            if (entropy_out) entropy_out[row] = entropy;
            if (lr_out) lr_out[row] = lr;
            if (ratio_out) ratio_out[row] = ratio;
            if (D == 1)
                value_term(pre_cv, pre_ret, pre_ov, d_value ? d_value + row : nullptr, p, acc[0], acc[4]);
            else
                value_terms(ret, curr_value, old_value, d_value, row, D, p, acc[0], acc[4]);
        }
        dlp_row[threadIdx.x] = dlp;
    }
    __syncthreads();

    float4 *__restrict__ dm4 = reinterpret_cast<float4 *>(d_mean);
    float4 *__restrict__ ds4 = reinterpret_cast<float4 *>(d_std);
    // std-vector mode: a lane's k-th chunk belongs to column group (k * 256 + tid) % LPR; its d_std contribution is
    // accumulated into that group's slot (compile-time slots, selected by comparison — a few v_cndmask, no scratch),
    // so that afterwards plain wave-wide shuffle sums give the wave's [A] column sums: no [256, A] LDS tile, one barrier.
    float4 ds_acc[kStdVec ? LPR : 1];
#pragma unroll
    for (int g = 0; g < (kStdVec ? LPR : 1); ++g) ds_acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < LPR; ++k) {
        const int local = k * kBlock + threadIdx.x;
        const int64_t q = chunk0 + local;
        if (q < total_chunks) {
            const float dlp = dlp_row[local / LPR];
            const float xs[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
            const float ms[4] = {mu[k].x, mu[k].y, mu[k].z, mu[k].w};
            const float ss[4] = {sg[k].x, sg[k].y, sg[k].z, sg[k].w};
            float gm[4], gs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float diff = xs[j] - ms[j], var = ss[j] * ss[j];
                gm[j] = dlp * (diff / var);                                           // d logp / d mean
                gs[j] = dlp * ((diff * diff) / (var * ss[j]) - 1.0f / ss[j]) + p.g_ent / ss[j];  // + d entropy / d std
            }
            if (d_mean) dm4[q] = make_float4(gm[0], gm[1], gm[2], gm[3]);
            if (kStdVec) {
                const int group = local % LPR;
#pragma unroll
                for (int g = 0; g < LPR; ++g) {
                    const bool hit = group == g;
                    ds_acc[g].x += hit ? gs[0] : 0.f, ds_acc[g].y += hit ? gs[1] : 0.f;
                    ds_acc[g].z += hit ? gs[2] : 0.f, ds_acc[g].w += hit ? gs[3] : 0.f;
                }
            } else if (d_std) {
                ds4[q] = make_float4(gs[0], gs[1], gs[2], gs[3]);
            }
        }
    }
    if (kStdVec) {  // column sums of the block's [256, A] d_std tile, fixed order: lanes of a wave, then the 4 waves
        const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
#pragma unroll
        for (int g = 0; g < LPR; ++g) {
            const float4 total = make_float4(wave_sum(ds_acc[g].x), wave_sum(ds_acc[g].y), wave_sum(ds_acc[g].z),
                                             wave_sum(ds_acc[g].w));
            if (lane == 0) ds_wave[wave * LPR + g] = total;
        }
        __syncthreads();
        if (t < LPR && d_std_partials) {
            float4 total = ds_wave[t];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) {
                const float4 v = ds_wave[w * LPR + t];
                total.x += v.x, total.y += v.y, total.z += v.z, total.w += v.w;
            }
            float4 *slot = reinterpret_cast<float4 *>(d_std_partials) + int64_t(blockIdx.x) * LPR + t;
            if (ticket) {  // published as two 8-byte agent-scope stores (see load_published)
                unsigned long long *q = reinterpret_cast<unsigned long long *>(slot);
                __hip_atomic_store(q, (unsigned long long)__float_as_uint(total.x) | ((unsigned long long)__float_as_uint(total.y) << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(total.z) | ((unsigned long long)__float_as_uint(total.w) << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                *slot = total;
            }
        }
    }
    if (ticket) write_block_partials<true>(acc, partials);
    else write_block_partials<false>(acc, partials);
    if (ticket && last_block_done(ticket))  // uniform per block
        finalize_losses<true>(partials, gridDim.x, B, D, p, losses_out, (kStdVec && d_std) ? d_std_partials : nullptr,
                              LPR * 4, d_std);
}

// Any action width: one lane per row, scalar accesses.
__global__ __launch_bounds__(kBlock) void ppo_loss_rowwise_kernel(
    const float *__restrict__ advantage, const float *__restrict__ old_logp, const float *__restrict__ action,
    const float *__restrict__ mean, const float *__restrict__ std, const float *__restrict__ ret,
    const float *__restrict__ curr_value, const float *__restrict__ old_value, int64_t B, int A, int D, LossParams p,
    float *__restrict__ logp_out, float *__restrict__ entropy_out, float *__restrict__ lr_out,
    float *__restrict__ ratio_out, float *__restrict__ d_mean, float *__restrict__ d_std,
    float *__restrict__ d_value, double *__restrict__ partials, unsigned int *__restrict__ ticket,
    float *__restrict__ losses_out) {
    double acc[kLossSums] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (row < B) {
        float logp = 0.0f, entropy = 0.0f;
        for (int a = 0; a < A; ++a) {
            const int64_t i = row * A + a;
            const float diff = action[i] - mean[i], sg = std[i], ls = logf(sg);
            logp += -(diff * diff) / (2.0f * (sg * sg)) - ls - log_sqrt_2pi();
            entropy += entropy_const() + ls;
        }
        float ratio, lr;
        const float dlp = row_terms(logp, entropy, old_logp[row], advantage[row], p, acc[1], acc[2], acc[3], ratio, lr);
        if (logp_out) logp_out[row] = logp;
        if (entropy_out) entropy_out[row] = entropy;
        if (lr_out) lr_out[row] = lr;
        if (ratio_out) ratio_out[row] = ratio;
        for (int a = 0; a < A; ++a) {
            const int64_t i = row * A + a;
            const float diff = action[i] - mean[i], sg = std[i], var = sg * sg;
            if (d_mean) d_mean[i] = dlp * (diff / var);
            if (d_std) d_std[i] = dlp * ((diff * diff) / (var * sg) - 1.0f / sg) + p.g_ent / sg;
        }
        value_terms(ret, curr_value, old_value, d_value, row, D, p, acc[0], acc[4]);
    }
    if (ticket) write_block_partials<true>(acc, partials);
    else write_block_partials<false>(acc, partials);
    if (ticket && last_block_done(ticket))
        finalize_losses<true>(partials, gridDim.x, B, D, p, losses_out, nullptr, A, nullptr);
}

// One-hot categorical policies (discrete action spaces, cusrl/nn/module/distribution.py:332-366 on top of
// torch.distributions.OneHotCategorical): log-prob = log-softmax(logits)[argmax(action)], entropy = -sum p log p,
// d logp / d logits = onehot - p, d entropy / d logits_j = -p_j (log p_j + entropy).  One lane per row; the row is
// walked three times (max, normaliser + the taken action, gradients) — the second and third walk hit L1.
__global__ __launch_bounds__(kBlock) void ppo_loss_categorical_kernel(
    const float *__restrict__ advantage, const float *__restrict__ old_logp, const float *__restrict__ action,
    const float *__restrict__ logits, const float *__restrict__ ret, const float *__restrict__ curr_value,
    const float *__restrict__ old_value, int64_t B, int A, int D, LossParams p, float *__restrict__ logp_out,
    float *__restrict__ entropy_out, float *__restrict__ lr_out, float *__restrict__ ratio_out,
    float *__restrict__ d_logits, float *__restrict__ d_value, double *__restrict__ partials) {
    double acc[kLossSums] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (row < B) {
        const float *z = logits + row * A, *a = action + row * A;
        float zmax = z[0];
        for (int j = 1; j < A; ++j) zmax = fmaxf(zmax, z[j]);
        float sum = 0.0f, best = a[0];
        int taken = 0;  // value.max(-1)[1]: first index of the largest entry of the (one-hot) action
        for (int j = 0; j < A; ++j) {
            sum += expf(z[j] - zmax);
            if (a[j] > best) best = a[j], taken = j;
        }
        const float log_norm = zmax + logf(sum);  // logsumexp
        // A masked action (logit = -inf) has p = 0 and log p = -inf: torch.distributions.Categorical.entropy clamps
        // the log-prob to finfo.min before the product, so the term is 0 * (-3.4e38) = -0 instead of 0 * (-inf) = NaN —
        // and so are the row's entropy, the loss and every d_logits of the row.
        float entropy = 0.0f;
        for (int j = 0; j < A; ++j) {
            const float lp = z[j] - log_norm;
            entropy -= expf(lp) * fmaxf(lp, -FLT_MAX);  // -(probs * clamp(logits, min=finfo.min)).sum(-1)
        }
        const float logp = z[taken] - log_norm;
        float ratio, lr;
        const float dlp = row_terms(logp, entropy, old_logp[row], advantage[row], p, acc[1], acc[2], acc[3], ratio, lr);
        if (logp_out) logp_out[row] = logp;
        if (entropy_out) entropy_out[row] = entropy;
        if (lr_out) lr_out[row] = lr;
        if (ratio_out) ratio_out[row] = ratio;
        if (d_logits) {
            for (int j = 0; j < A; ++j) {
                const float lp = z[j] - log_norm, pj = expf(lp);
                d_logits[row * A + j] =
                    dlp * ((j == taken ? 1.0f : 0.0f) - pj) - p.g_ent * (pj * (fmaxf(lp, -FLT_MAX) + entropy));
            }
        }
        value_terms(ret, curr_value, old_value, d_value, row, D, p, acc[0], acc[4]);
    }
    write_block_partials<false>(acc, partials);
}

// Column sums of `rows` partial rows [rows][A] (A <= 32) by one block, fixed order: lane (a, g) walks rows g, g + 8, ...
// four loads at a time, the 8 row groups are combined through LDS.  out[a] for a < A.
constexpr int kStdSliceRows = 128;  // partial rows one block of the staged reduction takes

template <bool kPublished>
__device__ __forceinline__ void reduce_std_rows(const float *in, int64_t rows, int A, float *__restrict__ out) {
    __shared__ float part[kBlock];
    const int a = threadIdx.x & 31, g = threadIdx.x >> 5;
    auto at = [&](int64_t i) { return kPublished ? load_published(in + i) : in[i]; };
    float total = 0.f;
    if (a < A) {
        int64_t r = g;
        for (; r + 24 < rows; r += 32) {
            const float v0 = at(r * A + a), v1 = at((r + 8) * A + a), v2 = at((r + 16) * A + a), v3 = at((r + 24) * A + a);
            total += (v0 + v1) + (v2 + v3);
        }
        for (; r < rows; r += 8) total += at(r * A + a);
    }
    part[threadIdx.x] = total;
    __syncthreads();
    if (g == 0 && a < A) {
        float sum = part[a];
#pragma unroll
        for (int k = 1; k < kBlock / 32; ++k) sum += part[k * 32 + a];
        out[a] = sum;
    }
    __syncthreads();
}

// Staged form for many partial rows: block b reduces rows [b * 128, b * 128 + 128) into stage[b][A].
__global__ __launch_bounds__(kBlock) void std_rows_stage_kernel(const float *__restrict__ in, int64_t rows, int A,
                                                                float *__restrict__ stage) {
    const int64_t first = int64_t(blockIdx.x) * kStdSliceRows;
    reduce_std_rows<false>(in + first * A, min(int64_t(kStdSliceRows), rows - first), A, stage + int64_t(blockIdx.x) * A);
}

__global__ __launch_bounds__(kBlock) void std_rows_final_kernel(const float *__restrict__ stage, int64_t rows, int A,
                                                                float *__restrict__ out) {
    reduce_std_rows<false>(stage, rows, A, out);
}

// Many blocks (> 256, i.e. minibatches beyond 65 536 rows): one block cannot walk all partial rows at memory latency,
// so slices of 256 rows are first reduced to one row each by as many blocks, and the finalize below reads those.
__global__ __launch_bounds__(kBlock) void loss_partials_stage_kernel(const double *__restrict__ partials, int64_t P,
                                                                     double *__restrict__ stage) {
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    double acc[kLossSums];
#pragma unroll
    for (int k = 0; k < kLossSums; ++k) acc[k] = row < P ? partials[row * kLossSums + k] : 0.0;
    write_block_partials(acc, stage);
}

template <bool kPublished>
__device__ __forceinline__ void finalize_losses(const double *partials, int64_t P, int64_t B, int D,
                                                const LossParams &p, float *__restrict__ losses_out,
                                                const float *d_std_partials, int A, float *__restrict__ d_std_vector) {
    __shared__ double scratch[kWavesPerBlock];
    if (d_std_partials) reduce_std_rows<kPublished>(d_std_partials, P, A, d_std_vector);  // std-vector mode, few blocks
    double sums[kLossSums];
#pragma unroll
    for (int k = 0; k < kLossSums; ++k) {
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < P; i += kBlock)
            s += kPublished ? load_published(partials + i * kLossSums + k) : partials[i * kLossSums + k];
        sums[k] = block_sum(s, scratch);
    }
    if (threadIdx.x == 0) {
        losses_out[0] = float(sums[0] / double(B * D)) * p.w_val;   // mean * weight              value.py:137
        losses_out[1] = -float(sums[1] / double(B)) * p.w_sur;      // -mean(min(...)) * weight   ppo.py:13-18,55
        losses_out[2] = -float(sums[2] / double(B)) * p.w_ent;      // -mean(entropy) * weight    ppo.py:83-84
        losses_out[3] = float(sums[3] / double(B));                 // mean |logp ratio|          common.py:47 metric
        losses_out[4] = float(sums[2] / double(B));                 // mean entropy               common.py:48 metric
        losses_out[5] = float(sums[4] / double(B));                 // mean value.sum(-1)         value.py:141 metric
        // sum(objectives.values()) in the hooks' insertion order            actor_critic.py:309
        losses_out[6] = (losses_out[0] + losses_out[1]) + losses_out[2];
    }
}

__global__ __launch_bounds__(kBlock) void ppo_loss_finalize_kernel(const double *__restrict__ partials, int64_t P,
                                                                   int64_t B, int D, LossParams p,
                                                                   float *__restrict__ losses_out,
                                                                   const float *__restrict__ d_std_partials, int A,
                                                                   float *__restrict__ d_std_vector) {
    finalize_losses<false>(partials, P, B, D, p, losses_out, d_std_partials, A, d_std_vector);
}

}  // namespace cusrl

using namespace cusrl;

static int64_t loss_blocks(int64_t B) { return B <= 0 ? 0 : ceil_div(B, kRowsPerBlock); }

// rows of the fp64 workspace: one per block, plus one per 256-block slice when the reduction is staged
extern "C" int64_t cusrl_ppo_loss_num_partials(int64_t B) {
    const int64_t blocks = loss_blocks(B);
    return blocks + (blocks > kBlock ? ceil_div(blocks, kBlock) : 0);
}

extern "C" int64_t cusrl_ppo_loss_std_partial_rows(int64_t B) {
    const int64_t blocks = loss_blocks(B);
    return blocks + (blocks > kStdSliceRows ? ceil_div(blocks, kStdSliceRows) : 0);
}

extern "C" int cusrl_ppo_loss_categorical_fwd_bwd(const float *advantage, const float *old_logp, const float *action,
                                                  const float *logits, const float *ret, const float *curr_value,
                                                  const float *old_value, int64_t B, int64_t A, int64_t D, double clip,
                                                  double value_clip, double w_sur, double w_val, double w_ent,
                                                  float *losses_out, float *logp_out, float *entropy_out,
                                                  float *logp_ratio_out, float *ratio_out, float *d_logits,
                                                  float *d_value, double *partials, void *stream) {
    if (B <= 0 || A <= 0 || D <= 0) return CUSRL_E_INVALID;
    if (!advantage || !old_logp || !action || !logits || !ret || !curr_value || !losses_out || !partials)
        return CUSRL_E_INVALID;
    if (value_clip >= 0.0 && !old_value) return CUSRL_E_INVALID;
    if (A > INT32_MAX || D > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    LossParams p;
    p.lo = float(1.0 - clip);
    p.hi = float(1.0 + clip);
    p.value_clip = value_clip < 0.0 ? -1.0f : float(value_clip);
    p.g_sur = float(-w_sur / double(B));
    p.g_ent = float(-w_ent / double(B));
    p.g_val = float(w_val / double(B * D));
    p.w_sur = float(w_sur);
    p.w_val = float(w_val);
    p.w_ent = float(w_ent);
    hipStream_t s = as_stream(stream);
    const int64_t blocks = loss_blocks(B);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(ppo_loss_categorical_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, s, advantage, old_logp,
                       action, logits, ret, curr_value, old_value, B, int(A), int(D), p, logp_out, entropy_out,
                       logp_ratio_out, ratio_out, d_logits, d_value, partials);
    if (int rc = launch_status()) return rc;
    const double *loss_rows = partials;
    int64_t num_loss_rows = blocks;
    if (blocks > kBlock) {
        num_loss_rows = ceil_div(blocks, kBlock);
        double *stage = partials + blocks * kLossSums;
        hipLaunchKernelGGL(loss_partials_stage_kernel, dim3(uint32_t(num_loss_rows)), dim3(kBlock), 0, s, partials, blocks,
                           stage);
        if (int rc = launch_status()) return rc;
        loss_rows = stage;
    }
    hipLaunchKernelGGL(ppo_loss_finalize_kernel, dim3(1), dim3(kBlock), 0, s, loss_rows, num_loss_rows, B, int(D), p,
                       losses_out, nullptr, int(A), nullptr);
    return launch_status();
}

#define CUSRL_LAUNCH_CHUNKED(LPR)                                                                                      \
    if (std_vector)                                                                                                    \
        hipLaunchKernelGGL((ppo_loss_chunked_kernel<LPR, true>), dim3(uint32_t(blocks)), dim3(kBlock), 0, s, advantage, \
                           old_logp, action, mean, std, ret, curr_value, old_value, B, int(D), p, logp_out,            \
                           entropy_out, logp_ratio_out, ratio_out, d_mean, d_std, d_value, partials, d_std_partials,   \
                           in_kernel, losses_out);                                                                     \
    else                                                                                                               \
        hipLaunchKernelGGL((ppo_loss_chunked_kernel<LPR, false>), dim3(uint32_t(blocks)), dim3(kBlock), 0, s,           \
                           advantage, old_logp, action, mean, std, ret, curr_value, old_value, B, int(D), p, logp_out, \
                           entropy_out, logp_ratio_out, ratio_out, d_mean, d_std, d_value, partials, d_std_partials,   \
                           in_kernel, losses_out)

extern "C" int cusrl_ppo_loss_fwd_bwd(const float *advantage, const float *old_logp, const float *action,
                                      const float *mean, const float *std, const float *ret, const float *curr_value,
                                      const float *old_value, int64_t B, int64_t A, int64_t D, double clip,
                                      double value_clip, double w_sur, double w_val, double w_ent, float *losses_out,
                                      float *logp_out, float *entropy_out, float *logp_ratio_out, float *ratio_out,
                                      float *d_mean, float *d_std, float *d_value, double *partials,
                                      int64_t std_rows, float *d_std_partials, uint32_t *ticket, void *stream) {
    if (B <= 0 || A <= 0 || D <= 0) return CUSRL_E_INVALID;
    if (std_rows != B && std_rows != 1) return CUSRL_E_INVALID;
    const bool std_vector = std_rows == 1 && B != 1;
    if (std_vector && d_std && !d_std_partials) return CUSRL_E_INVALID;
    if (!advantage || !old_logp || !action || !mean || !std || !ret || !curr_value || !losses_out || !partials)
        return CUSRL_E_INVALID;
    if (value_clip >= 0.0 && !old_value) return CUSRL_E_INVALID;
    if (A > INT32_MAX || D > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    LossParams p;
    p.lo = float(1.0 - clip);
    p.hi = float(1.0 + clip);
    p.value_clip = value_clip < 0.0 ? -1.0f : float(value_clip);
    p.g_sur = float(-w_sur / double(B));
    p.g_ent = float(-w_ent / double(B));
    p.g_val = float(w_val / double(B * D));
    p.w_sur = float(w_sur);
    p.w_val = float(w_val);
    p.w_ent = float(w_ent);
    hipStream_t s = as_stream(stream);
    const int64_t blocks = loss_blocks(B);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const bool chunked = A % 4 == 0 && A / 4 <= 8 && aligned(action, 16) && aligned(mean, 16) && aligned(std, 16) &&
                         (!d_mean || aligned(d_mean, 16)) && (!d_std || aligned(d_std, 16));
    if (std_vector && !chunked) return CUSRL_E_UNSUPPORTED;  // the row-vector form exists for the 16-byte-chunk layout
    // few enough blocks for ONE block to reduce every partial row at memory latency: the last block to finish does it
    // inside the launch (ticket); larger minibatches keep the staged reduction launches below
    const bool fused_finalize = ticket && blocks <= kBlock && (!(std_vector && d_std) || blocks <= kStdSliceRows);
    unsigned int *in_kernel = fused_finalize ? ticket : nullptr;
    if (chunked) {
        switch (A / 4) {
            case 1: CUSRL_LAUNCH_CHUNKED(1); break;
            case 2: CUSRL_LAUNCH_CHUNKED(2); break;
            case 3: CUSRL_LAUNCH_CHUNKED(3); break;
            case 4: CUSRL_LAUNCH_CHUNKED(4); break;
            case 5: CUSRL_LAUNCH_CHUNKED(5); break;
            case 6: CUSRL_LAUNCH_CHUNKED(6); break;
            case 7: CUSRL_LAUNCH_CHUNKED(7); break;
            default: CUSRL_LAUNCH_CHUNKED(8); break;
        }
    } else {
        hipLaunchKernelGGL(ppo_loss_rowwise_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, s, advantage, old_logp,
                           action, mean, std, ret, curr_value, old_value, B, int(A), int(D), p, logp_out, entropy_out,
                           logp_ratio_out, ratio_out, d_mean, d_std, d_value, partials, in_kernel, losses_out);
    }
    if (int rc = launch_status()) return rc;
    if (fused_finalize) return 0;
    // gradient of the std vector = column sums of the per-block sums: inside the finalize launch for up to 128 blocks
    // (a 32 768-row minibatch), staged over 128-row slices beyond that
    const bool reduce_std = std_vector && d_std;
    const bool staged = reduce_std && blocks > kStdSliceRows;
    const double *loss_rows = partials;
    int64_t num_loss_rows = blocks;
    if (blocks > kBlock) {
        num_loss_rows = ceil_div(blocks, kBlock);
        double *stage = partials + blocks * kLossSums;
        hipLaunchKernelGGL(loss_partials_stage_kernel, dim3(uint32_t(num_loss_rows)), dim3(kBlock), 0, s, partials, blocks,
                           stage);
        if (int rc = launch_status()) return rc;
        loss_rows = stage;
    }
    hipLaunchKernelGGL(ppo_loss_finalize_kernel, dim3(1), dim3(kBlock), 0, s, loss_rows, num_loss_rows, B, int(D), p,
                       losses_out, (reduce_std && !staged) ? d_std_partials : nullptr, int(A), d_std);
    if (int rc = launch_status()) return rc;
    if (staged) {
        const int64_t slices = ceil_div(blocks, kStdSliceRows);
        float *stage = d_std_partials + blocks * A;  // the workspace holds the slices' rows behind the blocks' rows
        hipLaunchKernelGGL(std_rows_stage_kernel, dim3(uint32_t(slices)), dim3(kBlock), 0, s, d_std_partials, blocks,
                           int(A), stage);
        if (int rc = launch_status()) return rc;
        hipLaunchKernelGGL(std_rows_final_kernel, dim3(1), dim3(kBlock), 0, s, stage, slices, int(A), d_std);
    }
    return launch_status();
}
