// Post-update policy statistics (hook/on_policy/stats.py:28-40) for Gaussian policies in one pass:
//   kl_divergence                 = mean_b  sum_a 0.5 (r + ((mu_p - mu_q) / s_q)^2 - 1 - log r),  r = (s_p / s_q)^2
//   importance_weighted_advantage = mean    adv * exp(logp_q(action) - old_logp)
//   action_std                    = mean_{b,a} s_q
// p = the behaviour policy stored in the buffer, q = the updated actor's output on the same observations.
// As torch ops this is ~25 elementwise / reduction launches over [98304, 12] issued eagerly once per update; here
// every row is read once (5 x 4A + 4 + 4D bytes), block partials in fp64, fixed summation order.
#include "common.hpp"

namespace cusrl {

constexpr int kStatSums = 3;

__global__ __launch_bounds__(kBlock) void policy_stats_kernel(const float *__restrict__ old_mean,
                                                              const float *__restrict__ old_std,
                                                              const float *__restrict__ new_mean,
                                                              const float *__restrict__ new_std,
                                                              const float *__restrict__ action,
                                                              const float *__restrict__ old_logp,
                                                              const float *__restrict__ advantage, int64_t B, int A,
                                                              int D, double *__restrict__ partials) {
    __shared__ double scratch[kWavesPerBlock][kStatSums];
    double acc[kStatSums] = {0.0, 0.0, 0.0};
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (row < B) {
        float kl = 0.f, logp = 0.f, std_sum = 0.f;
        const int64_t base = row * A;
        for (int a = 0; a < A; ++a) {
            const float mp = old_mean[base + a], sp = old_std[base + a];
            const float mq = new_mean[base + a], sq = new_std[base + a], x = action[base + a];
            const float ratio = sp / sq, var_ratio = ratio * ratio;            // torch.distributions.kl: _kl_normal_normal
            const float z = (mp - mq) / sq, t1 = z * z;
            kl += 0.5f * (var_ratio + t1 - 1.0f - logf(var_ratio));
            const float diff = x - mq;                                          // Normal.log_prob
            logp += -(diff * diff) / (2.0f * (sq * sq)) - logf(sq) - 0.918938533204672741780329736406f;
            std_sum += sq;
        }
        const float weight = expf(logp - old_logp[row]);
        float iw = 0.f;
        for (int d = 0; d < D; ++d) iw += advantage[row * D + d] * weight;
        acc[0] = double(kl), acc[1] = double(iw), acc[2] = double(std_sum);
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < kStatSums; ++k) {
        const double total = wave_sum(acc[k]);
        if (lane == 0) scratch[wave][k] = total;
    }
    __syncthreads();
    if (threadIdx.x < kStatSums) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) total += scratch[w][threadIdx.x];
        partials[int64_t(blockIdx.x) * kStatSums + threadIdx.x] = total;
    }
}

// One-hot categorical policies (torch.distributions.kl._kl_categorical_categorical, OneHotCategorical.log_prob):
//   kl = sum_j p_j (log p_j - log q_j)  [inf where q_j = 0 < p_j, 0 where p_j = 0],  logp_q(action) = log q_taken.
// The third sum stays zero (no `action_std` for this family).
__global__ __launch_bounds__(kBlock) void categorical_policy_stats_kernel(const float *__restrict__ old_logits,
                                                                          const float *__restrict__ new_logits,
                                                                          const float *__restrict__ action,
                                                                          const float *__restrict__ old_logp,
                                                                          const float *__restrict__ advantage,
                                                                          int64_t B, int A, int D,
                                                                          double *__restrict__ partials) {
    __shared__ double scratch[kWavesPerBlock][kStatSums];
    double acc[kStatSums] = {0.0, 0.0, 0.0};
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (row < B) {
        const float *zp = old_logits + row * A, *zq = new_logits + row * A, *a = action + row * A;
        float mp = zp[0], mq = zq[0];
        for (int j = 1; j < A; ++j) mp = fmaxf(mp, zp[j]), mq = fmaxf(mq, zq[j]);
        float sp = 0.f, sq = 0.f, best = a[0];
        int taken = 0;
        for (int j = 0; j < A; ++j) {
            sp += expf(zp[j] - mp), sq += expf(zq[j] - mq);
            if (a[j] > best) best = a[j], taken = j;
        }
        const float np = mp + logf(sp), nq = mq + logf(sq);
        float kl = 0.f;
        for (int j = 0; j < A; ++j) {
            const float lp = zp[j] - np, lq = zq[j] - nq, pj = expf(lp);
            float t = pj * (lp - lq);
            if (expf(lq) == 0.0f) t = INFINITY;  // t[(q.probs == 0)] = inf
            if (pj == 0.0f) t = 0.0f;            // t[(p.probs == 0)] = 0
            kl += t;
        }
        const float weight = expf((zq[taken] - nq) - old_logp[row]);
        float iw = 0.f;
        for (int d = 0; d < D; ++d) iw += advantage[row * D + d] * weight;
        acc[0] = double(kl), acc[1] = double(iw);
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < kStatSums; ++k) {
        const double total = wave_sum(acc[k]);
        if (lane == 0) scratch[wave][k] = total;
    }
    __syncthreads();
    if (threadIdx.x < kStatSums) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) total += scratch[w][threadIdx.x];
        partials[int64_t(blockIdx.x) * kStatSums + threadIdx.x] = total;
    }
}

__global__ __launch_bounds__(kBlock) void policy_stats_finalize_kernel(const double *__restrict__ partials, int64_t P,
                                                                       int64_t B, int A, int D,
                                                                       float *__restrict__ out) {
    __shared__ double scratch[kWavesPerBlock];
    double sums[kStatSums];
#pragma unroll
    for (int k = 0; k < kStatSums; ++k) {
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < P; i += kBlock) s += partials[i * kStatSums + k];
        sums[k] = block_sum(s, scratch);
    }
    if (threadIdx.x == 0) {
        out[0] = float(sums[0] / double(B));      // kl_divergence [B, 1] -> mean
        out[1] = float(sums[1] / double(B * D));  // importance-weighted advantage [B, D] -> mean
        out[2] = float(sums[2] / double(B * A));  // action std [B, A] -> mean
    }
}

}  // namespace cusrl

extern "C" int64_t cusrl_policy_stats_num_partials(int64_t B) { return B <= 0 ? 0 : cusrl::ceil_div(B, cusrl::kBlock); }

extern "C" int cusrl_policy_stats(const float *old_mean, const float *old_std, const float *new_mean,
                                  const float *new_std, const float *action, const float *old_logp,
                                  const float *advantage, int64_t B, int64_t A, int64_t D, double *partials,
                                  float *out, void *stream) {
    using namespace cusrl;
    if (!old_mean || !old_std || !new_mean || !new_std || !action || !old_logp || !advantage || !partials || !out)
        return CUSRL_E_INVALID;
    if (B <= 0 || A <= 0 || D <= 0 || A > INT32_MAX || D > INT32_MAX) return CUSRL_E_INVALID;
    const int64_t blocks = ceil_div(B, kBlock);
    hipStream_t s = as_stream(stream);
    policy_stats_kernel<<<uint32_t(blocks), kBlock, 0, s>>>(old_mean, old_std, new_mean, new_std, action, old_logp,
                                                             advantage, B, int(A), int(D), partials);
    if (int rc = launch_status()) return rc;
    policy_stats_finalize_kernel<<<1, kBlock, 0, s>>>(partials, blocks, B, int(A), int(D), out);
    return launch_status();
}

extern "C" int cusrl_categorical_policy_stats(const float *old_logits, const float *new_logits, const float *action,
                                              const float *old_logp, const float *advantage, int64_t B, int64_t A,
                                              int64_t D, double *partials, float *out, void *stream) {
    using namespace cusrl;
    if (!old_logits || !new_logits || !action || !old_logp || !advantage || !partials || !out) return CUSRL_E_INVALID;
    if (B <= 0 || A <= 0 || D <= 0 || A > INT32_MAX || D > INT32_MAX) return CUSRL_E_INVALID;
    const int64_t blocks = ceil_div(B, kBlock);
    hipStream_t s = as_stream(stream);
    categorical_policy_stats_kernel<<<uint32_t(blocks), kBlock, 0, s>>>(old_logits, new_logits, action, old_logp,
                                                                         advantage, B, int(A), int(D), partials);
    if (int rc = launch_status()) return rc;
    policy_stats_finalize_kernel<<<1, kBlock, 0, s>>>(partials, blocks, B, int(A), int(D), out);
    return launch_status();
}
