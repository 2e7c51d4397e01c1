// Differentiable policy terms of one minibatch (OnPolicyPreparation.objective, cusrl/hook/on_policy/common.py:29-43):
//   logp       = sum_a log N(action | mean, std)                      (cusrl/nn/module/distribution.py:207-209)
//   entropy    = sum_a (0.5 + 0.5 log(2 pi) + log std)                (distribution.py:211-213)
//   logp_ratio = logp - old_logp,   prob_ratio = exp(logp_ratio)      (common.py:33,41)
// and the one-hot categorical twin (distribution.py:354-362).  The stock PPO composition never launches these: its one
// fused objective kernel (ppo_loss.hip) evaluates the same terms and their gradients in a single pass.  They exist for
// compositions in which a user hook defines `objective` as well and may read or differentiate the four tensors above:
// forward = ONE launch, backward = ONE launch that folds the four incoming gradients
//   G = g_logp + g_logp_ratio + g_prob_ratio * prob_ratio
//   d mean = G (x - mu) / sigma^2,   d std = G ((x - mu)^2 / sigma^3 - 1 / sigma) + g_entropy / sigma
// (a std handed over as its [A] vector gets its column sums: fixed-order block partials + a one-block finalize).
// One lane per row (rows are 4A <= a few hundred bytes: every fetched line is consumed by neighbouring lanes).
#include "common.hpp"

namespace cusrl {

constexpr int kTermsMaxVectorA = 64;  // widest action vector whose std may be a shared [A] vector

__device__ __forceinline__ float terms_log_sqrt_2pi() { return 0.918938533204672741780329736406f; }
__device__ __forceinline__ float terms_entropy_const() { return 1.418938533204672741780329736406f; }  // 0.5 + 0.5 log(2 pi)

__global__ __launch_bounds__(kBlock) void policy_terms_fwd_kernel(const float *__restrict__ mean,
                                                                  const float *__restrict__ std, int std_is_vector,
                                                                  const float *__restrict__ action,
                                                                  const float *__restrict__ old_logp, int64_t B, int A,
                                                                  float *__restrict__ logp_out,
                                                                  float *__restrict__ entropy_out,
                                                                  float *__restrict__ lr_out,
                                                                  float *__restrict__ ratio_out) {
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (row >= B) return;
    const float *mu = mean + row * A, *x = action + row * A, *sg = std_is_vector ? std : std + row * A;
    // The per-element terms are fp32 like the reference's; their ROW SUMS are carried in fp64: ratio = exp(logp - old_logp)
    // turns every bit of absolute rounding noise of a |logp| ~ A sum into relative noise of the whole row's gradients
    // (DESIGN.md section 4), and 2A double additions per row cost nothing in a launch that waits for memory.
    double logp = 0.0, entropy = 0.0;
    for (int a = 0; a < A; ++a) {
        const float diff = x[a] - mu[a], s = sg[a], ls = logf(s);
        logp += double(-(diff * diff) / (2.0f * (s * s)) - ls - terms_log_sqrt_2pi());  // Normal.log_prob
        entropy += double(terms_entropy_const() + ls);                                  // Normal.entropy
    }
    const float lr = float(logp - double(old_logp[row]));
    logp_out[row] = float(logp);
    entropy_out[row] = float(entropy);
    lr_out[row] = lr;
    ratio_out[row] = expf(lr);
}

// g_* may be null (that output received no gradient).  d_std: [B, A], or — std vector — partial column sums per block
// in d_std_partials [blocks, A] (reduced by policy_terms_std_finalize_kernel).
__global__ __launch_bounds__(kBlock) void policy_terms_bwd_kernel(const float *__restrict__ mean,
                                                                  const float *__restrict__ std, int std_is_vector,
                                                                  const float *__restrict__ action,
                                                                  const float *__restrict__ ratio,
                                                                  const float *__restrict__ g_logp,
                                                                  const float *__restrict__ g_entropy,
                                                                  const float *__restrict__ g_lr,
                                                                  const float *__restrict__ g_ratio, int64_t B, int A,
                                                                  float *__restrict__ d_mean, float *__restrict__ d_std,
                                                                  float *__restrict__ d_std_partials) {
    __shared__ double wave_cols[kWavesPerBlock][kTermsMaxVectorA];
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const bool live = row < B;
    float G = 0.0f, ge = 0.0f;
    if (live) {
        if (g_logp) G += g_logp[row];
        if (g_lr) G += g_lr[row];
        if (g_ratio) G += g_ratio[row] * ratio[row];
        if (g_entropy) ge = g_entropy[row];
    }
    const int64_t safe = live ? row : 0;
    const float *mu = mean + safe * A, *x = action + safe * A, *sg = std_is_vector ? std : std + safe * A;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int a = 0; a < A; ++a) {
        const float s = sg[a], inv = 1.0f / s, z = (x[a] - mu[a]) * inv;
        const float gm = G * (z * inv);
        const float gs = live ? inv * (G * (z * z - 1.0f) + ge) : 0.0f;
        if (live) d_mean[row * A + a] = gm;
        if (!std_is_vector) {
            if (live) d_std[row * A + a] = gs;
        } else {
            const double total = wave_sum(double(gs));  // fixed order: lanes, then waves, then blocks
            if (lane == 0) wave_cols[wave][a] = total;
        }
    }
    if (std_is_vector) {
        __syncthreads();
        if (threadIdx.x < A) {
            double total = 0.0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) total += wave_cols[w][threadIdx.x];
            d_std_partials[int64_t(blockIdx.x) * A + threadIdx.x] = float(total);
        }
    }
}

__global__ __launch_bounds__(kBlock) void policy_terms_std_finalize_kernel(const float *__restrict__ partials,
                                                                           int64_t blocks, int A,
                                                                           float *__restrict__ d_std) {
    // one wave per column group would be faster; A <= 64 columns x a few hundred block rows is microseconds either way
    __shared__ double scratch[kWavesPerBlock];
    for (int a = 0; a < A; ++a) {
        double s = 0.0;
        for (int64_t b = threadIdx.x; b < blocks; b += kBlock) s += double(partials[b * A + a]);
        s = block_sum(s, scratch);
        if (threadIdx.x == 0) d_std[a] = float(s);
    }
}

// ---- one-hot categorical policy: logits [B, A], action one-hot [B, A]
//   logp = z[taken] - logsumexp(z),  entropy = -sum_j p_j log p_j  (log p clamped to the smallest finite float like
//   torch.distributions.Categorical.entropy, so a masked action — logit -inf — contributes exactly 0)
__global__ __launch_bounds__(kBlock) void categorical_terms_fwd_kernel(const float *__restrict__ logits,
                                                                       const float *__restrict__ action,
                                                                       const float *__restrict__ old_logp, int64_t B, int A,
                                                                       float *__restrict__ logp_out,
                                                                       float *__restrict__ entropy_out,
                                                                       float *__restrict__ lr_out,
                                                                       float *__restrict__ ratio_out) {
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (row >= B) return;
    const float *z = logits + row * A, *x = action + row * A;
    float m = z[0], best = x[0];
    int taken = 0;
    for (int j = 1; j < A; ++j) {
        m = fmaxf(m, z[j]);
        if (x[j] > best) best = x[j], taken = j;
    }
    float sum = 0.0f;
    for (int j = 0; j < A; ++j) sum += expf(z[j] - m);
    const float norm = m + logf(sum);
    float entropy = 0.0f;
    for (int j = 0; j < A; ++j) {
        const float lp = fmaxf(z[j] - norm, -3.402823466e+38f);
        entropy -= expf(lp) * lp;
    }
    const float logp = z[taken] - norm, lr = logp - old_logp[row];
    logp_out[row] = logp;
    entropy_out[row] = entropy;
    lr_out[row] = lr;
    ratio_out[row] = expf(lr);
}

__global__ __launch_bounds__(kBlock) void categorical_terms_bwd_kernel(const float *__restrict__ logits,
                                                                       const float *__restrict__ action,
                                                                       const float *__restrict__ ratio,
                                                                       const float *__restrict__ g_logp,
                                                                       const float *__restrict__ g_entropy,
                                                                       const float *__restrict__ g_lr,
                                                                       const float *__restrict__ g_ratio, int64_t B, int A,
                                                                       float *__restrict__ d_logits) {
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (row >= B) return;
    float G = 0.0f, ge = 0.0f;
    if (g_logp) G += g_logp[row];
    if (g_lr) G += g_lr[row];
    if (g_ratio) G += g_ratio[row] * ratio[row];
    if (g_entropy) ge = g_entropy[row];
    const float *z = logits + row * A, *x = action + row * A;
    float m = z[0], best = x[0];
    int taken = 0;
    for (int j = 1; j < A; ++j) {
        m = fmaxf(m, z[j]);
        if (x[j] > best) best = x[j], taken = j;
    }
    float sum = 0.0f;
    for (int j = 0; j < A; ++j) sum += expf(z[j] - m);
    const float norm = m + logf(sum);
    float entropy = 0.0f;
    for (int j = 0; j < A; ++j) {
        const float lp = fmaxf(z[j] - norm, -3.402823466e+38f);
        entropy -= expf(lp) * lp;
    }
    for (int j = 0; j < A; ++j) {
        const float lp = fmaxf(z[j] - norm, -3.402823466e+38f), p = expf(lp);
        // d logp / d z_j = [j == taken] - p_j;   d entropy / d z_j = -p_j (log p_j + entropy)
        d_logits[row * A + j] = G * ((j == taken ? 1.0f : 0.0f) - p) - ge * p * (lp + entropy);
    }
}

}  // namespace cusrl

using namespace cusrl;

extern "C" int64_t cusrl_policy_terms_std_partial_rows(int64_t B) { return B <= 0 ? 0 : ceil_div(B, kBlock); }

extern "C" int cusrl_policy_terms_fwd(const float *mean, const float *std, int64_t std_rows, const float *action,
                                      const float *old_logp, int64_t B, int64_t A, float *logp_out, float *entropy_out,
                                      float *logp_ratio_out, float *ratio_out, void *stream) {
    if (B < 0 || A <= 0 || A > INT32_MAX) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (std_rows != B && std_rows != 1) return CUSRL_E_INVALID;
    if (!mean || !std || !action || !old_logp || !logp_out || !entropy_out || !logp_ratio_out || !ratio_out)
        return CUSRL_E_INVALID;
    const int64_t blocks = ceil_div(B, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(policy_terms_fwd_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), mean, std,
                       int(std_rows == 1 && B != 1), action, old_logp, B, int(A), logp_out, entropy_out, logp_ratio_out,
                       ratio_out);
    return launch_status();
}

extern "C" int cusrl_policy_terms_bwd(const float *mean, const float *std, int64_t std_rows, const float *action,
                                      const float *ratio, const float *g_logp, const float *g_entropy,
                                      const float *g_logp_ratio, const float *g_ratio, int64_t B, int64_t A, float *d_mean,
                                      float *d_std, float *d_std_partials, void *stream) {
    if (B < 0 || A <= 0 || A > INT32_MAX) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (std_rows != B && std_rows != 1) return CUSRL_E_INVALID;
    if (!mean || !std || !action || !d_mean || !d_std || (g_ratio && !ratio)) return CUSRL_E_INVALID;
    const bool vector = std_rows == 1 && B != 1;
    if (vector && (!d_std_partials || A > kTermsMaxVectorA)) return vector && A > kTermsMaxVectorA ? CUSRL_E_UNSUPPORTED : CUSRL_E_INVALID;
    const int64_t blocks = ceil_div(B, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(policy_terms_bwd_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, s, mean, std, int(vector), action,
                       ratio, g_logp, g_entropy, g_logp_ratio, g_ratio, B, int(A), d_mean, d_std, d_std_partials);
    if (int rc = launch_status()) return rc;
    if (vector) {
        hipLaunchKernelGGL(policy_terms_std_finalize_kernel, dim3(1), dim3(kBlock), 0, s, d_std_partials, blocks, int(A),
                           d_std);
        return launch_status();
    }
    return 0;
}

extern "C" int cusrl_categorical_terms_fwd(const float *logits, const float *action, const float *old_logp, int64_t B,
                                           int64_t A, float *logp_out, float *entropy_out, float *logp_ratio_out,
                                           float *ratio_out, void *stream) {
    if (B < 0 || A <= 0 || A > INT32_MAX) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!logits || !action || !old_logp || !logp_out || !entropy_out || !logp_ratio_out || !ratio_out)
        return CUSRL_E_INVALID;
    const int64_t blocks = ceil_div(B, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(categorical_terms_fwd_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), logits,
                       action, old_logp, B, int(A), logp_out, entropy_out, logp_ratio_out, ratio_out);
    return launch_status();
}

extern "C" int cusrl_categorical_terms_bwd(const float *logits, const float *action, const float *ratio,
                                           const float *g_logp, const float *g_entropy, const float *g_logp_ratio,
                                           const float *g_ratio, int64_t B, int64_t A, float *d_logits, void *stream) {
    if (B < 0 || A <= 0 || A > INT32_MAX) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!logits || !action || !d_logits || (g_ratio && !ratio)) return CUSRL_E_INVALID;
    const int64_t blocks = ceil_div(B, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(categorical_terms_bwd_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), logits,
                       action, ratio, g_logp, g_entropy, g_logp_ratio, g_ratio, B, int(A), d_logits);
    return launch_status();
}
