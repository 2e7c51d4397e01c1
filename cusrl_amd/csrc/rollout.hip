// Rollout-side kernels for gfx950: fused Normal sample + log-prob (a12, acting side) and one-launch episode
// statistics (removes the per-step device->host sync of the reference trainer loop).
#include "common.hpp"
#include "push_body.hpp"

namespace cusrl {

__device__ __forceinline__ float log_sqrt_2pi_r() { return 0.918938533204672741780329736406f; }

// One lane per row for any A; rows are short (A = 12 -> 48 B) and the batch is one env step (N rows), so this is
// latency-bound, not bandwidth-bound.  A % 4 == 0 rows are read as 16 B chunks.
// `std_rows` = 1: std is the [A] vector a state-independent std repeats for every row (distribution.py:228-247); the kernel
// broadcasts it and, with `std_out`, also writes the repeated [B, A] matrix the rollout buffer stores as a leaf — the
// `repeat` launch of the acting path folded into this one.
template <bool kVec4>
__global__ __launch_bounds__(kBlock) void normal_sample_logp_kernel(const float *__restrict__ mean,
                                                                    const float *__restrict__ std,
                                                                    const float *__restrict__ eps,
                                                                    float *__restrict__ action,
                                                                    float *__restrict__ logp, int64_t B, int A,
                                                                    int std_is_vector, float *__restrict__ std_out,
                                                                    const float *__restrict__ mean_bias,
                                                                    float *__restrict__ mean_out) {
    // mean_bias (optional, [A]): `mean` is the policy head's product WITHOUT its bias; the bias is added here and the
    // finished mean goes to mean_out [B, A] — for a 12-column head the library adds the bias by broadcasting it into the
    // output with a copy launch before the GEMM, which costs more than the GEMM itself.
    const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (row >= B) return;
    float lp = 0.0f;
    const int64_t std_row = std_is_vector ? 0 : row;
    if constexpr (kVec4) {
        const float4 *m4 = reinterpret_cast<const float4 *>(mean + row * A);
        const float4 *s4 = reinterpret_cast<const float4 *>(std + std_row * A);
        const float4 *e4 = reinterpret_cast<const float4 *>(eps + row * A);
        float4 *a4 = reinterpret_cast<float4 *>(action + row * A);
        float4 *so4 = std_out ? reinterpret_cast<float4 *>(std_out + row * A) : nullptr;
        const float4 *b4 = reinterpret_cast<const float4 *>(mean_bias);
        float4 *mo4 = reinterpret_cast<float4 *>(mean_out + (mean_out ? row * A : 0));
        for (int c = 0; c < A / 4; ++c) {
            float4 m = m4[c];
            const float4 s = s4[c], e = e4[c];
            if (mean_bias) {
                const float4 b = b4[c];
                m.x += b.x, m.y += b.y, m.z += b.z, m.w += b.w;
                mo4[c] = m;
            }
            if (so4) so4[c] = s;
            const float ms[4] = {m.x, m.y, m.z, m.w}, ss[4] = {s.x, s.y, s.z, s.w}, es[4] = {e.x, e.y, e.z, e.w};
            float as[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                as[j] = ms[j] + es[j] * ss[j];  // rsample: loc + eps * scale
                const float diff = as[j] - ms[j];
                lp += -(diff * diff) / (2.0f * (ss[j] * ss[j])) - logf(ss[j]) - log_sqrt_2pi_r();
            }
            a4[c] = make_float4(as[0], as[1], as[2], as[3]);
        }
    } else {
        for (int a = 0; a < A; ++a) {
            const int64_t i = row * A + a;
            const float sg = std[std_row * A + a];
            const float mu = mean_bias ? mean[i] + mean_bias[a] : mean[i];
            if (mean_bias) mean_out[i] = mu;
            const float act = mu + eps[i] * sg;
            const float diff = act - mu;
            action[i] = act;
            if (std_out) std_out[i] = sg;
            lp += -(diff * diff) / (2.0f * (sg * sg)) - logf(sg) - log_sqrt_2pi_r();
        }
    }
    logp[row] = lp;
}

// Acting side of a one-hot categorical policy (cusrl/nn/module/distribution.py:332-366, OneHotCategorical.sample +
// log_prob), the draw done the way torch.multinomial draws ONE sample on the device: idx = argmax_j p_j / q_j with
// q ~ Exp(1) taken from torch's generator by the caller (the exponential race; p may stay unnormalised, a common
// positive factor does not move the argmax).  action = one_hot(idx), logp = logits[idx] - logsumexp(logits).
// kWavePerRow = false: one lane per row (A <= 32, an env step is N short rows: latency-bound); true: one wave per row,
// lanes stride over the categories, max / sum / arg-max by xor shuffles.  Ties go to the lower index.
template <bool kWavePerRow>
__global__ __launch_bounds__(kBlock) void categorical_sample_logp_kernel(const float *__restrict__ logits,
                                                                         const float *__restrict__ noise,
                                                                         float *__restrict__ action,
                                                                         float *__restrict__ logp, int64_t B, int A) {
    const int64_t thread = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if constexpr (!kWavePerRow) {
        const int64_t row = thread;
        if (row >= B) return;
        const float *z = logits + row * A, *q = noise + row * A;
        float m = z[0];
        for (int j = 1; j < A; ++j) m = fmaxf(m, z[j]);
        float sum = 0.0f;
        for (int j = 0; j < A; ++j) sum += expf(z[j] - m);
        float best = -INFINITY;
        int taken = 0;
        for (int j = 0; j < A; ++j) {
            const float race = expf(z[j] - m) / q[j];
            if (race > best) best = race, taken = j;
        }
        for (int j = 0; j < A; ++j) action[row * A + j] = j == taken ? 1.0f : 0.0f;
        logp[row] = z[taken] - (m + logf(sum));
    } else {
        const int64_t row = thread / kWave;
        const int lane = threadIdx.x & (kWave - 1);
        if (row >= B) return;  // uniform over the wave
        const float *z = logits + row * A, *q = noise + row * A;
        float m = -INFINITY;
        for (int j = lane; j < A; j += kWave) m = fmaxf(m, z[j]);
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
        float sum = 0.0f, best = -INFINITY;
        int taken = INT32_MAX;
        for (int j = lane; j < A; j += kWave) {
            const float e = expf(z[j] - m);
            sum += e;
            const float race = e / q[j];
            if (race > best) best = race, taken = j;
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            sum += __shfl_xor(sum, off, kWave);
            const float other = __shfl_xor(best, off, kWave);
            const int other_taken = __shfl_xor(taken, off, kWave);
            if (other > best || (other == best && other_taken < taken)) best = other, taken = other_taken;
        }
        if (taken == INT32_MAX) taken = 0;  // every race was NaN
        for (int j = lane; j < A; j += kWave) action[row * A + j] = j == taken ? 1.0f : 0.0f;
        if (lane == 0) logp[row] = z[taken] - (m + logf(sum));
    }
}

// Ring slots are handed out in ascending env order, like the reference's `(arange(count) + num_episodes) % R`
// (trainer.py:63-70): a block's first slot = episodes so far + finished envs in all EARLIER blocks, which every block
// counts for itself from the flag bytes (b x 256 bytes, 16 per lane-load, L2-resident; no count launch, no ticket).
// The running episode counter is double-buffered — every block reads counter[parity], the last block writes
// counter[parity ^ 1] — so no block can observe this step's update.  kOrdered = false (N > 262 144, where the
// quadratic flag re-read would matter) falls back to an atomic ticket: same slots, unspecified order within the step.
constexpr int64_t kOrderedStatsMaxEnvs = 262144;

template <bool kOrdered>
__device__ __forceinline__ void episode_stats_body(const int block, const int num_blocks, uint8_t *__restrict__ done_copy,
                                                   const float *__restrict__ reward,
                                                   const uint8_t *__restrict__ done,
                                                   const uint8_t *__restrict__ done_b, uint8_t *__restrict__ done_out,
                                                   float *__restrict__ episode_rew, float *__restrict__ episode_len,
                                                   float *__restrict__ ring_rew, float *__restrict__ ring_len,
                                                   unsigned long long *__restrict__ num_episodes,
                                                   double *__restrict__ step_reward_sum,
                                                   int64_t *__restrict__ indices_out, int32_t *__restrict__ count_out,
                                                   int64_t N, int D, int64_t R, int parity) {
    // `block` of `num_blocks`: the launch's block index (the plain launch) or the epilogue part of a fused launch;
    // done_copy != NULL: the flag is ALSO stored there (the rollout buffer's `done` slab of this step)
    // `done_b` != NULL: the flag of env n is done[n] | done_b[n] (terminated | truncated, actor_critic.py:277), written
    // to done_out; indices_out / count_out != NULL: the finished envs in ascending order + their number
    // (environment.py:356-362 `get_done_indices`), the count with a system-scope store (may be pinned host memory).
    __shared__ double scratch[kWavesPerBlock];
    __shared__ int iscratch[kWavesPerBlock];
    const int64_t n = int64_t(block) * kBlock + threadIdx.x;
    const bool active = n < N;
    bool finished = false;
    if (active) {
        finished = done[n] != 0 || (done_b && done_b[n] != 0);
        if (done_out) done_out[n] = finished ? 1 : 0;
        if (done_copy) done_copy[n] = finished ? 1 : 0;
    }
    unsigned long long slot = 0;
    float len = 0.0f;
    if (active) len = episode_len[n] + 1.0f;
    if constexpr (kOrdered) {
        int earlier = 0;
        const int64_t chunks = int64_t(block) * (kBlock / 16);  // 16-flag chunks in front of this block
        const bool vec = ((reinterpret_cast<uintptr_t>(done) | reinterpret_cast<uintptr_t>(done_b)) & 15) == 0;
        for (int64_t c = threadIdx.x; c < chunks; c += kBlock) {
            if (vec) {
                uint4 v = reinterpret_cast<const uint4 *>(done)[c];
                if (done_b) {
                    const uint4 u = reinterpret_cast<const uint4 *>(done_b)[c];
                    v.x |= u.x, v.y |= u.y, v.z |= u.z, v.w |= u.w;
                }
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t x = w[k];
                    x |= x >> 4, x |= x >> 2, x |= x >> 1;  // any bit of a byte -> its bit 0
                    earlier += __popc(x & 0x01010101u);
                }
            } else {
                for (int j = 0; j < 16; ++j) earlier += (done[c * 16 + j] != 0 || (done_b && done_b[c * 16 + j] != 0));
            }
        }
        int own_total;
        const int own_prefix = block_exclusive_scan(finished ? 1 : 0, iscratch, own_total);
        const int before = block_sum(earlier, iscratch);
        __shared__ int s_before;
        if (threadIdx.x == 0) s_before = before;
        __syncthreads();
        const unsigned long long base = num_episodes[parity];
        if (finished) {
            slot = (base + (unsigned long long)(s_before + own_prefix)) % (unsigned long long)R;
            if (indices_out) indices_out[s_before + own_prefix] = n;
        }
        if (block == num_blocks - 1 && threadIdx.x == 0) {
            num_episodes[parity ^ 1] = base + (unsigned long long)(s_before + own_total);
            if (count_out) __hip_atomic_store(count_out, s_before + own_total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else {
        if (finished) slot = atomicAdd(num_episodes + parity, 1ull) % (unsigned long long)R;
    }
    if (finished) ring_len[slot] = len;
    for (int d = 0; d < D; ++d) {
        float r = 0.0f;
        if (active) {
            r = reward[n * D + d];
            const float total = episode_rew[n * D + d] + r;
            if (finished) ring_rew[slot * D + d] = total;
            episode_rew[n * D + d] = finished ? 0.0f : total;
        }
        const double block_total = block_sum(double(r), scratch);
        if (threadIdx.x == 0) atomicAdd(step_reward_sum + d, block_total);
    }
    if (active) episode_len[n] = finished ? 0.0f : len;
}


template <bool kOrdered>
__global__ __launch_bounds__(kBlock) void episode_stats_kernel(const float *__restrict__ reward,
                                                               const uint8_t *__restrict__ done,
                                                               const uint8_t *__restrict__ done_b,
                                                               uint8_t *__restrict__ done_out,
                                                               float *__restrict__ episode_rew,
                                                               float *__restrict__ episode_len,
                                                               float *__restrict__ ring_rew,
                                                               float *__restrict__ ring_len,
                                                               unsigned long long *__restrict__ num_episodes,
                                                               double *__restrict__ step_reward_sum,
                                                               int64_t *__restrict__ indices_out,
                                                               int32_t *__restrict__ count_out, int64_t N, int D,
                                                               int64_t R, int parity) {
    episode_stats_body<kOrdered>(blockIdx.x, gridDim.x, nullptr, reward, done, done_b, done_out, episode_rew, episode_len,
                                 ring_rew, ring_len, num_episodes, step_reward_sum, indices_out, count_out, N, D, R, parity);
}

// The step epilogue AND the buffer append of the same env step as ONE launch: the first `epilogue_blocks` blocks are the
// epilogue (they also store the `done` flag straight into the buffer's slab, so that leaf needs no copy that would have to
// wait for them), the remaining blocks are the push of every other leaf — no block of one part reads what the other writes.
struct EpilogueArgs {
    const float *reward;
    const uint8_t *terminated, *truncated;
    uint8_t *done_out, *done_slab;
    float *episode_rew, *episode_len, *ring_rew, *ring_len;
    unsigned long long *num_episodes;
    double *step_reward_sum;
    int64_t *indices_out;
    int32_t *count_out;
    int64_t N, R;
    int D, parity, epilogue_blocks;
};

__global__ __launch_bounds__(kBlock) void step_epilogue_push_kernel(const EpilogueArgs e, const PushTable tab) {
    if (int(blockIdx.x) < e.epilogue_blocks)
        episode_stats_body<true>(blockIdx.x, e.epilogue_blocks, e.done_slab, e.reward, e.terminated, e.truncated, e.done_out,
                                 e.episode_rew, e.episode_len, e.ring_rew, e.ring_len, e.num_episodes, e.step_reward_sum,
                                 e.indices_out, e.count_out, e.N, e.D, e.R, e.parity);
    else
        push_body<0>(tab, int(blockIdx.x) - e.epilogue_blocks);
}

// ---- intrinsic-reward epilogues: one launch instead of sub / square / mean / mul / add_ (RND) or
// exp / add / reciprocal / rsub / clamp / log / neg / mul / add_ (AMP)
__global__ __launch_bounds__(kBlock) void rnd_reward_kernel(const float *__restrict__ target,
                                                            const float *__restrict__ prediction,
                                                            float *__restrict__ reward, float *__restrict__ bonus_out,
                                                            float scale, int64_t rows, int K) {
    const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= rows) return;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float d = target[i * K + k] - prediction[i * K + k];
        acc += d * d;
    }
    const float bonus = scale * (acc / float(K));  // reward_scale * (target - prediction).square().mean(-1)
    reward[i] += bonus;
    if (bonus_out) bonus_out[i] = bonus;
}

__global__ __launch_bounds__(kBlock) void amp_style_reward_kernel(const float *__restrict__ logit,
                                                                  float *__restrict__ reward,
                                                                  float *__restrict__ bonus_out, float scale,
                                                                  int64_t rows) {
    const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= rows) return;
    // reward_scale * -log(clamp(1 - 1 / (1 + exp(-logit)), min=1e-4)), evaluated in the reference's order
    const float p = 1.0f - 1.0f / (1.0f + expf(-logit[i]));
    const float bonus = scale * -logf(fmaxf(p, 1e-4f));
    reward[i] += bonus;
    if (bonus_out) bonus_out[i] = bonus;
}

}  // namespace cusrl

using namespace cusrl;

extern "C" int cusrl_rnd_reward(const float *target, const float *prediction, float *reward, float *bonus_out,
                                float scale, int64_t rows, int64_t K, void *stream) {
    if (rows < 0 || K <= 0) return CUSRL_E_INVALID;
    if (rows == 0) return 0;
    if (!target || !prediction || !reward) return CUSRL_E_INVALID;
    if (K > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(rnd_reward_kernel, dim3(uint32_t(ceil_div(rows, kBlock))), dim3(kBlock), 0, as_stream(stream),
                       target, prediction, reward, bonus_out, scale, rows, int(K));
    return launch_status();
}

extern "C" int cusrl_amp_style_reward(const float *logit, float *reward, float *bonus_out, float scale, int64_t rows,
                                      void *stream) {
    if (rows < 0) return CUSRL_E_INVALID;
    if (rows == 0) return 0;
    if (!logit || !reward) return CUSRL_E_INVALID;
    hipLaunchKernelGGL(amp_style_reward_kernel, dim3(uint32_t(ceil_div(rows, kBlock))), dim3(kBlock), 0,
                       as_stream(stream), logit, reward, bonus_out, scale, rows);
    return launch_status();
}

extern "C" int cusrl_normal_sample_logp(const float *mean, const float *std, const float *eps, float *action,
                                        float *logp, int64_t B, int64_t A, int64_t std_rows, float *std_out,
                                        const float *mean_bias, float *mean_out, void *stream) {
    if (B < 0 || A <= 0 || (std_rows != B && std_rows != 1)) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!mean || !std || !eps || !action || !logp || (mean_bias != nullptr) != (mean_out != nullptr)) return CUSRL_E_INVALID;
    if (A > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const int64_t blocks = ceil_div(B, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const int vector = std_rows == 1 && B != 1;
    const bool vec4 = A % 4 == 0 && aligned(mean, 16) && aligned(std, 16) && aligned(eps, 16) && aligned(action, 16) &&
                      (!std_out || aligned(std_out, 16)) && (!mean_bias || (aligned(mean_bias, 16) && aligned(mean_out, 16)));
    if (vec4)
        hipLaunchKernelGGL(normal_sample_logp_kernel<true>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream),
                           mean, std, eps, action, logp, B, int(A), vector, std_out, mean_bias, mean_out);
    else
        hipLaunchKernelGGL(normal_sample_logp_kernel<false>, dim3(uint32_t(blocks)), dim3(kBlock), 0,
                           as_stream(stream), mean, std, eps, action, logp, B, int(A), vector, std_out, mean_bias, mean_out);
    return launch_status();
}

extern "C" int cusrl_categorical_sample_logp(const float *logits, const float *noise, float *action, float *logp,
                                             int64_t B, int64_t A, void *stream) {
    if (B < 0 || A <= 0) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!logits || !noise || !action || !logp) return CUSRL_E_INVALID;
    if (A > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const bool wave_per_row = A > 32;
    const int64_t blocks = ceil_div(wave_per_row ? B * kWave : B, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    if (wave_per_row)
        hipLaunchKernelGGL(categorical_sample_logp_kernel<true>, dim3(uint32_t(blocks)), dim3(kBlock), 0,
                           as_stream(stream), logits, noise, action, logp, B, int(A));
    else
        hipLaunchKernelGGL(categorical_sample_logp_kernel<false>, dim3(uint32_t(blocks)), dim3(kBlock), 0,
                           as_stream(stream), logits, noise, action, logp, B, int(A));
    return launch_status();
}

extern "C" int cusrl_episode_stats(const float *reward, const uint8_t *done, float *episode_rew, float *episode_len,
                                   float *ring_rew, float *ring_len, uint64_t *num_episodes, double *step_reward_sum,
                                   int64_t N, int64_t D, int64_t R, int parity, void *stream) {
    if (N < 0 || D <= 0 || R <= 0 || (parity != 0 && parity != 1)) return CUSRL_E_INVALID;
    if (N == 0) return 0;
    if (!reward || !done || !episode_rew || !episode_len || !ring_rew || !ring_len || !num_episodes || !step_reward_sum)
        return CUSRL_E_INVALID;
    if (D > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const int64_t blocks = ceil_div(N, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    unsigned long long *counter = reinterpret_cast<unsigned long long *>(num_episodes);
    if (N <= kOrderedStatsMaxEnvs) {
        hipLaunchKernelGGL(episode_stats_kernel<true>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), reward,
                           done, nullptr, nullptr, episode_rew, episode_len, ring_rew, ring_len, counter, step_reward_sum,
                           nullptr, nullptr, N, int(D), R, parity);
    } else {  // ticket form: the counter of this step is first copied over, then bumped atomically
        if (hipError_t e = hipMemcpyAsync(counter + (parity ^ 1), counter + parity, sizeof(unsigned long long),
                                          hipMemcpyDeviceToDevice, as_stream(stream)))
            return int(e);
        hipLaunchKernelGGL(episode_stats_kernel<false>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), reward,
                           done, nullptr, nullptr, episode_rew, episode_len, ring_rew, ring_len, counter, step_reward_sum,
                           nullptr, nullptr, N, int(D), R, parity ^ 1);
    }
    return launch_status();
}

extern "C" int64_t cusrl_step_epilogue_max_envs(void) { return kOrderedStatsMaxEnvs; }

extern "C" int cusrl_step_epilogue_push(const float *reward, const uint8_t *terminated, const uint8_t *truncated,
                                        uint8_t *done_out, float *episode_rew, float *episode_len, float *ring_rew,
                                        float *ring_len, uint64_t *num_episodes, double *step_reward_sum,
                                        int64_t *indices_out, int32_t *count_out, int64_t N, int64_t D, int64_t R,
                                        int parity, const cusrl_field_t *fields, int n_fields, int done_field,
                                        int64_t cursor, void *stream) {
    if (N <= 0 || D <= 0 || R <= 0 || (parity != 0 && parity != 1)) return CUSRL_E_INVALID;
    if (!reward || !terminated || !truncated || !done_out || !episode_rew || !episode_len || !ring_rew || !ring_len ||
        !num_episodes || !step_reward_sum || !indices_out || !count_out)
        return CUSRL_E_INVALID;
    if (D > INT32_MAX || N > kOrderedStatsMaxEnvs) return CUSRL_E_UNSUPPORTED;
    if (!fields || done_field < 0 || done_field >= n_fields || fields[done_field].row_bytes != 1 || !fields[done_field].dst)
        return CUSRL_E_INVALID;  // the `done` leaf: one flag byte per env, stored by the epilogue blocks themselves
    PushTable tab;
    int32_t push_blocks = 0;
    int64_t step_bytes = 0;
    if (int rc = build_push_table(fields, n_fields, cursor, N, nullptr, 0, nullptr, done_field, tab, push_blocks, step_bytes))
        return rc;
    EpilogueArgs e;
    e.reward = reward, e.terminated = terminated, e.truncated = truncated, e.done_out = done_out;
    e.done_slab = static_cast<uint8_t *>(fields[done_field].dst) + cursor * N;
    e.episode_rew = episode_rew, e.episode_len = episode_len, e.ring_rew = ring_rew, e.ring_len = ring_len;
    e.num_episodes = reinterpret_cast<unsigned long long *>(num_episodes);
    e.step_reward_sum = step_reward_sum, e.indices_out = indices_out, e.count_out = count_out;
    e.N = N, e.R = R, e.D = int(D), e.parity = parity;
    e.epilogue_blocks = int(ceil_div(N, kBlock));
    const int64_t blocks = int64_t(e.epilogue_blocks) + push_blocks;
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(step_epilogue_push_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), e, tab);
    return launch_status();
}

extern "C" int cusrl_step_epilogue(const float *reward, const uint8_t *terminated, const uint8_t *truncated,
                                   uint8_t *done_out, float *episode_rew, float *episode_len, float *ring_rew,
                                   float *ring_len, uint64_t *num_episodes, double *step_reward_sum,
                                   int64_t *indices_out, int32_t *count_out, int64_t N, int64_t D, int64_t R, int parity,
                                   void *stream) {
    if (N <= 0 || D <= 0 || R <= 0 || (parity != 0 && parity != 1)) return CUSRL_E_INVALID;
    if (!reward || !terminated || !truncated || !done_out || !episode_rew || !episode_len || !ring_rew || !ring_len ||
        !num_episodes || !step_reward_sum || !indices_out || !count_out)
        return CUSRL_E_INVALID;
    if (D > INT32_MAX || N > kOrderedStatsMaxEnvs) return CUSRL_E_UNSUPPORTED;
    const int64_t blocks = ceil_div(N, kBlock);
    hipLaunchKernelGGL(episode_stats_kernel<true>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), reward,
                       terminated, truncated, done_out, episode_rew, episode_len, ring_rew, ring_len,
                       reinterpret_cast<unsigned long long *>(num_episodes), step_reward_sum, indices_out, count_out, N,
                       int(D), R, parity);
    return launch_status();
}
