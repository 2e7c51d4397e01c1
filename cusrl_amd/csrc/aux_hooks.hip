// Reward-side hooks of a captured env step and the auxiliary objectives of a minibatch step as
// single launches:
//   cusrl_reward_shaping        RewardShaping.post_step         cusrl/hook/mdp/reward.py:43-47   (mul_, add_, clamp_)
//   cusrl_amp_prepare           AdversarialMotionPrior.post_step cusrl/hook/auxiliary/amp.py:112-128
//                               state[idx] || next_state[idx], dataset[indices], two running-statistics updates, two
//                               normalisations: TWO launches instead of cat + index + 2 x (stats, finalize, merge) + 2 x normalise
//   cusrl_amp_style_reward_mean the style reward (amp.py:130-134) + the mean the metric records, one launch
//   cusrl_mse_loss_fwd_bwd      nn.MSELoss(predictor(x), target(x)) of RandomNetworkDistillation.objective (rnd.py:78-81):
//                               loss and d loss / d prediction in one pass
// All of these are a few hundred KB per launch at 4096 envs: latency-bound chains of tiny kernels in the reference's form,
// so what counts is the NUMBER of dependent launches inside the captured step (>= 1.5 us each + their own latency).
#include "common.hpp"

namespace cusrl {

// --------------------------------------------------------------------------------------------- reward shaping
// reward.mul_(scale).add_(shift) [two roundings, like the two torch ops], then clamp_(min, max) in torch's order:
// min(max(x, lo), hi), NaN propagates
__global__ __launch_bounds__(kBlock) void reward_shaping_kernel(float *__restrict__ reward, float scale, float shift,
                                                                float lower, float upper, int has_lower, int has_upper,
                                                                int64_t n) {
    const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    float r = __fadd_rn(__fmul_rn(reward[i], scale), shift);
    if (has_lower) r = (r != r) ? r : (r < lower ? lower : r);
    if (has_upper) r = (r != r) ? r : (r > upper ? upper : r);
    reward[i] = r;
}

// --------------------------------------------------------------------------------------------- AMP transition preparation
// Two launches: (A) every block assembles its slice of the agent rows and fetches its slice of the expert rows (raw, into the
// output buffers) and leaves fp64 {sum, sumsq} per channel of both; block 0 also snapshots the running statistics.
// (B) every block folds the <= 64 partial rows itself (L2-resident, fixed order), replays the two Chan merges from the
// snapshot — so no block depends on another block's result — normalises its slice in place; block 0 publishes the new
// running statistics.  (A single-workgroup form of the whole step measured 118 us: one CU's latency chain.)
constexpr int kPrepMaxC = 128;
constexpr int kPrepMaxBlocks = 64;
constexpr int kPrepItems = 8;  // elements per thread and tensor

// workspace layout (doubles): [P][2 tensors][C][2] partial sums, then the snapshot: mean[C], var[C], count
__device__ __forceinline__ int64_t prep_snapshot_offset(int P, int C) { return int64_t(P) * 2 * C * 2; }

__global__ __launch_bounds__(kBlock) void amp_assemble_kernel(
    const float *__restrict__ state, const float *__restrict__ next_state, int state_pitch,
    const int32_t *__restrict__ columns, int K, const float *__restrict__ agent_raw, const float *__restrict__ dataset,
    const int64_t *__restrict__ indices, const float *__restrict__ expert_raw, int E, int C,
    const float *__restrict__ mean, const float *__restrict__ var, const double *__restrict__ count,
    float *__restrict__ agent_out, float *__restrict__ expert_out, double *__restrict__ work) {
    __shared__ double red[kBlock][4];
    const int threads = int(gridDim.x) * kBlock;
    const int stride = threads / C * C;  // a lane stays on one channel (threads >= kBlock > C)
    const int tid = int(blockIdx.x) * kBlock + threadIdx.x;
    double sa = 0.0, qa = 0.0, se = 0.0, qe = 0.0;
    if (tid < stride) {
        const int c = tid % C;
        const int k = c < K ? c : c - K;
        const int col = agent_raw ? 0 : (columns ? columns[k] : k);
        const float *__restrict__ side = c < K ? state : next_state;
        // rounds of kPrepItems elements: every index, then every row element is requested before the first store (the
        // launch is a chain of dependent memory round trips: index -> dataset row -> store); out-of-range items are clamped
        for (int i0 = tid; i0 < E; i0 += stride * kPrepItems) {
            int idx[kPrepItems], row[kPrepItems];
            int64_t pick[kPrepItems];
            float a[kPrepItems], e[kPrepItems];
#pragma unroll
            for (int j = 0; j < kPrepItems; ++j) {
                idx[j] = min(i0 + j * stride, E - 1);
                row[j] = idx[j] / C;  // 32-bit: E <= 2^20
                if (!expert_raw) pick[j] = indices[row[j]];
            }
#pragma unroll
            for (int j = 0; j < kPrepItems; ++j) {
                // torch.cat([state[..., idx], next_state[..., idx]], -1)  amp.py:116-120;  dataset[randint]  amp.py:160-163
                const int i = row[j] * C + c;  // the clamped item re-reads this lane's channel of the last row
                a[j] = agent_raw ? agent_raw[i] : side[int64_t(row[j]) * state_pitch + col];
                e[j] = expert_raw ? expert_raw[i] : dataset[pick[j] * C + c];
            }
#pragma unroll
            for (int j = 0; j < kPrepItems; ++j) {
                const int i = i0 + j * stride;
                if (i < E) {
                    agent_out[i] = a[j];
                    expert_out[i] = e[j];
                    sa += double(a[j]), qa += double(a[j]) * double(a[j]);
                    se += double(e[j]), qe += double(e[j]) * double(e[j]);
                }
            }
        }
    }
    red[threadIdx.x][0] = sa, red[threadIdx.x][1] = qa, red[threadIdx.x][2] = se, red[threadIdx.x][3] = qe;
    __syncthreads();
    // channel c of this block = lanes whose global id is congruent to c mod C
    if (int(threadIdx.x) < C) {
        const int base = int(blockIdx.x) * kBlock;
        const int first = (int(threadIdx.x) - base % C + C) % C;
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int l = first; l < kBlock; l += C)
            for (int j = 0; j < 4; ++j) t[j] += red[l][j];
        double *out = work + (int64_t(blockIdx.x) * 2 * C + threadIdx.x) * 2;
        out[0] = t[0], out[1] = t[1];
        out[int64_t(C) * 2] = t[2], out[int64_t(C) * 2 + 1] = t[3];
    }
    if (blockIdx.x == 0) {  // the statistics every block of launch B merges into (nothing in THIS launch writes them)
        double *snap = work + prep_snapshot_offset(int(gridDim.x), C);
        for (int c = threadIdx.x; c < C; c += kBlock) snap[c] = double(mean[c]), snap[C + c] = double(var[c]);
        if (threadIdx.x == 0) snap[2 * C] = *count;
    }
}

__global__ __launch_bounds__(kBlock) void amp_normalize_kernel(const double *__restrict__ work, int P, int E, int C, int rows,
                                                               float eps, double max_count, float clamp,
                                                               float *__restrict__ mean, float *__restrict__ var,
                                                               float *__restrict__ std, double *__restrict__ count,
                                                               float *__restrict__ agent_out, float *__restrict__ expert_out) {
    __shared__ float s_mean[kPrepMaxC], s_std[kPrepMaxC];
    __shared__ double fold[kBlock][4];
    {
        // partial rows spread over the block: lane (group, c) folds rows group, group + G, ...; then one lane per channel
        // folds the G groups — both in fixed order
        const int groups = kBlock / C, c = threadIdx.x % C, group = threadIdx.x / C;
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        if (group < groups)
            for (int p = group; p < P; p += groups) {
                const double *row = work + (int64_t(p) * 2 * C + c) * 2;
                t[0] += row[0], t[1] += row[1];
                t[2] += row[int64_t(C) * 2], t[3] += row[int64_t(C) * 2 + 1];
            }
        for (int j = 0; j < 4; ++j) fold[threadIdx.x][j] = t[j];
    }
    __syncthreads();
    if (int(threadIdx.x) < C) {
        const int c = threadIdx.x, groups = kBlock / C;
        const double *snap = work + prep_snapshot_offset(P, C);
        float m = float(snap[c]), v = float(snap[C + c]);
        double running = snap[2 * C];
        // transition_rms.update(agent); transition_rms.update(expert)   amp.py:122-124 — the arithmetic of
        // masked_stats_finalize_kernel + rms_merge_kernel (population variance; Chan merge, weights count : rows, fp32)
        for (int which = 0; which < 2; ++which) {
            double s = 0.0, q = 0.0;
            for (int g = 0; g < groups; ++g) s += fold[g * C + c][2 * which], q += fold[g * C + c][2 * which + 1];
            const double n = double(rows), bm = s / n;
            double bv = q / n - bm * bm;
            if (bv < 0.0) bv = 0.0;
            const float batch_mean = float(bm), batch_var = float(bv);
            const double w_sum = running + n;
            const float w_new = float(n / w_sum), w_cross = float((running / w_sum) * (n / w_sum));
            const float delta = batch_mean - m;
            const float nm = m + delta * w_new;
            const float nv = v + ((batch_var - v) * w_new + (delta * delta) * w_cross);
            m = nm, v = nv;
            running = w_sum;
            if (max_count > 0.0 && running > max_count) running = max_count;
        }
        const float sd = sqrtf(v + eps);
        s_mean[c] = m, s_std[c] = sd;
        if (blockIdx.x == 0) {
            mean[c] = m, var[c] = v, std[c] = sd;
            if (c == 0) *count = running;
        }
    }
    __syncthreads();
    // normalise both with the statistics after BOTH updates   amp.py:125-128, rms.py:198-203
    const int threads = int(gridDim.x) * kBlock;
    const int stride = threads / C * C;
    const int tid = int(blockIdx.x) * kBlock + threadIdx.x;
    if (tid < stride) {
        const float m = s_mean[tid % C], sd = s_std[tid % C];
        for (int i = tid; i < E; i += stride) {
            float a = (agent_out[i] - m) / sd, e = (expert_out[i] - m) / sd;
            if (clamp > 0.0f) a = fminf(fmaxf(a, -clamp), clamp), e = fminf(fmaxf(e, -clamp), clamp);
            agent_out[i] = a;
            expert_out[i] = e;
        }
    }
}

// --------------------------------------------------------------------------------------------- AMP style reward + its mean
constexpr int kPrepThreads = 1024;  // one workgroup: rows of ONE env step, a single pass

__global__ __launch_bounds__(kPrepThreads) void amp_style_reward_mean_kernel(const float *__restrict__ logit,
                                                                             float *__restrict__ reward,
                                                                             float *__restrict__ bonus_out, float scale,
                                                                             int64_t rows, float *__restrict__ mean_out) {
    __shared__ double scratch[kPrepThreads / kWave];
    double total = 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += kPrepThreads) {
        // reward_scale * -log(clamp(1 - 1 / (1 + exp(-logit)), min=1e-4)), evaluated in the reference's order
        const float p = 1.0f - 1.0f / (1.0f + expf(-logit[i]));
        const float bonus = scale * -logf(fmaxf(p, 1e-4f));
        reward[i] += bonus;
        if (bonus_out) bonus_out[i] = bonus;
        total += double(bonus);
    }
    total = wave_sum(total);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) scratch[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kPrepThreads / kWave; ++w) s += scratch[w];
        *mean_out = float(s / double(rows));
    }
}

// --------------------------------------------------------------------------------------------- MSE loss, forward + backward
constexpr int kMseMaxBlocks = 1024;

// target == nullptr: zeros (a plain scaled sum of squares).  gridDim.x == 1: the block finalises itself (loss_out).
__global__ __launch_bounds__(kBlock) void mse_fwd_bwd_kernel(const float *__restrict__ prediction,
                                                             const float *__restrict__ target, int64_t n, float grad_scale,
                                                             float *__restrict__ d_prediction,
                                                             double *__restrict__ partials, double loss_scale,
                                                             float *__restrict__ loss_out) {
    __shared__ double scratch[kWavesPerBlock];
    const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x, stride = int64_t(gridDim.x) * kBlock;
    double acc = 0.0;
    if ((n & 3) == 0 && aligned_ptr16(prediction) && (!target || aligned_ptr16(target)) && aligned_ptr16(d_prediction)) {
        for (int64_t i = tid; i < n / 4; i += stride) {
            const float4 p = reinterpret_cast<const float4 *>(prediction)[i];
            const float4 t = target ? reinterpret_cast<const float4 *>(target)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float d[4] = {p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w};
            acc += double(d[0] * d[0]) + double(d[1] * d[1]) + double(d[2] * d[2]) + double(d[3] * d[3]);
            reinterpret_cast<float4 *>(d_prediction)[i] =
                make_float4(grad_scale * d[0], grad_scale * d[1], grad_scale * d[2], grad_scale * d[3]);
        }
    } else {
        for (int64_t i = tid; i < n; i += stride) {
            const float d = prediction[i] - (target ? target[i] : 0.0f);
            acc += double(d * d);
            d_prediction[i] = grad_scale * d;  // d mean((p - t)^2) / d p = 2 (p - t) / n
        }
    }
    const double total = block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        if (gridDim.x == 1)
            *loss_out = float(total * loss_scale);
        else
            partials[blockIdx.x] = total;
    }
}

__global__ __launch_bounds__(kBlock) void mse_finalize_kernel(const double *__restrict__ partials, int blocks,
                                                              double loss_scale, float *__restrict__ loss_out) {
    __shared__ double scratch[kWavesPerBlock];
    double acc = 0.0;
    for (int b = threadIdx.x; b < blocks; b += kBlock) acc += partials[b];
    const double total = block_sum(acc, scratch);
    if (threadIdx.x == 0) *loss_out = float(total * loss_scale);
}

// --------------------------------------------------------------------------------------------- BCE-with-logits of a joint batch
// logit [2 * rows]: the first `rows` are agent transitions (target 0), the rest expert transitions (target 1).
//   loss = mean_i ((1 - t_i) x_i - log_sigmoid(x_i))     torch.nn.functional.binary_cross_entropy_with_logits
//   d loss / d x_i = (sigmoid(x_i) - t_i) / (2 rows)
// scaled by `weight` — (BCE(D(agent), 0) + BCE(D(expert), 1)) / 2 * loss_weight of amp.py:143-147 is exactly the mean over
// the joint batch times loss_weight.  One workgroup (a discriminator batch is 2 x 512 rows).
__global__ __launch_bounds__(kPrepThreads) void bce_pair_fwd_bwd_kernel(const float *__restrict__ logit, int64_t rows,
                                                                        float weight, float *__restrict__ loss_out,
                                                                        float *__restrict__ d_logit) {
    __shared__ double scratch[kPrepThreads / kWave];
    const int64_t n = 2 * rows;
    const float grad_scale = weight / float(n);
    double total = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kPrepThreads) {
        const float x = logit[i], t = i < rows ? 0.0f : 1.0f;
        const float log_sigmoid = fminf(x, 0.0f) - log1pf(expf(-fabsf(x)));
        total += double((1.0f - t) * x - log_sigmoid);
        d_logit[i] = (1.0f / (1.0f + expf(-x)) - t) * grad_scale;
    }
    total = wave_sum(total);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) scratch[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kPrepThreads / kWave; ++w) s += scratch[w];
        *loss_out = float(s / double(n) * double(weight));
    }
}

// --------------------------------------------------------------------------------------------- synthetic benchmark env
// One step of the i.i.d. benchmark task of BASELINE.json config 2 (cusrl_amd/testing/environment.py): next observation
// ~ N(0, 1) [N, obs], reward ~ N(0, 1) [N, R], terminated ~ Bernoulli(p_term), truncated ~ Bernoulli(p_trunc), and one fresh
// N(0, 1) row per env for the resets — ONE launch instead of five generator launches (rand, compare, 3 x randn).
// Counter-based Philox4x32-10 keyed on (seed, step counter, stream, element): the step counter lives in device memory and is
// advanced by the kernel itself, so a hipGraph replay draws fresh numbers like an eager call does.
struct Philox {
    uint32_t c[4];
};
__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = uint64_t(0xD2511F53u) * c0, p1 = uint64_t(0xCD9E8D57u) * c2;
        const uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0, n1 = uint32_t(p1), n2 = uint32_t(p0 >> 32) ^ c3 ^ k1, n3 = uint32_t(p0);
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    return Philox{{c0, c1, c2, c3}};
}
__device__ __forceinline__ float philox_uniform(uint32_t x) { return float(x >> 8) * 5.9604645e-8f + 2.9802322e-8f; }  // (0, 1)
__device__ __forceinline__ float4 philox_normal4(const Philox &r) {  // two Box-Muller pairs
    const float r0 = sqrtf(-2.0f * logf(philox_uniform(r.c[0]))), r1 = sqrtf(-2.0f * logf(philox_uniform(r.c[2])));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * philox_uniform(r.c[1]), &s0, &c0);
    sincosf(6.283185307179586f * philox_uniform(r.c[3]), &s1, &c1);
    return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

__global__ __launch_bounds__(kBlock) void synthetic_env_step_kernel(uint64_t seed, unsigned long long *__restrict__ counter,
                                                                    int64_t N, int obs, int R, float p_term, float p_trunc,
                                                                    float *__restrict__ next_obs, float *__restrict__ reward,
                                                                    uint8_t *__restrict__ terminated,
                                                                    uint8_t *__restrict__ truncated,
                                                                    float *__restrict__ reset_rows, int64_t obs_quads,
                                                                    int64_t reward_quads) {
    // counter[0] = step number, counter[1] = arrival ticket.  Every block reads the step number first; the LAST block to
    // arrive at the ticket (all others have read by then) advances it — so a hipGraph replay draws fresh numbers like an
    // eager call, with no second launch and no host-side parity.  (Control flow only: no value depends on arrival order.)
    const unsigned long long step = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t k0 = uint32_t(seed), k1 = uint32_t(seed >> 32);
    const uint32_t s0 = uint32_t(step), s1 = uint32_t(step >> 32);
    const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t obs_total = N * obs, reward_total = N * R;
    auto store4 = [](float *dst, int64_t base, int64_t total, const float4 &v) {
        if (base + 3 < total && (reinterpret_cast<uintptr_t>(dst + base) & 15) == 0) {
            *reinterpret_cast<float4 *>(dst + base) = v;
        } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 4 && base + j < total; ++j) dst[base + j] = e[j];
        }
    };
    if (i < obs_quads) {  // stream 0: next observation, stream 1: reset rows
        store4(next_obs, i * 4, obs_total, philox_normal4(philox4x32_10(uint32_t(i), uint32_t(i >> 32) | 0x00000000u, s0, s1, k0, k1)));
        store4(reset_rows, i * 4, obs_total, philox_normal4(philox4x32_10(uint32_t(i), uint32_t(i >> 32) | 0x40000000u, s0, s1, k0, k1)));
    }
    if (i < reward_quads)  // stream 2
        store4(reward, i * 4, reward_total, philox_normal4(philox4x32_10(uint32_t(i), uint32_t(i >> 32) | 0x80000000u, s0, s1, k0, k1)));
    if (i < N) {  // stream 3: the two flags of env i
        const Philox r = philox4x32_10(uint32_t(i), uint32_t(i >> 32) | 0xC0000000u, s0, s1, k0, k1);
        terminated[i] = philox_uniform(r.c[0]) < p_term ? 1 : 0;
        truncated[i] = philox_uniform(r.c[1]) < p_trunc ? 1 : 0;
    }
    __syncthreads();  // the whole block has read `step`
    if (threadIdx.x == 0) {
        const unsigned long long arrived = atomicAdd(counter + 1, 1ull);
        if (arrived == gridDim.x - 1) {
            __hip_atomic_store(counter + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(counter, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// --------------------------------------------------------------------------------------------- metric taps of a captured step
// accumulator[i] += *value[i] for up to 32 scalars whose addresses travel by value in the kernarg segment: what a captured
// step does with the 0-d metrics its hooks record (torch: stack + add_, two launches per step).
constexpr int kMaxTapScalars = 32;
struct ScalarTable {
    const float *value[kMaxTapScalars];
};

__global__ void accumulate_scalars_kernel(const ScalarTable table, int n, float *__restrict__ accumulator) {
    const int i = threadIdx.x;
    if (i < n) accumulator[i] += *table.value[i];
}

}  // namespace cusrl

using namespace cusrl;

extern "C" int cusrl_accumulate_scalars(const float *const *values, int n, float *accumulator, void *stream) {
    if (n < 0 || n > kMaxTapScalars) return n < 0 ? CUSRL_E_INVALID : CUSRL_E_TOO_MANY;
    if (n == 0) return 0;
    if (!values || !accumulator) return CUSRL_E_INVALID;
    ScalarTable table;
    for (int i = 0; i < kMaxTapScalars; ++i) table.value[i] = i < n ? values[i] : nullptr;
    for (int i = 0; i < n; ++i)
        if (!table.value[i]) return CUSRL_E_INVALID;
    hipLaunchKernelGGL(accumulate_scalars_kernel, dim3(1), dim3(kWave), 0, as_stream(stream), table, n, accumulator);
    return launch_status();
}

extern "C" int cusrl_reward_shaping(float *reward, float scale, float shift, float lower, float upper, int has_lower,
                                    int has_upper, int64_t n, void *stream) {
    if (n < 0) return CUSRL_E_INVALID;
    if (n == 0) return 0;
    if (!reward) return CUSRL_E_INVALID;
    hipLaunchKernelGGL(reward_shaping_kernel, dim3(uint32_t(ceil_div(n, kBlock))), dim3(kBlock), 0, as_stream(stream),
                       reward, scale, shift, lower, upper, has_lower, has_upper, n);
    return launch_status();
}

extern "C" int64_t cusrl_amp_prepare_max_elements(void) { return int64_t(1) << 20; }

static int prep_blocks(int64_t elements) {
    const int64_t want = ceil_div(elements, int64_t(kBlock) * kPrepItems);
    return int(want < 1 ? 1 : (want > kPrepMaxBlocks ? kPrepMaxBlocks : want));
}

extern "C" int64_t cusrl_amp_prepare_workspace(int64_t N, int64_t C) {
    if (N <= 0 || C <= 0) return 0;
    return int64_t(prep_blocks(N * C)) * 2 * C * 2 + 2 * C + 1;  // doubles
}

extern "C" int cusrl_amp_prepare(const float *state, const float *next_state, int64_t state_pitch, const int32_t *columns,
                                 int64_t K, const float *agent_raw, const float *dataset, const int64_t *indices,
                                 const float *expert_raw, int64_t N, int64_t C, float *mean, float *var, float *std,
                                 double *count, float eps, double max_count, float clamp, float *agent_out,
                                 float *expert_out, double *workspace, void *stream) {
    if (N <= 0 || C <= 0) return CUSRL_E_INVALID;
    if (!mean || !var || !std || !count || !agent_out || !expert_out || !workspace) return CUSRL_E_INVALID;
    if (!agent_raw && (!state || !next_state || 2 * K != C || state_pitch < K || state_pitch > INT32_MAX)) return CUSRL_E_INVALID;
    if (!expert_raw && (!dataset || !indices)) return CUSRL_E_INVALID;
    if (C > kPrepMaxC || N * C > cusrl_amp_prepare_max_elements()) return CUSRL_E_UNSUPPORTED;
    const int P = prep_blocks(N * C);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(amp_assemble_kernel, dim3(P), dim3(kBlock), 0, s, state, next_state, int(state_pitch), columns, int(K),
                       agent_raw, dataset, indices, expert_raw, int(N * C), int(C), mean, var, count, agent_out, expert_out,
                       workspace);
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(amp_normalize_kernel, dim3(P), dim3(kBlock), 0, s, workspace, P, int(N * C), int(C), int(N), eps,
                       max_count, clamp, mean, var, std, count, agent_out, expert_out);
    return launch_status();
}

extern "C" int cusrl_amp_style_reward_mean(const float *logit, float *reward, float *bonus_out, float scale, int64_t rows,
                                           float *mean_out, void *stream) {
    if (rows <= 0) return CUSRL_E_INVALID;
    if (!logit || !reward || !mean_out) return CUSRL_E_INVALID;
    if (rows > (int64_t(1) << 20)) return CUSRL_E_UNSUPPORTED;  // one workgroup: beyond this use cusrl_amp_style_reward
    hipLaunchKernelGGL(amp_style_reward_mean_kernel, dim3(1), dim3(kPrepThreads), 0, as_stream(stream), logit, reward,
                       bonus_out, scale, rows, mean_out);
    return launch_status();
}

extern "C" int64_t cusrl_mse_loss_num_partials(int64_t n) {
    if (n <= 0) return 0;
    if (n <= int64_t(kBlock) * 64) return 1;  // small batches (a discriminator batch: 6 K elements): one block finalises itself
    const int64_t want = ceil_div(n, int64_t(kBlock) * 8);
    return want > kMseMaxBlocks ? kMseMaxBlocks : want;
}

static int launch_sumsq(const float *x, const float *target, int64_t n, double loss_scale, float grad_scale, float *loss_out,
                        float *grad_out, double *partials, hipStream_t s) {
    const int64_t blocks = cusrl_mse_loss_num_partials(n);
    hipLaunchKernelGGL(mse_fwd_bwd_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, s, x, target, n, grad_scale, grad_out,
                       partials, loss_scale, loss_out);
    if (int rc = launch_status()) return rc;
    if (blocks == 1) return 0;  // the one block finalised itself
    hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(kBlock), 0, s, partials, int(blocks), loss_scale, loss_out);
    return launch_status();
}

extern "C" int cusrl_mse_loss_fwd_bwd(const float *prediction, const float *target, int64_t n, float *loss_out,
                                      float *d_prediction, double *partials, void *stream) {
    if (n <= 0) return CUSRL_E_INVALID;
    if (!prediction || !target || !loss_out || !d_prediction || !partials) return CUSRL_E_INVALID;
    return launch_sumsq(prediction, target, n, 1.0 / double(n), float(2.0 / double(n)), loss_out, d_prediction, partials,
                        as_stream(stream));
}

extern "C" int cusrl_sumsq_fwd_bwd(const float *x, int64_t n, double loss_scale, double grad_scale, float *loss_out,
                                   float *grad_out, double *partials, void *stream) {
    if (n <= 0) return CUSRL_E_INVALID;
    if (!x || !loss_out || !grad_out || !partials) return CUSRL_E_INVALID;
    return launch_sumsq(x, nullptr, n, loss_scale, float(grad_scale), loss_out, grad_out, partials, as_stream(stream));
}

extern "C" int cusrl_bce_pair_fwd_bwd(const float *logit, int64_t rows, float weight, float *loss_out, float *d_logit,
                                      void *stream) {
    if (rows <= 0) return CUSRL_E_INVALID;
    if (!logit || !loss_out || !d_logit) return CUSRL_E_INVALID;
    if (rows > (int64_t(1) << 19)) return CUSRL_E_UNSUPPORTED;  // one workgroup
    hipLaunchKernelGGL(bce_pair_fwd_bwd_kernel, dim3(1), dim3(kPrepThreads), 0, as_stream(stream), logit, rows, weight,
                       loss_out, d_logit);
    return launch_status();
}

extern "C" int cusrl_synthetic_env_step(uint64_t seed, uint64_t *counter, int64_t N, int64_t obs_dim, int64_t reward_dim,
                                        float p_terminate, float p_truncate, float *next_observation, float *reward,
                                        uint8_t *terminated, uint8_t *truncated, float *reset_rows, void *stream) {
    if (N <= 0 || obs_dim <= 0 || reward_dim <= 0) return CUSRL_E_INVALID;
    if (!counter || !next_observation || !reward || !terminated || !truncated || !reset_rows) return CUSRL_E_INVALID;
    if (obs_dim > INT32_MAX || reward_dim > INT32_MAX || N * obs_dim > (int64_t(1) << 40)) return CUSRL_E_UNSUPPORTED;
    const int64_t obs_quads = ceil_div(N * obs_dim, 4), reward_quads = ceil_div(N * reward_dim, 4);
    int64_t threads = obs_quads > reward_quads ? obs_quads : reward_quads;
    if (N > threads) threads = N;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(synthetic_env_step_kernel, dim3(uint32_t(ceil_div(threads, kBlock))), dim3(kBlock), 0, s, seed,
                       reinterpret_cast<unsigned long long *>(counter), N, int(obs_dim), int(reward_dim), p_terminate,
                       p_truncate, next_observation, reward, terminated, truncated, reset_rows, obs_quads, reward_quads);
    return launch_status();
}
