/*
 * cusrl_oracle.c — CPU restatement of the reference's rollout + PPO-update hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cusrl_amd/ may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
 * only as the checker.  Every function cites the reference lines it restates
 * (paths relative to /root/reference).  The reference is pure Python on top of
 * PyTorch (torch>=2.5, requirements.txt:7; torch 2.10.0 here), so the arithmetic that
 * lives inside torch (randperm, var_mean, Normal.log_prob …) is restated from torch's
 * published algorithms and PINNED by the golden vectors in tests/golden/ (npz files), which
 * were produced by running the reference itself (tests/golden/make_golden.py).
 *
 * Plain C99, no dependencies: `gcc -O2 -ffp-contract=off -shared -fPIC`.
 * -ffp-contract=off matters: the reference evaluates `a + b*c` as two rounded fp32
 * operations (SURVEY.md §7 "FMA contraction breaks parity").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ a1: Buffer.push
 * cusrl/template/buffer.py:124-151 — `storage[cursor] = value` for every leaf.
 * One leaf: step [N*row_bytes] -> storage[cursor] of [T, N*row_bytes]. */
void oracle_buffer_push(const void *step, void *storage, int64_t cursor, int64_t step_bytes) {
    memcpy((char *)storage + cursor * step_bytes, step, (size_t)step_bytes);
}

/* ------------------------------------------------------------------ a3: next_value
 * cusrl/hook/on_policy/value.py:56-82
 *   next_value[:-1] = value[1:]                         (:66)
 *   next_value[-1]  = critic(next_state[-1])            (:68)  -> last_value [N,D]
 *   next_value[terminated] = termination_value          (:69-70)
 *   if truncated.any():                                 (:71)
 *     bootstrap:  next_value[truncated] = critic(next_state[truncated])  (:72-78)
 *                 -> trunc_values [K,D], rows in ascending flat (t*N+n) order
 *     otherwise:  next_value[truncated] = value[truncated]               (:79-80)
 * Returns the number of truncated slots K. */
int64_t oracle_next_value(const float *value, const uint8_t *terminated, const uint8_t *truncated,
                          const float *last_value, const float *trunc_values, int bootstrap,
                          float termination_value, float *next_value, int64_t T, int64_t N, int64_t D) {
    int64_t row = N * D;
    for (int64_t t = 0; t + 1 < T; ++t) memcpy(next_value + t * row, value + (t + 1) * row, sizeof(float) * row);
    memcpy(next_value + (T - 1) * row, last_value, sizeof(float) * row);
    for (int64_t s = 0; s < T * N; ++s)
        if (terminated[s])
            for (int64_t d = 0; d < D; ++d) next_value[s * D + d] = termination_value;
    int64_t k = 0;
    for (int64_t s = 0; s < T * N; ++s) {
        if (!truncated[s]) continue;
        for (int64_t d = 0; d < D; ++d)
            next_value[s * D + d] = bootstrap ? trunc_values[k * D + d] : value[s * D + d];
        ++k;
    }
    return k;
}

/* ------------------------------------------------------------------ a4: GAE(lambda)
 * cusrl/hook/on_policy/gae.py:8-20
 *   advantage = reward + next_value * gamma - value
 *   for step in T-2..0: advantage[step] += not_done[step] * (gamma * lamda) * advantage[step + 1]
 * Exact fp32 order (verified bit-exact against the reference):
 *   delta = (r + nv * (float)gamma) - v ;  c = (float)((double)gamma * (double)lamda)
 *   `not_done * c` is bool*python-float -> fp32 tensor (c or 0), then * A[t+1], then +=.
 * done is [T,N,1] and broadcasts over the D value channels. */
void oracle_gae_scan(const float *reward, const uint8_t *done, const float *value, const float *next_value,
                     double gamma, double lamda, float *advantage, int64_t T, int64_t N, int64_t D) {
    const float g = (float)gamma;
    const float c = (float)(gamma * lamda);
    for (int64_t i = 0; i < T * N * D; ++i) {
        float nvg = next_value[i] * g;
        float s = reward[i] + nvg;
        advantage[i] = s - value[i];
    }
    for (int64_t t = T - 2; t >= 0; --t)
        for (int64_t n = 0; n < N; ++n) {
            float coef = done[t * N + n] ? 0.0f : 1.0f;
            coef = coef * c;
            for (int64_t d = 0; d < D; ++d) {
                int64_t i = (t * N + n) * D + d;
                float carry = coef * advantage[i + N * D];
                advantage[i] = advantage[i] + carry;
            }
        }
}

/* cusrl/hook/on_policy/gae.py:85-110 — advantage with lamda; return = value + advantage, or
 * value + (second scan with lamda_value) when lamda_value >= 0 (None is passed as < 0). */
void oracle_gae(const float *reward, const uint8_t *done, const float *value, const float *next_value,
                double gamma, double lamda, double lamda_value, float *advantage, float *ret, int64_t T,
                int64_t N, int64_t D) {
    int64_t total = T * N * D;
    oracle_gae_scan(reward, done, value, next_value, gamma, lamda, advantage, T, N, D);
    if (lamda_value < 0) {
        for (int64_t i = 0; i < total; ++i) ret[i] = value[i] + advantage[i];
    } else {
        float *tmp = (float *)malloc(sizeof(float) * (size_t)total);
        oracle_gae_scan(reward, done, value, next_value, gamma, lamda_value, tmp, T, N, D);
        for (int64_t i = 0; i < total; ++i) ret[i] = value[i] + tmp[i];
        free(tmp);
    }
}

/* ------------------------------------------------------------------ a5: advantage normalisation
 * cusrl/hook/on_policy/advantage.py:108-115
 *   var, mean = torch.var_mean(advantage, dim=all-but-last)   (unbiased, correction=1)
 *   std = (var + 1e-8).sqrt();  advantage.sub_(mean).div_(std)
 * torch's reduction order is unspecified -> two-pass in double here; parity is by tolerance. */
void oracle_var_mean(const float *x, int64_t rows, int64_t D, float *mean, float *var) {
    for (int64_t d = 0; d < D; ++d) {
        double s = 0.0;
        for (int64_t r = 0; r < rows; ++r) s += (double)x[r * D + d];
        double m = s / (double)rows, q = 0.0;
        for (int64_t r = 0; r < rows; ++r) {
            double e = (double)x[r * D + d] - m;
            q += e * e;
        }
        mean[d] = (float)m;
        var[d] = (float)(q / (double)(rows - 1)); /* rows==1 -> nan, like torch */
    }
}

void oracle_normalize(float *x, int64_t rows, int64_t D, const float *mean, const float *var) {
    for (int64_t d = 0; d < D; ++d) {
        float std = sqrtf(var[d] + 1e-8f);
        for (int64_t r = 0; r < rows; ++r) {
            float c = x[r * D + d] - mean[d];
            x[r * D + d] = c / std;
        }
    }
}

/* ------------------------------------------------------------------ a6: cross-rank merge
 * cusrl/utils/distributed.py:175-183 — equal-weight merge, NOT the pooled variance:
 *   mean = avg_r(mean_r);  var = avg_r(var_r + (mean_r - mean)^2) */
void oracle_merge_mean_var(const float *means, const float *vars, int64_t W, int64_t D, float *mean, float *var) {
    for (int64_t d = 0; d < D; ++d) {
        float s = 0.0f;
        for (int64_t r = 0; r < W; ++r) s += means[r * D + d];
        float m = s / (float)W;
        float q = 0.0f;
        for (int64_t r = 0; r < W; ++r) {
            float e = means[r * D + d] - m;
            q += vars[r * D + d] + e * e;
        }
        mean[d] = m;
        var[d] = q / (float)W;
    }
}

/* ------------------------------------------------------------------ a7/a8: minibatch gather
 * cusrl/sampler/mini_batch_sampler.py:87-89   data.flatten(0,1)[indices]   (temporal == 0)
 * cusrl/sampler/mini_batch_sampler.py:113-114 data[:, indices]            (temporal != 0)
 * One leaf of row_bytes per (t, n) slot.  out is [B,row] or [T,B,row]. */
void oracle_gather_rows(const void *storage, const int64_t *indices, void *out, int64_t B, int64_t T,
                        int64_t N, int64_t row_bytes, int temporal) {
    const char *src = (const char *)storage;
    char *dst = (char *)out;
    if (!temporal) {
        for (int64_t b = 0; b < B; ++b) memcpy(dst + b * row_bytes, src + indices[b] * row_bytes, (size_t)row_bytes);
    } else {
        for (int64_t t = 0; t < T; ++t)
            for (int64_t b = 0; b < B; ++b)
                memcpy(dst + (t * B + b) * row_bytes, src + (t * N + indices[b]) * row_bytes, (size_t)row_bytes);
    }
}

/* ------------------------------------------------------------------ a7: torch.randperm on CPU
 * cusrl/sampler/mini_batch_sampler.py:56,68 call torch.randperm from the global generator.
 * Third-party algorithm (PyTorch 2.10.0, aten/src/ATen/native/TensorFactories.cpp `randperm_cpu`,
 * "small n" branch, n < UINT32_MAX/20): identity fill, then a forward Fisher-Yates
 *   for i in 0..n-2: z = mt19937() % (n - i); swap(r[i], r[i+z])
 * with the generator being a standard MT19937 seeded by torch.manual_seed(seed) (init_genrand on
 * the low 32 bits).  `state` is 625 uint32 words (624 + position) so consecutive draws continue
 * the stream, like the reference's `out=` re-draws for epoch > 0. */
void oracle_mt19937_seed(uint32_t *state, uint64_t seed) {
    state[0] = (uint32_t)(seed & 0xffffffffu);
    for (int i = 1; i < 624; ++i) state[i] = 1812433253u * (state[i - 1] ^ (state[i - 1] >> 30)) + (uint32_t)i;
    state[624] = 624;
}

static uint32_t mt19937_next(uint32_t *st) {
    if (st[624] >= 624) {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (st[k] & 0x80000000u) | (st[(k + 1) % 624] & 0x7fffffffu);
            st[k] = st[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        st[624] = 0;
    }
    uint32_t y = st[st[624]++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

void oracle_randperm(uint32_t *state, int64_t n, int64_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    for (int64_t i = 0; i + 1 < n; ++i) {
        int64_t z = (int64_t)(mt19937_next(state) % (uint64_t)(n - i));
        int64_t sav = out[i];
        out[i] = out[z + i];
        out[z + i] = sav;
    }
}

/* ------------------------------------------------------------------ a12/a11: Normal log-prob, entropy, KL
 * cusrl/nn/module/distribution.py:195-218 on top of torch.distributions.Normal:
 *   log_prob = -((x - mu)^2) / (2 sigma^2) - log(sigma) - log(sqrt(2 pi))     summed over action dims
 *   entropy  = 0.5 + 0.5 log(2 pi) + log(sigma)                                summed over action dims
 *   kl(p||q) = 0.5 * ((sp/sq)^2 + ((mp - mq)/sq)^2 - 1 - log((sp/sq)^2))      summed over action dims */
void oracle_normal_logp_entropy(const float *action, const float *mean, const float *std, float *logp,
                                float *entropy, int64_t B, int64_t A) {
    const float log_sqrt_2pi = (float)log(sqrt(2.0 * M_PI));
    const float ent_const = (float)(0.5 + 0.5 * log(2.0 * M_PI));
    for (int64_t b = 0; b < B; ++b) {
        double lp = 0.0, en = 0.0;
        for (int64_t a = 0; a < A; ++a) {
            float mu = mean[b * A + a], sg = std[b * A + a], x = action[b * A + a];
            float var = sg * sg, ls = logf(sg), diff = x - mu;
            float term = -(diff * diff) / (2.0f * var) - ls - log_sqrt_2pi;
            lp += (double)term;
            en += (double)(ent_const + ls);
        }
        if (logp) logp[b] = (float)lp;
        if (entropy) entropy[b] = (float)en;
    }
}

void oracle_normal_kl(const float *mean_p, const float *std_p, const float *mean_q, const float *std_q,
                      float *kl, int64_t B, int64_t A) {
    for (int64_t b = 0; b < B; ++b) {
        double acc = 0.0;
        for (int64_t a = 0; a < A; ++a) {
            float r = std_p[b * A + a] / std_q[b * A + a];
            float var_ratio = r * r;
            float t = (mean_p[b * A + a] - mean_q[b * A + a]) / std_q[b * A + a];
            float t1 = t * t;
            acc += (double)(0.5f * (var_ratio + t1 - 1.0f - logf(var_ratio)));
        }
        kl[b] = (float)acc;
    }
}

/* ------------------------------------------------------------------ a9-a13: PPO objective, forward + backward
 * cusrl/hook/on_policy/common.py:29-43      logp, entropy, ratio = exp(logp - old_logp)
 * cusrl/hook/on_policy/ppo.py:10-18,50-55   surrogate = -mean(min(A r, A clamp(r, 1-eps, 1+eps))) * w_sur
 * cusrl/hook/on_policy/value.py:85-89,121-137  mse(return, v) * w_val  or the clipped form
 * cusrl/hook/on_policy/ppo.py:82-84         entropy_loss = -mean(entropy) * w_ent
 * cusrl/template/actor_critic.py:309        loss = ((0 + value_loss) + surrogate_loss) + entropy_loss
 * Gradients are those torch autograd produces for d(loss)/d{mean,std,curr_value}; `min`/`max` ties split
 * the gradient evenly (derivatives.yaml minimum/maximum), clamp passes it on the closed interval.
 * value_clip < 0 means "None" (plain MSE).  losses_out = {value_loss, surrogate_loss, entropy_loss}. */
void oracle_ppo_loss(const float *advantage, const float *old_logp, const float *action, const float *mean,
                     const float *std, const float *ret, const float *curr_value, const float *old_value,
                     int64_t B, int64_t A, int64_t D, double clip, double value_clip, double w_sur,
                     double w_val, double w_ent, float *losses_out, float *logp_out, float *entropy_out,
                     float *ratio_out, float *d_mean, float *d_std, float *d_value) {
    const float lo = (float)(1.0 - clip), hi = (float)(1.0 + clip);
    const float log_sqrt_2pi = (float)log(sqrt(2.0 * M_PI));
    const float ent_const = (float)(0.5 + 0.5 * log(2.0 * M_PI));
    double sur_acc = 0.0, ent_acc = 0.0, val_acc = 0.0;
    const float g_sur = (float)(-w_sur / (double)B);      /* d loss / d min(...)_b */
    const float g_ent = (float)(-w_ent / (double)B);      /* d loss / d entropy_b  */
    const float g_val = (float)(w_val / (double)(B * D)); /* d loss / d sq_err_bd  */
    for (int64_t b = 0; b < B; ++b) {
        double lp = 0.0, en = 0.0;
        for (int64_t a = 0; a < A; ++a) {
            float mu = mean[b * A + a], sg = std[b * A + a], x = action[b * A + a];
            float diff = x - mu, ls = logf(sg);
            lp += (double)(-(diff * diff) / (2.0f * (sg * sg)) - ls - log_sqrt_2pi);
            en += (double)(ent_const + ls);
        }
        float logp = (float)lp, entropy = (float)en;
        float ratio = expf(logp - old_logp[b]);
        float adv = advantage[b];
        float s1 = adv * ratio;
        float rc = ratio < lo ? lo : (ratio > hi ? hi : ratio);
        float s2 = adv * rc;
        sur_acc += (double)(s1 < s2 ? s1 : s2);
        ent_acc += (double)entropy;
        /* d min / d ratio */
        int inside = (ratio >= lo) && (ratio <= hi);
        float d_ratio;
        if (s1 < s2) d_ratio = adv;
        else if (s1 > s2) d_ratio = inside ? adv : 0.0f;
        else d_ratio = 0.5f * adv + (inside ? 0.5f * adv : 0.0f);
        float d_logp = g_sur * d_ratio * ratio;
        if (logp_out) logp_out[b] = logp;
        if (entropy_out) entropy_out[b] = entropy;
        if (ratio_out) ratio_out[b] = ratio;
        for (int64_t a = 0; a < A; ++a) {
            float mu = mean[b * A + a], sg = std[b * A + a], x = action[b * A + a];
            float diff = x - mu, var = sg * sg;
            if (d_mean) d_mean[b * A + a] = d_logp * (diff / var);
            if (d_std) d_std[b * A + a] = d_logp * ((diff * diff) / (var * sg) - 1.0f / sg) + g_ent / sg;
        }
        for (int64_t d = 0; d < D; ++d) {
            float cv = curr_value[b * D + d], R = ret[b * D + d];
            float e1 = cv - R, l1 = e1 * e1, g1 = 2.0f * e1;
            if (value_clip < 0) {
                /* mse_loss(return, curr_value): (R - cv)^2, same value and gradient */
                val_acc += (double)l1;
                if (d_value) d_value[b * D + d] = g_val * g1;
            } else {
                float c = (float)value_clip, v = old_value[b * D + d];
                float dv = cv - v;
                float dvc = dv < -c ? -c : (dv > c ? c : dv);
                float e2 = (v + dvc) - R, l2 = e2 * e2;
                float g2 = (dv >= -c && dv <= c) ? 2.0f * e2 : 0.0f;
                val_acc += (double)(l1 > l2 ? l1 : l2);
                float g = l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * (g1 + g2));
                if (d_value) d_value[b * D + d] = g_val * g;
            }
        }
    }
    losses_out[0] = (float)(val_acc / (double)(B * D)) * (float)w_val;
    losses_out[1] = -(float)(sur_acc / (double)B) * (float)w_sur;
    losses_out[2] = -(float)(ent_acc / (double)B) * (float)w_ent;
}
