/*
 * cusrl_hip.h — C ABI of libcusrl_hip.so: the MI355X (gfx950) rollout + PPO-update hot path.
 *
 * Drop-in boundary.  The reference (chengruiz/cusrl) has no FFI: its hot path is
 * chains of PyTorch ops behind a Python plugin API (Buffer / Sampler / Hook).  Each entry point below
 * replaces the torch-op chain of ONE reference function, cited as file:line relative to the
 * reference repo root; the Python host (cusrl_amd/) keeps the reference's class and method names
 * and binds these symbols with ctypes (INTEGRATION.md shows the binding a cusrl maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked "host";
 *   - the caller owns all memory (no allocation, no hidden synchronisation, no host<->device copies);
 *   - every launch goes to `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative CUSRL_E_* code;
 *   - tensors are contiguous, row-major, fp32 unless stated; flags are 1-byte bools;
 *   - layouts follow the reference Buffer: a leaf is [T, N, C] (capacity, parallelism, channels),
 *     a "slot" is one (t, n) pair, its flat index is t * N + n (buffer.py:124-151).
 */
#ifndef CUSRL_HIP_H
#define CUSRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUSRL_ABI_VERSION 6
#define CUSRL_MAX_FIELDS 24 /* leaves per push / gather launch; larger tables are split by the host */
#define CUSRL_MAX_PACKED 16 /* 1-8 byte entries of the per-slot record (cusrl_pack_rows); wide fields count as leaves */
#define CUSRL_MAX_RECORD_BYTES 1024

#define CUSRL_E_INVALID (-1)     /* NULL pointer, negative size, inconsistent arguments */
#define CUSRL_E_TOO_MANY (-2)    /* n_fields > CUSRL_MAX_FIELDS */
#define CUSRL_E_UNSUPPORTED (-3) /* shape outside what the kernels handle */
#define CUSRL_E_COMM (-4)        /* RCCL unavailable or an RCCL call failed: see cusrl_comm_last_error() */

/* One leaf of a multi-leaf copy.  `row_bytes` = bytes of one slot (C * element size). */
typedef struct {
    const void *src;
    void *dst;
    int64_t row_bytes;
} cusrl_field_t;

int cusrl_abi_version(void);
/* Human-readable text for a return code (host string, static storage). */
const char *cusrl_error_string(int code);

/* Launch-shape and cache-policy overrides (ABI 6).  Every kernel chooses its launch shape and cache policy by a measured rule
 * of its own (footprint against the 256 MB Infinity Cache, rows per block); these exist for A/B measurements and sweeps.  Up
 * to ABI 5 the library read them from CUSRL_* environment variables inside the launch entry points — invisible to a caller
 * binding this header; now they are part of it, and the only environment variable the library itself reads is
 * CUSRL_RCCL_LIBRARY (where to find RCCL, cusrl_comm_*).  value 0 = back to the kernel's own rule.
 *   "gae_policy"    1 + {0, 5, 7}: cache policy of cusrl_gae's scan (bit 0 non-temporal loads, bit 1 `advantage`, bit 2 `return`)
 *   "gae_block"     128 | 256 threads per block of the scan
 *   "loss_policy"   1 default policy | 2 non-temporal [B, A] streams in cusrl_ppo_loss_fwd_bwd
 *   "push_policy"   1 default policy | 2 streaming in cusrl_buffer_push*
 *   "colsum_rows"   rows per block of cusrl_relu_bwd_colsum (4 .. 4096)
 *   "head_rows"     rows per block of cusrl_narrow_linear_bwd (8 .. 4096)
 *   "gru_bias_rows" rows per partial row of cusrl_gru_gates_bwd_bias (4 | 8 | 16 | 32)
 * Returns 0, CUSRL_E_INVALID for an unknown key, CUSRL_E_UNSUPPORTED for a value outside the list.  Host-side, thread-safe
 * (relaxed atomics); takes effect at the next launch.  Every setting changes how bytes move, never a result bit
 * (tests/test_hip_kernels.py: *_every_cache_policy_*, *_cache_policies_change_no_bit). */
int cusrl_set_option(const char *key, int64_t value);
int cusrl_get_option(const char *key, int64_t *value_out);

/* Node census of a captured hipGraph (`graph` = hipGraph_t): a host-side walk, no launch, no stream.  The reference's
 * `compile=True` (cusrl/template/actor_critic.py:217-220) hands its loops to torch.compile; here they are hipGraphs, and
 * what a captured step is made of is checkable: type_counts[k] (HOST, n_types entries) = number of nodes of
 * hipGraphNodeType k (0 kernel, 1 memcpy, 2 memset, ...); `names` (HOST, `capacity` bytes, may be NULL with capacity 0)
 * receives the mangled names of the kernel nodes in node order, '\n'-terminated, truncated at capacity;
 * *names_len = bytes the full list needs.  The host (template/graphs.py `_Capture.capture`) walks every captured region with
 * this before it instantiates it and rewrites the region's memset nodes (cusrl_graph_replace_memsets below: they do not
 * replay reliably on this stack); ATen global-reduce kernels are allowed (their semaphore memsets are what gets rewritten),
 * the tests assert that the stock compositions contain none. */
int cusrl_graph_census(void *graph, int64_t *type_counts, int n_types, char *names, int64_t capacity, int64_t *names_len);

/* Replace every memset node of a captured, not yet instantiated hipGraph by a kernel node (a plain fill kernel) with the
 * same destination, value, extent and edges; *replaced_out (HOST, may be NULL) = how many.  Host-side graph surgery, no
 * launch.  What torch.compile does for the reference's `compile=True` (cusrl/template/hook.py:396-399) is a hipGraph here,
 * and the ROCm 7.0 runtime of PyTorch 2.10 does not replay memset nodes reliably (scripts/probe_aten_reduce_capture.py:
 * ATen's split reductions zero their semaphores with hipMemsetAsync, Reduce.cuh:1294-1301) — a captured region is only
 * instantiated once it has none.  CUSRL_E_UNSUPPORTED: a memset node with an element size other than 1 / 2 / 4. */
int cusrl_graph_replace_memsets(void *graph, int64_t *replaced_out);

/* ---- a1  Buffer.push — cusrl/template/buffer.py:124-151 (`storage[cursor] = value` per leaf) ----
 * fields[i].src = step leaf [N, row_bytes];  fields[i].dst = storage leaf base [T, N, row_bytes].
 * Copies every leaf's step into slot row `cursor` in ONE launch.  `fields` is a HOST array. */
int cusrl_buffer_push(const cusrl_field_t *fields, int n_fields, int64_t cursor, int64_t N, void *stream);

/* The same append with WRITE-THROUGH into the per-slot record of cusrl_pack_rows / cusrl_gather_rows_packed (below):
 * record_offset[i] >= 0 (n_fields HOST entries): leaf i also lives in the record at that byte offset, and its step row n
 * is additionally stored at record[(cursor * N + n) * record_bytes + record_offset[i]] from the registers that already
 * hold it; -1: the leaf is not in the record.  Only whole 16-byte chunks are written through (row_bytes and the offset
 * multiples of 16, 16-byte aligned pointers: the wide leaves — observation, action); narrow leaves are produced or
 * edited at update time anyway and keep going through cusrl_pack_rows.  With this the once-per-update pack of the
 * `ppo` record moves 13 instead of 253 bytes per slot. */
int cusrl_buffer_push_through(const cusrl_field_t *fields, int n_fields, int64_t cursor, int64_t N, void *record,
                              int64_t record_bytes, const int32_t *record_offset, void *stream);

/* ---- a3  ValueComputation.pre_update — cusrl/hook/on_policy/value.py:56-82 ----
 * next_value[:-1] = value[1:]; next_value[-1] = last_value [N,D]; next_value[terminated] = termination_value;
 * truncated slots: mode 0 = left for the caller to bootstrap (value.py:72-78), mode 1 = value[truncated]
 * (value.py:79-80).  Also counts truncated slots per block into `block_counts`
 * (cusrl_flag_blocks(T*N) int32 entries) for cusrl_compact_flags. */
int cusrl_next_value(const float *value, const uint8_t *terminated, const uint8_t *truncated,
                     const float *last_value, float termination_value, int truncated_mode, float *next_value,
                     int32_t *block_counts, int64_t T, int64_t N, int64_t D, void *stream);

/* Number of per-block counters cusrl_next_value / cusrl_compact_flags use for `n` flags. */
int64_t cusrl_flag_blocks(int64_t n);

/* Ordered stream compaction of a flag array (replaces `next_state[truncated]` boolean-mask indexing,
 * value.py:75): indices_out[0..count) = ascending flat slots with flags != 0, *count_out = count.
 * `block_counts` must have been filled by cusrl_next_value over the same flags, or pass recount != 0
 * to have this call count first.  `count_out` is written with a system-scope store and may point to pinned host
 * memory (device-mapped): the host can then poll it instead of a device->host copy + stream synchronisation (the
 * trainer's per-step read of the finished-env count, trainer.py:365-372). */
int cusrl_compact_flags(const uint8_t *flags, int64_t n, int32_t *block_counts, int recount,
                        int64_t *indices_out, int32_t *count_out, void *stream);

/* dst[indices[k], :] = src[k, :] for k < K (row_bytes per row): `next_value[truncated] = critic(...)`,
 * value.py:78.  If count_dev != NULL the effective K is min(K, *count_dev) read on the device. */
int cusrl_scatter_rows(const void *src, const int64_t *indices, void *dst, int64_t K, int64_t row_bytes,
                       const int32_t *count_dev, void *stream);

/* ---- the next act input of a vectorised rollout — cusrl/template/environment.py:365-379 (`update_observation_and_state`:
 * `last_observation[indices] = init_observation`) fused with the copy into the act step's input buffer ----
 * dst[n] = src[n] for every env n with done[n] == 0, and dst[indices[k]] = init[k] for k < min(N, *count_dev): `done`
 * [N] and (indices, count_dev) must describe the same set of finished envs (cusrl_step_epilogue emits both), src / init
 * / dst are [N, row_bytes] with dst distinct from src and init; the count is read by the kernel (device or pinned host
 * memory), so a captured env step needs no host read to reset its finished envs. */
int cusrl_splice_rows(const void *src, const void *init, const int64_t *indices, const int32_t *count_dev,
                      const uint8_t *done, void *dst, int64_t N, int64_t row_bytes, void *stream);

/* ---- a4  GAE(lambda) + return — cusrl/hook/on_policy/gae.py:8-20, 85-110 ----
 * delta = (r + nv*gamma) - v;  A[T-1] = delta;  A[t] = delta[t] + ((done[t] ? 0 : gamma*lamda) * A[t+1])
 * (separate multiply and add, bit-exact with the reference);  ret = value + A, or value + A' where A' is the
 * same scan with lamda_value when lamda_value >= 0 (pass < 0 for None).  done is [T,N,1].
 * stat_partials (optional): [cusrl_gae_num_partials(T,N,D)] rows of {sum, sumsq} per channel, i.e.
 * double[num_partials][D][2], of the advantage — consumed by cusrl_stats_finalize (fuses the statistics
 * pass of advantage.py:111 into the scan). */
int cusrl_gae(const float *reward, const float *value, const float *next_value, const uint8_t *done,
              float *advantage, float *ret, double *stat_partials, int64_t T, int64_t N, int64_t D,
              double gamma, double lamda, double lamda_value, void *stream);
int64_t cusrl_gae_num_partials(int64_t T, int64_t N, int64_t D);

/* ---- a5  AdvantageNormalization.normalize_ — cusrl/hook/on_policy/advantage.py:108-115 ----
 * Column statistics of x [rows, D] as per-block {sum, sumsq} partials (double[num_partials][D][2]). */
int cusrl_col_stats(const float *x, int64_t rows, int64_t D, double *stat_partials, void *stream);
int64_t cusrl_col_stats_num_partials(int64_t rows, int64_t D);
/* mean[d], var[d] (unbiased, correction = 1, like torch.var_mean) from partials, fixed summation order.  mean and var may be the
 * halves of ONE float[2 D] row (var = mean + D): what a cross-rank merge all-gathers (cusrl_normalize_from_gathered). */
int cusrl_stats_finalize(const double *stat_partials, int64_t num_partials, int64_t D, int64_t count,
                         float *mean, float *var, void *stream);
/* x = (x - mean) / sqrt(var + eps) in place, true division (advantage.py:114-115). */
int cusrl_normalize(float *x, const float *mean, const float *var, float eps, int64_t rows, int64_t D,
                    void *stream);

/* The two steps above in ONE launch for the single-process case (no cross-rank merge between statistics and their
 * use): every block reduces the partial rows itself (same fixed order), then normalises its share of x; mean_out /
 * var_out [D] receive the statistics (block 0).  Bit-identical to cusrl_stats_finalize + cusrl_normalize. */
int cusrl_normalize_from_partials(float *x, const double *stat_partials, int64_t num_partials, int64_t count, float eps,
                                  int64_t rows, int64_t D, float *mean_out, float *var_out, void *stream);

/* ---- a6  reduce_mean_var_ merge — cusrl/utils/distributed.py:175-183 ----
 * gathered = [W, 2*D] rows of cat(mean_r, var_r) (the all_gather result); writes the equal-weight merge
 * mean = avg_r mean_r, var = avg_r (var_r + (mean_r - mean)^2) into mean[D], var[D]. */
int cusrl_merge_mean_var(const float *gathered, int64_t W, int64_t D, float *mean, float *var, void *stream);

/* cusrl_merge_mean_var + cusrl_normalize as ONE launch (ABI 6, second part of round 6): the advantage normalisation of a job with
 * several ranks (hook/on_policy/advantage.py:108-115 with `reduce_mean_var_`, distributed.py:175-183, between the statistics and
 * their use).  gathered: float[W][2 D], every rank's mean | var (cusrl_stats_finalize writes both into one [2 D] row, which
 * cusrl_allgather collects); every block re-derives the merged statistics — the same operations in the same order as
 * cusrl_merge_mean_var — and normalises its share of x [rows, D] in place; mean_out / var_out [D] receive the merged statistics.
 * Bit-identical to cusrl_merge_mean_var + cusrl_normalize.  D <= 256. */
int cusrl_normalize_from_gathered(float *x, const float *gathered, int64_t W, float eps, int64_t rows, int64_t D,
                                  float *mean_out, float *var_out, void *stream);

/* ---- a7/a8  minibatch gather — cusrl/sampler/mini_batch_sampler.py:87-89, 113-114; buffer.py:153-162 ----
 * For every leaf i: temporal == 0:  dst_i[b]       = src_i.flatten(0,1)[indices[b]]      b < B
 *                   temporal != 0:  dst_i[t, b]    = src_i[t, indices[b]]                t < T, b < B
 * All leaves in ONE launch; `fields` is a HOST array; indices are int64 (torch.randperm's dtype). */
int cusrl_gather_rows(const cusrl_field_t *fields, int n_fields, const int64_t *indices, int64_t B, int64_t T,
                      int64_t N, int temporal, void *stream);

/* ---- a7/a8 with the narrow leaves packed — same reference lines, different byte traffic ----
 * A minibatch row of a 1-8 byte leaf (action_logp, value, reward, next_value, advantage, return, the three flags of
 * the `ppo` buffer) costs one memory sector per leaf when fetched from a random slot.  cusrl_pack_rows interleaves
 * such leaves ONCE per update into one record per slot — record[s] = { field_0[s], field_1[s], ... } at the given
 * byte offsets, record_bytes a multiple of 16 up to CUSRL_MAX_RECORD_BYTES, `record` 16-byte aligned — and
 * cusrl_gather_rows_packed gathers the plain leaves as cusrl_gather_rows does plus, for every sampled slot, ONE record
 * read fanned out to the separate contiguous batch tensors (dst_k[b] = field_k of record[slot(b)]): identical results.
 * Wide leaves (a multiple of 16 bytes per slot: observation, action) may live in the record as well: when the record
 * holds exactly what a training step reads (253 B for the `ppo` preset -> 256 B, two 128-byte memory lines), a sampled
 * slot costs 256 fetched bytes instead of ~550 (measured 128 B per random row of ANY size <= 128 B on MI355X:
 * profiles/r02/pmc_summary.json).  The host keeps the record valid (rebuilds it after any write to a packed leaf). */
typedef struct {
    void *ptr;      /* pack: source leaf base [rows, width] (read only); gather: destination base [B or T*B, width] */
    int32_t offset; /* byte offset of the field inside the record; a multiple of min(width, 16) */
    int32_t width;  /* 1, 2, 4, 8 bytes, or any multiple of 16 (wide field) */
} cusrl_packed_field_t;
int cusrl_pack_rows(const cusrl_packed_field_t *fields, int n_fields, void *record, int64_t record_bytes,
                    int64_t rows, void *stream);
/* cusrl_pack_rows where the caller owns `owned_chunks` (1 or 2) whole 16-byte chunks of every record, starting at chunk
 * `owned_first_chunk`: every narrow field of the call lies inside them and nothing else lives there (no other leaf of
 * the record, no wide field).  The narrow fields then leave as ONE 16-byte store per chunk instead of one 1-4 byte store
 * per field (bytes of the owned chunks no field covers become zero).  The update-time leaves of the `ppo` hot record
 * (action_logp, advantage, return, done: 13 bytes in the record's last chunk) are the case it is there for. */
int cusrl_pack_rows_owned(const cusrl_packed_field_t *fields, int n_fields, void *record, int64_t record_bytes,
                          int64_t rows, int32_t owned_first_chunk, int32_t owned_chunks, void *stream);
int cusrl_gather_rows_packed(const cusrl_field_t *fields, int n_fields, const void *record, int64_t record_bytes,
                             const cusrl_packed_field_t *packed, int n_packed, const int64_t *indices, int64_t B,
                             int64_t T, int64_t N, int temporal, void *stream);

/* ---- random temporal windows — cusrl/sampler/random_sampler.py:95-113 (`TemporalRandomSampler`) ----
 * out[t * B + b] = ((cursor + start[b] + t) % T) * N + env[b] for t < L, b < B: the flat slots of B windows of L
 * consecutive steps, window b belonging to env[b] and starting at LOGICAL step start[b] (physical row `cursor` is
 * logical step 0 of a full ring; pass cursor = 0 while the ring is still filling).  The list feeds cusrl_gather_rows
 * (temporal = 0, B' = L * B) — `data[time_indices, env_indices]` of the reference for every leaf in one launch. */
int cusrl_window_indices(const int64_t *start, const int64_t *env, int64_t *out, int64_t B, int64_t L, int64_t T,
                         int64_t N, int64_t cursor, void *stream);

/* ---- a9-a13  PPO objective, forward + backward ----
 * common.py:29-43 (Normal log-prob / entropy / ratio), ppo.py:10-18,50-55 (clipped surrogate),
 * value.py:85-89,121-137 (MSE or clipped value loss), ppo.py:82-84 (entropy bonus),
 * actor_critic.py:309 (sum).  Shapes: advantage, old_logp [B,1]; action, mean, std [B,A];
 * ret, curr_value, old_value [B,D] (old_value may be NULL when value_clip < 0 = None).
 * Outputs: losses_out[7] = {value_loss, surrogate_loss, entropy_loss} (already weighted), the three per-minibatch
 * metrics the hooks record — mean |logp ratio|, mean entropy, mean curr_value.sum(-1) — and the total loss
 * (value + surrogate) + entropy that the agent differentiates (actor_critic.py:309);
 * logp_out, entropy_out, logp_ratio_out, ratio_out [B] (each optional);
 * d_mean, d_std [B,A], d_value [B,D] = d(value_loss + surrogate_loss + entropy_loss)/d(.) .
 * partials: double[cusrl_ppo_loss_num_partials(B)][5] workspace.
 * std_rows = B: `std` is the [B,A] matrix of the reference's repeated std vector (distribution.py:228-247);
 * std_rows = 1: `std` is that vector itself, [A] — it is broadcast inside the kernel and d_std is the gradient of the
 * vector, [A] (= the column sums a sum(0) over [B,A] would give); needs A % 4 == 0, A <= 32 and the workspace
 * d_std_partials: float[cusrl_ppo_loss_std_partial_rows(B)][A] (may be NULL otherwise).
 * flags: 0, or CUSRL_LOSS_DEFER — ONE launch, no finalize: nothing inside an optimizer step consumes the loss VALUES
 * (actor_critic.py:311-312 differentiates them with a unit gradient; the values only feed the metrics read once per
 * update), so the caller may own the reductions instead of paying a one-block launch per minibatch step:
 *   - every block ADDS its five partial sums {sum sq. value error, sum min(..) surrogate, sum entropy, sum |logp ratio|,
 *     sum value} to ITS row of `partials` (cusrl_ppo_loss_blocks(B, A) rows; zero-fill once): launches of the same
 *     shape on one stream build running sums in a fixed order, the caller forms the means whenever it reads them;
 *   - with a std vector every block leaves its [A] column sums of d_std in its row of d_std_partials — exactly the
 *     "partial rows" form cusrl_assemble_gradients reduces into the parameter's slot (row_stride = A);
 *   - losses_out and (std vector) d_std are not written and may be NULL. */
#define CUSRL_LOSS_DEFER 1
int cusrl_ppo_loss_fwd_bwd(const float *advantage, const float *old_logp, const float *action, const float *mean,
                           const float *std, const float *ret, const float *curr_value, const float *old_value,
                           int64_t B, int64_t A, int64_t D, double clip, double value_clip, double w_sur,
                           double w_val, double w_ent, float *losses_out, float *logp_out, float *entropy_out,
                           float *logp_ratio_out, float *ratio_out, float *d_mean, float *d_std, float *d_value,
                           double *partials, int64_t std_rows, float *d_std_partials, int flags, void *stream);
/* workspace rows (enough for any action width) and the exact number of blocks = partial rows of one launch
 * (A = 0: the categorical form) */
int64_t cusrl_ppo_loss_num_partials(int64_t B);
int64_t cusrl_ppo_loss_std_partial_rows(int64_t B);
int64_t cusrl_ppo_loss_blocks(int64_t B, int64_t A);

/* D == 0 in cusrl_ppo_loss_fwd_bwd / cusrl_ppo_loss_categorical_fwd_bwd (ABI 6): the launch carries NO value term — ret,
 * curr_value, old_value, d_value may be NULL, losses_out[0] = losses_out[5] = 0 and losses_out[6] = surrogate + entropy.
 * The value term then comes from the launch below.
 *
 * ValueLoss.objective on its own — cusrl/hook/on_policy/value.py:121-137 (`mse_loss(return, curr_value) * weight`) and the
 * clipped form `_clipped_value_loss` value.py:85-89 — forward AND backward in one pass over ret / curr_value / old_value
 * [B, D]: d_value [B, D] = d(value_loss)/d(curr_value) (optional), losses_out[0] = the weighted loss, losses_out[1] = mean of
 * curr_value.sum(-1) (the `value` metric, value.py:139-141).  The sum `value + surrogate + entropy` the agent differentiates
 * with a unit gradient (cusrl/template/actor_critic.py:309-312) splits into its summands; the host evaluates this one on the
 * stream the critic runs on, so that critic forward -> value term -> critic backward is one branch of the captured
 * minibatch step.  Same per-element arithmetic as the one-launch objective (d_value bit-identical).
 * partials: double[cusrl_value_loss_blocks(B, D)][2]; flags: 0 or CUSRL_LOSS_DEFER (every block ADDS {sum sq. error, sum value}
 * to its row; losses_out may be NULL). */
int cusrl_value_loss_fwd_bwd(const float *ret, const float *curr_value, const float *old_value, int64_t B, int64_t D,
                             double value_clip, double w_val, float *losses_out, float *d_value, double *partials, int flags,
                             void *stream);
int64_t cusrl_value_loss_blocks(int64_t B, int64_t D);

/* The same objective for one-hot categorical policies (discrete action spaces) —
 * cusrl/nn/module/distribution.py:332-366 over torch.distributions.OneHotCategorical: action [B,A] one-hot (the taken
 * action is its first arg-max), logits [B,A] unnormalised; logp = log_softmax(logits)[taken],
 * entropy = -sum_j p_j log p_j, ratio / surrogate / value / entropy terms and losses_out[7] exactly as above;
 * d_logits [B,A], d_value [B,D] = d(total loss)/d(.).  partials: double[cusrl_ppo_loss_num_partials(B)][5];
 * flags as above (CUSRL_LOSS_DEFER: block rows accumulate into `partials`, cusrl_ppo_loss_blocks(B, 0) of them;
 * a masked action's logit may be -inf: p = 0 and, as in torch.distributions.Categorical.entropy, its log p is clamped to
 * the smallest finite float before the product, so every output of the row stays finite and its d_logits is 0). */
int cusrl_ppo_loss_categorical_fwd_bwd(const float *advantage, const float *old_logp, const float *action,
                                       const float *logits, const float *ret, const float *curr_value,
                                       const float *old_value, int64_t B, int64_t A, int64_t D, double clip,
                                       double value_clip, double w_sur, double w_val, double w_ent, float *losses_out,
                                       float *logp_out, float *entropy_out, float *logp_ratio_out, float *ratio_out,
                                       float *d_logits, float *d_value, double *partials, int flags, void *stream);

/* ---- §8f row 1: GRU time step (torch.nn.GRU, the recurrent backbone of cusrl/nn/module/rnn.py:21-120) ----
 * One pass over the gates of one time step, between the rocBLAS GEMMs that produce gi = W_ih x + b_ih [B, 3H] (all steps
 * at once) and gh = W_hh h [B, 3H] (per step):
 *   r = sigmoid(gi_r + gh_r + b_hr); z = sigmoid(gi_z + gh_z + b_hz); n = tanh(gi_n + r * (gh_n + b_hn));
 *   h <- (1 - z) * n + z * h (in place); out = h.           b_hh may be NULL (bias=False).
 * lengths (int64[B], may be NULL): a sequence with t >= lengths[b] keeps h, emits out = 0 (the semantics of a
 * PackedSequence, cusrl/nn/module/rnn.py:273-291, without packing or a host read of the lengths).
 * cusrl_gru_gates_bwd consumes the saved pre-activations IN PLACE: gi <- dL/dgi, gh <- dL/dgh, and
 * dh <- (dh + d_out) * z, the direct path to h_{t-1} (the caller adds dL/dgh @ W_hh); d_out may be NULL; ended
 * sequences get zero gate gradients and leave dh untouched. */
int cusrl_gru_gates_fwd(const float *gi, const float *gh, const float *b_hh, float *h, float *out,
                        const int64_t *lengths, int64_t t, int64_t B, int64_t H, void *stream);
int cusrl_gru_gates_bwd(float *gi, float *gh, const float *b_hh, const float *h_prev, const float *d_out, float *dh,
                        const int64_t *lengths, int64_t t, int64_t B, int64_t H, void *stream);
/* cusrl_gru_gates_bwd with the bias gradients folded in: every block of 16 rows also leaves the column sums of what it
 * wrote, bias_partials[block][4H] = {sum d_r, sum d_z, sum d_n, sum d_q} (block = row / 16, cusrl_gru_bias_partial_rows(B)
 * rows); summed over all blocks and steps, d b_ih = {r, z, n} and d b_hh = {r, z, q} — instead of two column-sum passes
 * over the [L * B, 3H] gradient arrays.  Needs H / 4 (H for unaligned rows) to divide 256; CUSRL_E_UNSUPPORTED otherwise. */
int cusrl_gru_gates_bwd_bias(float *gi, float *gh, const float *b_hh, const float *h_prev, const float *d_out, float *dh,
                             const int64_t *lengths, int64_t t, int64_t B, int64_t H, float *bias_partials, void *stream);
int64_t cusrl_gru_bias_partial_rows(int64_t B);
/* 1 when cusrl_gru_gates_bwd_bias accepts a launch with this hidden size and these (device) pointers — alignment and the
 * column-chunk tiling decide — 0 when it would return CUSRL_E_UNSUPPORTED: the host asks instead of restating the rule. */
int cusrl_gru_bias_supported(int64_t H, const float *gi, const float *gh, const float *b_hh, const float *h_prev,
                             const float *d_out, const float *dh, const float *bias_partials);

/* The same for torch.nn.LSTM (gate order i, f, g, o), the default core of RecurrentPpoAgentFactory (cusrl/preset/ppo.py:189):
 *   pre = gi + gh + b_hh; c <- sigmoid(pre_f) * c + sigmoid(pre_i) * tanh(pre_g); h <- sigmoid(pre_o) * tanh(c); out = h.
 * With c_saved != NULL (training) the new cell state is also stored there and `pre` is written over gi — both biases are
 * additive, so d(pre) is the gradient of gi and of gh alike.  cusrl_lstm_gates_bwd: pre <- dL/dpre in place,
 * dc <- dL/dc_{t-1}, dh <- the part of the state gradient that bypasses the step (everything for an ended sequence,
 * zero otherwise; the caller adds dL/dpre @ W_hh).  c_prev / c_next: cell state before / after the step. */
int cusrl_lstm_gates_fwd(float *gi, const float *gh, const float *b_hh, float *h, float *c, float *out, float *c_saved,
                         const int64_t *lengths, int64_t t, int64_t B, int64_t H, void *stream);
int cusrl_lstm_gates_bwd(float *pre, const float *c_prev, const float *c_next, const float *d_out, float *dh, float *dc,
                         const int64_t *lengths, int64_t t, int64_t B, int64_t H, void *stream);

/* ... and for torch.nn.RNN (relu != 0: ReLU, else tanh): h <- act(gi + gh + b_hh), out = h.  The output is the only
 * saved state (act' is a function of it); cusrl_rnn_cell_bwd writes dL/dpre = (dh + d_out) * act'(out) into d_pre (the
 * caller passes the gi slice) and leaves in dh what bypasses the step, as the LSTM form does. */
int cusrl_rnn_cell_fwd(const float *gi, const float *gh, const float *b_hh, float *h, float *out, const int64_t *lengths,
                       int64_t t, int64_t B, int64_t H, int relu, void *stream);
int cusrl_rnn_cell_bwd(float *d_pre, const float *out, const float *d_out, float *dh, const int64_t *lengths, int64_t t,
                       int64_t B, int64_t H, int relu, void *stream);

/* ---- rollout-side: sampling and episode statistics ----
 * Normal sample + log-prob of the sample in one pass — cusrl/nn/module/distribution.py:198-205 (`rsample`, then
 * `log_prob(sample).sum(-1, keepdim)`): action = mean + eps * std with eps ~ N(0,1) supplied by the caller (drawn
 * from torch's generator so the random stream is the reference's); logp[B] as in cusrl_ppo_loss_fwd_bwd.
 * std_rows = B: std [B,A]; std_rows = 1: std is the [A] vector a state-independent std repeats for every row
 * (distribution.py:228-247, `param.repeat(B, 1)`) — broadcast inside the kernel, and std_out [B,A] (optional) receives
 * the repeated matrix the rollout buffer stores as a leaf, so the acting path needs no `repeat` launch.
 * mean_bias [A] + mean_out [B,A] (both or neither): `mean` is the policy head's product WITHOUT its bias
 * (`latent @ W^T`, distribution.py:195-197 `mean_head`); the bias is added here and the finished mean written to
 * mean_out — the library adds a 12-column bias by broadcasting it into the output with a copy launch before the GEMM. */
int cusrl_normal_sample_logp(const float *mean, const float *std, const float *eps, float *action, float *logp,
                             int64_t B, int64_t A, int64_t std_rows, float *std_out, const float *mean_bias,
                             float *mean_out, void *stream);

/* One-hot categorical sample + its log-prob in one pass — cusrl/nn/module/distribution.py:332-366
 * (`OneHotCategorical(logits).sample()`, `log_prob(sample)`): idx = argmax_j softmax(logits)_j / noise_j with
 * noise ~ Exp(1) supplied by the caller from torch's generator (the way torch.multinomial draws one sample on the
 * device); action[B, A] = one_hot(idx), logp[B] = logits[idx] - logsumexp(logits).  Ties go to the lower index. */
int cusrl_categorical_sample_logp(const float *logits, const float *noise, float *action, float *logp, int64_t B,
                                  int64_t A, void *stream);

/* EnvironmentStats.track_step + track_episode — cusrl/template/trainer.py:54-76, in one launch and without the
 * host round trip of `get_done_indices(...).tolist()` (environment.py:356-362):
 *   episode_rew[n] += reward[n]; episode_len[n] += 1; step_reward_sum[d] += sum_n reward[n,d];
 *   for done[n], in ASCENDING n: slot = (num_episodes++) % R; ring_rew[slot] = episode_rew[n];
 *                ring_len[slot] = episode_len[n]; episode_rew[n] = 0; episode_len[n] = 0.
 * reward, episode_rew [N,D]; done [N] bytes; episode_len [N]; ring_rew [R,D]; ring_len [R];
 * num_episodes: device uint64[2], double-buffered — the launch reads num_episodes[parity] and leaves the updated count
 * in num_episodes[parity ^ 1]; the caller flips `parity` (0 / 1) every call; step_reward_sum: device double[D].
 * (Beyond 262 144 envs slots are drawn by atomic ticket: same set of slots, unspecified order within the step.) */
int cusrl_episode_stats(const float *reward, const uint8_t *done, float *episode_rew, float *episode_len,
                        float *ring_rew, float *ring_len, uint64_t *num_episodes, double *step_reward_sum,
                        int64_t N, int64_t D, int64_t R, int parity, void *stream);

/* The whole per-env-step epilogue of the rollout loop in ONE launch (cusrl/template/actor_critic.py:277,
 * cusrl/template/trainer.py:300-313, cusrl/template/environment.py:356-362):
 *   done_out[n] = terminated[n] | truncated[n];   episode statistics exactly as cusrl_episode_stats with that flag;
 *   indices_out[0 .. count) = the finished envs in ascending order (`get_done_indices`), *count_out = their number.
 * count_out is written with a system-scope store and may be pinned host memory: the host launches this right after
 * env.step, runs agent.step (hooks, buffer push) meanwhile, and finds the count waiting instead of blocking on a
 * device->host copy.  Needs N <= cusrl_step_epilogue_max_envs(). */
int cusrl_step_epilogue(const float *reward, const uint8_t *terminated, const uint8_t *truncated, uint8_t *done_out,
                        float *episode_rew, float *episode_len, float *ring_rew, float *ring_len, uint64_t *num_episodes,
                        double *step_reward_sum, int64_t *indices_out, int32_t *count_out, int64_t N, int64_t D, int64_t R,
                        int parity, void *stream);
int64_t cusrl_step_epilogue_max_envs(void);
/* cusrl_step_epilogue AND cusrl_buffer_push of the same env step as ONE launch (actor_critic.py:273-282 + trainer.py:296-321):
 * the epilogue blocks store `done` both to done_out and straight into the buffer's slab of fields[done_field] (whose src
 * is ignored); every other field is appended as by cusrl_buffer_push.  For steps whose post_step hooks do no device
 * work between the two (nothing may read the flag or edit the reward in between). */
int cusrl_step_epilogue_push(const float *reward, const uint8_t *terminated, const uint8_t *truncated, uint8_t *done_out,
                             float *episode_rew, float *episode_len, float *ring_rew, float *ring_len, uint64_t *num_episodes,
                             double *step_reward_sum, int64_t *indices_out, int32_t *count_out, int64_t N, int64_t D, int64_t R,
                             int parity, const cusrl_field_t *fields, int n_fields, int done_field, int64_t cursor,
                             void *stream);

/* ---- post-update policy statistics — cusrl/hook/on_policy/stats.py:28-40 for Normal policies ----
 * out[0] = mean_b KL(N(old_mean, old_std) || N(new_mean, new_std)) summed over the A action dims (kl_divergence),
 * out[1] = mean of advantage * exp(log N(action; new_mean, new_std) - old_logp)  (importance_weighted_advantage),
 * out[2] = mean of new_std (action_std).  [B, A] matrices, old_logp [B], advantage [B, D];
 * partials: double[cusrl_policy_stats_num_partials(B)][3] workspace.  Fixed summation order (deterministic). */
int cusrl_policy_stats(const float *old_mean, const float *old_std, const float *new_mean, const float *new_std,
                       const float *action, const float *old_logp, const float *advantage, int64_t B, int64_t A,
                       int64_t D, double *partials, float *out, void *stream);
int64_t cusrl_policy_stats_num_partials(int64_t B);
/* The same hook for one-hot categorical policies (cusrl/nn/module/distribution.py:332-366; torch's
 * _kl_categorical_categorical): out[0] = mean_b KL(softmax(old_logits) || softmax(new_logits)),
 * out[1] = mean of advantage * exp(log_softmax(new_logits)[taken] - old_logp), out[2] = 0 (no action std). */
int cusrl_categorical_policy_stats(const float *old_logits, const float *new_logits, const float *action,
                                   const float *old_logp, const float *advantage, int64_t B, int64_t A, int64_t D,
                                   double *partials, float *out, void *stream);

/* ---- differentiable policy terms — OnPolicyPreparation.objective, cusrl/hook/on_policy/common.py:29-43 ----
 * logp = sum_a log N(action | mean, std) (cusrl/nn/module/distribution.py:207-209), entropy (distribution.py:211-213),
 * logp_ratio = logp - old_logp, ratio = exp(logp_ratio); outputs [B].  The stock `ppo` composition gets these (and their
 * gradients) from cusrl_ppo_loss_fwd_bwd; this pair serves compositions in which a further hook defines `objective` and
 * may read or differentiate them (one launch forward, one backward instead of the reference's ~15 torch ops + autograd).
 * std: [B, A] (std_rows = B) or ONE [A] vector shared by all rows (std_rows = 1).
 * Backward: g_* = gradients wrt the four outputs, each [B] or NULL; `ratio` = the forward's output (needed with g_ratio).
 * d_mean [B, A]; d_std [B, A], or [A] for a std vector (column sums in fixed order through d_std_partials:
 * float[cusrl_policy_terms_std_partial_rows(B)][A], A <= 64). */
int cusrl_policy_terms_fwd(const float *mean, const float *std, int64_t std_rows, const float *action,
                           const float *old_logp, int64_t B, int64_t A, float *logp_out, float *entropy_out,
                           float *logp_ratio_out, float *ratio_out, void *stream);
int cusrl_policy_terms_bwd(const float *mean, const float *std, int64_t std_rows, const float *action, const float *ratio,
                           const float *g_logp, const float *g_entropy, const float *g_logp_ratio, const float *g_ratio,
                           int64_t B, int64_t A, float *d_mean, float *d_std, float *d_std_partials, void *stream);
int64_t cusrl_policy_terms_std_partial_rows(int64_t B);
/* The same pair for one-hot categorical policies (distribution.py:354-362): logits / one-hot action [B, A]. */
int cusrl_categorical_terms_fwd(const float *logits, const float *action, const float *old_logp, int64_t B, int64_t A,
                                float *logp_out, float *entropy_out, float *logp_ratio_out, float *ratio_out, void *stream);
int cusrl_categorical_terms_bwd(const float *logits, const float *action, const float *ratio, const float *g_logp,
                                const float *g_entropy, const float *g_logp_ratio, const float *g_ratio, int64_t B,
                                int64_t A, float *d_logits, void *stream);

/* ---- MLP backward epilogues (callers of the path: torch.nn.Linear / ReLU backward of the actor-critic) ----
 * Bias gradient = column sums of grad [rows, H]; with `output` != NULL the ReLU backward mask is applied first
 * (grad_in = grad * (output > 0), written to grad_in) and the column sums are taken of the masked gradient, i.e.
 * threshold_backward + sum(0) of autograd in one pass.  partials: float[cusrl_colsum_num_partials(rows, H)][H]
 * workspace; colsum: float[H] (fixed summation order: deterministic), or NULL to get the partial rows only (H % 4 == 0
 * layouts): their column sums are then taken by cusrl_assemble_gradients. */
int cusrl_relu_bwd_colsum(const float *grad, const float *output, float *grad_in, float *partials, float *colsum,
                          int64_t rows, int64_t H, void *stream);
int64_t cusrl_colsum_num_partials(int64_t rows, int64_t H);

/* ---- backward of the FIRST layer of an MLP (ABI 6): y = relu(x W^T + b) whose input needs no gradient — the observation layer
 * of the actor / critic backbones (cusrl/nn/module/mlp.py:89-90; what autograd runs there: threshold_backward, sum(0), the
 * weight-gradient GEMM).  ONE pass over grad_out [rows, H], output [rows, H] (the ReLU's output: grad is masked by output > 0;
 * NULL: no activation) and input [rows, K] produces dW [H, K] = masked^T x and db [H] = column sums of masked — nothing is
 * written back per row (no masked-gradient matrix: dX is not needed).  The products run on v_mfma_f32_16x16x4_f32 (exact f32,
 * the column of ones of db included); the kernel is bound by the 4 (2 H + K) bytes per row it streams.
 * Result: grads float[H * K + H] = dW (row-major) | db — what cusrl_assemble_gradients copies into the two parameters' slots
 * (offset 0 / numel H*K and offset H*K / numel H).  Two launches on `stream`: the pass (per block one [64, K + 1] piece of a
 * partial row) and a small fixed-order sum of the partial rows (a kernel boundary is the cheap cross-XCD fence here).
 * partials: float[cusrl_input_layer_row_blocks(rows, H)][H * K + H] workspace.
 * Supported (cusrl_input_layer_supported): K % 4 == 0, K <= 60, H % 64 == 0, H <= 4096; all pointers 16-byte aligned.
 * Deterministic (fixed summation order). */
int cusrl_input_layer_bwd(const float *grad_out, const float *output, const float *input, int64_t rows, int64_t in_features,
                          int64_t out_features, float *partials, float *grads, void *stream);
int cusrl_input_layer_supported(int64_t in_features, int64_t out_features);
int64_t cusrl_input_layer_row_blocks(int64_t rows, int64_t out_features);

/* ---- backward of a narrow linear layer (policy-mean / value heads; torch.nn.Linear backward with out_features <= 16)
 * One pass over the minibatch produces all three gradients of y = x W^T + b:
 *   grad_input [rows, K] = grad_out W   (skipped when NULL),   dW [O, K] = grad_out^T x,   db [O] = sum_rows grad_out.
 * relu_input != 0: x is the output of a ReLU (x > 0 <=> the ReLU passed); grad_input is then written already
 * masked — the ReLU's backward — and its column sums, the producer layer's bias gradient, are returned too.
 * packed: float[(O + 1) * K + 16] = dW (row-major) | column sums of the masked grad_input (zeros when
 * relu_input == 0) | db, zero padded (NULL: partial rows only, reduced later by cusrl_assemble_gradients);
 * partials: float[cusrl_narrow_linear_num_partials(rows)][(O + 1) * K + 16].
 * Supported shapes (cusrl_narrow_linear_supported): 1 <= O <= 16, K in {32, 64, ..., 1024} a power of two; all
 * pointers 16-byte aligned.  Fixed summation order (deterministic). */
int cusrl_narrow_linear_bwd(const float *grad_out, const float *input, const float *weight, float *grad_input,
                            float *partials, float *packed, int64_t rows, int64_t in_features, int64_t out_features,
                            int relu_input, void *stream);
int64_t cusrl_narrow_linear_num_partials(int64_t rows);
int cusrl_narrow_linear_supported(int64_t in_features, int64_t out_features);
/* Forward of a ONE-output linear layer — the value head (cusrl/nn/module/critic.py:87-88), a discriminator's logit
 * (cusrl/hook/auxiliary/amp.py:138-147): output[b] = input[b, :] . weight + bias[0] (bias may be NULL), one launch instead
 * of torch's broadcast-bias copy + skinny GEMM.  out_features must be 1 (wider heads keep the library GEMM and its bias
 * epilogue), in_features as for cusrl_narrow_linear_supported; input / weight 16-byte aligned. */
int cusrl_narrow_linear_fwd(const float *input, const float *weight, const float *bias, float *output, int64_t rows,
                            int64_t in_features, int64_t out_features, void *stream);

/* ---- inference pass of a two-hidden-layer ReLU MLP + head, one launch (csrc/mlp_forward.hip) ----
 * output[b, :] = w3 relu(w2 relu(w1 input[b, :] + b1) + b2) + b3 — the Linear / ReLU stack of cusrl/nn/module/mlp.py:74-93
 * behind the policy-mean head (cusrl/nn/module/actor.py:69-92, distribution.py:256-262) or the value head
 * (cusrl/nn/module/critic.py:87-88), evaluated WITHOUT autograd: acting (ActorCritic.act, actor_critic.py:226-244), the value
 * targets (hook/on_policy/value.py:58-79), the post-update statistics (hook/on_policy/stats.py:29-40).  Replaces three library
 * GEMMs (+ the sampling launch): every intermediate stays on chip, a workgroup keeps its weight slices in registers and walks
 * 16-row tiles.  fp32 MFMA accumulation; agrees with the library GEMMs to a few ulp of the accumulated magnitude.
 * Weights are torch.nn.Linear's: w1 [hidden1, in], w2 [hidden2, hidden1], w3 [out, hidden2], row-major; b3 may be NULL.
 * Sampling epilogue (eps != NULL; std [out] vector, eps [rows, out]): action = mean + eps * std, logp[b] = sum_a log N(action |
 * mean, std) with the expressions and the summation order of cusrl_normal_sample_logp, std_out (optional) = std repeated per
 * row; `output` (optional then) receives the mean.  Without eps: std / action / logp / std_out must be NULL and output given.
 * Supported (cusrl_mlp2_forward_supported): in % 4 == 0, 4 <= in <= 64; hidden1 in {128, 256}; hidden2 in {64, 128};
 * 1 <= out <= 16; input / weights / biases 16-byte aligned (output too when out % 4 == 0). */
int cusrl_mlp2_forward(const float *input, int64_t rows, int64_t in_features, const float *w1, const float *b1, int64_t hidden1,
                       const float *w2, const float *b2, int64_t hidden2, const float *w3, const float *b3, int64_t out_features,
                       float *output, const float *std, const float *eps, float *action, float *logp, float *std_out,
                       void *stream);
int cusrl_mlp2_forward_supported(int64_t in_features, int64_t hidden1, int64_t hidden2, int64_t out_features);

/* ---- gradient-norm clipping (hook/on_policy/gradient_clipping.py:67-83 -> torch.nn.utils.clip_grad_norm_) ----
 * norm_out[0] = ||grad||_2 (pre-clip, the `grad_norm/default` metric); grad *= min(max_norm / (norm + 1e-6), 1).
 * max_norm < 0: measure only.  grad: float[n], 16-byte aligned (the flat gradient buffer every .grad aliases);
 * partials: double[cusrl_clip_grad_norm_num_partials(n)] workspace.  Fixed summation order (deterministic). */
int cusrl_clip_grad_norm(float *grad, int64_t n, float max_norm, double *partials, float *norm_out, void *stream);
int64_t cusrl_clip_grad_norm_num_partials(int64_t n);

/* ---- flat gradient assembly (the buffer behind actor_critic.py:311-314: backward, all-reduce, clip, step) ----
 * flat[offset .. offset + numel) = sum over `splits` stacked slabs of src [splits][numel]; splits = 1 copies a
 * plain gradient, splits = 0 writes zeros (src ignored).  The split-batch weight-gradient GEMMs leave their
 * [S, out, in] partial products here instead of running one sum(0) each; the column-sum kernels (bias gradients,
 * cusrl_relu_bwd_colsum / cusrl_narrow_linear_bwd called without their output pointer) leave their per-block partial
 * rows here instead of running a finalize launch each (a column window of a wider partial row is addressed through
 * `src` + `row_stride`).  One launch per 24 pieces.  Fixed order.
 * sumsq_partials (optional): double[cusrl_assemble_gradients_blocks(pieces, num_pieces)] — every block also stores the sum
 * of squares of the gradient elements it wrote; summed, that is the squared gradient norm of clip_grad_norm_
 * (gradient_clipping.py:74) and cusrl_adam_step takes the rows as its clip_partials, so a single-process optimizer step
 * needs no separate squared-sum pass (with several ranks the all-reduce changes the gradients in between). */
typedef struct {
    const void *src;
    int64_t offset;     /* element offset of the parameter's slot in `flat` */
    int64_t numel;
    int64_t splits;
    int64_t row_stride; /* elements between consecutive slabs; 0 = numel (densely stacked slabs) */
} cusrl_grad_piece_t;
int cusrl_assemble_gradients(const cusrl_grad_piece_t *pieces, int64_t num_pieces, float *flat, double *sumsq_partials,
                             void *stream);
int64_t cusrl_assemble_gradients_blocks(const cusrl_grad_piece_t *pieces, int64_t num_pieces);

/* Block partials of sum(grad^2) only (the first half of cusrl_clip_grad_norm): the caller hands them to
 * cusrl_adam_step, which applies the clipping coefficient while it streams the gradient. */
int cusrl_grad_sumsq(const float *grad, int64_t n, double *partials, void *stream);

/* ---- optimizer step on flat buffers (torch.optim.Adam / AdamW, the optimizer of cusrl/preset/ppo.py) ----
 * One launch: step += 1; g = grad * clip (clip = min(max_norm / (||grad|| + 1e-6), 1) from `clip_partials`, or 1
 * when NULL; max_norm < 0: norm only); AdamW: param *= 1 - lr wd, Adam: g += wd param;
 * exp_avg += (g - exp_avg)(1 - beta1); exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) g^2;
 * param -= lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps)   (torch's fused kernel).
 * Hyper-parameters are doubles like torch's: 1 - beta and the bias corrections are formed in double, then rounded.
 * param / grad / exp_avg / exp_avg_sq: float[n], 16-byte aligned; step, lr: device float[1] (hipGraph replays see
 * schedule changes); norm_out: device float[1] or NULL (receives ||grad||, the `grad_norm` metric);
 * norm_accumulator (ABI 6): device float[1] or NULL — ||grad|| is ADDED to it (the running sum a captured step keeps of the
 * metric over its replays, cusrl_accumulate_scalars' job without its launch);
 * ticket: device uint32[1], zero-initialised once by the caller and owned by this entry point afterwards. */
int cusrl_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, const float *lr,
                    int64_t n, double beta1, double beta2, double eps, double weight_decay, int decoupled_weight_decay,
                    int maximize, const double *clip_partials, int64_t num_clip_partials, float max_norm,
                    float *norm_out, float *norm_accumulator, uint32_t *ticket, void *stream);

/* cusrl_adam_step over ONE WINDOW of the flat buffers (round 6): the optimizer step of the parameters of one network, issued on
 * the stream that produced their gradients — the critic's on the critic's branch of a captured minibatch step, the others' on
 * the main stream — so that the two branches of the step never join (actor_critic.py:311-320 runs clip + step once, behind one
 * backward).  The clipping coefficient is the GLOBAL one (gradient_clipping.py:74): the squared norm is the sum of
 * clip_partials_a[0 .. num_a) followed by clip_partials_b[0 .. num_b) — the partial rows of the two windows' gradient
 * assemblies, in the order ONE assembly of all parameters would have left them; both launches of a step are handed the same
 * two arrays and compute the same norm to the bit.  param / grad / exp_avg / exp_avg_sq point at the window's first element
 * (16-byte aligned), n is its length; every window has its own `step` counter and `ticket`; step_mirror (optional): a second
 * device float[1] that receives the new step count too (the other window's counter, when a launch covers both windows). */
int cusrl_adam_step_window(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, const float *lr,
                           int64_t n, double beta1, double beta2, double eps, double weight_decay, int decoupled_weight_decay,
                           int maximize, const double *clip_partials_a, int64_t num_a, const double *clip_partials_b,
                           int64_t num_b, float max_norm, float *norm_out, float *norm_accumulator, float *step_mirror,
                           uint32_t *ticket, void *stream);

/* cusrl_adam_step_window that measures the gradient norm ITSELF (ABI 6, second part of round 6): the step of a job with several
 * ranks, where the gradient all-reduce (cusrl/utils/distributed.py:145-172, actor_critic.py:314) sits between the gradient assembly
 * — whose free partial sums are of the un-averaged gradients — and the clipping (gradient_clipping.py:74), so that the squared
 * norm used to be a launch of its own (cusrl_grad_sumsq) on the serial tail of every minibatch step.  One launch: every block sums
 * the squares of its share of norm_grad[0 .. norm_n) (the WHOLE flat gradient buffer, also when the launch steps one window of
 * it), publishes its partial sum in `workspace`, waits for the launch's other blocks, and derives the coefficient from all
 * partial sums in fixed order; then the step of cusrl_adam_step_window over param / grad / exp_avg / exp_avg_sq [0 .. n).  The
 * grid depends on norm_n and the device alone (at most half a block per CU, <= 256: the blocks of a launch meet inside it, and two
 * launches may do so side by side), so the launches of a
 * step's windows find the same norm to the bit.  workspace: cusrl_adam_step_normed_workspace_bytes() device bytes, 8-byte
 * aligned, filled with 0xFF bytes ONCE by the caller and owned by the entry point afterwards (self-re-arming: safe to replay
 * from a hipGraph); two launches that may run side by side need a workspace each.  norm_grad: 16-byte aligned.  Everything else
 * as cusrl_adam_step_window; max_norm < 0: measure only. */
int cusrl_adam_step_normed(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, const float *lr,
                           int64_t n, double beta1, double beta2, double eps, double weight_decay, int decoupled_weight_decay,
                           int maximize, const float *norm_grad, int64_t norm_n, void *workspace, float max_norm, float *norm_out,
                           float *norm_accumulator, float *step_mirror, uint32_t *ticket, void *stream);
int64_t cusrl_adam_step_normed_workspace_bytes(void);

/* ---- running observation statistics ----
 * cusrl/nn/utils/normalization.py:15-50 `mean_var_count` of x [rows, C] restricted to rows with mask != 0 (mask may
 * be NULL = all rows; replaces the host-synchronising boolean-mask select of hook/mdp/observation.py:206-208):
 * batch_mean[C], batch_var[C] (population variance, correction = 0), batch_count (device double[1]);
 * an empty selection yields mean 0, var 1, count 0 like the reference.
 * partials: double[cusrl_masked_stats_num_partials(rows, C)][C + 1][2] workspace. */
int cusrl_masked_col_stats(const float *x, const uint8_t *mask, int64_t rows, int64_t C, double *partials,
                           float *batch_mean, float *batch_var, double *batch_count, void *stream);
int64_t cusrl_masked_stats_num_partials(int64_t rows, int64_t C);

/* cusrl/nn/utils/normalization.py:80-93 `merge_mean_var_` + cusrl/nn/layer/rms.py:163-167: merge batch statistics into
 * the running mean / var (weights count : batch_count), std = sqrt(var + eps), count += batch_count (capped at
 * max_count when max_count > 0); a zero batch_count leaves everything unchanged.  count, batch_count: device double[1]. */
int cusrl_rms_merge(float *mean, float *var, float *std, double *count, const float *batch_mean,
                    const float *batch_var, const double *batch_count, float eps, double max_count, int64_t C,
                    void *stream);

/* cusrl/nn/layer/rms.py:198-203 `normalize`: out = clamp((x - mean) / std, -clamp, clamp) (clamp <= 0: no clamp). */
int cusrl_rms_normalize(const float *x, const float *mean, const float *std, float clamp, float *out, int64_t rows,
                        int64_t C, void *stream);

/* ---- recurrent BPTT data movement ----
 * Done-split sequence layout of a temporal batch — cusrl/nn/utils/recurrent.py:63-92 (`compute_sequence_lengths`),
 * :215-252 (`split_and_pad_sequences`), :160-199 (`scatter_memory`).  done is [L, N] bytes (the last step always ends
 * a sequence).  Sequences are numbered env-major (all segments of env 0 in time order, then env 1, ...).
 *  phase 1  cusrl_sequence_count: sequences per env -> exclusive prefix within each block of 256 envs
 *           (env_prefix[N] int32), block totals (block_totals[cusrl_sequence_blocks(N)] int32) and the total number
 *           of sequences Ns (num_sequences, device int32[1]) — the host reads Ns to size the padded tensors;
 *  phase 2  cusrl_sequence_layout: dest[t*N + n] = pos * Ns + seq (int64: row of slot (t, n) inside the padded
 *           [L, Ns] layout), first_seq[n] = index of env n's first sequence, and mask[pos * Ns + seq] = 1 for valid
 *           padded positions (mask [L, Ns] bytes must be zeroed by the caller); optionally (NULL to skip)
 *           seq_lengths[seq] = number of valid steps of every sequence (int64[Ns], `compute_sequence_lengths`) and
 *           last_seq[n] = index of the sequence still running at the end of env n's column (int64[N]).
 * Packing / unpacking any [L, N, ...] tensor is then cusrl_scatter_rows / cusrl_gather_rows with `dest`.
 * cusrl_gather_memory — cusrl/nn/utils/recurrent.py:124-157 (`gather_memory`, the inverse of scatter_memory used by
 * the packed-sequence forward, cusrl/nn/module/rnn.py:273-291): out[n] = memory[last_seq[n]], zeroed where
 * done_last[n] (the env finished exactly at the last step); rows of row_bytes bytes. */
int cusrl_sequence_count(const uint8_t *done, int64_t L, int64_t N, int32_t *env_prefix, int32_t *block_totals,
                         int32_t *num_sequences, void *stream);
int64_t cusrl_sequence_blocks(int64_t N);
int cusrl_sequence_layout(const uint8_t *done, int64_t L, int64_t N, const int32_t *env_prefix,
                          const int32_t *block_totals, int64_t Ns, int64_t *dest, int64_t *first_seq, uint8_t *mask,
                          int64_t *seq_lengths, int64_t *last_seq, void *stream);
int cusrl_gather_memory(const void *memory, const int64_t *last_seq, const uint8_t *done_last, void *out, int64_t N,
                        int64_t row_bytes, void *stream);

/* ---- intrinsic-reward epilogues ----
 * RND, cusrl/hook/auxiliary/rnd.py:71-74: reward[i] += scale * mean_k (target[i,k] - prediction[i,k])^2 for i < rows
 * (reward has one channel); bonus_out (optional, [rows]) receives the added term for the `rnd_reward` metric. */
int cusrl_rnd_reward(const float *target, const float *prediction, float *reward, float *bonus_out, float scale,
                     int64_t rows, int64_t K, void *stream);
/* AMP, cusrl/hook/auxiliary/amp.py:134-136: reward[i] += scale * -log(max(1 - 1/(1 + exp(-logit[i])), 1e-4));
 * bonus_out as above (`amp_reward`). */
int cusrl_amp_style_reward(const float *logit, float *reward, float *bonus_out, float scale, int64_t rows, void *stream);
/* The style reward together with the mean the metric `amp_reward` records (amp.py:134-136 + agent.record): one
 * single-workgroup launch for rows <= 2^20 (an env step), instead of the reward launch + a reduction. mean_out: float[1]. */
int cusrl_amp_style_reward_mean(const float *logit, float *reward, float *bonus_out, float scale, int64_t rows,
                                float *mean_out, void *stream);

/* ---- AdversarialMotionPrior.post_step up to the discriminator — cusrl/hook/auxiliary/amp.py:112-128 ----
 * agent rows  = cat(state[n, columns], next_state[n, columns])  (columns: K int32 entries, NULL = 0..K-1; state rows are
 *               `state_pitch` floats apart), or `agent_raw` [N, C] when the env provides `amp_obs` (state = NULL);
 * expert rows = dataset[indices[n]] (indices: the int64 draws of torch.randint, so the random stream is the
 *               reference's), or `expert_raw` [N, C] when a demonstration sampler provides them;
 * then transition_rms.update(agent), transition_rms.update(expert) (population statistics, Chan merge with weights
 * count : N, count capped at max_count if > 0 — the arithmetic of cusrl_masked_col_stats + cusrl_rms_merge) and both
 * normalised + clamped (clamp <= 0: none) with the statistics after BOTH updates, into agent_out / expert_out [N, C].
 * C = 2K <= 128, N * C <= cusrl_amp_prepare_max_elements(): TWO launches (assemble + block partial sums; merge +
 * normalise, every block folding the <= 64 partial rows itself) instead of the reference's cat + index + 2 x
 * var_mean/merge + 2 x normalise (~25 torch launches; 10 launches of the per-op kernels above).
 * workspace: double[cusrl_amp_prepare_workspace(N, C)]. */
int cusrl_amp_prepare(const float *state, const float *next_state, int64_t state_pitch, const int32_t *columns, int64_t K,
                      const float *agent_raw, const float *dataset, const int64_t *indices, const float *expert_raw,
                      int64_t N, int64_t C, float *mean, float *var, float *std, double *count, float eps,
                      double max_count, float clamp, float *agent_out, float *expert_out, double *workspace, void *stream);
int64_t cusrl_amp_prepare_max_elements(void);
int64_t cusrl_amp_prepare_workspace(int64_t N, int64_t C);

/* ---- the synthetic benchmark env of BASELINE.json config 2 (cusrl_amd/testing/environment.py; the reference ships
 * no such env — its step is this package's definition of the workload): next_observation ~ N(0, 1) [N, obs_dim],
 * reward ~ N(0, 1) [N, reward_dim], terminated / truncated ~ Bernoulli(p) [N] bytes, reset_rows ~ N(0, 1) [N, obs_dim] (one
 * fresh row per env for the resets), from Philox4x32-10 keyed on (seed, step number, stream, element) in ONE launch.
 * counter: uint64[2] device words {step number, arrival ticket}, zero-initialised by the caller; the launch advances the
 * step number itself (its last block), so a hipGraph replay draws fresh numbers. */
int cusrl_synthetic_env_step(uint64_t seed, uint64_t *counter, int64_t N, int64_t obs_dim, int64_t reward_dim,
                             float p_terminate, float p_truncate, float *next_observation, float *reward,
                             uint8_t *terminated, uint8_t *truncated, float *reset_rows, void *stream);

/* accumulator[i] += *values[i] for i < n <= 32 (values: HOST array of device pointers to fp32 scalars): the per-step
 * running sums of the metrics a captured minibatch / env step records (cusrl/utils/metrics.py:17-28 keeps running means
 * with four torch launches per metric and step).  One launch. */
int cusrl_accumulate_scalars(const float *const *values, int n, float *accumulator, void *stream);

/* ---- RewardShaping.post_step — cusrl/hook/mdp/reward.py:43-47 ----
 * reward = clamp(reward * scale + shift, lower, upper) in place (the product and the sum rounded separately like the two
 * torch ops; has_lower / has_upper = 0: that bound is None). */
int cusrl_reward_shaping(float *reward, float scale, float shift, float lower, float upper, int has_lower, int has_upper,
                         int64_t n, void *stream);

/* ---- nn.MSELoss(prediction, target) forward AND backward — RandomNetworkDistillation.objective, rnd.py:78-81 ----
 * loss_out[0] = mean((prediction - target)^2) over n elements (fp64 block partials, fixed order),
 * d_prediction = 2 (prediction - target) / n.  partials: double[cusrl_mse_loss_num_partials(n)]. */
int cusrl_mse_loss_fwd_bwd(const float *prediction, const float *target, int64_t n, float *loss_out, float *d_prediction,
                           double *partials, void *stream);
int64_t cusrl_mse_loss_num_partials(int64_t n);
/* loss_out[0] = loss_scale * sum(x^2), grad_out = grad_scale * x — the gradient penalty of AMP's discriminator objective
 * (mean_n ||dD/dx_n||^2, cusrl/nn/layer/loss.py:10-56) and what it sends back into the input gradient; same workspace. */
int cusrl_sumsq_fwd_bwd(const float *x, int64_t n, double loss_scale, double grad_scale, float *loss_out, float *grad_out,
                        double *partials, void *stream);
/* AMP's discrimination loss over the joint batch logit[2 * rows] (first half agent = target 0, second half expert =
 * target 1): loss_out[0] = weight * mean BCE-with-logits = (BCE(D(agent), 0) + BCE(D(expert), 1)) / 2 * loss_weight
 * (amp.py:143-147), d_logit = weight * (sigmoid(logit) - target) / (2 rows).  One launch instead of ~10 torch ops. */
int cusrl_bce_pair_fwd_bwd(const float *logit, int64_t rows, float weight, float *loss_out, float *d_logit, void *stream);


/* ---- a6 / a14  data-parallel exchange over RCCL / xGMI — cusrl/utils/distributed.py:58-63, 101-110, 145-183 ----
 * One process per GPU (cusrl/utils/config.py:31-44); a communicator spans all ranks of the job and binds to the
 * device that is current when it is created.  Rank 0 draws the 128-byte id (cusrl_comm_unique_id), the host hands it
 * to the other ranks by any means (the Python host uses its torch.distributed store), every rank calls
 * cusrl_comm_create.  The three collectives only ENQUEUE work on `stream` (no host synchronisation, no allocation), so
 * they may be issued during hipGraph capture and then replay as nodes of the captured minibatch step:
 *   cusrl_allreduce_mean  buffer[i] = mean over ranks of buffer[i], in place, fp32 — `reduce_gradients` on the flat
 *                         gradient buffer (distributed.py:145-172; caller template/actor_critic.py:314);
 *   cusrl_allgather       output[r * bytes : (r + 1) * bytes] = rank r's input — `gather_stack` of cat(mean, var) for
 *                         `reduce_mean_var_` (distributed.py:101-110, 175-183), merged by cusrl_merge_mean_var;
 *   cusrl_broadcast       rank `root`'s buffer to every rank, in place — `broadcast_parameters` (distributed.py:58-63).
 * RCCL is resolved at run time from the copy already loaded into the process (PyTorch-ROCm's), else the system one;
 * cusrl_comm_available() == 0 when neither exists.  A failed RCCL call returns CUSRL_E_COMM and leaves its
 * ncclResult_t text in cusrl_comm_last_error() (host string, valid until the thread's next cusrl_comm_* call). */
typedef struct cusrl_comm cusrl_comm_t;
int cusrl_comm_available(void);
const char *cusrl_comm_last_error(void);
int cusrl_comm_unique_id(void *id_out /* host, 128 bytes */);
int cusrl_comm_create(const void *id /* host, 128 bytes */, int world_size, int rank, cusrl_comm_t **comm_out);
int cusrl_comm_destroy(cusrl_comm_t *comm);
/* Abandon a communicator whose enqueued collectives may never complete (a peer rank failed before issuing its half):
 * ncclCommAbort — stops in-flight kernels instead of waiting for them — then frees the handle. */
int cusrl_comm_abort(cusrl_comm_t *comm);
int cusrl_comm_world_size(const cusrl_comm_t *comm);
int cusrl_allreduce_mean(float *buffer, int64_t count, cusrl_comm_t *comm, void *stream);
int cusrl_allgather(const void *input, void *output, int64_t bytes_per_rank, cusrl_comm_t *comm, void *stream);
int cusrl_broadcast(void *buffer, int64_t bytes, int root, cusrl_comm_t *comm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CUSRL_HIP_H */
