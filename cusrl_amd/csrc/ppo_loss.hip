// Fused PPO objective for gfx950 (a9-a13): Normal log-prob + entropy + probability ratio, clipped surrogate,
// (clipped) value loss and entropy bonus, forward AND backward in one pass over the minibatch.
//
// Reads  advantage 4 + old_logp 4 + action/mean/std 3*4A + return 4D + curr_value 4D (+ old_value 4D) bytes,
// writes d_mean/d_std 2*4A + d_value 4D bytes per sample (264 B at A=12, D=1) — HBM-bound, ~60 flops/sample.
//
// Layout (A % 4 == 0, A <= 32 — ppo_loss_rowgroup_kernel, round 3): ROW GROUPS.  The LPR = A/4 16-byte chunks of a row sit
// in LPR adjacent lanes of one wave (63 of 64 lanes busy at A = 12), so a wave's load covers ~1 KB of contiguous bytes;
// the row sums of log-prob / entropy are LPR neighbour shuffles and EVERY lane of the row evaluates the scalar part (ratio,
// clipping, d loss / d logp) itself — no LDS tile, ONE barrier per block (for the block's fp64 partial sums).  A std handed
// over as its [A] vector keeps d_std in four registers per lane, reduced over the wave's rows by a stride-LPR shuffle tree.
// Per element one v_rcp_f32 of sigma and multiplications instead of five IEEE divisions (the textbook form was bound by its
// own VALU instructions).  Any other width: ppo_loss_rowwise_kernel (one lane per row).  Scalar statistics: fp32 lane sums,
// fp64 wave / block / grid sums in fixed order.
#include <float.h>
#include <stdlib.h>

#include "common.hpp"

namespace cusrl {

// the [B, A] streams with the non-temporal hint (kStream); clang's builtins take native vectors
typedef float loss_native_float4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 loss_nt_load(const float4 *p) {
    const loss_native_float4 v = __builtin_nontemporal_load(reinterpret_cast<const loss_native_float4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void loss_nt_store(float4 *p, const float4 &v) {
    loss_native_float4 n;
    n.x = v.x, n.y = v.y, n.z = v.z, n.w = v.w;
    __builtin_nontemporal_store(n, reinterpret_cast<loss_native_float4 *>(p));
}

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

// All five sums through ONE LDS exchange: wave-shuffle each, lane 0 of every wave parks its five totals, the first five
// threads add the four waves up in fixed order.  kAccumulate: the block ADDS to its row instead of overwriting it — the
// row belongs to this block alone, so a plain read-modify-write is race-free and launches of the same grid, ordered on
// one stream, build up per-block running sums in a fixed order (CUSRL_LOSS_DEFER, see cusrl_ppo_loss_fwd_bwd).
// Wave-wide sum of a double by DPP moves of its two halves (quad swaps, row rotations, row broadcasts): six steps of
// 2 v_mov_dpp + 1 v_add_f64 instead of six ds_bpermute round trips per half; fixed order; the total lands in lane 63.
template <int kCtrl>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), kCtrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), kCtrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_to_last_lane(double v) {
    v += dpp_move<0xb1>(v);   // quad_perm:[1,0,3,2]
    v += dpp_move<0x4e>(v);   // quad_perm:[2,3,0,1]
    v += dpp_move<0x124>(v);  // row_ror:4
    v += dpp_move<0x128>(v);  // row_ror:8   -> every lane holds its row's (16 lanes) sum
    v += dpp_move<0x142>(v);  // row_bcast:15 -> rows 1 and 3 add the row in front of them
    v += dpp_move<0x143>(v);  // row_bcast:31 -> the upper half adds lane 31: lane 63 holds the wave's sum
    return v;
}

__device__ __forceinline__ void park_wave_sums(const double (&acc)[kLossSums], double (*scratch)[kLossSums]) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < kLossSums; ++k) {
        const double total = wave_sum_to_last_lane(acc[k]);
        if (lane == kWave - 1) scratch[wave][k] = total;
    }
}

__device__ __forceinline__ void store_block_partials(double (*scratch)[kLossSums], double *__restrict__ partials,
                                                     bool accumulate) {
    if (threadIdx.x < kLossSums) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) total += scratch[w][threadIdx.x];
        double *slot = partials + int64_t(blockIdx.x) * kLossSums + threadIdx.x;
        *slot = accumulate ? *slot + total : total;
    }
}

__device__ __forceinline__ void write_block_partials(const double (&acc)[kLossSums], double *__restrict__ partials,
                                                     bool accumulate = false) {
    __shared__ double scratch[kWavesPerBlock][kLossSums];
    park_wave_sums(acc, scratch);
    __syncthreads();
    store_block_partials(scratch, partials, accumulate);
}

__device__ __forceinline__ void reduce_std_rows(const float *in, int64_t rows, int A, float *__restrict__ out);
__device__ __forceinline__ void finalize_losses(const double *partials, int64_t P, int64_t B, int D,
                                                const LossParams &p, float *__restrict__ losses_out,
                                                const float *d_std_partials, int A, float *__restrict__ d_std_vector);

// ---- the row-group layout of the [B, A] streams (A = 4 * LPR) --------------------------------------------------------
// A row is LPR 16-byte chunks.  The LPR chunks of a row are held by LPR ADJACENT LANES OF ONE WAVE, a wave covers
// kRowsPerWave = 64 / LPR consecutive rows per round (64 % LPR lanes idle: 1 of 64 at A = 12), so
//   * every global access is a dwordx4 and a wave's load covers kActive * 16 contiguous bytes (coalesced);
//   * the row reductions (log-prob, entropy) are LPR wave shuffles among neighbours — no LDS tile, no barrier — and every
//     lane of the row then evaluates the scalar part (ratio, clip, d loss / d logp) itself: nothing to publish;
//   * a lane's chunks all belong to ONE column group (lane % LPR), so with a std vector its d_std contributions add up
//     in four registers and the column sums of a wave are a strided shuffle tree — no select chain.
// One barrier per block (the cross-wave exchange of the five loss sums and the d_std column sums), against four in the
// flat-chunk layout this replaces (round 2: 0.41 of the HBM roofline with a std vector, bound by its own barriers).
// kRounds chunks per lane are requested before anything is computed (memory-level parallelism).
template <int LPR>
struct RowGroup {
    static constexpr int kRowsPerWave = kWave / LPR;
    static constexpr int kActive = kRowsPerWave * LPR;
    static constexpr int kRounds = LPR >= 4 ? 4 : LPR;
    static constexpr int kRowsPerBlock = kWavesPerBlock * kRowsPerWave * kRounds;
    static constexpr int kTreeStart = kRowsPerWave > 32 ? 32 : kRowsPerWave > 16 ? 16 : kRowsPerWave > 8 ? 8 : 4;
};

static int loss_rows_per_block(int64_t A) {
    if (A % 4 != 0 || A / 4 > 8) return kBlock;  // row-wise kernel: one lane per row
    switch (A / 4) {
        case 1: return RowGroup<1>::kRowsPerBlock;
        case 2: return RowGroup<2>::kRowsPerBlock;
        case 3: return RowGroup<3>::kRowsPerBlock;
        case 4: return RowGroup<4>::kRowsPerBlock;
        case 5: return RowGroup<5>::kRowsPerBlock;
        case 6: return RowGroup<6>::kRowsPerBlock;
        case 7: return RowGroup<7>::kRowsPerBlock;
        default: return RowGroup<8>::kRowsPerBlock;
    }
}
constexpr int kMinLossRowsPerBlock = RowGroup<8>::kRowsPerBlock;  // the smallest of them: workspace bound for any A

// kStdVec: `std` is ONE row [A] shared by every sample (a state-independent std vector, distribution.py:228-247)
// instead of a [B, A] matrix: it is read once per lane instead of streamed, and d_std leaves the kernel as per-block
// column sums [A] (the gradient of the vector) instead of a [B, A] matrix that a sum(0) launch would have to reduce —
// 96 of the 264 bytes per sample disappear.
// The scalar part of a row with float results (the row-group kernel sums at most kRounds of them per lane in fp32 before the
// fp64 wave / block reduction): same decisions as row_terms / value_term above.
struct RowScalars {
    float dlp, ratio, lr, min_term, abs_lr;
};

__device__ __forceinline__ RowScalars row_scalars(float logp, float old_logp, float adv, const LossParams &p) {
    RowScalars r;
    r.lr = logp - old_logp;                             // action_logp_ratio        common.py:35
    r.abs_lr = fabsf(r.lr);                             // metric `ratio` = |logp ratio|   common.py:47
    r.ratio = expf(r.lr);                               // action_prob_ratio        common.py:41
    const float s1 = adv * r.ratio;                     // ppo.py:14
    const float rc = fminf(fmaxf(r.ratio, p.lo), p.hi); // clamp                    ppo.py:16
    const float s2 = adv * rc;
    r.min_term = fminf(s1, s2);
    const bool inside = r.ratio >= p.lo && r.ratio <= p.hi;
    // autograd of min(): ties split evenly, clamp passes on the closed interval
    const float d_ratio = s1 < s2 ? adv : (s1 > s2 ? (inside ? adv : 0.0f) : 0.5f * adv + (inside ? 0.5f * adv : 0.0f));
    r.dlp = p.g_sur * d_ratio * r.ratio;
    return r;
}

__device__ __forceinline__ void value_scalars(float cv, float R, float v, const LossParams &p, float &loss, float &grad) {
    const float e1 = cv - R, l1 = e1 * e1, g1 = 2.0f * e1;
    if (p.value_clip < 0.0f) {  // uniform
        loss = l1, grad = p.g_val * g1;  // mse_loss(return, curr_value)             value.py:132
        return;
    }
    const float c = p.value_clip, dv = cv - v;
    const float e2 = (v + fminf(fmaxf(dv, -c), c)) - R, l2 = e2 * e2;   // value.py:85-89
    const float g2 = (dv >= -c && dv <= c) ? 2.0f * e2 : 0.0f;
    loss = fmaxf(l1, l2);
    grad = p.g_val * (l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * (g1 + g2)));
}

// kFull: every optional output is wanted (the training step) — no per-store pointer tests.
// kWaveRows (round 5): a wave's rows of ALL its rounds are one contiguous run (R * kRowsPerWave <= 64 rows), so the nine
// per-row scalar streams (advantage, old log-prob, return, value[, old value] in; log-prob, entropy, log-ratio, ratio,
// d_value out) move as ONE dword access per stream and wave — lane L holds row L of the run, values travel between the
// row-major lanes and the row groups by wave shuffles — instead of one access per stream and ROUND with a third of the
// lanes active: 53 -> 20 vector-memory instructions per wave at A = 12 (the kernel moved 1.02x its algorithmic bytes
// and still sat at 0.61-0.67 of the HBM roofline: it was bound by memory INSTRUCTIONS, not bytes).
// kStream (round 5): the [B, A] streams — action, mean, (std,) d_mean, (d_std) — with the non-temporal hint, loads AND stores,
// chosen by footprint (loss_streaming below).  Beyond the 256 MB Infinity Cache every line these streams leave in the caches
// is dead weight that evicts somebody's dirty line: 0.60 -> 0.70 of the HBM roofline in the std-vector form, 0.62-0.68 ->
// 0.71 in the matrix form on one box (profiles/r05/loss_variants_ab.txt).  Both halves are needed: non-temporal loads alone
// are SLOWER than the default policy (0.59) and non-temporal stores alone neutral — which is what round 4's policy sweep
// ran into.  The per-row scalar streams (4 B per row each) keep the default policy.
template <int LPR, bool kStdVec, bool kFull, bool kWaveRows, bool kStream>
__global__ __launch_bounds__(kBlock) void ppo_loss_rowgroup_kernel(
    const float *__restrict__ advantage, const float *__restrict__ old_logp, const float *__restrict__ action,
    const float *__restrict__ mean, const float *__restrict__ std, const float *__restrict__ ret,
    const float *__restrict__ curr_value, const float *__restrict__ old_value, int64_t B, int D, LossParams p,
    float *__restrict__ logp_out, float *__restrict__ entropy_out, float *__restrict__ lr_out,
    float *__restrict__ ratio_out, float *__restrict__ d_mean, float *__restrict__ d_std,
    float *__restrict__ d_value, double *__restrict__ partials, float *__restrict__ d_std_partials, int accumulate) {
    using G = RowGroup<LPR>;
    constexpr int R = G::kRounds;
    __shared__ double acc_wave[kWavesPerBlock][kLossSums];
    __shared__ float4 ds_wave[kStdVec ? kWavesPerBlock * LPR : 1];
    const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const unsigned sub = lane % LPR, rloc = lane / LPR;  // chunk within the row, row within the wave's round
    const bool holder = lane < unsigned(G::kActive);
    // block-relative addressing: the block's base pointers are uniform (scalar registers) and everything a lane adds to
    // them is an unsigned 32-bit offset — no 64-bit vector address arithmetic in the rounds
    const int64_t block_row0 = int64_t(blockIdx.x) * G::kRowsPerBlock;
    const unsigned rows_here = unsigned(min(int64_t(G::kRowsPerBlock), B - block_row0));
    const float4 *__restrict__ xb = reinterpret_cast<const float4 *>(action) + block_row0 * LPR;
    const float4 *__restrict__ mb = reinterpret_cast<const float4 *>(mean) + block_row0 * LPR;
    const float4 *__restrict__ sb = reinterpret_cast<const float4 *>(std) + (kStdVec ? 0 : block_row0 * LPR);
    const float *__restrict__ advb = advantage + block_row0, *__restrict__ olpb = old_logp + block_row0;
    const float *__restrict__ retb = ret + block_row0 * D, *__restrict__ cvb = curr_value + block_row0 * D;
    const float *__restrict__ ovb = old_value ? old_value + block_row0 * D : nullptr;
    float4 *__restrict__ dmb = d_mean ? reinterpret_cast<float4 *>(d_mean) + block_row0 * LPR : nullptr;
    float4 *__restrict__ dsb = (!kStdVec && d_std) ? reinterpret_cast<float4 *>(d_std) + block_row0 * LPR : nullptr;
    float *__restrict__ dvb = d_value ? d_value + block_row0 * D : nullptr;
    float *__restrict__ lpo = logp_out ? logp_out + block_row0 : nullptr, *__restrict__ eno = entropy_out ? entropy_out + block_row0 : nullptr;
    float *__restrict__ lro = lr_out ? lr_out + block_row0 : nullptr, *__restrict__ rao = ratio_out ? ratio_out + block_row0 : nullptr;

    // ---- every load of the block's rows is requested up front, unpredicated (rows past the end re-read the block's last
    // row; their results are never stored): the matrix chunks and the per-row scalars of the scalar part
    unsigned lrow[R], q[R];
    bool valid[R];
    float4 x[R], mu[R], sg[kStdVec ? 1 : R];
    float adv[R], olp[R], pre_ret[R], pre_cv[R], pre_ov[R];
    if (kStdVec) sg[0] = sb[sub];
    constexpr unsigned kRunRows = unsigned(R * G::kRowsPerWave);  // rows of one wave over all its rounds (kWaveRows)
    const unsigned run_row0 = wave * kRunRows;
    // kWaveRows: lane L requests the scalars of row L of the wave's run (clamped like every other load)
    float s_adv = 0.f, s_olp = 0.f, s_ret = 0.f, s_cv = 0.f, s_ov = 0.f;
    if (kWaveRows) {
        const unsigned srow = min(run_row0 + lane, rows_here - 1u);
        s_adv = advb[srow];
        s_olp = olpb[srow];
        if (D == 1) {  // uniform
            s_ret = retb[srow];
            s_cv = cvb[srow];
            if (p.value_clip >= 0.0f) s_ov = ovb[srow];
        }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        lrow[k] = kWaveRows ? run_row0 + unsigned(k) * G::kRowsPerWave + rloc
                            : (unsigned(k) * kWavesPerBlock + wave) * G::kRowsPerWave + rloc;
        valid[k] = holder && lrow[k] < rows_here;
        const unsigned crow = min(lrow[k], rows_here - 1u);
        q[k] = crow * LPR + sub;
        if (kStream) {
            x[k] = loss_nt_load(xb + q[k]);
            mu[k] = loss_nt_load(mb + q[k]);
            if (!kStdVec) sg[k] = loss_nt_load(sb + q[k]);
        } else {
            x[k] = xb[q[k]];
            mu[k] = mb[q[k]];
            if (!kStdVec) sg[k] = sb[q[k]];
        }
        pre_ret[k] = pre_cv[k] = pre_ov[k] = 0.f;
        if (!kWaveRows) {
            adv[k] = advb[crow];
            olp[k] = olpb[crow];
            if (D == 1) {  // uniform
                pre_ret[k] = retb[crow];
                pre_cv[k] = cvb[crow];
                if (p.value_clip >= 0.0f) pre_ov[k] = ovb[crow];
            }
        }
    }
    if (kWaveRows) {  // row-major lanes -> row groups (a clamped row's lane holds the clamped row's values: same semantics)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int src = int(unsigned(k) * G::kRowsPerWave + rloc) & (kWave - 1);
            adv[k] = __shfl(s_adv, src, kWave);
            olp[k] = __shfl(s_olp, src, kWave);
            if (D == 1) {
                pre_ret[k] = __shfl(s_ret, src, kWave);
                pre_cv[k] = __shfl(s_cv, src, kWave);
                if (p.value_clip >= 0.0f) pre_ov[k] = __shfl(s_ov, src, kWave);
            }
        }
    }

    // Arithmetic: per element ONE hardware reciprocal of sigma (v_rcp_f32, <= 1 ulp) and multiplications instead of the five
    // IEEE divisions of the textbook form — with those the kernel is bound by its VALU instructions, not by HBM (round 2:
    // matrix and vector form both took 340 us at 1 M envs for 1.74 and 1.13 GB).  z = (x - mu) / sigma:
    //   log-prob term  -z^2 / 2 - log sigma - log sqrt(2 pi)                 distribution.py:207-209
    //   d logp/d mu = z / sigma,  d logp/d sigma = (z^2 - 1) / sigma,  d entropy/d sigma = 1 / sigma
    // A few ulp from the reference's op order, far inside the 1e-5 the losses and gradients are held to.  With a std vector
    // reciprocal, logarithm and the row's entropy do not depend on the row at all: once per lane.
    // A lane sums its (at most kRounds) rows' scalars in fp32; waves and blocks are reduced in fp64.
    float lane_sum[kLossSums] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float4 ds_acc = make_float4(0.f, 0.f, 0.f, 0.f);  // std-vector mode: this lane's column group, summed over its rows
    const unsigned row_lane0 = lane - sub;
    float inv[4], ls[4], entropy = 0.0f;
    auto prepare_std = [&](const float4 &s) {
        const float ss[4] = {s.x, s.y, s.z, s.w};
        float en = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            inv[j] = __builtin_amdgcn_rcpf(ss[j]);
            ls[j] = logf(ss[j]);
            en += entropy_const() + ls[j];  // distribution.py:211-213
        }
        entropy = 0.0f;  // the row's sum, chunk 0 first (the order of a sequential sum over the row)
#pragma unroll
        for (int j = 0; j < LPR; ++j) entropy += __shfl(en, int(row_lane0) + j, kWave);
    };
    if (kStdVec) prepare_std(sg[0]);
    float o_logp = 0.f, o_entropy = 0.f, o_lr = 0.f, o_ratio = 0.f, o_vgrad = 0.f;  // kWaveRows: row (run_row0 + lane)'s outputs
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (!kStdVec) prepare_std(sg[k]);
        const float xs[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
        const float ms[4] = {mu[k].x, mu[k].y, mu[k].z, mu[k].w};
        float z[4], zz[4], lp = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            z[j] = (xs[j] - ms[j]) * inv[j];
            zz[j] = z[j] * z[j];
            lp += (-0.5f * zz[j] - ls[j]) - log_sqrt_2pi();
        }
        float logp = 0.0f;
#pragma unroll
        for (int j = 0; j < LPR; ++j) logp += __shfl(lp, int(row_lane0) + j, kWave);
        // every lane of the row evaluates the scalar part; lane `sub == 0` owns the row's outputs and sums
        const RowScalars r = row_scalars(logp, olp[k], adv[k], p);
        float v_loss = 0.0f, v_grad = 0.0f;
        if (D == 1) value_scalars(pre_cv[k], pre_ret[k], pre_ov[k], p, v_loss, v_grad);
        float gm[4], gs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gm[j] = r.dlp * (z[j] * inv[j]);                       // d logp / d mean
            gs[j] = inv[j] * (r.dlp * (zz[j] - 1.0f) + p.g_ent);  // d logp / d std + d entropy / d std
        }
        const float own = (valid[k] && sub == 0) ? 1.0f : 0.0f, live = valid[k] ? 1.0f : 0.0f;
        lane_sum[1] += own * r.min_term, lane_sum[2] += own * entropy, lane_sum[3] += own * r.abs_lr;
        if (D == 1) lane_sum[0] += own * v_loss, lane_sum[4] += own * pre_cv[k];  // metric `value` = curr_value.sum(-1)
        if (kStdVec) ds_acc.x += live * gs[0], ds_acc.y += live * gs[1], ds_acc.z += live * gs[2], ds_acc.w += live * gs[3];
        if (valid[k]) {
            if (kStream) {
                if (kFull || dmb) loss_nt_store(dmb + (lrow[k] * LPR + sub), make_float4(gm[0], gm[1], gm[2], gm[3]));
                if (!kStdVec && (kFull || dsb)) loss_nt_store(dsb + (lrow[k] * LPR + sub), make_float4(gs[0], gs[1], gs[2], gs[3]));
            } else {
                if (kFull || dmb) dmb[lrow[k] * LPR + sub] = make_float4(gm[0], gm[1], gm[2], gm[3]);
                if (!kStdVec && (kFull || dsb)) dsb[lrow[k] * LPR + sub] = make_float4(gs[0], gs[1], gs[2], gs[3]);
            }
            if (!kWaveRows && sub == 0) {
                if (kFull || lpo) lpo[lrow[k]] = logp;
                if (kFull || eno) eno[lrow[k]] = entropy;
                if (kFull || lro) lro[lrow[k]] = r.lr;
                if (kFull || rao) rao[lrow[k]] = r.ratio;
                if (D == 1 && (kFull || dvb)) dvb[lrow[k]] = v_grad;
            }
        }
        if (kWaveRows) {  // row groups -> row-major lanes: lane L collects row L of the run from the group that holds it
            const int src = int((lane % G::kRowsPerWave) * LPR) & (kWave - 1);
            const bool mine = lane / G::kRowsPerWave == unsigned(k);
            const float c_logp = __shfl(logp, src, kWave), c_lr = __shfl(r.lr, src, kWave), c_ratio = __shfl(r.ratio, src, kWave);
            if (mine) o_logp = c_logp, o_lr = c_lr, o_ratio = c_ratio;
            if (!kStdVec) {
                const float c_en = __shfl(entropy, src, kWave);
                if (mine) o_entropy = c_en;
            }
            if (D == 1) {
                const float c_vg = __shfl(v_grad, src, kWave);
                if (mine) o_vgrad = c_vg;
            }
        }
    }
    if (kWaveRows) {
        const unsigned orow = run_row0 + lane;
        if (lane < kRunRows && orow < rows_here) {
            if (kFull || lpo) lpo[orow] = o_logp;
            if (kFull || eno) eno[orow] = kStdVec ? entropy : o_entropy;  // a std vector: the same entropy for every row
            if (kFull || lro) lro[orow] = o_lr;
            if (kFull || rao) rao[orow] = o_ratio;
            if (D == 1 && (kFull || dvb)) dvb[orow] = o_vgrad;
        }
    }
    double acc[kLossSums];
#pragma unroll
    for (int k = 0; k < kLossSums; ++k) acc[k] = double(lane_sum[k]);
    if (D != 1) {  // several value channels (uniform, rare): the generic per-row walk, fp64 sums
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (valid[k] && sub == 0)
                value_terms(ret, curr_value, old_value, d_value, block_row0 + lrow[k], D, p, acc[0], acc[4]);
    }

    if (kStdVec) {
        // column sums over the wave's rows: lanes of equal `sub` sit LPR apart -> a shuffle tree with stride LPR; the
        // first step folds the rows beyond the largest power of two below kRowsPerWave.  Fixed order.
#pragma unroll
        for (int off = G::kTreeStart; off >= 1; off >>= 1) {
            if (off >= G::kRowsPerWave) continue;
            const float ox = __shfl_down(ds_acc.x, off * LPR, kWave), oy = __shfl_down(ds_acc.y, off * LPR, kWave);
            const float oz = __shfl_down(ds_acc.z, off * LPR, kWave), ow = __shfl_down(ds_acc.w, off * LPR, kWave);
            if (holder && rloc + off < G::kRowsPerWave) ds_acc.x += ox, ds_acc.y += oy, ds_acc.z += oz, ds_acc.w += ow;
        }
        if (lane < LPR) ds_wave[wave * LPR + lane] = ds_acc;
    }
    park_wave_sums(acc, acc_wave);
    __syncthreads();  // the only barrier of the block
    store_block_partials(acc_wave, partials, accumulate != 0);
    if (kStdVec && d_std_partials && threadIdx.x < LPR) {
        const int t = threadIdx.x;
        float4 total = ds_wave[t];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) {
            const float4 v = ds_wave[w * LPR + t];
            total.x += v.x, total.y += v.y, total.z += v.z, total.w += v.w;
        }
        reinterpret_cast<float4 *>(d_std_partials)[int64_t(blockIdx.x) * LPR + t] = total;
    }
}

// Any action width: one lane per row, scalar accesses.
__global__ __launch_bounds__(kBlock) void ppo_loss_rowwise_kernel(
    const float *__restrict__ advantage, const float *__restrict__ old_logp, const float *__restrict__ action,
    const float *__restrict__ mean, const float *__restrict__ std, const float *__restrict__ ret,
    const float *__restrict__ curr_value, const float *__restrict__ old_value, int64_t B, int A, int D, LossParams p,
    float *__restrict__ logp_out, float *__restrict__ entropy_out, float *__restrict__ lr_out,
    float *__restrict__ ratio_out, float *__restrict__ d_mean, float *__restrict__ d_std,
    float *__restrict__ d_value, double *__restrict__ partials, int accumulate) {
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
    write_block_partials(acc, partials, accumulate != 0);
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
    float *__restrict__ d_logits, float *__restrict__ d_value, double *__restrict__ partials, int accumulate) {
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
    write_block_partials(acc, partials, accumulate != 0);
}

// Column sums of `rows` partial rows [rows][A] (A <= 32) by one block, fixed order: lane (a, g) walks rows g, g + 8, ...
// four loads at a time, the 8 row groups are combined through LDS.  out[a] for a < A.
constexpr int kStdSliceRows = 128;  // partial rows one block of the staged reduction takes

__device__ __forceinline__ void reduce_std_rows(const float *in, int64_t rows, int A, float *__restrict__ out) {
    __shared__ float part[kBlock];
    const int a = threadIdx.x & 31, g = threadIdx.x >> 5;
    float total = 0.f;
    if (a < A) {
        int64_t r = g;
        for (; r + 24 < rows; r += 32) {
            const float v0 = in[r * A + a], v1 = in[(r + 8) * A + a], v2 = in[(r + 16) * A + a], v3 = in[(r + 24) * A + a];
            total += (v0 + v1) + (v2 + v3);
        }
        for (; r < rows; r += 8) total += in[r * A + a];
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
    reduce_std_rows(in + first * A, min(int64_t(kStdSliceRows), rows - first), A, stage + int64_t(blockIdx.x) * A);
}

__global__ __launch_bounds__(kBlock) void std_rows_final_kernel(const float *__restrict__ stage, int64_t rows, int A,
                                                                float *__restrict__ out) {
    reduce_std_rows(stage, rows, A, out);
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

__device__ __forceinline__ void finalize_losses(const double *partials, int64_t P, int64_t B, int D,
                                                const LossParams &p, float *__restrict__ losses_out,
                                                const float *d_std_partials, int A, float *__restrict__ d_std_vector) {
    __shared__ double scratch[kWavesPerBlock];
    if (d_std_partials) reduce_std_rows(d_std_partials, P, A, d_std_vector);  // std-vector mode, few blocks
    double sums[kLossSums];
#pragma unroll
    for (int k = 0; k < kLossSums; ++k) {
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < P; i += kBlock) s += partials[i * kLossSums + k];
        sums[k] = block_sum(s, scratch);
    }
    if (threadIdx.x == 0) {
        // (D == 0: the value term is evaluated by cusrl_value_loss_fwd_bwd on the critic's stream — it contributes nothing here)
        losses_out[0] = D > 0 ? float(sums[0] / double(B * D)) * p.w_val : 0.0f;   // mean * weight   value.py:137
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
    finalize_losses(partials, P, B, D, p, losses_out, d_std_partials, A, d_std_vector);
}


// ---- the value term on its own (round 6) -----------------------------------------------------------------------------
// ValueLoss.objective, cusrl/hook/on_policy/value.py:85-89,121-137: mean((cv - R)^2) * w, or the clipped form.  The critic and
// the actor share nothing until the losses are summed (actor_critic.py:309), and a sum differentiated with a unit gradient
// splits into its summands: with the critic on its own stream of a captured minibatch step (template/graphs.py) the value term
// is evaluated THERE, right behind the value head — critic forward -> this launch -> critic backward never meets the actor's
// stream in the middle of the step (one fork and one join per step instead of two each).  Same per-element arithmetic as
// value_scalars above: d_value is bit-identical to what the one-launch objective writes.
constexpr int kValueSums = 2;             // sum of the (clipped) squared errors, sum of the values (metric `value`, value.py:141)
constexpr int kValueElemsPerBlock = kBlock * 4;

__global__ __launch_bounds__(kBlock) void value_loss_kernel(const float *__restrict__ ret, const float *__restrict__ curr_value,
                                                            const float *__restrict__ old_value, int64_t n, LossParams p,
                                                            float *__restrict__ d_value, double *__restrict__ partials,
                                                            int accumulate) {
    __shared__ double scratch[kWavesPerBlock][kValueSums];
    const int64_t base = int64_t(blockIdx.x) * kValueElemsPerBlock + threadIdx.x;
    float cv[4], R[4], ov[4];
    bool live[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // every load requested before anything is computed
        const int64_t i = base + int64_t(k) * kBlock;
        live[k] = i < n;
        const int64_t c = live[k] ? i : n - 1;
        cv[k] = curr_value[c], R[k] = ret[c];
        ov[k] = p.value_clip >= 0.0f ? old_value[c] : 0.0f;
    }
    float loss_sum = 0.f, value_sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float loss, grad;
        value_scalars(cv[k], R[k], ov[k], p, loss, grad);
        if (live[k]) {
            loss_sum += loss, value_sum += cv[k];
            if (d_value) d_value[base + int64_t(k) * kBlock] = grad;
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const double l = wave_sum_to_last_lane(double(loss_sum)), v = wave_sum_to_last_lane(double(value_sum));
    if (lane == kWave - 1) scratch[wave][0] = l, scratch[wave][1] = v;
    __syncthreads();
    if (threadIdx.x < kValueSums) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) total += scratch[w][threadIdx.x];
        double *slot = partials + int64_t(blockIdx.x) * kValueSums + threadIdx.x;
        *slot = accumulate ? *slot + total : total;
    }
}

// losses_out[0] = weighted value loss, losses_out[1] = mean of curr_value.sum(-1)
__global__ __launch_bounds__(kBlock) void value_loss_finalize_kernel(const double *__restrict__ partials, int64_t P, int64_t B,
                                                                     int64_t D, float w_val, float *__restrict__ losses_out) {
    __shared__ double scratch[kWavesPerBlock];
    for (int k = 0; k < kValueSums; ++k) {
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < P; i += kBlock) s += partials[i * kValueSums + k];
        const double total = block_sum(s, scratch);
        if (threadIdx.x == 0) losses_out[k] = k == 0 ? float(total / double(B * D)) * w_val : float(total / double(B));
    }
}

}  // namespace cusrl

using namespace cusrl;

static LossParams loss_params(int64_t B, int64_t D, double clip, double value_clip, double w_sur, double w_val, double w_ent);

extern "C" int64_t cusrl_value_loss_blocks(int64_t B, int64_t D) {
    return B <= 0 || D <= 0 ? 0 : ceil_div(B * D, kValueElemsPerBlock);
}

extern "C" int cusrl_value_loss_fwd_bwd(const float *ret, const float *curr_value, const float *old_value, int64_t B, int64_t D,
                                        double value_clip, double w_val, float *losses_out, float *d_value, double *partials,
                                        int flags, void *stream) {
    if (B <= 0 || D <= 0) return CUSRL_E_INVALID;
    const bool defer = (flags & CUSRL_LOSS_DEFER) != 0;
    if (!ret || !curr_value || !partials || (!losses_out && !defer)) return CUSRL_E_INVALID;
    if (value_clip >= 0.0 && !old_value) return CUSRL_E_INVALID;
    const int64_t blocks = cusrl_value_loss_blocks(B, D);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const LossParams p = loss_params(B, D, 0.2, value_clip, 0.0, w_val, 0.0);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(value_loss_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, s, ret, curr_value, old_value, B * D, p, d_value,
                       partials, int(defer));
    if (int rc = launch_status()) return rc;
    if (defer) return 0;
    hipLaunchKernelGGL(value_loss_finalize_kernel, dim3(1), dim3(kBlock), 0, s, partials, blocks, B, D, float(w_val), losses_out);
    return launch_status();
}

// partial rows (= blocks) the main kernel of a [B, A] minibatch writes; A = 0: the categorical / row-wise form
extern "C" int64_t cusrl_ppo_loss_blocks(int64_t B, int64_t A) {
    return B <= 0 ? 0 : ceil_div(B, A > 0 ? loss_rows_per_block(A) : kBlock);
}

// rows of the fp64 workspace, enough for any action width: one per block, plus one per 256-block slice when the
// reduction is staged
extern "C" int64_t cusrl_ppo_loss_num_partials(int64_t B) {
    const int64_t blocks = B <= 0 ? 0 : ceil_div(B, kMinLossRowsPerBlock);
    return blocks + (blocks > kBlock ? ceil_div(blocks, kBlock) : 0);
}

extern "C" int64_t cusrl_ppo_loss_std_partial_rows(int64_t B) {
    const int64_t blocks = B <= 0 ? 0 : ceil_div(B, kMinLossRowsPerBlock);
    return blocks + (blocks > kStdSliceRows ? ceil_div(blocks, kStdSliceRows) : 0);
}

static LossParams loss_params(int64_t B, int64_t D, double clip, double value_clip, double w_sur, double w_val, double w_ent) {
    LossParams p;
    p.lo = float(1.0 - clip);
    p.hi = float(1.0 + clip);
    p.value_clip = value_clip < 0.0 ? -1.0f : float(value_clip);
    p.g_sur = float(-w_sur / double(B));
    p.g_ent = float(-w_ent / double(B));
    p.g_val = D > 0 ? float(w_val / double(B * D)) : 0.0f;
    p.w_sur = float(w_sur);
    p.w_val = float(w_val);
    p.w_ent = float(w_ent);
    return p;
}

// losses_out from the block partial rows (staged over 256-row slices beyond 256 blocks), optionally with the std-vector
// column sums of up to kStdSliceRows rows
static int launch_finalize(double *partials, int64_t blocks, int64_t B, int64_t A, int64_t D, const LossParams &p,
                           float *losses_out, const float *d_std_partials, float *d_std, hipStream_t s) {
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
                       losses_out, d_std_partials, int(A), d_std);
    return launch_status();
}

extern "C" int cusrl_ppo_loss_categorical_fwd_bwd(const float *advantage, const float *old_logp, const float *action,
                                                  const float *logits, const float *ret, const float *curr_value,
                                                  const float *old_value, int64_t B, int64_t A, int64_t D, double clip,
                                                  double value_clip, double w_sur, double w_val, double w_ent,
                                                  float *losses_out, float *logp_out, float *entropy_out,
                                                  float *logp_ratio_out, float *ratio_out, float *d_logits,
                                                  float *d_value, double *partials, int flags, void *stream) {
    if (B <= 0 || A <= 0 || D < 0) return CUSRL_E_INVALID;  // D == 0: no value term (see cusrl_value_loss_fwd_bwd)
    const bool defer = (flags & CUSRL_LOSS_DEFER) != 0;
    if (!advantage || !old_logp || !action || !logits || (D > 0 && (!ret || !curr_value)) || (!losses_out && !defer) || !partials)
        return CUSRL_E_INVALID;
    if (D > 0 && value_clip >= 0.0 && !old_value) return CUSRL_E_INVALID;
    if (D == 0) value_clip = -1.0, d_value = nullptr;
    if (A > INT32_MAX || D > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const LossParams p = loss_params(B, D, clip, value_clip, w_sur, w_val, w_ent);
    hipStream_t s = as_stream(stream);
    const int64_t blocks = ceil_div(B, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(ppo_loss_categorical_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, s, advantage, old_logp,
                       action, logits, ret, curr_value, old_value, B, int(A), int(D), p, logp_out, entropy_out,
                       logp_ratio_out, ratio_out, d_logits, d_value, partials, int(defer));
    if (int rc = launch_status()) return rc;
    if (defer) return 0;
    return launch_finalize(partials, blocks, B, A, D, p, losses_out, nullptr, nullptr, s);
}

// The instantiated forms (round 6: what measured slower or neutral in rounds 4-5 is no longer compiled — the per-round scalar
// streams, kWaveRows = false, profiles/r05/loss_layout_ab.txt): the scalar streams of a wave's rows always move as one access
// per stream; the training step's launch (kFull) exists with and without the non-temporal [B, A] streams.
#define CUSRL_LAUNCH_ROWGROUP_AS(LPR, VEC, FULL, STREAM)                                                                \
    hipLaunchKernelGGL((ppo_loss_rowgroup_kernel<LPR, VEC, FULL, true, STREAM>), dim3(uint32_t(blocks)), dim3(kBlock), 0, s, \
                       advantage, old_logp, action, mean, std, ret, curr_value, old_value, B, int(D), p, logp_out,          \
                       entropy_out, logp_ratio_out, ratio_out, d_mean, d_std, d_value, partials, d_std_partials, int(defer))
#define CUSRL_LAUNCH_ROWGROUP_FULL(LPR, VEC)                                                                            \
    if (full && streaming) CUSRL_LAUNCH_ROWGROUP_AS(LPR, VEC, true, true);                                              \
    else if (full) CUSRL_LAUNCH_ROWGROUP_AS(LPR, VEC, true, false);                                                     \
    else CUSRL_LAUNCH_ROWGROUP_AS(LPR, VEC, false, false)
#define CUSRL_LAUNCH_ROWGROUP(LPR)                                                                                     \
    if (std_vector) { CUSRL_LAUNCH_ROWGROUP_FULL(LPR, true); }                                                         \
    else { CUSRL_LAUNCH_ROWGROUP_FULL(LPR, false); }

// Cache policy by footprint, like the GAE scan's (advantage.hip): while the launch's bytes fit the 256 MB Infinity Cache the
// default policy is the fastest (config 2's in-step launch moves 4.4 MB out of L2); beyond it the [B, A] streams go past the
// caches.  cusrl_set_option("loss_policy", 1 | 2) forces the default / the streaming form (A/B measurements).
static bool loss_streaming(int64_t B, int64_t A, int64_t D, bool std_vector) {
    if (const int64_t forced = option(kOptLossPolicy)) return forced == 2;
    const int64_t per_row = 8 + (std_vector ? 8 : 12) * A + 8 * D + (std_vector ? 4 : 8) * A + 4 * D + 16;
    return B * per_row > (int64_t(256) << 20);
}

extern "C" int cusrl_ppo_loss_fwd_bwd(const float *advantage, const float *old_logp, const float *action,
                                      const float *mean, const float *std, const float *ret, const float *curr_value,
                                      const float *old_value, int64_t B, int64_t A, int64_t D, double clip,
                                      double value_clip, double w_sur, double w_val, double w_ent, float *losses_out,
                                      float *logp_out, float *entropy_out, float *logp_ratio_out, float *ratio_out,
                                      float *d_mean, float *d_std, float *d_value, double *partials,
                                      int64_t std_rows, float *d_std_partials, int flags, void *stream) {
    if (B <= 0 || A <= 0 || D < 0) return CUSRL_E_INVALID;  // D == 0: no value term (see cusrl_value_loss_fwd_bwd)
    if (std_rows != B && std_rows != 1) return CUSRL_E_INVALID;
    const bool defer = (flags & CUSRL_LOSS_DEFER) != 0;
    const bool std_vector = std_rows == 1 && B != 1;
    if (std_vector && (d_std || defer) && !d_std_partials) return CUSRL_E_INVALID;
    if (!advantage || !old_logp || !action || !mean || !std || (D > 0 && (!ret || !curr_value)) || (!losses_out && !defer) || !partials)
        return CUSRL_E_INVALID;
    if (D > 0 && value_clip >= 0.0 && !old_value) return CUSRL_E_INVALID;
    if (A > INT32_MAX || D > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    if (D == 0) value_clip = -1.0, d_value = nullptr;
    const LossParams p = loss_params(B, D, clip, value_clip, w_sur, w_val, w_ent);
    hipStream_t s = as_stream(stream);
    const bool chunked = A % 4 == 0 && A / 4 <= 8 && aligned(action, 16) && aligned(mean, 16) && aligned(std, 16) &&
                         (!d_mean || aligned(d_mean, 16)) && (!d_std || std_vector || aligned(d_std, 16)) &&
                         (!d_std_partials || aligned(d_std_partials, 16));
    if (std_vector && !chunked) return CUSRL_E_UNSUPPORTED;  // the row-vector form exists for the 16-byte-chunk layout
    const int64_t blocks = ceil_div(B, chunked ? loss_rows_per_block(A) : kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    // the training step wants every output: that variant carries no per-store pointer tests
    const bool full = logp_out && entropy_out && logp_ratio_out && ratio_out && d_mean && (d_value || D == 0) && (std_vector || d_std);
    const bool streaming = loss_streaming(B, A, D, std_vector);
    if (chunked) {
        switch (A / 4) {
            case 1: CUSRL_LAUNCH_ROWGROUP(1); break;
            case 2: CUSRL_LAUNCH_ROWGROUP(2); break;
            case 3: CUSRL_LAUNCH_ROWGROUP(3); break;
            case 4: CUSRL_LAUNCH_ROWGROUP(4); break;
            case 5: CUSRL_LAUNCH_ROWGROUP(5); break;
            case 6: CUSRL_LAUNCH_ROWGROUP(6); break;
            case 7: CUSRL_LAUNCH_ROWGROUP(7); break;
            default: CUSRL_LAUNCH_ROWGROUP(8); break;
        }
    } else {
        hipLaunchKernelGGL(ppo_loss_rowwise_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, s, advantage, old_logp,
                           action, mean, std, ret, curr_value, old_value, B, int(A), int(D), p, logp_out, entropy_out,
                           logp_ratio_out, ratio_out, d_mean, d_std, d_value, partials, int(defer));
    }
    if (int rc = launch_status()) return rc;
    if (defer) return 0;  // the caller reduces the block rows itself (losses: whenever it wants totals; d_std: assembly)
    // gradient of the std vector = column sums of the per-block sums: inside the finalize launch for up to 128 blocks
    // (a 32 768-row minibatch), staged over 128-row slices beyond that
    const bool reduce_std = std_vector && d_std;
    const bool staged = reduce_std && blocks > kStdSliceRows;
    if (int rc = launch_finalize(partials, blocks, B, A, D, p, losses_out, (reduce_std && !staged) ? d_std_partials : nullptr,
                                 d_std, s))
        return rc;
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
