// Gradient-norm clipping of the flat gradient buffer (a14 neighbourhood: runs between the gradient all-reduce and
// the optimizer step).  Counterpart of hook/on_policy/gradient_clipping.py:67-83, which calls
// torch.nn.utils.clip_grad_norm_: total = ||g||_2, g *= min(max_norm / (total + 1e-6), 1).  As torch ops that is
// norm + add + reciprocal + mul + clamp + mul = six launches over a 370 KB buffer; here two: block partials of the
// squared sum, then every block re-derives the (uniform) coefficient from the partials and scales its slice.
#include "common.hpp"

namespace cusrl {

constexpr int kNormMaxBlocks = 64;           // blocks of the stand-alone squared-sum pass
constexpr int kMaxClipPartials = 1 << 16;    // partial rows a consumer (scale pass / Adam step) walks with one wave
constexpr int kNormFloatsPerBlock = 256 * 16;  // 4 float4 per thread

__global__ __launch_bounds__(kBlock) void sumsq_partials_kernel(const float *__restrict__ g, int64_t n,
                                                                double *__restrict__ partials) {
    __shared__ double scratch[kWavesPerBlock];
    double acc = 0.0;
    const int64_t n4 = n / 4;
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(g);
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += int64_t(gridDim.x) * kBlock) {
        const float4 v = g4[i];
        acc += double(v.x * v.x + v.y * v.y) + double(v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {  // ragged tail (n % 4 elements)
        const float v = g[n4 * 4 + threadIdx.x];
        acc += double(v * v);
    }
    const double total = block_sum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void clip_scale_kernel(float *__restrict__ g, int64_t n,
                                                            const double *__restrict__ partials, int num_partials,
                                                            float max_norm, float *__restrict__ norm_out) {
    __shared__ float coef_shared;
    if (threadIdx.x < kWave) {  // wave 0: fixed-order sum of the partials
        double p = 0.0;
        for (int i = threadIdx.x; i < num_partials; i += kWave) p += partials[i];
        p = wave_sum(p);
        if (threadIdx.x == 0) {
            const float norm = float(sqrt(p));
            const float coef = max_norm / (norm + 1e-6f);  // clip_grad_norm_: max_norm / (total_norm + 1e-6)
            coef_shared = coef < 1.0f ? coef : 1.0f;       // clamp(max=1): a NaN coefficient propagates like torch
            if (coef != coef) coef_shared = coef;
            if (blockIdx.x == 0) norm_out[0] = norm;
        }
    }
    __syncthreads();
    const float coef = coef_shared;
    if (max_norm < 0.0f || coef == 1.0f) return;  // negative limit = measure only; x * 1.0f is the identity
    const int64_t n4 = n / 4;
    float4 *__restrict__ g4 = reinterpret_cast<float4 *>(g);
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += int64_t(gridDim.x) * kBlock) {
        float4 v = g4[i];
        v.x *= coef, v.y *= coef, v.z *= coef, v.w *= coef;
        g4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) g[n4 * 4 + threadIdx.x] *= coef;
}

static int norm_blocks(int64_t n) {
    const int64_t blocks = ceil_div(n, kNormFloatsPerBlock);
    return int(blocks < 1 ? 1 : (blocks > kNormMaxBlocks ? kNormMaxBlocks : blocks));
}

}  // namespace cusrl

extern "C" int64_t cusrl_clip_grad_norm_num_partials(int64_t n) { return n < 0 ? 0 : cusrl::norm_blocks(n); }

extern "C" int cusrl_clip_grad_norm(float *grad, int64_t n, float max_norm, double *partials, float *norm_out,
                                    void *stream) {
    using namespace cusrl;
    if (n < 0 || !partials || !norm_out || (n > 0 && !grad)) return CUSRL_E_INVALID;
    if (n > 0 && !aligned(grad, 16)) return CUSRL_E_UNSUPPORTED;
    const int blocks = norm_blocks(n);
    sumsq_partials_kernel<<<blocks, kBlock, 0, as_stream(stream)>>>(grad, n, partials);
    // the scale pass wants the whole chip, not just the (<= 64) reduction blocks
    const int64_t scale_blocks = ceil_div(n / 4 > 0 ? n / 4 : 1, kBlock);
    clip_scale_kernel<<<int(scale_blocks > 1024 ? 1024 : scale_blocks), kBlock, 0, as_stream(stream)>>>(
        grad, n, partials, blocks, max_norm, norm_out);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Adam / AdamW step over flat buffers (every parameter, its gradient and both moments alias one buffer each:
// cusrl_amd/utils/flat_optimizer.py).  torch.optim.Adam(fused=True) is a multi-tensor-apply launch plus a foreach
// launch for the step counters — 22 + 5 us for the 92 569 parameters of the ppo preset's networks (13 tensors);
// one streaming pass over four 370 KB buffers is launch-bound at a few us.  The pending clipping coefficient
// (partials of cusrl_grad_sumsq) is applied on the fly, so "clip + step" is two launches instead of six + two.
namespace cusrl {

struct AdamParams {
    double beta1, beta2;  // doubles like torch's python floats: 1 - beta is formed in double, THEN rounded to fp32
    float eps, weight_decay, max_norm;
    int decoupled, maximize, num_clip_partials;
};

__global__ __launch_bounds__(kBlock) void adam_step_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                           float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                           float *__restrict__ step, const float *__restrict__ lr,
                                                           const double *__restrict__ clip_partials,
                                                           const double *__restrict__ clip_partials_b, int num_b,
                                                           float *__restrict__ norm_out, float *__restrict__ norm_accumulator,
                                                           float *__restrict__ step_mirror, unsigned int *__restrict__ ticket,
                                                           int64_t n, AdamParams a) {
    __shared__ float shared[4];  // clip coefficient, step size, sqrt(bias_correction2), new step count
    // The four streams of this thread's FIRST float4 are requested before anything else: they do not depend on the clipping
    // coefficient, and the chain partial rows -> norm -> coefficient below is a memory round trip of its own (round 6: the
    // launch is latency, not bandwidth — 370 KB per buffer — and sits on the serial tail of every minibatch step).
    const int64_t n4 = n / 4;
    float4 *p4 = reinterpret_cast<float4 *>(param), *m4 = reinterpret_cast<float4 *>(exp_avg),
           *v4 = reinterpret_cast<float4 *>(exp_avg_sq);
    const float4 *g4 = reinterpret_cast<const float4 *>(grad);
    const int64_t i0 = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), m0 = p0, v0 = p0, g0 = p0;
    if (i0 < n4) p0 = p4[i0], m0 = m4[i0], v0 = v4[i0], g0 = g4[i0];
    if (threadIdx.x < kWave) {
        float coef = 1.0f;
        if (clip_partials) {  // uniform branch; the partials of cusrl_grad_sumsq (<= 64) or of the gradient assembly
            // (a second array continues the first: the partial rows of two gradient assemblies — one per network window,
            // cusrl_adam_step_window — are summed exactly as ONE assembly's rows would be)
            double p = 0.0;
            const int first = a.num_clip_partials, total = first + num_b;
            for (int i = threadIdx.x; i < total; i += kWave) p += i < first ? clip_partials[i] : clip_partials_b[i - first];
            p = wave_sum(p);
            const float norm = float(sqrt(p));
            if (a.max_norm >= 0.0f) {
                const float c = a.max_norm / (norm + 1e-6f);  // clip_grad_norm_: max_norm / (total_norm + 1e-6)
                coef = c < 1.0f ? c : (c != c ? c : 1.0f);
            }
            if (threadIdx.x == 0 && blockIdx.x == 0) {
                if (norm_out) norm_out[0] = norm;
                if (norm_accumulator) norm_accumulator[0] += norm;  // the running sum a captured step's metric tap keeps
            }
        }
        if (threadIdx.x == 0) {
            // every block reads the counter here, before the LAST block to finish bumps it (see the ticket below)
            const float t = step[0] + 1.0f;
            const double bc1 = 1.0 - pow(a.beta1, double(t)), bc2 = 1.0 - pow(a.beta2, double(t));
            shared[0] = coef;
            shared[1] = float(double(lr[0]) / bc1);  // step_size = lr / bias_correction1
            shared[2] = float(sqrt(bc2));
            shared[3] = t;
        }
    }
    __syncthreads();
    const float coef = shared[0], step_size = shared[1], bc2_sqrt = shared[2], t = shared[3];
    const float beta2 = float(a.beta2), omb1 = float(1.0 - a.beta1), omb2 = float(1.0 - a.beta2);
    const float decay = a.decoupled ? 1.0f - lr[0] * a.weight_decay : 1.0f;
    const float l2 = a.decoupled ? 0.0f : a.weight_decay;
    const float sign = a.maximize ? -coef : coef;

    auto update = [&](float &p, float g, float &m, float &v) {
        g *= sign;
        p *= decay;                                   // AdamW: param *= 1 - lr * weight_decay
        g += l2 * p;                                  // Adam:  grad += weight_decay * param
        m += (g - m) * omb1;                          // exp_avg.lerp_(grad, 1 - beta1)
        v = beta2 * v + omb2 * (g * g);               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        p -= step_size * (m / (sqrtf(v) / bc2_sqrt + a.eps));
    };
    for (int64_t i = i0; i < n4; i += int64_t(gridDim.x) * kBlock) {
        float4 p = p0, m = m0, v = v0;
        const float4 g = g0;
        const int64_t next = i + int64_t(gridDim.x) * kBlock;
        if (next < n4) p0 = p4[next], m0 = m4[next], v0 = v4[next], g0 = g4[next];
        update(p.x, g.x, m.x, v.x), update(p.y, g.y, m.y, v.y), update(p.z, g.z, m.z, v.z), update(p.w, g.w, m.w, v.w);
        p4[i] = p, m4[i] = m, v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
        const int64_t i = n4 * 4 + threadIdx.x;
        update(param[i], grad[i], exp_avg[i], exp_avg_sq[i]);
    }
    // The last block to finish publishes the new step count and re-arms the ticket (self-resetting, graph-safe).  No fence
    // around the ticket (round 6): what has to be ordered is every block's READ of step[0] — consumed by the stores above,
    // hence complete before this barrier — in front of the last block's write, and the ticket's atomic provides exactly that;
    // the written values are only read by LATER launches.  An agent-scope fence writes back and invalidates the XCD's whole L2
    // on this part, once per block, on the serial tail of every minibatch step.
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            step[0] = t;
            if (step_mirror) step_mirror[0] = t;  // a second counter kept equal (the other window's, cusrl_adam_step_window)
            *ticket = 0u;
        }
    }
}


// ---- the Adam step that measures the gradient norm ITSELF (multi-rank steps: the all-reduce sits between the gradient
// assembly, whose free partial sums are of the un-averaged gradients, and the clipping — so the norm used to be a launch of
// its own on the serial tail of every minibatch step, gradient_clipping.py:74 behind distributed.py:145-172).  One launch: every
// block sums the squares of its share of `norm_grad` (the WHOLE flat gradient buffer, also when the launch steps one window),
// publishes the partial sum in a slot of the workspace, waits until every slot of the launch is filled, and derives the — to the
// bit identical — coefficient from the slots in fixed order.  A grid-wide meeting inside a plain launch: the grid is at most half
// a block per CU of the device (and at most kNormedMaxBlocks), so every block is resident, or becomes resident as unrelated work
// drains, while the others wait; two such launches side by side (the two windows of an unjoined step) are 2 x 91 blocks of the ppo
// preset's networks on 256 CUs.
// No fence anywhere: a slot IS its own flag (device-scope read-modify-writes on one 8-byte word; "empty" is a bit pattern no
// sum of squares has), the slots of the NEXT launch are re-armed by this one (two sets, chosen by a launch counter the last
// block to finish bumps), so the entry point stays self-resetting and graph-safe like the ticket.
constexpr int kNormedMaxBlocks = 256;
constexpr unsigned long long kSlotEmpty = ~0ull;                     // (the workspace starts as 0xFF bytes)
constexpr unsigned long long kCanonicalNan = 0x7ff8000000000000ull;
constexpr int kNormedSpinLimit = 1 << 20;  // a slot that never fills (cannot happen while the grid is resident): NaN, loudly

struct NormedWorkspace {
    unsigned long long slots[2][kNormedMaxBlocks];
    unsigned int launches;  // parity = the set this launch publishes into
    unsigned int pad[3];
};
static_assert(kNormedMaxBlocks == kBlock, "block 0 re-arms one slot per thread");

__global__ __launch_bounds__(kBlock) void adam_step_normed_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                                  float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                                  float *__restrict__ step, const float *__restrict__ lr,
                                                                  const float *__restrict__ norm_grad, int64_t norm_n,
                                                                  NormedWorkspace *__restrict__ ws, float *__restrict__ norm_out,
                                                                  float *__restrict__ norm_accumulator,
                                                                  float *__restrict__ step_mirror, unsigned int *__restrict__ ticket,
                                                                  int64_t n, AdamParams a) {
    __shared__ float shared[4];  // clip coefficient, step size, sqrt(bias_correction2), new step count
    __shared__ double scratch[kWavesPerBlock];
    // (as in adam_step_kernel: the window's first float4s are requested before anything else)
    const int64_t n4 = n / 4;
    float4 *p4 = reinterpret_cast<float4 *>(param), *m4 = reinterpret_cast<float4 *>(exp_avg),
           *v4 = reinterpret_cast<float4 *>(exp_avg_sq);
    const float4 *g4 = reinterpret_cast<const float4 *>(grad);
    const int64_t i0 = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), m0 = p0, v0 = p0, g0 = p0;
    if (i0 < n4) p0 = p4[i0], m0 = m4[i0], v0 = v4[i0], g0 = g4[i0];
    // every block reads the launch counter here, before the LAST block to finish bumps it (behind the meeting below)
    const unsigned int set = ws->launches & 1u;
    unsigned long long *mine = ws->slots[set];
    if (blockIdx.x == 0) atomicExch(&ws->slots[set ^ 1u][threadIdx.x], kSlotEmpty);  // the next launch's set (the previous
                                                                                      // launch, its last user, is complete)
    double acc = 0.0;
    const int64_t nn4 = norm_n / 4;
    const float4 *__restrict__ ng4 = reinterpret_cast<const float4 *>(norm_grad);
    for (int64_t i = i0; i < nn4; i += int64_t(gridDim.x) * kBlock) {
        const float4 v = ng4[i];
        acc += double(v.x * v.x + v.y * v.y) + double(v.z * v.z + v.w * v.w);  // (sumsq_partials_kernel's terms)
    }
    if (blockIdx.x == 0 && threadIdx.x < norm_n - nn4 * 4) {
        const float v = norm_grad[nn4 * 4 + threadIdx.x];
        acc += double(v * v);
    }
    const double total = block_sum(acc, scratch);
    if (threadIdx.x < kWave) {
        if (threadIdx.x == 0) {
            unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(total));
            atomicExch(&mine[blockIdx.x], bits == kSlotEmpty ? kCanonicalNan : bits);
        }
        double p = 0.0;
        for (int i = threadIdx.x; i < int(gridDim.x); i += kWave) {
            unsigned long long bits;
            int spins = 0;
            while ((bits = atomicAdd(&mine[i], 0ull)) == kSlotEmpty) {
                if (++spins > kNormedSpinLimit) {
                    bits = kCanonicalNan;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            p += __longlong_as_double(static_cast<long long>(bits));
        }
        p = wave_sum(p);
        if (threadIdx.x == 0) {
            const float norm = float(sqrt(p));
            float coef = 1.0f;
            if (a.max_norm >= 0.0f) {
                const float c = a.max_norm / (norm + 1e-6f);  // clip_grad_norm_: max_norm / (total_norm + 1e-6)
                coef = c < 1.0f ? c : (c != c ? c : 1.0f);
            }
            if (blockIdx.x == 0) {
                if (norm_out) norm_out[0] = norm;
                if (norm_accumulator) norm_accumulator[0] += norm;
            }
            const float t = step[0] + 1.0f;
            const double bc1 = 1.0 - pow(a.beta1, double(t)), bc2 = 1.0 - pow(a.beta2, double(t));
            shared[0] = coef;
            shared[1] = float(double(lr[0]) / bc1);
            shared[2] = float(sqrt(bc2));
            shared[3] = t;
        }
    }
    __syncthreads();
    const float coef = shared[0], step_size = shared[1], bc2_sqrt = shared[2], t = shared[3];
    const float beta2 = float(a.beta2), omb1 = float(1.0 - a.beta1), omb2 = float(1.0 - a.beta2);
    const float decay = a.decoupled ? 1.0f - lr[0] * a.weight_decay : 1.0f;
    const float l2 = a.decoupled ? 0.0f : a.weight_decay;
    const float sign = a.maximize ? -coef : coef;

    auto update = [&](float &p, float g, float &m, float &v) {  // (adam_step_kernel's, operation for operation)
        g *= sign;
        p *= decay;
        g += l2 * p;
        m += (g - m) * omb1;
        v = beta2 * v + omb2 * (g * g);
        p -= step_size * (m / (sqrtf(v) / bc2_sqrt + a.eps));
    };
    for (int64_t i = i0; i < n4; i += int64_t(gridDim.x) * kBlock) {
        float4 p = p0, m = m0, v = v0;
        const float4 g = g0;
        const int64_t next = i + int64_t(gridDim.x) * kBlock;
        if (next < n4) p0 = p4[next], m0 = m4[next], v0 = v4[next], g0 = g4[next];
        update(p.x, g.x, m.x, v.x), update(p.y, g.y, m.y, v.y), update(p.z, g.z, m.z, v.z), update(p.w, g.w, m.w, v.w);
        p4[i] = p, m4[i] = m, v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
        const int64_t i = n4 * 4 + threadIdx.x;
        update(param[i], grad[i], exp_avg[i], exp_avg_sq[i]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {  // (every block has read step[0] and the launch counter by now)
            step[0] = t;
            if (step_mirror) step_mirror[0] = t;
            ws->launches += 1u;
            *ticket = 0u;
        }
    }
}

}  // namespace cusrl

extern "C" int cusrl_grad_sumsq(const float *grad, int64_t n, double *partials, void *stream) {
    using namespace cusrl;
    if (n < 0 || !partials || (n > 0 && !grad)) return CUSRL_E_INVALID;
    if (n > 0 && !aligned(grad, 16)) return CUSRL_E_UNSUPPORTED;
    sumsq_partials_kernel<<<norm_blocks(n), kBlock, 0, as_stream(stream)>>>(grad, n, partials);
    return launch_status();
}

extern "C" int cusrl_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step,
                               const float *lr, int64_t n, double beta1, double beta2, double eps, double weight_decay,
                               int decoupled_weight_decay, int maximize, const double *clip_partials,
                               int64_t num_clip_partials, float max_norm, float *norm_out, float *norm_accumulator,
                               uint32_t *ticket, void *stream) {
    using namespace cusrl;
    if (n <= 0 || !param || !grad || !exp_avg || !exp_avg_sq || !step || !lr || !ticket) return CUSRL_E_INVALID;
    if (clip_partials && (num_clip_partials < 1 || num_clip_partials > kMaxClipPartials)) return CUSRL_E_INVALID;
    if (!aligned(param, 16) || !aligned(grad, 16) || !aligned(exp_avg, 16) || !aligned(exp_avg_sq, 16))
        return CUSRL_E_UNSUPPORTED;
    AdamParams a{beta1, beta2, float(eps), float(weight_decay), max_norm, decoupled_weight_decay, maximize,
                 int(num_clip_partials)};
    const int64_t blocks = ceil_div(n / 4 > 0 ? n / 4 : 1, kBlock);
    adam_step_kernel<<<int(blocks > 1024 ? 1024 : blocks), kBlock, 0, as_stream(stream)>>>(
        param, grad, exp_avg, exp_avg_sq, step, lr, clip_partials, nullptr, 0, norm_out, norm_accumulator, nullptr, ticket, n, a);
    return launch_status();
}

extern "C" int cusrl_adam_step_window(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step,
                                      const float *lr, int64_t n, double beta1, double beta2, double eps, double weight_decay,
                                      int decoupled_weight_decay, int maximize, const double *clip_partials_a, int64_t num_a,
                                      const double *clip_partials_b, int64_t num_b, float max_norm, float *norm_out,
                                      float *norm_accumulator, float *step_mirror, uint32_t *ticket, void *stream) {
    using namespace cusrl;
    if (n <= 0 || !param || !grad || !exp_avg || !exp_avg_sq || !step || !lr || !ticket) return CUSRL_E_INVALID;
    if (num_a < 0 || num_b < 0 || (num_a > 0 && !clip_partials_a) || (num_b > 0 && !clip_partials_b)) return CUSRL_E_INVALID;
    if (num_b > 0 && num_a == 0) return CUSRL_E_INVALID;  // (the second array continues the first)
    if (num_a + num_b > kMaxClipPartials) return CUSRL_E_INVALID;
    if (!aligned(param, 16) || !aligned(grad, 16) || !aligned(exp_avg, 16) || !aligned(exp_avg_sq, 16))
        return CUSRL_E_UNSUPPORTED;
    AdamParams a{beta1, beta2, float(eps), float(weight_decay), max_norm, decoupled_weight_decay, maximize, int(num_a)};
    const int64_t blocks = ceil_div(n / 4 > 0 ? n / 4 : 1, kBlock);
    adam_step_kernel<<<int(blocks > 1024 ? 1024 : blocks), kBlock, 0, as_stream(stream)>>>(
        param, grad, exp_avg, exp_avg_sq, step, lr, num_a > 0 ? clip_partials_a : nullptr, clip_partials_b, int(num_b), norm_out,
        norm_accumulator, step_mirror, ticket, n, a);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Assembly of the flat gradient buffer: every parameter's gradient lands in its slot of the one buffer that the
// all-reduce, the clipping norm and the optimizer step work on.  A piece is `splits` stacked partial gradients
// (the split-batch weight-gradient GEMMs leave [S, out, in] slabs: their sum IS the gradient), a plain gradient
// (splits = 1) or nothing (splits = 0: zeros).  One launch replaces one torch.cat plus one sum(0) per split GEMM.
namespace cusrl {

struct GradPiece {
    const float *src;
    int64_t offset, numel, row_stride;  // slab s, element e lives at src[s * row_stride + e]
    int32_t splits, wide;               // wide: many slabs (block partials of a column-sum kernel), reduced cooperatively
};

struct GradTable {
    int32_t n;
    int32_t block_start[CUSRL_MAX_FIELDS + 1];
    GradPiece piece[CUSRL_MAX_FIELDS];
};

constexpr int kAssemblePerBlock = kBlock * 4;  // elements of one piece per block (few slabs)
constexpr int kAssembleWideCols = 16;          // elements per block of a WIDE piece: 16 columns x 16 slab groups
constexpr int kAssembleWideSplits = 16;        // more slabs than this -> wide

// sumsq (optional): partial row blockIdx.x + sumsq_base receives the sum of squares of the elements this block wrote —
// the squared gradient norm the clipping needs, taken while the values are in registers (one launch fewer per optimizer
// step whenever nothing — no cross-rank reduction — changes the gradients between assembly and clipping).
__global__ __launch_bounds__(kBlock) void assemble_gradients_kernel(const GradTable tab, float *__restrict__ flat,
                                                                    double *__restrict__ sumsq, int sumsq_base) {
    __shared__ double sq_scratch[kWavesPerBlock];
    double sq = 0.0;
    const int blk = blockIdx.x;
    int f = 0;
#pragma unroll
    for (int i = 1; i < CUSRL_MAX_FIELDS; ++i) f += (i < tab.n && blk >= tab.block_start[i]) ? 1 : 0;
    const GradPiece piece = tab.piece[f];
    const float *__restrict__ src = piece.src;
    const int64_t n = piece.numel;
    const int64_t stride = piece.row_stride;
    if (piece.wide) {  // uniform per block: hundreds of partial rows (bias / head gradients), 16 slab groups in flight
        __shared__ float red[kBlock];
        const int c = threadIdx.x % kAssembleWideCols, g = threadIdx.x / kAssembleWideCols;
        constexpr int kGroups = kBlock / kAssembleWideCols;
        const int64_t e = int64_t(blk - tab.block_start[f]) * kAssembleWideCols + c;
        float total = 0.f;
        if (e < n) {
            int s = g;
            for (; s + 3 * kGroups < piece.splits; s += 4 * kGroups) {
                const float a = src[int64_t(s) * stride + e], b = src[int64_t(s + kGroups) * stride + e],
                            c2 = src[int64_t(s + 2 * kGroups) * stride + e], d = src[int64_t(s + 3 * kGroups) * stride + e];
                total += (a + b) + (c2 + d);
            }
            for (; s < piece.splits; s += kGroups) total += src[int64_t(s) * stride + e];
        }
        red[threadIdx.x] = total;
        __syncthreads();
        if (g == 0 && e < n) {
            float sum = red[c];
#pragma unroll
            for (int k = 1; k < kGroups; ++k) sum += red[k * kAssembleWideCols + c];
            flat[piece.offset + e] = sum;
            sq = double(sum * sum);
        }
        if (sumsq) {  // uniform
            const double total_sq = block_sum(sq, sq_scratch);
            if (threadIdx.x == 0) sumsq[sumsq_base + blk] = total_sq;
        }
        return;
    }
    const int64_t base = int64_t(blk - tab.block_start[f]) * kAssemblePerBlock + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t e = base + int64_t(k) * kBlock;
        if (e >= n) break;
        float total = 0.f;
        int s = 0;
        for (; s + 4 <= piece.splits; s += 4) {  // fixed order, four independent loads in flight
            const float a = src[int64_t(s) * stride + e], b = src[int64_t(s + 1) * stride + e],
                        c = src[int64_t(s + 2) * stride + e], d = src[int64_t(s + 3) * stride + e];
            total += (a + b) + (c + d);
        }
        for (; s < piece.splits; ++s) total += src[int64_t(s) * stride + e];
        flat[piece.offset + e] = total;
        sq += double(total * total);
    }
    if (sumsq) {  // uniform
        const double total_sq = block_sum(sq, sq_scratch);
        if (threadIdx.x == 0) sumsq[sumsq_base + blk] = total_sq;
    }
}

}  // namespace cusrl

static int64_t assemble_blocks(const cusrl_grad_piece_t &p) {
    return cusrl::ceil_div(p.numel, p.splits > cusrl::kAssembleWideSplits ? cusrl::kAssembleWideCols : cusrl::kAssemblePerBlock);
}

extern "C" int64_t cusrl_assemble_gradients_blocks(const cusrl_grad_piece_t *pieces, int64_t num_pieces) {
    if (num_pieces < 0 || (num_pieces > 0 && !pieces)) return -1;
    int64_t blocks = 0;
    for (int64_t i = 0; i < num_pieces; ++i) {
        if (pieces[i].numel < 0 || pieces[i].splits < 0) return -1;
        blocks += assemble_blocks(pieces[i]);
    }
    return blocks;
}

extern "C" int cusrl_assemble_gradients(const cusrl_grad_piece_t *pieces, int64_t num_pieces, float *flat,
                                        double *sumsq_partials, void *stream) {
    using namespace cusrl;
    if (num_pieces < 0 || (num_pieces > 0 && (!pieces || !flat))) return CUSRL_E_INVALID;
    int64_t sumsq_base = 0;
    for (int64_t first = 0; first < num_pieces; first += CUSRL_MAX_FIELDS) {
        GradTable tab;
        tab.n = int32_t(num_pieces - first < CUSRL_MAX_FIELDS ? num_pieces - first : CUSRL_MAX_FIELDS);
        int64_t blocks = 0;
        for (int i = 0; i < tab.n; ++i) {
            const cusrl_grad_piece_t &p = pieces[first + i];
            if (p.numel < 0 || p.offset < 0 || p.splits < 0 || (p.splits > 0 && !p.src)) return CUSRL_E_INVALID;
            const int64_t row_stride = p.row_stride > 0 ? p.row_stride : p.numel;
            if (row_stride < p.numel || p.splits > INT32_MAX) return CUSRL_E_INVALID;
            const bool wide = p.splits > kAssembleWideSplits;
            tab.block_start[i] = int32_t(blocks);
            tab.piece[i] = GradPiece{static_cast<const float *>(p.src), p.offset, p.numel, row_stride, int32_t(p.splits), wide};
            blocks += ceil_div(p.numel, wide ? kAssembleWideCols : kAssemblePerBlock);
            if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
        }
        tab.block_start[tab.n] = int32_t(blocks);
        if (blocks == 0) continue;
        if (sumsq_base + blocks > kMaxClipPartials && sumsq_partials) return CUSRL_E_UNSUPPORTED;
        assemble_gradients_kernel<<<uint32_t(blocks), kBlock, 0, as_stream(stream)>>>(tab, flat, sumsq_partials, int(sumsq_base));
        if (int rc = launch_status()) return rc;
        sumsq_base += blocks;
    }
    return 0;
}

extern "C" int64_t cusrl_adam_step_normed_workspace_bytes(void) { return int64_t(sizeof(cusrl::NormedWorkspace)); }

extern "C" int cusrl_adam_step_normed(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step,
                                      const float *lr, int64_t n, double beta1, double beta2, double eps, double weight_decay,
                                      int decoupled_weight_decay, int maximize, const float *norm_grad, int64_t norm_n,
                                      void *workspace, float max_norm, float *norm_out, float *norm_accumulator,
                                      float *step_mirror, uint32_t *ticket, void *stream) {
    using namespace cusrl;
    if (n <= 0 || !param || !grad || !exp_avg || !exp_avg_sq || !step || !lr || !ticket) return CUSRL_E_INVALID;
    if (norm_n <= 0 || !norm_grad || !workspace) return CUSRL_E_INVALID;
    if (!aligned(param, 16) || !aligned(grad, 16) || !aligned(exp_avg, 16) || !aligned(exp_avg_sq, 16) ||
        !aligned(norm_grad, 16) || !aligned(workspace, 8))
        return CUSRL_E_UNSUPPORTED;
    AdamParams a{beta1, beta2, float(eps), float(weight_decay), max_norm, decoupled_weight_decay, maximize, 0};
    // the grid follows the NORM's buffer alone: the launches of a step's windows split it alike and find the same norm, bit for bit
    // — and is at most HALF a block per CU of the device this call runs on (a partitioned part has 32 CUs, not 256): the two
    // launches of an unjoined step meet inside themselves side by side, and every block of both must find a place while the others
    // wait (a CU holds eight such blocks; unrelated kernels drain, a waiting block does not)
    int device = 0, cus = 0;
    if (hipGetDevice(&device) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 2)
        cus = 2;
    const int64_t most = cus / 2 < kNormedMaxBlocks ? cus / 2 : kNormedMaxBlocks;
    int64_t blocks = ceil_div(norm_n / 4 > 0 ? norm_n / 4 : 1, kBlock);
    if (blocks > most) blocks = most;
    adam_step_normed_kernel<<<int(blocks), kBlock, 0, as_stream(stream)>>>(
        param, grad, exp_avg, exp_avg_sq, step, lr, norm_grad, norm_n, static_cast<NormedWorkspace *>(workspace), norm_out,
        norm_accumulator, step_mirror, ticket, n, a);
    return launch_status();
}
