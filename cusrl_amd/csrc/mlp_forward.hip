// Inference pass of the actor / critic MLP for gfx950 (round 6): out = W3 relu(W2 relu(W1 x + b1) + b2) + b3 — the
// Linear / ReLU stack of cusrl/nn/module/mlp.py:74-93 behind a distribution head (cusrl/nn/module/actor.py:69-92,
// distribution.py:256-262) or a value head (critic.py:87-88) — as ONE launch, optionally with the acting path's sampling
// epilogue (Normal.rsample + log_prob, distribution.py:195-218) behind it.
//
// Why hand-written.  Acting on 4096 envs is three dependent library GEMMs of 0.05 / 0.27 / 0.01 GFLOP (7.4 + 8.0 + 4.7 us
// inside the captured env step, launch- and tail-bound: the 256 -> 128 layer runs 128 workgroups on 256 CUs) plus the sampling
// launch (4.2 us): 24 us of the 37 us a captured env step takes, 24 times per iteration.  The whole stack for 16 rows is
// 0.75 MFLOP and 182 KB of weights that every row tile re-uses: one workgroup per 16 rows keeps every intermediate on chip and
// streams the weights ONCE into registers.
//
// Formulation (transposed: features x rows).  A layer is H^T = W X^T, so that with v_mfma_f32_16x16x4_f32
//   A[i][kk] = W[16 tile + i][k],  B[kk][m] = X^T[k][m],  D[4 kk + reg][m] = H^T[16 tile + 4 kk + reg][m]    (i = m = lane & 15, kk = lane >> 4)
// and the FOUR k values a lane supplies to four consecutive MFMAs chosen as k = 16 t + 4 kk + j (j = 0..3; any assignment of
// k to (instruction, kk) is a valid order of the sum as long as A and B agree):
//   * the A operands of k-block t are ONE float4 of the weight row — W[row][16 t + 4 kk .. + 3], a plain row-major load;
//   * the B operands of the NEXT layer's k-block t are the four accumulator registers the same lane holds of THIS layer's
//     output tile t — D[4 kk + j][m] is H^T[16 t + 4 kk + j][m] — i.e. a layer's result feeds the next layer without being
//     moved, transposed or even re-indexed.
// The 4 waves of a workgroup split a layer's OUTPUT features (wave w owns tiles T w .. T w + T - 1): layer 1's tiles meet in
// LDS as [tile][lane] float4 (written and read with the same lane index: conflict-free 16-byte accesses) because layer 2 needs
// all of them; the head contracts over layer 2's features, so every wave multiplies the tiles it already holds by its slice
// of the head's weight and only the four [out, 16] partial sums meet in LDS.  Two barriers per row tile.
//
// Persistent weights: a wave loads its slices of W1 / W2 / W3 and the biases ONCE (48 + 128 + 8 + 28 registers for the preset's
// 48 -> 256 -> 128 -> 12 actor) and walks row tiles with stride gridDim.x: the acting launch is one tile per workgroup, the
// passes over the whole buffer (value targets in pre_update, the statistics pass) re-use the registers for 24+ tiles.
//
// Numerics: fp32 MFMA accumulation (the instruction the library's fp32 GEMMs use); the summation order differs from the
// library's, results agree to a few ulp of the accumulated magnitude.  The sampling epilogue evaluates action and log-prob with
// the expressions and the order of cusrl_normal_sample_logp (rollout.hip), so given the same mean it is bit-identical to it.
#include "common.hpp"

namespace cusrl {

typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMlpWaves = 4;
constexpr int kMlpThreads = kMlpWaves * kWave;
constexpr int kMlpRows = 16;  // rows of one tile (the N dimension of the MFMA)

struct MlpForwardArgs {
    const float *x;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    float *out;        // [rows, A]: head output incl. bias (NULL: not stored — only legal with the sampling epilogue's mean_out)
    const float *std;  // sampling epilogue (all NULL: none): [A] vector
    const float *eps;  // [rows, A]
    float *action, *logp, *std_out;
    int64_t rows;
    int K, A;
};

__device__ __forceinline__ float log_sqrt_2pi_f() { return 0.91893853320467274178f; }

// relu as torch evaluates it: a NaN stays a NaN (v <= 0 is false for it), -0 becomes +0
__device__ __forceinline__ mfma_f32x4 relu4(mfma_f32x4 v) {
    v[0] = v[0] <= 0.f ? 0.f : v[0], v[1] = v[1] <= 0.f ? 0.f : v[1], v[2] = v[2] <= 0.f ? 0.f : v[2], v[3] = v[3] <= 0.f ? 0.f : v[3];
    return v;
}

// T1 / T2: 16-feature tiles of the first / second hidden layer per wave (H1 = 64 T1, H2 = 64 T2); KB: 16-wide k-blocks of the
// input (K <= 16 KB, K % 4 == 0).
template <int T1, int T2, int KB>
__global__ __launch_bounds__(kMlpThreads) void mlp2_forward_kernel(const MlpForwardArgs p) {
    constexpr int NT1 = kMlpWaves * T1;  // k-blocks of layer 2 = tiles of layer 1
    __shared__ float4 hidden[NT1][kWave];
    __shared__ float4 partial[kMlpWaves][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int i = lane & 15, kk = lane >> 4;
    const int K = p.K, A = p.A;
    constexpr int H1 = 16 * NT1, H2 = 16 * kMlpWaves * T2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- this wave's weights, once
    float4 a1[T1][KB], a2[T2][NT1], a3[T2];
    mfma_f32x4 c1[T1], c2[T2], c3;
#pragma unroll
    for (int t1 = 0; t1 < T1; ++t1) {
        const int n = 16 * (T1 * wave + t1);
#pragma unroll
        for (int t = 0; t < KB; ++t) {
            // (unpredicated 16-byte load from a clamped address, then a select: a predicated load is split into four guarded ones)
            const int k = 16 * t + 4 * kk;
            const float4 v = *reinterpret_cast<const float4 *>(p.w1 + int64_t(n + i) * K + (k < K ? k : 0));
            a1[t1][t] = k < K ? v : zero4;
        }
        const float4 b = *reinterpret_cast<const float4 *>(p.b1 + n + 4 * kk);
        c1[t1] = mfma_f32x4{b.x, b.y, b.z, b.w};
    }
    const int64_t tiles = (p.rows + kMlpRows - 1) / kMlpRows;
    // the input rows of a tile are requested one tile ahead: with one workgroup per CU nothing else hides the load's latency
    float4 ahead[KB];
    auto request = [&](int64_t tile) {
        const int64_t row = min(tile * kMlpRows + i, p.rows - 1);  // (rows past the end re-read the last row; nothing is stored for them)
#pragma unroll
        for (int t = 0; t < KB; ++t) {
            const int k = 16 * t + 4 * kk;
            ahead[t] = *reinterpret_cast<const float4 *>(p.x + row * K + (k < K ? k : 0));
        }
    };
    // (the first tile's rows in front of the second layer's weights: loads return in order, and layer 1 needs only these)
    request(min(int64_t(blockIdx.x), tiles - 1));
#pragma unroll
    for (int t2 = 0; t2 < T2; ++t2) {
        const int n = 16 * (T2 * wave + t2);
#pragma unroll
        for (int t = 0; t < NT1; ++t) a2[t2][t] = *reinterpret_cast<const float4 *>(p.w2 + int64_t(n + i) * H1 + 16 * t + 4 * kk);
        const float4 b = *reinterpret_cast<const float4 *>(p.b2 + n + 4 * kk);
        c2[t2] = mfma_f32x4{b.x, b.y, b.z, b.w};
        // the head: rows = outputs (i < A), k-block = this wave's own tile t2 of the second hidden layer
        const float4 v = *reinterpret_cast<const float4 *>(p.w3 + int64_t(i < A ? i : 0) * H2 + n + 4 * kk);
        a3[t2] = i < A ? v : zero4;
    }
    c3 = mfma_f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave == 0 && p.b3) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) c3[reg] = 4 * kk + reg < A ? p.b3[4 * kk + reg] : 0.f;
    }

    // sampling epilogue (wave 0): this lane's four std values once, its four eps values requested at the top of every tile
    float sigma[4] = {1.f, 1.f, 1.f, 1.f}, noise[4] = {0.f, 0.f, 0.f, 0.f};
    const bool sampling = wave == 0 && p.eps != nullptr;
    if (sampling) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) sigma[reg] = p.std[min(4 * kk + reg, A - 1)];
    }

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row0 = tile * kMlpRows;
        if (sampling) {
            const int64_t first = min(row0 + i, p.rows - 1) * A;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) noise[reg] = p.eps[first + min(4 * kk + reg, A - 1)];
        }
        float4 xb[KB];
#pragma unroll
        for (int t = 0; t < KB; ++t) xb[t] = 16 * t + 4 * kk < K ? ahead[t] : zero4;
        request(min(tile + gridDim.x, tiles - 1));
        // ---- layer 1: this wave's T1 tiles of H1^T, bias in the accumulator, ReLU, to LDS
#pragma unroll
        for (int t1 = 0; t1 < T1; ++t1) {
            mfma_f32x4 acc = c1[t1];
#pragma unroll
            for (int t = 0; t < KB; ++t) {
                const float av[4] = {a1[t1][t].x, a1[t1][t].y, a1[t1][t].z, a1[t1][t].w};
                const float bv[4] = {xb[t].x, xb[t].y, xb[t].z, xb[t].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
            }
            acc = relu4(acc);
            hidden[T1 * wave + t1][lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        // ---- layer 2: this wave's T2 tiles of H2^T over all NT1 k-blocks
        mfma_f32x4 h2[T2];
#pragma unroll
        for (int t2 = 0; t2 < T2; ++t2) h2[t2] = c2[t2];
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const float4 hb = hidden[t][lane];
            const float bv[4] = {hb.x, hb.y, hb.z, hb.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t2 = 0; t2 < T2; ++t2) {
                    const float av = j == 0 ? a2[t2][t].x : j == 1 ? a2[t2][t].y : j == 2 ? a2[t2][t].z : a2[t2][t].w;
                    h2[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], h2[t2], 0, 0, 0);
                }
        }
        // ---- head: partial product over this wave's own k-blocks (its T2 tiles), the four partials meet in LDS
        mfma_f32x4 head = c3;
#pragma unroll
        for (int t2 = 0; t2 < T2; ++t2) {
            const mfma_f32x4 h = relu4(h2[t2]);
            const float av[4] = {a3[t2].x, a3[t2].y, a3[t2].z, a3[t2].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) head = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], h[j], head, 0, 0, 0);
        }
        partial[wave][lane] = make_float4(head[0], head[1], head[2], head[3]);
        __syncthreads();
        if (wave == 0) {
            // out^T[a = 4 kk + reg][m = i]: waves added in wave order (fixed)
            float4 sum = partial[0][lane];
#pragma unroll
            for (int w = 1; w < kMlpWaves; ++w) {
                const float4 v = partial[w][lane];
                sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
            }
            const bool live = row0 + i < p.rows;
            const int a0 = 4 * kk;
            const int64_t at = (row0 + i) * A + a0;
            const float mu[4] = {sum.x, sum.y, sum.z, sum.w};
            if (live && p.out) {
                if (A % 4 == 0) {
                    if (a0 < A) *reinterpret_cast<float4 *>(p.out + at) = sum;
                } else {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        if (a0 + reg < A) p.out[at + reg] = mu[reg];
                }
            }
            if (p.eps) {
                // Normal.rsample + log_prob with the expressions of normal_sample_logp_kernel (rollout.hip): a lane evaluates its
                // four actions, lane kk == 0 of a row adds the row's terms up in action order (its own, then the other three
                // lanes' through shuffles): the association of the one-lane-per-row kernel
                float term[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int a = a0 + reg;
                    if (a < A) {
                        const float sg = sigma[reg];
                        const float act = mu[reg] + noise[reg] * sg;
                        const float diff = act - mu[reg];
                        term[reg] = -(diff * diff) / (2.0f * (sg * sg)) - logf(sg) - log_sqrt_2pi_f();
                        if (live) {
                            p.action[at + reg] = act;
                            if (p.std_out) p.std_out[at + reg] = sg;
                        }
                    }
                }
                float lp = 0.f;
#pragma unroll
                for (int src = 0; src < 4; ++src)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float t = __shfl(term[reg], i + 16 * src, kWave);
                        if (4 * src + reg < A) lp += t;
                    }
                if (live && kk == 0) p.logp[row0 + i] = lp;
            }
        }
    }
}

template <int T1, int T2>
static int launch_mlp2(const MlpForwardArgs &p, int kb, dim3 grid, hipStream_t s) {
    switch (kb) {
        case 1: hipLaunchKernelGGL((mlp2_forward_kernel<T1, T2, 1>), grid, dim3(kMlpThreads), 0, s, p); break;
        case 2: hipLaunchKernelGGL((mlp2_forward_kernel<T1, T2, 2>), grid, dim3(kMlpThreads), 0, s, p); break;
        case 3: hipLaunchKernelGGL((mlp2_forward_kernel<T1, T2, 3>), grid, dim3(kMlpThreads), 0, s, p); break;
        case 4: hipLaunchKernelGGL((mlp2_forward_kernel<T1, T2, 4>), grid, dim3(kMlpThreads), 0, s, p); break;
        default: return CUSRL_E_UNSUPPORTED;
    }
    return launch_status();
}

}  // namespace cusrl

using namespace cusrl;

extern "C" int cusrl_mlp2_forward_supported(int64_t in_features, int64_t hidden1, int64_t hidden2, int64_t out_features) {
    return in_features >= 4 && in_features % 4 == 0 && in_features <= 64 && (hidden1 == 128 || hidden1 == 256) &&
           (hidden2 == 64 || hidden2 == 128) && out_features >= 1 && out_features <= 16;
}

extern "C" int cusrl_mlp2_forward(const float *input, int64_t rows, int64_t in_features, const float *w1, const float *b1,
                                  int64_t hidden1, const float *w2, const float *b2, int64_t hidden2, const float *w3,
                                  const float *b3, int64_t out_features, float *output, const float *std, const float *eps,
                                  float *action, float *logp, float *std_out, void *stream) {
    if (rows < 0 || !input || !w1 || !b1 || !w2 || !b2 || !w3) return CUSRL_E_INVALID;
    if (rows == 0) return 0;
    if (!cusrl_mlp2_forward_supported(in_features, hidden1, hidden2, out_features)) return CUSRL_E_UNSUPPORTED;
    const bool sampling = eps != nullptr;
    if (sampling ? !(std && action && logp) : (std || action || logp || std_out)) return CUSRL_E_INVALID;
    if (!sampling && !output) return CUSRL_E_INVALID;
    if (!aligned(input, 16) || !aligned(w1, 16) || !aligned(b1, 16) || !aligned(w2, 16) || !aligned(b2, 16) || !aligned(w3, 16) ||
        (output && out_features % 4 == 0 && !aligned(output, 16)))
        return CUSRL_E_UNSUPPORTED;
    MlpForwardArgs p;
    p.x = input, p.w1 = w1, p.b1 = b1, p.w2 = w2, p.b2 = b2, p.w3 = w3, p.b3 = b3, p.out = output;
    p.std = std, p.eps = eps, p.action = action, p.logp = logp, p.std_out = std_out;
    p.rows = rows, p.K = int(in_features), p.A = int(out_features);
    const int64_t tiles = ceil_div(rows, kMlpRows);
    // one workgroup (4 waves of ~300 registers: one per SIMD) per CU; beyond that the tiles are walked with the weights in place
    // (a second round of workgroups would pay the weight loads again: 20.5 us instead of 14 us for 512 tiles)
    static int cus = 0;
    if (cus == 0) {
        int device = 0, count = 0;
        if (hipGetDevice(&device) != hipSuccess || hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || count <= 0)
            count = 256;
        cus = count;
    }
    const dim3 grid(uint32_t(tiles < cus ? tiles : cus));
    const int kb = int(ceil_div(in_features, 16));
    hipStream_t s = as_stream(stream);
    if (hidden1 == 256 && hidden2 == 128) return launch_mlp2<4, 2>(p, kb, grid, s);
    if (hidden1 == 256 && hidden2 == 64) return launch_mlp2<4, 1>(p, kb, grid, s);
    if (hidden1 == 128 && hidden2 == 128) return launch_mlp2<2, 2>(p, kb, grid, s);
    return launch_mlp2<2, 1>(p, kb, grid, s);
}
