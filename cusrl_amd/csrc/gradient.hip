// Gradient-norm clipping of the flat gradient buffer (a14 neighbourhood: runs between the gradient all-reduce and
// the optimizer step).  Counterpart of hook/on_policy/gradient_clipping.py:67-83, which calls
// torch.nn.utils.clip_grad_norm_: total = ||g||_2, g *= min(max_norm / (total + 1e-6), 1).  As torch ops that is
// norm + add + reciprocal + mul + clamp + mul = six launches over a 370 KB buffer; here two: block partials of the
// squared sum, then every block re-derives the (uniform) coefficient from the partials and scales its slice.
#include "common.hpp"

namespace cusrl {

constexpr int kNormMaxBlocks = 64;           // partials fit one wave's lanes in the second kernel
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
    if (threadIdx.x < kWave) {  // wave 0: fixed-order sum of <= 64 partials
        double p = int(threadIdx.x) < num_partials ? partials[threadIdx.x] : 0.0;
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
