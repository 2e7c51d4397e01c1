// Backward epilogues of the actor-critic MLP for gfx950: ReLU mask + bias gradient (column sums) in one pass.
// autograd runs threshold_backward (read g, y; write g') and then sum(0) (read g') as two kernels — 23.8 + 11.7 us
// for a [24576, 256] layer; here g and y are read once, g' written once, and the column sums ride along.
#include <stdlib.h>

#include "common.hpp"

namespace cusrl {

// rows of the matrix one block reduces (64: 24576 rows -> 384 blocks); cusrl_set_option("colsum_rows", n) overrides it for sweeps
static int col_rows_per_block() {
    const int v = int(option(kOptColsumRows));
    return v >= 4 && v <= 4096 ? v : 64;
}
constexpr int kColBatch = 4;          // row passes whose loads are issued together

__device__ __forceinline__ void pin4(float4 (&r)[kColBatch]) {
    asm volatile(""
                 : "+v"(r[0].x), "+v"(r[0].y), "+v"(r[0].z), "+v"(r[0].w), "+v"(r[1].x), "+v"(r[1].y), "+v"(r[1].z),
                   "+v"(r[1].w), "+v"(r[2].x), "+v"(r[2].y), "+v"(r[2].z), "+v"(r[2].w), "+v"(r[3].x), "+v"(r[3].y),
                   "+v"(r[3].z), "+v"(r[3].w));
}

// H % 4 == 0 and (kBlock % (H/4) == 0): lanes stream 16 B chunks; a lane always lands on the same column group.
// Loads are unpredicated (rows past the end are clamped and contribute 0) and issued kColBatch passes at a time.
template <bool kMask>
__global__ __launch_bounds__(kBlock) void colsum_chunked_kernel(const float *__restrict__ grad,
                                                                const float *__restrict__ output,
                                                                float *__restrict__ grad_in,
                                                                float *__restrict__ partials, int64_t rows, int H,
                                                                int rows_per_block) {
    const int lpr = H / 4;                   // 16 B chunks per row
    const int rows_per_pass = kBlock / lpr;  // rows covered by the block per pass
    const int col = threadIdx.x % lpr, sub = threadIdx.x / lpr;
    const int64_t row0 = int64_t(blockIdx.x) * rows_per_block;
    const int64_t row_end = min(row0 + rows_per_block, rows);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t base = row0 + sub; base < row_end; base += int64_t(kColBatch) * rows_per_pass) {
        float4 g[kColBatch], y[kColBatch];
        int64_t q[kColBatch];
        float live[kColBatch];
#pragma unroll
        for (int k = 0; k < kColBatch; ++k) {
            const int64_t r = base + int64_t(k) * rows_per_pass;
            live[k] = r < row_end ? 1.f : 0.f;
            q[k] = min(r, row_end - 1) * lpr + col;
            g[k] = reinterpret_cast<const float4 *>(grad)[q[k]];
            if constexpr (kMask) y[k] = reinterpret_cast<const float4 *>(output)[q[k]];
        }
        pin4(g);
        if constexpr (kMask) pin4(y);
#pragma unroll
        for (int k = 0; k < kColBatch; ++k) {
            float4 v = g[k];
            if constexpr (kMask) {
                v.x = y[k].x > 0.f ? v.x : 0.f;
                v.y = y[k].y > 0.f ? v.y : 0.f;
                v.z = y[k].z > 0.f ? v.z : 0.f;
                v.w = y[k].w > 0.f ? v.w : 0.f;
                reinterpret_cast<float4 *>(grad_in)[q[k]] = v;  // clamped duplicates rewrite identical bytes
            }
            acc.x += live[k] * v.x, acc.y += live[k] * v.y, acc.z += live[k] * v.z, acc.w += live[k] * v.w;
        }
    }
    // combine the kBlock / lpr sub-rows through LDS (fixed order)
    __shared__ float4 red[kBlock];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < lpr) {
        float4 total = red[threadIdx.x];
        for (int s = 1; s < rows_per_pass; ++s) {
            const float4 v = red[s * lpr + threadIdx.x];
            total.x += v.x, total.y += v.y, total.z += v.z, total.w += v.w;
        }
        reinterpret_cast<float4 *>(partials + int64_t(blockIdx.x) * H)[threadIdx.x] = total;
    }
}

// Any H: one lane per row, H accumulators walked column by column through LDS (narrow heads: H = 12, 1, ...).
template <bool kMask>
__global__ __launch_bounds__(kBlock) void colsum_rowwise_kernel(const float *__restrict__ grad,
                                                                const float *__restrict__ output,
                                                                float *__restrict__ grad_in,
                                                                float *__restrict__ partials, int64_t rows, int H) {
    __shared__ float scratch[kWavesPerBlock];
    const int64_t row0 = int64_t(blockIdx.x) * kBlock * 4;
    for (int h = 0; h < H; ++h) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t r = row0 + int64_t(k) * kBlock + threadIdx.x;
            if (r < rows) {
                float g = grad[r * H + h];
                if constexpr (kMask) {
                    g = output[r * H + h] > 0.f ? g : 0.f;
                    grad_in[r * H + h] = g;
                }
                acc += g;
            }
        }
        const float total = block_sum(acc, scratch);
        if (threadIdx.x == 0) partials[int64_t(blockIdx.x) * H + h] = total;
    }
}

// partials [P, H] -> colsum [H].  H % 4 == 0: one block per 64 columns = 16 float4 lanes x 16 row groups; every lane
// walks its partial rows 4 loads at a time (independent, pipelined), groups are combined through LDS in fixed order.
__global__ __launch_bounds__(kBlock) void colsum_finalize_vec_kernel(const float *__restrict__ partials, int64_t P,
                                                                     int H, float *__restrict__ colsum) {
    __shared__ float4 red[kBlock];
    const int c4 = threadIdx.x & 15, group = threadIdx.x >> 4;
    const int h = blockIdx.x * 64 + c4 * 4;
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h < H) {
        const float4 *base = reinterpret_cast<const float4 *>(partials + h);
        const int64_t stride = H / 4;  // float4 elements per partial row
        int64_t p = group;
        for (; p + 48 < P; p += 64) {
            const float4 a = base[p * stride], b = base[(p + 16) * stride], c = base[(p + 32) * stride],
                         d = base[(p + 48) * stride];
            total.x += (a.x + b.x) + (c.x + d.x), total.y += (a.y + b.y) + (c.y + d.y);
            total.z += (a.z + b.z) + (c.z + d.z), total.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; p < P; p += 16) {
            const float4 a = base[p * stride];
            total.x += a.x, total.y += a.y, total.z += a.z, total.w += a.w;
        }
    }
    red[threadIdx.x] = total;
    __syncthreads();
    if (group == 0 && h < H) {
        float4 sum = red[c4];
        for (int g = 1; g < 16; ++g) {
            const float4 v = red[g * 16 + c4];
            sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
        }
        *reinterpret_cast<float4 *>(colsum + h) = sum;
    }
}

// any H: one block per 64 columns; 4 partial-row groups x 64 columns
__global__ __launch_bounds__(kBlock) void colsum_finalize_kernel(const float *__restrict__ partials, int64_t P, int H,
                                                                 float *__restrict__ colsum) {
    __shared__ float red[kBlock];
    const int c = threadIdx.x & 63, group = threadIdx.x >> 6;
    const int h = blockIdx.x * 64 + c;
    float total = 0.f;
    if (h < H)
        for (int64_t p = group; p < P; p += 4) total += partials[p * H + h];
    red[threadIdx.x] = total;
    __syncthreads();
    if (group == 0 && h < H) colsum[h] = (red[c] + red[64 + c]) + (red[128 + c] + red[192 + c]);
}

static bool colsum_chunked(int64_t H) { return H % 4 == 0 && H / 4 <= kBlock && kBlock % (H / 4) == 0; }

}  // namespace cusrl

using namespace cusrl;

extern "C" int64_t cusrl_colsum_num_partials(int64_t rows, int64_t H) {
    if (rows <= 0 || H <= 0) return 0;
    return colsum_chunked(H) ? ceil_div(rows, int64_t(cusrl::col_rows_per_block())) : ceil_div(rows, int64_t(kBlock) * 4);
}

extern "C" int cusrl_relu_bwd_colsum(const float *grad, const float *output, float *grad_in, float *partials,
                                     float *colsum, int64_t rows, int64_t H, void *stream) {
    if (rows <= 0 || H <= 0) return CUSRL_E_INVALID;
    if (!grad || !partials || (output && !grad_in)) return CUSRL_E_INVALID;
    if (H > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    const int64_t P = cusrl_colsum_num_partials(rows, H);
    if (P > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const bool chunked = colsum_chunked(H) && aligned(grad, 16) && (!output || (aligned(output, 16) && aligned(grad_in, 16))) &&
                         aligned(partials, 16);
    if (chunked) {
        if (output)
            hipLaunchKernelGGL(colsum_chunked_kernel<true>, dim3(uint32_t(P)), dim3(kBlock), 0, s, grad, output, grad_in,
                               partials, rows, int(H), col_rows_per_block());
        else
            hipLaunchKernelGGL(colsum_chunked_kernel<false>, dim3(uint32_t(P)), dim3(kBlock), 0, s, grad, output, grad_in,
                               partials, rows, int(H), col_rows_per_block());
    } else {
        if (!colsum) return CUSRL_E_UNSUPPORTED;  // partials-only mode exists for the chunked layout
        // the partial count was sized for the layout chosen by H alone; recompute for the row-wise launch shape
        const int64_t Pr = ceil_div(rows, int64_t(kBlock) * 4);
        if (Pr > P) return CUSRL_E_UNSUPPORTED;
        if (output)
            hipLaunchKernelGGL(colsum_rowwise_kernel<true>, dim3(uint32_t(Pr)), dim3(kBlock), 0, s, grad, output, grad_in,
                               partials, rows, int(H));
        else
            hipLaunchKernelGGL(colsum_rowwise_kernel<false>, dim3(uint32_t(Pr)), dim3(kBlock), 0, s, grad, output,
                               grad_in, partials, rows, int(H));
        if (int rc = launch_status()) return rc;
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(uint32_t(ceil_div(H, 64))), dim3(kBlock), 0, s, partials, Pr,
                           int(H), colsum);
        return launch_status();
    }
    if (int rc = launch_status()) return rc;
    if (!colsum) return 0;  // the caller reduces the P partial rows itself (cusrl_assemble_gradients)
    if (aligned(colsum, 16))
        hipLaunchKernelGGL(colsum_finalize_vec_kernel, dim3(uint32_t(ceil_div(H, 64))), dim3(kBlock), 0, s, partials, P,
                           int(H), colsum);
    else
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(uint32_t(ceil_div(H, 64))), dim3(kBlock), 0, s, partials, P,
                           int(H), colsum);
    return launch_status();
}
