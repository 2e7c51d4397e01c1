// Backward of the FIRST layer of the actor-critic MLPs for gfx950 (round 6): y = relu(x W^T + b) with an input that needs no
// gradient (the observation) — torch.nn.Linear / ReLU backward of cusrl/nn/module/mlp.py:89-90 at the bottom of the stack.
//
// What the reference's autograd runs there: threshold_backward (read g, y; write g'), sum(0) of g' (the bias gradient) and
// the weight-gradient GEMM g'^T x.  Rounds 2-5 had the first two as one pass (cusrl_relu_bwd_colsum) in front of a
// split-batch rocBLAS GEMM: g and y read once, g' written once and read back by the GEMM — 75 MB + 30 MB for the
// [24576, 256] x [24576, 48] layer of BASELINE config 2, 20 + 17 us inside a step, the largest hand-written launch of the
// minibatch step.  dX is not needed here, so NOTHING has to be written back: this kernel streams g, y and x ONCE, masks in
// registers, and accumulates dW (and db, as the product with a column of ones) on the matrix cores — 55 MB read, a few MB
// of slab partials written.  The contraction runs over the ROWS of the minibatch, i.e. it is a reduction kernel that
// happens to use MFMA for its arithmetic (0.6 GFLOP: 5 us of v_mfma_f32_16x16x4_f32 spread under the 55 MB stream), not a
// GEMM tiling — HBM / L2 bound like every other kernel of the path.
//
// Layout.  A block owns ONE group of 64 output columns h and a range of rows; its 8 waves take alternate 16-row chunks of
// that range (two waves per SIMD: one's MFMA burst — 16 rows = 64 instructions = 2048 cycles — runs under the other's loads).
// Per step of 4 rows a lane (i = lane & 15, kk = lane >> 4) loads ONE float4 of g and of y — row r0 + kk, columns
// 64 cg + 4 i .. + 3: the wave's load is four 256-byte row segments — and one float4 of x (row r0 + kk, features 4 i .. + 3;
// lane i = K / 4 supplies (1, 0, 0, 0): the column of ones whose product is the bias gradient).  The four components of the g
// load are the A operands of four MFMA tiles (tile t holds columns 64 cg + 4 i + t), the four components of the x load the
// B operands of four tiles (tile u holds features 4 j + u): 16 v_mfma_f32_16x16x4_f32 per 4 rows and wave, 64 accumulator
// registers.  A[i][kk] = g[r0 + kk][h(i, t)], B[kk][j] = x[r0 + kk][f(j, u)], D[i][j] += sum_kk A[i][kk] B[kk][j]  =>  dW[h][f].
//
// Reduction, two launches, fixed order.  (1) The 8 waves of a block park their 16 tiles in LDS (128 KB) and every thread adds
// up two tiles over the waves: the block's [64, K + 1] piece of partial row `row block`.  (2) A second, tiny launch
// (input_layer_sum_rows_kernel, same stream) adds the P partial rows up into the ONE [H * K + H] row that
// cusrl_assemble_gradients copies into the parameters' slots.  (The first version did step 2 inside the kernel — the last
// block of a run to finish, found by a ticket, added the run up: correct, and 2x slower than the whole rest of the kernel,
// because the agent-scope release / acquire fences that must bracket the ticket write back and invalidate the XCD's whole
// L2, once per block: 28.6 us with it, 15.2 us without, profiles/r06/input_layer_ab.txt.  A kernel boundary is the cheaper
// fence on an 8-XCD part.)
#include "common.hpp"

namespace cusrl {

typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kInColsPerWave = 64;

__device__ __forceinline__ float4 masked(const float4 &g, const float4 &y) {
    return make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
}

constexpr int kInWaves = 8;
constexpr int kInThreads = kInWaves * kWave;  // 512
constexpr int kInQuads = 4;                   // row quads (4 rows each) requested before anything is computed: 16 rows
constexpr int kInChunk = 4 * kInQuads;        // rows of one chunk

// kMask: y != NULL (ReLU behind the layer).
template <bool kMask>
__global__ __launch_bounds__(kInThreads) void input_layer_bwd_kernel(const float *__restrict__ grad, const float *__restrict__ output,
                                                                     const float *__restrict__ input, int64_t rows, int H, int K,
                                                                     int rows_per_block, float *__restrict__ partials,
                                                                     int64_t stride) {
    __shared__ float4 exchange[kInWaves][16][kWave];  // every wave's 16 tiles: 128 KB
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int i = lane & 15, kk = lane >> 4;
    const int cg = blockIdx.y;  // this block's group of 64 output columns
    const int k4 = K / 4, h4 = H / 4;
    const int64_t row0 = int64_t(blockIdx.x) * rows_per_block;
    const int64_t row_end = min(row0 + rows_per_block, rows);
    mfma_f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = mfma_f32x4{0.f, 0.f, 0.f, 0.f};
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(grad) + cg * (kInColsPerWave / 4) + i;
    const float4 *__restrict__ y4 = kMask ? reinterpret_cast<const float4 *>(output) + cg * (kInColsPerWave / 4) + i : nullptr;
    const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(input) + (i < k4 ? i : 0);
    for (int64_t r0 = row0 + int64_t(wave) * kInChunk; r0 < row_end; r0 += int64_t(kInWaves) * kInChunk) {
        float4 g[kInQuads], y[kInQuads], x[kInQuads];
        bool live[kInQuads];
#pragma unroll
        for (int q = 0; q < kInQuads; ++q) {  // unpredicated loads: rows past the end re-read the block's last row
            const int64_t r = r0 + 4 * q + kk;
            live[q] = r < row_end;
            const int64_t c = live[q] ? r : row_end - 1;
            g[q] = g4[c * h4];
            if (kMask) y[q] = y4[c * h4];
            x[q] = x4[c * k4];
        }
#pragma unroll
        for (int q = 0; q < kInQuads; ++q) {
            float4 a = kMask ? masked(g[q], y[q]) : g[q];
            if (!live[q]) a = make_float4(0.f, 0.f, 0.f, 0.f);
            // lanes i < K/4 carry four features, lane i == K/4 the column of ones, the rest nothing
            const float4 b = i < k4 ? x[q] : make_float4(i == k4 ? 1.f : 0.f, 0.f, 0.f, 0.f);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[u], acc[t][u], 0, 0, 0);
        }
    }
    // ---- (1) the block's piece: every thread adds two tiles (t = wave / 2, u = 2 (wave % 2) + {0, 1}) up over the 8 waves
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            exchange[wave][4 * t + u][lane] = make_float4(acc[t][u][0], acc[t][u][1], acc[t][u][2], acc[t][u][3]);
    __syncthreads();
    {
        const int t = wave >> 1, u0 = 2 * (wave & 1);
        float4 lo = exchange[0][4 * t + u0][lane], hi = exchange[0][4 * t + u0 + 1][lane];
#pragma unroll
        for (int w = 1; w < kInWaves; ++w) {
            const float4 a = exchange[w][4 * t + u0][lane], b = exchange[w][4 * t + u0 + 1][lane];
            lo.x += a.x, lo.y += a.y, lo.z += a.z, lo.w += a.w;
            hi.x += b.x, hi.y += b.y, hi.z += b.z, hi.w += b.w;
        }
        // D element (row = 4 kk + reg, col = lane & 15) of tile (t, u) is dW[h][f] with h = 64 cg + 4 (4 kk + reg) + t, f = 4 i + u;
        // f == K (lane i == K/4, u == 0) is the bias gradient of column h
        float *__restrict__ mine = partials + int64_t(blockIdx.x) * stride;
        const float lo_r[4] = {lo.x, lo.y, lo.z, lo.w}, hi_r[4] = {hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int h = cg * kInColsPerWave + 4 * (4 * kk + reg) + t;
            if (i < k4)
                *reinterpret_cast<float2 *>(mine + int64_t(h) * K + 4 * i + u0) = make_float2(lo_r[reg], hi_r[reg]);
            else if (i == k4 && u0 == 0)
                mine[int64_t(H) * K + h] = lo_r[reg];
        }
    }
}

// (2) out[e] = sum over the P partial rows, in row order: a block is 16 float4 columns x 16 row groups; a thread requests its
// group's rows four at a time, the 16 groups meet in LDS.
constexpr int kInReduceCols = 16, kInReduceGroups = kBlock / kInReduceCols;

__global__ __launch_bounds__(kBlock) void input_layer_sum_rows_kernel(const float *__restrict__ partials, int num_rows, int64_t stride,
                                                                    int64_t width, float *__restrict__ out) {
    __shared__ float4 meet[kInReduceGroups][kInReduceCols];
    const int c = threadIdx.x % kInReduceCols, g = threadIdx.x / kInReduceCols;
    const int64_t e = (int64_t(blockIdx.x) * kInReduceCols + c) * 4;
    const int per_group = (num_rows + kInReduceGroups - 1) / kInReduceGroups;  // (ceil_div is a host helper)
    const int first = g * per_group, end = min(first + per_group, num_rows);
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < width) {
        for (int r = first; r < end; r += 4) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4 *>(partials + int64_t(min(r + k, end - 1)) * stride + e);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (r + k < end) total.x += v[k].x, total.y += v[k].y, total.z += v[k].z, total.w += v[k].w;
        }
    }
    meet[g][c] = total;
    __syncthreads();
    if (g == 0 && e < width) {
        float4 sum = meet[0][c];
#pragma unroll
        for (int k = 1; k < kInReduceGroups; ++k) {
            const float4 v = meet[k][c];
            sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + e) = sum;
    }
}

// rows one block walks: enough blocks (row blocks x column groups) to put bytes in flight on every CU, few enough partial rows
// that their traffic stays a few MB
static int input_layer_rows_per_block(int64_t rows, int64_t H) {
    const int64_t column_groups = H / kInColsPerWave;
    int64_t row_blocks = ceil_div(256, column_groups);                      // fill the chip ...
    if (ceil_div(rows, 512) > row_blocks) row_blocks = ceil_div(rows, 512);  // ... and keep a block's walk short
    const int64_t most = ceil_div(rows, kInChunk);                           // at least one chunk per block
    if (row_blocks > most) row_blocks = most;
    return int(ceil_div(ceil_div(rows, row_blocks), kInChunk) * kInChunk);
}

}  // namespace cusrl

using namespace cusrl;

extern "C" int cusrl_input_layer_supported(int64_t in_features, int64_t out_features) {
    return in_features >= 4 && in_features % 4 == 0 && in_features / 4 <= 15 && out_features >= kInColsPerWave &&
           out_features % kInColsPerWave == 0 && out_features <= 4096;
}

extern "C" int64_t cusrl_input_layer_row_blocks(int64_t rows, int64_t out_features) {
    return rows <= 0 || out_features < kInColsPerWave ? 0 : ceil_div(rows, input_layer_rows_per_block(rows, out_features));
}

extern "C" int cusrl_input_layer_bwd(const float *grad_out, const float *output, const float *input, int64_t rows,
                                     int64_t in_features, int64_t out_features, float *partials, float *grads, void *stream) {
    if (rows <= 0 || !grad_out || !input || !partials || !grads) return CUSRL_E_INVALID;
    if (!cusrl_input_layer_supported(in_features, out_features)) return CUSRL_E_UNSUPPORTED;
    if (!aligned(grad_out, 16) || !aligned(input, 16) || (output && !aligned(output, 16)) || !aligned(partials, 16) ||
        !aligned(grads, 16))
        return CUSRL_E_UNSUPPORTED;
    const int per_block = input_layer_rows_per_block(rows, out_features);
    const int64_t blocks = ceil_div(rows, per_block);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const int64_t width = out_features * in_features + out_features;  // one partial row: dW [H, K] | db [H]
    const dim3 grid(uint32_t(blocks), uint32_t(out_features / kInColsPerWave));
    hipStream_t s = as_stream(stream);
    if (output)
        hipLaunchKernelGGL((input_layer_bwd_kernel<true>), grid, dim3(kInThreads), 0, s, grad_out, output, input, rows,
                           int(out_features), int(in_features), per_block, partials, width);
    else
        hipLaunchKernelGGL((input_layer_bwd_kernel<false>), grid, dim3(kInThreads), 0, s, grad_out, output, input, rows,
                           int(out_features), int(in_features), per_block, partials, width);
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(input_layer_sum_rows_kernel, dim3(uint32_t(ceil_div(width / 4, kInReduceCols))), dim3(kBlock), 0, s, partials,
                       int(blocks), width, width, grads);
    return launch_status();
}
