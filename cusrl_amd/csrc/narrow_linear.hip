// Backward of a NARROW linear layer (policy-mean / value heads: out_features O <= 16, e.g. 128 -> 12 and 128 -> 1)
// for gfx950.  autograd issues three library calls for it — dX = dY W (a [B,O]x[O,K] GEMM), dW = dY^T X (a GEMM with
// an O x K output and the whole minibatch as reduction dim) and db = dY.sum(0) — 8 + 19 + 6 us at B = 24576, K = 128,
// each 3-7x above its HBM floor because an O <= 16 wide operand cannot fill an MFMA tile.  With O this small the
// products are plain FMAs on data already in registers, so ONE pass does all three: a lane owns one 16 B chunk of
// the K axis, keeps its O x 4 slice of W and of the dW accumulator in VGPRs, streams X rows (coalesced float4,
// 4 rows in flight), writes dX and accumulates dW / db.  HBM traffic = X read + dX write (+ dY), the floor.
#include <stdlib.h>

#include "common.hpp"

namespace cusrl {

// rows of the minibatch one block reduces: 96 -> a 24576-row minibatch is 256 blocks, one per CU (cusrl_set_option("head_rows", n)
// overrides it for sweeps: profiles/r04/narrow_head_rows.txt)
static int head_rows_per_block() {
    const int v = int(option(kOptHeadRows));
    return v >= 8 && v <= 4096 ? v : 96;
}
constexpr int head_batch(int O) { return O > 8 ? 2 : 4; }  // X rows in flight per lane (VGPR budget: O x 12 + ...)
constexpr int kHeadBiasPad = 16;  // db rides behind dW in the same partial row, padded to keep float4 alignment

// kReluInput: the layer's input is a ReLU output (x > 0 <=> the ReLU passed), so the kernel also plays the ReLU's
// backward for the producer: grad_input comes out already masked and its column sums (= the producer layer's bias
// gradient) ride along as one more accumulator slice — the separate mask + column-sum pass over [rows, K] disappears.
template <int O, bool kReluInput>
__global__ __launch_bounds__(kBlock) void narrow_linear_bwd_kernel(const float *__restrict__ grad_out,
                                                                   const float *__restrict__ input,
                                                                   const float *__restrict__ weight,
                                                                   float *__restrict__ grad_input,
                                                                   float *__restrict__ partials, int64_t rows, int K,
                                                                   int rows_per_block) {
    extern __shared__ float4 red[];  // [groups][O + 1][lpr] dW (+ masked-dX column sum) slices, then [groups][16] db
    constexpr int SL = O + 1;        // accumulator slices per lane: O rows of dW and the column sums of the masked dX
    const int lpr = K / 4;
    const int groups = kBlock / lpr;
    const int col = threadIdx.x % lpr, sub = threadIdx.x / lpr;
    const int64_t row0 = int64_t(blockIdx.x) * rows_per_block;
    const int64_t row_end = min(row0 + rows_per_block, rows);

    constexpr int kHeadBatch = head_batch(O);
    float4 w[O], dw[SL];
    float db[O];  // every lane of a row group reads the same dY row: each keeps the group's db sums (O adds per row)
#pragma unroll
    for (int o = 0; o < O; ++o) w[o] = reinterpret_cast<const float4 *>(weight)[o * lpr + col], db[o] = 0.f;
#pragma unroll
    for (int o = 0; o < SL; ++o) dw[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t base = row0 + sub; base < row_end; base += int64_t(kHeadBatch) * groups) {
        float4 x[kHeadBatch];
        float g[kHeadBatch][O];
        int64_t r[kHeadBatch];
        bool live[kHeadBatch];
#pragma unroll
        for (int k = 0; k < kHeadBatch; ++k) {  // all loads of the batch first (rows past the end: clamped, zeroed)
            const int64_t row = base + int64_t(k) * groups;
            live[k] = row < row_end;
            r[k] = min(row, row_end - 1);
            x[k] = reinterpret_cast<const float4 *>(input)[r[k] * lpr + col];
            if constexpr (O % 4 == 0) {  // a dY row is a whole number of 16 B chunks (same address in every lane)
#pragma unroll
                for (int o4 = 0; o4 < O / 4; ++o4) {
                    const float4 v = reinterpret_cast<const float4 *>(grad_out)[r[k] * (O / 4) + o4];
                    g[k][o4 * 4 + 0] = v.x, g[k][o4 * 4 + 1] = v.y, g[k][o4 * 4 + 2] = v.z, g[k][o4 * 4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int o = 0; o < O; ++o) g[k][o] = grad_out[r[k] * O + o];
            }
        }
#pragma unroll
        for (int k = 0; k < kHeadBatch; ++k) {
            float4 dx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const float go = live[k] ? g[k][o] : 0.f;
                db[o] += go;
                dx.x = fmaf(go, w[o].x, dx.x), dx.y = fmaf(go, w[o].y, dx.y);
                dx.z = fmaf(go, w[o].z, dx.z), dx.w = fmaf(go, w[o].w, dx.w);
                dw[o].x = fmaf(go, x[k].x, dw[o].x), dw[o].y = fmaf(go, x[k].y, dw[o].y);
                dw[o].z = fmaf(go, x[k].z, dw[o].z), dw[o].w = fmaf(go, x[k].w, dw[o].w);
            }
            if constexpr (kReluInput) {  // rows past the end have dx == 0: they add nothing to the sums
                dx.x = x[k].x > 0.f ? dx.x : 0.f, dx.y = x[k].y > 0.f ? dx.y : 0.f;
                dx.z = x[k].z > 0.f ? dx.z : 0.f, dx.w = x[k].w > 0.f ? dx.w : 0.f;
                dw[O].x += dx.x, dw[O].y += dx.y, dw[O].z += dx.z, dw[O].w += dx.w;
            }
            if (grad_input && live[k]) reinterpret_cast<float4 *>(grad_input)[r[k] * lpr + col] = dx;
        }
    }

    // combine the row groups of the block in fixed order (group 0 first): every group parks its slices in LDS, then the
    // SL x lpr float4 outputs are spread over ALL lanes of the block (2 outputs x `groups` LDS reads per lane at K = 128,
    // O = 12 — round 3 had the 32 lanes of group 0 walk 13 x 7 reads each while the other three waves had retired)
    float *red_bias = reinterpret_cast<float *>(red + groups * SL * lpr);  // [groups][16 outputs]
#pragma unroll
    for (int o = 0; o < SL; ++o) red[(sub * SL + o) * lpr + col] = dw[o];
    if (col == 0) {
#pragma unroll
        for (int o = 0; o < kHeadBiasPad; ++o) red_bias[sub * kHeadBiasPad + o] = 0.f;
#pragma unroll
        for (int o = 0; o < O; ++o) red_bias[sub * kHeadBiasPad + o] = db[o];
    }
    __syncthreads();
    float *out = partials + int64_t(blockIdx.x) * (SL * K + kHeadBiasPad);
    const int slab = SL * lpr;  // float4 outputs of the block (dW rows, then the masked-dX column sums)
    for (int i = threadIdx.x; i < slab; i += kBlock) {
        float4 total = red[i];
#pragma unroll 8
        for (int s = 1; s < groups; ++s) {
            const float4 v = red[s * slab + i];
            total.x += v.x, total.y += v.y, total.z += v.z, total.w += v.w;
        }
        reinterpret_cast<float4 *>(out)[i] = total;
    }
    if (threadIdx.x >= kBlock - kHeadBiasPad) {  // the last 16 threads finish db (columns >= O are zero padding)
        const int o = threadIdx.x - (kBlock - kHeadBiasPad);
        float total = 0.f;
        for (int s = 0; s < groups; ++s) total += red_bias[s * kHeadBiasPad + o];
        out[SL * K + o] = total;
    }
}

// partials [P, H] -> out [H] (H % 4 == 0), same scheme as the bias-gradient finalize of mlp_epilogue.hip:
// one block per 64 columns = 16 float4 lanes x 16 partial-row groups, groups combined through LDS in fixed order.
__global__ __launch_bounds__(kBlock) void narrow_linear_finalize_kernel(const float *__restrict__ partials, int64_t P,
                                                                        int H, float *__restrict__ out) {
    __shared__ float4 red[kBlock];
    const int c4 = threadIdx.x & 15, group = threadIdx.x >> 4;
    const int h = blockIdx.x * 64 + c4 * 4;
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h < H) {
        const float4 *base = reinterpret_cast<const float4 *>(partials + h);
        const int64_t stride = H / 4;
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
        *reinterpret_cast<float4 *>(out + h) = sum;
    }
}


// FORWARD of a one-output linear layer (the value head, a discriminator's logit): y[b] = x[b, :] . w + bias — a row dot
// product, memory-bound (4K bytes read per 4 written).  torch's addmm has no bias epilogue for a one-column output: it
// materialises the broadcast bias with a copy launch and then runs a skinny GEMM (4.8 + 4.5 us at [24576, 128]); here the
// kLanes = min(64, K / 4) lanes of a row group each take 16-byte chunks of the row (one chunk for K <= 256), the group is
// reduced by log2(kLanes) wave shuffles and its first lane adds the bias and stores.  Four row groups' loads in flight per lane.
template <int kLanes>
__global__ __launch_bounds__(kBlock) void narrow_dot_fwd_kernel(const float *__restrict__ input, const float *__restrict__ weight,
                                                                const float *__restrict__ bias, float *__restrict__ output,
                                                                int64_t rows, int K) {
    constexpr int kRowsPerWave = kWave / kLanes, kInFlight = 4;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane % kLanes, rloc = lane / kLanes;
    const int chunks = K / 4;  // a multiple of kLanes
    const int64_t wave = int64_t(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
    const int64_t row0 = wave * (kRowsPerWave * kInFlight);
    if (row0 >= rows) return;  // uniform per wave
    const float b = bias ? bias[0] : 0.f;
    float acc[kInFlight] = {0.f, 0.f, 0.f, 0.f};
    for (int c = sub; c < chunks; c += kLanes) {
        const float4 w = reinterpret_cast<const float4 *>(weight)[c];
        float4 x[kInFlight];
#pragma unroll
        for (int k = 0; k < kInFlight; ++k) {  // rows past the end: clamped (their sums are never stored)
            const int64_t row = min(row0 + int64_t(k) * kRowsPerWave + rloc, rows - 1);
            x[k] = reinterpret_cast<const float4 *>(input)[row * chunks + c];
        }
#pragma unroll
        for (int k = 0; k < kInFlight; ++k)
            acc[k] += (x[k].x * w.x + x[k].y * w.y) + (x[k].z * w.z + x[k].w * w.w);
    }
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) {
        float v = acc[k];
#pragma unroll
        for (int off = kLanes / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
        const int64_t row = row0 + int64_t(k) * kRowsPerWave + rloc;
        if (sub == 0 && row < rows) output[row] = v + b;
    }
}

static bool narrow_shape_ok(int64_t K, int64_t O) {
    if (O < 1 || O > 16 || K < 32 || K > 1024 || K % 4) return false;
    const int64_t lpr = K / 4;
    return (lpr & (lpr - 1)) == 0;  // power of two: divides the 256-lane block
}

template <int O, bool kReluInput>
static int launch_narrow_bwd_as(const float *grad_out, const float *input, const float *weight, float *grad_input,
                                float *partials, int64_t rows, int K, int64_t blocks, hipStream_t s) {
    const int lpr = K / 4, groups = kBlock / lpr;
    const size_t lds = size_t(groups) * (O + 1) * lpr * sizeof(float4) + size_t(groups) * kHeadBiasPad * sizeof(float);
    if (lds > 64 * 1024) {  // more than the default dynamic-LDS window: opt in (the CU has 160 KB)
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&narrow_linear_bwd_kernel<O, kReluInput>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (err != hipSuccess) return int(err);
    }
    hipLaunchKernelGGL((narrow_linear_bwd_kernel<O, kReluInput>), dim3(uint32_t(blocks)), dim3(kBlock), lds, s, grad_out,
                       input, weight, grad_input, partials, rows, K, head_rows_per_block());
    return launch_status();
}

template <int O>
static int launch_narrow_bwd(const float *grad_out, const float *input, const float *weight, float *grad_input,
                             float *partials, int64_t rows, int K, int64_t blocks, bool relu_input, hipStream_t s) {
    return relu_input ? launch_narrow_bwd_as<O, true>(grad_out, input, weight, grad_input, partials, rows, K, blocks, s)
                      : launch_narrow_bwd_as<O, false>(grad_out, input, weight, grad_input, partials, rows, K, blocks, s);
}

}  // namespace cusrl

extern "C" int cusrl_narrow_linear_supported(int64_t in_features, int64_t out_features) {
    return cusrl::narrow_shape_ok(in_features, out_features) ? 1 : 0;
}

extern "C" int64_t cusrl_narrow_linear_num_partials(int64_t rows) {
    return rows <= 0 ? 0 : cusrl::ceil_div(rows, int64_t(cusrl::head_rows_per_block()));
}

extern "C" int cusrl_narrow_linear_bwd(const float *grad_out, const float *input, const float *weight,
                                       float *grad_input, float *partials, float *packed, int64_t rows,
                                       int64_t in_features, int64_t out_features, int relu_input, void *stream) {
    using namespace cusrl;
    if (!grad_out || !input || !weight || !partials || rows <= 0) return CUSRL_E_INVALID;
    if (!narrow_shape_ok(in_features, out_features)) return CUSRL_E_UNSUPPORTED;
    if (out_features % 4 == 0 && !aligned(grad_out, 16)) return CUSRL_E_UNSUPPORTED;
    if (!aligned(input, 16) || !aligned(weight, 16) || !aligned(partials, 16) || (packed && !aligned(packed, 16)) ||
        (grad_input && !aligned(grad_input, 16)))
        return CUSRL_E_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    const int K = int(in_features);
    const int64_t blocks = ceil_div(rows, int64_t(head_rows_per_block()));
    const bool relu = relu_input != 0;
    int rc = 0;
    switch (out_features) {
#define CUSRL_NARROW_CASE(N) \
    case N: rc = launch_narrow_bwd<N>(grad_out, input, weight, grad_input, partials, rows, K, blocks, relu, s); break;
        CUSRL_NARROW_CASE(1) CUSRL_NARROW_CASE(2) CUSRL_NARROW_CASE(3) CUSRL_NARROW_CASE(4)
        CUSRL_NARROW_CASE(5) CUSRL_NARROW_CASE(6) CUSRL_NARROW_CASE(7) CUSRL_NARROW_CASE(8)
        CUSRL_NARROW_CASE(9) CUSRL_NARROW_CASE(10) CUSRL_NARROW_CASE(11) CUSRL_NARROW_CASE(12)
        CUSRL_NARROW_CASE(13) CUSRL_NARROW_CASE(14) CUSRL_NARROW_CASE(15) CUSRL_NARROW_CASE(16)
#undef CUSRL_NARROW_CASE
        default: return CUSRL_E_UNSUPPORTED;
    }
    if (rc) return rc;
    if (!packed) return 0;  // the caller reduces the partial rows itself (cusrl_assemble_gradients)
    const int H = (int(out_features) + 1) * K + kHeadBiasPad;
    hipLaunchKernelGGL(narrow_linear_finalize_kernel, dim3(uint32_t(ceil_div(H, 64))), dim3(kBlock), 0, s, partials,
                       blocks, H, packed);
    return launch_status();
}

extern "C" int cusrl_narrow_linear_fwd(const float *input, const float *weight, const float *bias, float *output,
                                       int64_t rows, int64_t in_features, int64_t out_features, void *stream) {
    using namespace cusrl;
    if (!input || !weight || !output || rows <= 0) return CUSRL_E_INVALID;
    if (out_features != 1 || !narrow_shape_ok(in_features, 1)) return CUSRL_E_UNSUPPORTED;  // wider heads: the library GEMM
    if (!aligned(input, 16) || !aligned(weight, 16)) return CUSRL_E_UNSUPPORTED;
    const int K = int(in_features), lanes = K / 4 >= kWave ? kWave : K / 4;
    const int64_t waves = ceil_div(rows, int64_t(kWave / lanes) * 4), blocks = ceil_div(waves, int64_t(kWavesPerBlock));
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    switch (lanes) {
#define CUSRL_DOT_CASE(N)                                                                                              \
    case N:                                                                                                            \
        hipLaunchKernelGGL(narrow_dot_fwd_kernel<N>, dim3(uint32_t(blocks)), dim3(kBlock), 0, s, input, weight, bias,   \
                           output, rows, K);                                                                           \
        break;
        CUSRL_DOT_CASE(8) CUSRL_DOT_CASE(16) CUSRL_DOT_CASE(32) CUSRL_DOT_CASE(64)
#undef CUSRL_DOT_CASE
        default: return CUSRL_E_UNSUPPORTED;
    }
    return launch_status();
}
