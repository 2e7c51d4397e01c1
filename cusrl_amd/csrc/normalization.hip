// Running observation statistics for gfx950: masked column mean / variance of one env step,
// Chan merge into the running statistics, and the normalise-and-clamp pass — three launches, no host round trip
// (the reference: ~15 torch launches and a boolean-mask select that synchronises, twice per env step).
#include "common.hpp"

namespace cusrl {

constexpr int kRmsMaxBlocks = 256;

// partials[block][c][{sum, sumsq}] for c < C, partials[block][C][0] = number of selected rows in this block's rows.
__global__ __launch_bounds__(kBlock) void masked_col_stats_kernel(const float *__restrict__ x,
                                                                  const uint8_t *__restrict__ mask, int64_t rows, int C,
                                                                  double *__restrict__ partials) {
    // a lane keeps one channel: its stride over the flat [rows * C] array is a multiple of C
    const int64_t threads = int64_t(gridDim.x) * kBlock;
    const int64_t stride = threads / C * C;
    const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t E = rows * C;
    double sum = 0.0, sumsq = 0.0, count = 0.0;
    for (int64_t i = tid < stride ? tid : E; i < E; i += stride) {
        const int64_t row = i / C;
        if (!mask || mask[row]) {
            const double v = double(x[i]);
            sum += v;
            sumsq += v * v;
            count += 1.0;
        }
    }
    __shared__ double red[kBlock][3];
    red[threadIdx.x][0] = sum;
    red[threadIdx.x][1] = sumsq;
    red[threadIdx.x][2] = count;
    __syncthreads();
    // Fold the kBlock / C lanes of every channel in two levels (with 2 .. 12 channels a single loop per channel is a
    // chain of 20 .. 128 dependent LDS reads: 9 us for the toy config's 2-channel observation): `groups` lanes per
    // column take every groups-th lane of that channel, then one lane per column folds the groups.
    const int64_t base = int64_t(blockIdx.x) * kBlock;
    const int columns = C + 1;
    const int groups = kBlock / columns < 16 ? kBlock / columns : 16;  // >= 1 because C < kBlock
    const int column = threadIdx.x % columns, group = threadIdx.x / columns;
    // column c < C folds the lanes of channel c; column C folds the row counts of channel 0's lanes
    const int channel = column < C ? column : 0;
    const int first = int((int64_t(channel) - base % C + C) % C);
    double s = 0.0, q = 0.0, n = 0.0;
    if (group < groups) {
        for (int k = first + group * C; k < kBlock; k += groups * C) {
            s += red[k][0];
            q += red[k][1];
            n += red[k][2];
        }
    }
    __syncthreads();
    if (group < groups) {
        red[threadIdx.x][0] = s;
        red[threadIdx.x][1] = q;
        red[threadIdx.x][2] = n;
    }
    __syncthreads();
    if (threadIdx.x <= C) {
        s = q = n = 0.0;
        for (int g = 0; g < groups; ++g) {
            s += red[g * columns + threadIdx.x][0];
            q += red[g * columns + threadIdx.x][1];
            n += red[g * columns + threadIdx.x][2];
        }
        double *out = partials + (int64_t(blockIdx.x) * (C + 1) + threadIdx.x) * 2;
        if (threadIdx.x < C) {
            out[0] = s;
            out[1] = q;
        } else {
            out[0] = n;
            out[1] = 0.0;
        }
    }
}

// One block folds the P block partials: the C + 1 columns (C channels + the row count) are spread over the whole block,
// kBlock / (C + 1) lanes per column each taking every G-th partial, so the serial chain is P / G fp64 loads long instead
// of P (P = 24 .. 256; as a single loop per channel this launch was 15 us inside every captured env step).
__global__ __launch_bounds__(kBlock) void masked_stats_finalize_kernel(const double *__restrict__ partials, int P, int C,
                                                                       float *__restrict__ mean, float *__restrict__ var,
                                                                       double *__restrict__ count) {
    __shared__ double folded[kBlock][2];
    const int columns = C + 1;            // C < kBlock (checked by the entry point)
    const int groups = kBlock / columns;  // >= 1
    const int column = threadIdx.x % columns, group = threadIdx.x / columns;
    double s = 0.0, q = 0.0;
    if (group < groups) {
        for (int p = group; p < P; p += groups) {
            const double2 v = reinterpret_cast<const double2 *>(partials)[int64_t(p) * columns + column];
            s += v.x;
            q += v.y;
        }
    }
    folded[threadIdx.x][0] = s;
    folded[threadIdx.x][1] = q;
    __syncthreads();
    if (threadIdx.x < C) {
        double n = 0.0;
        s = q = 0.0;
        for (int g = 0; g < groups; ++g) {
            s += folded[g * columns + threadIdx.x][0];
            q += folded[g * columns + threadIdx.x][1];
            n += folded[g * columns + C][0];
        }
        if (n > 0.0) {
            const double m = s / n;
            const double v = q / n - m * m;  // population variance (correction = 0)
            mean[threadIdx.x] = float(m);
            var[threadIdx.x] = float(v < 0.0 ? 0.0 : v);
        } else {
            mean[threadIdx.x] = 0.0f;
            var[threadIdx.x] = 1.0f;
        }
        if (threadIdx.x == 0) *count = n;
    }
}

__global__ __launch_bounds__(kBlock) void rms_merge_kernel(float *__restrict__ mean, float *__restrict__ var,
                                                           float *__restrict__ std, double *__restrict__ count,
                                                           const float *__restrict__ batch_mean,
                                                           const float *__restrict__ batch_var,
                                                           const double *__restrict__ batch_count, float eps,
                                                           double max_count, int C) {
    const double w_old_count = *count, w_new_count = *batch_count;
    __syncthreads();  // every lane has read the counts before lane 0 updates them
    if (w_new_count <= 0.0) return;
    const double w_sum = w_old_count + w_new_count;
    const float w_new = float(w_new_count / w_sum);
    const float w_cross = float((w_old_count / w_sum) * (w_new_count / w_sum));
    for (int c = threadIdx.x; c < C; c += kBlock) {
        const float delta = batch_mean[c] - mean[c];
        const float m = mean[c] + delta * w_new;
        const float v = var[c] + ((batch_var[c] - var[c]) * w_new + (delta * delta) * w_cross);
        mean[c] = m;
        var[c] = v;
        std[c] = sqrtf(v + eps);
    }
    if (threadIdx.x == 0) {
        double total = w_sum;
        if (max_count > 0.0 && total > max_count) total = max_count;
        *count = total;
    }
}

__global__ __launch_bounds__(kBlock) void rms_normalize_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                               const float *__restrict__ std, float clamp,
                                                               float *__restrict__ out, int64_t E, int C, int vec4) {
    const int64_t tid = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    if (vec4) {  // C % 4 == 0: the 4 elements of a chunk are 4 adjacent channels of one row
        const int64_t n4 = E / 4;
        for (int64_t i = tid; i < n4; i += stride) {
            const int c = int((i * 4) % C);
            const float4 v = reinterpret_cast<const float4 *>(x)[i];
            const float4 m = *reinterpret_cast<const float4 *>(mean + c);
            const float4 s = *reinterpret_cast<const float4 *>(std + c);
            float r[4] = {(v.x - m.x) / s.x, (v.y - m.y) / s.y, (v.z - m.z) / s.z, (v.w - m.w) / s.w};
            if (clamp > 0.0f) {
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = fminf(fmaxf(r[j], -clamp), clamp);
            }
            reinterpret_cast<float4 *>(out)[i] = make_float4(r[0], r[1], r[2], r[3]);
        }
    } else {
        for (int64_t i = tid; i < E; i += stride) {
            const int c = int(i % C);
            float r = (x[i] - mean[c]) / std[c];
            if (clamp > 0.0f) r = fminf(fmaxf(r, -clamp), clamp);
            out[i] = r;
        }
    }
}

}  // namespace cusrl

using namespace cusrl;

extern "C" int64_t cusrl_masked_stats_num_partials(int64_t rows, int64_t C) {
    if (rows <= 0 || C <= 0) return 0;
    const int64_t want = ceil_div(rows * C, int64_t(kBlock) * 8);
    return want < 1 ? 1 : (want > kRmsMaxBlocks ? kRmsMaxBlocks : want);
}

extern "C" int cusrl_masked_col_stats(const float *x, const uint8_t *mask, int64_t rows, int64_t C, double *partials,
                                      float *batch_mean, float *batch_var, double *batch_count, void *stream) {
    if (rows < 0 || C <= 0) return CUSRL_E_INVALID;
    if (!partials || !batch_mean || !batch_var || !batch_count || (rows > 0 && !x)) return CUSRL_E_INVALID;
    if (C >= kBlock) return CUSRL_E_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    const int64_t P = rows == 0 ? 0 : cusrl_masked_stats_num_partials(rows, C);
    if (P > 0) {
        hipLaunchKernelGGL(masked_col_stats_kernel, dim3(uint32_t(P)), dim3(kBlock), 0, s, x, mask, rows, int(C),
                           partials);
        if (int rc = launch_status()) return rc;
    }
    hipLaunchKernelGGL(masked_stats_finalize_kernel, dim3(1), dim3(kBlock), 0, s, partials, int(P), int(C), batch_mean,
                       batch_var, batch_count);
    return launch_status();
}

extern "C" int cusrl_rms_merge(float *mean, float *var, float *std, double *count, const float *batch_mean,
                               const float *batch_var, const double *batch_count, float eps, double max_count,
                               int64_t C, void *stream) {
    if (C <= 0 || !mean || !var || !std || !count || !batch_mean || !batch_var || !batch_count) return CUSRL_E_INVALID;
    if (C > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(rms_merge_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), mean, var, std, count, batch_mean,
                       batch_var, batch_count, eps, max_count, int(C));
    return launch_status();
}

extern "C" int cusrl_rms_normalize(const float *x, const float *mean, const float *std, float clamp, float *out,
                                   int64_t rows, int64_t C, void *stream) {
    if (rows < 0 || C <= 0) return CUSRL_E_INVALID;
    if (rows == 0) return 0;
    if (!x || !mean || !std || !out) return CUSRL_E_INVALID;
    if (C > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const int64_t E = rows * C;
    const int vec4 = C % 4 == 0 && aligned(x, 16) && aligned(out, 16) && aligned(mean, 16) && aligned(std, 16);
    int64_t blocks = ceil_div(vec4 ? E / 4 : E, kBlock);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rms_normalize_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), x, mean, std,
                       clamp, out, E, int(C), vec4);
    return launch_status();
}
