// Stand-alone sweep of GAE kernel shapes at roofline scale (no torch): which launch shape / load schedule gets the
// 21 B/slot scan closest to what the memory system gives a plain float4 copy on this box.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/gae_sweep.hip -o scripts/_bin/gae_sweep
//   scripts/_bin/gae_sweep [envs=1048576] [T=24]
//
// Every variant computes the same bit pattern (checked against variant 0 on the device's own output).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int kWave = 64;

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

typedef float native_f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ float4 ld4(const float *p) {
    if constexpr (NT) {
        const native_f4 v = __builtin_nontemporal_load(reinterpret_cast<const native_f4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else
        return *reinterpret_cast<const float4 *>(p);
}
template <bool NT>
__device__ __forceinline__ uint32_t ld1(const uint8_t *p) {
    if constexpr (NT)
        return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
    else
        return *reinterpret_cast<const uint32_t *>(p);
}
template <bool NT>
__device__ __forceinline__ void st4(float *p, float4 v) {
    if constexpr (NT) {
        native_f4 n = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(n, reinterpret_cast<native_f4 *>(p));
    } else
        *reinterpret_cast<float4 *>(p) = v;
}

// step of the recurrence for 4 columns
__device__ __forceinline__ void gae_step(const float4 &r, const float4 &v, const float4 &nv, uint32_t dn, bool last,
                                         float gamma, float c_adv, float (&carry)[4], float4 &adv, float4 &ret,
                                         double &sum, double &sumsq) {
    const float rr[4] = {r.x, r.y, r.z, r.w}, vv[4] = {v.x, v.y, v.z, v.w}, nn[4] = {nv.x, nv.y, nv.z, nv.w};
    float aa[4], rt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float delta = __fsub_rn(__fadd_rn(rr[j], __fmul_rn(nn[j], gamma)), vv[j]);
        float a = delta;
        if (!last) {
            const float coef = ((dn >> (8 * j)) & 0xffu) ? 0.0f : c_adv;
            a = __fadd_rn(delta, __fmul_rn(coef, carry[j]));
        }
        carry[j] = a;
        aa[j] = a;
        rt[j] = __fadd_rn(vv[j], a);
        sum += double(a);
        sumsq += double(a) * double(a);
    }
    adv = make_float4(aa[0], aa[1], aa[2], aa[3]);
    ret = make_float4(rt[0], rt[1], rt[2], rt[3]);
}

template <int BLK>
__device__ __forceinline__ void write_partials(double sum, double sumsq, double *partials) {
    __shared__ double scratch[BLK / kWave][2];
    const double s = wave_sum(sum), q = wave_sum(sumsq);
    if ((threadIdx.x & (kWave - 1)) == 0) scratch[threadIdx.x / kWave][0] = s, scratch[threadIdx.x / kWave][1] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int w = 0; w < BLK / kWave; ++w) ts += scratch[w][0], tq += scratch[w][1];
        partials[int64_t(blockIdx.x) * 2 + 0] = ts;
        partials[int64_t(blockIdx.x) * 2 + 1] = tq;
    }
}

// ---- variant A: the shipped schedule (load a chunk of TC steps, wait, scan, store), parameterised
template <int TC, int BLK, bool NTL, bool NTS>
__global__ __launch_bounds__(BLK) void gae_plain(const float *__restrict__ reward, const float *__restrict__ value,
                                                 const float *__restrict__ next_value, const uint8_t *__restrict__ done,
                                                 float *__restrict__ advantage, float *__restrict__ ret,
                                                 double *__restrict__ partials, int T, int64_t C, float gamma,
                                                 float c_adv) {
    const int64_t col = (int64_t(blockIdx.x) * BLK + threadIdx.x) * 4;
    float carry[4] = {0, 0, 0, 0};
    double sum = 0.0, sumsq = 0.0;
    if (col < C) {
        for (int t_hi = T; t_hi > 0; t_hi -= TC) {
            float4 r[TC], v[TC], nv[TC];
            uint32_t dn[TC];
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                const int t = t_hi - 1 - k;
                if (t >= 0) {
                    const int64_t off = int64_t(t) * C + col;
                    r[k] = ld4<NTL>(reward + off);
                    v[k] = ld4<NTL>(value + off);
                    nv[k] = ld4<NTL>(next_value + off);
                    dn[k] = ld1<NTL>(done + off);
                }
            }
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                const int t = t_hi - 1 - k;
                if (t >= 0) {
                    float4 a, rt;
                    gae_step(r[k], v[k], nv[k], dn[k], t == T - 1, gamma, c_adv, carry, a, rt, sum, sumsq);
                    const int64_t off = int64_t(t) * C + col;
                    st4<NTS>(advantage + off, a);
                    st4<NTS>(ret + off, rt);
                }
            }
        }
    }
    write_partials<BLK>(sum, sumsq, partials);
}

// ---- variant B: software pipelined — the next chunk's loads are issued before the current chunk is scanned and stored,
// so a wave always has reads in flight (T must be a multiple of 2*TC for the ping-pong below; checked by the host)
template <int TC, int BLK, bool NTL, bool NTS>
__global__ __launch_bounds__(BLK) void gae_pipelined(const float *__restrict__ reward, const float *__restrict__ value,
                                                     const float *__restrict__ next_value,
                                                     const uint8_t *__restrict__ done, float *__restrict__ advantage,
                                                     float *__restrict__ ret, double *__restrict__ partials, int T,
                                                     int64_t C, float gamma, float c_adv) {
    const int64_t col = (int64_t(blockIdx.x) * BLK + threadIdx.x) * 4;
    float carry[4] = {0, 0, 0, 0};
    double sum = 0.0, sumsq = 0.0;
    if (col < C) {
        float4 r0[TC], v0[TC], n0[TC], r1[TC], v1[TC], n1[TC];
        uint32_t d0[TC], d1[TC];
        auto load = [&](float4(&r)[TC], float4(&v)[TC], float4(&nv)[TC], uint32_t(&dn)[TC], int t_hi) {
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                const int t = t_hi - 1 - k;
                if (t >= 0) {
                    const int64_t off = int64_t(t) * C + col;
                    r[k] = ld4<NTL>(reward + off);
                    v[k] = ld4<NTL>(value + off);
                    nv[k] = ld4<NTL>(next_value + off);
                    dn[k] = ld1<NTL>(done + off);
                }
            }
        };
        auto scan = [&](float4(&r)[TC], float4(&v)[TC], float4(&nv)[TC], uint32_t(&dn)[TC], int t_hi) {
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                const int t = t_hi - 1 - k;
                if (t >= 0) {
                    float4 a, rt;
                    gae_step(r[k], v[k], nv[k], dn[k], t == T - 1, gamma, c_adv, carry, a, rt, sum, sumsq);
                    const int64_t off = int64_t(t) * C + col;
                    st4<NTS>(advantage + off, a);
                    st4<NTS>(ret + off, rt);
                }
            }
        };
        load(r0, v0, n0, d0, T);
        for (int t_hi = T; t_hi > 0; t_hi -= 2 * TC) {
            load(r1, v1, n1, d1, t_hi - TC);
            scan(r0, v0, n0, d0, t_hi);
            load(r0, v0, n0, d0, t_hi - 2 * TC);
            scan(r1, v1, n1, d1, t_hi - TC);
        }
    }
    write_partials<BLK>(sum, sumsq, partials);
}

// ---- reference points
__global__ __launch_bounds__(256) void fill4(float4 *__restrict__ dst, int64_t n, float v) {
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) dst[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void copy4(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n) {
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
// same six streams with the same bytes, but every block walks ONE contiguous span of the flattened arrays (what the
// access pattern would cost without the 4 MB row stride)
__global__ __launch_bounds__(256) void six_streams_linear(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                          const float4 *__restrict__ c, const uint32_t *__restrict__ d,
                                                          float4 *__restrict__ o1, float4 *__restrict__ o2, int64_t n4,
                                                          int per_block) {
    const int64_t base = int64_t(blockIdx.x) * per_block * 256;
    for (int k = 0; k < per_block; k += 6) {
        float4 x[6], y[6], z[6];
        uint32_t f[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int64_t i = base + int64_t(k + j) * 256 + threadIdx.x;
            if (i < n4) x[j] = a[i], y[j] = b[i], z[j] = c[i], f[j] = d[i];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int64_t i = base + int64_t(k + j) * 256 + threadIdx.x;
            if (i < n4) {
                float s = f[j] ? 1.0f : 0.5f;
                o1[i] = make_float4(x[j].x + y[j].x * s, x[j].y + y[j].y * s, x[j].z + y[j].z * s, x[j].w + y[j].w * s);
                o2[i] = make_float4(z[j].x - y[j].x, z[j].y - y[j].y, z[j].z - y[j].z, z[j].w - y[j].w);
            }
        }
    }
}

struct Variant {
    const char *name;
    void (*launch)(const float *, const float *, const float *, const uint8_t *, float *, float *, double *, int, int64_t,
                   hipStream_t);
};

template <int TC, int BLK, bool NTL, bool NTS>
void launch_plain(const float *r, const float *v, const float *nv, const uint8_t *d, float *a, float *rt, double *p, int T,
                  int64_t C, hipStream_t s) {
    const int64_t blocks = (C / 4 + BLK - 1) / BLK;
    hipLaunchKernelGGL((gae_plain<TC, BLK, NTL, NTS>), dim3(uint32_t(blocks)), dim3(BLK), 0, s, r, v, nv, d, a, rt, p, T, C,
                       0.99f, float(0.99 * 0.95));
}
template <int TC, int BLK, bool NTL, bool NTS>
void launch_pipe(const float *r, const float *v, const float *nv, const uint8_t *d, float *a, float *rt, double *p, int T,
                 int64_t C, hipStream_t s) {
    if (T % (2 * TC)) return;
    const int64_t blocks = (C / 4 + BLK - 1) / BLK;
    hipLaunchKernelGGL((gae_pipelined<TC, BLK, NTL, NTS>), dim3(uint32_t(blocks)), dim3(BLK), 0, s, r, v, nv, d, a, rt, p, T,
                       C, 0.99f, float(0.99 * 0.95));
}

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 1048576;
    const int T = argc > 2 ? atoi(argv[2]) : 24;
    const int64_t S = N * T;
    const int iters = 20;
    const bool cold = argc > 3 && atoi(argv[3]) != 0;  // evict the caches (1 GB of writes) before every timed launch
    float4 *scratch = nullptr;
    const int64_t scratch4 = int64_t(1) << 26;  // 1 GiB
    if (cold) CHECK(hipMalloc(&scratch, scratch4 * 16));
    float *reward, *value, *nv, *adv, *ret, *adv0, *ret0;
    uint8_t *done;
    double *partials;
    CHECK(hipMalloc(&reward, S * 4));
    CHECK(hipMalloc(&value, S * 4));
    CHECK(hipMalloc(&nv, S * 4));
    CHECK(hipMalloc(&adv, S * 4));
    CHECK(hipMalloc(&ret, S * 4));
    CHECK(hipMalloc(&adv0, S * 4));
    CHECK(hipMalloc(&ret0, S * 4));
    CHECK(hipMalloc(&done, S));
    CHECK(hipMalloc(&partials, (N / 64 + 1) * 16));
    {
        std::vector<float> h(S);
        std::vector<uint8_t> hd(S);
        uint32_t x = 12345;
        auto rnd = [&]() {
            x = x * 1664525u + 1013904223u;
            return float(x >> 8) / float(1 << 24) * 2.0f - 1.0f;
        };
        for (auto &e : h) e = rnd();
        CHECK(hipMemcpy(reward, h.data(), S * 4, hipMemcpyHostToDevice));
        for (auto &e : h) e = rnd();
        CHECK(hipMemcpy(value, h.data(), S * 4, hipMemcpyHostToDevice));
        for (auto &e : h) e = rnd();
        CHECK(hipMemcpy(nv, h.data(), S * 4, hipMemcpyHostToDevice));
        for (auto &e : hd) e = rnd() > 0.97f;
        CHECK(hipMemcpy(done, hd.data(), S, hipMemcpyHostToDevice));
    }
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double bytes = double(S) * 21.0;

    std::vector<Variant> variants = {
        {"plain  TC6  B256 (shipped)", launch_plain<6, 256, false, false>},
        {"plain  TC6  B256 nt-load", launch_plain<6, 256, true, false>},
        {"plain  TC6  B256 nt-both", launch_plain<6, 256, true, true>},
        {"plain  TC6  B128", launch_plain<6, 128, false, false>},
        {"plain  TC6  B128 nt-load", launch_plain<6, 128, true, false>},
        {"plain  TC6  B128 nt-both", launch_plain<6, 128, true, true>},
        {"plain  TC4  B128 nt-load", launch_plain<4, 128, true, false>},
        {"plain  TC8  B128 nt-load", launch_plain<8, 128, true, false>},
        {"plain  TC12 B128 nt-load", launch_plain<12, 128, true, false>},
        {"plain  TC6  B64  nt-load", launch_plain<6, 64, true, false>},
        {"plain  TC4  B256 nt-load", launch_plain<4, 256, true, false>},
        {"plain  TC8  B256 nt-load", launch_plain<8, 256, true, false>},
        {"plain  TC12 B256 nt-load", launch_plain<12, 256, true, false>},
        {"pipe   TC3  B256 nt-load", launch_pipe<3, 256, true, false>},
        {"pipe   TC4  B128 nt-load", launch_pipe<4, 128, true, false>},
        {"pipe   TC6  B128 nt-load", launch_pipe<6, 128, true, false>},
    };
    printf("GAE sweep: N = %lld envs, T = %d, %.1f MB algorithmic per launch; peak 8000 GB/s; %s\n", (long long)N, T, bytes / 1e6, cold ? "COLD: 1 GiB written before every timed launch, one launch per event pair" : "hot loop: 20 launches back to back");
    // reference output
    variants[0].launch(reward, value, nv, done, adv0, ret0, partials, T, N, stream);
    CHECK(hipStreamSynchronize(stream));
    std::vector<float> h0(S), h1(S);
    CHECK(hipMemcpy(h0.data(), adv0, S * 4, hipMemcpyDeviceToHost));
    for (int round = 0; round < 2; ++round) {
        printf("-- round %d\n", round);
        for (auto &var : variants) {
            CHECK(hipMemsetAsync(adv, 0, S * 4, stream));
            for (int i = 0; i < 2; ++i) var.launch(reward, value, nv, done, adv, ret, partials, T, N, stream);
            double us;
            if (!cold) {
                CHECK(hipEventRecord(e0, stream));
                for (int i = 0; i < iters; ++i) var.launch(reward, value, nv, done, adv, ret, partials, T, N, stream);
                CHECK(hipEventRecord(e1, stream));
                CHECK(hipStreamSynchronize(stream));
                CHECK(hipGetLastError());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                us = ms * 1e3 / iters;
            } else {
                double total = 0;
                for (int i = 0; i < 8; ++i) {
                    hipLaunchKernelGGL(fill4, dim3(4096), dim3(256), 0, stream, scratch, scratch4, float(i));
                    CHECK(hipEventRecord(e0, stream));
                    var.launch(reward, value, nv, done, adv, ret, partials, T, N, stream);
                    CHECK(hipEventRecord(e1, stream));
                    CHECK(hipStreamSynchronize(stream));
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    total += ms * 1e3;
                }
                us = total / 8;
            }
            const char *ok = "";
            if (round == 0) {
                CHECK(hipMemcpy(h1.data(), adv, S * 4, hipMemcpyDeviceToHost));
                ok = memcmp(h0.data(), h1.data(), S * 4) == 0 ? " bit-exact" : " MISMATCH";
            }
            printf("%-34s %8.1f us  %7.1f GB/s  %.3f%s\n", var.name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0, ok);
        }
        // reference points
        {
            const int64_t n4 = int64_t(bytes / 2 / 16);
            for (int grid : {2048, 8192}) {
                CHECK(hipEventRecord(e0, stream));
                for (int i = 0; i < iters; ++i)
                    hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, stream, (const float4 *)reward, (float4 *)adv, n4 > S / 4 ? S / 4 : n4);
                CHECK(hipEventRecord(e1, stream));
                CHECK(hipStreamSynchronize(stream));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double moved = double(n4 > S / 4 ? S / 4 : n4) * 32.0;
                const double us = ms * 1e3 / iters;
                printf("%-34s %8.1f us  %7.1f GB/s  %.3f\n", grid == 2048 ? "float4 copy grid 2048" : "float4 copy grid 8192", us,
                       moved / us / 1e3, moved / us / 1e3 / 8000.0);
            }
            const int64_t s4 = S / 4;
            for (int per_block : {6, 24}) {
                const int64_t blocks = (s4 + int64_t(per_block) * 256 - 1) / (int64_t(per_block) * 256);
                CHECK(hipEventRecord(e0, stream));
                for (int i = 0; i < iters; ++i)
                    hipLaunchKernelGGL(six_streams_linear, dim3(uint32_t(blocks)), dim3(256), 0, stream, (const float4 *)reward,
                                       (const float4 *)value, (const float4 *)nv, (const uint32_t *)done, (float4 *)adv,
                                       (float4 *)ret, s4, per_block);
                CHECK(hipEventRecord(e1, stream));
                CHECK(hipStreamSynchronize(stream));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / iters;
                printf("six linear streams, %2d rounds/blk   %8.1f us  %7.1f GB/s  %.3f\n", per_block, us, bytes / us / 1e3,
                       bytes / us / 1e3 / 8000.0);
            }
        }
    }
    return 0;
}
