// GRU / LSTM time-step epilogues for gfx950 (recurrent BPTT, config 4).
//
// torch.nn.GRU on ROCm is MIOpen's RNN: per time step it issues generic tensor kernels (Op2dTensorLite,
// SubTensorOpWithSubTensor2d: ~24 000 launches per iteration of config 4) and reduces the bias gradients with
// Op2dTensorSquash at 7.6 ms a call — 45 % of the device time of that config (profiles/r02/config4_miopen_kernel_stats.csv).
// Here the layer is what it is algebraically: ONE input-projection GEMM for all time steps, per step one recurrent GEMM
// (rocBLAS, [B, H] x [H, 3H]) and ONE pass over the gates; backward mirrors it, and the weight / bias gradients are
// batched GEMMs / column sums over all steps at once.  These kernels are that one pass:
//
//   r = sigmoid(gi_r + gh_r + b_hr)      z = sigmoid(gi_z + gh_z + b_hz)
//   q = gh_n + b_hn                      n = tanh(gi_n + r * q)
//   h' = (1 - z) * n + z * h             (torch.nn.GRU; gi = W_ih x + b_ih, gh = W_hh h)
//
// `lengths` (optional) gives packed-sequence semantics without packing: a sequence that has ended (t >= lengths[b]) keeps
// its state, emits zeros, and lets the state gradient pass through untouched — so the final state is the state at each
// sequence's own last step (cusrl/nn/module/rnn.py:273-291 obtains that through a PackedSequence and a host read of the
// lengths).  Streaming, HBM-bound: 36 B per state element forward, 64 B backward; 16-byte lanes when H % 4 == 0.
#include <stdlib.h>

#include "common.hpp"

namespace cusrl {

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct GruGates {
    float r, z, q, n;
};

__device__ __forceinline__ GruGates gru_gates(float gi_r, float gi_z, float gi_n, float gh_r, float gh_z, float gh_n,
                                              float b_r, float b_z, float b_n) {
    GruGates g;
    g.r = gru_sigmoid(gi_r + gh_r + b_r);
    g.z = gru_sigmoid(gi_z + gh_z + b_z);
    g.q = gh_n + b_n;
    g.n = tanhf(gi_n + g.r * g.q);
    return g;
}

// V = float4 (H % 4 == 0, aligned rows) or float.
template <typename V>
__global__ __launch_bounds__(kBlock) void gru_gates_fwd_kernel(const float *__restrict__ gi, const float *__restrict__ gh,
                                                               const float *__restrict__ b_hh, float *__restrict__ h,
                                                               float *__restrict__ out,
                                                               const int64_t *__restrict__ lengths, int64_t t,
                                                               int64_t B, int H) {
    constexpr int kW = sizeof(V) / sizeof(float);
    const int cols = H / kW;
    const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= B * cols) return;
    const int64_t b = e / cols;
    const int j = int(e - b * cols) * kW;
    const float *gi_row = gi + b * 3 * H, *gh_row = gh + b * 3 * H;
    V gir = *reinterpret_cast<const V *>(gi_row + j), giz = *reinterpret_cast<const V *>(gi_row + H + j),
      gin = *reinterpret_cast<const V *>(gi_row + 2 * H + j);
    V ghr = *reinterpret_cast<const V *>(gh_row + j), ghz = *reinterpret_cast<const V *>(gh_row + H + j),
      ghn = *reinterpret_cast<const V *>(gh_row + 2 * H + j);
    V hp = *reinterpret_cast<const V *>(h + b * H + j);
    V br, bz, bn;
    if (b_hh) {
        br = *reinterpret_cast<const V *>(b_hh + j), bz = *reinterpret_cast<const V *>(b_hh + H + j),
        bn = *reinterpret_cast<const V *>(b_hh + 2 * H + j);
    }
    const bool live = !lengths || t < lengths[b];
    V hn, o;
    const float *pgir = reinterpret_cast<const float *>(&gir), *pgiz = reinterpret_cast<const float *>(&giz),
                *pgin = reinterpret_cast<const float *>(&gin), *pghr = reinterpret_cast<const float *>(&ghr),
                *pghz = reinterpret_cast<const float *>(&ghz), *pghn = reinterpret_cast<const float *>(&ghn),
                *php = reinterpret_cast<const float *>(&hp), *pbr = reinterpret_cast<const float *>(&br),
                *pbz = reinterpret_cast<const float *>(&bz), *pbn = reinterpret_cast<const float *>(&bn);
    float *phn = reinterpret_cast<float *>(&hn), *po = reinterpret_cast<float *>(&o);
#pragma unroll
    for (int k = 0; k < kW; ++k) {
        const GruGates g = gru_gates(pgir[k], pgiz[k], pgin[k], pghr[k], pghz[k], pghn[k], b_hh ? pbr[k] : 0.0f,
                                     b_hh ? pbz[k] : 0.0f, b_hh ? pbn[k] : 0.0f);
        const float next = (1.0f - g.z) * g.n + g.z * php[k];
        phn[k] = live ? next : php[k];
        po[k] = live ? next : 0.0f;
    }
    *reinterpret_cast<V *>(h + b * H + j) = hn;
    *reinterpret_cast<V *>(out + b * H + j) = o;
}

// In place: gi <- d(loss)/d(gi), gh <- d(loss)/d(gh) (the saved pre-activations are consumed), dh <- dh_t * z (the
// direct path to h_{t-1}; the caller adds d_gh @ W_hh).  dh on entry = gradient arriving from step t + 1.
template <typename V>
__global__ __launch_bounds__(kBlock) void gru_gates_bwd_kernel(float *__restrict__ gi, float *__restrict__ gh,
                                                               const float *__restrict__ b_hh,
                                                               const float *__restrict__ h_prev,
                                                               const float *__restrict__ d_out, float *__restrict__ dh,
                                                               const int64_t *__restrict__ lengths, int64_t t,
                                                               int64_t B, int H) {
    constexpr int kW = sizeof(V) / sizeof(float);
    const int cols = H / kW;
    const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= B * cols) return;
    const int64_t b = e / cols;
    const int j = int(e - b * cols) * kW;
    float *gi_row = gi + b * 3 * H, *gh_row = gh + b * 3 * H;
    const bool live = !lengths || t < lengths[b];
    if (!live) {  // ended sequence: no gate gradients, the state gradient passes through (dh stays as it is)
        V zero;
        float *pz = reinterpret_cast<float *>(&zero);
#pragma unroll
        for (int k = 0; k < kW; ++k) pz[k] = 0.0f;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            *reinterpret_cast<V *>(gi_row + g * H + j) = zero;
            *reinterpret_cast<V *>(gh_row + g * H + j) = zero;
        }
        return;
    }
    V gir = *reinterpret_cast<const V *>(gi_row + j), giz = *reinterpret_cast<const V *>(gi_row + H + j),
      gin = *reinterpret_cast<const V *>(gi_row + 2 * H + j);
    V ghr = *reinterpret_cast<const V *>(gh_row + j), ghz = *reinterpret_cast<const V *>(gh_row + H + j),
      ghn = *reinterpret_cast<const V *>(gh_row + 2 * H + j);
    V hp = *reinterpret_cast<const V *>(h_prev + b * H + j);
    V dhn = *reinterpret_cast<const V *>(dh + b * H + j);
    V dout;
    if (d_out) dout = *reinterpret_cast<const V *>(d_out + b * H + j);
    V br, bz, bn;
    if (b_hh) {
        br = *reinterpret_cast<const V *>(b_hh + j), bz = *reinterpret_cast<const V *>(b_hh + H + j),
        bn = *reinterpret_cast<const V *>(b_hh + 2 * H + j);
    }
    const float *pgir = reinterpret_cast<const float *>(&gir), *pgiz = reinterpret_cast<const float *>(&giz),
                *pgin = reinterpret_cast<const float *>(&gin), *pghr = reinterpret_cast<const float *>(&ghr),
                *pghz = reinterpret_cast<const float *>(&ghz), *pghn = reinterpret_cast<const float *>(&ghn),
                *php = reinterpret_cast<const float *>(&hp), *pdhn = reinterpret_cast<const float *>(&dhn),
                *pdout = reinterpret_cast<const float *>(&dout), *pbr = reinterpret_cast<const float *>(&br),
                *pbz = reinterpret_cast<const float *>(&bz), *pbn = reinterpret_cast<const float *>(&bn);
    V d_r, d_z, d_n, d_q, d_h;
    float *pdr = reinterpret_cast<float *>(&d_r), *pdz = reinterpret_cast<float *>(&d_z),
          *pdn = reinterpret_cast<float *>(&d_n), *pdq = reinterpret_cast<float *>(&d_q),
          *pdh = reinterpret_cast<float *>(&d_h);
#pragma unroll
    for (int k = 0; k < kW; ++k) {
        const GruGates g = gru_gates(pgir[k], pgiz[k], pgin[k], pghr[k], pghz[k], pghn[k], b_hh ? pbr[k] : 0.0f,
                                     b_hh ? pbz[k] : 0.0f, b_hh ? pbn[k] : 0.0f);
        const float dht = pdhn[k] + (d_out ? pdout[k] : 0.0f);
        const float dn_pre = dht * (1.0f - g.z) * (1.0f - g.n * g.n);
        pdn[k] = dn_pre;
        pdq[k] = dn_pre * g.r;
        pdr[k] = dn_pre * g.q * g.r * (1.0f - g.r);
        pdz[k] = dht * (php[k] - g.n) * g.z * (1.0f - g.z);
        pdh[k] = dht * g.z;
    }
    *reinterpret_cast<V *>(gi_row + j) = d_r;
    *reinterpret_cast<V *>(gi_row + H + j) = d_z;
    *reinterpret_cast<V *>(gi_row + 2 * H + j) = d_n;
    *reinterpret_cast<V *>(gh_row + j) = d_r;
    *reinterpret_cast<V *>(gh_row + H + j) = d_z;
    *reinterpret_cast<V *>(gh_row + 2 * H + j) = d_q;
    *reinterpret_cast<V *>(dh + b * H + j) = d_h;
}

// The same pass with the bias gradients folded in: a block owns kBiasRows consecutive rows, every thread keeps one column
// chunk (kBlock % cols == 0) and sums what it writes — d_r, d_z, d_n, d_q — over its rows; the block's row groups are
// folded through LDS and the block leaves ONE partial row [4H] = {sum d_r, sum d_z, sum d_n, sum d_q}.  The caller sums the
// L x ceil(B / kBiasRows) partial rows once per layer: d b_ih = {r, z, n}, d b_hh = {r, z, q}.  Replaces two column-sum
// passes over the [L * B, 3H] gradient arrays (config 4: 2 x 408 MB read again per layer at the HBM roofline, 14 ms per
// iteration) by 2 x 17 MB of partial rows.  Fixed summation order (deterministic).
constexpr int kBiasRowsDefault = 16;
static int bias_rows() {
    const int r = int(option(kOptGruBiasRows));  // cusrl_set_option("gru_bias_rows", 4 | 8 | 16 | 32)
    return (r == 4 || r == 8 || r == 16 || r == 32) ? r : kBiasRowsDefault;
}

template <typename V>
__global__ __launch_bounds__(kBlock) void gru_gates_bwd_bias_kernel(float *__restrict__ gi, float *__restrict__ gh,
                                                                    const float *__restrict__ b_hh,
                                                                    const float *__restrict__ h_prev,
                                                                    const float *__restrict__ d_out, float *__restrict__ dh,
                                                                    const int64_t *__restrict__ lengths, int64_t t,
                                                                    int64_t B, int H, float *__restrict__ bias_partials,
                                                                    int kBiasRows) {
    constexpr int kW = sizeof(V) / sizeof(float);
    const int cols = H / kW;              // divides kBlock (checked by the entry point)
    const int groups = kBlock / cols;     // rows in flight per pass of the block
    const int col = threadIdx.x % cols, group = threadIdx.x / cols;
    const int j = col * kW;
    const int64_t row0 = int64_t(blockIdx.x) * kBiasRows;
    float acc[4][kW];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < kW; ++k) acc[g][k] = 0.0f;
    V br, bz, bn;
    if (b_hh) {
        br = *reinterpret_cast<const V *>(b_hh + j), bz = *reinterpret_cast<const V *>(b_hh + H + j),
        bn = *reinterpret_cast<const V *>(b_hh + 2 * H + j);
    }
    const float *pbr = reinterpret_cast<const float *>(&br), *pbz = reinterpret_cast<const float *>(&bz),
                *pbn = reinterpret_cast<const float *>(&bn);
    for (int r = group; r < kBiasRows; r += groups) {
        const int64_t b = row0 + r;
        if (b >= B) break;
        float *gi_row = gi + b * 3 * H, *gh_row = gh + b * 3 * H;
        const bool live = !lengths || t < lengths[b];
        V d_r, d_z, d_n, d_q;
        float *pdr = reinterpret_cast<float *>(&d_r), *pdz = reinterpret_cast<float *>(&d_z),
              *pdn = reinterpret_cast<float *>(&d_n), *pdq = reinterpret_cast<float *>(&d_q);
        if (!live) {  // ended sequence: no gate gradients, the state gradient passes through (dh stays as it is)
#pragma unroll
            for (int k = 0; k < kW; ++k) pdr[k] = pdz[k] = pdn[k] = pdq[k] = 0.0f;
        } else {
            const V gir = *reinterpret_cast<const V *>(gi_row + j), giz = *reinterpret_cast<const V *>(gi_row + H + j),
                    gin = *reinterpret_cast<const V *>(gi_row + 2 * H + j);
            const V ghr = *reinterpret_cast<const V *>(gh_row + j), ghz = *reinterpret_cast<const V *>(gh_row + H + j),
                    ghn = *reinterpret_cast<const V *>(gh_row + 2 * H + j);
            const V hp = *reinterpret_cast<const V *>(h_prev + b * H + j);
            const V dhn = *reinterpret_cast<const V *>(dh + b * H + j);
            V dout;
            if (d_out) dout = *reinterpret_cast<const V *>(d_out + b * H + j);
            const float *pgir = reinterpret_cast<const float *>(&gir), *pgiz = reinterpret_cast<const float *>(&giz),
                        *pgin = reinterpret_cast<const float *>(&gin), *pghr = reinterpret_cast<const float *>(&ghr),
                        *pghz = reinterpret_cast<const float *>(&ghz), *pghn = reinterpret_cast<const float *>(&ghn),
                        *php = reinterpret_cast<const float *>(&hp), *pdhn = reinterpret_cast<const float *>(&dhn),
                        *pdout = reinterpret_cast<const float *>(&dout);
            V d_h;
            float *pdh = reinterpret_cast<float *>(&d_h);
#pragma unroll
            for (int k = 0; k < kW; ++k) {
                const GruGates g = gru_gates(pgir[k], pgiz[k], pgin[k], pghr[k], pghz[k], pghn[k], b_hh ? pbr[k] : 0.0f,
                                             b_hh ? pbz[k] : 0.0f, b_hh ? pbn[k] : 0.0f);
                const float dht = pdhn[k] + (d_out ? pdout[k] : 0.0f);
                const float dn_pre = dht * (1.0f - g.z) * (1.0f - g.n * g.n);
                pdn[k] = dn_pre;
                pdq[k] = dn_pre * g.r;
                pdr[k] = dn_pre * g.q * g.r * (1.0f - g.r);
                pdz[k] = dht * (php[k] - g.n) * g.z * (1.0f - g.z);
                pdh[k] = dht * g.z;
                acc[0][k] += pdr[k], acc[1][k] += pdz[k], acc[2][k] += pdn[k], acc[3][k] += pdq[k];
            }
            *reinterpret_cast<V *>(dh + b * H + j) = d_h;
        }
        *reinterpret_cast<V *>(gi_row + j) = d_r;
        *reinterpret_cast<V *>(gi_row + H + j) = d_z;
        *reinterpret_cast<V *>(gi_row + 2 * H + j) = d_n;
        *reinterpret_cast<V *>(gh_row + j) = d_r;
        *reinterpret_cast<V *>(gh_row + H + j) = d_z;
        *reinterpret_cast<V *>(gh_row + 2 * H + j) = d_q;
    }
    // fold the block's row groups (same column chunk: threads col, col + cols, ...), group 0 first
    __shared__ float fold[kBlock][4 * kW];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < kW; ++k) fold[threadIdx.x][g * kW + k] = acc[g][k];
    __syncthreads();
    if (group == 0) {
        float *out = bias_partials + int64_t(blockIdx.x) * 4 * H;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float total[kW];
#pragma unroll
            for (int k = 0; k < kW; ++k) total[k] = 0.0f;
            for (int q = 0; q < groups; ++q)
#pragma unroll
                for (int k = 0; k < kW; ++k) total[k] += fold[q * cols + col][g * kW + k];
#pragma unroll
            for (int k = 0; k < kW; ++k) out[g * H + j + k] = total[k];
        }
    }
}


// ---- LSTM (the default core of RecurrentPpoAgentFactory, cusrl/preset/ppo.py:189) --------------------------------------
//   pre = gi + gh + b_hh   (gi = W_ih x + b_ih, gh = W_hh h; gate order i, f, g, o as torch.nn.LSTM)
//   i, f, o = sigmoid(pre_i, pre_f, pre_o);  g = tanh(pre_g);  c' = f * c + i * g;  h' = o * tanh(c')
// Both biases are additive, so d(pre) is the gradient of gi AND of gh: the forward stores `pre` over gi (kSave) and the
// backward overwrites it with d(pre) — one [B, 4H] array per step is all the layer keeps besides the cell states.
template <typename V, bool kSave>
__global__ __launch_bounds__(kBlock) void lstm_gates_fwd_kernel(float *__restrict__ gi, const float *__restrict__ gh,
                                                                const float *__restrict__ b_hh, float *__restrict__ h,
                                                                float *__restrict__ c, float *__restrict__ out,
                                                                float *__restrict__ c_saved,
                                                                const int64_t *__restrict__ lengths, int64_t t,
                                                                int64_t B, int H) {
    constexpr int kW = sizeof(V) / sizeof(float);
    const int cols = H / kW;
    const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= B * cols) return;
    const int64_t b = e / cols;
    const int j = int(e - b * cols) * kW;
    float *gi_row = gi + b * 4 * H;
    const float *gh_row = gh + b * 4 * H;
    V pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const V a = *reinterpret_cast<const V *>(gi_row + g * H + j), r = *reinterpret_cast<const V *>(gh_row + g * H + j);
        V bias;
        if (b_hh) bias = *reinterpret_cast<const V *>(b_hh + g * H + j);
        const float *pa = reinterpret_cast<const float *>(&a), *pr = reinterpret_cast<const float *>(&r),
                    *pb = reinterpret_cast<const float *>(&bias);
        float *pp = reinterpret_cast<float *>(&pre[g]);
#pragma unroll
        for (int k = 0; k < kW; ++k) pp[k] = pa[k] + pr[k] + (b_hh ? pb[k] : 0.0f);
    }
    const V hp = *reinterpret_cast<const V *>(h + b * H + j), cp = *reinterpret_cast<const V *>(c + b * H + j);
    const bool live = !lengths || t < lengths[b];
    V hn, cn, o;
    const float *pi = reinterpret_cast<const float *>(&pre[0]), *pf = reinterpret_cast<const float *>(&pre[1]),
                *pg = reinterpret_cast<const float *>(&pre[2]), *po = reinterpret_cast<const float *>(&pre[3]),
                *php = reinterpret_cast<const float *>(&hp), *pcp = reinterpret_cast<const float *>(&cp);
    float *phn = reinterpret_cast<float *>(&hn), *pcn = reinterpret_cast<float *>(&cn), *pout = reinterpret_cast<float *>(&o);
#pragma unroll
    for (int k = 0; k < kW; ++k) {
        const float c_next = gru_sigmoid(pf[k]) * pcp[k] + gru_sigmoid(pi[k]) * tanhf(pg[k]);
        const float h_next = gru_sigmoid(po[k]) * tanhf(c_next);
        pcn[k] = live ? c_next : pcp[k];
        phn[k] = live ? h_next : php[k];
        pout[k] = live ? h_next : 0.0f;
    }
    *reinterpret_cast<V *>(h + b * H + j) = hn;
    *reinterpret_cast<V *>(c + b * H + j) = cn;
    *reinterpret_cast<V *>(out + b * H + j) = o;
    if constexpr (kSave) {
        *reinterpret_cast<V *>(c_saved + b * H + j) = cn;
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<V *>(gi_row + g * H + j) = pre[g];
    }
}

// pre <- d(loss)/d(pre) in place; dc <- gradient of c_{t-1}; dh <- the part of the state gradient that bypasses this step
// (all of it for an ended sequence, nothing otherwise): the caller then adds d(pre) @ W_hh.
template <typename V>
__global__ __launch_bounds__(kBlock) void lstm_gates_bwd_kernel(float *__restrict__ pre, const float *__restrict__ c_prev,
                                                                const float *__restrict__ c_next,
                                                                const float *__restrict__ d_out, float *__restrict__ dh,
                                                                float *__restrict__ dc,
                                                                const int64_t *__restrict__ lengths, int64_t t,
                                                                int64_t B, int H) {
    constexpr int kW = sizeof(V) / sizeof(float);
    const int cols = H / kW;
    const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= B * cols) return;
    const int64_t b = e / cols;
    const int j = int(e - b * cols) * kW;
    float *row = pre + b * 4 * H;
    const bool live = !lengths || t < lengths[b];
    V zero;
    {
        float *pz = reinterpret_cast<float *>(&zero);
#pragma unroll
        for (int k = 0; k < kW; ++k) pz[k] = 0.0f;
    }
    if (!live) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<V *>(row + g * H + j) = zero;
        return;  // dh and dc pass through untouched
    }
    const V vi = *reinterpret_cast<const V *>(row + j), vf = *reinterpret_cast<const V *>(row + H + j),
            vg = *reinterpret_cast<const V *>(row + 2 * H + j), vo = *reinterpret_cast<const V *>(row + 3 * H + j);
    const V vcp = *reinterpret_cast<const V *>(c_prev + b * H + j), vcn = *reinterpret_cast<const V *>(c_next + b * H + j);
    const V vdh = *reinterpret_cast<const V *>(dh + b * H + j), vdc = *reinterpret_cast<const V *>(dc + b * H + j);
    V vdo;
    if (d_out) vdo = *reinterpret_cast<const V *>(d_out + b * H + j);
    const float *pi = reinterpret_cast<const float *>(&vi), *pf = reinterpret_cast<const float *>(&vf),
                *pg = reinterpret_cast<const float *>(&vg), *po = reinterpret_cast<const float *>(&vo),
                *pcp = reinterpret_cast<const float *>(&vcp), *pcn = reinterpret_cast<const float *>(&vcn),
                *pdh = reinterpret_cast<const float *>(&vdh), *pdc = reinterpret_cast<const float *>(&vdc),
                *pdo = reinterpret_cast<const float *>(&vdo);
    V di, df, dg, dO, dcp;
    float *qi = reinterpret_cast<float *>(&di), *qf = reinterpret_cast<float *>(&df), *qg = reinterpret_cast<float *>(&dg),
          *qo = reinterpret_cast<float *>(&dO), *qc = reinterpret_cast<float *>(&dcp);
#pragma unroll
    for (int k = 0; k < kW; ++k) {
        const float i = gru_sigmoid(pi[k]), f = gru_sigmoid(pf[k]), g = tanhf(pg[k]), o = gru_sigmoid(po[k]);
        const float tc = tanhf(pcn[k]);
        const float dht = pdh[k] + (d_out ? pdo[k] : 0.0f);
        const float dct = pdc[k] + dht * o * (1.0f - tc * tc);
        qo[k] = dht * tc * o * (1.0f - o);
        qi[k] = dct * g * i * (1.0f - i);
        qf[k] = dct * pcp[k] * f * (1.0f - f);
        qg[k] = dct * i * (1.0f - g * g);
        qc[k] = dct * f;
    }
    *reinterpret_cast<V *>(row + j) = di;
    *reinterpret_cast<V *>(row + H + j) = df;
    *reinterpret_cast<V *>(row + 2 * H + j) = dg;
    *reinterpret_cast<V *>(row + 3 * H + j) = dO;
    *reinterpret_cast<V *>(dc + b * H + j) = dcp;
    *reinterpret_cast<V *>(dh + b * H + j) = zero;
}

// ---- vanilla RNN (torch.nn.RNN, tanh | relu): h' = act(gi + gh + b_hh).  The output IS the saved state: the backward
// needs act'(h') only, so nothing but `out` is kept; d(pre) is written over gi.
template <typename V, bool kRelu>
__global__ __launch_bounds__(kBlock) void rnn_cell_fwd_kernel(const float *__restrict__ gi, const float *__restrict__ gh,
                                                              const float *__restrict__ b_hh, float *__restrict__ h,
                                                              float *__restrict__ out,
                                                              const int64_t *__restrict__ lengths, int64_t t, int64_t B,
                                                              int H) {
    constexpr int kW = sizeof(V) / sizeof(float);
    const int cols = H / kW;
    const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= B * cols) return;
    const int64_t b = e / cols;
    const int j = int(e - b * cols) * kW;
    const V a = *reinterpret_cast<const V *>(gi + b * H + j), r = *reinterpret_cast<const V *>(gh + b * H + j);
    const V hp = *reinterpret_cast<const V *>(h + b * H + j);
    V bias;
    if (b_hh) bias = *reinterpret_cast<const V *>(b_hh + j);
    const bool live = !lengths || t < lengths[b];
    const float *pa = reinterpret_cast<const float *>(&a), *pr = reinterpret_cast<const float *>(&r),
                *pb = reinterpret_cast<const float *>(&bias), *php = reinterpret_cast<const float *>(&hp);
    V hn, o;
    float *phn = reinterpret_cast<float *>(&hn), *po = reinterpret_cast<float *>(&o);
#pragma unroll
    for (int k = 0; k < kW; ++k) {
        const float pre = pa[k] + pr[k] + (b_hh ? pb[k] : 0.0f);
        const float next = kRelu ? fmaxf(pre, 0.0f) : tanhf(pre);
        phn[k] = live ? next : php[k];
        po[k] = live ? next : 0.0f;
    }
    *reinterpret_cast<V *>(h + b * H + j) = hn;
    *reinterpret_cast<V *>(out + b * H + j) = o;
}

template <typename V, bool kRelu>
__global__ __launch_bounds__(kBlock) void rnn_cell_bwd_kernel(float *__restrict__ d_pre, const float *__restrict__ out,
                                                              const float *__restrict__ d_out, float *__restrict__ dh,
                                                              const int64_t *__restrict__ lengths, int64_t t, int64_t B,
                                                              int H) {
    constexpr int kW = sizeof(V) / sizeof(float);
    const int cols = H / kW;
    const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= B * cols) return;
    const int64_t b = e / cols;
    const int j = int(e - b * cols) * kW;
    const bool live = !lengths || t < lengths[b];
    V g;
    float *pg = reinterpret_cast<float *>(&g);
    if (!live) {
#pragma unroll
        for (int k = 0; k < kW; ++k) pg[k] = 0.0f;
        *reinterpret_cast<V *>(d_pre + b * H + j) = g;
        return;  // dh passes through
    }
    const V y = *reinterpret_cast<const V *>(out + b * H + j), vdh = *reinterpret_cast<const V *>(dh + b * H + j);
    V vdo;
    if (d_out) vdo = *reinterpret_cast<const V *>(d_out + b * H + j);
    const float *py = reinterpret_cast<const float *>(&y), *pdh = reinterpret_cast<const float *>(&vdh),
                *pdo = reinterpret_cast<const float *>(&vdo);
    V zero;
    float *pz = reinterpret_cast<float *>(&zero);
#pragma unroll
    for (int k = 0; k < kW; ++k) {
        const float dht = pdh[k] + (d_out ? pdo[k] : 0.0f);
        pg[k] = kRelu ? (py[k] > 0.0f ? dht : 0.0f) : dht * (1.0f - py[k] * py[k]);
        pz[k] = 0.0f;
    }
    *reinterpret_cast<V *>(d_pre + b * H + j) = g;
    *reinterpret_cast<V *>(dh + b * H + j) = zero;
}

inline bool gru_vec4(int64_t H, const void *a, const void *b, const void *c, const void *d, const void *e,
                     const void *f) {
    auto ok = [](const void *p) { return p == nullptr || aligned(p, 16); };
    return H % 4 == 0 && ok(a) && ok(b) && ok(c) && ok(d) && ok(e) && ok(f);
}

}  // namespace cusrl

extern "C" int cusrl_gru_gates_fwd(const float *gi, const float *gh, const float *b_hh, float *h, float *out,
                                   const int64_t *lengths, int64_t t, int64_t B, int64_t H, void *stream) {
    using namespace cusrl;
    if (B < 0 || H <= 0 || t < 0) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!gi || !gh || !h || !out) return CUSRL_E_INVALID;
    if (H > INT32_MAX / 3) return CUSRL_E_UNSUPPORTED;
    const bool vec4 = gru_vec4(H, gi, gh, b_hh, h, out, nullptr);
    const int64_t blocks = ceil_div(B * (vec4 ? H / 4 : H), kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    if (vec4)
        hipLaunchKernelGGL(gru_gates_fwd_kernel<float4>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), gi,
                           gh, b_hh, h, out, lengths, t, B, int(H));
    else
        hipLaunchKernelGGL(gru_gates_fwd_kernel<float>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), gi,
                           gh, b_hh, h, out, lengths, t, B, int(H));
    return launch_status();
}

extern "C" int cusrl_gru_gates_bwd(float *gi, float *gh, const float *b_hh, const float *h_prev, const float *d_out,
                                   float *dh, const int64_t *lengths, int64_t t, int64_t B, int64_t H, void *stream) {
    using namespace cusrl;
    if (B < 0 || H <= 0 || t < 0) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!gi || !gh || !h_prev || !dh) return CUSRL_E_INVALID;
    if (H > INT32_MAX / 3) return CUSRL_E_UNSUPPORTED;
    const bool vec4 = gru_vec4(H, gi, gh, b_hh, h_prev, d_out, dh);
    const int64_t blocks = ceil_div(B * (vec4 ? H / 4 : H), kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    if (vec4)
        hipLaunchKernelGGL(gru_gates_bwd_kernel<float4>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), gi,
                           gh, b_hh, h_prev, d_out, dh, lengths, t, B, int(H));
    else
        hipLaunchKernelGGL(gru_gates_bwd_kernel<float>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), gi,
                           gh, b_hh, h_prev, d_out, dh, lengths, t, B, int(H));
    return launch_status();
}

extern "C" int64_t cusrl_gru_bias_partial_rows(int64_t B) { return B <= 0 ? 0 : cusrl::ceil_div(B, cusrl::bias_rows()); }

// 1: cusrl_gru_gates_bwd_bias takes a launch with these pointers (every other argument valid); 0: CUSRL_E_UNSUPPORTED.
// The ONE statement of that eligibility rule: the host asks here instead of restating it.
extern "C" int cusrl_gru_bias_supported(int64_t H, const float *gi, const float *gh, const float *b_hh, const float *h_prev,
                                        const float *d_out, const float *dh, const float *bias_partials) {
    using namespace cusrl;
    if (H <= 0 || H > INT32_MAX / 3) return 0;
    const bool vec4 = gru_vec4(H, gi, gh, b_hh, h_prev, d_out, dh) && aligned(bias_partials, 16);
    const int64_t cols = vec4 ? H / 4 : H;
    return cols <= kBlock && kBlock % cols == 0 && kBlock / cols <= bias_rows();
}

extern "C" int cusrl_gru_gates_bwd_bias(float *gi, float *gh, const float *b_hh, const float *h_prev, const float *d_out,
                                        float *dh, const int64_t *lengths, int64_t t, int64_t B, int64_t H,
                                        float *bias_partials, void *stream) {
    using namespace cusrl;
    if (B < 0 || H <= 0 || t < 0) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!gi || !gh || !h_prev || !dh || !bias_partials) return CUSRL_E_INVALID;
    // (wide or narrow layers whose column chunks do not tile a block: the caller keeps the plain pass + column sums)
    if (!cusrl_gru_bias_supported(H, gi, gh, b_hh, h_prev, d_out, dh, bias_partials)) return CUSRL_E_UNSUPPORTED;
    const bool vec4 = gru_vec4(H, gi, gh, b_hh, h_prev, d_out, dh) && aligned(bias_partials, 16);
    const int rows_per_block = bias_rows();
    const int64_t blocks = ceil_div(B, rows_per_block);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    if (vec4)
        hipLaunchKernelGGL(gru_gates_bwd_bias_kernel<float4>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), gi,
                           gh, b_hh, h_prev, d_out, dh, lengths, t, B, int(H), bias_partials, rows_per_block);
    else
        hipLaunchKernelGGL(gru_gates_bwd_bias_kernel<float>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), gi,
                           gh, b_hh, h_prev, d_out, dh, lengths, t, B, int(H), bias_partials, rows_per_block);
    return launch_status();
}

extern "C" int cusrl_lstm_gates_fwd(float *gi, const float *gh, const float *b_hh, float *h, float *c, float *out,
                                    float *c_saved, const int64_t *lengths, int64_t t, int64_t B, int64_t H,
                                    void *stream) {
    using namespace cusrl;
    if (B < 0 || H <= 0 || t < 0) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!gi || !gh || !h || !c || !out) return CUSRL_E_INVALID;
    if (H > INT32_MAX / 4) return CUSRL_E_UNSUPPORTED;
    const bool vec4 = gru_vec4(H, gi, gh, b_hh, h, c, out) && (c_saved == nullptr || aligned(c_saved, 16));
    const int64_t blocks = ceil_div(B * (vec4 ? H / 4 : H), kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const dim3 grid{uint32_t(blocks)}, block{kBlock};
    hipStream_t s = as_stream(stream);
    if (vec4 && c_saved)
        hipLaunchKernelGGL((lstm_gates_fwd_kernel<float4, true>), grid, block, 0, s, gi, gh, b_hh, h, c, out, c_saved, lengths, t, B, int(H));
    else if (vec4)
        hipLaunchKernelGGL((lstm_gates_fwd_kernel<float4, false>), grid, block, 0, s, gi, gh, b_hh, h, c, out, c_saved, lengths, t, B, int(H));
    else if (c_saved)
        hipLaunchKernelGGL((lstm_gates_fwd_kernel<float, true>), grid, block, 0, s, gi, gh, b_hh, h, c, out, c_saved, lengths, t, B, int(H));
    else
        hipLaunchKernelGGL((lstm_gates_fwd_kernel<float, false>), grid, block, 0, s, gi, gh, b_hh, h, c, out, c_saved, lengths, t, B, int(H));
    return launch_status();
}

extern "C" int cusrl_lstm_gates_bwd(float *pre, const float *c_prev, const float *c_next, const float *d_out, float *dh,
                                    float *dc, const int64_t *lengths, int64_t t, int64_t B, int64_t H, void *stream) {
    using namespace cusrl;
    if (B < 0 || H <= 0 || t < 0) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!pre || !c_prev || !c_next || !dh || !dc) return CUSRL_E_INVALID;
    if (H > INT32_MAX / 4) return CUSRL_E_UNSUPPORTED;
    const bool vec4 = gru_vec4(H, pre, c_prev, c_next, d_out, dh, dc);
    const int64_t blocks = ceil_div(B * (vec4 ? H / 4 : H), kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    if (vec4)
        hipLaunchKernelGGL(lstm_gates_bwd_kernel<float4>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), pre,
                           c_prev, c_next, d_out, dh, dc, lengths, t, B, int(H));
    else
        hipLaunchKernelGGL(lstm_gates_bwd_kernel<float>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), pre,
                           c_prev, c_next, d_out, dh, dc, lengths, t, B, int(H));
    return launch_status();
}

extern "C" int cusrl_rnn_cell_fwd(const float *gi, const float *gh, const float *b_hh, float *h, float *out,
                                  const int64_t *lengths, int64_t t, int64_t B, int64_t H, int relu, void *stream) {
    using namespace cusrl;
    if (B < 0 || H <= 0 || t < 0) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!gi || !gh || !h || !out) return CUSRL_E_INVALID;
    if (H > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const bool vec4 = gru_vec4(H, gi, gh, b_hh, h, out, nullptr);
    const int64_t blocks = ceil_div(B * (vec4 ? H / 4 : H), kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const dim3 grid{uint32_t(blocks)}, block{kBlock};
    hipStream_t s = as_stream(stream);
    if (vec4 && relu) hipLaunchKernelGGL((rnn_cell_fwd_kernel<float4, true>), grid, block, 0, s, gi, gh, b_hh, h, out, lengths, t, B, int(H));
    else if (vec4) hipLaunchKernelGGL((rnn_cell_fwd_kernel<float4, false>), grid, block, 0, s, gi, gh, b_hh, h, out, lengths, t, B, int(H));
    else if (relu) hipLaunchKernelGGL((rnn_cell_fwd_kernel<float, true>), grid, block, 0, s, gi, gh, b_hh, h, out, lengths, t, B, int(H));
    else hipLaunchKernelGGL((rnn_cell_fwd_kernel<float, false>), grid, block, 0, s, gi, gh, b_hh, h, out, lengths, t, B, int(H));
    return launch_status();
}

extern "C" int cusrl_rnn_cell_bwd(float *d_pre, const float *out, const float *d_out, float *dh, const int64_t *lengths,
                                  int64_t t, int64_t B, int64_t H, int relu, void *stream) {
    using namespace cusrl;
    if (B < 0 || H <= 0 || t < 0) return CUSRL_E_INVALID;
    if (B == 0) return 0;
    if (!d_pre || !out || !dh) return CUSRL_E_INVALID;
    if (H > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const bool vec4 = gru_vec4(H, d_pre, out, d_out, dh, nullptr, nullptr);
    const int64_t blocks = ceil_div(B * (vec4 ? H / 4 : H), kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const dim3 grid{uint32_t(blocks)}, block{kBlock};
    hipStream_t s = as_stream(stream);
    if (vec4 && relu) hipLaunchKernelGGL((rnn_cell_bwd_kernel<float4, true>), grid, block, 0, s, d_pre, out, d_out, dh, lengths, t, B, int(H));
    else if (vec4) hipLaunchKernelGGL((rnn_cell_bwd_kernel<float4, false>), grid, block, 0, s, d_pre, out, d_out, dh, lengths, t, B, int(H));
    else if (relu) hipLaunchKernelGGL((rnn_cell_bwd_kernel<float, true>), grid, block, 0, s, d_pre, out, d_out, dh, lengths, t, B, int(H));
    else hipLaunchKernelGGL((rnn_cell_bwd_kernel<float, false>), grid, block, 0, s, d_pre, out, d_out, dh, lengths, t, B, int(H));
    return launch_status();
}
