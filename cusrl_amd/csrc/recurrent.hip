// Done-split sequence layout for recurrent (BPTT) minibatches on gfx950.
// The reference builds it from nonzero / cumsum / boolean-mask scatters with several host synchronisations
// (cusrl/nn/utils/recurrent.py:63-92, 160-252); here one lane walks the L steps of one env (L = 24), a block scan and
// a block-totals prefix give the env-major sequence numbering, and a second launch emits, for every slot (t, n), its
// row in the padded [L, Ns] layout.  Moving the data itself is cusrl_scatter_rows / cusrl_gather_rows.
#include "common.hpp"

namespace cusrl {

__global__ __launch_bounds__(kBlock) void sequence_count_kernel(const uint8_t *__restrict__ done, int L, int64_t N,
                                                                int32_t *__restrict__ env_prefix,
                                                                int32_t *__restrict__ block_totals) {
    __shared__ int scratch[kWavesPerBlock];
    const int64_t n = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    int count = 0;
    if (n < N) {
        count = 1;  // the last step always closes a sequence (recurrent.py:83)
        for (int t = 0; t + 1 < L; ++t) count += done[int64_t(t) * N + n] != 0;
    }
    int total;
    const int prefix = block_exclusive_scan(count, scratch, total);
    if (n < N) env_prefix[n] = prefix;
    if (threadIdx.x == 0) block_totals[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void sequence_total_kernel(const int32_t *__restrict__ block_totals, int blocks,
                                                                int32_t *__restrict__ num_sequences) {
    __shared__ int scratch[kWavesPerBlock];
    int s = 0;
    for (int i = threadIdx.x; i < blocks; i += kBlock) s += block_totals[i];
    const int total = block_sum(s, scratch);
    if (threadIdx.x == 0) *num_sequences = total;
}

__global__ __launch_bounds__(kBlock) void sequence_layout_kernel(const uint8_t *__restrict__ done, int L, int64_t N,
                                                                 const int32_t *__restrict__ env_prefix,
                                                                 const int32_t *__restrict__ block_totals, int64_t Ns,
                                                                 int64_t *__restrict__ dest,
                                                                 int64_t *__restrict__ first_seq,
                                                                 uint8_t *__restrict__ mask,
                                                                 int64_t *__restrict__ seq_lengths,
                                                                 int64_t *__restrict__ last_seq) {
    __shared__ int scratch[kWavesPerBlock];
    __shared__ int block_offset;
    int before = 0;
    for (int i = threadIdx.x; i < int(blockIdx.x); i += kBlock) before += block_totals[i];
    const int offset = block_sum(before, scratch);
    if (threadIdx.x == 0) block_offset = offset;
    __syncthreads();
    const int64_t n = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (n >= N) return;
    int64_t seq = int64_t(block_offset) + env_prefix[n];
    if (first_seq) first_seq[n] = seq;
    int pos = 0;
    for (int t = 0; t < L; ++t) {
        const int64_t slot = int64_t(pos) * Ns + seq;
        dest[int64_t(t) * N + n] = slot;
        if (mask) mask[slot] = 1;
        if (done[int64_t(t) * N + n] || t == L - 1) {  // the last step always closes a sequence (recurrent.py:83)
            if (seq_lengths) seq_lengths[seq] = pos + 1;
            if (t == L - 1 && last_seq) last_seq[n] = seq;
            ++seq;
            pos = 0;
        } else {
            ++pos;
        }
    }
}

// gather_memory (recurrent.py:124-157): dst[n] = src[last_seq[n]] — the state of the sequence still running at the end
// of env n's column — cleared where the env finished exactly at the last step.  16-byte lanes, rows of H floats.
template <typename V>
__global__ __launch_bounds__(kBlock) void gather_memory_kernel(const char *__restrict__ src,
                                                               const int64_t *__restrict__ last_seq,
                                                               const uint8_t *__restrict__ done_last,
                                                               char *__restrict__ dst, int64_t N, int lpr,
                                                               int64_t row_bytes) {
    const int64_t op = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (op >= N * lpr) return;
    const int64_t n = op / lpr, col = op - n * lpr;
    V value = *reinterpret_cast<const V *>(src + last_seq[n] * row_bytes + col * int64_t(sizeof(V)));
    if (done_last[n]) value = V{};
    *reinterpret_cast<V *>(dst + n * row_bytes + col * int64_t(sizeof(V))) = value;
}

}  // namespace cusrl

using namespace cusrl;

extern "C" int64_t cusrl_sequence_blocks(int64_t N) { return N <= 0 ? 0 : ceil_div(N, kBlock); }

extern "C" int cusrl_sequence_count(const uint8_t *done, int64_t L, int64_t N, int32_t *env_prefix,
                                    int32_t *block_totals, int32_t *num_sequences, void *stream) {
    if (L <= 0 || N <= 0 || !done || !env_prefix || !block_totals || !num_sequences) return CUSRL_E_INVALID;
    if (L > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    const int64_t blocks = cusrl_sequence_blocks(N);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(sequence_count_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, s, done, int(L), N, env_prefix,
                       block_totals);
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(sequence_total_kernel, dim3(1), dim3(kBlock), 0, s, block_totals, int(blocks), num_sequences);
    return launch_status();
}

extern "C" int cusrl_sequence_layout(const uint8_t *done, int64_t L, int64_t N, const int32_t *env_prefix,
                                     const int32_t *block_totals, int64_t Ns, int64_t *dest, int64_t *first_seq,
                                     uint8_t *mask, int64_t *seq_lengths, int64_t *last_seq, void *stream) {
    if (L <= 0 || N <= 0 || Ns <= 0 || !done || !env_prefix || !block_totals || !dest) return CUSRL_E_INVALID;
    const int64_t blocks = cusrl_sequence_blocks(N);
    hipLaunchKernelGGL(sequence_layout_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), done, int(L),
                       N, env_prefix, block_totals, Ns, dest, first_seq, mask, seq_lengths, last_seq);
    return launch_status();
}

#define CUSRL_LAUNCH_GATHER_MEMORY(V)                                                                                 \
    hipLaunchKernelGGL(gather_memory_kernel<V>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream),           \
                       static_cast<const char *>(memory), last_seq, done_last, static_cast<char *>(out), N, lpr, row_bytes)

extern "C" int cusrl_gather_memory(const void *memory, const int64_t *last_seq, const uint8_t *done_last, void *out,
                                   int64_t N, int64_t row_bytes, void *stream) {
    if (N == 0 || row_bytes == 0) return 0;
    if (!memory || !last_seq || !done_last || !out || N < 0 || row_bytes < 0) return CUSRL_E_INVALID;
    int unit = 1;
    for (int u : {16, 8, 4, 2})
        if (row_bytes % u == 0 && aligned(memory, u) && aligned(out, u)) {
            unit = u;
            break;
        }
    const int lpr = int(row_bytes / unit);
    const int64_t blocks = ceil_div(N * lpr, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    switch (unit) {
        case 16: CUSRL_LAUNCH_GATHER_MEMORY(uint4); break;
        case 8: CUSRL_LAUNCH_GATHER_MEMORY(uint2); break;
        case 4: CUSRL_LAUNCH_GATHER_MEMORY(uint32_t); break;
        case 2: CUSRL_LAUNCH_GATHER_MEMORY(uint16_t); break;
        default: CUSRL_LAUNCH_GATHER_MEMORY(uint8_t); break;
    }
    return launch_status();
}
