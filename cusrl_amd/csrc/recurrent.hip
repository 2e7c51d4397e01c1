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
                                                                 uint8_t *__restrict__ mask) {
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
        if (done[int64_t(t) * N + n]) {
            ++seq;
            pos = 0;
        } else {
            ++pos;
        }
    }
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
                                     uint8_t *mask, void *stream) {
    if (L <= 0 || N <= 0 || Ns <= 0 || !done || !env_prefix || !block_totals || !dest) return CUSRL_E_INVALID;
    const int64_t blocks = cusrl_sequence_blocks(N);
    hipLaunchKernelGGL(sequence_layout_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), done, int(L),
                       N, env_prefix, block_totals, Ns, dest, first_seq, mask);
    return launch_status();
}
