// Step append of every leaf of a transition (a1, cusrl/template/buffer.py:124-151) — shared by the plain push launch
// (buffer.hip) and the fused step-epilogue + push launch (rollout.hip): table layout, per-block body, host-side table build.
#pragma once

#include <stdlib.h>

#include "common.hpp"

namespace cusrl {

constexpr int64_t kPushBlockBytes = int64_t(kBlock) * 16 * 2;  // two independent 16 B transactions per lane

// Leaf table passed BY VALUE in the kernarg segment.  Kernarg reads are scalar loads that miss to (host-visible)
// kernarg memory with microsecond latency, so the layout and the lookup are built for at most TWO dependent round
// trips per wave: (1) `n` + the whole block_start[] prefix array (104 B, fetched by a few wide independent
// s_loads, the block->leaf search is branch-free over all entries), (2) the one 24 B descriptor of the leaf found.
// (A first version walked the prefix array with a dependent load per entry: 13 serial misses = 17-30 us per launch.)
struct PushLeaf {
    const char *src;
    char *dst;
    int64_t bytes;
    // write-through into the per-slot record (cusrl_buffer_push_through): the step's row n additionally goes to
    // dst2 + n * pitch2; nullptr = the leaf does not live in the record
    char *dst2;
    uint32_t row_bytes, pitch2;
};

struct PushTable {
    int32_t n;
    int32_t block_start[CUSRL_MAX_FIELDS + 1];
    PushLeaf leaf[CUSRL_MAX_FIELDS];
};

template <typename Table>
__device__ __forceinline__ int find_leaf(const Table &tab, int blk) {
    int f = 0;
#pragma unroll
    for (int i = 1; i < CUSRL_MAX_FIELDS; ++i) f += (i < tab.n && blk >= tab.block_start[i]) ? 1 : 0;
    return f;
}

typedef uint32_t native_u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 stream_load16(const char *p) {
    if constexpr (NT) {
        const native_u4 v = __builtin_nontemporal_load(reinterpret_cast<const native_u4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const uint4 *>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void stream_store16(char *p, const uint4 &v) {
    if constexpr (NT) {
        const native_u4 n = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(n, reinterpret_cast<native_u4 *>(p));
    } else {
        *reinterpret_cast<uint4 *>(p) = v;
    }
}

// POLICY: bit 0 = non-temporal loads of the step's tensors, bit 1 = non-temporal stores into the buffer's slabs (chosen by
// the step's footprint, see push_fields).
template <int POLICY>
__device__ __forceinline__ void push_body(const PushTable &tab, const int blk) {
    const int f = find_leaf(tab, blk);
    const PushLeaf leaf = tab.leaf[f];
    const char *__restrict__ src = leaf.src;
    char *__restrict__ dst = leaf.dst;
    const int64_t total = leaf.bytes;
    const int64_t begin = int64_t(blk - tab.block_start[f]) * kPushBlockBytes;
    const int64_t end = min(begin + kPushBlockBytes, total);
    const uintptr_t align = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | uintptr_t(total);
    if ((align & 15) == 0) {
        const int64_t o0 = begin + int64_t(threadIdx.x) * 16;
        const int64_t o1 = o0 + int64_t(kBlock) * 16;
        uint4 r0, r1;
        const bool p0 = o0 < end, p1 = o1 < end;
        if (p0) r0 = stream_load16<(POLICY & 1) != 0>(src + o0);
        if (p1) r1 = stream_load16<(POLICY & 1) != 0>(src + o1);
        if (p0) stream_store16<(POLICY & 2) != 0>(dst + o0, r0);
        if (p1) stream_store16<(POLICY & 2) != 0>(dst + o1, r1);
        if (leaf.dst2) {  // uniform per block: the same 16 bytes, once more, at the row's place inside its record
            const uint32_t rb = leaf.row_bytes;  // a multiple of 16, so a 16-byte lane-op never straddles two rows
            if (p0) {
                const uint32_t row = uint32_t(o0) / rb;
                *reinterpret_cast<uint4 *>(leaf.dst2 + uint64_t(row) * leaf.pitch2 + (uint32_t(o0) - row * rb)) = r0;
            }
            if (p1) {
                const uint32_t row = uint32_t(o1) / rb;
                *reinterpret_cast<uint4 *>(leaf.dst2 + uint64_t(row) * leaf.pitch2 + (uint32_t(o1) - row * rb)) = r1;
            }
        }
    } else if ((align & 3) == 0) {
        for (int64_t o = begin + int64_t(threadIdx.x) * 4; o < end; o += int64_t(kBlock) * 4)
            *reinterpret_cast<uint32_t *>(dst + o) = *reinterpret_cast<const uint32_t *>(src + o);
    } else {
        for (int64_t o = begin + threadIdx.x; o < end; o += kBlock) dst[o] = src[o];
    }
}


template <int POLICY>
__global__ __launch_bounds__(kBlock) void push_kernel(const PushTable tab) {
    push_body<POLICY>(tab, blockIdx.x);
}

// a step whose read + written bytes do not fit the 256 MB Infinity Cache streams past the caches (0.72 -> 0.79 of the
// roofline at 1 M envs, profiles/r04/push_policy.txt); cusrl_set_option("push_policy", 1 | 2) forces the default / the
// streaming form (A/B measurements)
inline bool push_streams(int64_t step_bytes) {
    const int64_t forced = option(kOptPushPolicy);
    return forced ? forced == 2 : 2 * step_bytes >= (int64_t(256) << 20);
}

// Fills `tab` from the caller's field list (validation included); `skip` = index of a field that is NOT to be copied (the
// fused step epilogue writes that leaf itself), -1 = none.  Returns 0 or a CUSRL_E_* code.
inline int build_push_table(const cusrl_field_t *fields, int n_fields, int64_t cursor, int64_t N, char *record,
                            int64_t record_bytes, const int32_t *record_offset, int skip, PushTable &tab, int32_t &blocks,
                            int64_t &step_bytes) {
    if (!fields || n_fields < 0 || cursor < 0 || N < 0) return CUSRL_E_INVALID;
    if (n_fields > CUSRL_MAX_FIELDS) return CUSRL_E_TOO_MANY;
    if (record_offset && (!record || record_bytes < 16 || record_bytes % 16 != 0 || record_bytes > CUSRL_MAX_RECORD_BYTES ||
                          !aligned(record, 16)))
        return CUSRL_E_INVALID;
    blocks = 0;
    step_bytes = 0;
    int n = 0;
    for (int i = 0; i < n_fields; ++i) {
        const int64_t bytes = N * fields[i].row_bytes;
        if (fields[i].row_bytes < 0 || (bytes > 0 && (!fields[i].src || !fields[i].dst))) return CUSRL_E_INVALID;
        const int32_t offset = record_offset ? record_offset[i] : -1;
        if (offset >= 0) {
            // a leaf written through must take the 16-byte-lane path: whole 16-byte chunks at a 16-byte offset of the record
            const int64_t rb = fields[i].row_bytes;
            if (rb <= 0 || rb % 16 != 0 || offset % 16 != 0 || offset + rb > record_bytes) return CUSRL_E_INVALID;
            if (!aligned(fields[i].src, 16) || !aligned(fields[i].dst, 16) || bytes > int64_t(UINT32_MAX))
                return CUSRL_E_UNSUPPORTED;
        }
        if (bytes == 0 || i == skip) continue;
        step_bytes += bytes;
        tab.leaf[n].src = static_cast<const char *>(fields[i].src);
        tab.leaf[n].dst = static_cast<char *>(fields[i].dst) + cursor * bytes;
        tab.leaf[n].bytes = bytes;
        tab.leaf[n].dst2 = offset >= 0 ? record + cursor * N * record_bytes + offset : nullptr;
        tab.leaf[n].row_bytes = uint32_t(fields[i].row_bytes);
        tab.leaf[n].pitch2 = uint32_t(record_bytes);
        tab.block_start[n] = blocks;
        const int64_t nb = ceil_div(bytes, kPushBlockBytes);
        if (nb + blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
        blocks += int32_t(nb);
        ++n;
    }
    for (int i = n; i <= CUSRL_MAX_FIELDS; ++i) tab.block_start[i] = blocks;
    tab.n = n;
    return 0;
}

}  // namespace cusrl
