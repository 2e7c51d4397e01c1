// Rollout-buffer data movement for gfx950: step append (a1), minibatch gather (a7/a8), ordered flag
// compaction and row scatter (a3 support).  All kernels are HBM-bound byte movers: 16 B per lane where
// alignment allows, every leaf of a transition handled by ONE launch through a by-value leaf table
// (no device-side table upload, no per-leaf launches).
#include <stdlib.h>

#include "common.hpp"
#include "push_body.hpp"

namespace cusrl {

// --------------------------------------------------------------------------------------------- gather
constexpr int kGatherItems = 4;                       // lane-ops per thread, all loads issued before any store
constexpr int kGatherOpsPerBlock = kBlock * kGatherItems;

struct GatherLeaf {
    const char *src;
    char *dst;
    int32_t src_pitch;      // bytes between consecutive source rows: the row size, or the record size for a leaf that
                            // is read out of (dst_pitch: written into) the per-slot record
    int32_t unit;           // bytes per lane-op: 16 / 8 / 4 / 2 / 1;  0 = "four 1-byte rows packed"
    int32_t lanes_per_row;  // row bytes / unit
    int32_t dst_pitch;
};

struct GatherTable {  // same two-round-trip kernarg layout as PushTable
    int32_t n;
    int32_t block_start[CUSRL_MAX_FIELDS + 1];
    GatherLeaf leaf[CUSRL_MAX_FIELDS];
};

// Memory-level parallelism is the whole game here (every access is a dependent idx -> row -> store chain and, at
// config-2 sizes, the launch is latency-bound): loads are never predicated — out-of-range lane-ops are CLAMPED to
// the last valid op so the compiler can issue all index loads of a lane back to back, then all row loads, and only
// then the stores.  (Predicated loads made it wait after every single load: 27-40 us per 27 MB launch.)
// One empty asm statement that "uses and redefines" every loaded register: the loads must all be issued (and
// waited for) before it, the stores can only come after it.
__device__ __forceinline__ void pin_loaded(uint4 (&r)[kGatherItems]) {
    static_assert(kGatherItems == 4, "operand list below is written for 4 items");
    asm volatile(""
                 : "+v"(r[0].x), "+v"(r[0].y), "+v"(r[0].z), "+v"(r[0].w), "+v"(r[1].x), "+v"(r[1].y), "+v"(r[1].z),
                   "+v"(r[1].w), "+v"(r[2].x), "+v"(r[2].y), "+v"(r[2].z), "+v"(r[2].w), "+v"(r[3].x), "+v"(r[3].y),
                   "+v"(r[3].z), "+v"(r[3].w));
}
__device__ __forceinline__ void pin_loaded(uint2 (&r)[kGatherItems]) {
    asm volatile(""
                 : "+v"(r[0].x), "+v"(r[0].y), "+v"(r[1].x), "+v"(r[1].y), "+v"(r[2].x), "+v"(r[2].y), "+v"(r[3].x),
                   "+v"(r[3].y));
}
__device__ __forceinline__ void pin_loaded(uint32_t (&r)[kGatherItems]) {
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
}
__device__ __forceinline__ void pin_loaded(uint16_t (&r)[kGatherItems]) {
    uint32_t t[4] = {r[0], r[1], r[2], r[3]};
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
    for (int i = 0; i < 4; ++i) r[i] = uint16_t(t[i]);
}
__device__ __forceinline__ void pin_loaded(uint8_t (&r)[kGatherItems]) {
    uint32_t t[4] = {r[0], r[1], r[2], r[3]};
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
    for (int i = 0; i < 4; ++i) r[i] = uint8_t(t[i]);
}

template <typename V>
__device__ __forceinline__ void gather_unit(const char *__restrict__ src, char *__restrict__ dst,
                                            const int64_t *__restrict__ idx, int64_t ops, int64_t op0, int lpr,
                                            int64_t src_pitch, int64_t dst_pitch, int64_t B, int64_t N, bool temporal) {
    int64_t row[kGatherItems], col[kGatherItems], src_row[kGatherItems];
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it) {
        const int64_t op = min(op0 + int64_t(it) * kBlock, ops - 1);
        if (lpr == 1) {
            row[it] = op;
            col[it] = 0;
        } else {
            row[it] = op / lpr;
            col[it] = op - row[it] * lpr;
        }
    }
    if (idx == nullptr) {  // identity: a strided row copy (building the per-slot record from the leaves)
#pragma unroll
        for (int it = 0; it < kGatherItems; ++it) src_row[it] = row[it];
    } else if (temporal) {
#pragma unroll
        for (int it = 0; it < kGatherItems; ++it) {
            const int64_t t = row[it] / B, b = row[it] - t * B;
            src_row[it] = t * N + idx[b];
        }
    } else {
#pragma unroll
        for (int it = 0; it < kGatherItems; ++it) src_row[it] = idx[row[it]];
    }
    V regs[kGatherItems];
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it)
        regs[it] = *reinterpret_cast<const V *>(src + src_row[it] * src_pitch + col[it] * int64_t(sizeof(V)));
    pin_loaded(regs);  // all row loads are in flight before the first store (hipcc otherwise re-interleaves
                       // load / wait / store per item, i.e. one exposed memory latency per item)
    // stores are unconditional as well: a clamped lane-op rewrites the last element with the identical bytes, which
    // keeps the whole body branch-free (with masked stores LLVM sinks each load into its store's block again)
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it)
        *reinterpret_cast<V *>(dst + row[it] * dst_pitch + col[it] * int64_t(sizeof(V))) = regs[it];
}

// 1-byte leaves (terminated / truncated / done): one lane gathers 4 consecutive output rows (4 index loads, then 4
// byte loads, all unpredicated) and issues a single 4-byte store instead of four byte stores.
__device__ __forceinline__ void gather_bytes_packed(const char *__restrict__ src, char *__restrict__ dst,
                                                    const int64_t *__restrict__ idx, int64_t rows, int64_t op,
                                                    int64_t B, int64_t N, bool temporal) {
    const int64_t r0 = op * 4;
    if (r0 >= rows) return;
    int64_t src_row[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t r = min(r0 + j, rows - 1);
        if (temporal) {
            const int64_t t = r / B, b = r - t * B;
            src_row[j] = t * N + idx[b];
        } else {
            src_row[j] = idx[r];
        }
    }
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) packed |= uint32_t(uint8_t(src[src_row[j]])) << (8 * j);
    if (r0 + 4 <= rows) {
        *reinterpret_cast<uint32_t *>(dst + r0) = packed;
    } else {
        for (int j = 0; r0 + j < rows; ++j) dst[r0 + j] = char(packed >> (8 * j));
    }
}

// ---- packed narrow leaves -----------------------------------------------------------------------------------------
// A random row of a 1-4 byte leaf costs a whole memory sector, and the ppo buffer has nine such leaves (logp, value,
// reward, next_value, advantage, return, three flags): ten sectors fetched for 27 useful bytes per sampled slot.
// cusrl_pack_rows interleaves them ONCE per update into one record per slot (27 -> 32 B, one sector), and the gather
// reads the record and fans the fields out to their separate, contiguous batch tensors (coalesced stores).
// Descriptor table, by value in the kernarg segment, laid out for FEW WIDE independent scalar loads (kernarg misses
// cost microseconds — see PushTable).  Entries are sorted by width on the host — [0, n4) move 4 bytes, [n4, n4 + n2)
// 2 bytes, the rest 1 byte; an 8-byte leaf is two 4-byte entries with a row stride of 8 — so the kernels are three
// straight-line, compile-time-indexed passes without any per-entry switch: every descriptor read is a load at a
// static kernarg offset and no register image of the record has to be indexed at run time (that would live in scratch).
constexpr int kMaxWideInRecord = 8;

struct WideInRecord {  // a wide leaf (a multiple of 16 bytes per slot) living in the record: chunks [first, first + count)
    char *ptr;         // gather: destination tensor
    int32_t pitch;     // its row size in bytes
    uint16_t first_chunk, num_chunks;
};

struct RecordTable {
    int32_t n4, n2, n1;
    int32_t record_bytes;               // a multiple of 16, at most CUSRL_MAX_RECORD_BYTES
    int32_t n_wide, used_chunks;        // gather: wide leaves in the record; 16-byte chunks that hold anything
    uint16_t offset[CUSRL_MAX_PACKED];  // byte offset inside the record
    uint8_t stride[CUSRL_MAX_PACKED];   // bytes between consecutive rows of the leaf (its row size)
    char *ptr[CUSRL_MAX_PACKED];        // pack: source leaf (+ byte offset);  gather: destination tensor (+ byte offset)
    WideInRecord wide[kMaxWideInRecord];
};

constexpr int kRecordTableDwords = sizeof(RecordTable) / 4;
static_assert(sizeof(RecordTable) % 4 == 0 && kRecordTableDwords <= 2 * kWave, "two dwords of the table per lane");

// The table as seen by one wave: lane k holds dword k of the kernarg copy, fetched by ONE vector load (a single
// round trip to kernarg memory; scalar loads sunk next to each use would be a chain of microsecond misses), and every
// descriptor is then a v_readlane with a compile-time lane index, i.e. a scalar value again.
struct WaveRecordTable {
    uint32_t lo, hi;  // lane k holds dwords k and k + 64 of the table
    __device__ __forceinline__ explicit WaveRecordTable(size_t kernarg_offset) {
        const uint32_t *karg = (const uint32_t *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + kernarg_offset);
        const int lane = threadIdx.x & (kWave - 1);
        lo = karg[lane];
        hi = karg[lane + kWave < kRecordTableDwords ? lane + kWave : 0];
    }
    __device__ __forceinline__ uint32_t dword(int k) const {
        return k < kWave ? __builtin_amdgcn_readlane(lo, k) : __builtin_amdgcn_readlane(hi, k - kWave);
    }
    __device__ __forceinline__ int n4() const { return int(dword(0)); }
    __device__ __forceinline__ int n2() const { return int(dword(1)); }
    __device__ __forceinline__ int n1() const { return int(dword(2)); }
    __device__ __forceinline__ int record_bytes() const { return int(dword(3)); }
    __device__ __forceinline__ int n_wide() const { return int(dword(4)); }
    __device__ __forceinline__ int used_chunks() const { return int(dword(5)); }
    __device__ __forceinline__ int offset(int f) const { return (dword(6 + f / 2) >> (16 * (f % 2))) & 0xffff; }
    __device__ __forceinline__ int stride(int f) const {
        return (dword(6 + CUSRL_MAX_PACKED / 2 + f / 4) >> (8 * (f % 4))) & 0xff;
    }
    __device__ __forceinline__ char *pointer_at(int base) const {
        return reinterpret_cast<char *>(uint64_t(dword(base)) | (uint64_t(dword(base + 1)) << 32));
    }
    __device__ __forceinline__ char *ptr(int f) const { return pointer_at(6 + CUSRL_MAX_PACKED / 2 + CUSRL_MAX_PACKED / 4 + 2 * f); }
    static constexpr int kWideBase = 6 + CUSRL_MAX_PACKED / 2 + CUSRL_MAX_PACKED / 4 + 2 * CUSRL_MAX_PACKED;
    __device__ __forceinline__ char *wide_ptr(int k) const { return pointer_at(kWideBase + 4 * k); }
    __device__ __forceinline__ int wide_pitch(int k) const { return int(dword(kWideBase + 4 * k + 2)); }
    __device__ __forceinline__ int wide_first(int k) const { return int(dword(kWideBase + 4 * k + 3) & 0xffff); }
    __device__ __forceinline__ int wide_count(int k) const { return int(dword(kWideBase + 4 * k + 3) >> 16); }
};
static_assert(offsetof(RecordTable, offset) == 24 && offsetof(RecordTable, stride) == 24 + 2 * CUSRL_MAX_PACKED &&
                  offsetof(RecordTable, ptr) == 24 + 3 * CUSRL_MAX_PACKED &&
                  offsetof(RecordTable, wide) == 24 + 3 * CUSRL_MAX_PACKED + 8 * CUSRL_MAX_PACKED && sizeof(WideInRecord) == 16,
              "WaveRecordTable decodes this layout");

// Record-major gather: a lane-op is one 16-byte chunk of one sampled slot's record, consecutive lanes take consecutive
// chunks of the same record.  Every memory line of a record is therefore requested exactly once, by one CU — when the
// wide leaves of a record were moved by their own blocks (possibly on other XCDs, each with its own L2), the line
// shared by the observation tail, the action and the narrow fields was measured fetched three times.  A chunk is
// either part of a wide leaf (stored as a whole to that leaf's batch tensor) or holds narrow entries (fanned out).
//
// What a chunk is comes from a map the HOST resolves (chunk -> destination / entry mask) and passes by value.  Each wave
// fetches it with ONE vector load — lane c holds chunk c's 16-byte descriptor, lane f entry f's — issued together with
// the data loads, and a lane-op then looks its chunk up with four wave shuffles: no LDS, no barrier.  (Two earlier
// forms, kept here as a warning: resolving per lane-op from scalar registers with static loops over all descriptors made
// the kernel instruction-bound, 21 us for a 24 576-slot minibatch; staging the map in LDS behind two barriers, 8.7 us.)
constexpr int kRecordOpsPerBlock = kBlock * kGatherItems;
constexpr int kMaxRecordChunks = CUSRL_MAX_RECORD_BYTES / 16;
static_assert(kMaxRecordChunks == kWave, "one chunk descriptor per lane");

struct ChunkInfo {   // 16 bytes
    char *ptr;       // wide chunk: destination tensor of its leaf; narrow chunk: nullptr
    int32_t pitch;   // wide: row size of the leaf
    int32_t detail;  // wide: first chunk of the leaf inside the record; narrow: bit f set <=> entry f lives in this chunk
};

struct EntryInfo {  // 16 bytes: one narrow entry
    char *ptr;
    int32_t offset;        // byte offset inside the record
    int32_t stride_width;  // stride | width << 8
};

struct RecordMap {
    int32_t chunks, record_bytes, n_entries, pad;
    ChunkInfo chunk[kMaxRecordChunks];
    EntryInfo entry[CUSRL_MAX_PACKED];
};

__device__ __forceinline__ void gather_record_major(size_t map_kernarg_offset, int chunks, int64_t pitch, int n_entries,
                                                    const char *__restrict__ src, const int64_t *__restrict__ idx,
                                                    int64_t rows, int64_t op0, int64_t B, int64_t N, bool temporal) {
    const char *karg = (const char *)__builtin_amdgcn_kernarg_segment_ptr() + map_kernarg_offset;
    const int lane = threadIdx.x & (kWave - 1);
    const uint4 my_chunk = reinterpret_cast<const uint4 *>(karg + offsetof(RecordMap, chunk))[lane];
    const uint4 my_entry = reinterpret_cast<const uint4 *>(karg + offsetof(RecordMap, entry))[lane & (CUSRL_MAX_PACKED - 1)];
    const int64_t ops = rows * chunks;
    int64_t out_row[kGatherItems], src_row[kGatherItems];
    int chunk[kGatherItems];
    // a power-of-two chunk count (16 for the 256-byte `ppo` record, 2 for the 32-byte narrow record) divides the block
    // size: shifts instead of 64-bit divisions, and a lane keeps the SAME chunk index for all its items
    const bool pow2 = (chunks & (chunks - 1)) == 0;
    const int shift = 31 - __clz(chunks);
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it) {
        // clamped, unpredicated (see gather_unit); in the power-of-two form an out-of-range op is folded onto the LAST
        // record at the lane's own chunk index, so the lane's chunk stays the same for all items (duplicates rewrite
        // identical bytes)
        int64_t op = op0 + int64_t(it) * kBlock;
        if (pow2) op = op < ops ? op : ops - chunks + (op & (chunks - 1));
        else op = min(op, ops - 1);
        out_row[it] = pow2 ? (op >> shift) : op / chunks;
        chunk[it] = int(op - out_row[it] * chunks);
    }
    if (temporal) {
#pragma unroll
        for (int it = 0; it < kGatherItems; ++it) {
            const int64_t t = out_row[it] / B, b = out_row[it] - t * B;
            src_row[it] = t * N + idx[b];
        }
    } else {
#pragma unroll
        for (int it = 0; it < kGatherItems; ++it) src_row[it] = idx[out_row[it]];
    }
    uint4 regs[kGatherItems];
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it)
        regs[it] = *reinterpret_cast<const uint4 *>(src + src_row[it] * pitch + int64_t(chunk[it]) * 16);
    // chunk descriptors by wave shuffle (all lanes active here: a bpermute reads nothing from a disabled lane)
    uint4 info[kGatherItems];
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it) {
        if (it > 0 && pow2) {  // same chunk as item 0 (the clamp at the very end only ever repeats the last op)
            info[it] = info[0];
            continue;
        }
        info[it].x = __shfl(my_chunk.x, chunk[it], kWave);
        info[it].y = __shfl(my_chunk.y, chunk[it], kWave);
        info[it].z = __shfl(my_chunk.z, chunk[it], kWave);
        info[it].w = __shfl(my_chunk.w, chunk[it], kWave);
    }
    pin_loaded(regs);
    // wide chunks: one 16-byte store per item
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it) {
        char *wide_dst = reinterpret_cast<char *>(uint64_t(info[it].x) | (uint64_t(info[it].y) << 32));
        if (wide_dst)
            *reinterpret_cast<uint4 *>(wide_dst + out_row[it] * int(info[it].z) + int64_t(chunk[it] - int(info[it].w)) * 16) = regs[it];
    }
    // narrow entries: a wave-uniform walk over the entries (descriptor of entry f from lane f by v_readlane); every
    // entry is tested against the four items of the lane (by name: indexing `regs` with a loop variable here once moved
    // the array to LDS)
    auto fan_out = [&](const uint4 value, const uint4 desc, const int64_t row, int f, char *base, int offset, int stride_width) {
        const bool narrow = (desc.x | desc.y) == 0u;
        if (narrow && ((desc.w >> f) & 1u)) {
            const int sel = (offset >> 2) & 3;
            uint32_t word = value.x;
            word = sel == 1 ? value.y : word;
            word = sel == 2 ? value.z : word;
            word = sel == 3 ? value.w : word;
            word >>= (offset & 3) * 8;
            char *out = base + row * (stride_width & 0xff);
            const int width = stride_width >> 8;
            if (width == 4) *reinterpret_cast<uint32_t *>(out) = word;
            else if (width == 2) *reinterpret_cast<uint16_t *>(out) = uint16_t(word);
            else *reinterpret_cast<uint8_t *>(out) = uint8_t(word);
        }
    };
    static_assert(kGatherItems == 4, "the four items are visited by name");
    for (int f = 0; f < n_entries; ++f) {
        // (v_readlane returns a signed int: go through uint32_t, or a pointer whose low word has bit 31 set sign-extends)
        char *base = reinterpret_cast<char *>(uint64_t(uint32_t(__builtin_amdgcn_readlane(my_entry.x, f))) |
                                              (uint64_t(uint32_t(__builtin_amdgcn_readlane(my_entry.y, f))) << 32));
        const int offset = int(__builtin_amdgcn_readlane(my_entry.z, f));
        const int stride_width = int(__builtin_amdgcn_readlane(my_entry.w, f));
        fan_out(regs[0], info[0], out_row[0], f, base, offset, stride_width);
        fan_out(regs[1], info[1], out_row[1], f, base, offset, stride_width);
        fan_out(regs[2], info[2], out_row[2], f, base, offset, stride_width);
        fan_out(regs[3], info[3], out_row[3], f, base, offset, stride_width);
    }
}

// kOwnedChunks > 0: the caller owns `kOwnedChunks` whole 16-byte chunks of every record starting at chunk `owned_first` —
// they hold the listed narrow fields and nothing else — so the fields are assembled in registers and leave as ONE
// 16-byte store per chunk.  Separate 1-4 byte stores at a 256-byte pitch are one memory request each per row and field
// (four requests and four partial-line writes per slot of the `ppo` record's 13 narrow bytes: 1.4 ms at 25 M slots).
// The descriptors are wave-uniform scalars, so placing a field into its dword of the image is a scalar branch.
template <int kOwnedChunks>
__global__ __launch_bounds__(kBlock) void pack_rows_kernel(const RecordTable rec_arg, char *__restrict__ record,
                                                           int64_t rows, int owned_first) {
    const WaveRecordTable rec(0);  // rec_arg is the first kernel argument: kernarg offset 0
    const int64_t row = min(int64_t(blockIdx.x) * kBlock + threadIdx.x, rows - 1);  // clamped: duplicates rewrite equal bytes
    char *out = record + row * rec.record_bytes();
    const int n4 = rec.n4(), n42 = n4 + rec.n2(), n = n42 + rec.n1();
    uint32_t word[CUSRL_MAX_PACKED];
    // unit-stride across the wave for 1-channel leaves: coalesced reads, all issued before the stores
#pragma unroll
    for (int f = 0; f < CUSRL_MAX_PACKED; ++f) {
        word[f] = 0;
        if (f < n) {
            const char *src = rec.ptr(f) + row * rec.stride(f);
            if (f < n4) word[f] = *reinterpret_cast<const uint32_t *>(src);
            else if (f < n42) word[f] = *reinterpret_cast<const uint16_t *>(src);
            else word[f] = *reinterpret_cast<const uint8_t *>(src);
        }
    }
    if constexpr (kOwnedChunks > 0) {
        uint32_t image[4 * kOwnedChunks];
#pragma unroll
        for (int d = 0; d < 4 * kOwnedChunks; ++d) image[d] = 0;
#pragma unroll
        for (int f = 0; f < CUSRL_MAX_PACKED; ++f) {
            if (f < n) {
                const int at = rec.offset(f) - owned_first * 16;  // wave-uniform
                const uint32_t bits = word[f] << (8 * (at & 3));   // (zero-extended above; a 4-byte entry has at % 4 == 0)
#pragma unroll
                for (int d = 0; d < 4 * kOwnedChunks; ++d)
                    if ((at >> 2) == d) image[d] |= bits;
            }
        }
#pragma unroll
        for (int c = 0; c < kOwnedChunks; ++c)
            *reinterpret_cast<uint4 *>(out + (owned_first + c) * 16) =
                uint4{image[4 * c], image[4 * c + 1], image[4 * c + 2], image[4 * c + 3]};
        return;
    }
    // narrow strided stores, merged in L2 (this runs once per update)
#pragma unroll
    for (int f = 0; f < CUSRL_MAX_PACKED; ++f) {
        if (f < n) {
            char *dst = out + rec.offset(f);
            if (f < n4) *reinterpret_cast<uint32_t *>(dst) = word[f];
            else if (f < n42) *reinterpret_cast<uint16_t *>(dst) = uint16_t(word[f]);
            else *reinterpret_cast<uint8_t *>(dst) = uint8_t(word[f]);
        }
    }
}

struct GatherArgs {  // both tables in ONE kernel argument, so that the record map's kernarg offset is offsetof()
    GatherTable tab;
    RecordMap map;
};

__global__ __launch_bounds__(kBlock) void gather_kernel(const GatherArgs args, const char *__restrict__ record,
                                                        const int64_t *__restrict__ idx, int64_t B, int64_t T,
                                                        int64_t N, int temporal) {
    const GatherTable &tab = args.tab;
    const int blk = blockIdx.x;
    if (blk >= tab.block_start[CUSRL_MAX_FIELDS]) {  // the blocks behind the last plain leaf unpack the record
        const int64_t rows = temporal ? T * B : B;
        const int64_t op0 = int64_t(blk - tab.block_start[CUSRL_MAX_FIELDS]) * kRecordOpsPerBlock + threadIdx.x;
        gather_record_major(offsetof(GatherArgs, map), args.map.chunks, args.map.record_bytes, args.map.n_entries, record, idx,
                            rows, op0, B, N, temporal != 0);
        return;
    }
    const int f = find_leaf(tab, blk);
    const GatherLeaf leaf = tab.leaf[f];
    const char *__restrict__ src = leaf.src;
    char *__restrict__ dst = leaf.dst;
    const int unit = leaf.unit;
    const int lpr = leaf.lanes_per_row;
    const int64_t src_pitch = leaf.src_pitch, dst_pitch = leaf.dst_pitch;
    const int64_t rows = temporal ? T * B : B;
    const bool temp = temporal != 0;
    if (unit == 0) {  // one packed lane-op per lane: kBlock ops per block
        gather_bytes_packed(src, dst, idx, rows, int64_t(blk - tab.block_start[f]) * kBlock + threadIdx.x, B, N, temp);
        return;
    }
    const int64_t op0 = int64_t(blk - tab.block_start[f]) * kGatherOpsPerBlock + threadIdx.x;
    const int64_t ops = rows * lpr;
    switch (unit) {
        case 16: gather_unit<uint4>(src, dst, idx, ops, op0, lpr, src_pitch, dst_pitch, B, N, temp); break;
        case 8: gather_unit<uint2>(src, dst, idx, ops, op0, lpr, src_pitch, dst_pitch, B, N, temp); break;
        case 4: gather_unit<uint32_t>(src, dst, idx, ops, op0, lpr, src_pitch, dst_pitch, B, N, temp); break;
        case 2: gather_unit<uint16_t>(src, dst, idx, ops, op0, lpr, src_pitch, dst_pitch, B, N, temp); break;
        default: gather_unit<uint8_t>(src, dst, idx, ops, op0, lpr, src_pitch, dst_pitch, B, N, temp); break;
    }
}

// --------------------------------------------------------------------------------------------- compaction
constexpr int kFlagChunk = kBlock * 16;  // flags per block: one 16 B load per lane

__device__ __forceinline__ int load_flags16(const uint8_t *__restrict__ flags, int64_t base, int64_t n,
                                            uint32_t (&w)[4]) {
    // returns the number of set flags among the 16 bytes starting at `base` (bounds-checked)
    int count = 0;
    if (base + 16 <= n && (reinterpret_cast<uintptr_t>(flags + base) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(flags + base);
        w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
        for (int j = 0; j < 16; ++j)
            if (base + j < n && flags[base + j]) w[j >> 2] |= 1u << (8 * (j & 3));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // bools are 0/1 bytes in torch, but treat any non-zero byte as set
        uint32_t x = w[k];
        x |= x >> 4;
        x |= x >> 2;
        x |= x >> 1;
        x &= 0x01010101u;
        w[k] = x;
        count += __popc(x);
    }
    return count;
}

__global__ __launch_bounds__(kBlock) void count_flags_kernel(const uint8_t *__restrict__ flags, int64_t n,
                                                             int32_t *__restrict__ block_counts) {
    __shared__ int scratch[kWavesPerBlock];
    uint32_t w[4];
    const int64_t base = int64_t(blockIdx.x) * kFlagChunk + int64_t(threadIdx.x) * 16;
    const int c = base < n ? load_flags16(flags, base, n, w) : 0;
    const int total = block_sum(c, scratch);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void compact_flags_kernel(const uint8_t *__restrict__ flags, int64_t n,
                                                               const int32_t *__restrict__ block_counts,
                                                               int64_t *__restrict__ indices_out,
                                                               int32_t *__restrict__ count_out) {
    __shared__ int scratch[kWavesPerBlock];
    // offset of this block = sum of the counts of all earlier blocks (fixed order -> deterministic output)
    int before = 0;
    for (int i = threadIdx.x; i < int(blockIdx.x); i += kBlock) before += block_counts[i];
    __shared__ int block_offset;
    const int prefix = block_sum(before, scratch);
    if (threadIdx.x == 0) block_offset = prefix;
    __syncthreads();

    uint32_t w[4];
    const int64_t base = int64_t(blockIdx.x) * kFlagChunk + int64_t(threadIdx.x) * 16;
    const int c = base < n ? load_flags16(flags, base, n, w) : 0;
    int total;
    int pos = block_offset + block_exclusive_scan(c, scratch, total);
    if (c) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if ((w[j >> 2] >> (8 * (j & 3))) & 1u) indices_out[pos++] = base + j;
    }
    // system-scope store: `count_out` may be pinned HOST memory that the caller polls instead of paying a device->host
    // copy + stream synchronisation; consumers of indices_out are stream-ordered behind this kernel either way
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        __hip_atomic_store(count_out, block_offset + total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// --------------------------------------------------------------------------------------------- scatter
template <typename V>
__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(const char *__restrict__ src,
                                                              const int64_t *__restrict__ indices,
                                                              char *__restrict__ dst, int64_t K, int lpr,
                                                              int64_t row_bytes, const int32_t *__restrict__ count_dev) {
    const int64_t limit = count_dev ? min(K, int64_t(*count_dev)) : K;
    const int64_t ops = limit * lpr;
    if (ops <= 0) return;
    // same shape as the gather: clamped, unpredicated, index loads -> row loads -> stores (duplicates are identical)
    const int64_t op0 = int64_t(blockIdx.x) * kGatherOpsPerBlock + threadIdx.x;
    int64_t row[kGatherItems], col[kGatherItems], dst_row[kGatherItems];
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it) {
        const int64_t op = min(op0 + int64_t(it) * kBlock, ops - 1);
        row[it] = lpr == 1 ? op : op / lpr;
        col[it] = op - row[it] * lpr;
    }
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it) dst_row[it] = indices[row[it]];
    V regs[kGatherItems];
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it)
        regs[it] = *reinterpret_cast<const V *>(src + row[it] * row_bytes + col[it] * int64_t(sizeof(V)));
    pin_loaded(regs);
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it)
        *reinterpret_cast<V *>(dst + dst_row[it] * row_bytes + col[it] * int64_t(sizeof(V))) = regs[it];
}

// The act input of the next env step in ONE pass (cusrl/template/environment.py:365-379 `update_observation_and_state`
// + the copy into the act step's static input): dst[n] = src[n] for every env that did not finish, and
// dst[indices[k]] = init[k] for k < *count — the finished envs, in the order the step epilogue listed them.  `done` and
// (indices, count) describe the same set (cusrl_step_epilogue writes both), so every destination row has one writer.
template <typename V>
__global__ __launch_bounds__(kBlock) void splice_rows_kernel(const char *__restrict__ src, const char *__restrict__ init,
                                                             const int64_t *__restrict__ indices,
                                                             const int32_t *__restrict__ count_dev,
                                                             const uint8_t *__restrict__ done, char *__restrict__ dst,
                                                             int64_t N, int lpr, int64_t row_bytes) {
    const int64_t ops = N * lpr;
    const int64_t resets = min(N, int64_t(*count_dev)) * lpr;
    const int64_t op0 = int64_t(blockIdx.x) * kGatherOpsPerBlock + threadIdx.x;
    int64_t row[kGatherItems], col[kGatherItems];
    V kept[kGatherItems], fresh[kGatherItems];
    int64_t target[kGatherItems];
    bool finished[kGatherItems];
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it) {
        const int64_t op = min(op0 + int64_t(it) * kBlock, ops - 1);  // clamped, unpredicated loads (see gather_unit)
        row[it] = lpr == 1 ? op : op / lpr;
        col[it] = op - row[it] * lpr;
    }
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it) {
        finished[it] = done[row[it]] != 0;
        target[it] = indices[row[it]];  // only meaningful below `resets`; the index buffer always holds valid env ids
        kept[it] = *reinterpret_cast<const V *>(src + row[it] * row_bytes + col[it] * int64_t(sizeof(V)));
        fresh[it] = *reinterpret_cast<const V *>(init + row[it] * row_bytes + col[it] * int64_t(sizeof(V)));
    }
#pragma unroll
    for (int it = 0; it < kGatherItems; ++it) {
        const int64_t op = op0 + int64_t(it) * kBlock;
        if (op < ops && !finished[it])
            *reinterpret_cast<V *>(dst + row[it] * row_bytes + col[it] * int64_t(sizeof(V))) = kept[it];
        if (op < resets)
            *reinterpret_cast<V *>(dst + target[it] * row_bytes + col[it] * int64_t(sizeof(V))) = fresh[it];
    }
}

// Flat slot list of random temporal windows (cusrl/sampler/random_sampler.py:95-109): window b covers `L` consecutive
// steps of env[b] from logical step start[b]; logical time 0 is physical row `cursor` once the ring is full.
// out[t * B + b] = ((cursor + start[b] + t) % T) * N + env[b].
__global__ __launch_bounds__(kBlock) void window_indices_kernel(const int64_t *__restrict__ start,
                                                                const int64_t *__restrict__ env,
                                                                int64_t *__restrict__ out, int64_t B, int64_t L,
                                                                int64_t T, int64_t N, int64_t cursor) {
    const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= B * L) return;
    const int64_t t = i / B, b = i - t * B;
    out[i] = ((cursor + start[b] + t) % T) * N + env[b];
}

static int pick_unit(const void *a, const void *b, int64_t row_bytes) {
    for (int unit : {16, 8, 4, 2})
        if (row_bytes % unit == 0 && aligned(a, unit) && aligned(b, unit)) return unit;
    return 1;
}

}  // namespace cusrl

using namespace cusrl;

static int push_fields(const cusrl_field_t *fields, int n_fields, int64_t cursor, int64_t N, char *record,
                       int64_t record_bytes, const int32_t *record_offset, void *stream) {
    if (n_fields == 0) return 0;
    PushTable tab;
    int32_t blocks = 0;
    int64_t step_bytes = 0;
    if (int rc = build_push_table(fields, n_fields, cursor, N, record, record_bytes, record_offset, -1, tab, blocks, step_bytes))
        return rc;
    if (tab.n == 0) return 0;
    if (push_streams(step_bytes))
        hipLaunchKernelGGL(push_kernel<3>, dim3(blocks), dim3(kBlock), 0, as_stream(stream), tab);
    else
        hipLaunchKernelGGL(push_kernel<0>, dim3(blocks), dim3(kBlock), 0, as_stream(stream), tab);
    return launch_status();
}

extern "C" int cusrl_buffer_push(const cusrl_field_t *fields, int n_fields, int64_t cursor, int64_t N, void *stream) {
    return push_fields(fields, n_fields, cursor, N, nullptr, 0, nullptr, stream);
}

extern "C" int cusrl_buffer_push_through(const cusrl_field_t *fields, int n_fields, int64_t cursor, int64_t N,
                                         void *record, int64_t record_bytes, const int32_t *record_offset,
                                         void *stream) {
    if (!record_offset) return CUSRL_E_INVALID;
    return push_fields(fields, n_fields, cursor, N, static_cast<char *>(record), record_bytes, record_offset, stream);
}

// Splits the packed-field list into (a) narrow entries (1 / 2 / 4 / 8 bytes) for the record kernels' RecordTable and
// (b) wide fields (a multiple of 16 bytes at a 16-byte-aligned offset: observation, action rows), which are moved as
// ordinary 16-byte-lane leaves whose source (gather) or destination (pack) pitch is the record size.
struct WideField {
    char *ptr;
    int32_t offset, width;
};

static int fill_record_table(const cusrl_packed_field_t *packed, int n_packed, int64_t record_bytes,
                             RecordTable &rec, WideField *wide, int &n_wide) {
    rec.n4 = rec.n2 = rec.n1 = 0;
    rec.record_bytes = int32_t(record_bytes);
    rec.n_wide = rec.used_chunks = 0;
    n_wide = 0;
    for (int i = 0; i < CUSRL_MAX_PACKED; ++i) rec.offset[i] = 0, rec.stride[i] = 4, rec.ptr[i] = nullptr;
    for (int i = 0; i < kMaxWideInRecord; ++i) rec.wide[i] = WideInRecord{nullptr, 16, 0, 0};
    if (n_packed < 0 || n_packed > CUSRL_MAX_PACKED + CUSRL_MAX_FIELDS) return CUSRL_E_TOO_MANY;
    if (n_packed == 0) return 0;
    if (!packed || record_bytes < 16 || record_bytes % 16 != 0 || record_bytes > CUSRL_MAX_RECORD_BYTES)
        return CUSRL_E_INVALID;
    uint8_t used[CUSRL_MAX_RECORD_BYTES] = {0};  // fields must not overlap
    int entries = 0;
    for (int i = 0; i < n_packed; ++i) {
        const int32_t w = packed[i].width, off = packed[i].offset;
        const bool narrow = w == 1 || w == 2 || w == 4 || w == 8;
        const bool is_wide = w >= 16 && w % 16 == 0;
        if (!packed[i].ptr || (!narrow && !is_wide) || off < 0 || off + w > record_bytes) return CUSRL_E_INVALID;
        if (narrow && (off % w != 0 || !aligned(packed[i].ptr, uintptr_t(w)))) return CUSRL_E_INVALID;
        if (is_wide && (off % 16 != 0 || !aligned(packed[i].ptr, 16))) return CUSRL_E_INVALID;
        for (int b = off; b < off + w; ++b) {
            if (used[b]) return CUSRL_E_INVALID;
            used[b] = 1;
        }
        if (is_wide) {
            if (n_wide == CUSRL_MAX_FIELDS) return CUSRL_E_TOO_MANY;
            wide[n_wide++] = WideField{static_cast<char *>(packed[i].ptr), off, w};
        } else {
            entries += w == 8 ? 2 : 1;
        }
    }
    if (entries > CUSRL_MAX_PACKED) return CUSRL_E_TOO_MANY;
    int last = 0;
    for (int b = 0; b < record_bytes; ++b)
        if (used[b]) last = b;
    rec.used_chunks = last / 16 + 1;  // trailing padding chunks are never read
    int at = 0;
    for (int pass_width : {4, 2, 1}) {  // 4-byte entries first (an 8-byte leaf = two of them), then 2-byte, then 1-byte
        for (int i = 0; i < n_packed; ++i) {
            const int32_t w = packed[i].width, off = packed[i].offset;
            if (w >= 16 || (w == 8 ? 4 : w) != pass_width) continue;
            for (int half = 0; half < (w == 8 ? 2 : 1); ++half) {
                rec.offset[at] = uint16_t(off + 4 * half);
                rec.stride[at] = uint8_t(w);
                rec.ptr[at] = static_cast<char *>(packed[i].ptr) + 4 * half;
                ++at;
            }
        }
        (pass_width == 4 ? rec.n4 : pass_width == 2 ? rec.n2 : rec.n1) = at - (pass_width == 4 ? 0 : pass_width == 2 ? rec.n4 : rec.n4 + rec.n2);
    }
    return 0;
}

static void set_block_tail(GatherTable &tab, int n, int64_t blocks) {
    for (int i = n; i <= CUSRL_MAX_FIELDS; ++i) tab.block_start[i] = int32_t(blocks);
    tab.n = n;
}

static int pack_rows(const cusrl_packed_field_t *fields, int n_fields, void *record, int64_t record_bytes, int64_t rows,
                     int owned_first, int owned_chunks, void *stream) {
    if (n_fields == 0 || rows == 0) return 0;
    if (!record || rows < 0 || !aligned(record, 16)) return CUSRL_E_INVALID;
    if (owned_chunks < 0 || owned_chunks > 2 || (owned_chunks > 0 && owned_first < 0)) return CUSRL_E_INVALID;
    GatherArgs args;
    RecordTable rec;
    WideField wide[CUSRL_MAX_FIELDS];
    int n_wide = 0;
    if (int rc = fill_record_table(fields, n_fields, record_bytes, rec, wide, n_wide)) return rc;
    if (owned_chunks > 0) {  // every narrow entry inside the owned chunks, no wide leaf over them
        if (int64_t(owned_first + owned_chunks) * 16 > record_bytes) return CUSRL_E_INVALID;
        for (int f = 0; f < rec.n4 + rec.n2 + rec.n1; ++f) {
            const int width = f < rec.n4 ? 4 : f < rec.n4 + rec.n2 ? 2 : 1;
            if (rec.offset[f] < owned_first * 16 || rec.offset[f] + width > (owned_first + owned_chunks) * 16)
                return CUSRL_E_INVALID;
        }
        for (int i = 0; i < n_wide; ++i)
            if (wide[i].offset < (owned_first + owned_chunks) * 16 && wide[i].offset + wide[i].width > owned_first * 16)
                return CUSRL_E_INVALID;
    }
    if (n_wide > 0) {  // wide leaves: a strided row copy leaf -> record (the gather kernel without an index vector)
        GatherTable &tab = args.tab;
        int64_t blocks = 0;
        for (int i = 0; i < n_wide; ++i) {
            GatherLeaf &leaf = tab.leaf[i];
            leaf.src = wide[i].ptr;
            leaf.dst = static_cast<char *>(record) + wide[i].offset;
            leaf.src_pitch = wide[i].width;
            leaf.dst_pitch = int32_t(record_bytes);
            leaf.unit = 16;
            leaf.lanes_per_row = wide[i].width / 16;
            tab.block_start[i] = int32_t(blocks);
            blocks += ceil_div(rows * leaf.lanes_per_row, kGatherOpsPerBlock);
            if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
        }
        set_block_tail(tab, n_wide, blocks);
        args.map.chunks = args.map.n_entries = 0;
        args.map.record_bytes = int32_t(record_bytes);
        hipLaunchKernelGGL(gather_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), args,
                           static_cast<const char *>(nullptr), static_cast<const int64_t *>(nullptr), rows, int64_t(1),
                           rows, 0);
        if (int rc = launch_status()) return rc;
    }
    if (rec.n4 + rec.n2 + rec.n1 > 0) {
        const int64_t blocks = ceil_div(rows, kBlock);
        if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
        auto kernel = owned_chunks == 2 ? pack_rows_kernel<2> : owned_chunks == 1 ? pack_rows_kernel<1> : pack_rows_kernel<0>;
        hipLaunchKernelGGL(kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), rec,
                           static_cast<char *>(record), rows, owned_first);
    }
    return launch_status();
}

extern "C" int cusrl_pack_rows(const cusrl_packed_field_t *fields, int n_fields, void *record, int64_t record_bytes,
                               int64_t rows, void *stream) {
    return pack_rows(fields, n_fields, record, record_bytes, rows, 0, 0, stream);
}

extern "C" int cusrl_pack_rows_owned(const cusrl_packed_field_t *fields, int n_fields, void *record, int64_t record_bytes,
                                     int64_t rows, int32_t owned_first_chunk, int32_t owned_chunks, void *stream) {
    if (owned_chunks < 1) return CUSRL_E_INVALID;
    return pack_rows(fields, n_fields, record, record_bytes, rows, owned_first_chunk, owned_chunks, stream);
}

extern "C" int cusrl_gather_rows_packed(const cusrl_field_t *fields, int n_fields, const void *record,
                                        int64_t record_bytes, const cusrl_packed_field_t *packed, int n_packed,
                                        const int64_t *indices, int64_t B, int64_t T, int64_t N, int temporal,
                                        void *stream) {
    if ((n_fields == 0 && n_packed == 0) || B == 0) return 0;
    if ((n_fields > 0 && !fields) || !indices || n_fields < 0 || B < 0 || T < 1 || N < 1) return CUSRL_E_INVALID;
    if (n_fields > CUSRL_MAX_FIELDS) return CUSRL_E_TOO_MANY;
    if (n_packed > 0 && (!record || !aligned(record, 16))) return CUSRL_E_INVALID;
    const int64_t rows = temporal ? T * B : B;
    GatherArgs args;
    GatherTable &tab = args.tab;
    RecordTable rec;
    WideField wide[CUSRL_MAX_FIELDS];
    int n_wide = 0;
    if (int rc = fill_record_table(packed, n_packed, n_packed > 0 ? record_bytes : 16, rec, wide, n_wide)) return rc;
    int64_t blocks = 0;
    int n = 0;
    for (int i = 0; i < n_fields; ++i) {
        const int64_t rb = fields[i].row_bytes;
        if (rb < 0 || rb > INT32_MAX || (rb > 0 && (!fields[i].src || !fields[i].dst))) return CUSRL_E_INVALID;
        if (rb == 0) continue;
        GatherLeaf &leaf = tab.leaf[n];
        leaf.src = static_cast<const char *>(fields[i].src);
        leaf.dst = static_cast<char *>(fields[i].dst);
        leaf.src_pitch = leaf.dst_pitch = int32_t(rb);
        int64_t ops;
        if (rb == 1 && aligned(leaf.dst, 4)) {
            leaf.unit = 0;
            leaf.lanes_per_row = 1;
            ops = (rows + 3) / 4;
        } else {
            const int unit = pick_unit(leaf.src, leaf.dst, rb);
            leaf.unit = unit;
            leaf.lanes_per_row = int32_t(rb / unit);
            ops = rows * (rb / unit);
        }
        tab.block_start[n] = int32_t(blocks);
        blocks += ceil_div(ops, leaf.unit == 0 ? kBlock : kGatherOpsPerBlock);
        if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
        ++n;
    }
    // the chunk map of the record-major pass: chunk -> wide leaf (destination, pitch, first chunk) or entry mask
    RecordMap &map = args.map;
    map.chunks = n_packed > 0 ? rec.used_chunks : 0;
    map.record_bytes = int32_t(record_bytes);
    map.n_entries = rec.n4 + rec.n2 + rec.n1;
    map.pad = 0;
    for (int c = 0; c < kMaxRecordChunks; ++c) map.chunk[c] = ChunkInfo{nullptr, 0, 0};
    for (int f = 0; f < CUSRL_MAX_PACKED; ++f) map.entry[f] = EntryInfo{nullptr, 0, 4 | (4 << 8)};
    for (int i = 0; i < n_wide; ++i)
        for (int c = wide[i].offset / 16; c < (wide[i].offset + wide[i].width) / 16; ++c)
            map.chunk[c] = ChunkInfo{wide[i].ptr, wide[i].width, wide[i].offset / 16};
    for (int f = 0; f < map.n_entries; ++f) {
        const int width = f < rec.n4 ? 4 : (f < rec.n4 + rec.n2 ? 2 : 1);
        map.entry[f] = EntryInfo{rec.ptr[f], rec.offset[f], int32_t(rec.stride[f]) | (width << 8)};
        map.chunk[rec.offset[f] / 16].detail |= 1 << f;
    }
    set_block_tail(tab, n, blocks);
    if (n_packed > 0) blocks += ceil_div(rows * map.chunks, kRecordOpsPerBlock);
    if (blocks == 0) return 0;
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(gather_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), args,
                       static_cast<const char *>(record), indices, B, T, N, temporal);
    return launch_status();
}

extern "C" int cusrl_gather_rows(const cusrl_field_t *fields, int n_fields, const int64_t *indices, int64_t B,
                                 int64_t T, int64_t N, int temporal, void *stream) {
    if (n_fields == 0 || B == 0) return 0;
    return cusrl_gather_rows_packed(fields, n_fields, nullptr, 0, nullptr, 0, indices, B, T, N, temporal, stream);
}

extern "C" int cusrl_window_indices(const int64_t *start, const int64_t *env, int64_t *out, int64_t B, int64_t L,
                                    int64_t T, int64_t N, int64_t cursor, void *stream) {
    if (B == 0 || L == 0) return 0;
    if (!start || !env || !out || B < 0 || L < 0 || T < 1 || N < 1 || cursor < 0 || cursor >= T) return CUSRL_E_INVALID;
    const int64_t blocks = ceil_div(B * L, kBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    hipLaunchKernelGGL(window_indices_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), start, env, out,
                       B, L, T, N, cursor);
    return launch_status();
}

__global__ void zero_count_kernel(int32_t *count) { __hip_atomic_store(count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

extern "C" int64_t cusrl_flag_blocks(int64_t n) { return n <= 0 ? 0 : ceil_div(n, kFlagChunk); }

extern "C" int cusrl_compact_flags(const uint8_t *flags, int64_t n, int32_t *block_counts, int recount,
                                   int64_t *indices_out, int32_t *count_out, void *stream) {
    if (n < 0 || !count_out) return CUSRL_E_INVALID;
    if (n == 0) {  // (a kernel, not hipMemsetAsync: no memset nodes in captured regions)
        hipLaunchKernelGGL(zero_count_kernel, dim3(1), dim3(1), 0, as_stream(stream), count_out);
        return launch_status();
    }
    if (!flags || !block_counts || !indices_out) return CUSRL_E_INVALID;
    const int64_t blocks = cusrl_flag_blocks(n);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    if (recount && blocks > 1) {  // a single block never reads the per-block counts (its offset is 0)
        hipLaunchKernelGGL(count_flags_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), flags, n,
                           block_counts);
        if (int rc = launch_status()) return rc;
    }
    hipLaunchKernelGGL(compact_flags_kernel, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream), flags, n,
                       block_counts, indices_out, count_out);
    return launch_status();
}

#define CUSRL_LAUNCH_SCATTER(V)                                                                                      \
    hipLaunchKernelGGL(scatter_rows_kernel<V>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream),           \
                       static_cast<const char *>(src), indices, static_cast<char *>(dst), K, lpr, row_bytes, count_dev)

#define CUSRL_LAUNCH_SPLICE(V)                                                                                       \
    hipLaunchKernelGGL(splice_rows_kernel<V>, dim3(uint32_t(blocks)), dim3(kBlock), 0, as_stream(stream),           \
                       static_cast<const char *>(src), static_cast<const char *>(init), indices, count_dev, done,    \
                       static_cast<char *>(dst), N, lpr, row_bytes)

extern "C" int cusrl_splice_rows(const void *src, const void *init, const int64_t *indices, const int32_t *count_dev,
                                 const uint8_t *done, void *dst, int64_t N, int64_t row_bytes, void *stream) {
    if (N == 0 || row_bytes == 0) return 0;
    if (!src || !init || !indices || !count_dev || !done || !dst || N < 0 || row_bytes < 0) return CUSRL_E_INVALID;
    int unit = 1;
    for (int u : {16, 8, 4, 2})
        if (row_bytes % u == 0 && aligned(src, u) && aligned(init, u) && aligned(dst, u)) {
            unit = u;
            break;
        }
    const int lpr = int(row_bytes / unit);
    const int64_t blocks = ceil_div(N * lpr, kGatherOpsPerBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    switch (unit) {
        case 16: CUSRL_LAUNCH_SPLICE(uint4); break;
        case 8: CUSRL_LAUNCH_SPLICE(uint2); break;
        case 4: CUSRL_LAUNCH_SPLICE(uint32_t); break;
        case 2: CUSRL_LAUNCH_SPLICE(uint16_t); break;
        default: CUSRL_LAUNCH_SPLICE(uint8_t); break;
    }
    return launch_status();
}

extern "C" int cusrl_scatter_rows(const void *src, const int64_t *indices, void *dst, int64_t K, int64_t row_bytes,
                                  const int32_t *count_dev, void *stream) {
    if (K == 0 || row_bytes == 0) return 0;
    if (!src || !indices || !dst || K < 0 || row_bytes < 0) return CUSRL_E_INVALID;
    const int unit = pick_unit(src, dst, row_bytes);
    const int lpr = int(row_bytes / unit);
    const int64_t blocks = ceil_div(K * lpr, kGatherOpsPerBlock);
    if (blocks > INT32_MAX) return CUSRL_E_UNSUPPORTED;
    switch (unit) {
        case 16: CUSRL_LAUNCH_SCATTER(uint4); break;
        case 8: CUSRL_LAUNCH_SCATTER(uint2); break;
        case 4: CUSRL_LAUNCH_SCATTER(uint32_t); break;
        case 2: CUSRL_LAUNCH_SCATTER(uint16_t); break;
        default: CUSRL_LAUNCH_SCATTER(uint8_t); break;
    }
    return launch_status();
}
