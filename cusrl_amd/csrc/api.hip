// ABI bookkeeping for libcusrl_hip.so.
#include "common.hpp"

#include <atomic>
#include <string.h>

extern "C" int cusrl_abi_version(void) { return CUSRL_ABI_VERSION; }

// ---- options: what rounds 2-5 read from CUSRL_* environment variables INSIDE the launch entry points ----------------------
namespace cusrl {
static std::atomic<int64_t> g_options[kNumOptions];
static const char *const kOptionNames[kNumOptions] = {"gae_policy", "gae_block", "loss_policy", "push_policy",
                                                      "colsum_rows", "head_rows", "gru_bias_rows"};
int64_t option(Option which) { return g_options[which].load(std::memory_order_relaxed); }

static bool option_accepts(int which, int64_t value) {
    if (value == 0) return true;  // back to the kernel's own rule
    switch (which) {
        case kOptGaePolicy: return value == 1 || value == 6 || value == 8;  // 1 + {0, 5, 7}
        case kOptGaeBlock: return value == 128 || value == 256;
        case kOptLossPolicy:
        case kOptPushPolicy: return value == 1 || value == 2;
        case kOptColsumRows: return value >= 4 && value <= 4096;
        case kOptHeadRows: return value >= 8 && value <= 4096;
        case kOptGruBiasRows: return value == 4 || value == 8 || value == 16 || value == 32;
        default: return false;
    }
}
}  // namespace cusrl

extern "C" int cusrl_set_option(const char *key, int64_t value) {
    if (!key) return CUSRL_E_INVALID;
    for (int k = 0; k < cusrl::kNumOptions; ++k)
        if (strcmp(key, cusrl::kOptionNames[k]) == 0) {
            if (!cusrl::option_accepts(k, value)) return CUSRL_E_UNSUPPORTED;
            cusrl::g_options[k].store(value, std::memory_order_relaxed);
            return 0;
        }
    return CUSRL_E_INVALID;
}

extern "C" int cusrl_get_option(const char *key, int64_t *value_out) {
    if (!key || !value_out) return CUSRL_E_INVALID;
    for (int k = 0; k < cusrl::kNumOptions; ++k)
        if (strcmp(key, cusrl::kOptionNames[k]) == 0) {
            *value_out = cusrl::option(cusrl::Option(k));
            return 0;
        }
    return CUSRL_E_INVALID;
}

extern "C" const char *cusrl_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case CUSRL_E_INVALID: return "invalid argument (null pointer, negative size or inconsistent shapes)";
        case CUSRL_E_TOO_MANY: return "too many leaves in one launch (CUSRL_MAX_FIELDS)";
        case CUSRL_E_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case CUSRL_E_COMM: return "RCCL unavailable or an RCCL call failed (cusrl_comm_last_error has the text)";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown cusrl error";
}

// Node census of a captured hipGraph: what a captured step really consists of (host-side walk, no launch).
extern "C" int cusrl_graph_census(void *graph, int64_t *type_counts, int n_types, char *names, int64_t capacity,
                                  int64_t *names_len) {
    if (!graph || !type_counts || n_types <= 0 || capacity < 0 || (capacity > 0 && !names)) return CUSRL_E_INVALID;
    hipGraph_t g = static_cast<hipGraph_t>(graph);
    size_t count = 0;
    if (hipError_t e = hipGraphGetNodes(g, nullptr, &count)) return static_cast<int>(e);
    hipGraphNode_t *nodes = count ? new hipGraphNode_t[count] : nullptr;
    if (count)
        if (hipError_t e = hipGraphGetNodes(g, nodes, &count)) {
            delete[] nodes;
            return static_cast<int>(e);
        }
    for (int k = 0; k < n_types; ++k) type_counts[k] = 0;
    int64_t used = 0;
    auto append = [&](const char *text) {
        for (const char *c = text; *c; ++c, ++used)
            if (used < capacity) names[used] = *c;
        if (used < capacity) names[used] = '\n';
        ++used;
    };
    int rc = 0;
    for (size_t i = 0; i < count && rc == 0; ++i) {
        hipGraphNodeType type;
        if (hipError_t e = hipGraphNodeGetType(nodes[i], &type)) {
            rc = static_cast<int>(e);
            break;
        }
        if (int(type) >= 0 && int(type) < n_types) ++type_counts[int(type)];
        if (type != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams params;
        if (hipGraphKernelNodeGetParams(nodes[i], &params) != hipSuccess || !params.func) {
            append("?");
            continue;
        }
        // a kernel launched through its host stub (ATen, this library) resolves by pointer; module launches
        // (rocBLAS / hipBLASLt code objects) carry a hipFunction_t instead
        const char *name = hipKernelNameRefByPtr(params.func, nullptr);
        if (!name) name = hipKernelNameRef(static_cast<hipFunction_t>(params.func));
        append(name ? name : "?");
    }
    delete[] nodes;
    if (names_len) *names_len = used;
    return rc;
}

namespace cusrl {

// What a memset node does, as a kernel: `width` elements of `element_size` bytes (1 / 2 / 4) per row, `height` rows `pitch`
// bytes apart, every element = the low bytes of `value`.
__global__ __launch_bounds__(kBlock) void graph_fill_kernel(unsigned char *__restrict__ dst, unsigned int value,
                                                            unsigned int element_size, size_t width, size_t height,
                                                            size_t pitch) {
    const size_t total = width * height;
    for (size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x; i < total; i += size_t(gridDim.x) * kBlock) {
        const size_t row = i / width, col = i - row * width;
        unsigned char *p = dst + row * pitch + col * element_size;
        if (element_size == 4)
            *reinterpret_cast<unsigned int *>(p) = value;
        else if (element_size == 2)
            *reinterpret_cast<unsigned short *>(p) = static_cast<unsigned short>(value);
        else
            *p = static_cast<unsigned char>(value);
    }
}

}  // namespace cusrl

// Replace every memset node of a captured (not yet instantiated) hipGraph by a kernel node with the same effect and the same
// edges.  Why: on this stack (ROCm 7.0 runtime of PyTorch 2.10) a replayed graph does not execute its memset nodes reliably —
// scripts/probe_aten_reduce_capture.py: plain torch, `x.sum(0)` of [1024, 128] captured and replayed 2000 times is wrong in
// 4-100 % of the replays, every shape whose reduction needs no semaphore memset is right in all of them (DESIGN.md
// section 5).  ATen's global reductions (Reduce.cuh:1294-1301) zero their semaphores with hipMemsetAsync, so any hook that
// calls `.sum()` / `.mean()` over >= ~1024 rows inside a captured phase brings such nodes in.
extern "C" int cusrl_graph_replace_memsets(void *graph, int64_t *replaced_out) {
    if (!graph) return CUSRL_E_INVALID;
    hipGraph_t g = static_cast<hipGraph_t>(graph);
    size_t count = 0;
    if (hipError_t e = hipGraphGetNodes(g, nullptr, &count)) return static_cast<int>(e);
    if (replaced_out) *replaced_out = 0;
    if (count == 0) return 0;
    hipGraphNode_t *nodes = new hipGraphNode_t[count];
    int rc = 0;
    int64_t replaced = 0;
    if (hipError_t e = hipGraphGetNodes(g, nodes, &count)) rc = static_cast<int>(e);
    for (size_t i = 0; i < count && rc == 0; ++i) {
        hipGraphNodeType type;
        if (hipError_t e = hipGraphNodeGetType(nodes[i], &type)) {
            rc = static_cast<int>(e);
            break;
        }
        if (type != hipGraphNodeTypeMemset) continue;
        hipMemsetParams m;
        if (hipError_t e = hipGraphMemsetNodeGetParams(nodes[i], &m)) {
            rc = static_cast<int>(e);
            break;
        }
        if (m.elementSize != 1 && m.elementSize != 2 && m.elementSize != 4) {
            rc = CUSRL_E_UNSUPPORTED;
            break;
        }
        size_t n_in = 0, n_out = 0;
        if (hipError_t e = hipGraphNodeGetDependencies(nodes[i], nullptr, &n_in)) {
            rc = static_cast<int>(e);
            break;
        }
        if (hipError_t e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &n_out)) {
            rc = static_cast<int>(e);
            break;
        }
        hipGraphNode_t *in = new hipGraphNode_t[n_in + 1], *out = new hipGraphNode_t[n_out + 1];
        hipGraphNode_t *from = new hipGraphNode_t[n_out + 1];
        hipError_t e = hipSuccess;
        if (n_in) e = hipGraphNodeGetDependencies(nodes[i], in, &n_in);
        if (e == hipSuccess && n_out) e = hipGraphNodeGetDependentNodes(nodes[i], out, &n_out);
        if (e == hipSuccess) {
            unsigned char *dst = static_cast<unsigned char *>(m.dst);
            unsigned int value = m.value, element_size = m.elementSize;
            size_t width = m.width, height = m.height ? m.height : 1, pitch = m.pitch;
            void *args[] = {&dst, &value, &element_size, &width, &height, &pitch};
            const size_t total = width * height;
            const size_t blocks = total ? (total + cusrl::kBlock - 1) / cusrl::kBlock : 1;
            hipKernelNodeParams k = {};
            k.func = reinterpret_cast<void *>(cusrl::graph_fill_kernel);
            k.gridDim = dim3(static_cast<unsigned int>(blocks > 1024 ? 1024 : blocks));
            k.blockDim = dim3(cusrl::kBlock);
            k.sharedMemBytes = 0;
            k.kernelParams = args;
            k.extra = nullptr;
            hipGraphNode_t fill;
            e = hipGraphAddKernelNode(&fill, g, n_in ? in : nullptr, n_in, &k);
            if (e == hipSuccess && n_out) {
                for (size_t j = 0; j < n_out; ++j) from[j] = fill;
                e = hipGraphAddDependencies(g, from, out, n_out);
            }
            if (e == hipSuccess) e = hipGraphDestroyNode(nodes[i]);
            if (e == hipSuccess) ++replaced;
        }
        delete[] in;
        delete[] out;
        delete[] from;
        if (e != hipSuccess) rc = static_cast<int>(e);
    }
    delete[] nodes;
    if (replaced_out) *replaced_out = replaced;
    return rc;
}
