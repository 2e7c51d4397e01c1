// ABI bookkeeping for libcusrl_hip.so.
#include "common.hpp"

extern "C" int cusrl_abi_version(void) { return CUSRL_ABI_VERSION; }

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
