// ABI bookkeeping for librlt_hip.so.
#include "common.h"

extern "C" {

int rlt_abi_version(void) { return RLT_ABI_VERSION; }

const char* rlt_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case RLT_E_ARG: return "RLT_E_ARG: null pointer or non-positive dimension";
        case RLT_E_SHAPE: return "RLT_E_SHAPE: dimension outside the supported range";
        case RLT_E_WORKSPACE: return "RLT_E_WORKSPACE: workspace too small";
        case RLT_E_ALIGN: return "RLT_E_ALIGN: pointer or leading dimension not 16-byte aligned";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown rlt error";
}

}  // extern "C"
