// ABI bookkeeping for librlt_hip.so.
#include "common.h"

#include <stdlib.h>
#include <string.h>

// precision mode of the MFMA contractions: 0 = exact fp32 MFMA, 1 = split-bf16 (3 bf16 products, fp32 accumulate),
// 2 = fp32-faithful six-product split for the GEMM family (attention and BiLSTM on the exact fp32 MFMA kernels)
static int g_precision = -1;
int rlt_precision() {
    if (g_precision < 0) {
        const char* e = getenv("RLT_PRECISION");
        if (e && (!strcmp(e, "fp32") || !strcmp(e, "0"))) g_precision = RLT_PRECISION_FP32;
        else if (e && (!strcmp(e, "bf16x6") || !strcmp(e, "2"))) g_precision = RLT_PRECISION_BF16X6;
        else g_precision = RLT_PRECISION_BF16X3;
    }
    return g_precision;
}

extern "C" {

int rlt_set_precision(int mode) {
    if (mode != RLT_PRECISION_FP32 && mode != RLT_PRECISION_BF16X3 && mode != RLT_PRECISION_BF16X6) return RLT_E_ARG;
    g_precision = mode;
    return 0;
}
int rlt_get_precision(void) { return rlt_precision(); }

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
