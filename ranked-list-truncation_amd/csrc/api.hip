// ABI bookkeeping for librlt_hip.so.
#include "common.h"

#include <atomic>
#include <stdlib.h>
#include <string.h>

// Precision mode of the MFMA contractions: RLT_PRECISION_FP32 = exact fp32 MFMA, RLT_PRECISION_BF16X3 = split-bf16 (3 bf16
// products, 16 operand bits), RLT_PRECISION_BF16X6 = fp32-faithful six-product split (all 24 operand bits; GEMM family, list
// attention at head dims <= 64, BiLSTM recurrences).
//   g_default  the process default: env RLT_PRECISION at first use, else BF16X6 (the reference computes in fp32 end to end,
//              models/AttnCut.py:8-14, so the default of a drop-in is a reference-faithful mode; bf16x3 is opt-in); written
//              only by rlt_set_precision
//   tl_scope   the mode of the entry-point call running on this thread (common.h: RltPrecScope), -1 outside any call
static std::atomic<int> g_default{-1};      // (atomic: the first call of any thread may initialise it while another thread sets it)
static thread_local int tl_scope = -1;
static int default_precision() {
    int d = g_default.load(std::memory_order_relaxed);
    if (d < 0) {
        const char* e = getenv("RLT_PRECISION");
        int from_env = RLT_PRECISION_BF16X6;
        if (e && (!strcmp(e, "fp32") || !strcmp(e, "0"))) from_env = RLT_PRECISION_FP32;
        else if (e && (!strcmp(e, "bf16x3") || !strcmp(e, "1"))) from_env = RLT_PRECISION_BF16X3;
        // a concurrent rlt_set_precision wins over the environment
        if (g_default.compare_exchange_strong(d, from_env, std::memory_order_relaxed)) d = from_env;
    }
    return d;
}
int rlt_precision() { return tl_scope >= 0 ? tl_scope : default_precision(); }
RltPrecScope::RltPrecScope(int p) : saved(tl_scope), set(p >= 0) { if (set) tl_scope = p; }
RltPrecScope::~RltPrecScope() { if (set) tl_scope = saved; }

extern "C" {

int rlt_set_precision(int mode) {
    if (mode != RLT_PRECISION_FP32 && mode != RLT_PRECISION_BF16X3 && mode != RLT_PRECISION_BF16X6) return RLT_E_ARG;
    g_default.store(mode, std::memory_order_relaxed);
    return 0;
}
int rlt_get_precision(void) { return default_precision(); }

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
