// api.hip — version / error strings of the C ABI.
#include "common.h"

extern "C" int gode_abi_version(void) { return GODE_ABI_VERSION; }

extern "C" const char* gode_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case GODE_E_NULLPTR: return "graphode: required pointer is NULL";
        case GODE_E_SHAPE: return "graphode: invalid shape / leading dimension";
        case GODE_E_ALIGN: return "graphode: operand not 16-byte aligned";
        case GODE_E_RANGE: return "graphode: argument out of range";
        case GODE_E_UNSUPPORTED: return "graphode: unsupported configuration";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "graphode: unknown error";
}

// ---- run-time options (tuning switches shared by the translation units) ---------------------------------
#include "options.h"
#include <cstdlib>
#include <cstring>

namespace {
int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
int g_gemm_split = -1, g_overlap = -1, g_wgrad_split = -1, g_wgrad_split_small = 0, g_bwd_pc = -1, g_fwd_pc = -1, g_small_fused = -1, g_bwd_wgrad = -1, g_y2_colsum = -1;
}

int gode_opt_gemm_split() { if (g_gemm_split < 0) { const int v = env_int("GODE_GEMM_SPLIT", 2); g_gemm_split = (v == 1 || v == 2) ? v : 0; } return g_gemm_split; }
int gode_opt_wgrad_split() {
    if (g_wgrad_split < 0) { const int v = env_int("GODE_WGRAD_SPLIT", 8); g_wgrad_split = (v == 6 || v == 8) ? v : 0; }
    return g_wgrad_split;
}
int gode_opt_wgrad_split_small() { return g_wgrad_split_small; }
int gode_opt_bwd_pc() { if (g_bwd_pc < 0) g_bwd_pc = env_int("GODE_BWD_PC", 1) != 0; return g_bwd_pc; }
int gode_opt_fwd_pc() { if (g_fwd_pc < 0) g_fwd_pc = env_int("GODE_FWD_PC", 3) & 3; return g_fwd_pc; }
int gode_opt_small_fused() { if (g_small_fused < 0) g_small_fused = env_int("GODE_SMALL_FUSED", 1) != 0; return g_small_fused; }
int gode_opt_bwd_wgrad() { if (g_bwd_wgrad < 0) g_bwd_wgrad = env_int("GODE_BWD_WGRAD", 1) != 0; return g_bwd_wgrad; }
int gode_opt_y2_colsum() { if (g_y2_colsum < 0) g_y2_colsum = env_int("GODE_Y2_COLSUM", 1) != 0; return g_y2_colsum; }
int gode_opt_overlap() { if (g_overlap < 0) g_overlap = env_int("GODE_OVERLAP", 1) != 0; return g_overlap; }

extern "C" int gode_set_option(const char* name, int value) {
    if (!name) return GODE_E_NULLPTR;
    if (!strcmp(name, "gemm_split")) { if (value < 0 || value > 2) return GODE_E_UNSUPPORTED; g_gemm_split = value; return 0; }
    if (!strcmp(name, "overlap")) { g_overlap = value != 0; return 0; }
    if (!strcmp(name, "wgrad_split_small")) { g_wgrad_split_small = value != 0; return 0; }
    if (!strcmp(name, "bwd_pc")) { g_bwd_pc = value != 0; return 0; }
    if (!strcmp(name, "small_fused")) { g_small_fused = value != 0; return 0; }
    if (!strcmp(name, "bwd_wgrad")) { g_bwd_wgrad = value != 0; return 0; }
    if (!strcmp(name, "y2_colsum")) { g_y2_colsum = value != 0; return 0; }
    if (!strcmp(name, "fwd_pc")) { if (value < 0 || value > 3) return GODE_E_UNSUPPORTED; g_fwd_pc = value; return 0; }
    if (!strcmp(name, "wgrad_split")) { if (value != 0 && value != 6 && value != 8) return GODE_E_UNSUPPORTED; g_wgrad_split = value; return 0; }
    return GODE_E_UNSUPPORTED;
}
extern "C" int gode_get_option(const char* name) {
    if (!name) return GODE_E_NULLPTR;
    if (!strcmp(name, "gemm_split")) return gode_opt_gemm_split();
    if (!strcmp(name, "overlap")) return gode_opt_overlap();
    if (!strcmp(name, "wgrad_split")) return gode_opt_wgrad_split();
    if (!strcmp(name, "wgrad_split_small")) return gode_opt_wgrad_split_small();
    if (!strcmp(name, "bwd_pc")) return gode_opt_bwd_pc();
    if (!strcmp(name, "small_fused")) return gode_opt_small_fused();
    if (!strcmp(name, "bwd_wgrad")) return gode_opt_bwd_wgrad();
    if (!strcmp(name, "y2_colsum")) return gode_opt_y2_colsum();
    if (!strcmp(name, "fwd_pc")) return gode_opt_fwd_pc();
    return GODE_E_UNSUPPORTED;
}
