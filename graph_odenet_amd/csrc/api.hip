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
