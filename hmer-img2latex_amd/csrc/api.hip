// Version / error strings of libimg2latex_hip.so.
#include "common.h"

extern "C" int i2l_version(void) { return 100; }

extern "C" const char* i2l_error_string(int code) {
    switch (code) {
        case I2L_OK: return "ok";
        case I2L_ERR_ARG: return "invalid argument (null pointer or non-positive dimension)";
        case I2L_ERR_UNSUPPORTED: return "dimension not supported by the gfx950 kernels";
        case I2L_ERR_WORKSPACE: return "workspace missing or too small";
        case I2L_ERR_LAUNCH: return "HIP launch error";
        default: return "unknown error";
    }
}
