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

// ------------------------------------------------------------------ side stream of the training backward pass
// Weight gradients do not feed the backward chain: with I2L_FLAG_SIDE_WGRAD the backward entry points enqueue them on
// this non-blocking stream (one per device), forked from the caller's stream by an event, so that they fill the launch
// gaps and tile tails of the data-gradient chain.  i2l_side_stream_join(stream) makes `stream` wait for all of it.
#include <mutex>
namespace {
struct Side { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool tried = false; };
Side g_side[16];
std::mutex g_side_mu;
Side* side_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Side& sd = g_side[dev];
    if (!sd.tried) {
        sd.tried = true;
        if (hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking) != hipSuccess) { sd.s = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess) { sd.s = nullptr; return nullptr; }
    }
    return sd.s ? &sd : nullptr;
}
}  // namespace

hipStream_t i2l_side_fork(hipStream_t main) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    Side* sd = side_of_current_device();
    if (!sd) return nullptr;
    if (hipEventRecord(sd->fork, main) != hipSuccess) return nullptr;
    if (hipStreamWaitEvent(sd->s, sd->fork, 0) != hipSuccess) return nullptr;
    return sd->s;
}

extern "C" int i2l_side_stream_join(i2l_stream_t stream) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    Side* sd = side_of_current_device();
    if (!sd) return I2L_OK;                              // nothing was ever forked
    if (hipEventRecord(sd->join, sd->s) != hipSuccess) return I2L_ERR_LAUNCH;
    if (hipStreamWaitEvent(i2l_s(stream), sd->join, 0) != hipSuccess) return I2L_ERR_LAUNCH;
    return I2L_OK;
}
