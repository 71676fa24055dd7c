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

// ------------------------------------------------------------------ side streams of the training backward pass
// Weight gradients do not feed the backward chain: with I2L_FLAG_SIDE_WGRAD the backward entry points enqueue them on
// one of two non-blocking streams per device (lane 0: decoder and FC layer, lane 1: conv blocks -- so that a conv weight
// gradient does not queue behind the decoder's), forked from the caller's stream by an event, where they fill the launch
// gaps and tile tails of the data-gradient chain.  i2l_side_stream_join(stream) makes `stream` wait for all of it.
#include <mutex>
namespace {
constexpr int N_LANES = 2;
struct Side { hipStream_t s[N_LANES] = {}; hipEvent_t fork[N_LANES] = {}, join[N_LANES] = {}; bool tried = false, ok = false; };
Side g_side[16];
std::mutex g_side_mu;
Side* side_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Side& sd = g_side[dev];
    if (!sd.tried) {
        sd.tried = true;
        sd.ok = true;
        for (int i = 0; i < N_LANES; ++i)
            if (hipStreamCreateWithFlags(&sd.s[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&sd.fork[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&sd.join[i], hipEventDisableTiming) != hipSuccess)
                sd.ok = false;
    }
    return sd.ok ? &sd : nullptr;
}
}  // namespace

hipStream_t i2l_side_fork(hipStream_t main, int lane) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    Side* sd = side_of_current_device();
    if (!sd || lane < 0 || lane >= N_LANES) return nullptr;
    if (hipEventRecord(sd->fork[lane], main) != hipSuccess) return nullptr;
    if (hipStreamWaitEvent(sd->s[lane], sd->fork[lane], 0) != hipSuccess) return nullptr;
    return sd->s[lane];
}

extern "C" int i2l_side_stream_join(i2l_stream_t stream) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    Side* sd = side_of_current_device();
    if (!sd) return I2L_OK;                              // nothing was ever forked
    for (int i = 0; i < N_LANES; ++i) {
        if (hipEventRecord(sd->join[i], sd->s[i]) != hipSuccess) return I2L_ERR_LAUNCH;
        if (hipStreamWaitEvent(i2l_s(stream), sd->join[i], 0) != hipSuccess) return I2L_ERR_LAUNCH;
    }
    return I2L_OK;
}

// ------------------------------------------------------------------ a short device-side delay on a stream
// GreedyPipeline holds the encoder of batch i + 1 back until the decode of batch i has been launched (an event) PLUS a few
// tens of microseconds, so that the decode's 256 workgroups are resident before the first conv workgroup asks for a CU:
// launched the other way round the two kernels settle into a schedule that is 15 - 20 % slower (profiles/r03/ramp.txt).
namespace {
__global__ void spin_kernel(long long ticks) {
    const long long t0 = (long long)wall_clock64();      // constant 100 MHz counter
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

extern "C" int i2l_stream_spin_us(float microseconds, i2l_stream_t stream) {
    if (!(microseconds >= 0.f) || microseconds > 10000.f) return I2L_ERR_ARG;      // bounded: at most 10 ms
    if (microseconds == 0.f) return I2L_OK;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, i2l_s(stream), (long long)(microseconds * 100.f));
    return hipGetLastError() == hipSuccess ? I2L_OK : I2L_ERR_LAUNCH;
}
