// Version / error strings of libimg2latex_hip.so, caller-owned side lanes, the stream-waits-for-a-word primitive.
#include "common.h"
#include <new>

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

// ------------------------------------------------------------------ side lanes of the training backward pass
// Weight gradients do not feed the backward chain: given an i2l_lanes object the backward entry points enqueue them on
// the CALLER'S side streams (lane 0: decoder and FC layer, lane 1: conv blocks -- so that a conv weight gradient does
// not queue behind the decoder's), forked from the main stream by an event, where they fill the launch gaps and tile
// tails of the data-gradient chain.  i2l_lanes_join(lanes, stream) makes `stream` wait for all of it.  The object holds
// the caller's stream handles and the events the library records on them; it belongs to the caller (create / destroy),
// the library keeps NO state of its own (r03 kept a per-device table behind a mutex).
struct i2l_lanes {
    int n = 0;
    hipStream_t s[I2L_MAX_LANES] = {};
    hipEvent_t fork[I2L_MAX_LANES] = {}, join[I2L_MAX_LANES] = {};
};

extern "C" int i2l_lanes_create(const i2l_stream_t* streams, int n, i2l_lanes** out) {
    if (!streams || !out || n < 1 || n > I2L_MAX_LANES) return I2L_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (!streams[i]) return I2L_ERR_ARG;                 // the default stream would serialise the lanes with everything
    i2l_lanes* l = new (std::nothrow) i2l_lanes();
    if (!l) return I2L_ERR_WORKSPACE;
    l->n = n;
    bool ok = true;
    for (int i = 0; i < n; ++i) {
        l->s[i] = i2l_s(streams[i]);
        ok = ok && hipEventCreateWithFlags(&l->fork[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&l->join[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { i2l_lanes_destroy(l); return I2L_ERR_LAUNCH; }
    *out = l;
    return I2L_OK;
}

extern "C" int i2l_lanes_destroy(i2l_lanes* l) {
    if (!l) return I2L_OK;
    for (int i = 0; i < I2L_MAX_LANES; ++i) {
        if (l->fork[i]) (void)hipEventDestroy(l->fork[i]);
        if (l->join[i]) (void)hipEventDestroy(l->join[i]);
    }
    delete l;
    return I2L_OK;
}

hipStream_t i2l_side_fork(i2l_lanes* l, hipStream_t main, int lane) {
    if (!l || lane < 0) return nullptr;
    if (lane >= l->n) lane = l->n - 1;                       // fewer lanes than roles: share the last one
    if (hipEventRecord(l->fork[lane], main) != hipSuccess) return nullptr;
    if (hipStreamWaitEvent(l->s[lane], l->fork[lane], 0) != hipSuccess) return nullptr;
    return l->s[lane];
}

extern "C" int i2l_lanes_join(i2l_lanes* l, i2l_stream_t stream) {
    if (!l) return I2L_ERR_ARG;
    for (int i = 0; i < l->n; ++i) {
        if (hipEventRecord(l->join[i], l->s[i]) != hipSuccess) return I2L_ERR_LAUNCH;
        if (hipStreamWaitEvent(i2l_s(stream), l->join[i], 0) != hipSuccess) return I2L_ERR_LAUNCH;
    }
    return I2L_OK;
}

// ------------------------------------------------------------------ a stream waits for a device word
// GreedyPipeline holds the encoder of batch i + 1 back until the decode of batch i OWNS its compute units: the grouped
// decode kernels publish `resident_value` to a caller-owned word once every group of the launch has been through its
// placement exchange (group_common.inc.h: count_resident_group), and the encoder stream waits for that word here.
// Launched the other way round -- conv workgroups on the CUs before the decode's -- the two kernels settle into a
// schedule that is 15 - 20 % slower (profiles/r03/ramp.txt).  r03 approximated the dependency with a 30 us delay kernel.
// The wait is one wave polling with s_sleep; it is BOUNDED (timeout_us <= 100 ms): a signal that never comes -- a launch
// that fell back to another kernel, a timed-out group -- costs time, never a hang.  Values compare as a wrapping
// sequence: the wait ends when (int32)(*flag - value) >= 0.
namespace {
__global__ void wait_value32_kernel(const unsigned* flag, unsigned value, long long ticks) {
    const long long t0 = (long long)wall_clock64();          // constant 100 MHz counter
    for (;;) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - value) >= 0) return;
        if ((long long)wall_clock64() - t0 > ticks) return;
        __builtin_amdgcn_s_sleep(4);
    }
}
}  // namespace

extern "C" int i2l_stream_wait_value32(const uint32_t* flag, uint32_t value, float timeout_us, i2l_stream_t stream) {
    if (!flag || !(timeout_us >= 0.f) || timeout_us > 100000.f) return I2L_ERR_ARG;
    hipLaunchKernelGGL(wait_value32_kernel, dim3(1), dim3(1), 0, i2l_s(stream), flag, value, (long long)(timeout_us * 100.f));
    return hipGetLastError() == hipSuccess ? I2L_OK : I2L_ERR_LAUNCH;
}
