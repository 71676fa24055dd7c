// Can a decode-shaped workgroup and a conv workgroup share every CU, and what does each lose?  (the
// co-resident decode.)  `fake_decode` has the resource footprint and the rhythm of the proposed 8-member grouped
// decode -- 256 threads, ~200 live registers, 85 KB of LDS, per step a burst of v_pk_fma_f32 at the vector peak (the
// recurrent product + logits: `fmas` packed FMAs per lane) followed by an idle stretch of `idle_ns` (the two exchange
// waits, cell and arg max) -- but no exchange: it isolates the question of resource sharing from the kernel itself.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcoresident.so coresident.hip   (driven by coresident.py)
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void fake_decode(int steps, int fmas, long long idle_ticks, float* sink) {
    extern __shared__ float lds[];
    f32x2 w[80];                                   // 160 registers of "weights"
    f32x2 acc[16];
    const float seed = 1.0f + 1e-3f * (float)(threadIdx.x & 63);
#pragma unroll
    for (int i = 0; i < 80; ++i) w[i] = f32x2{seed + i * 1e-4f, seed - i * 1e-4f};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 0.f};
    lds[threadIdx.x] = seed;
    __syncthreads();
    for (int t = 0; t < steps; ++t) {
        const f32x2 h = f32x2{lds[(threadIdx.x + t) & 255], lds[(threadIdx.x + 2 * t) & 255]};
        for (int r = 0; r < fmas / 80; ++r)
#pragma unroll
            for (int i = 0; i < 80; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i & 15]) : "v"(w[i]), "v"(h));
        const long long t0 = (long long)wall_clock64();
        while ((long long)wall_clock64() - t0 < idle_ticks) __builtin_amdgcn_s_sleep(2);
        lds[threadIdx.x] = acc[t & 15].x * 1e-30f + seed;
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    if (s == 12345.f) sink[threadIdx.x] = s;
}

extern "C" int launch_fake_decode(int workgroups, int steps, int fmas, int idle_ns, int lds_bytes, float* sink, void* stream) {
    static bool once = hipFuncSetAttribute(reinterpret_cast<const void*>(fake_decode), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024) == hipSuccess;
    (void)once;
    hipLaunchKernelGGL(fake_decode, dim3(workgroups), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), steps, fmas,
                       (long long)idle_ns / 10, sink);          // wall_clock64 ticks at 100 MHz
    return (int)hipGetLastError();
}
