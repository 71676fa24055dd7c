// VERDICT r03 item 4, the micro-experiment: would the decode's per-step products be better off on the matrix pipe?
// `fake_decode_mfma` is profiles/micro/coresident.hip's decode-shaped dummy with the vector-FMA burst replaced by a burst of
// v_mfma_f32_16x16x32_bf16 (the split-bf16 form of the same products: 6 partial products per fp32 product, M = 16 rows --
// half of them padding when a group has 8 rows), same footprint (256 threads, ~200 registers: 144 of them "weight"
// operands, 85 KB of LDS), same idle stretch for the two exchange waits / cell / arg max.  No exchange, no real data flow:
// it isolates (a) how long the burst takes alone, (b) what it costs the real conv encoder beside it, (c) what the conv
// kernels cost IT -- against the FMA-burst dummy of r03.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcoresident_mfma.so coresident_mfma.hip   (driven by coresident3.py)
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 mk(unsigned s) {          // pseudo-random operands: the matrix cores run ~25 % faster on constants
    union { bf16x8 v; unsigned u[4]; } x;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; x.u[i] = (s & 0x7FFF7FFFu) | 0x3C003C00u; x.u[i] &= 0x3FFF3FFFu; x.u[i] |= 0x38003800u; }
    return x.v;
}

// per step: `mfmas` MFMAs per wave (4 independent accumulators), then `idle_ticks` of sleeping, then a barrier
__global__ __launch_bounds__(256) void fake_decode_mfma(int steps, int mfmas, long long idle_ticks, float* sink, long long* burst_ticks) {
    extern __shared__ float lds[];
    bf16x8 w[36];                                              // 144 registers of "weights" (B operands)
#pragma unroll
    for (int i = 0; i < 36; ++i) w[i] = mk(threadIdx.x * 977u + i * 131u + blockIdx.x);
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    lds[threadIdx.x] = 1.0f + 1e-3f * (float)(threadIdx.x & 63);
    __syncthreads();
    long long burst = 0;
    for (int t = 0; t < steps; ++t) {
        const bf16x8 a = mk(__float_as_uint(lds[(threadIdx.x + t) & 255]) + t);      // the step's "h pieces" (A operand)
        const long long b0 = (long long)wall_clock64();
        for (int r = 0; r < mfmas / 36; ++r)
#pragma unroll
            for (int i = 0; i < 36; ++i)
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w[i], acc[i & 3], 0, 0, 0);
        asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));   // results needed here
        const long long t0 = (long long)wall_clock64();
        burst += t0 - b0;
        while ((long long)wall_clock64() - t0 < idle_ticks) __builtin_amdgcn_s_sleep(2);
        lds[threadIdx.x] = acc[t & 3].x * 1e-30f + 1.0f;
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y;
    if (s == 12345.f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0 && burst_ticks) burst_ticks[blockIdx.x] = burst;
}

// the r03 dummy (vector-FMA burst), kept here so that one driver runs both on the same box
__global__ __launch_bounds__(256) void fake_decode_fma(int steps, int fmas, long long idle_ticks, float* sink, long long* burst_ticks) {
    extern __shared__ float lds[];
    f32x2 w[80];
    f32x2 acc[16];
    const float seed = 1.0f + 1e-3f * (float)(threadIdx.x & 63);
#pragma unroll
    for (int i = 0; i < 80; ++i) w[i] = f32x2{seed + i * 1e-4f, seed - i * 1e-4f};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 0.f};
    lds[threadIdx.x] = seed;
    __syncthreads();
    long long burst = 0;
    for (int t = 0; t < steps; ++t) {
        const f32x2 h = f32x2{lds[(threadIdx.x + t) & 255], lds[(threadIdx.x + 2 * t) & 255]};
        const long long b0 = (long long)wall_clock64();
        for (int r = 0; r < fmas / 80; ++r)
#pragma unroll
            for (int i = 0; i < 80; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i & 15]) : "v"(w[i]), "v"(h));
        const long long t0 = (long long)wall_clock64();
        burst += t0 - b0;
        while ((long long)wall_clock64() - t0 < idle_ticks) __builtin_amdgcn_s_sleep(2);
        lds[threadIdx.x] = acc[t & 15].x * 1e-30f + seed;
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    if (s == 12345.f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0 && burst_ticks) burst_ticks[blockIdx.x] = burst;
}

extern "C" int launch_fake(int kind, int workgroups, int steps, int ops, int idle_ns, int lds_bytes, float* sink, long long* burst_ticks,
                           void* stream) {
    auto fn = kind == 0 ? fake_decode_fma : fake_decode_mfma;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    hipLaunchKernelGGL(fn, dim3(workgroups), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), steps, ops,
                       (long long)idle_ns / 10, sink, burst_ticks);          // wall_clock64 ticks at 100 MHz
    return (int)hipGetLastError();
}
