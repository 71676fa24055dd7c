// Bare MFMA loops on every CU: v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 (same FLOPs per instruction),
// operands in registers, 8 independent accumulator chains per wave, 1 or 2 waves per SIMD.  Prints achieved TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip && ./mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    // operand data: random normal-range bf16 per lane and per element, four different operand pairs in rotation (the data
    // toggles the multiplier arrays as real activations / weights do; constant operands draw far less power)
    bf16x8 av[4], bv[4];
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; av[q][i] = (short)(((h >> 16) & 0x80ff) | 0x3f00);
            h = h * 1664525u + 1013904223u; bv[q][i] = (short)(((h >> 16) & 0x80ff) | 0x3f00);
        }
    if (RANDOM == 0) for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { av[q][i] = 0x3f80; bv[q][i] = 0x3f00; }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[c & 3], bv[(c + 1) & 3], acc[c], 0, 0, 0);
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    } else {
        f32x4 acc[16];
        for (int c = 0; c < 16; ++c) for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[c & 3], bv[(c + 1) & 3], acc[c], 0, 0, 0);
        for (int c = 0; c < 16; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rnd : {0, 1}) for (int wgs : {256, 512}) {
        for (int shape : {32, 16}) {
            const int iters = 20000;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (shape == 32 && rnd) hipLaunchKernelGGL((k<32, 1>), dim3(wgs), dim3(256), 0, 0, out, iters);
                else if (shape == 32) hipLaunchKernelGGL((k<32, 0>), dim3(wgs), dim3(256), 0, 0, out, iters);
                else if (rnd) hipLaunchKernelGGL((k<16, 1>), dim3(wgs), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL((k<16, 0>), dim3(wgs), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double mf = shape == 32 ? 4.0 : 16.0;          // MFMAs per iteration per wave
                const double flops = (double)wgs * 4 * iters * mf * (shape == 32 ? 32768.0 : 16384.0);
                if (rep == 2) printf("%s operands, wgs %d shape %dx%d: %.3f ms  %.0f TFLOP/s\n", rnd ? "random" : "constant", wgs, shape, shape, ms, flops / ms / 1e9);
            }
        }
    }
    return 0;
}
