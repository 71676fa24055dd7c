// Shared host-side helpers for libimg2latex_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/img2latex_hip.h"

#define I2L_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return I2L_ERR_LAUNCH; \
    } while (0)

static inline hipStream_t i2l_s(i2l_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
// side lanes (api.hip): the caller's stream of that lane, made to wait for everything enqueued on `main` so far.
// Returns nullptr without lanes or on a HIP error (the caller then stays on `main`).  i2l_lanes_join() is the way back.
hipStream_t i2l_side_fork(i2l_lanes* lanes, hipStream_t main, int lane);   // lane 0: decoder + FC weight gradients, lane 1: conv blocks
// hipFuncAttributeMaxDynamicSharedMemorySize is per device: set it once per (call site, device).  `done` is the call site's
// function-local cache (bit d = set on device d); an idempotent cache, not a switch.
#include <atomic>
static inline bool i2l_lds_attr(const void* fn, size_t bytes, std::atomic<unsigned>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev >= 0 && dev < 32 && ((done.load(std::memory_order_relaxed) >> dev) & 1u)) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    if (dev >= 0 && dev < 32) done.fetch_or(1u << dev, std::memory_order_relaxed);
    return true;
}
static inline int i2l_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t i2l_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- generic fp32 MFMA GEMM (gemm.hip):
//   C[m][n] (+)= alpha * act(sum_z sum_k A_z(m,k) * W_z(n,k) + bias[n] + bias2[n])
//   a_kc != 0: A(m,k) = A[m*lda + k] (K-contiguous) else A[k*lda + m];  same for W with n.
//   perm_h > 0: output column n = g*perm_h + j is stored at column 4*j + g (LSTM gate interleave).
struct GemmArgs {
    const float* A; long lda; int a_kc;
    const float* W; long ldw; int w_kc;
    long bsa, bsw; int nz;    // nz operand pairs (A + z*bsa, W + z*bsw) are summed
    const float* bias;        // may be null
    const float* bias2;       // may be null (added as well)
    float* C; int ldc;
    int M, N, K;
    int relu;
    int perm_h;
    float alpha;
    int accumulate;           // C += result instead of C = result
    int split_bf16;           // allow the split-bf16 matrix-core kernel (fp32-grade, not an fmaf chain); gemm.hip
    int conv_h, conv_w;       // > 0 (split-bf16 kernel only): W is an implicit im2col^T of a (Cin, conv_h, conv_w) image per
                              // operand pair: W(n = ci*9 + tap, k = y*conv_w + x) = img[ci][y + tap/3 - 1][x + tap%3 - 1], 0 outside
    // with conv_h > 0 and pool_y != null: A is NOT the un-pooled gradient but the POOLED one, (M, conv_h/2, conv_w/2) per
    // operand pair (lda = bsa / M = pooled plane size), and is un-pooled on the fly:
    //   A(m, k = y*conv_w + x) = (pool_y > 0 && pool_am == 2*(y&1) + (x&1)) ? A_pooled[m][y/2][x/2] : 0      (conv_w % 8 == 0)
    const float* pool_y;
    const unsigned char* pool_am;
};
static inline GemmArgs gemm_args() {
    GemmArgs g{};
    g.a_kc = 1; g.w_kc = 1; g.nz = 1; g.alpha = 1.f;
    return g;
}
size_t i2l_gemm_workspace_bytes(int M, int N, int K, int nz = 1);
int i2l_gemm(const GemmArgs& g, void* ws, size_t ws_bytes, hipStream_t s);
bool i2l_gemm_split_bf16_ok(const GemmArgs& g);   // would i2l_gemm run this on the split-bf16 kernel?

// ---- 3 x bf16 split conv block on the bf16 matrix cores (conv_bf16x3.hip); inference forward of blocks with
//      Cin % 16 == 0 and Cout % 64 == 0 (shape test only: the caller decides on I2L_FLAG_EXACT_FP32)
bool i2l_conv_bf16x3_applicable(int Cin, int Cout);
bool i2l_conv_bf16x3_full_applicable(int Cin, int Cout);   // the full-resolution (data gradient) flavour: also Cout == 32
size_t i2l_conv_bf16x3_workspace_bytes(int Cin, int Cout);
int i2l_conv_bf16x3_run(const float* x, const float* w, const float* bias, float* y, unsigned char* amax, int B, int Cin,
                        int H, int W, int Cout, void* workspace, size_t workspace_bytes, hipStream_t s, int full = 0,
                        int weights_packed = 0);
// (full != 0: plain full-resolution conv with the flipped / transposed filter = a block's data gradient; H, W even)
// first conv block (Cin <= 3): whole K in one or two bf16 k-steps, filters in registers, im2col image in LDS
bool i2l_conv_smallk_applicable(int Cin, int Cout);
int i2l_conv_smallk_run(const float* x, const float* w, const float* bias, float* y, unsigned char* amax, int B, int Cin,
                        int H, int W, int Cout, void* workspace, size_t workspace_bytes, hipStream_t s);
// amax != NULL (training forward): both split kernels list the pooling windows whose arg max / ReLU decision is a
// near-tie and re-evaluate them in fp32; the list lives in the workspace (after the packed weights of the bf16x3 kernel)
size_t i2l_conv_fixlist_bytes(void);
