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
static inline int i2l_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t i2l_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- generic fp32 NT GEMM (gemm.hip):  C[m][n] = act(sum_k A[m][k] * W[n][k] + bias[n])
// perm_h > 0: output column n = g*perm_h + j is stored at column 4*j + g (LSTM gate interleave).
struct GemmArgs {
    const float* A; int lda;
    const float* W; int ldw;
    const float* bias;        // may be null
    const float* bias2;       // may be null (added as well)
    float* C; int ldc;
    int M, N, K;
    int relu;
    int perm_h;
};
size_t i2l_gemm_workspace_bytes(int M, int N, int K);
int i2l_gemm_nt(const GemmArgs& g, void* ws, size_t ws_bytes, hipStream_t s);
