// Fused Conv2d(3x3, pad 1, stride 1) + bias + ReLU + MaxPool2d(2) for NCHW fp32.
// Replaces the three separate full-tensor passes ATen makes per block of
// CNNEncoder.cnn_layers (encoder.py:78-95, run at :122): the pre-pool activation
// never leaves registers, so each block reads its input once and writes only the
// pooled output (the "fused" byte count of SURVEY.md 8d).
//
// General direct kernel (any Cin/Cout/H/W, floor pooling):
//   workgroup = 256 threads = an 8 x 32 tile of POOLED outputs of one image and one
//   group of CO_T output channels.  The 18 x 66 input patch (1-pixel halo, zero
//   padding = Conv2d padding) of CI_T input channels at a time is staged in LDS
//   with coalesced row loads; every thread keeps the 2x2 pre-pool quad of CO_T
//   channels in registers, pools in registers (no cross-lane traffic) and applies
//   bias+ReLU after the max (max and +bias/ReLU commute: both are monotone).
//   Weights are wave-uniform, so the compiler reads them through the scalar cache.
#include "common.h"

namespace {

constexpr int TPH = 8, TPW = 32;           // pooled tile
constexpr int TIH = 2 * TPH + 2;           // 18 input rows
constexpr int TIW = 2 * TPW + 2;           // 66 input cols
constexpr int TIWP = TIW + 2;              // padded row stride (68 floats: rows stay 8B aligned)
constexpr int CI_T = 8;

template <int CO_T>
__global__ __launch_bounds__(256) void conv3x3_relu_pool2_direct(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int B, int Cin, int H, int W, int Cout, int Hp, int Wp, int co_groups) {
    __shared__ float tile[CI_T * TIH * TIWP];
    const int tid = threadIdx.x;
    const int px = tid & (TPW - 1), py = tid >> 5;
    const int b = blockIdx.x / co_groups;                    // co-groups of one image are adjacent:
    const int co0 = (blockIdx.x - b * co_groups) * CO_T;     // they re-read the same patch from L2
    const int PX0 = blockIdx.y * TPW, PY0 = blockIdx.z * TPH;
    const int ix0 = 2 * PX0 - 1, iy0 = 2 * PY0 - 1;          // top-left of the input patch

    float acc[CO_T][4];
#pragma unroll
    for (int c = 0; c < CO_T; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[c][q] = 0.f;

    const float* xb = x + (size_t)b * Cin * H * W;
    for (int ci0 = 0; ci0 < Cin; ci0 += CI_T) {
        const int nci = min(CI_T, Cin - ci0);
        __syncthreads();
        for (int idx = tid; idx < nci * TIH * TIW; idx += 256) {
            const int ci = idx / (TIH * TIW);
            const int rem = idx - ci * (TIH * TIW);
            const int r = rem / TIW, c = rem - r * TIW;
            const int iy = iy0 + r, ix = ix0 + c;
            float v = 0.f;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xb[((size_t)(ci0 + ci) * H + iy) * W + ix];
            tile[(ci * TIH + r) * TIWP + c] = v;
        }
        __syncthreads();
        for (int ci = 0; ci < nci; ++ci) {
            float in[4][4];
            const float* tp = &tile[(ci * TIH + 2 * py) * TIWP + 2 * px];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 a = *reinterpret_cast<const float2*>(tp + r * TIWP);
                const float2 c2 = *reinterpret_cast<const float2*>(tp + r * TIWP + 2);
                in[r][0] = a.x; in[r][1] = a.y; in[r][2] = c2.x; in[r][3] = c2.y;
            }
#pragma unroll
            for (int c = 0; c < CO_T; ++c) {
                if (co0 + c < Cout) {                       // block-uniform
                    const float* wp = w + ((size_t)(co0 + c) * Cin + (ci0 + ci)) * 9;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float wv = wp[ky * 3 + kx];
                            acc[c][0] = fmaf(in[ky][kx], wv, acc[c][0]);
                            acc[c][1] = fmaf(in[ky][kx + 1], wv, acc[c][1]);
                            acc[c][2] = fmaf(in[ky + 1][kx], wv, acc[c][2]);
                            acc[c][3] = fmaf(in[ky + 1][kx + 1], wv, acc[c][3]);
                        }
                }
            }
        }
    }
    const int PY = PY0 + py, PX = PX0 + px;
    if (PY < Hp && PX < Wp) {
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            if (co0 + c < Cout) {
                float v = fmaxf(fmaxf(acc[c][0], acc[c][1]), fmaxf(acc[c][2], acc[c][3])) + bias[co0 + c];
                y[(((size_t)b * Cout + co0 + c) * Hp + PY) * Wp + PX] = fmaxf(v, 0.f);
            }
        }
    }
}

}  // namespace

extern "C" int i2l_conv3x3_relu_pool2_fwd(const float* x, const float* w, const float* bias, float* y,
                                           int B, int Cin, int H, int W, int Cout, i2l_stream_t stream) {
    if (!x || !w || !bias || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H < 2 || W < 2) return I2L_ERR_ARG;
    const int Hp = H / 2, Wp = W / 2;
    constexpr int CO_T = 8;
    const int co_groups = i2l_cdiv(Cout, CO_T);
    if (i2l_cdiv(Wp, TPW) > 65535 || i2l_cdiv(Hp, TPH) > 65535) return I2L_ERR_UNSUPPORTED;
    dim3 grid(B * co_groups, i2l_cdiv(Wp, TPW), i2l_cdiv(Hp, TPH));
    hipLaunchKernelGGL(conv3x3_relu_pool2_direct<CO_T>, grid, dim3(256), 0, i2l_s(stream), x, w, bias, y, B, Cin,
                       H, W, Cout, Hp, Wp, co_groups);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
