// Fused Conv2d(3x3, pad 1, stride 1) + bias + ReLU + MaxPool2d(2) for NCHW fp32.
// Replaces the three separate full-tensor passes ATen makes per block of
// CNNEncoder.cnn_layers (encoder.py:78-95, run at :122): the pre-pool activation
// never leaves registers, so each block reads its input once and writes only the
// pooled output (the "fused" byte count of SURVEY.md 8d).
//
// Two kernels behind one entry point:
//
// (1) conv3x3_mfma_kernel -- implicit GEMM on the fp32 matrix cores (Cout % 32 == 0).
//     v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fmaf chain, so this keeps
//     exact fp32 numerics at up to 2.3x the VALU rate.
//       M (32 rows of an MFMA tile) = 2 conv rows x 16 conv cols, ordered (pooled x, dy, dx):
//         row i = 4*pp + 2*dy + dx.  In the 32x32 C/D layout a lane then holds, for ITS output
//         channel, whole 2x2 pooling quads in consecutive registers: pooling is 3 v_max per
//         output, no cross-lane traffic, no LDS round trip.
//       N (32 cols) = 32 output channels;  K = (ci pair, tap): lanes 0-31 feed input channel
//         2*cp, lanes 32-63 channel 2*cp+1 of the same tap, so every A/B operand read is
//         ds_read_b32 <per-lane base> offset:<immediate>.
//     Workgroup (4 waves, 2x2) = 8 conv rows x 32 conv cols x 64 output channels; each wave
//     2 M-tiles x 2 N-tiles = 4 accumulators.  Input channels are consumed in chunks of CI_BLK:
//     the 10 x 34 input patch (halo = Conv2d zero padding) and the pre-packed weight slab of
//     the chunk are staged in LDS (row stride 48 floats: the two conv rows of an M-tile land
//     on disjoint bank halves -> conflict-free operand reads).
//
// (2) conv3x3_relu_pool2_direct -- general VALU kernel (any Cout), used when Cout % 32 != 0.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// (1) fp32-MFMA implicit GEMM
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int M_TR = 8, M_TC = 32;             // conv rows / cols per workgroup
constexpr int M_IR = M_TR + 2;                 // 10 input rows
constexpr int M_IC = M_TC + 2;                 // 34 input cols
constexpr int M_ST = 48;                       // LDS row stride (floats), == 16 mod 32
constexpr int M_CH = M_IR * M_ST;              // 480 floats per staged channel

// wpack[co_blk][chunk][cp][tap][h][CO_BLK]: value = w[co_blk*CO_BLK + co][chunk*CI_BLK + 2*cp + h][tap], 0 outside.
__global__ void conv_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wpack, int Cin, int Cout,
                                         int ci_blk, int n_chunks, int CO_BLK, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co_l = idx % CO_BLK;
    int r = idx / CO_BLK;
    const int h = r % 2; r /= 2;
    const int tap = r % 9; r /= 9;
    const int cp = r % (ci_blk / 2); r /= (ci_blk / 2);
    const int chunk = r % n_chunks;
    const int cb = r / n_chunks;
    const int co = cb * CO_BLK + co_l, ci = chunk * ci_blk + 2 * cp + h;
    wpack[idx] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
}

template <int CI_BLK, int NTL>      // NTL = 32-channel N-tiles per wave (workgroup covers CO_BLK = 32*NTL channels)
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
    float* __restrict__ y, int Cin, int H, int W, int Cout, int Hp, int Wp, int tiles_x, int n_chunks) {
    constexpr int CO_BLK = 32 * NTL;
    constexpr int W_SLAB = (CI_BLK / 2) * 9 * 2 * CO_BLK;          // floats of packed weights per chunk
    __shared__ __attribute__((aligned(16))) float in_s[CI_BLK * M_CH];
    __shared__ __attribute__((aligned(16))) float w_s[W_SLAB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int cb = blockIdx.x;                                     // channel block (fastest: shares the input patch in L2)
    const int ty = blockIdx.y / tiles_x, tx = blockIdx.y - ty * tiles_x;
    const int b = blockIdx.z;
    const int y0 = ty * M_TR, x0 = tx * M_TC;                      // conv-output origin of the tile

    const int i = lane & 31, h = lane >> 5;
    const int dx = i & 1, dy = (i >> 1) & 1, pp = i >> 2;
    // per-lane operand bases (floats)
    const int a_base = h * M_CH + (wy * 4 + dy) * M_ST + wx * 16 + 2 * pp + dx;
    const int b_base = h * CO_BLK + i;

    f32x16 acc[2][NTL];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NTL; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const float* xb = x + (size_t)b * Cin * H * W;
    const float* wp_cb = wpack + (size_t)cb * n_chunks * W_SLAB;

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int ci0 = chunk * CI_BLK;
        __syncthreads();                                           // previous chunk's operand reads are done
        // ---- stage the input patch: CI_BLK x 10 x 34, zero outside the image / beyond Cin
        for (int idx = tid; idx < CI_BLK * M_IR * M_IC; idx += 256) {
            const int c = idx / (M_IR * M_IC);
            const int rem = idx - c * (M_IR * M_IC);
            const int r = rem / M_IC, col = rem - r * M_IC;
            const int iy = y0 - 1 + r, ix = x0 - 1 + col, ci = ci0 + c;
            float v = 0.f;
            if (ci < Cin && iy >= 0 && iy < H && ix >= 0 && ix < W) v = xb[((size_t)ci * H + iy) * W + ix];
            in_s[c * M_CH + r * M_ST + col] = v;
        }
        // ---- stage the packed weight slab of this chunk (contiguous copy)
        const float4* wsrc = reinterpret_cast<const float4*>(wp_cb + (size_t)chunk * W_SLAB);
        for (int idx = tid; idx < W_SLAB / 4; idx += 256) reinterpret_cast<float4*>(w_s)[idx] = wsrc[idx];
        __syncthreads();
        // ---- (CI_BLK/2) x 9 k-steps, 4 MFMAs each
#pragma unroll
        for (int cp = 0; cp < CI_BLK / 2; ++cp) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int aoff = cp * 2 * M_CH + ky * M_ST + kx;
                const int boff = (cp * 9 + tap) * 2 * CO_BLK;
                const float a0 = in_s[a_base + aoff];
                const float a1 = in_s[a_base + aoff + 2 * M_ST];
#pragma unroll
                for (int n = 0; n < NTL; ++n) {
                    const float bn = w_s[b_base + boff + 32 * n];
                    acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bn, acc[0][n], 0, 0, 0);
                    acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bn, acc[1][n], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue.  C/D layout: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5):
    // registers 4q..4q+3 of a lane are the 2x2 quad of pooled column pp = 2q + h.
#pragma unroll
    for (int n = 0; n < NTL; ++n) {
        const int co = cb * CO_BLK + n * 32 + i;
        if (co >= Cout) continue;
        const float bv = bias[co];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int py = (y0 >> 1) + wy * 2 + m;
            if (py >= Hp) continue;
            float* yrow = y + (((size_t)b * Cout + co) * Hp + py) * Wp;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int px = (x0 >> 1) + wx * 8 + 2 * q + h;
                const float v = fmaxf(fmaxf(acc[m][n][4 * q], acc[m][n][4 * q + 1]),
                                      fmaxf(acc[m][n][4 * q + 2], acc[m][n][4 * q + 3])) + bv;
                if (px < Wp) yrow[px] = fmaxf(v, 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// (2) general direct kernel
//   workgroup = 256 threads = an 8 x 32 tile of POOLED outputs of one image and one
//   group of CO_T output channels.  The 18 x 66 input patch of CI_T input channels at a
//   time is staged in LDS with coalesced row loads; every thread keeps the 2x2 pre-pool
//   quad of CO_T channels in registers, pools in registers and applies bias+ReLU after
//   the max (max and +bias/ReLU commute: both are monotone).  Weights are wave-uniform,
//   so the compiler reads them through the scalar cache.
// ------------------------------------------------------------------------------------------
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

int mfma_ci_blk(int Cin) { return Cin <= 4 ? 4 : 8; }
int mfma_co_blk(int Cout) { return Cout % 64 == 0 ? 64 : 32; }

template <int CI_BLK, int NTL>
void launch_mfma(dim3 grid, hipStream_t s, const float* x, const float* wpack, const float* bias, float* y, int Cin,
                 int H, int W, int Cout, int Hp, int Wp, int tiles_x, int n_chunks) {
    hipLaunchKernelGGL((conv3x3_mfma_kernel<CI_BLK, NTL>), grid, dim3(256), 0, s, x, wpack, bias, y, Cin, H, W, Cout,
                       Hp, Wp, tiles_x, n_chunks);
}

}  // namespace

extern "C" size_t i2l_conv_workspace_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cout % 32 != 0) return 0;
    const int cb = mfma_ci_blk(Cin), cob = mfma_co_blk(Cout);
    const int n_chunks = i2l_cdiv(Cin, cb), co_blocks = i2l_cdiv(Cout, cob);
    return i2l_align((size_t)co_blocks * n_chunks * (cb / 2) * 9 * 2 * cob * sizeof(float));
}

extern "C" int i2l_conv3x3_relu_pool2_fwd(const float* x, const float* w, const float* bias, float* y,
                                           int B, int Cin, int H, int W, int Cout, void* workspace,
                                           size_t workspace_bytes, i2l_stream_t stream) {
    if (!x || !w || !bias || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H < 2 || W < 2) return I2L_ERR_ARG;
    const int Hp = H / 2, Wp = W / 2;
    hipStream_t s = i2l_s(stream);
    if (Cout % 32 == 0) {
        const size_t need = i2l_conv_workspace_bytes(Cin, Cout);
        if (!workspace || workspace_bytes < need) return I2L_ERR_WORKSPACE;
        const int cbk = mfma_ci_blk(Cin), cob = mfma_co_blk(Cout);
        const int n_chunks = i2l_cdiv(Cin, cbk), co_blocks = i2l_cdiv(Cout, cob);
        const int total = co_blocks * n_chunks * (cbk / 2) * 9 * 2 * cob;
        float* wpack = static_cast<float*>(workspace);
        hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(i2l_cdiv(total, 256)), dim3(256), 0, s, w, wpack, Cin, Cout,
                           cbk, n_chunks, cob, total);
        I2L_CHECK_LAUNCH();
        // tiles cover the conv positions that feed a pooled output (floor pooling drops an odd last row/col)
        const int tiles_x = i2l_cdiv(2 * Wp, M_TC), tiles_y = i2l_cdiv(2 * Hp, M_TR);
        if ((long long)tiles_x * tiles_y > 65535 || B > 65535) return I2L_ERR_UNSUPPORTED;
        dim3 grid(co_blocks, tiles_x * tiles_y, B);
        if (cbk == 4 && cob == 32) launch_mfma<4, 1>(grid, s, x, wpack, bias, y, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks);
        else if (cbk == 4) launch_mfma<4, 2>(grid, s, x, wpack, bias, y, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks);
        else if (cob == 32) launch_mfma<8, 1>(grid, s, x, wpack, bias, y, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks);
        else launch_mfma<8, 2>(grid, s, x, wpack, bias, y, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks);
        I2L_CHECK_LAUNCH();
        return I2L_OK;
    }
    constexpr int CO_T = 8;
    const int co_groups = i2l_cdiv(Cout, CO_T);
    if (i2l_cdiv(Wp, TPW) > 65535 || i2l_cdiv(Hp, TPH) > 65535) return I2L_ERR_UNSUPPORTED;
    dim3 grid(B * co_groups, i2l_cdiv(Wp, TPW), i2l_cdiv(Hp, TPH));
    hipLaunchKernelGGL(conv3x3_relu_pool2_direct<CO_T>, grid, dim3(256), 0, s, x, w, bias, y, B, Cin, H, W, Cout,
                       Hp, Wp, co_groups);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
