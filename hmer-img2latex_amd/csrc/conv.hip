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

// Workgroup tile = (4/WX * 4) conv rows x (WX * 16) conv cols; the 4 waves sit WX across, 4/WX down.
// WX = 2: 8 x 32 (default), WX = 1: 16 x 16, WX = 4: 4 x 64 -- picked per launch to waste the fewest tiles.
template <int WX> struct Tile {
    static constexpr int TR = (4 / WX) * 4, TC = WX * 16;      // conv rows / cols per workgroup
    static constexpr int IR = TR + 2, IC = TC + 2;             // staged input rows / cols (1-pixel halo)
    static constexpr int ST = WX == 4 ? 80 : 48;               // LDS row stride (floats), == 16 mod 32, >= IC
    static constexpr int CH = IR * ST;                         // floats per staged channel
};

// wpack[co_blk][chunk][cp][tap][h][CO_BLK]: value = w[co_blk*CO_BLK + co][chunk*CI_BLK + 2*cp + h][tap], 0 outside.
// flip != 0 packs the weights of the DATA-GRADIENT convolution: w is the forward filter (Cin_f = Cout, Cout_f = Cin),
// value = w[ci][co][2-ky][2-kx].
__global__ void conv_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wpack, int Cin, int Cout,
                                         int ci_blk, int n_chunks, int CO_BLK, int total, int flip) {
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
    float v = 0.f;
    if (co < Cout && ci < Cin)
        v = flip ? w[((size_t)ci * Cout + co) * 9 + (8 - tap)] : w[((size_t)co * Cin + ci) * 9 + tap];
    wpack[idx] = v;
}

// POOL = true : y = maxpool2(relu(conv + bias)) (B,Cout,Hp,Wp), optional argmax (0..3 = 2*dy+dx, first max wins)
// POOL = false: y = conv (B,Cout,H,W), no bias / activation (the data-gradient convolution of the backward pass)
template <int CI_BLK, int NTL, bool POOL, int WX>      // NTL = 32-channel N-tiles per wave (workgroup covers CO_BLK = 32*NTL channels)
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
    float* __restrict__ y, unsigned char* __restrict__ amax, int Cin, int H, int W, int Cout, int Hp, int Wp,
    int tiles_x, int n_chunks) {
    constexpr int M_TR = Tile<WX>::TR, M_TC = Tile<WX>::TC, M_IR = Tile<WX>::IR, M_IC = Tile<WX>::IC;
    constexpr int M_ST = Tile<WX>::ST, M_CH = Tile<WX>::CH;
    constexpr int PR = M_TR / 2, PC = M_TC / 2;                    // pooled tile (PR * PC == 64)
    constexpr int CO_BLK = 32 * NTL;
    constexpr int W_SLAB = (CI_BLK / 2) * 9 * 2 * CO_BLK;          // floats of packed weights per chunk
    constexpr int OUT_S = CO_BLK * 65;                             // pooled output tile staged for coalesced stores
    constexpr int IN_S = CI_BLK * M_CH > OUT_S ? CI_BLK * M_CH : OUT_S;
    __shared__ __attribute__((aligned(16))) float in_s[IN_S];
    __shared__ __attribute__((aligned(16))) float w_s[W_SLAB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave / WX, wx = wave % WX;
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

    // Per-thread staging plan, computed once: element e of this thread goes to LDS offset loff[e] and comes
    // from global offset goff[e] inside the chunk's first channel plane (ok bit e: inside the image).
    constexpr int NE = (CI_BLK * M_IR * M_IC + 255) / 256;
    constexpr int NWV = (W_SLAB / 4 + 255) / 256;
    int loff[NE], goff[NE], cch[NE];
    unsigned okmask = 0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * 256;
        const int c = idx / (M_IR * M_IC);
        const int rem = idx - c * (M_IR * M_IC);
        const int r = rem / M_IC, col = rem - r * M_IC;
        const int iy = y0 - 1 + r, ix = x0 - 1 + col;
        loff[e] = c * M_CH + r * M_ST + col;
        goff[e] = (c * H + iy) * W + ix;
        cch[e] = c;
        if (idx < CI_BLK * M_IR * M_IC && iy >= 0 && iy < H && ix >= 0 && ix < W) okmask |= 1u << e;
        if (idx >= CI_BLK * M_IR * M_IC) loff[e] = -1;
    }
    float xin[NE];
    float4 win[NWV];
    auto fetch = [&](int chunk) {          // global -> registers (latency overlaps the MFMA loop of the previous chunk)
        const int ci0 = chunk * CI_BLK;
        const float* xc = xb + (size_t)ci0 * H * W;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            xin[e] = (((okmask >> e) & 1u) && ci0 + cch[e] < Cin) ? xc[goff[e]] : 0.f;
        const float4* wsrc = reinterpret_cast<const float4*>(wp_cb + (size_t)chunk * W_SLAB);
#pragma unroll
        for (int e = 0; e < NWV; ++e) {
            const int idx = tid + e * 256;
            win[e] = idx < W_SLAB / 4 ? wsrc[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {                  // registers -> LDS
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (loff[e] >= 0) in_s[loff[e]] = xin[e];
#pragma unroll
        for (int e = 0; e < NWV; ++e) {
            const int idx = tid + e * 256;
            if (idx < W_SLAB / 4) reinterpret_cast<float4*>(w_s)[idx] = win[e];
        }
    };

    fetch(0);
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        __syncthreads();                                           // previous chunk's operand reads are done
        commit();
        __syncthreads();
        if (chunk + 1 < n_chunks) fetch(chunk + 1);                // in flight during the MFMAs below
        // ---- (CI_BLK/2) x 9 k-steps, 2*NTL MFMAs each
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
    // Fast store path (inference, interior tiles): pooled tile -> LDS [channel][4 rows x 16 cols] -> 64-byte runs.
    const int py0 = y0 >> 1, px0 = x0 >> 1;
    if (POOL && amax == nullptr && (Wp & 3) == 0 && py0 + PR <= Hp && px0 + PC <= Wp && (cb + 1) * CO_BLK <= Cout) {
        __syncthreads();                                           // all operand reads of in_s are done
#pragma unroll
        for (int n = 0; n < NTL; ++n) {
            const float bv = bias[cb * CO_BLK + n * 32 + i];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = fmaxf(fmaxf(acc[m][n][4 * q], acc[m][n][4 * q + 1]),
                                          fmaxf(acc[m][n][4 * q + 2], acc[m][n][4 * q + 3])) + bv;
                    in_s[(n * 32 + i) * 65 + (wy * 2 + m) * PC + wx * 8 + 2 * q + h] = fmaxf(v, 0.f);
                }
        }
        __syncthreads();
        for (int idx = tid; idx < CO_BLK * 16; idx += 256) {       // (channel, row, 4-column group)
            constexpr int XG = PC / 4;
            const int x4 = idx % XG, py = (idx / XG) % PR, co_l = idx >> 4;
            const float* sp = &in_s[co_l * 65 + py * PC + 4 * x4];
            float4 v = make_float4(sp[0], sp[1], sp[2], sp[3]);
            *reinterpret_cast<float4*>(y + (((size_t)b * Cout + cb * CO_BLK + co_l) * Hp + py0 + py) * Wp + px0 + 4 * x4) = v;
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NTL; ++n) {
        const int co = cb * CO_BLK + n * 32 + i;
        if (co >= Cout) continue;
        if (POOL) {
            const float bv = bias[co];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int py = (y0 >> 1) + wy * 2 + m;
                if (py >= Hp) continue;
                const size_t rowoff = (((size_t)b * Cout + co) * Hp + py) * Wp;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int px = (x0 >> 1) + wx * 8 + 2 * q + h;
                    float best = acc[m][n][4 * q];
                    int bi = 0;
#pragma unroll
                    for (int e = 1; e < 4; ++e)
                        if (acc[m][n][4 * q + e] > best) { best = acc[m][n][4 * q + e]; bi = e; }
                    if (px < Wp) {
                        y[rowoff + px] = fmaxf(best + bv, 0.f);
                        if (amax) amax[rowoff + px] = (unsigned char)bi;
                    }
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ri = (r & 3) + 8 * (r >> 2) + 4 * h;          // MFMA row -> (pp, dy, dx)
                    const int yy = y0 + wy * 4 + m * 2 + ((ri >> 1) & 1);
                    const int xx = x0 + wx * 16 + 2 * (ri >> 2) + (ri & 1);
                    if (yy < H && xx < W) y[(((size_t)b * Cout + co) * H + yy) * W + xx] = acc[m][n][r];
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

// flip != 0: data-gradient convolution, weight element = w[ci][co][2-ky][2-kx] (w is the forward filter).
template <int CO_T, bool POOL>
__global__ __launch_bounds__(256) void conv3x3_relu_pool2_direct(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, unsigned char* __restrict__ amax, int B, int Cin, int H, int W, int Cout, int Hp, int Wp,
    int co_groups, int flip) {
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
                    const float* wp = flip ? w + ((size_t)(ci0 + ci) * Cout + (co0 + c)) * 9
                                           : w + ((size_t)(co0 + c) * Cin + (ci0 + ci)) * 9;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float wv = flip ? wp[8 - (ky * 3 + kx)] : wp[ky * 3 + kx];
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
    if (POOL) {
        if (PY < Hp && PX < Wp) {
#pragma unroll
            for (int c = 0; c < CO_T; ++c) {
                if (co0 + c < Cout) {
                    float best = acc[c][0];
                    int bi = 0;
#pragma unroll
                    for (int e = 1; e < 4; ++e)
                        if (acc[c][e] > best) { best = acc[c][e]; bi = e; }
                    const size_t o = (((size_t)b * Cout + co0 + c) * Hp + PY) * Wp + PX;
                    y[o] = fmaxf(best + bias[co0 + c], 0.f);
                    if (amax) amax[o] = (unsigned char)bi;
                }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            if (co0 + c < Cout) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int yy = 2 * PY + (e >> 1), xx = 2 * PX + (e & 1);
                    if (yy < H && xx < W) y[(((size_t)b * Cout + co0 + c) * H + yy) * W + xx] = acc[c][e];
                }
            }
        }
    }
}

int mfma_ci_blk(int Cin) { return Cin <= 4 ? 4 : 8; }
int mfma_co_blk(int Cout) { return Cout % 64 == 0 ? 64 : 32; }

template <int CI_BLK, int NTL, int WX>
void launch_mfma_wx(bool pool, dim3 grid, hipStream_t s, const float* x, const float* wpack, const float* bias, float* y,
                    unsigned char* amax, int Cin, int H, int W, int Cout, int Hp, int Wp, int tiles_x, int n_chunks) {
    if (pool)
        hipLaunchKernelGGL((conv3x3_mfma_kernel<CI_BLK, NTL, true, WX>), grid, dim3(256), 0, s, x, wpack, bias, y, amax,
                           Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks);
    else
        hipLaunchKernelGGL((conv3x3_mfma_kernel<CI_BLK, NTL, false, WX>), grid, dim3(256), 0, s, x, wpack, bias, y, amax,
                           Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks);
}

// tile shape with the fewest wasted positions for a rows x cols output plane (ties: the default 8 x 32)
int pick_wx(int rows, int cols) {
    int best = 2;
    long bestw = -1;
    const int cand[3] = {2, 1, 4};
    for (int c = 0; c < 3; ++c) {
        const int wx = cand[c], tr = (4 / wx) * 4, tc = wx * 16;
        const long cover = (long)i2l_cdiv(rows, tr) * tr * i2l_cdiv(cols, tc) * tc;
        if (bestw < 0 || cover < bestw) { bestw = cover; best = wx; }
    }
    return best;
}

template <int CI_BLK, int NTL>
int launch_mfma(bool pool, int co_blocks, int B, hipStream_t s, const float* x, const float* wpack, const float* bias,
                float* y, unsigned char* amax, int Cin, int H, int W, int Cout, int Hp, int Wp, int n_chunks) {
    // pooled: tiles cover the conv positions that feed a pooled output (floor pooling drops an odd last
    // row/col); plain: every position
    const int rows = pool ? 2 * Hp : H, cols = pool ? 2 * Wp : W;
    const int wx = pick_wx(rows, cols);
    const int tr = (4 / wx) * 4, tc = wx * 16;
    const int tiles_x = i2l_cdiv(cols, tc), tiles_y = i2l_cdiv(rows, tr);
    if ((long long)tiles_x * tiles_y > 65535 || B > 65535) return I2L_ERR_UNSUPPORTED;
    dim3 grid(co_blocks, tiles_x * tiles_y, B);
    if (wx == 2) launch_mfma_wx<CI_BLK, NTL, 2>(pool, grid, s, x, wpack, bias, y, amax, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks);
    else if (wx == 1) launch_mfma_wx<CI_BLK, NTL, 1>(pool, grid, s, x, wpack, bias, y, amax, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks);
    else launch_mfma_wx<CI_BLK, NTL, 4>(pool, grid, s, x, wpack, bias, y, amax, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks);
    return I2L_OK;
}

// One 3x3 / pad 1 convolution launch.  pool: fused bias+ReLU+maxpool2 (+argmax); !pool: plain full-resolution
// conv with the flipped/transposed filter (data gradient).  `wpack_ws` must hold i2l_conv_workspace_bytes().
int run_conv(bool pool, const float* x, const float* w, const float* bias, float* y, unsigned char* amax, int B,
             int Cin, int H, int W, int Cout, void* wpack_ws, size_t ws_bytes, hipStream_t s, bool exact);

}  // namespace

extern "C" size_t i2l_conv_workspace_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cout % 32 != 0) return 0;
    const int cb = mfma_ci_blk(Cin), cob = mfma_co_blk(Cout);
    const int n_chunks = i2l_cdiv(Cin, cb), co_blocks = i2l_cdiv(Cout, cob);
    size_t need = i2l_align((size_t)co_blocks * n_chunks * (cb / 2) * 9 * 2 * cob * sizeof(float));
    if (i2l_conv_bf16x3_full_applicable(Cin, Cout)) {
        size_t n3 = i2l_conv_bf16x3_workspace_bytes(Cin, Cout);
        if (i2l_conv_bf16x3_applicable(Cin, Cout)) n3 += i2l_conv_fixlist_bytes();     // training forward: near-decision list
        if (n3 > need) need = n3;
    }
    if (i2l_conv_smallk_applicable(Cin, Cout) && i2l_conv_fixlist_bytes() > need) need = i2l_conv_fixlist_bytes();
    return need;
}

namespace {

int run_conv(bool pool, const float* x, const float* w, const float* bias, float* y, unsigned char* amax, int B,
             int Cin, int H, int W, int Cout, void* workspace, size_t workspace_bytes, hipStream_t s, bool exact) {
    const int Hp = H / 2, Wp = W / 2;
    // data gradient on the split-bf16 matrix-core kernel (fp32-grade, like the weight-gradient GEMMs) where the
    // channel counts and even H, W allow
    if (!exact && !pool && !bias && !amax && ((H | W) & 1) == 0 && i2l_conv_bf16x3_full_applicable(Cin, Cout)) {
        if (!workspace || workspace_bytes < i2l_conv_workspace_bytes(Cin, Cout)) return I2L_ERR_WORKSPACE;
        return i2l_conv_bf16x3_run(x, w, nullptr, y, nullptr, B, Cin, H, W, Cout, workspace, workspace_bytes, s, 1);
    }
    if (Cout % 32 == 0) {
        const size_t need = i2l_conv_workspace_bytes(Cin, Cout);
        if (!workspace || workspace_bytes < need) return I2L_ERR_WORKSPACE;
        const int cbk = mfma_ci_blk(Cin), cob = mfma_co_blk(Cout);
        const int n_chunks = i2l_cdiv(Cin, cbk), co_blocks = i2l_cdiv(Cout, cob);
        const int total = co_blocks * n_chunks * (cbk / 2) * 9 * 2 * cob;
        float* wpack = static_cast<float*>(workspace);
        hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(i2l_cdiv(total, 256)), dim3(256), 0, s, w, wpack, Cin, Cout,
                           cbk, n_chunks, cob, total, pool ? 0 : 1);
        I2L_CHECK_LAUNCH();
        int rc;
        if (cbk == 4 && cob == 32) rc = launch_mfma<4, 1>(pool, co_blocks, B, s, x, wpack, bias, y, amax, Cin, H, W, Cout, Hp, Wp, n_chunks);
        else if (cbk == 4) rc = launch_mfma<4, 2>(pool, co_blocks, B, s, x, wpack, bias, y, amax, Cin, H, W, Cout, Hp, Wp, n_chunks);
        else if (cob == 32) rc = launch_mfma<8, 1>(pool, co_blocks, B, s, x, wpack, bias, y, amax, Cin, H, W, Cout, Hp, Wp, n_chunks);
        else rc = launch_mfma<8, 2>(pool, co_blocks, B, s, x, wpack, bias, y, amax, Cin, H, W, Cout, Hp, Wp, n_chunks);
        if (rc != I2L_OK) return rc;
        I2L_CHECK_LAUNCH();
        return I2L_OK;
    }
    constexpr int CO_T = 8;
    const int co_groups = i2l_cdiv(Cout, CO_T);
    // the direct kernel walks 2x2 quads: ceil so that an odd last row/col is produced in the plain mode
    const int qh = pool ? Hp : i2l_cdiv(H, 2), qw = pool ? Wp : i2l_cdiv(W, 2);
    if (i2l_cdiv(qw, TPW) > 65535 || i2l_cdiv(qh, TPH) > 65535) return I2L_ERR_UNSUPPORTED;
    dim3 grid(B * co_groups, i2l_cdiv(qw, TPW), i2l_cdiv(qh, TPH));
    if (pool)
        hipLaunchKernelGGL((conv3x3_relu_pool2_direct<CO_T, true>), grid, dim3(256), 0, s, x, w, bias, y, amax, B, Cin,
                           H, W, Cout, Hp, Wp, co_groups, 0);
    else
        hipLaunchKernelGGL((conv3x3_relu_pool2_direct<CO_T, false>), grid, dim3(256), 0, s, x, w, bias, y, amax, B, Cin,
                           H, W, Cout, Hp, Wp, co_groups, 1);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

}  // namespace

extern "C" int i2l_conv3x3_relu_pool2_fwd(const float* x, const float* w, const float* bias, float* y,
                                           unsigned char* argmax_out, int B, int Cin, int H, int W, int Cout,
                                           void* workspace, size_t workspace_bytes, int flags,
                                           i2l_stream_t stream) {
    if (!x || !w || !bias || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H < 2 || W < 2) return I2L_ERR_ARG;
    const bool exact = (flags & I2L_FLAG_EXACT_FP32) != 0;
    // split-bf16 matrix-core kernels where the channel counts allow, for inference and for the training forward
    // (argmax_out != NULL).  The conv gradients are non-smooth in the forward values (ReLU gate, pooling arg max), so
    // there the kernels list every window whose decision is closer than 2^-13 and a fix-up pass re-evaluates those in
    // fp32 (conv_bf16x3.hip): the discrete decisions are an fp32 computation's, the rest differs by ~1e-7.
    const bool split = !exact;
    if (split && i2l_conv_smallk_applicable(Cin, Cout))
        return i2l_conv_smallk_run(x, w, bias, y, argmax_out, B, Cin, H, W, Cout, workspace, workspace_bytes, i2l_s(stream));
    if (split && i2l_conv_bf16x3_applicable(Cin, Cout))
        return i2l_conv_bf16x3_run(x, w, bias, y, argmax_out, B, Cin, H, W, Cout, workspace, workspace_bytes, i2l_s(stream), 0,
                                   (flags & I2L_FLAG_WEIGHTS_PACKED) ? 1 : 0);
    return run_conv(true, x, w, bias, y, argmax_out, B, Cin, H, W, Cout, workspace, workspace_bytes, i2l_s(stream), exact);
}

// ------------------------------------------------------------------------------------------
// Backward of one block
// ------------------------------------------------------------------------------------------
namespace {

// dyp (B,C,H,W) = gradient w.r.t. the conv output: the pooled gradient goes to the window's argmax
// position if the pooled (post-ReLU) output is positive, everything else is 0 (ReLU' and MaxPool').
__global__ __launch_bounds__(256) void unpool_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              const unsigned char* __restrict__ amax,
                                                              float* __restrict__ dyp, size_t planes, int H, int W,
                                                              int Hp, int Wp) {
    const size_t total = planes * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pl = i / ((size_t)H * W);
        const int rem = (int)(i - pl * H * W);
        const int yy = rem / W, xx = rem - yy * W;
        const int py = yy >> 1, px = xx >> 1;
        float v = 0.f;
        if (py < Hp && px < Wp) {
            const size_t o = (pl * Hp + py) * Wp + px;
            if (y[o] > 0.f && (int)amax[o] == 2 * (yy & 1) + (xx & 1)) v = dy[o];
        }
        dyp[i] = v;
    }
}

// same, four output columns (two pooling windows) per thread and 16-byte stores; needs W % 4 == 0
__global__ __launch_bounds__(256) void unpool_relu_bwd4_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               const unsigned char* __restrict__ amax,
                                                               float* __restrict__ dyp, size_t planes, int H, int W,
                                                               int Hp, int Wp) {
    const int W4 = W >> 2;
    const size_t total = planes * H * W4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t prow = i / W4;                    // (plane, yy)
        const int x4 = (int)(i - prow * W4);
        const size_t pl = prow / H;
        const int yy = (int)(prow - pl * H);
        const int py = yy >> 1, ry = yy & 1;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (py < Hp) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int px = 2 * x4 + e;
                if (px < Wp) {
                    const size_t o = (pl * Hp + py) * Wp + px;
                    const int am = (int)amax[o];
                    if (y[o] > 0.f && (am >> 1) == ry) v[2 * e + (am & 1)] = dy[o];
                }
            }
        }
        *reinterpret_cast<float4*>(dyp + (prow * W + 4 * (size_t)x4)) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// colT[b][(ci*9+tap)][pos] = x[b][ci][y+ky-1][x+kx-1] (0 outside) for a chunk of images
__global__ __launch_bounds__(256) void im2col_t_kernel(const float* __restrict__ x, float* __restrict__ colT, int nb,
                                                       int Cin, int H, int W) {
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)nb * Cin * 9 * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / HW;                     // (b, ci, tap)
        const int pos = (int)(i - row * HW);
        const int tap = (int)(row % 9);
        const size_t bc = row / 9;                     // b*Cin + ci
        const int yy = pos / W + tap / 3 - 1, xx = pos % W + tap % 3 - 1;
        colT[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[(bc * H + yy) * W + xx] : 0.f;
    }
}

// db[c] = sum_{b,pos} dyp[b][c][pos]: grid (C, B) partial sums per image, then an ordered sum over images
// (deterministic), taken from the POOLED tensors: the un-pooled gradient holds, per 2x2 window, dy where y > 0 and zeros
// elsewhere, so db[c] = sum over pooled positions of (y > 0 ? dy : 0) -- a quarter of the bytes
__global__ __launch_bounds__(256) void plane_sum_pooled_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               double* __restrict__ partial, int B, int C, size_t HWp) {
    __shared__ double red[256];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* pd = dy + ((size_t)b * C + c) * HWp;
    const float* py = y + ((size_t)b * C + c) * HWp;
    double s = 0.0;
    for (size_t i = threadIdx.x; i < HWp; i += 256) s += py[i] > 0.f ? pd[i] : 0.f;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)c * B + b] = red[0];
}
__global__ void plane_sum_final_kernel(const double* __restrict__ partial, float* __restrict__ db, int B, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += partial[(size_t)c * B + b];
    db[c] = (float)s;
}


// ---- weight + bias gradient of the FIRST block (no data gradient wanted, Cin <= 3): the sparse form --------------------
// Max-pooling sends each pooled gradient to ONE of its four positions, so dW[co][ci][ky][kx] needs one product per pooled
// cell, channel and tap -- a quarter of what the implicit-im2col GEMM over the un-pooled gradient multiplies, and at
// Cout x 27 outputs that GEMM wastes most of its 128 x 64 tile besides.  Exact fp32 on the vector ALUs:
//   workgroup = (image, band of FB_PB pooled rows, block of 32 output channels); the band's input rows (+ halo, zero padded)
//   and its masked gradients / arg-max codes sit in LDS; thread (co = tid & 31, grp = tid >> 5) walks cells grp, grp + 8, ..
//   with its Cin x 9 sums in registers (the four arg-max variants of a cell's window land in four different banks, the
//   gradient rows have an odd stride); the 8 groups are summed in a fixed order and every workgroup writes its partial
//   [32][Cin*9 + 1] (last = bias); conv_wgrad_first_reduce_kernel adds the partials in double, in a fixed order.
constexpr int FB_PB = 1;
template <int CIN>
__global__ __launch_bounds__(256) void conv_wgrad_first_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ y,
                                                               const unsigned char* __restrict__ amax,
                                                               float* __restrict__ partial, int H, int W, int Cout, int Hp,
                                                               int Wp, int bands, int fast) {
    constexpr int K9 = CIN * 9, XR = 2 * FB_PB + 2, NO = K9 + 1;
    extern __shared__ __attribute__((aligned(16))) float fb_sm[];
    const int XW = W + 2, NC = FB_PB * Wp, GS = NC | 1;
    float* x_s = fb_sm;                                         // [CIN][XR][XW], column 0 = image column -1
    float* g_s = x_s + CIN * XR * XW;                           // [32][GS] masked pooled gradient; later the group sums
    unsigned char* a_s = reinterpret_cast<unsigned char*>(g_s + (32 * GS > 8 * 32 * NO ? 32 * GS : 8 * 32 * NO));
    const int tid = threadIdx.x;
    const int b = blockIdx.x / bands, band = blockIdx.x - b * bands, cb = blockIdx.y;
    const int p0 = band * FB_PB, r0 = 2 * p0 - 1;
    // staging.  Fast path (W % 8 == 0, W <= 512, 16-byte aligned tensors; FB_PB == 1): whole rows as 16-byte pieces, every
    // global load of the workgroup issued before the first LDS store -- thread (row = tid >> 3, seg = tid & 7) takes pieces
    // seg, seg + 8, .. of gradient row `row` (y, dy: float4, arg max: 4 codes), thread (xr = tid >> 4, xs = tid & 15) pieces
    // xs, xs + 16, .. of window row xr; all loads unconditional from clamped addresses, masks applied at the store
    const int n_x = CIN * XR * XW, n_g = 32 * NC;
    if (fast) {
        constexpr int KG = 8, KX = 8;
        const int grow = tid >> 3, seg = tid & 7, nq = Wp >> 2;
        const int cog = cb * 32 + grow;
        const bool gvalid = p0 < Hp && cog < Cout;
        const size_t orow = (((size_t)b * Cout + min(cog, Cout - 1)) * Hp + min(p0, Hp - 1)) * Wp;
        const float4* y4 = reinterpret_cast<const float4*>(y + orow);
        const float4* d4 = reinterpret_cast<const float4*>(dy + orow);
        const unsigned* a4 = reinterpret_cast<const unsigned*>(amax + orow);
        const int xr = tid >> 4, xs = tid & 15, nxq = W >> 2;
        const int xci = min(xr / XR, CIN - 1), xrr = xr - (xr / XR) * XR, Y = r0 + xrr;
        const bool xvalid = xr < CIN * XR && (unsigned)Y < (unsigned)H;
        const float4* x4 = reinterpret_cast<const float4*>(x + (((size_t)b * CIN + xci) * H + min(max(Y, 0), H - 1)) * W);
        float4 vy[KG], vd[KG], vx[KX];
        unsigned va[KG];
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            const int q = min(seg + 8 * k, nq - 1);
            vy[k] = y4[q];
            vd[k] = d4[q];
            va[k] = a4[q];
        }
#pragma unroll
        for (int k = 0; k < KX; ++k) vx[k] = x4[min(xs + 16 * k, nxq - 1)];
        if (xr < CIN * XR) {
            float* xrow = x_s + xr * XW;
            if (xs == 0) xrow[0] = 0.f;
            if (xs == 1) xrow[W + 1] = 0.f;
#pragma unroll
            for (int k = 0; k < KX; ++k) {
                const int q = xs + 16 * k;
                if (q < nxq) {
                    xrow[1 + 4 * q] = xvalid ? vx[k].x : 0.f;
                    xrow[2 + 4 * q] = xvalid ? vx[k].y : 0.f;
                    xrow[3 + 4 * q] = xvalid ? vx[k].z : 0.f;
                    xrow[4 + 4 * q] = xvalid ? vx[k].w : 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            const int q = seg + 8 * k;
            if (q < nq) {
                float* gr = g_s + grow * GS + 4 * q;
                unsigned char* ar = a_s + grow * GS + 4 * q;
                gr[0] = gvalid && vy[k].x > 0.f ? vd[k].x : 0.f;
                gr[1] = gvalid && vy[k].y > 0.f ? vd[k].y : 0.f;
                gr[2] = gvalid && vy[k].z > 0.f ? vd[k].z : 0.f;
                gr[3] = gvalid && vy[k].w > 0.f ? vd[k].w : 0.f;
                ar[0] = gvalid ? (unsigned char)(va[k] & 0xFF) : (unsigned char)0;
                ar[1] = gvalid ? (unsigned char)((va[k] >> 8) & 0xFF) : (unsigned char)0;
                ar[2] = gvalid ? (unsigned char)((va[k] >> 16) & 0xFF) : (unsigned char)0;
                ar[3] = gvalid ? (unsigned char)(va[k] >> 24) : (unsigned char)0;
            }
        }
    } else {
        // any other shape: element by element, batches of 8 loads per thread
        for (int base0 = 0; base0 < n_x; base0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base0 + u * 256 + tid;
                const int ci = idx / (XR * XW), rem = idx - ci * XR * XW;
                const int rr = rem / XW, cc = rem - rr * XW;
                const int Y = r0 + rr, X = cc - 1;
                const bool in = idx < n_x && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
                const int cic = ci < CIN ? ci : CIN - 1, Yc = min(max(Y, 0), H - 1), Xc = min(max(X, 0), W - 1);
                const float t = x[(((size_t)b * CIN + cic) * H + Yc) * W + Xc];
                v[u] = in ? t : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (base0 + u * 256 + tid < n_x) x_s[base0 + u * 256 + tid] = v[u];
        }
        for (int base0 = 0; base0 < n_g; base0 += 8 * 256) {
            float gy[8], gd[8];
            unsigned char ga[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base0 + u * 256 + tid;
                const int co = idx / NC, c = idx - co * NC;
                const int pr = c / Wp, px = c - pr * Wp, P = p0 + pr, cog = cb * 32 + co;
                const bool in = idx < n_g && P < Hp && cog < Cout;
                const size_t o = (((size_t)b * Cout + min(cog, Cout - 1)) * Hp + min(P, Hp - 1)) * Wp + px;   // always valid
                const float ty = y[o], td = dy[o];
                const unsigned char ta = amax[o];
                gy[u] = in && ty > 0.f ? td : 0.f;
                gd[u] = 0.f;
                ga[u] = in ? ta : (unsigned char)0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base0 + u * 256 + tid;
                if (idx < n_g) {
                    const int co = idx / NC, c = idx - co * NC;
                    g_s[co * GS + c] = gy[u];
                    a_s[co * GS + c] = ga[u];
                }
            }
        }
    }
    __syncthreads();
    const int co = tid & 31, grp = tid >> 5;
    float acc[K9], accb = 0.f;
#pragma unroll
    for (int j = 0; j < K9; ++j) acc[j] = 0.f;
    int pr = 0, px = grp;
    while (px >= Wp) { px -= Wp; ++pr; }
    for (int c = grp; c < NC; c += 8) {
        const float g = g_s[co * GS + c];
        const int a = a_s[co * GS + c];
        const float* xw = x_s + (2 * pr + (a >> 1)) * XW + 2 * px + (a & 1);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc[ci * 9 + ky * 3 + kx] = fmaf(g, xw[(ci * XR + ky) * XW + kx], acc[ci * 9 + ky * 3 + kx]);
        accb += g;
        px += 8;
        while (px >= Wp) { px -= Wp; ++pr; }
    }
    __syncthreads();                                            // everyone is done with g_s: it becomes the group sums
    float* red = g_s;
#pragma unroll
    for (int j = 0; j < K9; ++j) red[(grp * 32 + co) * NO + j] = acc[j];
    red[(grp * 32 + co) * NO + K9] = accb;
    __syncthreads();
    float* out = partial + ((size_t)blockIdx.x * gridDim.y + cb) * (32 * NO);
    for (int o = tid; o < 32 * NO; o += 256) {
        float sum = 0.f;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) sum += red[g8 * 32 * NO + o];
        out[o] = sum;
    }
}

// dw[co][j] (j < K9) and db[co] (j == K9) = sum over the n_wg partials, 8 outputs x 32 slices per workgroup, in double
__global__ __launch_bounds__(256) void conv_wgrad_first_reduce_kernel(const float* __restrict__ partial, int n_wg, int co_blocks,
                                                                      int K9, int Cout, float* __restrict__ dw,
                                                                      float* __restrict__ db) {
    __shared__ double red[32][9];
    const int NO = K9 + 1, per_cb = 32 * NO, n_out = co_blocks * per_cb;
    const int ol = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int o = blockIdx.x * 8 + ol;
    const int cb = o < n_out ? o / per_cb : 0, r = o - cb * per_cb;
    double s = 0.0;
    if (o < n_out) {
        const float* src = partial + (size_t)cb * per_cb + r;
        const size_t step = (size_t)co_blocks * per_cb;
        int wgi = sl;
        for (; wgi + 7 * 32 < n_wg; wgi += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(wgi + 32 * u) * step];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; wgi < n_wg; wgi += 32) s += (double)src[(size_t)wgi * step];
    }
    red[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && o < n_out) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][ol];
        const int co = cb * 32 + r / NO, j = r - (r / NO) * NO;
        if (co < Cout) {
            if (j < K9) dw[(size_t)co * K9 + j] = (float)t;
            else db[co] = (float)t;
        }
    }
}

constexpr int WG_CHUNK = 32;    // images per weight-gradient GEMM call

struct BwdLayout { size_t dyp, colT, gemm, wpack, psum, first, total; size_t gemm_bytes, wpack_bytes, first_bytes; int chunk; };

// conv_wgrad_first_kernel: LDS bytes for this width, 0 when the shape is not its business
size_t wgrad_first_lds(int Cin, int W, int Cout) {
    if (Cin < 1 || Cin > 3 || Cout % 32 != 0 || W < 2) return 0;
    const int Wp = W / 2, NC = FB_PB * Wp, GS = NC | 1, NO = Cin * 9 + 1;
    const size_t gs = (size_t)(32 * GS > 8 * 32 * NO ? 32 * GS : 8 * 32 * NO);
    const size_t b = ((size_t)Cin * (2 * FB_PB + 2) * (W + 2) + gs) * sizeof(float) + (size_t)32 * GS;
    return b <= 80 * 1024 ? b : 0;
}

BwdLayout bwd_layout(int B, int Cin, int H, int W, int Cout) {
    BwdLayout o{};
    const size_t HW = (size_t)H * W;
    o.chunk = B < WG_CHUNK ? B : WG_CHUNK;
    size_t off = 0;
    o.dyp = off; off += i2l_align((size_t)B * Cout * HW * sizeof(float));
    o.colT = off; off += i2l_align((size_t)o.chunk * Cin * 9 * HW * sizeof(float));
    o.gemm_bytes = i2l_gemm_workspace_bytes(Cout, Cin * 9, (int)HW, o.chunk);
    {   // the whole batch in one implicit-im2col GEMM (split-bf16 kernel)
        const size_t all = i2l_gemm_workspace_bytes(Cout, Cin * 9, (int)HW, B);
        if (all > o.gemm_bytes) o.gemm_bytes = all;
    }
    o.gemm = off; off += i2l_align(o.gemm_bytes);
    o.wpack_bytes = i2l_conv_workspace_bytes(Cout, Cin);         // data-gradient conv: Cout -> Cin channels
    o.wpack = off; off += i2l_align(o.wpack_bytes);
    o.psum = off; off += i2l_align((size_t)B * Cout * sizeof(double));
    if (wgrad_first_lds(Cin, W, Cout)) {         // partial sums of the first block's sparse weight-gradient kernel
        o.first_bytes = (size_t)B * i2l_cdiv(H / 2, FB_PB) * (Cout / 32) * 32 * (Cin * 9 + 1) * sizeof(float);
        o.first = off; off += i2l_align(o.first_bytes);
    }
    o.total = off;
    return o;
}

}  // namespace

extern "C" size_t i2l_conv_bwd_workspace_bytes(int B, int Cin, int H, int W, int Cout) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H < 2 || W < 2) return 0;
    return bwd_layout(B, Cin, H, W, Cout).total;
}

extern "C" int i2l_conv3x3_relu_pool2_bwd(const float* x, const float* w, const float* y, const uint8_t* argmax,
                                           const float* dy, float* dx, float* dw, float* db, int B, int Cin, int H,
                                           int W, int Cout, void* workspace, size_t workspace_bytes, int flags,
                                           i2l_lanes* lanes, i2l_stream_t stream) {
    if (!x || !w || !y || !argmax || !dy || !dw || !db || B <= 0 || Cin <= 0 || Cout <= 0 || H < 2 || W < 2)
        return I2L_ERR_ARG;
    const bool exact = (flags & I2L_FLAG_EXACT_FP32) != 0;
    const int split = exact ? 0 : 1;
    const BwdLayout lo = bwd_layout(B, Cin, H, W, Cout);
    if (!workspace || workspace_bytes < lo.total) return I2L_ERR_WORKSPACE;
    char* base = static_cast<char*>(workspace);
    float* dyp = reinterpret_cast<float*>(base + lo.dyp);
    float* colT = reinterpret_cast<float*>(base + lo.colT);
    hipStream_t s = i2l_s(stream);
    const int Hp = H / 2, Wp = W / 2;
    const size_t HW = (size_t)H * W;
    // Block 0 needs no data gradient (dx == null): its weight-gradient GEMM un-pools dy on the fly (GemmArgs::pool_y),
    // so the un-pooled gradient -- 168 MB at 64 x 32 x 64 x 320 -- is never written
    GemmArgs gp = gemm_args();
    gp.A = dy; gp.lda = (long)Hp * Wp; gp.bsa = (long)Cout * Hp * Wp;
    gp.pool_y = y; gp.pool_am = argmax;
    gp.W = x; gp.ldw = (long)HW; gp.bsw = (long)(Cin * HW);
    gp.conv_h = H; gp.conv_w = W;
    gp.nz = B;
    gp.split_bf16 = split;
    gp.C = dw; gp.ldc = Cin * 9;
    gp.M = Cout; gp.N = Cin * 9; gp.K = (int)HW;
    const size_t first_lds = (dx || (flags & I2L_FLAG_CONV_NO_SPARSE_WGRAD) || Hp < 1 || Wp < 1 || !lo.first_bytes)
                                 ? 0 : wgrad_first_lds(Cin, W, Cout);      // the sparse first-block kernel reads dy itself
    const bool fused_unpool = !dx && (H % 2) == 0 && (W % 2) == 0 && i2l_gemm_split_bf16_ok(gp);
    if (!fused_unpool && !first_lds) {
        const size_t total = (size_t)B * Cout * HW;
        size_t blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        if (W % 4 == 0 && reinterpret_cast<uintptr_t>(dyp) % 16 == 0) {
            size_t b4 = (total / 4 + 255) / 256;
            if (b4 > 16384) b4 = 16384;
            hipLaunchKernelGGL(unpool_relu_bwd4_kernel, dim3((unsigned)b4), dim3(256), 0, s, dy, y, argmax, dyp,
                               (size_t)B * Cout, H, W, Hp, Wp);
        } else {
            hipLaunchKernelGGL(unpool_relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, y, argmax, dyp,
                               (size_t)B * Cout, H, W, Hp, Wp);
        }
        I2L_CHECK_LAUNCH();
    }
    if (B > 65535) return I2L_ERR_UNSUPPORTED;
    // lanes != NULL: bias sums and the weight-gradient GEMM(s) leave the caller's stream here (their inputs -- dy,
    // y, x and the un-pooled gradient -- are complete on it); the data-gradient conv below stays on it.  The two use
    // disjoint workspace regions (psum / colT / gemm against wpack).
    hipStream_t sd = s;
    if (lanes && dx) {                                   // the first block (no dx) has nothing to run beside: it stays here
        hipStream_t f = i2l_side_fork(lanes, s, 1);
        if (f) sd = f;
    }
    const hipStream_t s_main = s;
    s = sd;
    if (first_lds) {
        const int bands = i2l_cdiv(Hp, FB_PB), cbs = Cout / 32;
        auto al = [](const void* q, uintptr_t a) { return reinterpret_cast<uintptr_t>(q) % a == 0; };
        const int fast = (FB_PB == 1 && W % 8 == 0 && W <= 512 && al(x, 16) && al(y, 16) && al(dy, 16) && al(argmax, 4)) ? 1 : 0;
        float* part = reinterpret_cast<float*>(base + lo.first);
        dim3 grid((unsigned)(B * bands), (unsigned)cbs);
#define I2L_FIRST(CI)                                                                                                  \
        do {                                                                                                           \
            static std::atomic<unsigned> attr_done{0};           /* per device */                         \
            if (!i2l_lds_attr(reinterpret_cast<const void*>(conv_wgrad_first_kernel<CI>), 80 * 1024, attr_done))       \
                return I2L_ERR_LAUNCH;                                                                                 \
            hipLaunchKernelGGL(conv_wgrad_first_kernel<CI>, grid, dim3(256), first_lds, s, x, dy, y, argmax, part, H, W, \
                               Cout, Hp, Wp, bands, fast);                                                                 \
        } while (0)
        if (Cin == 1) I2L_FIRST(1);
        else if (Cin == 2) I2L_FIRST(2);
        else I2L_FIRST(3);
#undef I2L_FIRST
        I2L_CHECK_LAUNCH();
        const int n_out = cbs * 32 * (Cin * 9 + 1);
        hipLaunchKernelGGL(conv_wgrad_first_reduce_kernel, dim3(i2l_cdiv(n_out, 8)), dim3(256), 0, s, (const float*)part,
                           B * bands, cbs, Cin * 9, Cout, dw, db);
        I2L_CHECK_LAUNCH();
        return I2L_OK;
    }
    double* psum = reinterpret_cast<double*>(base + lo.psum);
    hipLaunchKernelGGL(plane_sum_pooled_kernel, dim3(Cout, B), dim3(256), 0, s, dy, y, psum, B, Cout, (size_t)Hp * Wp);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(plane_sum_final_kernel, dim3(i2l_cdiv(Cout, 64)), dim3(64), 0, s, (const double*)psum, db, B, Cout);
    I2L_CHECK_LAUNCH();
    // weight gradient: dw[co][(ci,tap)] = sum_b sum_pos dyp[b][co][pos] * colT[b][(ci,tap)][pos]
    bool wgrad_done = false;
    if (fused_unpool) {
        const int rc = i2l_gemm(gp, base + lo.gemm, lo.gemm_bytes, s);
        if (rc != I2L_OK) return rc;
        wgrad_done = true;
    }
    if (!wgrad_done) {   // one GEMM over the whole batch, the im2col^T operand gathered inside the kernel (no colT image)
        GemmArgs g = gemm_args();
        g.A = dyp; g.lda = (long)HW; g.bsa = (long)(Cout * HW);
        g.W = x; g.ldw = (long)HW; g.bsw = (long)(Cin * HW);
        g.conv_h = H; g.conv_w = W;
        g.nz = B;
        g.split_bf16 = split;
        g.C = dw; g.ldc = Cin * 9;
        g.M = Cout; g.N = Cin * 9; g.K = (int)HW;
        if (i2l_gemm_split_bf16_ok(g)) {
            const int rc = i2l_gemm(g, base + lo.gemm, lo.gemm_bytes, s);
            if (rc != I2L_OK) return rc;
            wgrad_done = true;
        }
    }
    for (int b0 = 0; b0 < B && !wgrad_done; b0 += lo.chunk) {
        const int nb = B - b0 < lo.chunk ? B - b0 : lo.chunk;
        const size_t total = (size_t)nb * Cin * 9 * HW;
        size_t blocks = (total + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(im2col_t_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x + (size_t)b0 * Cin * HW, colT, nb,
                           Cin, H, W);
        I2L_CHECK_LAUNCH();
        GemmArgs g = gemm_args();
        g.A = dyp + (size_t)b0 * Cout * HW; g.lda = (long)HW; g.bsa = (long)(Cout * HW);
        g.W = colT; g.ldw = (long)HW; g.bsw = (long)(Cin * 9 * HW);
        g.nz = nb;
        g.split_bf16 = split;  // training GEMM: 3-way split bf16 matrix cores where the shape allows (gemm.hip)
        g.C = dw; g.ldc = Cin * 9;
        g.M = Cout; g.N = Cin * 9; g.K = (int)HW;
        g.accumulate = b0 > 0 ? 1 : 0;
        const int rc = i2l_gemm(g, base + lo.gemm, lo.gemm_bytes, s);
        if (rc != I2L_OK) return rc;
    }
    s = s_main;
    if (dx) {   // data gradient = conv3x3(dyp, flipped / transposed filter), Cout -> Cin channels
        const int rc = run_conv(false, dyp, w, nullptr, dx, nullptr, B, Cout, H, W, Cin, base + lo.wpack, lo.wpack_bytes, s,
                                exact);
        if (rc != I2L_OK) return rc;
    }
    return I2L_OK;
}
