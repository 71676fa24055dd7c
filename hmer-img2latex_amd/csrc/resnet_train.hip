// Training-mode pieces of the ResNet encoder (reference encoder.py:185-249 with model.train(): the torchvision trunk's
// BatchNorm2d layers normalise with BATCH statistics and update their running statistics -- also the frozen ones,
// `freeze_backbone` only clears requires_grad (encoder.py:201-210) -- and layer4 + the Linear receive gradients; with
// freeze_backbone=False, as the shipped config trains it (configs/config.yaml:43), every layer does).
//
// Layouts are those of the inference path (resnet.hip): activations NHWC bf16 = a row-major (M = B*H*W, C) matrix.
// The convolutions themselves stay on the bf16 GEMM kernels (i2l_conv_bn_act_bf16_fwd with an identity BatchNorm gives
// the raw conv output z); this file adds what surrounds them:
//   forward    z -> per-channel batch mean / biased variance (fp32 partial sums per row slab, combined in double),
//              running statistics (momentum, unbiased variance), y = act(gamma * (z - mean) * invstd + beta + residual)
//   backward   dy (fp32) -> ReLU mask from y, per-channel sums, dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)),
//              dgamma, dbeta, the masked gradient for the residual branch
//   conv grads explicit im2col (fp32, column order (ci, ky, kx) = the weight tensor's) so that the weight / data
//              gradients are the two GEMMs of i2l_linear_bias_act_bwd, and col2im as a GATHER (deterministic, no atomics)
//   pooling    max-pool 3x3/2 backward (first maximum of the window wins, as ATen), global average pool backward.
// Gradients are fp32 NHWC.  HBM-bound elementwise / reduction kernels: 16-byte accesses, one pass per tensor.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

constexpr int SLAB_ROWS = 128;          // rows of the (M, C) matrix one workgroup reduces (M / 128 workgroups: 640+ at B = 64)

// Per-channel partial sums over a slab of rows.  Thread = (row lane r of 256 / CG, channel group of 8 channels);
// MODE 0: (sum z, sum z^2); MODE 1: (sum g, sum g * xhat) with g = dy masked by y > 0.
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const bf16_t* __restrict__ z, const float* __restrict__ dy,
                                                         const bf16_t* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, long M, int C,
                                                         float* __restrict__ part) {
    __shared__ float red[2][256][8];
    const int groups = C / 8;                                    // C % 8 == 0
    const int cg_per_pass = groups < 256 ? groups : 256;
    const int lanes = 256 / cg_per_pass;                         // row lanes per channel group
    const int tid = threadIdx.x;
    const int cgl = tid % cg_per_pass, rl = tid / cg_per_pass;
    const long r0 = (long)blockIdx.x * SLAB_ROWS, r1 = r0 + SLAB_ROWS < M ? r0 + SLAB_ROWS : M;
    for (int cg0 = 0; cg0 < groups; cg0 += cg_per_pass) {
        const int cg = cg0 + cgl;
        float s0[8], s1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
        float mu[8], is[8];
        if (MODE == 1 && cg < groups && rl < lanes)
#pragma unroll
            for (int j = 0; j < 8; ++j) { mu[j] = mean[cg * 8 + j]; is[j] = invstd[cg * 8 + j]; }
        if (cg < groups && rl < lanes) {
            for (long r = r0 + rl; r < r1; r += lanes) {
                const uint4 zv = *reinterpret_cast<const uint4*>(z + r * C + cg * 8);
                const bf16_t* zp = reinterpret_cast<const bf16_t*>(&zv);
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float v = bf2f(zp[j]); s0[j] += v; s1[j] += v * v; }
                } else {
                    const float4 g0 = *reinterpret_cast<const float4*>(dy + r * C + cg * 8);
                    const float4 g1 = *reinterpret_cast<const float4*>(dy + r * C + cg * 8 + 4);
                    float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                    if (y) {
                        const uint4 yv = *reinterpret_cast<const uint4*>(y + r * C + cg * 8);
                        const bf16_t* yp = reinterpret_cast<const bf16_t*>(&yv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (!(bf2f(yp[j]) > 0.f)) g[j] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s0[j] += g[j]; s1[j] += g[j] * (bf2f(zp[j]) - mu[j]) * is[j]; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[0][tid][j] = s0[j]; red[1][tid][j] = s1[j]; }
        __syncthreads();
        if (rl == 0 && cg < groups) {                            // lanes added in lane order: deterministic
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = 0.f, b = 0.f;
                for (int l = 0; l < lanes; ++l) { a += red[0][l * cg_per_pass + cgl][j]; b += red[1][l * cg_per_pass + cgl][j]; }
                part[((size_t)blockIdx.x * 2 + 0) * C + cg * 8 + j] = a;
                part[((size_t)blockIdx.x * 2 + 1) * C + cg * 8 + j] = b;
            }
        }
        __syncthreads();
    }
}

// sums of the slab partials of 32 channels per workgroup: thread = (channel c = tid & 31, slab lane tid >> 5), slabs
// lane, lane + 8, ... added in double, the 8 lane sums in lane order (deterministic)
__device__ __forceinline__ void slab_sums(const float* __restrict__ part, int slabs, int C, int c, double (&red)[2][8][32],
                                          double& s, double& q) {
    const int cl = threadIdx.x & 31, l = threadIdx.x >> 5;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int sb = l; sb < slabs; sb += 8) { a += part[((size_t)sb * 2) * C + c]; b += part[((size_t)sb * 2 + 1) * C + c]; }
    red[0][l][cl] = a;
    red[1][l][cl] = b;
    __syncthreads();
    s = q = 0.0;
    if (l == 0)
        for (int k = 0; k < 8; ++k) { s += red[0][k][cl]; q += red[1][k][cl]; }
}

// forward finalize: mean, invstd, running statistics (nn.BatchNorm2d: momentum update with the UNBIASED variance)
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ part, int slabs, long M, int C, float eps,
                                                             float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var) {
    __shared__ double red[2][8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    double s, q;
    slab_sums(part, slabs, C, c, red, s, q);
    if ((threadIdx.x >> 5) != 0 || c >= C) return;
    const double mu = s / (double)M;
    double var = q / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ residual,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       int relu, long M, int C, bf16_t* __restrict__ y) {
    const long total = M * (C / 8);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % (C / 8));
        const long o = idx * 8;
        const uint4 zv = *reinterpret_cast<const uint4*>(z + o);
        const bf16_t* zp = reinterpret_cast<const bf16_t*>(&zv);
        uint4 rv = make_uint4(0, 0, 0, 0);
        if (residual) rv = *reinterpret_cast<const uint4*>(residual + o);
        const bf16_t* rp = reinterpret_cast<const bf16_t*>(&rv);
        uint4 ov;
        bf16_t* op = reinterpret_cast<bf16_t*>(&ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            // the normalised value is rounded to bf16 BEFORE the residual is added, like the inference epilogue
            float v = bf2f(f2bf((bf2f(zp[j]) - mean[c]) * invstd[c] * gamma[c] + beta[c]));
            if (residual) v += bf2f(rp[j]);
            if (relu) v = fmaxf(v, 0.f);
            op[j] = f2bf(v);
        }
        *reinterpret_cast<uint4*>(y + o) = ov;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ part, int slabs, int C, float* __restrict__ sums,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[2][8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    double s, q;
    slab_sums(part, slabs, C, c, red, s, q);
    if ((threadIdx.x >> 5) != 0 || c >= C) return;
    sums[c] = (float)s;
    sums[C + c] = (float)q;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const bf16_t* __restrict__ y,
                                                           const bf16_t* __restrict__ z, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ sums, long M, int C,
                                                           float* __restrict__ dz, float* __restrict__ dres, int dres_accumulate) {
    const long total = M * (C / 8);
    const float inv_m = 1.0f / (float)M;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % (C / 8));
        const long o = idx * 8;
        const uint4 zv = *reinterpret_cast<const uint4*>(z + o);
        const bf16_t* zp = reinterpret_cast<const bf16_t*>(&zv);
        const float4 g0 = *reinterpret_cast<const float4*>(dy + o), g1 = *reinterpret_cast<const float4*>(dy + o + 4);
        float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        if (y) {
            const uint4 yv = *reinterpret_cast<const uint4*>(y + o);
            const bf16_t* yp = reinterpret_cast<const bf16_t*>(&yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(bf2f(yp[j]) > 0.f)) g[j] = 0.f;
        }
        float out[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            const float xh = (bf2f(zp[j]) - mean[c]) * invstd[c];
            out[j] = gamma[c] * invstd[c] * (g[j] - sums[c] * inv_m - xh * sums[C + c] * inv_m);
        }
        *reinterpret_cast<float4*>(dz + o) = make_float4(out[0], out[1], out[2], out[3]);
        *reinterpret_cast<float4*>(dz + o + 4) = make_float4(out[4], out[5], out[6], out[7]);
        if (dres) {
            if (dres_accumulate) {
                const float4 a0 = *reinterpret_cast<const float4*>(dres + o), a1 = *reinterpret_cast<const float4*>(dres + o + 4);
                g[0] += a0.x; g[1] += a0.y; g[2] += a0.z; g[3] += a0.w; g[4] += a1.x; g[5] += a1.y; g[6] += a1.z; g[7] += a1.w;
            }
            *reinterpret_cast<float4*>(dres + o) = make_float4(g[0], g[1], g[2], g[3]);
            *reinterpret_cast<float4*>(dres + o + 4) = make_float4(g[4], g[5], g[6], g[7]);
        }
    }
}

// col[m = (b, oy, ox)][n = ci * kh*kw + ky * kw + kx] = x[b][oy*s + ky - p][ox*s + kx - p][ci], 0 outside
template <bool BF16, bool NCHW>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ xv, int B, int H, int W, int C, int kh, int kw,
                                                     int stride, int pad, int Ho, int Wo, float* __restrict__ col) {
    const int KK = kh * kw;
    const long N = (long)C * KK, total = (long)B * Ho * Wo * N;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / N;
        const int n = (int)(idx - m * N);
        const int ci = n / KK, tap = n - ci * KK;
        const int ky = tap / kw, kx = tap - ky * kw;
        const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long)Wo * Ho));
        const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const size_t o = NCHW ? (((size_t)b * C + ci) * H + iy) * W + ix : (((size_t)b * H + iy) * W + ix) * C + ci;
            v = BF16 ? bf2f(static_cast<const bf16_t*>(xv)[o]) : static_cast<const float*>(xv)[o];
        }
        col[idx] = v;
    }
}

// dx[b][iy][ix][ci] = sum over taps with (iy + p - ky) % s == 0 of dcol[(b, (iy+p-ky)/s, (ix+p-kx)/s)][ci*KK + tap]
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, int B, int H, int W, int C, int kh, int kw,
                                                     int stride, int pad, int Ho, int Wo, float* __restrict__ dx, int accumulate) {
    const int KK = kh * kw;
    const long N = (long)C * KK, total = (long)B * H * W * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ci = (int)(idx % C);
        const long pix = idx / C;
        const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
        float s = accumulate ? dx[idx] : 0.f;
        for (int ky = 0; ky < kh; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                s += dcol[(((long)b * Ho + oy) * Wo + ox) * N + (long)ci * KK + ky * kw + kx];
            }
        }
        dx[idx] = s;
    }
}

// nn.MaxPool2d(3, 2, 1) backward on NHWC: every input pixel collects the gradients of the windows whose FIRST maximum
// (row-major window scan, as ATen) it is
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ dy, int B, int H,
                                                          int W, int C, int Ho, int Wo, float* __restrict__ dx) {
    const long total = (long)B * H * W * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long pix = idx / C;
        const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
        float s = 0.f;
        // windows covering (iy, ix): rows 2*oy - 1 .. 2*oy + 1  ->  oy in [iy / 2, (iy + 1) / 2], same for ox
        for (int oy = iy / 2; oy <= (iy + 1) / 2 && oy < Ho; ++oy) {
            for (int ox = ix / 2; ox <= (ix + 1) / 2 && ox < Wo; ++ox) {
                float best = -INFINITY;
                int by = -1, bx = -1;
                for (int wy = 2 * oy - 1; wy <= 2 * oy + 1; ++wy) {
                    if (wy < 0 || wy >= H) continue;
                    for (int wx = 2 * ox - 1; wx <= 2 * ox + 1; ++wx) {
                        if (wx < 0 || wx >= W) continue;
                        const float v = bf2f(x[(((size_t)b * H + wy) * W + wx) * C + c]);
                        if (v > best || by < 0) { best = v; by = wy; bx = wx; }
                    }
                }
                if (by == iy && bx == ix) s += dy[(((size_t)b * Ho + oy) * Wo + ox) * C + c];
            }
        }
        dx[idx] = s;
    }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dfeat, int B, int HW, int C, float* __restrict__ dx) {
    const long total = (long)B * HW * C;
    const float inv = 1.0f / (float)HW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int b = (int)(idx / ((long)HW * C));
        dx[idx] = dfeat[(size_t)b * C + c] * inv;
    }
}

inline int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" size_t i2l_bn_train_workspace_bytes(int64_t M, int C) {
    if (M <= 0 || C <= 0) return 0;
    const size_t slabs = (size_t)((M + SLAB_ROWS - 1) / SLAB_ROWS);
    return i2l_align(slabs * 2 * (size_t)C * sizeof(float)) + i2l_align(2 * (size_t)C * sizeof(float));
}

extern "C" int i2l_bn_train_fwd_bf16(const void* z, const void* residual, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum, float eps, int relu,
                                     void* y, float* save_mean, float* save_invstd, int64_t M, int C, void* workspace,
                                     size_t workspace_bytes, i2l_stream_t stream) {
    if (!z || !gamma || !beta || !y || !save_mean || !save_invstd || M <= 0 || C <= 0) return I2L_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return I2L_ERR_ARG;
    if (C % 8) return I2L_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < i2l_bn_train_workspace_bytes(M, C)) return I2L_ERR_WORKSPACE;
    hipStream_t s = i2l_s(stream);
    const int slabs = (int)((M + SLAB_ROWS - 1) / SLAB_ROWS);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(slabs), dim3(256), 0, s, static_cast<const bf16_t*>(z), nullptr, nullptr,
                       nullptr, nullptr, (long)M, C, part);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(i2l_cdiv(C, 32)), dim3(256), 0, s, part, slabs, (long)M, C, eps, momentum,
                       save_mean, save_invstd, running_mean, running_var);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * (C / 8))), dim3(256), 0, s, static_cast<const bf16_t*>(z),
                       static_cast<const bf16_t*>(residual), gamma, beta, save_mean, save_invstd, relu ? 1 : 0, (long)M, C,
                       static_cast<bf16_t*>(y));
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_bn_train_bwd_bf16(const float* dy, const void* y_relu, const void* z, const float* gamma,
                                     const float* save_mean, const float* save_invstd, float* dz, float* dgamma,
                                     float* dbeta, float* dres, int dres_accumulate, int64_t M, int C, void* workspace,
                                     size_t workspace_bytes, i2l_stream_t stream) {
    if (!dy || !z || !gamma || !save_mean || !save_invstd || !dz || M <= 0 || C <= 0) return I2L_ERR_ARG;
    if (C % 8) return I2L_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < i2l_bn_train_workspace_bytes(M, C)) return I2L_ERR_WORKSPACE;
    hipStream_t s = i2l_s(stream);
    const int slabs = (int)((M + SLAB_ROWS - 1) / SLAB_ROWS);
    float* part = static_cast<float*>(workspace);
    float* sums = reinterpret_cast<float*>(static_cast<char*>(workspace) + i2l_align((size_t)slabs * 2 * C * sizeof(float)));
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(slabs), dim3(256), 0, s, static_cast<const bf16_t*>(z), dy,
                       static_cast<const bf16_t*>(y_relu), save_mean, save_invstd, (long)M, C, part);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(i2l_cdiv(C, 32)), dim3(256), 0, s, part, slabs, C, sums, dgamma, dbeta);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(M * (C / 8))), dim3(256), 0, s, dy, static_cast<const bf16_t*>(y_relu),
                       static_cast<const bf16_t*>(z), gamma, save_mean, save_invstd, sums, (long)M, C, dz, dres,
                       dres_accumulate ? 1 : 0);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_im2col_f32(const void* x, int x_kind, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                              float* col, i2l_stream_t stream) {
    if (!x || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return I2L_ERR_ARG;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return I2L_ERR_ARG;
    const long total = (long)B * Ho * Wo * C * kh * kw;
    hipStream_t s = i2l_s(stream);
    const dim3 g(grid_for(total)), b(256);
    if (x_kind == 0) hipLaunchKernelGGL((im2col_kernel<true, false>), g, b, 0, s, x, B, H, W, C, kh, kw, stride, pad, Ho, Wo, col);
    else if (x_kind == 1) hipLaunchKernelGGL((im2col_kernel<false, false>), g, b, 0, s, x, B, H, W, C, kh, kw, stride, pad, Ho, Wo, col);
    else if (x_kind == 2) hipLaunchKernelGGL((im2col_kernel<false, true>), g, b, 0, s, x, B, H, W, C, kh, kw, stride, pad, Ho, Wo, col);
    else return I2L_ERR_ARG;
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_col2im_f32(const float* dcol, int B, int H, int W, int C, int kh, int kw, int stride, int pad, float* dx,
                              int accumulate, i2l_stream_t stream) {
    if (!dcol || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return I2L_ERR_ARG;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return I2L_ERR_ARG;
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, i2l_s(stream), dcol, B, H, W, C, kh, kw,
                       stride, pad, Ho, Wo, dx, accumulate ? 1 : 0);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_maxpool3x3s2_bf16_bwd(const void* x, const float* dy, float* dx, int B, int H, int W, int C, i2l_stream_t stream) {
    if (!x || !dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return I2L_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, i2l_s(stream),
                       static_cast<const bf16_t*>(x), dy, B, H, W, C, Ho, Wo, dx);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_global_avgpool_bwd_f32(const float* dfeat, float* dx, int B, int H, int W, int C, i2l_stream_t stream) {
    if (!dfeat || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return I2L_ERR_ARG;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, i2l_s(stream), dfeat, B, H * W, C, dx);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
