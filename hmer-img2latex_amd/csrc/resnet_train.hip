// Training-mode pieces of the ResNet encoder (reference encoder.py:185-249 with model.train(): the torchvision trunk's
// BatchNorm2d layers normalise with BATCH statistics and update their running statistics -- also the frozen ones,
// `freeze_backbone` only clears requires_grad (encoder.py:201-210) -- and layer4 + the Linear receive gradients; with
// freeze_backbone=False, as the shipped config trains it (configs/config.yaml:43), every layer does).
//
// PRECISION (r04).  The training path is fp32-grade end to end, like the reference's fp32 branch (trainer.py:334-343):
// activations, raw conv outputs z and gradients are NHWC fp32 = row-major (M = B*H*W, C) matrices, and every convolution
// is a GEMM on the split-bf16 matrix-core kernel (3 x bf16 pieces per operand, 6 partial products, fp32 accumulation:
// the error class of an fp32 fmaf chain; gemm.hip).  r03 stored z and y in bf16 and took the batch statistics from the
// rounded z: a trunk with batch statistics amplifies a relative perturbation of 2^-9 by ~100x per 50 layers
// (profiles/micro/resnet_precision_cpu.py: even 16-bit operands leave gradient cosines of 0.988 against fp32), so
// nothing below fp32 grade reproduces the reference's gradients.  The bf16 inference path (resnet.hip) is untouched.
//   conv       z = col(x) W^T, col = the fp32 im2col image in the weight tensor's own (ci, ky, kx) column order (x itself
//              for 1x1 / stride 1); dw = dz^T col and dcol = dz W are the same kernel, col2im is a GATHER (deterministic)
//   forward    z -> per-channel batch mean / biased variance (fp32 partial sums per row slab, combined in double),
//              running statistics (momentum, unbiased variance), y = act(gamma * (z - mean) * invstd + beta + residual)
//   backward   dy (fp32) -> ReLU mask from y, per-channel sums, dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)),
//              dgamma, dbeta, the masked gradient for the residual branch
//   pooling    max-pool 3x3/2 forward / backward (first maximum of the window wins, as ATen), global average pool both ways.
// HBM-bound elementwise / reduction kernels: 16-byte accesses, one pass per tensor.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }

constexpr int SLAB_ROWS = 128;          // rows of the (M, C) matrix one workgroup reduces (M / 128 workgroups: 640+ at B = 64)

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(float* __restrict__ p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// Per-channel partial sums over a slab of rows.  Thread = (row lane r of 256 / CG, channel group of 8 channels);
// backward (MODE 1): (sum g, sum g * xhat) with g = dy masked by y > 0.  The forward statistics have their own kernel
// below: (sum d, sum d^2) with d = z - c, c = the channel's value in ROW 0 of the matrix -- a shift of the order of the mean,
// so that the variance comes out of ONE pass without the cancellation of sum z^2 / M - mean^2 (r04 first used a second
// pass over z - mean: two more launches per unit).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                         const float* __restrict__ y, const float* __restrict__ mean,
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
                float zv[8];
                load8(z + r * C + cg * 8, zv);
                {
                    float g[8];
                    load8(dy + r * C + cg * 8, g);
                    if (y) {
                        float yv[8];
                        load8(y + r * C + cg * 8, yv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (!(yv[j] > 0.f)) g[j] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s0[j] += g[j]; s1[j] += g[j] * (zv[j] - mu[j]) * is[j]; }
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

// The forward statistics' partial sums (sum d, sum d^2 with d = z - row 0, see above) accumulated and stored in DOUBLE: the
// kernel is bound by reading z once, the fp64 adds are free, and the one-pass variance then keeps ~1e-7 whatever the shift.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, long M, int C, double* __restrict__ part) {
    __shared__ double red[2][256][8];
    const int groups = C / 8;
    const int cg_per_pass = groups < 256 ? groups : 256;
    const int lanes = 256 / cg_per_pass;
    const int tid = threadIdx.x;
    const int cgl = tid % cg_per_pass, rl = tid / cg_per_pass;
    const long r0 = (long)blockIdx.x * SLAB_ROWS, r1 = r0 + SLAB_ROWS < M ? r0 + SLAB_ROWS : M;
    for (int cg0 = 0; cg0 < groups; cg0 += cg_per_pass) {
        const int cg = cg0 + cgl;
        double s0[8], s1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.0;
        if (cg < groups && rl < lanes) {
            float c0[8];
            load8(z + cg * 8, c0);                                // the shift: row 0
            for (long r = r0 + rl; r < r1; r += lanes) {
                float zv[8];
                load8(z + r * C + cg * 8, zv);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const double d = (double)(zv[j] - c0[j]); s0[j] += d; s1[j] += d * d; }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[0][tid][j] = s0[j]; red[1][tid][j] = s1[j]; }
        __syncthreads();
        if (rl == 0 && cg < groups) {                            // lanes added in lane order: deterministic
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double a = 0.0, b = 0.0;
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
template <typename T>
__device__ __forceinline__ void slab_sums(const T* __restrict__ part, int slabs, int C, int c, double (&red)[2][8][32],
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

// forward finalize: mean = c + sum d / M, biased variance = sum d^2 / M - (sum d / M)^2 (slab sums combined in double),
// invstd, running statistics (nn.BatchNorm2d: momentum update with the UNBIASED variance)
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double* __restrict__ part, const float* __restrict__ z_row0,
                                                             int slabs, long M, int C, float eps, float momentum,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var) {
    __shared__ double red[2][8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    double s, q;
    slab_sums(part, slabs, C, c, red, s, q);
    if ((threadIdx.x >> 5) != 0 || c >= C) return;
    const double dm = s / (double)M;
    const double mu = (double)z_row0[c] + dm;
    double var = q / (double)M - dm * dm;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ residual,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       int relu, long M, int C, float* __restrict__ y) {
    const long total = M * (C / 8);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % (C / 8));
        const long o = idx * 8;
        float zv[8], rv[8], out[8];
        load8(z + o, zv);
        if (residual) load8(residual + o, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            float v = (zv[j] - mean[c]) * invstd[c] * gamma[c] + beta[c];
            if (residual) v += rv[j];
            if (relu) v = fmaxf(v, 0.f);
            out[j] = v;
        }
        store8(y + o, out);
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

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ z, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ sums, long M, int C,
                                                           float* __restrict__ dz, float* __restrict__ dres, int dres_accumulate) {
    const long total = M * (C / 8);
    const float inv_m = 1.0f / (float)M;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % (C / 8));
        const long o = idx * 8;
        float zv[8], g[8], out[8];
        load8(z + o, zv);
        load8(dy + o, g);
        if (y) {
            float yv[8];
            load8(y + o, yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(yv[j] > 0.f)) g[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            const float xh = (zv[j] - mean[c]) * invstd[c];
            out[j] = gamma[c] * invstd[c] * (g[j] - sums[c] * inv_m - xh * sums[C + c] * inv_m);
        }
        store8(dz + o, out);
        if (dres) {
            if (dres_accumulate) {
                float a[8];
                load8(dres + o, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] += a[j];
            }
            store8(dres + o, g);
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

// nn.MaxPool2d(3, 2, 1) on NHWC fp32, 4 channels (16 bytes) per thread
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W,
                                                          int C, int Ho, int Wo) {
    const int c4n = C / 4;
    const long total = (long)B * Ho * Wo * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        long t = i / c4n;
        const int xo = (int)(t % Wo); t /= Wo;
        const int yo = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int yi = 2 * yo - 1 + ky, xi = 2 * xo - 1 + kx;
                if (yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + yi) * W + xi) * C + c4 * 4);
                best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
            }
        *reinterpret_cast<float4*>(y + (((size_t)b * Ho + yo) * Wo + xo) * C + c4 * 4) = best;
    }
}

// nn.MaxPool2d(3, 2, 1) backward on NHWC: every input pixel collects the gradients of the windows whose FIRST maximum
// (row-major window scan, as ATen) it is
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H,
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
                        const float v = x[(((size_t)b * H + wy) * W + wx) * C + c];
                        if (v > best || by < 0) { best = v; by = wy; bx = wx; }
                    }
                }
                if (by == iy && bx == ix) s += dy[(((size_t)b * Ho + oy) * Wo + ox) * C + c];
            }
        }
        dx[idx] = s;
    }
}

// AdaptiveAvgPool2d(1) + Flatten: (B,H,W,C) fp32 -> (B,C); thread = (image, 4 channels), positions added in order
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW) {
    const int c4n = C / 4;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * c4n) return;
    const int b = i / c4n, c4 = (i - b * c4n) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < HW; ++p) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HW + p) * C + c4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float inv = 1.0f / (float)HW;
    *reinterpret_cast<float4*>(y + (size_t)b * C + c4) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
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

struct ConvGeom {
    int Ho, Wo;
    long M, Kc;
    bool direct;        // 1x1 / stride 1 / no padding on NHWC: the column image IS x
};
inline bool conv_geom(int x_kind, int B, int H, int W, int Cin, int kh, int kw, int stride, int pad, ConvGeom* g) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return false;
    if (x_kind != 1 && x_kind != 2) return false;
    g->Ho = (H + 2 * pad - kh) / stride + 1;
    g->Wo = (W + 2 * pad - kw) / stride + 1;
    if (g->Ho <= 0 || g->Wo <= 0) return false;
    g->M = (long)B * g->Ho * g->Wo;
    g->Kc = (long)Cin * kh * kw;
    g->direct = x_kind == 1 && kh == 1 && kw == 1 && stride == 1 && pad == 0;
    return g->M <= 0x7fffffffl && g->Kc <= 0x7fffffffl;
}
inline size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

int launch_im2col(const void* x, int x_kind, int B, int H, int W, int C, int kh, int kw, int stride, int pad, int Ho, int Wo,
                  float* col, hipStream_t s) {
    const long total = (long)B * Ho * Wo * C * kh * kw;
    const dim3 g(grid_for(total)), b(256);
    if (x_kind == 0) hipLaunchKernelGGL((im2col_kernel<true, false>), g, b, 0, s, x, B, H, W, C, kh, kw, stride, pad, Ho, Wo, col);
    else if (x_kind == 1) hipLaunchKernelGGL((im2col_kernel<false, false>), g, b, 0, s, x, B, H, W, C, kh, kw, stride, pad, Ho, Wo, col);
    else if (x_kind == 2) hipLaunchKernelGGL((im2col_kernel<false, true>), g, b, 0, s, x, B, H, W, C, kh, kw, stride, pad, Ho, Wo, col);
    else return I2L_ERR_ARG;
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

}  // namespace

// ---------------------------------------------------------------- convolution, fp32 grade (forward and both gradients)
extern "C" size_t i2l_conv_f32_workspace_bytes(int x_kind, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                               int pad, int backward_dx) {
    ConvGeom c;
    if (Cout <= 0 || !conv_geom(x_kind, B, H, W, Cin, kh, kw, stride, pad, &c)) return 0;
    const int M = (int)c.M, Kc = (int)c.Kc;
    const size_t col = c.direct ? 0 : i2l_align((size_t)M * Kc * sizeof(float));
    const size_t dcol = (backward_dx && !c.direct) ? i2l_align((size_t)M * Kc * sizeof(float)) : 0;
    const size_t gemm = i2l_align(max3(i2l_gemm_workspace_bytes(M, Cout, Kc), i2l_gemm_workspace_bytes(Cout, Kc, M),
                                       i2l_gemm_workspace_bytes(M, Kc, Cout)));
    return col + dcol + gemm + 256;
}

extern "C" int i2l_conv_f32_fwd(const float* x, int x_kind, const float* w, float* z, int B, int H, int W, int Cin, int Cout,
                                int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, int flags,
                                i2l_stream_t stream) {
    ConvGeom c;
    if (!x || !w || !z || Cout <= 0 || !conv_geom(x_kind, B, H, W, Cin, kh, kw, stride, pad, &c)) return I2L_ERR_ARG;
    if (!workspace || workspace_bytes < i2l_conv_f32_workspace_bytes(x_kind, B, H, W, Cin, Cout, kh, kw, stride, pad, 0))
        return I2L_ERR_WORKSPACE;
    hipStream_t s = i2l_s(stream);
    const int M = (int)c.M, Kc = (int)c.Kc;
    char* p = static_cast<char*>(workspace);
    const float* col = x;
    if (!c.direct) {
        float* cimg = reinterpret_cast<float*>(p);
        p += i2l_align((size_t)M * Kc * sizeof(float));
        const int rc = launch_im2col(x, x_kind, B, H, W, Cin, kh, kw, stride, pad, c.Ho, c.Wo, cimg, s);
        if (rc != I2L_OK) return rc;
        col = cimg;
    }
    GemmArgs g = gemm_args();                         // z[m][n] = sum_k col[m][k] * w[n][k]
    g.split_bf16 = (flags & I2L_FLAG_EXACT_FP32) ? 0 : 1;
    g.A = col; g.lda = Kc;
    g.W = w; g.ldw = Kc;
    g.C = z; g.ldc = Cout;
    g.M = M; g.N = Cout; g.K = Kc;
    return i2l_gemm(g, p, workspace_bytes - (size_t)(p - static_cast<char*>(workspace)), s);
}

extern "C" int i2l_conv_f32_bwd(const float* x, int x_kind, const float* w, const float* dz, float* dx, float* dw, int B, int H,
                                int W, int Cin, int Cout, int kh, int kw, int stride, int pad, void* workspace,
                                size_t workspace_bytes, int flags, i2l_stream_t stream) {
    ConvGeom c;
    if (!x || !w || !dz || (!dx && !dw) || Cout <= 0 || !conv_geom(x_kind, B, H, W, Cin, kh, kw, stride, pad, &c)) return I2L_ERR_ARG;
    if (dx && x_kind != 1) return I2L_ERR_UNSUPPORTED;                  // the images need no gradient
    if (!workspace || workspace_bytes < i2l_conv_f32_workspace_bytes(x_kind, B, H, W, Cin, Cout, kh, kw, stride, pad, dx ? 1 : 0))
        return I2L_ERR_WORKSPACE;
    hipStream_t s = i2l_s(stream);
    const int M = (int)c.M, Kc = (int)c.Kc;
    char* p = static_cast<char*>(workspace);
    const float* col = x;
    float* dcol = dx;
    if (!c.direct) {
        if (dw) {
            float* cimg = reinterpret_cast<float*>(p);
            if (!(flags & I2L_FLAG_CONV_COL_READY)) {        // else: i2l_conv_f32_fwd left it at the head of this workspace
                const int rc = launch_im2col(x, x_kind, B, H, W, Cin, kh, kw, stride, pad, c.Ho, c.Wo, cimg, s);
                if (rc != I2L_OK) return rc;
            }
            col = cimg;
        }
        p += i2l_align((size_t)M * Kc * sizeof(float));
        if (dx) {
            dcol = reinterpret_cast<float*>(p);
            p += i2l_align((size_t)M * Kc * sizeof(float));
        }
    }
    const size_t gws = workspace_bytes - (size_t)(p - static_cast<char*>(workspace));
    const int split = (flags & I2L_FLAG_EXACT_FP32) ? 0 : 1;
    if (dw) {   // dw[n][k] = sum_m dz[m][n] * col[m][k]
        GemmArgs g = gemm_args();
        g.split_bf16 = split;
        g.A = dz; g.lda = Cout; g.a_kc = 0;
        g.W = col; g.ldw = Kc; g.w_kc = 0;
        g.C = dw; g.ldc = Kc;
        g.M = Cout; g.N = Kc; g.K = M;
        const int rc = i2l_gemm(g, p, gws, s);
        if (rc != I2L_OK) return rc;
    }
    if (dx) {   // dcol[m][k] = sum_n dz[m][n] * w[n][k]
        GemmArgs g = gemm_args();
        g.split_bf16 = split;
        g.A = dz; g.lda = Cout;
        g.W = w; g.ldw = Kc; g.w_kc = 0;
        g.C = dcol; g.ldc = Kc;
        g.M = M; g.N = Kc; g.K = Cout;
        const int rc = i2l_gemm(g, p, gws, s);
        if (rc != I2L_OK) return rc;
        if (!c.direct) {
            hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((long)B * H * W * Cin)), dim3(256), 0, s, dcol, B, H, W, Cin, kh, kw,
                               stride, pad, c.Ho, c.Wo, dx, 0);
            I2L_CHECK_LAUNCH();
        }
    }
    return I2L_OK;
}

// ---------------------------------------------------------------- BatchNorm with batch statistics
extern "C" size_t i2l_bn_train_workspace_bytes(int64_t M, int C) {
    if (M <= 0 || C <= 0) return 0;
    const size_t slabs = (size_t)((M + SLAB_ROWS - 1) / SLAB_ROWS);
    return i2l_align(slabs * 2 * (size_t)C * sizeof(double)) + i2l_align(2 * (size_t)C * sizeof(float));   // fwd: double partials
}

extern "C" int i2l_bn_train_fwd_f32(const float* z, const float* residual, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, int relu,
                                    float* y, float* save_mean, float* save_invstd, int64_t M, int C, void* workspace,
                                    size_t workspace_bytes, i2l_stream_t stream) {
    if (!z || !gamma || !beta || !y || !save_mean || !save_invstd || M <= 0 || C <= 0) return I2L_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return I2L_ERR_ARG;
    if (C % 8) return I2L_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < i2l_bn_train_workspace_bytes(M, C)) return I2L_ERR_WORKSPACE;
    hipStream_t s = i2l_s(stream);
    const int slabs = (int)((M + SLAB_ROWS - 1) / SLAB_ROWS);
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(slabs), dim3(256), 0, s, z, (long)M, C, part);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(i2l_cdiv(C, 32)), dim3(256), 0, s, part, z, slabs, (long)M, C, eps, momentum,
                       save_mean, save_invstd, running_mean, running_var);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * (C / 8))), dim3(256), 0, s, z, residual, gamma, beta, save_mean,
                       save_invstd, relu ? 1 : 0, (long)M, C, y);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_bn_train_bwd_f32(const float* dy, const float* y_relu, const float* z, const float* gamma,
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
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(slabs), dim3(256), 0, s, z, dy, y_relu, save_mean, save_invstd, (long)M, C, part);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(i2l_cdiv(C, 32)), dim3(256), 0, s, part, slabs, C, sums, dgamma, dbeta);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(M * (C / 8))), dim3(256), 0, s, dy, y_relu, z, gamma, save_mean,
                       save_invstd, sums, (long)M, C, dz, dres, dres_accumulate ? 1 : 0);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_im2col_f32(const void* x, int x_kind, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                              float* col, i2l_stream_t stream) {
    if (!x || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return I2L_ERR_ARG;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return I2L_ERR_ARG;
    return launch_im2col(x, x_kind, B, H, W, C, kh, kw, stride, pad, Ho, Wo, col, i2l_s(stream));
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

// ---------------------------------------------------------------- pooling (fp32 NHWC)
extern "C" int i2l_maxpool3x3s2_f32_fwd(const float* x, float* y, int B, int H, int W, int C, i2l_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return I2L_ERR_ARG;
    if (C % 4) return I2L_ERR_UNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long)B * Ho * Wo * (C / 4))), dim3(256), 0, i2l_s(stream), x, y, B, H, W,
                       C, Ho, Wo);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_maxpool3x3s2_f32_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, i2l_stream_t stream) {
    if (!x || !dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return I2L_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, i2l_s(stream), x, dy, B, H, W, C, Ho,
                       Wo, dx);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_global_avgpool_f32_fwd(const float* x, float* y, int B, int H, int W, int C, i2l_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return I2L_ERR_ARG;
    if (C % 4) return I2L_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(i2l_cdiv(B * (C / 4), 256)), dim3(256), 0, i2l_s(stream), x, y, B, C, H * W);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_global_avgpool_bwd_f32(const float* dfeat, float* dx, int B, int H, int W, int C, i2l_stream_t stream) {
    if (!dfeat || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return I2L_ERR_ARG;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, i2l_s(stream), dfeat, B, H * W, C, dx);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
