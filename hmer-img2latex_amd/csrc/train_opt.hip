// Gradient clipping + Adam on flat fp32 buffers (trainer.py:338-342 with the optimizer of :91-93):
//   g      = grad * (1 / count)                     loss is a mean over non-PAD tokens: the backward kernels
//                                                   produce the gradient of the SUM, `count` arrives in device
//                                                   memory (after the data-parallel all-reduce it is the
//                                                   GLOBAL token count)
//   total  = ||g||_2 over all parameters            nn.utils.clip_grad_norm_: coef = min(1, max_norm/(total+1e-6))
//   g      = g * coef + weight_decay * p            torch.optim.Adam's L2 is coupled (added to the gradient)
//   m, v, p updated with bias correction            denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= lr/(1-b1^t) * m/denom
// Three launches, no host synchronisation; the sum of squares is reduced in a fixed order (deterministic).
// Non-finite total norm (Inf / NaN gradients: an fp16-scaled overflow, or the NaN a grouped training kernel writes
// when one of its bounded waits expires) => the call is a no-op on p, m, v: stats[3] = 1 and a counter in the
// workspace remembers how many calls were skipped, so the bias correction keeps counting APPLIED steps only.
#include <math.h>

#include "common.h"

namespace {

constexpr int SQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, size_t n4,
                                                            double* __restrict__ partial) {
    __shared__ double red[256];
    double s = 0.0;
    // n4 = number of 16-byte pieces read as such (0 when the buffer is not 16-byte aligned); the rest element by element
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = g4[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v = g[i];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// stats_out[0] = total gradient norm (after the 1/count scaling), [1] = clip coefficient, [2] = 1/count,
// [3] = 1 when the update is skipped; coefs (workspace) = {apply?, lr / bc1, sqrt(bc2)} of the applied-step number
__global__ __launch_bounds__(256) void norm_final_kernel(const double* __restrict__ partial, int nblk,
                                                         const float* __restrict__ count_ptr, float max_norm,
                                                         float* __restrict__ stats_out, unsigned* __restrict__ skipped,
                                                         int step, float lr, float b1, float b2,
                                                         float* __restrict__ coefs) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float cnt = count_ptr ? fmaxf(*count_ptr, 1.f) : 1.f;
        const float inv = 1.f / cnt;
        const float total = (float)sqrt(red[0]) * inv;
        float coef = 1.f;
        if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.f);
        const bool skip = !(total <= 3.0e38f);           // NaN or Inf
        unsigned nskip = *skipped;
        if (skip) { coef = 0.f; *skipped = ++nskip; }
        stats_out[0] = total;
        stats_out[1] = coef;
        stats_out[2] = inv;
        stats_out[3] = skip ? 1.f : 0.f;
        int applied = step - (int)nskip;                 // optimizer step number among the APPLIED updates
        if (applied < 1) applied = 1;
        coefs[0] = skip ? 0.f : 1.f;
        coefs[1] = lr / (1.f - powf(b1, (float)applied));
        coefs[2] = sqrtf(1.f - powf(b2, (float)applied));
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   const float* __restrict__ stats, const float* __restrict__ coefs,
                                                   float b1, float b2, float eps, float wd) {
    if (coefs[0] == 0.f) return;                     // non-finite gradients: leave p, m, v untouched
    const float gs = stats[1] * stats[2];            // clip coefficient * 1/count
    const float step = coefs[1], bc2_sqrt = coefs[2];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] * gs + wd * pi;
        const float mi = m[i] + (1.f - b1) * (gi - m[i]);          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

}  // namespace

// workspace: [SQ_BLOCKS doubles of partial sums][256 B: skipped-call counter, 3 step coefficients]
extern "C" size_t i2l_optimizer_workspace_bytes(void) { return i2l_align(SQ_BLOCKS * sizeof(double)) + 256; }

extern "C" int i2l_grad_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                        size_t n, const float* count_ptr, float max_norm, float lr, float beta1,
                                        float beta2, float eps, float weight_decay, int step, void* workspace,
                                        size_t workspace_bytes, float* stats_out, i2l_stream_t stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !stats_out || n == 0 || step < 1) return I2L_ERR_ARG;
    if (!workspace || workspace_bytes < i2l_optimizer_workspace_bytes()) return I2L_ERR_WORKSPACE;
    hipStream_t s = i2l_s(stream);
    double* partial = static_cast<double*>(workspace);
    char* tail = static_cast<char*>(workspace) + i2l_align(SQ_BLOCKS * sizeof(double));
    unsigned* skipped = reinterpret_cast<unsigned*>(tail);
    float* coefs = reinterpret_cast<float*>(tail + 16);
    size_t nb = (n + 255) / 256;
    const int nblk = (int)(nb > (size_t)SQ_BLOCKS ? SQ_BLOCKS : nb);
    const size_t n4 = reinterpret_cast<uintptr_t>(grads) % 16 == 0 ? n / 4 : 0;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(256), 0, s, grads, n, n4, partial);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, s, (const double*)partial, nblk, count_ptr, max_norm,
                       stats_out, skipped, step, lr, beta1, beta2, coefs);
    I2L_CHECK_LAUNCH();
    const int ablk = (int)(nb > 4096 ? 4096 : nb);
    hipLaunchKernelGGL(adam_kernel, dim3(ablk), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n,
                       (const float*)stats_out, (const float*)coefs, beta1, beta2, eps, weight_decay);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
