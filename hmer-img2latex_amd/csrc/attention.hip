// Additive attention for a general source length S (Attention.forward, decoder.py:312-343):
//   energy[s] = tanh(W [hidden ; enc[s]] + b)   (:329-332)      score[s] = v . energy[s]   (:335)
//   weights = softmax_s(score)                  (:338)          context = sum_s weights[s] enc[s]   (:341)
// One workgroup per batch row.  The hidden half of the Linear is the same for every s and is
// computed once.  In the reference's decoder S is always 1 (one encoder vector per image), where
// the result equals enc bit-for-bit; this kernel backs the Attention class for any S.
#include <math.h>

#include "common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(NT) void attention_kernel(const float* __restrict__ hidden, const float* __restrict__ enc,
                                                       const float* __restrict__ w_attn, const float* __restrict__ b_attn,
                                                       const float* __restrict__ v, float* __restrict__ context,
                                                       int S, int H, int E) {
    extern __shared__ float sm[];
    float* hpart = sm;            // [H]  W[:, :H] @ hidden + b
    float* score = hpart + H;     // [S]
    float* red = score + S;       // [NT/64]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = H + E;
    const float* hb = hidden + (size_t)b * H;
    const float* eb = enc + (size_t)b * S * E;
    for (int j = tid; j < H; j += NT) {
        float a = 0.f;
        const float* wr = w_attn + (size_t)j * ld;
        for (int k = 0; k < H; ++k) a = fmaf(wr[k], hb[k], a);
        hpart[j] = a;
    }
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        float part = 0.f;
        for (int j = tid; j < H; j += NT) {
            const float* wr = w_attn + (size_t)j * ld + H;
            float a = hpart[j];
            for (int k = 0; k < E; ++k) a = fmaf(wr[k], eb[(size_t)s * E + k], a);
            part += v[j] * tanhf(a + b_attn[j]);
        }
        part = wave_sum_f(part);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int wv = 0; wv < NT / 64; ++wv) t += red[wv];
            score[s] = t;
        }
        __syncthreads();
    }
    // softmax over s (every thread computes the same max / sum; S is small)
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, score[s]);
    float den = 0.f;
    for (int s = 0; s < S; ++s) den += expf(score[s] - m);
    for (int e = tid; e < E; e += NT) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a = fmaf(expf(score[s] - m) / den, eb[(size_t)s * E + e], a);
        context[(size_t)b * E + e] = a;
    }
}

}  // namespace

extern "C" int i2l_attention_context_fwd(const float* hidden, const float* enc, const float* w_attn,
                                         const float* b_attn, const float* v, float* context, int B, int S, int H,
                                         int E, i2l_stream_t stream) {
    if (!hidden || !enc || !w_attn || !b_attn || !v || !context || B <= 0 || S <= 0 || H <= 0 || E <= 0)
        return I2L_ERR_ARG;
    const size_t lds = ((size_t)H + S + NT / 64) * sizeof(float);
    if (lds > 64 * 1024) return I2L_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(attention_kernel, dim3(B), dim3(NT), lds, i2l_s(stream), hidden, enc, w_attn, b_attn, v,
                       context, S, H, E);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
