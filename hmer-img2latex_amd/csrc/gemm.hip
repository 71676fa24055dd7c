// fp32 "NT" GEMM on the gfx950 matrix cores:  C[m][n] = act(sum_k A[m][k] * W[n][k] + bias[n])
//
// Both operands are K-contiguous, which is how PyTorch stores nn.Linear / nn.LSTM
// weights (W is (out, in)), so y = x @ W^T needs no transpose of either operand.
// Arithmetic is exact fp32: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf
// chain (no TF32/xf32 on gfx950), so results differ from ATen's only by summation
// order.  Used for the encoder's Flatten+Linear+ReLU (encoder.py:125-127; M=B,
// N=E, K=40960 -> split-K), and for the decoder's loop-invariant GEMMs.
//
// Tile: 64x64x32 per 256-thread workgroup, 4 waves as 2x2, each wave one 32x32
// accumulator (16 VGPRs).  LDS rows are padded to 33 floats so that the MFMA
// operand read (32 lanes walk 32 rows at one k) is bank-conflict-free.
// Split-K: grid.z slices of K write fp32 partial slabs; a second kernel sums
// the slabs in slice order (deterministic) and applies bias/ReLU/permutation.
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32, LDP = BK + 1;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int perm_col(int n, int perm_h) {
    if (perm_h <= 0) return n;
    const int g = n / perm_h, j = n - g * perm_h;
    return 4 * j + g;
}

template <bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g, float* __restrict__ partial,
                                                      int k_chunk, int splits) {
    __shared__ float As[BM * LDP];
    __shared__ float Ws[BN * LDP];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, z = blockIdx.z;
    const int kbeg = z * k_chunk;
    const int kend = min(g.K, kbeg + k_chunk);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        // ---- stage A and W tiles (64 rows x 32 k) into LDS, zero-filled outside the matrix
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256;          // 0..511 : 64 rows x 8 float4
            const int row = idx >> 3, c4 = (idx & 7) * 4;
            const int k = k0 + c4;
            float va[4] = {0.f, 0.f, 0.f, 0.f}, vw[4] = {0.f, 0.f, 0.f, 0.f};
            if (VEC) {
                if (m0 + row < g.M && k < kend) {   // K % 4 == 0 and k_chunk % 4 == 0: all-or-nothing
                    const float4 t = *reinterpret_cast<const float4*>(g.A + (size_t)(m0 + row) * g.lda + k);
                    va[0] = t.x; va[1] = t.y; va[2] = t.z; va[3] = t.w;
                }
                if (n0 + row < g.N && k < kend) {
                    const float4 t = *reinterpret_cast<const float4*>(g.W + (size_t)(n0 + row) * g.ldw + k);
                    vw[0] = t.x; vw[1] = t.y; vw[2] = t.z; vw[3] = t.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (m0 + row < g.M && k + e < kend) va[e] = g.A[(size_t)(m0 + row) * g.lda + k + e];
                    if (n0 + row < g.N && k + e < kend) vw[e] = g.W[(size_t)(n0 + row) * g.ldw + k + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[row * LDP + c4 + e] = va[e];
                Ws[row * LDP + c4 + e] = vw[e];
            }
        }
        __syncthreads();
        const float* ap = &As[(wm * 32 + li) * LDP + lk];
        const float* wp = &Ws[(wn * 32 + li) * LDP + lk];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], wp[kk], acc, 0, 0, 0);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < g.M && n < g.N) {
            if (splits > 1) {
                partial[((size_t)z * g.M + m) * g.N + n] = acc[r];
            } else {
                float v = acc[r];
                if (g.bias) v += g.bias[n];
                if (g.bias2) v += g.bias2[n];
                if (g.relu) v = fmaxf(v, 0.f);
                g.C[(size_t)m * g.ldc + perm_col(n, g.perm_h)] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g, const float* __restrict__ partial,
                                                            int splits) {
    const size_t total = (size_t)g.M * g.N;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / g.N), n = (int)(i - (size_t)m * g.N);
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += partial[(size_t)z * total + i];   // fixed slice order
        if (g.bias) v += g.bias[n];
        if (g.bias2) v += g.bias2[n];
        if (g.relu) v = fmaxf(v, 0.f);
        g.C[(size_t)m * g.ldc + perm_col(n, g.perm_h)] = v;
    }
}

void plan(int M, int N, int K, int* splits, int* k_chunk) {
    const int tiles = i2l_cdiv(M, BM) * i2l_cdiv(N, BN);
    int s = 1;
    if (tiles < 256 && K >= 1024) {
        s = 512 / tiles;
        const int max_s = K / 512;           // keep >= 512 of K per slice
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
    }
    int kc = i2l_cdiv(i2l_cdiv(K, s), BK) * BK;
    s = i2l_cdiv(K, kc);
    *splits = s;
    *k_chunk = kc;
}

}  // namespace

size_t i2l_gemm_workspace_bytes(int M, int N, int K) {
    int s, kc;
    plan(M, N, K, &s, &kc);
    return s > 1 ? i2l_align((size_t)s * M * N * sizeof(float)) : 0;
}

int i2l_gemm_nt(const GemmArgs& g, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (!g.A || !g.W || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return I2L_ERR_ARG;
    int splits, kc;
    plan(g.M, g.N, g.K, &splits, &kc);
    if (splits > 1 && (!ws || ws_bytes < (size_t)splits * g.M * g.N * sizeof(float))) return I2L_ERR_WORKSPACE;
    const bool vec = (g.K % 4 == 0) && (g.lda % 4 == 0) && (g.ldw % 4 == 0) &&
                     (reinterpret_cast<uintptr_t>(g.A) % 16 == 0) && (reinterpret_cast<uintptr_t>(g.W) % 16 == 0);
    dim3 grid(i2l_cdiv(g.N, BN), i2l_cdiv(g.M, BM), splits);
    if (vec)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), 0, stream, g, (float*)ws, kc, splits);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), 0, stream, g, (float*)ws, kc, splits);
    I2L_CHECK_LAUNCH();
    if (splits > 1) {
        const size_t total = (size_t)g.M * g.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g, (const float*)ws, splits);
        I2L_CHECK_LAUNCH();
    }
    return I2L_OK;
}

extern "C" size_t i2l_linear_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return i2l_gemm_workspace_bytes(M, N, K);
}

extern "C" int i2l_linear_bias_act_fwd(const float* x, const float* w, const float* bias, float* y,
                                        int M, int K, int N, int relu, void* workspace,
                                        size_t workspace_bytes, i2l_stream_t stream) {
    if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0) return I2L_ERR_ARG;
    GemmArgs g{};
    g.A = x; g.lda = K;
    g.W = w; g.ldw = K;
    g.bias = bias; g.bias2 = nullptr;
    g.C = y; g.ldc = N;
    g.M = M; g.N = N; g.K = K;
    g.relu = relu ? 1 : 0;
    g.perm_h = 0;
    return i2l_gemm_nt(g, workspace, workspace_bytes, i2l_s(stream));
}
