// fp32 GEMM on the gfx950 matrix cores:
//     C[m][n] (+)= alpha * act( sum_z sum_k A_z(m,k) * W_z(n,k)  + bias[n] )
//
// Operands may be K-contiguous (A(m,k) = A[m*lda + k], how PyTorch stores nn.Linear / nn.LSTM
// weights and activations: y = x @ W^T needs no transpose) or M/N-contiguous
// (A(m,k) = A[k*lda + m], which gives the A^T B and A B forms the backward pass needs without
// materialising a transpose).  `nz` operand pairs (A + z*bsa, W + z*bsw) are summed into one
// result (the conv weight-gradient sums one GEMM per image).
// Arithmetic is exact fp32: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain
// (no TF32/xf32 on gfx950), so results differ from ATen's only by summation order.
//
// Tile: 64x64x32 per 256-thread workgroup, 4 waves as 2x2, each wave one 32x32
// accumulator (16 VGPRs).  LDS rows are padded to 33 floats so that the MFMA
// operand read (32 lanes walk 32 rows at one k) is bank-conflict-free.
// Split-K / batch: grid.z slices write fp32 partial slabs; a second kernel sums the slabs in
// slice order (deterministic) and applies bias/ReLU/permutation/alpha/accumulate.
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 64, LDP = BK + 1;
constexpr int NLD = BM * BK / 4 / 256;      // float4 loads per thread per operand tile
constexpr int KQ = BK / 4;                  // float4 chunks along k
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int perm_col(int n, int perm_h) {
    if (perm_h <= 0) return n;
    const int g = n / perm_h, j = n - g * perm_h;
    return 4 * j + g;
}

__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int m, int n, float v) {
    if (g.bias) v += g.bias[n];
    if (g.bias2) v += g.bias2[n];
    if (g.relu) v = fmaxf(v, 0.f);
    v *= g.alpha;
    float* c = g.C + (size_t)m * g.ldc + perm_col(n, g.perm_h);
    *c = g.accumulate ? *c + v : v;
}

// A (64 rows x 32 k) operand tile moves global -> registers -> LDS (row stride 33), zero-filled outside
// [rows) x [kbeg,kend).  The two halves are separate so the global loads of tile t+1 can be in flight
// while the MFMAs of tile t run (the LDS write happens after the barrier that ends tile t).
template <bool KC, bool VEC>
__device__ __forceinline__ void load_tile(float (&v)[NLD][4], const float* __restrict__ src, long ld, int row0, int rows,
                                          int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * 256;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = 0.f;
        if (KC) {                                       // 64 rows x KQ float4 along k
            const int row = idx / KQ, k = k0 + (idx % KQ) * 4;
            if (row0 + row < rows) {
                const float* p = src + (size_t)(row0 + row) * ld + k;
                if (VEC && k + 3 < kend) {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < kend) v[i][e] = p[e];
                }
            }
        } else {                                        // BK k x 16 float4 along the row index
            const int k = k0 + (idx >> 4), r4 = (idx & 15) * 4;
            if (k < kend) {
                const float* p = src + (size_t)k * ld + row0 + r4;
                if (VEC && row0 + r4 + 3 < rows) {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (row0 + r4 + e < rows) v[i][e] = p[e];
                }
            }
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* dst, const float (&v)[NLD][4], int tid) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * 256;
        if (KC) {
            const int row = idx / KQ, c4 = (idx % KQ) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[row * LDP + c4 + e] = v[i][e];
        } else {
            const int kk = idx >> 4, r4 = (idx & 15) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(r4 + e) * LDP + kk] = v[i][e];
        }
    }
}

template <bool A_KC, bool W_KC, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g, float* __restrict__ partial, int k_chunk,
                                                   int ksplits, int slabs) {
    __shared__ float As[BM * LDP];
    __shared__ float Ws[BN * LDP];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int slab = blockIdx.z;
    const int z = slab / ksplits, ks = slab - z * ksplits;
    const int kbeg = ks * k_chunk;
    const int kend = min(g.K, kbeg + k_chunk);
    const float* Az = g.A + (size_t)z * g.bsa;
    const float* Wz = g.W + (size_t)z * g.bsw;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    float va[NLD][4], vw[NLD][4];
    load_tile<A_KC, VEC>(va, Az, g.lda, m0, g.M, kbeg, kend, tid);
    load_tile<W_KC, VEC>(vw, Wz, g.ldw, n0, g.N, kbeg, kend, tid);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        store_tile<A_KC>(As, va, tid);
        store_tile<W_KC>(Ws, vw, tid);
        __syncthreads();
        if (k0 + BK < kend) {                           // next tile's loads fly during the MFMAs
            load_tile<A_KC, VEC>(va, Az, g.lda, m0, g.M, k0 + BK, kend, tid);
            load_tile<W_KC, VEC>(vw, Wz, g.ldw, n0, g.N, k0 + BK, kend, tid);
        }
        const float* ap = &As[(wm * 32 + li) * LDP + lk];
        const float* wp = &Ws[(wn * 32 + li) * LDP + lk];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], wp[kk], acc, 0, 0, 0);
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < g.M && n < g.N) {
            if (slabs > 1) partial[((size_t)slab * g.M + m) * g.N + n] = acc[r];
            else epilogue_store(g, m, n, acc[r]);
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g, const float* __restrict__ partial,
                                                            int slabs) {
    const size_t total = (size_t)g.M * g.N;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / g.N), n = (int)(i - (size_t)m * g.N);
        float v = 0.f;
        for (int s = 0; s < slabs; ++s) v += partial[(size_t)s * total + i];   // fixed slab order
        epilogue_store(g, m, n, v);
    }
}

void plan(int M, int N, int K, int nz, int* ksplits, int* k_chunk) {
    const int tiles = i2l_cdiv(M, BM) * i2l_cdiv(N, BN) * nz;
    int s = 1;
    if (tiles < 256 && K >= 1024) {
        s = 512 / tiles;
        const int max_s = K / 512;           // keep >= 512 of K per slice
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
    }
    int kc = i2l_cdiv(i2l_cdiv(K, s), BK) * BK;
    s = i2l_cdiv(K, kc);
    *ksplits = s;
    *k_chunk = kc;
}

template <bool A_KC, bool W_KC>
void launch(const GemmArgs& g, float* ws, int kc, int ks, int slabs, bool vec, dim3 grid, hipStream_t s) {
    if (vec) hipLaunchKernelGGL((gemm_kernel<A_KC, W_KC, true>), grid, dim3(256), 0, s, g, ws, kc, ks, slabs);
    else hipLaunchKernelGGL((gemm_kernel<A_KC, W_KC, false>), grid, dim3(256), 0, s, g, ws, kc, ks, slabs);
}

}  // namespace

size_t i2l_gemm_workspace_bytes(int M, int N, int K, int nz) {
    int ks, kc;
    plan(M, N, K, nz < 1 ? 1 : nz, &ks, &kc);
    const size_t slabs = (size_t)ks * (nz < 1 ? 1 : nz);
    return slabs > 1 ? i2l_align(slabs * M * N * sizeof(float)) : 0;
}

int i2l_gemm(const GemmArgs& g0, void* ws, size_t ws_bytes, hipStream_t stream) {
    GemmArgs g = g0;
    if (!g.A || !g.W || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return I2L_ERR_ARG;
    if (g.nz < 1) g.nz = 1;
    int ks, kc;
    plan(g.M, g.N, g.K, g.nz, &ks, &kc);
    const int slabs = ks * g.nz;
    if (slabs > 65535) return I2L_ERR_UNSUPPORTED;
    if (slabs > 1 && (!ws || ws_bytes < (size_t)slabs * g.M * g.N * sizeof(float))) return I2L_ERR_WORKSPACE;
    auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    const bool vec = al16(g.A) && al16(g.W) && g.lda % 4 == 0 && g.ldw % 4 == 0 && g.bsa % 4 == 0 && g.bsw % 4 == 0 &&
                     (g.a_kc && g.w_kc ? g.K % 4 == 0 : true);
    dim3 grid(i2l_cdiv(g.N, BN), i2l_cdiv(g.M, BM), slabs);
    float* wsf = static_cast<float*>(ws);
    if (g.a_kc && g.w_kc) launch<true, true>(g, wsf, kc, ks, slabs, vec, grid, stream);
    else if (g.a_kc) launch<true, false>(g, wsf, kc, ks, slabs, vec, grid, stream);
    else if (g.w_kc) launch<false, true>(g, wsf, kc, ks, slabs, vec, grid, stream);
    else launch<false, false>(g, wsf, kc, ks, slabs, vec, grid, stream);
    I2L_CHECK_LAUNCH();
    if (slabs > 1) {
        const size_t total = (size_t)g.M * g.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g, (const float*)ws, slabs);
        I2L_CHECK_LAUNCH();
    }
    return I2L_OK;
}

extern "C" size_t i2l_linear_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return i2l_gemm_workspace_bytes(M, N, K, 1);
}

extern "C" int i2l_linear_bias_act_fwd(const float* x, const float* w, const float* bias, float* y,
                                        int M, int K, int N, int relu, void* workspace,
                                        size_t workspace_bytes, i2l_stream_t stream) {
    if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0) return I2L_ERR_ARG;
    GemmArgs g = gemm_args();
    g.A = x; g.lda = K;
    g.W = w; g.ldw = K;
    g.bias = bias;
    g.C = y; g.ldc = N;
    g.M = M; g.N = N; g.K = K;
    g.relu = relu ? 1 : 0;
    return i2l_gemm(g, workspace, workspace_bytes, i2l_s(stream));
}

// ---------------------------------------------------------------- backward of y = act(x W^T + b)
namespace {
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = y[i] > 0.f ? dy[i] : 0.f;
}
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ A, int M, int N,
                                                           float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += A[(size_t)m * N + n];
    out[n] = s;
}
size_t lin_bwd_gemm_ws(int M, int K, int N) {
    const size_t a = i2l_gemm_workspace_bytes(N, K, M), b = i2l_gemm_workspace_bytes(M, K, N);
    return i2l_align(a > b ? a : b);
}
}  // namespace

extern "C" size_t i2l_linear_bwd_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return i2l_align((size_t)M * N * sizeof(float)) + lin_bwd_gemm_ws(M, K, N);
}

extern "C" int i2l_linear_bias_act_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx,
                                        float* dw, float* db, int M, int K, int N, int relu, void* workspace,
                                        size_t workspace_bytes, i2l_stream_t stream) {
    if (!x || !w || !dy || !dw || !db || (relu && !y) || M <= 0 || K <= 0 || N <= 0) return I2L_ERR_ARG;
    if (!workspace || workspace_bytes < i2l_linear_bwd_workspace_bytes(M, K, N)) return I2L_ERR_WORKSPACE;
    hipStream_t s = i2l_s(stream);
    float* dpre = static_cast<float*>(workspace);
    char* gws = static_cast<char*>(workspace) + i2l_align((size_t)M * N * sizeof(float));
    const size_t gws_bytes = lin_bwd_gemm_ws(M, K, N);
    const float* d = dy;
    if (relu) {
        const size_t n = (size_t)M * N;
        size_t blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, y, dpre, n);
        I2L_CHECK_LAUNCH();
        d = dpre;
    }
    hipLaunchKernelGGL(colsum_small_kernel, dim3(i2l_cdiv(N, 256)), dim3(256), 0, s, d, M, N, db);
    I2L_CHECK_LAUNCH();
    {   // dw[n][k] = sum_m d[m][n] * x[m][k]
        GemmArgs g = gemm_args();
        g.A = d; g.lda = N; g.a_kc = 0;
        g.W = x; g.ldw = K; g.w_kc = 0;
        g.C = dw; g.ldc = K;
        g.M = N; g.N = K; g.K = M;
        const int rc = i2l_gemm(g, gws, gws_bytes, s);
        if (rc != I2L_OK) return rc;
    }
    if (dx) {   // dx[m][k] = sum_n d[m][n] * w[n][k]
        GemmArgs g = gemm_args();
        g.A = d; g.lda = N;
        g.W = w; g.ldw = K; g.w_kc = 0;
        g.C = dx; g.ldc = K;
        g.M = M; g.N = K; g.K = N;
        const int rc = i2l_gemm(g, gws, gws_bytes, s);
        if (rc != I2L_OK) return rc;
    }
    return I2L_OK;
}
