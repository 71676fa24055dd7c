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
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 64, LDP = BK + 1;
constexpr int NLD = BM * BK / 4 / 256;      // float4 loads per thread per operand tile
constexpr int KQ = BK / 4;                  // float4 chunks along k
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Split-K slabs are written by concurrent workgroups and summed slab by slab at the same (m, n): a power-of-two slab
// stride (256 x 256 floats = 256 KB) puts all of those accesses on one memory channel (measured: 16.7 MB of slab
// writes took 42 us).  The stride is therefore an ODD number of 256-byte units.
__host__ __device__ __forceinline__ size_t slab_stride(int M, int N) {
    size_t units = ((size_t)M * N + 63) / 64;               // 256-byte units
    if ((units & 1) == 0) ++units;
    return units * 64;
}

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
            if (slabs > 1) partial[(size_t)slab * slab_stride(g.M, g.N) + (size_t)m * g.N + n] = acc[r];
            else epilogue_store(g, m, n, acc[r]);
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g, const float* __restrict__ partial,
                                                            int slabs) {
    const size_t total = (size_t)g.M * g.N, stride = slab_stride(g.M, g.N);
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / g.N), n = (int)(i - (size_t)m * g.N);
        float v = 0.f;
        int s = 0;
        for (; s + 8 <= slabs; s += 8) {                    // 8 loads in flight, added in slab order
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = partial[(size_t)(s + j) * stride + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) v += t[j];
        }
        for (; s < slabs; ++s) v += partial[(size_t)s * stride + i];
        epilogue_store(g, m, n, v);
    }
}

// Few outputs, many slabs (the conv weight gradients: 864 .. 73,728 outputs, hundreds of slabs): 8 lanes per output
// add the slabs r, r + 8, ... and their sums meet in lane order (fixed order: deterministic); one thread per output
// would walk the slabs alone on a handful of CUs.
__global__ __launch_bounds__(256) void splitk_reduce_lanes_kernel(GemmArgs g, const float* __restrict__ partial,
                                                                  int slabs) {
    __shared__ float sm[8][32];
    const size_t total = (size_t)g.M * g.N, stride = slab_stride(g.M, g.N);
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + c;
    float v = 0.f;
    if (i < total) {
        int s = r;
        for (; s + 24 < slabs; s += 32) {
            const float t0 = partial[(size_t)s * stride + i], t1 = partial[(size_t)(s + 8) * stride + i];
            const float t2 = partial[(size_t)(s + 16) * stride + i], t3 = partial[(size_t)(s + 24) * stride + i];
            v += t0; v += t1; v += t2; v += t3;
        }
        for (; s < slabs; s += 8) v += partial[(size_t)s * stride + i];
    }
    sm[r][c] = v;
    __syncthreads();
    if (r == 0 && i < total) {
        float t = sm[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sm[k][c];
        epilogue_store(g, (int)(i / g.N), (int)(i % g.N), t);
    }
}
void launch_splitk_reduce(const GemmArgs& g, const float* ws, int slabs, hipStream_t s) {
    const size_t total = (size_t)g.M * g.N;
    if (g.conv_h > 0 && total <= 131072 && slabs >= 32) {   // the conv weight gradients only: other sums keep their order
        hipLaunchKernelGGL(splitk_reduce_lanes_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, g, ws, slabs);
        return;
    }
    int rb = (int)((total + 255) / 256);
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb), dim3(256), 0, s, g, ws, slabs);
}

void plan(int M, int N, int K, int nz, int* ksplits, int* k_chunk) {
    const int tiles = i2l_cdiv(M, BM) * i2l_cdiv(N, BN) * nz;
    int s = 1;
    if (tiles < 256 && K >= 1024) {
        s = 512 / tiles;
        const int max_s = K / 512;           // keep >= 512 of K per slice
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
    } else if (tiles <= 64 && K >= 256) {    // a quarter of the chip or less (the decoder's per-batch Genc GEMM: 64 tiles,
        s = 256 / tiles;                     // K = 256): one 64-deep slice per workgroup fills it, the slab sum costs ~5 us
        const int max_s = K / 64;
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


// ---------------------------------------------------------------------------------------------------------------
// y = act(x W^T + b) with a long reduction (the encoder's Flatten -> Linear: K = 40,960) on the bf16 matrix cores at
// fp32-grade accuracy: both operands are split into 3 bf16 pieces on their way to LDS (x = x0 + x1 + x2, exact) and
// a product is the six partial products a_i b_j, i + j <= 2, on v_mfma_f32_32x32x16_bf16 -- see conv_bf16x3.hip.
// Tile 128 x 64 x 32 per 256-thread workgroup (waves 2 x 2, each 64 x 32 = 2 M-tiles), split-K into fp32 slabs that
// splitk_reduce_kernel sums in slab order.  LDS image per operand: [split][k group of 8][row][8 k] in 16-byte pieces,
// plane stride = rows + 2 pieces, so both the staging writes (4 lanes = the 4 k groups of one row) and the MFMA
// operand reads (32 lanes = 32 consecutive rows) are conflict-free.  Workgroups that share a K slice (same operand
// bytes) get block ids that are equal mod 8: one XCD, one L2 (speed only).
// ---------------------------------------------------------------------------------------------------------------
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v2_t __attribute__((ext_vector_type(2)));
typedef float f32v2_t __attribute__((ext_vector_type(2)));
constexpr int LBN = 64, LBK = 32, LPW = LBN + 2;

__device__ __forceinline__ void lsplit3_pair(float x0, float x1, unsigned& s0, unsigned& s1, unsigned& s2) {
    f32v2_t v = {x0, x1};
    bf16v2_t b0 = __builtin_convertvector(v, bf16v2_t);
    v -= __builtin_convertvector(b0, f32v2_t);
    bf16v2_t b1 = __builtin_convertvector(v, bf16v2_t);
    v -= __builtin_convertvector(b1, f32v2_t);
    bf16v2_t b2 = __builtin_convertvector(v, bf16v2_t);
    s0 = *reinterpret_cast<unsigned*>(&b0);
    s1 = *reinterpret_cast<unsigned*>(&b1);
    s2 = *reinterpret_cast<unsigned*>(&b2);
}
// 8 consecutive k of one row -> one 16-byte piece per split
__device__ __forceinline__ void lsplit_piece(const float4& lo, const float4& hi, uint4 (&out)[3]) {
    unsigned p[3][4];
    lsplit3_pair(lo.x, lo.y, p[0][0], p[1][0], p[2][0]);
    lsplit3_pair(lo.z, lo.w, p[0][1], p[1][1], p[2][1]);
    lsplit3_pair(hi.x, hi.y, p[0][2], p[1][2], p[2][2]);
    lsplit3_pair(hi.z, hi.w, p[0][3], p[1][3], p[2][3]);
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) out[sp] = make_uint4(p[sp][0], p[sp][1], p[sp][2], p[sp][3]);
}

// MT = 32-row MFMA tiles per wave: workgroup tile (64 MT) x 64.  PLAIN: one operand pair, no gathered / un-pooled
// operand (the decoder's and the FC layer's GEMMs): the fetch is two 16-byte loads per piece, no index arithmetic
template <bool A_KC, bool W_KC, int MT, bool PLAIN>
__global__ __launch_bounds__(256, 3) void gemm_bf16x3_kernel(GemmArgs g, float* __restrict__ partial, int k_chunk,
                                                             int S, int tiles_n, int tiles) {
    constexpr int LBM_ = 64 * MT, LPA_ = LBM_ + 2;
    __shared__ __attribute__((aligned(16))) uint4 a_s[3 * 4 * LPA_];
    __shared__ __attribute__((aligned(16))) uint4 w_s[3 * 4 * LPW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    // K slices a multiple of 8: workgroups that share a slice (same operand bytes) get block ids equal mod 8 = one XCD
    // under round-robin placement (speed only); otherwise the plain order (every XCD gets work)
    const int L = blockIdx.x;
    int tile, slice;
    if ((S & 7) == 0) { const int grp = L >> 3; tile = grp % tiles; slice = (L & 7) + 8 * (grp / tiles); }
    else if (S == 1 && (tiles & 7) == 0) { tile = (L & 7) * (tiles >> 3) + (L >> 3); slice = 0; }   // an XCD walks its own rows of A:
    else { tile = L % tiles; slice = L / tiles; }                                                   // every N tile of a row block in one L2 (r04: -1.2 % on the training step)
    if (slice >= S) return;
    const int m0 = (tile / tiles_n) * LBM_, n0 = (tile % tiles_n) * LBN;
    // the reduction runs over the flattened (operand pair z, k) range; K % 32 == 0 when nz > 1, so a chunk never
    // straddles two pairs
    const long kf_beg = (long)slice * k_chunk, kf_end = min((long)g.nz * g.K, kf_beg + k_chunk);

    // staging: piece = (row, k group kg of 8); A: 2 pieces per thread, W: 1.  K-contiguous operands: 4 lanes cover the
    // 4 k groups of one row (two float4 loads each); M/N-contiguous operands: consecutive lanes = consecutive rows
    // (8 strided 4-byte loads each, coalesced across lanes).  Either way a piece lands at (split, kg) plane + row.
    const int a_row0 = A_KC ? (tid >> 2) : (MT == 2 ? (tid & 127) : (tid & 63)), a_kg0 = A_KC ? (tid & 3) : (MT == 2 ? (tid >> 7) : (tid >> 6));
    const int a_row1 = A_KC ? 64 + (tid >> 2) : (tid & 127), a_kg1 = A_KC ? (tid & 3) : 2 + (tid >> 7);
    const int w_row = W_KC ? (tid >> 2) : (tid & 63), w_kg = W_KC ? (tid & 3) : (tid >> 6);
    const bool oka0 = m0 + a_row0 < g.M, oka1 = MT == 2 && m0 + a_row1 < g.M, okw = n0 + w_row < g.N;
    const float* pa0 = A_KC ? g.A + (size_t)min(m0 + a_row0, g.M - 1) * g.lda : g.A + min(m0 + a_row0, g.M - 1);
    const float* pa1 = A_KC ? g.A + (size_t)min(m0 + a_row1, g.M - 1) * g.lda : g.A + min(m0 + a_row1, g.M - 1);
    const float* pw = W_KC ? g.W + (size_t)min(n0 + w_row, g.N - 1) * g.ldw : g.W + min(n0 + w_row, g.N - 1);
    // two register stages, prefetch distance two: a stage's registers are free as soon as commit() has copied them to
    // LDS, so the loads of chunk c+2 go straight back into the stage of chunk c (one chunk of MFMAs is only ~0.4 us,
    // shorter than a memory round trip; a third stage cost 24 registers and three spills in the widest variant)
    struct Stage { float4 a[2][2], w[2]; };
    Stage st0, st1;
    auto load8 = [&](bool kc, const float* base, long ld, bool ok, int k, int kend, float4& lo, float4& hi) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        lo = z; hi = z;
        if (!ok) return;
        if (kc) {                                            // K % 8 == 0: a piece is inside or outside as a whole
            if (k < kend) {
                lo = *reinterpret_cast<const float4*>(base + k);
                hi = *reinterpret_cast<const float4*>(base + k + 4);
            }
        } else {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = k + j < kend ? base[(size_t)(k + j) * ld] : 0.f;
            lo = make_float4(v[0], v[1], v[2], v[3]);
            hi = make_float4(v[4], v[5], v[6], v[7]);
        }
    };
    auto fetch = [&](Stage& st, long kf0) {
        if constexpr (PLAIN) {
            const int k0 = (int)kf0, kend = (int)kf_end;
            load8(A_KC, pa0, g.lda, oka0, k0 + a_kg0 * 8, kend, st.a[0][0], st.a[0][1]);
            if (MT == 2) load8(A_KC, pa1, g.lda, oka1, k0 + a_kg1 * 8, kend, st.a[1][0], st.a[1][1]);
            load8(W_KC, pw, g.ldw, okw, k0 + w_kg * 8, kend, st.w[0], st.w[1]);
            return;
        }
        const int z = (int)(kf0 / g.K), k0 = (int)(kf0 - (long)z * g.K);
        const int kend = (int)min((long)g.K, kf_end - (long)z * g.K);
        const size_t oa = (size_t)z * g.bsa, ow = (size_t)z * g.bsw;
        if (A_KC && g.pool_y) {
            // un-pool on the fly: 8 consecutive positions of one image row <- 4 pooled windows (dy, y, arg max)
            auto unpool8 = [&](int row, bool ok, int k, float4& lo, float4& hi) {
                const float4 zz4 = make_float4(0.f, 0.f, 0.f, 0.f);
                lo = zz4; hi = zz4;
                if (!ok || k >= kend) return;
                const int yy = k / g.conv_w, xx0 = k - yy * g.conv_w;
                const size_t o = (size_t)z * g.bsa + (size_t)min(m0 + row, g.M - 1) * g.lda + (size_t)(yy >> 1) * (g.conv_w >> 1) + (xx0 >> 1);
                const float4 d4 = *reinterpret_cast<const float4*>(g.A + o);
                const float4 y4 = *reinterpret_cast<const float4*>(g.pool_y + o);
                const unsigned a4 = *reinterpret_cast<const unsigned*>(g.pool_am + o);
                const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int pc = j >> 1;
                    const unsigned am = (a4 >> (8 * pc)) & 0xFFu;
                    v[j] = (yv[pc] > 0.f && am == (unsigned)(2 * (yy & 1) + (j & 1))) ? dv[pc] : 0.f;
                }
                lo = make_float4(v[0], v[1], v[2], v[3]);
                hi = make_float4(v[4], v[5], v[6], v[7]);
            };
            unpool8(a_row0, oka0, k0 + a_kg0 * 8, st.a[0][0], st.a[0][1]);
            if (MT == 2) unpool8(a_row1, oka1, k0 + a_kg1 * 8, st.a[1][0], st.a[1][1]);
        } else {
        load8(A_KC, pa0 + oa, g.lda, oka0, k0 + a_kg0 * 8, kend, st.a[0][0], st.a[0][1]);
        if (MT == 2) load8(A_KC, pa1 + oa, g.lda, oka1, k0 + a_kg1 * 8, kend, st.a[1][0], st.a[1][1]);
        }
        if (g.conv_h > 0) {
            // implicit im2col^T: row n = n0 + w_row = (ci, tap), 8 consecutive positions k of image z
            const float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
            st.w[0] = zz; st.w[1] = zz;
            const int n = n0 + w_row, kk = k0 + w_kg * 8;
            if (okw && kk < kend) {
                const int ci = n / 9, tap = n - 9 * ci, dyy = tap / 3 - 1, dxx = tap - 3 * (tap / 3) - 1;
                const float* img = g.W + ow + (size_t)ci * g.conv_h * g.conv_w;
                float v[8];
                if ((g.conv_w & 7) == 0 && kk + 8 <= kend) {
                    // the 8 positions share one image row (kk % 8 == 0): one divide, one row test, and away from the
                    // left / right edge two 16-byte loads (4-byte aligned) instead of eight scalar ones
                    const int yy = kk / g.conv_w, xx0 = kk - yy * g.conv_w;
                    const int iy = yy + dyy;
                    if (iy >= 0 && iy < g.conv_h) {
                        const float* rowp = img + (size_t)iy * g.conv_w + xx0 + dxx;
                        if (xx0 + dxx >= 0 && xx0 + dxx + 8 <= g.conv_w) {
                            __builtin_memcpy(&st.w[0], rowp, 16);
                            __builtin_memcpy(&st.w[1], rowp + 4, 16);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int ix = xx0 + dxx + j;
                                v[j] = (ix >= 0 && ix < g.conv_w) ? rowp[j] : 0.f;
                            }
                            st.w[0] = make_float4(v[0], v[1], v[2], v[3]);
                            st.w[1] = make_float4(v[4], v[5], v[6], v[7]);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int pos = kk + j, yy = pos / g.conv_w, xx = pos - yy * g.conv_w;
                        const int iy = yy + dyy, ix = xx + dxx;
                        v[j] = (pos < kend && iy >= 0 && iy < g.conv_h && ix >= 0 && ix < g.conv_w) ? img[(size_t)iy * g.conv_w + ix] : 0.f;
                    }
                    st.w[0] = make_float4(v[0], v[1], v[2], v[3]);
                    st.w[1] = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
        } else {
            load8(W_KC, pw + ow, g.ldw, okw, k0 + w_kg * 8, kend, st.w[0], st.w[1]);
        }
    };
    auto commit = [&](const Stage& st) {
        uint4 o[3];
        lsplit_piece(st.a[0][0], st.a[0][1], o);
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) a_s[(sp * 4 + a_kg0) * LPA_ + a_row0] = o[sp];
        if (MT == 2) {
            lsplit_piece(st.a[1][0], st.a[1][1], o);
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) a_s[(sp * 4 + a_kg1) * LPA_ + a_row1] = o[sp];
        }
        lsplit_piece(st.w[0], st.w[1], o);
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) w_s[(sp * 4 + w_kg) * LPW + w_row] = o[sp];
    };

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    auto step = [&](Stage& cur, long k0) {
        commit(cur);
        __syncthreads();
        if (k0 + 2 * LBK < kf_end) fetch(cur, k0 + 2 * LBK);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t a[MT][3], b[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                const int plane = sp * 4 + s * 2 + h;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const uint4 t = a_s[plane * LPA_ + wm * 32 * MT + m * 32 + i];
                    a[m][sp] = *reinterpret_cast<const bf16x8_t*>(&t);
                }
                const uint4 t = w_s[plane * LPW + wn * 32 + i];
                b[sp] = *reinterpret_cast<const bf16x8_t*>(&t);
            }
            constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][TI[t]], b[TJ[t]], acc[m], 0, 0, 0);
        }
        __syncthreads();
    };
    fetch(st0, kf_beg);
    if (kf_beg + LBK < kf_end) fetch(st1, kf_beg + LBK);
    for (long k0 = kf_beg; k0 < kf_end; k0 += 2 * LBK) {
        step(st0, k0);
        if (k0 + LBK < kf_end) step(st1, k0 + LBK);
    }
    const int n = n0 + wn * 32 + i;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 * MT + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < g.M && n < g.N) {
                if (S > 1) partial[(size_t)slice * slab_stride(g.M, g.N) + (size_t)row * g.N + n] = acc[m][r];
                else epilogue_store(g, row, n, acc[m][r]);
            }
        }
}

// 64-row workgroup tiles for M <= 64 (the weight gradients of the first two conv blocks, the FC layer of a 64-image
// shard): half of a 128-row tile would be padding
int bf16x3_mt(int M) { return M <= 64 ? 1 : 2; }

void plan_bf16x3(int M, int N, long KT, int* S, int* k_chunk) {      // KT = nz * K, the flattened reduction length
    const int tiles = i2l_cdiv(M, 64 * bf16x3_mt(M)) * i2l_cdiv(N, LBN);
    // three workgroups per CU are resident (768 slots); outputs of 64 K elements and more pay too much for the extra
    // slabs (the encoder's FC layer: 55 -> 58 us at 96 slabs), they keep the 512-slot split
    long s = ((long)M * N >= 65536 ? 512 : 768) / tiles;
    const long max_s = KT / 256;              // keep >= 256 of the reduction per slice
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    const long kc = ((KT + s - 1) / s + LBK - 1) / LBK * LBK;
    *S = (int)((KT + kc - 1) / kc);
    *k_chunk = (int)kc;
}

// can this GEMM run on gemm_bf16x3_kernel?  (single operand pair, K-contiguous operands 16-byte aligned with K % 8 == 0)
bool bf16x3_applicable(const GemmArgs& g) {
    if (g.K < 64 || (g.nz > 1 && (g.K % LBK != 0 || (long)g.nz * g.K > 0x7fffffffl))) return false;
    auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (g.pool_y) {
        if (!(g.conv_h > 0 && g.a_kc && g.pool_am && (g.conv_w % 8) == 0 && (g.conv_h % 2) == 0 && al16(g.A) && al16(g.pool_y) &&
              reinterpret_cast<uintptr_t>(g.pool_am) % 4 == 0 && g.lda % 4 == 0 && g.bsa % 4 == 0))
            return false;
    } else if (g.a_kc && (!al16(g.A) || g.lda % 4 != 0 || g.K % 8 != 0)) return false;
    if (g.conv_h > 0) return g.w_kc && g.K == g.conv_h * g.conv_w;     // gathered W: no alignment requirement
    if (g.w_kc && (!al16(g.W) || g.ldw % 4 != 0 || g.K % 8 != 0)) return false;
    return true;
}

int run_bf16x3(const GemmArgs& g, void* ws, size_t ws_bytes, hipStream_t s) {
    int S, kc;
    plan_bf16x3(g.M, g.N, (long)g.nz * g.K, &S, &kc);
    if (S > 1 && (!ws || ws_bytes < (size_t)S * slab_stride(g.M, g.N) * sizeof(float))) return I2L_ERR_WORKSPACE;
    const int mt = bf16x3_mt(g.M);
    const int tiles_n = i2l_cdiv(g.N, LBN), tiles = i2l_cdiv(g.M, 64 * mt) * tiles_n;
    const long long blocks = (long long)tiles * S;
    if (blocks > 0x7fffffffll) return I2L_ERR_UNSUPPORTED;
    dim3 grid((unsigned)blocks);
    float* wsf = static_cast<float*>(ws);
    const bool plain = g.nz == 1 && !g.pool_y && g.conv_h <= 0;
#define I2L_LAUNCH_BF16X3(AK, WK)                                                                                          \
    do {                                                                                                                \
        if (mt == 2 && plain) hipLaunchKernelGGL((gemm_bf16x3_kernel<AK, WK, 2, true>), grid, dim3(256), 0, s, g, wsf, kc, S, tiles_n, tiles); \
        else if (mt == 2) hipLaunchKernelGGL((gemm_bf16x3_kernel<AK, WK, 2, false>), grid, dim3(256), 0, s, g, wsf, kc, S, tiles_n, tiles); \
        else if (plain) hipLaunchKernelGGL((gemm_bf16x3_kernel<AK, WK, 1, true>), grid, dim3(256), 0, s, g, wsf, kc, S, tiles_n, tiles); \
        else hipLaunchKernelGGL((gemm_bf16x3_kernel<AK, WK, 1, false>), grid, dim3(256), 0, s, g, wsf, kc, S, tiles_n, tiles);  \
    } while (0)
    if (g.a_kc && g.w_kc) I2L_LAUNCH_BF16X3(true, true);
    else if (g.a_kc) I2L_LAUNCH_BF16X3(true, false);
    else if (g.w_kc) I2L_LAUNCH_BF16X3(false, true);
    else I2L_LAUNCH_BF16X3(false, false);
#undef I2L_LAUNCH_BF16X3
    I2L_CHECK_LAUNCH();
    if (S > 1) {
        launch_splitk_reduce(g, (const float*)ws, S, s);
        I2L_CHECK_LAUNCH();
    }
    return I2L_OK;
}

}  // namespace

bool i2l_gemm_split_bf16_ok(const GemmArgs& g0) {
    GemmArgs g = g0;
    if (g.nz < 1) g.nz = 1;
    return g.split_bf16 && bf16x3_applicable(g);
}

size_t i2l_gemm_workspace_bytes(int M, int N, int K, int nz) {
    int ks, kc;
    plan(M, N, K, nz < 1 ? 1 : nz, &ks, &kc);
    const size_t slabs = (size_t)ks * (nz < 1 ? 1 : nz);
    size_t need = slabs > 1 ? i2l_align(slabs * slab_stride(M, N) * sizeof(float)) : 0;
    if (K >= 64) {                             // the split-bf16 kernel's slabs (GemmArgs::split_bf16)
        int S, kc;
        plan_bf16x3(M, N, (long)(nz < 1 ? 1 : nz) * K, &S, &kc);
        const size_t n3 = S > 1 ? i2l_align((size_t)S * slab_stride(M, N) * sizeof(float)) : 0;
        if (n3 > need) need = n3;
    }
    return need;
}

int i2l_gemm(const GemmArgs& g0, void* ws, size_t ws_bytes, hipStream_t stream) {
    GemmArgs g = g0;
    if (!g.A || !g.W || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return I2L_ERR_ARG;
    if (g.nz < 1) g.nz = 1;
    if (g.split_bf16 && bf16x3_applicable(g)) return run_bf16x3(g, ws, ws_bytes, stream);
    int ks, kc;
    plan(g.M, g.N, g.K, g.nz, &ks, &kc);
    const int slabs = ks * g.nz;
    if (slabs > 65535) return I2L_ERR_UNSUPPORTED;
    if (slabs > 1 && (!ws || ws_bytes < (size_t)slabs * slab_stride(g.M, g.N) * sizeof(float))) return I2L_ERR_WORKSPACE;
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
        launch_splitk_reduce(g, (const float*)ws, slabs, stream);
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
                                        size_t workspace_bytes, int flags, i2l_stream_t stream) {
    if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0) return I2L_ERR_ARG;
    GemmArgs g = gemm_args();
    g.A = x; g.lda = K;
    g.W = w; g.ldw = K;
    g.bias = bias;
    g.C = y; g.ldc = N;
    g.M = M; g.N = N; g.K = K;
    g.relu = relu ? 1 : 0;
    // long reductions (the encoder FC): bf16 matrix cores, fp32-grade result
    g.split_bf16 = (K >= 2048 && !(flags & I2L_FLAG_EXACT_FP32)) ? 1 : 0;
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
    __shared__ float sm[8][32];                             // 32 columns x 8 row lanes, lane sums added in lane order
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float s = 0.f;
    if (n < N)
        for (int m = r; m < M; m += 8) s += A[(size_t)m * N + n];
    sm[r][c] = s;
    __syncthreads();
    if (r == 0 && n < N) {
        float t = sm[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sm[k][c];
        out[n] = t;
    }
}
size_t lin_bwd_gemm_ws(int M, int K, int N) {
    const size_t a = i2l_gemm_workspace_bytes(N, K, M), b = i2l_gemm_workspace_bytes(M, K, N);
    return i2l_align(a > b ? a : b);
}
}  // namespace

extern "C" size_t i2l_linear_bwd_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return i2l_align((size_t)M * N * sizeof(float)) + 2 * lin_bwd_gemm_ws(M, K, N);   // one GEMM workspace per stream
}

extern "C" int i2l_linear_bias_act_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx,
                                        float* dw, float* db, int M, int K, int N, int relu, void* workspace,
                                        size_t workspace_bytes, int flags, i2l_lanes* lanes, i2l_stream_t stream) {
    if (!x || !w || !dy || !dw || (relu && !y) || M <= 0 || K <= 0 || N <= 0) return I2L_ERR_ARG;   // db may be NULL (no bias)
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
    // lanes != NULL: db and dw on the caller's side lane 0 (own GEMM workspace), dx stays on the main stream
    hipStream_t s_main = s;
    char* gws_main = gws;
    if (lanes && dx) {
        hipStream_t f = i2l_side_fork(lanes, s, 0);
        if (f) { s = f; gws = gws + gws_bytes; }
    }
    if (db) {
        hipLaunchKernelGGL(colsum_small_kernel, dim3(i2l_cdiv(N, 32)), dim3(256), 0, s, d, M, N, db);
        I2L_CHECK_LAUNCH();
    }
    {   // dw[n][k] = sum_m d[m][n] * x[m][k]
        GemmArgs g = gemm_args();
        g.split_bf16 = (flags & I2L_FLAG_EXACT_FP32) ? 0 : 1;
        g.A = d; g.lda = N; g.a_kc = 0;
        g.W = x; g.ldw = K; g.w_kc = 0;
        g.C = dw; g.ldc = K;
        g.M = N; g.N = K; g.K = M;
        const int rc = i2l_gemm(g, gws, gws_bytes, s);
        if (rc != I2L_OK) return rc;
    }
    s = s_main;
    gws = gws_main;
    if (dx) {   // dx[m][k] = sum_n d[m][n] * w[n][k]
        GemmArgs g = gemm_args();
        g.split_bf16 = (flags & I2L_FLAG_EXACT_FP32) ? 0 : 1;
        g.A = d; g.lda = N;
        g.W = w; g.ldw = K; g.w_kc = 0;
        g.C = dx; g.ldc = K;
        g.M = M; g.N = K; g.K = N;
        const int rc = i2l_gemm(g, gws, gws_bytes, s);
        if (rc != I2L_OK) return rc;
    }
    return I2L_OK;
}
