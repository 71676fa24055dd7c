// bf16 building blocks of the ResNet encoder (reference encoder.py:132-249: torchvision ResNet trunk
// minus fc, run at :242) for inference: activations live in HBM as NHWC bf16, every convolution is a
// GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate) with BatchNorm folded
// into a per-channel scale/bias and the residual add + ReLU fused into the epilogue.
//   1x1 stride-1 convs read the activation tensor directly ([B*H*W][Cin] is already the A matrix);
//   3x3 / 7x7 / strided convs go through an im2col image [M][taps*Cin] (K contiguous).
// Parity status: UNPINNED (torchvision is not installed and the reference fetches remote weights;
// SURVEY.md 8c) -- tests compare against this repo's own fp32 restatement of the v1.5 architecture.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);          // round to nearest even (finite inputs)
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }

// ------------------------------------------------------------------ bf16 GEMM with fused epilogue
// out[m][n] = act( acc[m][n] * scale[n] + bias[n] + res[m][n] ),  acc = sum_k A[m][k] * W[n][k]
constexpr int GM = 128, GK = 64, GLD = 72;      // LDS row stride 72 bf16 = 144 B: conflict-free b128 reads
constexpr int GNL = GM * GK / 8 / 256;                     // 16-byte loads per thread per operand tile

struct BfGemm {
    // conv != 0: A is never materialised -- row m = (b, yo, xo) and k = (tap, ci) index the NHWC activation
    // tensor directly (implicit GEMM; Cin % 64 == 0 so a 64-wide K tile lies inside one tap)
    int conv, iH, iW, Cin, Ho, Wo, kw, stride, pad;
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    const float* scale; const float* bias;
    const bf16_t* res; long ldr;
    bf16_t* C; long ldc;
    int M, N, K, relu;
};

template <int NTW>      // 32-column MFMA tiles per wave: workgroup tile = 128 x (64*NTW)
__global__ __launch_bounds__(256) void gemm_bf16_kernel(BfGemm g) {
    constexpr int GN = 64 * NTW;
    constexpr int GNLW = GN * GK / 8 / 256;
    __shared__ __attribute__((aligned(16))) bf16_t smem_ab[(GM + GN) * GLD];
    bf16_t* As = smem_ab;
    bf16_t* Ws = smem_ab + GM * GLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * GN, m0 = blockIdx.y * GM;

    f32x16 acc[2][NTW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    uint4 ra[GNL], rw[GNLW];
    // implicit-GEMM rows: output pixel of each of this thread's GNL staging rows
    int pb[GNL], py[GNL], px[GNL];
    if (g.conv) {
#pragma unroll
        for (int i = 0; i < GNL; ++i) {
            const int m = min(m0 + (tid + i * 256) / (GK / 8), g.M - 1);
            const int xo = m % g.Wo, t = m / g.Wo;
            pb[i] = t / g.Ho;
            py[i] = (t - pb[i] * g.Ho) * g.stride - g.pad;
            px[i] = xo * g.stride - g.pad;
        }
    }
    auto fetch = [&](int k0) {                             // global -> registers
        int ky = 0, kx = 0, c0 = 0;
        if (g.conv) {
            const int tap = k0 / g.Cin;
            c0 = k0 - tap * g.Cin;
            ky = tap / g.kw;
            kx = tap - ky * g.kw;
        }
#pragma unroll
        for (int i = 0; i < GNL; ++i) {
            const int idx = tid + i * 256;                 // 128 rows x (GK/8) chunks of 8 bf16
            const int row = idx / (GK / 8), c8 = (idx % (GK / 8)) * 8;
            ra[i] = make_uint4(0, 0, 0, 0);
            if (g.conv) {
                const int yi = py[i] + ky, xi = px[i] + kx;
                if (m0 + row < g.M && yi >= 0 && yi < g.iH && xi >= 0 && xi < g.iW)
                    ra[i] = *reinterpret_cast<const uint4*>(g.A + (((size_t)pb[i] * g.iH + yi) * g.iW + xi) * g.Cin + c0 + c8);
            } else if (k0 + c8 < g.K && m0 + row < g.M) {  // K % 8 == 0
                ra[i] = *reinterpret_cast<const uint4*>(g.A + (size_t)(m0 + row) * g.lda + k0 + c8);
            }
        }
#pragma unroll
        for (int i = 0; i < GNLW; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (GK / 8), c8 = (idx % (GK / 8)) * 8;
            rw[i] = make_uint4(0, 0, 0, 0);
            if (k0 + c8 < g.K && n0 + row < g.N)
                rw[i] = *reinterpret_cast<const uint4*>(g.W + (size_t)(n0 + row) * g.ldw + k0 + c8);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < g.K; k0 += GK) {
#pragma unroll
        for (int i = 0; i < GNL; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (GK / 8), c8 = (idx % (GK / 8)) * 8;
            *reinterpret_cast<uint4*>(&As[row * GLD + c8]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < GNLW; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (GK / 8), c8 = (idx % (GK / 8)) * 8;
            *reinterpret_cast<uint4*>(&Ws[row * GLD + c8]) = rw[i];
        }
        __syncthreads();
        if (k0 + GK < g.K) fetch(k0 + GK);                 // next tile in flight during the MFMAs
#pragma unroll
        for (int ks = 0; ks < GK; ks += 16) {
            bf16x8 a[2], b[NTW];
#pragma unroll
            for (int t = 0; t < 2; ++t)
                a[t] = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + t * 32 + li) * GLD + ks + 8 * lh]);
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                b[t] = *reinterpret_cast<const bf16x8*>(&Ws[(wn * 32 * NTW + t * 32 + li) * GLD + ks + 8 * lh]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
    // Epilogue.  The activations are far larger than the weights, so most layers are bound by writing C (and
    // reading the residual): stage the scaled tile in LDS as bf16 and move 16 bytes per lane, whole rows at a time.
    // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    constexpr int CLD = GN + 8;                               // bf16 row stride of the staged tile (16-byte aligned rows)
    bf16_t* Cs = As;                                          // As and Ws are adjacent: 128 x CLD bf16 fit in them
    static_assert(GM * CLD <= GM * GLD + 64 * NTW * GLD, "staged C tile must fit in the operand tiles");
    if (g.N % 8 == 0 && g.ldc % 8 == 0 && (g.res == nullptr || g.ldr % 8 == 0)) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int nl = wn * 32 * NTW + nt * 32 + li;
            const int n = n0 + nl;
            const float sc = (g.scale && n < g.N) ? g.scale[n] : 1.f, bi = (g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    // without a residual the value is final: apply ReLU before the single rounding to bf16
                    float v = acc[mt][nt][r] * sc + bi;
                    if (g.relu && !g.res) v = fmaxf(v, 0.f);
                    Cs[ml * CLD + nl] = f2bf(v);
                }
        }
        __syncthreads();
        constexpr int CHUNKS = GN / 8;                        // 16-byte chunks per row
        for (int idx = tid; idx < GM * CHUNKS; idx += 256) {
            const int ml = idx / CHUNKS, c8 = (idx - ml * CHUNKS) * 8;
            const int m = m0 + ml, n = n0 + c8;
            if (m >= g.M || n >= g.N) continue;
            uint4 v = *reinterpret_cast<const uint4*>(&Cs[ml * CLD + c8]);
            if (g.res) {
                const uint4 rr = *reinterpret_cast<const uint4*>(g.res + (size_t)m * g.ldr + n);
                bf16_t* hv = reinterpret_cast<bf16_t*>(&v);
                const bf16_t* hr = reinterpret_cast<const bf16_t*>(&rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = bf2f(hv[e]) + bf2f(hr[e]);
                    if (g.relu) f = fmaxf(f, 0.f);
                    hv[e] = f2bf(f);
                }
            }
            *reinterpret_cast<uint4*>(g.C + (size_t)m * g.ldc + n) = v;
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = n0 + wn * 32 * NTW + nt * 32 + li;
        if (n >= g.N) continue;
        const float sc = g.scale ? g.scale[n] : 1.f, bi = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= g.M) continue;
                float v = acc[mt][nt][r] * sc + bi;
                if (g.res) v += bf2f(g.res[(size_t)m * g.ldr + n]);
                if (g.relu) v = fmaxf(v, 0.f);
                g.C[(size_t)m * g.ldc + n] = f2bf(v);
            }
    }
}

// ------------------------------------------------------------------ ring-buffered bf16 GEMM
// Same contract as gemm_bf16_kernel for K % BK == 0.  The single-buffered kernel above spends a full
// memory round trip per 64-deep K tile (about 1.5 us against 0.25 us of MFMA work on the 3x3 layers);
// here the operand tiles go global -> LDS directly (global_load_lds_dwordx4, no registers) into a ring of
// NS stages, NS-1 tiles ahead of the MFMAs, with one raw barrier per K tile and counted vmcnt waits so the
// loads stay in flight across the barriers.  The LDS image of such a load is lane-linear (1 KiB per wave
// instruction), so the bank-conflict swizzle (16-byte chunk ^ row bits) is applied to the per-lane SOURCE
// address and again to the fragment reads.  Padding taps of the implicit-GEMM gather read a zero chunk.
__device__ __attribute__((aligned(16))) uint4 g_zero_chunk[1];

template <int BK>
__device__ __forceinline__ int ring_swz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NTW, int BK, int NS, bool CONV, bool RES, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void gemm_bf16_ring_kernel(BfGemm g, int nb_n, int total_tiles) {
    constexpr int BN = 64 * NTW, CH = BK / 8, ROWB = BK * 2;
    constexpr int A_BYTES = GM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
    constexpr int NLA = A_BYTES / 4096, NLW = W_BYTES / 4096, NL = NLA + NLW;   // loads per thread per stage
    constexpr int CLD = BN + 8, CHUNKS = BN / 8, NRES = GM * CHUNKS / 256;
    static_assert(NS >= 1 && NS <= 4 && (NS < 2 || NL * (NS - 2) < 64), "ring depth");   // NS == 1: a single K tile (K == BK)
    static_assert(GM * CLD * 2 <= NS * STAGE, "staged C tile must fit in the ring");
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    int t = blockIdx.x;
    if ((total_tiles & 7) == 0) t = (t & 7) * (total_tiles >> 3) + (t >> 3);   // tiles sharing A rows on one XCD
    const int mb = t / nb_n;
    const int n0 = (t - mb * nb_n) * BN, m0 = mb * GM;

    // residual tile: requested first, consumed in the epilogue
    uint4 rres[NRES];
    if constexpr (RES) {
#pragma unroll
        for (int j = 0; j < NRES; ++j) {
            const int idx = tid + j * 256;
            const int ml = idx / CHUNKS, c8 = (idx - ml * CHUNKS) * 8;
            const int m = min(m0 + ml, g.M - 1), n = min(n0 + c8, g.N - 8);
            rres[j] = *reinterpret_cast<const uint4*>(g.res + (size_t)m * g.ldr + n);
        }
    }

    // per-thread sources of the NL 16-byte pieces of a stage
    const bf16_t* srcA[NLA];
    const bf16_t* srcW[NLW];
    int py[NLA], px[NLA];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int s = i * 256 + tid, row = s / CH, c = (s % CH) ^ ring_swz<BK>(row);
        const int m = min(m0 + row, g.M - 1);
        if constexpr (CONV) {
            const int xo = m % g.Wo, tt = m / g.Wo;
            const int b = tt / g.Ho;
            py[i] = (tt - b * g.Ho) * g.stride - g.pad;
            px[i] = xo * g.stride - g.pad;
            srcA[i] = g.A + ((size_t)b * g.iH * g.iW) * g.Cin + c * 8;
        } else {
            py[i] = px[i] = 0;
            srcA[i] = g.A + (size_t)m * g.lda + c * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int s = i * 256 + tid, row = s / CH, c = (s % CH) ^ ring_swz<BK>(row);
        srcW[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
    }
    int is_k0 = 0, is_c0 = 0, is_ky = 0, is_kx = 0, is_slot = 0;      // state of the next stage to issue
    const bf16_t* const zero_chunk = reinterpret_cast<const bf16_t*>(g_zero_chunk);
    auto issue = [&]() {
        unsigned char* base = ring + is_slot * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const bf16_t* p;
            if constexpr (CONV) {
                const int yi = py[i] + is_ky, xi = px[i] + is_kx;
                const bool in = (unsigned)yi < (unsigned)g.iH && (unsigned)xi < (unsigned)g.iW;
                const int off = (yi * g.iW + xi) * g.Cin + is_c0;          // inside one image: fits an int
                p = in ? srcA[i] + off : zero_chunk;
            } else {
                p = srcA[i] + is_k0;
            }
            glds16(p, base + i * 4096);
        }
#pragma unroll
        for (int i = 0; i < NLW; ++i) glds16(srcW[i] + is_k0, base + A_BYTES + i * 4096);
        is_k0 += BK;
        is_c0 += BK;
        if (CONV && is_c0 >= g.Cin) {
            is_c0 = 0;
            if (++is_kx == g.kw) { is_kx = 0; ++is_ky; }
        }
        if (++is_slot == NS) is_slot = 0;
    };

    f32x16 acc[2][NTW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment read offsets inside a stage: row * ROWB + ((chunk ^ swz(row)) << 4), chunk = 2*ks + lh
    int offA[2], qA[2], offB[NTW], qB[NTW];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = wm * 64 + a * 32 + li;
        offA[a] = row * ROWB;
        qA[a] = lh ^ ring_swz<BK>(row);
    }
#pragma unroll
    for (int b = 0; b < NTW; ++b) {
        const int row = wn * 32 * NTW + b * 32 + li;
        offB[b] = A_BYTES + row * ROWB;
        qB[b] = lh ^ ring_swz<BK>(row);
    }

    const int KT = g.K / BK;
#pragma unroll
    for (int s = 0; s < (NS > 1 ? NS - 1 : 1); ++s)
        if (s < KT) issue();
    int slot = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const int rem = KT - 1 - kt;                       // stages issued after this one: up to NS-2 stay in flight
        if (NS >= 6 && rem >= 4) wait_vm<NL * 4>();
        else if (NS >= 5 && rem >= 3) wait_vm<NL * 3>();
        else if (NS >= 4 && rem >= 2) wait_vm<NL * 2>();
        else if (NS >= 3 && rem >= 1) wait_vm<NL>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                      // stage kt landed for every wave; slot kt-1 is free
        const unsigned char* st = ring + slot * STAGE;
        bf16x8 a[BK / 16][2], b[BK / 16][NTW];             // the whole tile's fragments, then the MFMAs
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
                a[ks][x] = *reinterpret_cast<const bf16x8*>(st + offA[x] + (((2 * ks) ^ qA[x]) << 4));
#pragma unroll
            for (int x = 0; x < NTW; ++x)
                b[ks][x] = *reinterpret_cast<const bf16x8*>(st + offB[x] + (((2 * ks) ^ qB[x]) << 4));
        }
        if (NS > 1 && kt + NS - 1 < KT) issue();
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][mt], b[ks][nt], acc[mt][nt], 0, 0, 0);
        if (++slot == NS) slot = 0;
    }
    __syncthreads();                                       // every wave is done with the ring
    // epilogue: scaled tile staged in LDS as bf16, then whole rows at 16 bytes per lane (N % 8 == 0)
    bf16_t* Cs = reinterpret_cast<bf16_t*>(ring);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int nl = wn * 32 * NTW + nt * 32 + li;
        const int n = n0 + nl;
        const float sc = (g.scale && n < g.N) ? g.scale[n] : 1.f, bi = (g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[mt][nt][r] * sc + bi;
                if (g.relu && !RES) v = fmaxf(v, 0.f);
                Cs[ml * CLD + nl] = f2bf(v);
            }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NRES; ++j) {
        const int idx = tid + j * 256;
        const int ml = idx / CHUNKS, c8 = (idx - ml * CHUNKS) * 8;
        const int m = m0 + ml, n = n0 + c8;
        if (m >= g.M || n >= g.N) continue;
        uint4 v = *reinterpret_cast<const uint4*>(&Cs[ml * CLD + c8]);
        if constexpr (RES) {
            bf16_t* hv = reinterpret_cast<bf16_t*>(&v);
            const bf16_t* hr = reinterpret_cast<const bf16_t*>(&rres[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = bf2f(hv[e]) + bf2f(hr[e]);
                if (g.relu) f = fmaxf(f, 0.f);
                hv[e] = f2bf(f);
            }
        }
        *reinterpret_cast<uint4*>(g.C + (size_t)m * g.ldc + n) = v;
    }
}

#include "resnet_patch.inc.h"
#include "resnet_join.inc.h"

// ------------------------------------------------------------------ fused stem (7x7, stride 2, pad 3, 3 -> 64)
// conv1 + bn1 + relu straight from the fp32 NCHW images to NHWC bf16: no im2col image in HBM (that image is
// 420 MB at B=256 and cost 0.6 ms to write and read back).  A workgroup walks 4x32-pixel output tiles, one
// output row per wave.  The 13x69 input window of a tile sits in LDS as bf16; the K axis is ordered
// (ci, ky, kx) with kx padded 7 -> 8, so one 8-wide MFMA operand chunk is 8 consecutive window columns
// (four aligned 4-byte LDS reads, columns 2*px .. 2*px+7; the 8th tap has zero weight).
constexpr int ST_TH = 4, ST_TW = 32;
constexpr int ST_WR = (ST_TH - 1) * 2 + 7;           // 13 window rows
constexpr int ST_WC = (ST_TW - 1) * 2 + 7;           // 69 window columns
constexpr int ST_WS = 72;                            // window row stride (column 69 = the zero tap's operand)
constexpr int ST_Q = 22;                             // chunks of 8: 21 (ci, ky) rows + 1 all-zero
constexpr int ST_K = ST_Q * 8;                       // 176 = 11 MFMA k-steps
constexpr int ST_WLD = ST_K + 8;                     // filter row stride in LDS: 368 B = odd multiple of 16
constexpr int ST_CLD = 72;

__global__ __launch_bounds__(256) void stem_conv7_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wp, int Kp,
                                                         const float* __restrict__ scale, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ y, int H, int W, int Ho, int Wo, int tiles_x,
                                                         int tiles_y, int n_tiles, int relu) {
    __shared__ __attribute__((aligned(16))) bf16_t win[3 * ST_WR * ST_WS];
    __shared__ __attribute__((aligned(16))) bf16_t wl[64 * ST_WLD];
    __shared__ __attribute__((aligned(16))) bf16_t Cs[ST_TH * ST_TW * ST_CLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // filter image: wl[n][(ci*7 + ky)*8 + kx] = wp[n][(ky*7 + kx)*3 + ci]   (packed layout of pack_conv_bf16_kernel)
    for (int i = tid; i < 64 * ST_K; i += 256) {
        const int n = i / ST_K, k = i - n * ST_K;
        const int q = k >> 3, kx = k & 7;
        bf16_t v = 0;
        if (q < 21 && kx < 7) {
            const int ci = q / 7, ky = q - ci * 7;
            v = wp[(size_t)n * Kp + (ky * 7 + kx) * 3 + ci];
        }
        wl[n * ST_WLD + k] = v;
    }
    float sc[2], bi[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        sc[nt] = scale[nt * 32 + li];
        bi[nt] = bias[nt * 32 + li];
    }
    const int abase = (2 * wave) * ST_WS + 2 * li;           // window element of this lane's pixel, tap (0, 0)
    constexpr int NWIN = (3 * ST_WR * ST_WS + 255) / 256;    // window elements per thread
    float nxt[NWIN];                                         // the next tile's window, in flight during this tile's math
    auto fetch_window = [&](int tile) {
        const int tx = tile % tiles_x, t2 = tile / tiles_x;
        const int ty = t2 % tiles_y, b = t2 / tiles_y;
        const int yi0 = ty * ST_TH * 2 - 3, xi0 = tx * ST_TW * 2 - 3;
#pragma unroll
        for (int j = 0; j < NWIN; ++j) {
            const int i = tid + j * 256;
            const int c = i % ST_WS, rr = i / ST_WS;
            const int r = rr % ST_WR, ci = rr / ST_WR;
            const int yi = yi0 + r, xi = xi0 + c;
            nxt[j] = 0.f;
            if (ci < 3 && c < ST_WC && (unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W)
                nxt[j] = x[(((size_t)b * 3 + ci) * H + yi) * W + xi];
        }
    };
    if ((int)blockIdx.x < n_tiles) fetch_window(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, t2 = tile / tiles_x;
        const int ty = t2 % tiles_y, b = t2 / tiles_y;
        const int yo0 = ty * ST_TH, xo0 = tx * ST_TW;
        __syncthreads();                                     // the previous tile is out of win / Cs (first pass: wl is complete)
#pragma unroll
        for (int j = 0; j < NWIN; ++j) {
            const int i = tid + j * 256;
            if (i < 3 * ST_WR * ST_WS) win[i] = f2bf(nxt[j]);
        }
        if (tile + (int)gridDim.x < n_tiles) fetch_window(tile + gridDim.x);
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < ST_K / 16; ++ks) {
            // chunk q = 2*ks + lh is window row (ci*13 + ky); the all-zero chunk 21 re-reads chunk 20's (finite) data
            constexpr int Q0 = 0;
            const int q0 = 2 * ks + Q0, q1 = (2 * ks + 1 > 20) ? 20 : 2 * ks + 1;
            const int rq = lh ? ((q1 / 7) * ST_WR + q1 % 7) : ((q0 / 7) * ST_WR + q0 % 7);
            const unsigned* pw = reinterpret_cast<const unsigned*>(win + abase + rq * ST_WS);
            union { unsigned u[4]; bf16x8 v; } a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a.u[e] = pw[e];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bf16x8 bb = *reinterpret_cast<const bf16x8*>(&wl[(nt * 32 + li) * ST_WLD + ks * 16 + lh * 8]);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, bb, acc[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = wave * ST_TW + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[nt][r] * sc[nt] + bi[nt];
                if (relu) v = fmaxf(v, 0.f);
                Cs[pix * ST_CLD + nt * 32 + li] = f2bf(v);
            }
        __syncthreads();
        for (int idx = tid; idx < ST_TH * ST_TW * 8; idx += 256) {
            const int pix = idx >> 3, c8 = (idx & 7) * 8;
            const int yo = yo0 + (pix >> 5), xo = xo0 + (pix & 31);
            if (yo < Ho && xo < Wo)
                *reinterpret_cast<uint4*>(y + (((size_t)b * Ho + yo) * Wo + xo) * 64 + c8) =
                    *reinterpret_cast<const uint4*>(&Cs[pix * ST_CLD + c8]);
        }
    }
}

// ------------------------------------------------------------------ layout / packing kernels
// wp[co][(tap*Cin + ci)] (K padded to Kp with zeros) = w[co][ci][tap]; scale/bias = folded BatchNorm
__global__ __launch_bounds__(256) void pack_conv_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp,
                                                             int Cout, int Cin, int taps, int Kp) {
    const size_t total = (size_t)Cout * Kp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i / Kp), k = (int)(i - (size_t)co * Kp);
        float v = 0.f;
        if (k < taps * Cin) {
            const int tap = k / Cin, ci = k - tap * Cin;
            v = w[((size_t)co * Cin + ci) * taps + tap];
        }
        wp[i] = f2bf(v);
    }
}
__global__ void fold_bn_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps,
                               float* __restrict__ scale, float* __restrict__ bias, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = s;
    bias[c] = beta[c] - mean[c] * s;
}

// im2col from NCHW fp32 (the images): col[m][(tap*Cin+ci)] bf16, K padded to Kp
__global__ __launch_bounds__(256) void im2col_nchw_f32_kernel(const float* __restrict__ x, bf16_t* __restrict__ col, int B,
                                                              int Cin, int H, int W, int Ho, int Wo, int kh, int kw,
                                                              int stride, int pad, int Kp) {
    const size_t total = (size_t)B * Ho * Wo * Kp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / Kp;
        const int k = (int)(i - m * Kp);
        float v = 0.f;
        if (k < kh * kw * Cin) {
            const int tap = k / Cin, ci = k - tap * Cin;
            const int ky = tap / kw, kx = tap - ky * kw;
            const int xo = (int)(m % Wo);
            const size_t t = m / Wo;
            const int yo = (int)(t % Ho), b = (int)(t / Ho);
            const int yi = yo * stride - pad + ky, xi = xo * stride - pad + kx;
            if (yi >= 0 && yi < H && xi >= 0 && xi < W) v = x[(((size_t)b * Cin + ci) * H + yi) * W + xi];
        }
        col[i] = f2bf(v);
    }
}

// Stem im2col (kh x kw up to 7x7, Cin <= 4, stride 2) with the input window staged in LDS: a workgroup builds
// 32 consecutive output columns of one output row; global reads run along image rows (coalesced).
constexpr int STEM_TW = 32;
__global__ __launch_bounds__(256) void im2col_stem_kernel(const float* __restrict__ x, bf16_t* __restrict__ col, int B,
                                                          int Cin, int H, int W, int Ho, int Wo, int kh, int kw,
                                                          int stride, int pad, int Kp) {
    extern __shared__ float win[];                         // [Cin][kh][wcols]
    const int wcols = (STEM_TW - 1) * stride + kw;
    const int xo0 = blockIdx.x * STEM_TW, yo = blockIdx.y, b = blockIdx.z;
    const int xi0 = xo0 * stride - pad, yi0 = yo * stride - pad;
    const int n = Cin * kh * wcols;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / (kh * wcols);
        const int rem = i - c * (kh * wcols);
        const int r = rem / wcols, cc = rem - r * wcols;
        const int yi = yi0 + r, xi = xi0 + cc;
        win[i] = (yi >= 0 && yi < H && xi >= 0 && xi < W) ? x[(((size_t)b * Cin + c) * H + yi) * W + xi] : 0.f;
    }
    __syncthreads();
    const int K = kh * kw * Cin;
    for (int i = threadIdx.x; i < STEM_TW * Kp; i += 256) {
        const int p = i / Kp, k = i - p * Kp;
        const int xo = xo0 + p;
        if (xo >= Wo) continue;
        float v = 0.f;
        if (k < K) {
            const int tap = k / Cin, ci = k - tap * Cin;
            const int ky = tap / kw, kx = tap - ky * kw;
            v = win[(ci * kh + ky) * wcols + p * stride + kx];
        }
        col[(((size_t)b * Ho + yo) * Wo + xo) * Kp + k] = f2bf(v);
    }
}

// im2col from NHWC bf16, 8 channels (16 bytes) per thread; Cin % 8 == 0
__global__ __launch_bounds__(256) void im2col_nhwc_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col,
                                                               int B, int Cin, int H, int W, int Ho, int Wo, int kh,
                                                               int kw, int stride, int pad) {
    const int c8n = Cin / 8;
    const size_t total = (size_t)B * Ho * Wo * kh * kw * c8n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        size_t t = i / c8n;
        const int tap = (int)(t % (kh * kw));
        const size_t m = t / (kh * kw);
        const int ky = tap / kw, kx = tap - ky * kw;
        const int xo = (int)(m % Wo);
        const size_t t2 = m / Wo;
        const int yo = (int)(t2 % Ho), b = (int)(t2 / Ho);
        const int yi = yo * stride - pad + ky, xi = xo * stride - pad + kx;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (yi >= 0 && yi < H && xi >= 0 && xi < W)
            v = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + yi) * W + xi) * Cin + c8 * 8);
        *reinterpret_cast<uint4*>(col + (m * kh * kw + tap) * Cin + c8 * 8) = v;
    }
}

// MaxPool2d(3, stride 2, padding 1) on NHWC bf16, 8 channels per thread
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                                int B, int C, int H, int W, int Ho, int Wo) {
    const int c8n = C / 8;
    const size_t total = (size_t)B * Ho * Wo * c8n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        size_t t = i / c8n;
        const int xo = (int)(t % Wo); t /= Wo;
        const int yo = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float best[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int yi = 2 * yo - 1 + ky, xi = 2 * xo - 1 + kx;
                if (yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + yi) * W + xi) * C + c8 * 8);
                const bf16_t* h = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
                for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], bf2f(h[e]));
            }
        bf16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(best[e]);
        *reinterpret_cast<uint4*>(y + (((size_t)b * Ho + yo) * Wo + xo) * C + c8 * 8) = *reinterpret_cast<const uint4*>(o);
    }
}

// AdaptiveAvgPool2d(1) + Flatten: (B,H,W,C) bf16 -> (B,C) fp32
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int B,
                                                           int C, int HW) {
    const int c8n = C / 8;                                   // 8 channels (16 bytes) per thread when C % 8 == 0
    if (C % 8 == 0) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= B * c8n) return;
        const int b = i / c8n, c8 = (i - b * c8n) * 8;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < HW; ++p) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + ((size_t)b * HW + p) * C + c8);
            const bf16_t* h = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += bf2f(h[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) y[(size_t)b * C + c8 + e] = s[e] / (float)HW;
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * C; i += gridDim.x * 256) {
        const int b = i / C, c = i - b * C;
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += bf2f(x[((size_t)b * HW + p) * C + c]);
        y[i] = s / (float)HW;
    }
}

int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

namespace {
struct PackLayout { size_t wp, scale, bias, total; int Kp; };
PackLayout pack_layout(int Cout, int Cin, int kh, int kw) {
    PackLayout o{};
    o.Kp = i2l_cdiv(kh * kw * Cin, 8) * 8;
    o.wp = 0;
    o.scale = i2l_align((size_t)Cout * o.Kp * sizeof(bf16_t));
    o.bias = o.scale + i2l_align((size_t)Cout * sizeof(float));
    o.total = o.bias + i2l_align((size_t)Cout * sizeof(float));
    return o;
}
}  // namespace

namespace {
bool stem_shape(int Cin, int Cout, int kh, int kw, int stride, int pad) {
    return Cin == 3 && Cout == 64 && kh == 7 && kw == 7 && stride == 2 && pad == 3;
}
// Stages of the operand ring (I2L_FLAG_RESNET_RING_DEPTH(n) overrides, 2..4).  Two stages are 64 KB of LDS, so two
// workgroups share a CU and cover each other's waits: measured faster than a deeper ring on every layer
// that has more tiles than CUs.  With at most one tile per CU only the ring hides latency: 4 stages.
int ring_depth(int kt, long tiles, int flags) {
    const int forced = (flags >> 8) & 0xF;
    if (forced >= 2 && forced <= 5) return forced;
    return (tiles <= 256 && kt >= 4) ? 4 : 2;
}
template <int NTW, int BK, int NS, bool CONV, bool RES, int OCC = 1>
int launch_ring4(const BfGemm& g, int nb_n, int total, hipStream_t s) {
    constexpr int lds = NS * (GM + 64 * NTW) * BK * 2;
    // once per template instantiation (a function-local static is initialised exactly once, thread-safe): the attribute
    // is a property of the kernel, not of the launch -- r02 asked the runtime again on each of the 53 launches per image batch
    static std::atomic<unsigned> attr{0};                       // ... and of the DEVICE: one bit per device
    if (!i2l_lds_attr(reinterpret_cast<const void*>(gemm_bf16_ring_kernel<NTW, BK, NS, CONV, RES, OCC>), lds, attr))
        return I2L_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_bf16_ring_kernel<NTW, BK, NS, CONV, RES, OCC>), dim3(total), dim3(256), lds, s, g, nb_n, total);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
template <int NTW, int BK, int NS>
int launch_ring2(const BfGemm& g, int nb_n, int total, hipStream_t s) {
    if (g.conv)
        return g.res ? launch_ring4<NTW, BK, NS, true, true>(g, nb_n, total, s)
                     : launch_ring4<NTW, BK, NS, true, false>(g, nb_n, total, s);
    return g.res ? launch_ring4<NTW, BK, NS, false, true>(g, nb_n, total, s)
                 : launch_ring4<NTW, BK, NS, false, false>(g, nb_n, total, s);
}
template <int NTW>
int launch_ring(const BfGemm& g, int depth, int nb_n, int total, hipStream_t s) {
    if (depth == 5) return launch_ring2<NTW, 32, 4>(g, nb_n, total, s);      // 32-deep K tiles, 4 stages: the LDS of <64, 2>, +-3 % per layer (r04)
    return depth == 4 ? launch_ring2<NTW, 64, 4>(g, nb_n, total, s)
         : depth == 3 ? launch_ring2<NTW, 64, 3>(g, nb_n, total, s) : launch_ring2<NTW, 64, 2>(g, nb_n, total, s);
}
}  // namespace

extern "C" size_t i2l_conv_bf16_packed_bytes(int Cout, int Cin, int kh, int kw) {
    if (Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0) return 0;
    return pack_layout(Cout, Cin, kh, kw).total;
}

// Weight-only preparation of one conv + BatchNorm pair: bf16 [Cout][taps*Cin] image and the folded
// scale/bias.  Valid until a weight or a BatchNorm statistic changes.
extern "C" int i2l_conv_bn_bf16_pack(const float* w, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                                      const float* bn_var, float bn_eps, void* packed, size_t packed_bytes, int Cout,
                                      int Cin, int kh, int kw, i2l_stream_t stream) {
    if (!w || !bn_weight || !bn_bias || !bn_mean || !bn_var || !packed || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0)
        return I2L_ERR_ARG;
    const PackLayout lo = pack_layout(Cout, Cin, kh, kw);
    if (packed_bytes < lo.total) return I2L_ERR_WORKSPACE;
    hipStream_t s = i2l_s(stream);
    char* base = static_cast<char*>(packed);
    hipLaunchKernelGGL(pack_conv_bf16_kernel, dim3(grid_for((size_t)Cout * lo.Kp)), dim3(256), 0, s, w,
                       reinterpret_cast<bf16_t*>(base + lo.wp), Cout, Cin, kh * kw, lo.Kp);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_bn_kernel, dim3(i2l_cdiv(Cout, 256)), dim3(256), 0, s, bn_weight, bn_bias, bn_mean, bn_var,
                       bn_eps, reinterpret_cast<float*>(base + lo.scale), reinterpret_cast<float*>(base + lo.bias), Cout);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" size_t i2l_conv_bf16_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                                int pad, int flags) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return 0;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const int Kp = i2l_cdiv(kh * kw * Cin, 8) * 8;
    const bool direct = kh == 1 && kw == 1 && stride == 1 && pad == 0 && Cin % 8 == 0;
    const bool implicit = Cin % 64 == 0;                   // NHWC input gathered inside the GEMM (no im2col image)
    const bool stem = !(flags & I2L_FLAG_RESNET_IM2COL_STEM) && stem_shape(Cin, Cout, kh, kw, stride, pad);
    return (direct || implicit || stem) ? 256 : i2l_align((size_t)B * Ho * Wo * Kp * sizeof(bf16_t));
}

// y = act( BN(conv(x, w)) + residual ), NHWC bf16 in/out (x may instead be the NCHW fp32 image batch);
// `packed` comes from i2l_conv_bn_bf16_pack for the same conv.
extern "C" int i2l_conv_bn_act_bf16_fwd(const void* x, int x_is_nchw_f32, const void* packed, const void* residual,
                                         void* y, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                         int pad, int relu, void* workspace, size_t workspace_bytes, int flags,
                                         i2l_stream_t stream) {
    if (!x || !packed || !y) return I2L_ERR_ARG;
    const size_t need = i2l_conv_bf16_workspace_bytes(B, H, W, Cin, Cout, kh, kw, stride, pad, flags);
    if (need == 0) return I2L_ERR_ARG;
    if (!workspace || workspace_bytes < need) return I2L_ERR_WORKSPACE;
    if (!x_is_nchw_f32 && Cin % 8 != 0) return I2L_ERR_UNSUPPORTED;
    if (Cout % 8 != 0) return I2L_ERR_UNSUPPORTED;
    hipStream_t s = i2l_s(stream);
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    const int taps = kh * kw;
    const PackLayout lo = pack_layout(Cout, Cin, kh, kw);
    const int Kp = lo.Kp;
    const char* pbase = static_cast<const char*>(packed);
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(pbase + lo.wp);
    const float* scale = reinterpret_cast<const float*>(pbase + lo.scale);
    const float* bias = reinterpret_cast<const float*>(pbase + lo.bias);
    bf16_t* col = static_cast<bf16_t*>(workspace);
    const size_t M = (size_t)B * Ho * Wo;
    if (M > 0x7fffffff) return I2L_ERR_UNSUPPORTED;
    if (x_is_nchw_f32 && !residual && !(flags & I2L_FLAG_RESNET_IM2COL_STEM) && stem_shape(Cin, Cout, kh, kw, stride, pad)) {
        const int tiles_x = i2l_cdiv(Wo, ST_TW), tiles_y = i2l_cdiv(Ho, ST_TH);
        const long n_tiles = (long)B * tiles_x * tiles_y;
        if (n_tiles > 0x7fffffff) return I2L_ERR_UNSUPPORTED;
        const int grid = (int)(n_tiles < 768 ? n_tiles : 768);          // 3 workgroups per CU walk the tiles
        hipLaunchKernelGGL(stem_conv7_kernel, dim3(grid), dim3(256), 0, s, static_cast<const float*>(x), wp, Kp, scale,
                           bias, static_cast<bf16_t*>(y), H, W, Ho, Wo, tiles_x, tiles_y, (int)n_tiles, relu);
        I2L_CHECK_LAUNCH();
        return I2L_OK;
    }
    BfGemm g{};
    const bool direct = !x_is_nchw_f32 && kh == 1 && kw == 1 && stride == 1 && pad == 0;
    const bool implicit = !x_is_nchw_f32 && !direct && Cin % 64 == 0;
    if (direct) {
        g.A = static_cast<const bf16_t*>(x); g.lda = Cin;
    } else if (implicit) {
        g.conv = 1; g.iH = H; g.iW = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.kw = kw; g.stride = stride; g.pad = pad;
        g.A = static_cast<const bf16_t*>(x); g.lda = 0;
    } else {
        const size_t stem_lds = (size_t)Cin * kh * ((STEM_TW - 1) * stride + kw) * sizeof(float);
        if (x_is_nchw_f32 && stem_lds <= 48 * 1024 && Ho <= 65535 && B <= 65535)
            hipLaunchKernelGGL(im2col_stem_kernel, dim3(i2l_cdiv(Wo, STEM_TW), Ho, B), dim3(256), stem_lds, s,
                               static_cast<const float*>(x), col, B, Cin, H, W, Ho, Wo, kh, kw, stride, pad, Kp);
        else if (x_is_nchw_f32)
            hipLaunchKernelGGL(im2col_nchw_f32_kernel, dim3(grid_for(M * Kp)), dim3(256), 0, s,
                               static_cast<const float*>(x), col, B, Cin, H, W, Ho, Wo, kh, kw, stride, pad, Kp);
        else
            hipLaunchKernelGGL(im2col_nhwc_bf16_kernel, dim3(grid_for(M * taps * (Cin / 8))), dim3(256), 0, s,
                               static_cast<const bf16_t*>(x), col, B, Cin, H, W, Ho, Wo, kh, kw, stride, pad);
        I2L_CHECK_LAUNCH();
        g.A = col; g.lda = Kp;
    }
    g.W = wp; g.ldw = Kp;
    g.scale = scale; g.bias = bias;
    g.res = static_cast<const bf16_t*>(residual); g.ldr = Cout;
    g.C = static_cast<bf16_t*>(y); g.ldc = Cout;
    g.M = (int)M; g.N = Cout; g.K = Kp; g.relu = relu;
    // short reductions (K <= 128: the 1x1 convs of layer1 / layer2 that expand 64 / 128 channels) are bound by HBM, not by
    // the matrix cores: 64-column tiles at THREE workgroups per CU (48 KB of LDS, <= 168 registers) keep more bytes in flight
    const bool short_k = direct && Kp <= 256 && !((flags >> 8) & 0xF) && (long)i2l_cdiv((int)M, GM) * i2l_cdiv(Cout, 64) >= 1024;
    int gn = (Cout <= 64 || short_k) ? 64 : 128;           // narrow tile for the 64-channel layers (no wasted MFMAs)
    if (gn == 128 && Cout % 128 == 0 && !(flags & I2L_FLAG_RESNET_WIDE_TILES)) {
        // balance: the layers of the trunk are one to five 128 x 128 tiles per CU, so the busiest CU decides (MFMAs alone take
        // 26 us where the mean load would take 17, profiles/r04/resnet_patch.txt): 64-column tiles when they lower its share
        const long t128 = (long)i2l_cdiv((int)M, GM) * (Cout / 128);
        const long c128 = ((t128 + 255) / 256) * 2, c64 = (2 * t128 + 255) / 256;
        if (c64 < c128) gn = 64;
    }
    if (implicit && kh == 3 && kw == 3 && stride == 1 && pad == 1 && !(flags & (I2L_FLAG_RESNET_NO_RING | I2L_FLAG_RESNET_NO_PATCH))) {
        // input patch staged in LDS, filters through the ring; the tile shape is picked for balance over the CUs
        const PatchPlan pp = patch_plan((long)M, H, W, Cin, Cout, 256, (flags >> 20) & 0xF);
        if (pp.shape >= 0) return launch_patch(g, pp, s);
    }
    if (!(flags & I2L_FLAG_RESNET_NO_RING) && (direct || implicit) && Kp % 64 == 0) {
        const int nb_n = i2l_cdiv(Cout, gn), nb_m = i2l_cdiv((int)M, GM);
        const long total = (long)nb_n * nb_m;
        const int depth = ring_depth(Kp / 64, total, flags);
        if (total > 0x7fffffff) return I2L_ERR_UNSUPPORTED;
        if (short_k && Kp == 64)          // one K tile: no ring at all, 24.5 KB of LDS, four workgroups per CU
            return g.res ? launch_ring4<1, 64, 1, false, true, 4>(g, nb_n, (int)total, s)
                         : launch_ring4<1, 64, 1, false, false, 4>(g, nb_n, (int)total, s);
        if (short_k)
            return g.res ? launch_ring4<1, 64, 2, false, true, 3>(g, nb_n, (int)total, s)
                         : launch_ring4<1, 64, 2, false, false, 3>(g, nb_n, (int)total, s);
        return gn == 64 ? launch_ring<1>(g, depth, nb_n, (int)total, s) : launch_ring<2>(g, depth, nb_n, (int)total, s);
    }
    dim3 grid(i2l_cdiv(Cout, gn), i2l_cdiv((int)M, GM));
    if (grid.y > 65535) return I2L_ERR_UNSUPPORTED;
    if (gn == 64) hipLaunchKernelGGL(gemm_bf16_kernel<1>, grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_bf16_kernel<2>, grid, dim3(256), 0, s, g);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_maxpool3x3s2_bf16_fwd(const void* x, void* y, int B, int H, int W, int C, i2l_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return I2L_ERR_ARG;
    if (C % 8 != 0) return I2L_ERR_UNSUPPORTED;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 8))), dim3(256), 0, i2l_s(stream),
                       static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y), B, C, H, W, Ho, Wo);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_global_avgpool_bf16_fwd(const void* x, float* y, int B, int H, int W, int C, i2l_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return I2L_ERR_ARG;
    hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3(i2l_cdiv(C % 8 == 0 ? B * (C / 8) : B * C, 256)), dim3(256), 0, i2l_s(stream),
                       static_cast<const bf16_t*>(x), y, B, C, H * W);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

// Bottleneck tail + the next block's head in one launch (resnet_join.inc.h): y = relu(bn3(conv3(o2)) + identity), NHWC bf16,
// and z = relu(bn1'(conv1'(y))).  `packed3` / `packed1_next` come from i2l_conv_bn_bf16_pack for the two 1x1 convs.  Built for
// the shapes whose y tile fits in LDS beside its filters: c_mid = 64 -> c_out = 256 -> n2 = 64 or 128 (layer1 of the bottleneck
// ResNets, and its hand-over to layer2); anything else returns I2L_ERR_UNSUPPORTED and the caller makes two launches.
extern "C" int i2l_bottleneck_join_bf16_fwd(const void* o2, const void* packed3, const void* identity, void* y,
                                             const void* packed1_next, void* z, long positions, int c_mid, int c_out, int n2,
                                             i2l_stream_t stream) {
    if (!o2 || !packed3 || !identity || !y || !packed1_next || !z || positions <= 0) return I2L_ERR_ARG;
    if (c_mid != 64 || c_out != 256 || (n2 != 64 && n2 != 128) || positions > 0x7fffffffL) return I2L_ERR_UNSUPPORTED;
    const PackLayout l3 = pack_layout(c_out, c_mid, 1, 1), l1 = pack_layout(n2, c_out, 1, 1);
    const char* p3 = static_cast<const char*>(packed3);
    const char* p1 = static_cast<const char*>(packed1_next);
    JoinArgs g{};
    g.A1 = static_cast<const bf16_t*>(o2);
    g.W1 = reinterpret_cast<const bf16_t*>(p3 + l3.wp);
    g.scale1 = reinterpret_cast<const float*>(p3 + l3.scale); g.bias1 = reinterpret_cast<const float*>(p3 + l3.bias);
    g.res = static_cast<const bf16_t*>(identity); g.Y = static_cast<bf16_t*>(y);
    g.W2 = reinterpret_cast<const bf16_t*>(p1 + l1.wp);
    g.scale2 = reinterpret_cast<const float*>(p1 + l1.scale); g.bias2 = reinterpret_cast<const float*>(p1 + l1.bias);
    g.Z = static_cast<bf16_t*>(z);
    g.M = (int)positions;
    g.n_tiles = i2l_cdiv((int)positions, JN_TM);
    const dim3 grid((unsigned)(g.n_tiles < 256 ? g.n_tiles : 256));      // one persistent workgroup per CU walks the tiles
    hipStream_t s = i2l_s(stream);
    if (n2 == 64) {
        static std::atomic<unsigned> attr{0};
        if (!i2l_lds_attr(reinterpret_cast<const void*>(bottleneck_join_kernel<1>), join_lds_bytes(64), attr)) return I2L_ERR_LAUNCH;
        hipLaunchKernelGGL(bottleneck_join_kernel<1>, grid, dim3(256), join_lds_bytes(64), s, g);
    } else {
        static std::atomic<unsigned> attr{0};
        if (!i2l_lds_attr(reinterpret_cast<const void*>(bottleneck_join_kernel<2>), join_lds_bytes(128), attr)) return I2L_ERR_LAUNCH;
        hipLaunchKernelGGL(bottleneck_join_kernel<2>, grid, dim3(256), join_lds_bytes(128), s, g);
    }
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
