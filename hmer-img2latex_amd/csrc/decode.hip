// LSTM decoder step loop as ONE persistent launch (reference decoder.py:197-284 driven
// by seq2seq.py:210-221 / predictor.py:283-347).
//
// Design (DESIGN.md "decode"): the recurrence of one batch row never touches another
// row, so rows are partitioned over workgroups and a workgroup runs ALL steps of its
// rows with no inter-workgroup communication: no grid barrier, no host sync, nothing
// that can deadlock.  Per step a workgroup streams the recurrent and projection
// weights (L2-resident: 1.5 MB at E=H=256,V=512) as coalesced 16-byte loads; the state
// (h, c) lives in LDS for the whole loop.
//
// Weight images built once per call by i2l_decoder_prepare():
//   WhhT[l][k][4j+g] = W_hh_l[g*H+j][k]      gate-interleaved transpose: thread j reads one
//   WihT[l][k][4j+g] = W_ih_l[g*H+j][k] (l>0) float4 per k = (i,f,g,o) of hidden unit j
//   P[v][4j+g]       = sum_e Emb[v][e] * W_ih_0[g*H+j][e]          token half of layer-0 gates
//   Genc[b][4j+g]    = sum_e enc[b][e] * W_ih_0[g*H+j][E+e] + b_ih_0 + b_hh_0   image half
//   WoutT[k][v]      = W_out[v][k], columns padded to a multiple of 512 (bias -inf)
// so a step is: gates = P[tok] + Genc + h @ W_hh^T (a k-ordered fmaf chain per gate),
// LSTM pointwise, logits = h' @ W_out^T + b, first-index argmax by wave shuffles.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int MAXL = I2L_MAX_LSTM_LAYERS;
constexpr int NT = 256;        // threads per workgroup
constexpr int VCHUNK = 2 * NT;  // vocab columns per projection pass

struct Layout {
    size_t P, Genc, WhhT[MAXL], WihT[MAXL], biasP[MAXL], WoutT, boutP, wpack16, gemm_ws, xchg, xchg_bytes, total;
    int Vp, n_groups;
};

// grouped decode (decode_group.inc.h): exchange granules [groups padded to 8][2][4][288] x 8 B + a status block
constexpr size_t GROUP_XCHG_PER_GROUP = (size_t)2 * 4 * 288 * 8;
constexpr size_t GROUP16_XCHG_BYTES = (size_t)2 * 16 * 288 * 8;   // decode_group16.inc.h: [2 parities][16 members][288 granules]
constexpr size_t GROUP_STATUS_BYTES = 2048;
inline bool group_shape_ok(int V, int H, int L) { return L == 1 && H == 256 && V <= 512; }

Layout make_layout(int rows, int V, int E, int H, int L) {
    Layout o{};
    size_t off = 0;
    auto take = [&](size_t floats) { size_t r = off; off += i2l_align(floats * sizeof(float)); return r; };
    const size_t G = 4 * (size_t)H;
    o.Vp = i2l_cdiv(V, VCHUNK) * VCHUNK;
    o.P = take((size_t)V * G);
    o.Genc = take((size_t)rows * G);
    for (int l = 0; l < L; ++l) o.WhhT[l] = take((size_t)H * G);
    for (int l = 1; l < L; ++l) o.WihT[l] = take((size_t)H * G);
    for (int l = 1; l < L; ++l) o.biasP[l] = take(G);
    o.WoutT = take((size_t)H * o.Vp);
    o.boutP = take(o.Vp);
    // decode_group16_kernel: the weights pre-split into bf16 pieces in the matrix cores' operand layout, [member 16][wave 4]
    // [fragment 36][lane 64] x 16 bytes (weight-only: built with the other weight images under I2L_PREP_WEIGHTS)
    if (group_shape_ok(V, H, L)) o.wpack16 = take((size_t)16 * 4 * 36 * 64 * 4);
    size_t g1 = i2l_gemm_workspace_bytes(V, 4 * H, E), g2 = i2l_gemm_workspace_bytes(rows, 4 * H, E);
    o.gemm_ws = off;
    off += i2l_align(g1 > g2 ? g1 : g2);
    if (group_shape_ok(V, H, L)) {
        o.n_groups = i2l_cdiv(rows, 4);
        o.xchg = off;
        const size_t x4 = (size_t)i2l_cdiv(o.n_groups, 8) * 8 * GROUP_XCHG_PER_GROUP;
        const size_t x8 = (size_t)i2l_cdiv(i2l_cdiv(rows, 8), 8) * 8 * (2 * GROUP_XCHG_PER_GROUP);   // 8 members x 8 rows (decode_group8)
        const size_t x16 = (size_t)i2l_cdiv(i2l_cdiv(rows, 16), 8) * 8 * GROUP16_XCHG_BYTES;         // 16 x 16 (decode_group16)
        const size_t x48 = x4 > x8 ? x4 : x8;
        o.xchg_bytes = GROUP_STATUS_BYTES + (x48 > x16 ? x48 : x16);
        off += i2l_align(o.xchg_bytes);
    }
    o.total = off;
    return o;
}

// out[k][c] = W[n(c)][k] for c < Npad; n(c) = (c&3)*perm_h + (c>>2) when perm_h > 0 else c; rows n >= N read as 0.
__global__ __launch_bounds__(256) void transpose_perm_kernel(const float* __restrict__ W, int N, int K,
                                                             float* __restrict__ out, int ld_out, int Npad,
                                                             int perm_h) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, k = k0 + tx;
        const int n = perm_h > 0 ? (c & 3) * perm_h + (c >> 2) : c;
        tile[i][tx] = (c < Npad && n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int k = k0 + i, c = c0 + tx;
        if (k < K && c < Npad) out[(size_t)k * ld_out + c] = tile[tx][i];
    }
}

__global__ void bias_perm_kernel(const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                 float* __restrict__ out, int H) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < 4 * H) {
        const int n = (c & 3) * H + (c >> 2);
        out[c] = b_ih[n] + b_hh[n];
    }
}

__global__ void bout_pad_kernel(const float* __restrict__ b_out, float* __restrict__ out, int V, int Vp) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < Vp) out[v] = v < V ? b_out[v] : -INFINITY;
}

struct StepWeights {
    int H, L, V, Vp;
    const float* P;
    const float* Genc;
    const float* WhhT[MAXL];
    const float* WihT[MAXL];
    const float* biasP[MAXL];
    const float* WoutT;
    const float* boutP;
    const void* wpack16;      // decode_group16_kernel's operand image (null when the dimensions have no grouped path)
};

struct DecodeParams {
    StepWeights w;
    int B, T;
    const int32_t* tok0;
    const int32_t* forced;
    const float* h0;
    const float* c0;
    float* h_out;
    float* c_out;
    int32_t* ids;
    float* logits;
    float temperature;
    int use_temp, select, stop, end_id;
    int top_k; float top_p; unsigned long long seed; float* probs;   // I2L_SELECT_SAMPLE only
};

// Gate functions of the search kernels (greedy, beam, row-per-workgroup): hardware exp2 / reciprocal (v_exp_f32,
// v_rcp_f32: ~1 ulp each) instead of libm's expf / tanhf and an IEEE division -- the LSTM cell is a chain of five of
// them on the critical path of every step (~150 dependent instructions -> ~30).  Absolute error <= ~2e-7 per call, the
// class of the vectorised expf / tanhf ATen itself uses; saturation: exp -> inf gives 1/inf = 0, exp -> 0 gives -1 / 0+.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

// acc[r] += sum_k W[k][0..3] * x[r][k], k ascending (one fmaf chain per output).
// The weight stream is software-pipelined: while the FMAs of one batch of PF rows run, the
// loads of the next batch are already in flight (two register sets, counted vmcnt by the
// compiler), so one wave per SIMD keeps ~PF 16-byte loads per lane outstanding.
constexpr int PF = 8;

template <int R, int NV>   // NV = 4 (float4: LSTM gates) or 2 (float2: vocabulary projection)
struct WVec;
template <int R>
struct WVec<R, 4> { typedef float4 type; };
template <int R>
struct WVec<R, 2> { typedef float2 type; };

template <int R>
__device__ __forceinline__ void fma_rows(float4 (&acc)[R], const float4 (&w)[PF], const float* xs, int H, int k) {   // H = row stride of xs
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int i4 = 0; i4 < PF; i4 += 4) {
            const float4 xa = *reinterpret_cast<const float4*>(xs + r * H + k + i4);
            const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[r].x = fmaf(w[i4 + i].x, xv[i], acc[r].x);
                acc[r].y = fmaf(w[i4 + i].y, xv[i], acc[r].y);
                acc[r].z = fmaf(w[i4 + i].z, xv[i], acc[r].z);
                acc[r].w = fmaf(w[i4 + i].w, xv[i], acc[r].w);
            }
        }
    }
}

template <typename VT>
__device__ __forceinline__ void load_batch(VT (&w)[PF], const float* __restrict__ Wcol, size_t ldw) {
#pragma unroll
    for (int i = 0; i < PF; ++i) w[i] = *reinterpret_cast<const VT*>(Wcol + (size_t)i * ldw);
}

// matvec whose first batch (rows 0..PF-1) was loaded earlier by load_batch (prefetch across a barrier).
template <int R, typename VT>
__device__ __forceinline__ void matvec_pre(VT (&acc)[R], VT (&wa)[PF], const float* __restrict__ Wcol, size_t ldw,
                                           const float* xs, int xstride, int H) {
    VT wb[PF];
    int k = 0;
    for (; k + 2 * PF < H; k += 2 * PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) wb[i] = *reinterpret_cast<const VT*>(Wcol + (size_t)(k + PF + i) * ldw);
        __builtin_amdgcn_sched_barrier(0);
        fma_rows<R>(acc, wa, xs, xstride, k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PF; ++i) wa[i] = *reinterpret_cast<const VT*>(Wcol + (size_t)(k + 2 * PF + i) * ldw);
        __builtin_amdgcn_sched_barrier(0);
        fma_rows<R>(acc, wb, xs, xstride, k + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) wb[i] = *reinterpret_cast<const VT*>(Wcol + (size_t)(k + PF + i) * ldw);
    __builtin_amdgcn_sched_barrier(0);
    fma_rows<R>(acc, wa, xs, xstride, k);
    fma_rows<R>(acc, wb, xs, xstride, k + PF);
}

template <int R, typename VT>
__device__ __forceinline__ void matvec(VT (&acc)[R], const float* __restrict__ Wcol, size_t ldw, const float* xs,
                                       int xstride, int H) {   // H = number of k (multiple of 32), xstride = row stride of xs
    VT wa[PF], wb[PF];
    // No load sits under a condition: the compiler can then count outstanding loads exactly
    // (s_waitcnt vmcnt(PF) before a batch is used) instead of draining to vmcnt(0) at a join.
#pragma unroll
    for (int i = 0; i < PF; ++i) wa[i] = *reinterpret_cast<const VT*>(Wcol + (size_t)i * ldw);
    int k = 0;
    for (; k + 2 * PF < H; k += 2 * PF) {                 // H % (2*PF) == 0
#pragma unroll
        for (int i = 0; i < PF; ++i) wb[i] = *reinterpret_cast<const VT*>(Wcol + (size_t)(k + PF + i) * ldw);
        __builtin_amdgcn_sched_barrier(0);                // keep the batch of loads ahead of the FMAs
        fma_rows<R>(acc, wa, xs, xstride, k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PF; ++i) wa[i] = *reinterpret_cast<const VT*>(Wcol + (size_t)(k + 2 * PF + i) * ldw);
        __builtin_amdgcn_sched_barrier(0);
        fma_rows<R>(acc, wb, xs, xstride, k + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) wb[i] = *reinterpret_cast<const VT*>(Wcol + (size_t)(k + PF + i) * ldw);
    __builtin_amdgcn_sched_barrier(0);
    fma_rows<R>(acc, wa, xs, xstride, k);
    fma_rows<R>(acc, wb, xs, xstride, k + PF);
}

template <int R>
__device__ __forceinline__ void matvec4(float4 (&acc)[R], const float* __restrict__ Wcol, size_t ldw,
                                        const float* xs, int H) {
    matvec<R, float4>(acc, Wcol, ldw, xs, H, H);
}

// (value, index) max with "first index wins" over the 64 lanes of a wave.
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// All L layers of one LSTM time step for R rows of one workgroup (decoder.py:247/277 at seq len 1).
// h is double-buffered in LDS: reads hs[par], writes hs[par^1]; c_in -> c_out (may alias: each
// element is read and written by the same thread).  tok[r] selects the row of P, grow[r] the row of Genc.
// Ends with a __syncthreads(): hs[par^1] is complete on return.
template <int R>
__device__ __forceinline__ void lstm_layers(const StepWeights& w, const int (&tok)[R], const int (&grow)[R],
                                            float* hs, const float* c_in, float* c_out, int par, int tid) {
    const int H = w.H, L = w.L;
    const size_t G = 4 * (size_t)H;
    for (int l = 0; l < L; ++l) {
        const float* h_old = hs + ((size_t)par * L + l) * R * H;
        float* h_new = hs + ((size_t)(par ^ 1) * L + l) * R * H;
        for (int j = tid; j < H; j += NT) {
            float4 acc[R];
            if (l == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float4 a = *reinterpret_cast<const float4*>(w.P + (size_t)tok[r] * G + 4 * j);
                    const float4 e = *reinterpret_cast<const float4*>(w.Genc + (size_t)grow[r] * G + 4 * j);
                    acc[r] = make_float4(a.x + e.x, a.y + e.y, a.z + e.z, a.w + e.w);
                }
            } else {
                const float4 bb = *reinterpret_cast<const float4*>(w.biasP[l] + 4 * j);
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = bb;
                matvec4<R>(acc, w.WihT[l] + 4 * j, G, hs + ((size_t)(par ^ 1) * L + (l - 1)) * R * H, H);
            }
            matvec4<R>(acc, w.WhhT[l] + 4 * j, G, h_old, H);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const size_t ci = ((size_t)l * R + r) * H + j;
                const float ig = sigmoidf_(acc[r].x), fg = sigmoidf_(acc[r].y);
                const float gg = tanhf_(acc[r].z), og = sigmoidf_(acc[r].w);
                const float cn = fg * c_in[ci] + ig * gg;
                c_out[ci] = cn;
                h_new[r * H + j] = og * tanhf_(cn);
            }
        }
        __syncthreads();
    }
}

// logits = h_top @ W_out^T + b for R rows, 512 vocabulary columns per pass.  16-byte loads only
// (8-byte streams run at ~0.55x the 16-byte rate on gfx950): thread t owns the 4 columns
// 4*(t&127).. and HALF of k -- threads 0-127 sum k < H/2 (starting from the bias), threads 128-255
// sum k >= H/2; the two partial chains meet through LDS (psum) in a fixed order, so results are
// deterministic.  Optionally writes raw logits to global (lp[r] != null), keeps logit/temperature
// in LDS (lg != null), and returns each thread's running (max, first index); upper-half threads
// return -inf.  Contains __syncthreads(); every thread of the workgroup must call it.
template <int R, bool PRE = false>
__device__ __forceinline__ void project(const StepWeights& w, const float* h_top, float* const (&lp)[R],
                                        float* lg, bool use_temp, float temperature, float (&best)[R],
                                        int (&besti)[R], float* psum, int tid, float4 (*pre)[PF] = nullptr) {
    const int half = tid >> 7, c4 = (tid & 127) * 4;
    const int Hh = w.H >> 1;
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = -INFINITY; besti[r] = 0x7fffffff; }
    for (int v0 = c4; v0 < w.Vp; v0 += VCHUNK) {
        float4 acc[R];
        const float4 bb = half == 0 ? *reinterpret_cast<const float4*>(w.boutP + v0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = bb;
        if (PRE && v0 == c4)      // first pass: rows 0..PF-1 were prefetched before the LSTM step
            matvec_pre<R, float4>(acc, *pre, w.WoutT + (size_t)half * Hh * w.Vp + v0, (size_t)w.Vp, h_top + half * Hh, w.H, Hh);
        else
            matvec<R, float4>(acc, w.WoutT + (size_t)half * Hh * w.Vp + v0, (size_t)w.Vp, h_top + half * Hh, w.H, Hh);
        if (half == 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(psum + ((size_t)r * 128 + (tid & 127)) * 4) = acc[r];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 u = *reinterpret_cast<const float4*>(psum + ((size_t)r * 128 + tid) * 4);
                float a[4] = {acc[r].x + u.x, acc[r].y + u.y, acc[r].z + u.z, acc[r].w + u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (lp[r] && v0 + e < w.V) lp[r][v0 + e] = a[e];
                    if (use_temp) a[e] = a[e] / temperature;
                    if (lg) lg[r * w.Vp + v0 + e] = a[e];
                    if (a[e] > best[r]) { best[r] = a[e]; besti[r] = v0 + e; }
                }
            }
        }
        if (v0 + VCHUNK < w.Vp) __syncthreads();          // psum is reused by the next pass
    }
}

// Fold per-thread (max, index) pairs over the workgroup; on return every thread holds the result.
// redv/redi: LDS [NT/64][R].  Contains two __syncthreads().
template <int R>
__device__ __forceinline__ void block_argmax(float (&best)[R], int (&besti)[R], float* redv, int* redi, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        wave_argmax(best[r], besti[r]);
        if (lane == 0) { redv[wave * R + r] = best[r]; redi[wave * R + r] = besti[r]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float bv = redv[r];
        int bi = redi[r];
#pragma unroll
        for (int wv = 1; wv < NT / 64; ++wv) {
            const float ov = redv[wv * R + r];
            const int oi = redi[wv * R + r];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        best[r] = bv; besti[r] = bi;
    }
    __syncthreads();
}

// workgroup-wide sum / max (every thread gets the result); red: LDS [NT/64]; contain two __syncthreads()
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < NT / 64; ++wv) s += red[wv];
    __syncthreads();
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red, int tid) {
    v = wave_max(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int wv = 1; wv < NT / 64; ++wv) s = fmaxf(s, red[wv]);
    __syncthreads();
    return s;
}
// counter-based uniform in [0,1): splitmix64 finaliser of (seed, row, step); 24 random bits
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned row, unsigned step) {
    unsigned long long z = seed + ((unsigned long long)row << 32 | step) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

#include "decode_group.inc.h"
#include "decode_group8.inc.h"
#include "decode_group16.inc.h"
#include "beam_group.inc.h"

// KR / KL > 0 (fast path for R == 1, L == 1, H <= 256: thread j owns hidden unit j for the whole loop):
// rows [0,KR) of WhhT stay in the thread's registers and rows [KR,KR+KL) in LDS for all steps, so only
// H-KR-KL rows are streamed from L2 per step.  The fmaf chain still runs k = 0..H-1 in order, so the
// results are bit-identical to the streaming-only kernel.
template <int R, int KR, int KL, bool SAMPLE = false>
__global__ __launch_bounds__(NT) void decode_kernel(DecodeParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const StepWeights& w = p.w;
    const int H = w.H, L = w.L, B = p.B, T = p.T, V = w.V;
    float* hs = smem;                          // [2][L][R][H]  double-buffered h
    float* cs = hs + 2 * L * R * H;            // [L][R][H]
    float* redv = cs + L * R * H;              // [4][R]
    int* redi = reinterpret_cast<int*>(redv + 4 * R);   // [4][R]
    int* tok_s = redi + 4 * R;                 // [R]
    int* fin_s = tok_s + R;                    // [R]
    float* psum = reinterpret_cast<float*>(fin_s + R + ((4 - ((10 * R) & 3)) & 3));   // [R][128][4], 16-byte aligned
    float* wl = psum + R * 512;                         // [KL][4H] resident rows of WhhT[0]
    float* lg = wl + (size_t)KL * 4 * H;                // [R][Vp], only with I2L_SELECT_SOFTMAX / _SAMPLE

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * R;
    int grow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) grow[r] = min(row0 + r, B - 1);

    for (int idx = tid; idx < L * R * H; idx += NT) {
        const int l = idx / (R * H);
        const int rem = idx - l * (R * H);
        const int r = rem / H, j = rem - r * H;
        const size_t g = ((size_t)l * B + min(row0 + r, B - 1)) * H + j;
        hs[idx] = p.h0 ? p.h0[g] : 0.f;
        cs[idx] = p.c0 ? p.c0[g] : 0.f;
    }
    if (tid < R) {
        tok_s[tid] = p.tok0[min(row0 + tid, B - 1)];
        fin_s[tid] = 0;
    }
    // resident slices of WhhT[0] (loaded once per launch)
    float4 wres[KR > 0 ? KR : 1];
    const int jj = min(tid, H - 1);
    if (KR > 0) {
#pragma unroll
        for (int k = 0; k < KR; ++k) wres[k] = *reinterpret_cast<const float4*>(w.WhhT[0] + (size_t)k * 4 * H + 4 * jj);
    }
    float4 genc_reg[R];                                   // image half of the layer-0 gates: constant over the loop
#pragma unroll
    for (int r = 0; r < R; ++r)
        genc_reg[r] = (KR + KL > 0) ? *reinterpret_cast<const float4*>(w.Genc + (size_t)grow[r] * 4 * H + 4 * jj)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    if (KL > 0) {
        const float4* src = reinterpret_cast<const float4*>(w.WhhT[0] + (size_t)KR * 4 * H);
        for (int idx = tid; idx < KL * H; idx += NT) reinterpret_cast<float4*>(wl)[idx] = src[idx];
    }
    __syncthreads();

    int par = 0;
    int t = 0;
    for (; t < T; ++t) {
        int tok[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int tk = p.forced ? p.forced[(size_t)grow[r] * T + t] : tok_s[r];
            tok[r] = min(max(tk, 0), V - 1);
        }
        float4 pre[PF];                                   // first batch of the projection stream (fast path)
        if (KR + KL > 0)
            load_batch<float4>(pre, w.WoutT + (size_t)(tid >> 7) * (H >> 1) * w.Vp + (tid & 127) * 4, (size_t)w.Vp);
        if (KR + KL > 0) {
            // L == 1, H <= NT: one LSTM layer, thread tid = hidden unit, R rows share every weight read
            const size_t G = 4 * (size_t)H;
            const float* h_old = hs + (size_t)par * R * H;
            float* h_new = hs + (size_t)(par ^ 1) * R * H;
            if (tid < H) {
                // the token-dependent row of P is only needed at the end of the chain: its L2 latency hides
                // behind the resident rows (the sum starts from the image half, which never changes)
                float4 a[R], acc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    a[r] = *reinterpret_cast<const float4*>(w.P + (size_t)tok[r] * G + 4 * tid);
                    acc[r] = genc_reg[r];
                }
#pragma unroll
                for (int k4 = 0; k4 < KR; k4 += 4) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float4 xa = *reinterpret_cast<const float4*>(h_old + r * H + k4);
                        const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[r].x = fmaf(wres[k4 + i].x, xv[i], acc[r].x);
                            acc[r].y = fmaf(wres[k4 + i].y, xv[i], acc[r].y);
                            acc[r].z = fmaf(wres[k4 + i].z, xv[i], acc[r].z);
                            acc[r].w = fmaf(wres[k4 + i].w, xv[i], acc[r].w);
                        }
                    }
                }
#pragma unroll 2
                for (int k4 = 0; k4 < KL; k4 += 4) {
                    float4 wv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) wv[i] = *reinterpret_cast<const float4*>(wl + (size_t)(k4 + i) * G + 4 * tid);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float4 xa = *reinterpret_cast<const float4*>(h_old + r * H + KR + k4);
                        const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[r].x = fmaf(wv[i].x, xv[i], acc[r].x);
                            acc[r].y = fmaf(wv[i].y, xv[i], acc[r].y);
                            acc[r].z = fmaf(wv[i].z, xv[i], acc[r].z);
                            acc[r].w = fmaf(wv[i].w, xv[i], acc[r].w);
                        }
                    }
                }
                matvec<R, float4>(acc, w.WhhT[0] + (size_t)(KR + KL) * G + 4 * tid, G, h_old + KR + KL, H, H - KR - KL);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r].x += a[r].x; acc[r].y += a[r].y; acc[r].z += a[r].z; acc[r].w += a[r].w;
                    const float ig = sigmoidf_(acc[r].x), fg = sigmoidf_(acc[r].y);
                    const float gg = tanhf_(acc[r].z), og = sigmoidf_(acc[r].w);
                    const float cn = fg * cs[r * H + tid] + ig * gg;
                    cs[r * H + tid] = cn;
                    h_new[r * H + tid] = og * tanhf_(cn);
                }
            }
            __syncthreads();
        } else {
            lstm_layers<R>(w, tok, grow, hs, cs, cs, par, tid);
        }

        // ---------------- output projection + token selection
        const float* h_top = hs + ((size_t)(par ^ 1) * L + (L - 1)) * R * H;
        float* lp[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
            lp[r] = (p.logits && row0 + r < B) ? p.logits + ((size_t)(row0 + r) * T + t) * V : nullptr;
        float best[R];
        int besti[R];
        if (KR + KL > 0)
            project<R, true>(w, h_top, lp, p.select != I2L_SELECT_LOGITS ? lg : nullptr, p.use_temp != 0,
                             p.temperature, best, besti, psum, tid, &pre);
        else
            project<R>(w, h_top, lp, p.select != I2L_SELECT_LOGITS ? lg : nullptr, p.use_temp != 0, p.temperature,
                       best, besti, psum, tid);
        block_argmax<R>(best, besti, redv, redi, tid);
        if (p.select == I2L_SELECT_SOFTMAX) {
            // argmax(softmax(x)): probabilities in fp32 as torch.softmax (exp(x - max) / sum), first index wins
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float s = 0.f;
                for (int v = tid; v < V; v += NT) s += expf(lg[r * w.Vp + v] - best[r]);
                s = wave_sum(s);
                if (lane == 0) redv[wave * R + r] = s;
            }
            __syncthreads();
            float pbest[R];
            int pbesti[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < NT / 64; ++wv) s += redv[wv * R + r];
                pbest[r] = -1.f; pbesti[r] = 0x7fffffff;
                for (int v = tid; v < V; v += NT) {
                    const float pr = expf(lg[r * w.Vp + v] - best[r]) / s;
                    if (pr > pbest[r]) { pbest[r] = pr; pbesti[r] = v; }
                }
            }
            __syncthreads();
            block_argmax<R>(pbest, pbesti, redv, redi, tid);
#pragma unroll
            for (int r = 0; r < R; ++r) besti[r] = pbesti[r];
        }
        if (SAMPLE) {
            // predictor.py:295-331: probs = softmax(logits / T); top-k mask (keep p >= k-th largest); top-p mask
            // (drop a token once the probability mass sorted before it exceeds top_p); renormalise; one multinomial
            // draw by inverse CDF with a counter-based uniform of (seed, row, step).
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float* pr = lg + (size_t)r * w.Vp;
                float sm = 0.f;
                for (int v = tid; v < V; v += NT) sm += expf(pr[v] - best[r]);
                sm = block_sum(sm, redv, tid);
                for (int v = tid; v < V; v += NT) pr[v] = expf(pr[v] - best[r]) / sm;
                __syncthreads();
                if (p.top_k > 0) {
                    const int kk = min(p.top_k, V);
                    float kth = -1.f;
                    for (int v = tid; v < V; v += NT) {
                        const float pv = pr[v];
                        int gt = 0, ge = 0;
                        for (int u = 0; u < V; ++u) { const float pu = pr[u]; gt += pu > pv; ge += pu >= pv; }
                        if (gt < kk && kk <= ge) kth = pv;
                    }
                    kth = block_max(kth, redv, tid);
                    float s2 = 0.f;
                    for (int v = tid; v < V; v += NT) { const float q = pr[v] < kth ? 0.f : pr[v]; pr[v] = q; s2 += q; }
                    s2 = block_sum(s2, redv, tid);
                    if (s2 > 0.f) for (int v = tid; v < V; v += NT) pr[v] = pr[v] / s2;
                    __syncthreads();
                }
                if (p.top_p > 0.f) {
                    float* nx = lg + (size_t)(R + r) * w.Vp;         // second buffer: masked values
                    float s2 = 0.f;
                    for (int v = tid; v < V; v += NT) {
                        const float pv = pr[v];
                        float before = 0.f;                          // mass sorted ahead of v (descending, index breaks ties)
                        for (int u = 0; u < V; ++u) { const float pu = pr[u]; if (pu > pv || (pu == pv && u < v)) before += pu; }
                        const float q = before > p.top_p ? 0.f : pv;
                        nx[v] = q;
                        s2 += q;
                    }
                    __syncthreads();
                    for (int v = tid; v < V; v += NT) pr[v] = nx[v];
                    s2 = block_sum(s2, redv, tid);
                    if (s2 > 0.f) for (int v = tid; v < V; v += NT) pr[v] = pr[v] / s2;
                    __syncthreads();
                }
                if (p.probs && row0 + r < B) {
                    float* po = p.probs + ((size_t)(row0 + r) * T + t) * V;
                    for (int v = tid; v < V; v += NT) po[v] = pr[v];
                }
                // inverse CDF: thread t owns the contiguous chunk [t*CH, (t+1)*CH)
                const int CH = (V + NT - 1) / NT;
                float part = 0.f;
                for (int v = tid * CH; v < min(V, (tid + 1) * CH); ++v) part += pr[v];
                float* scan = psum;                                  // [NT] scratch (psum is free here)
                scan[tid] = part;
                __syncthreads();
                if (tid == 0) {
                    float run = 0.f;
                    for (int i = 0; i < NT; ++i) { const float x = scan[i]; scan[i] = run; run += x; }
                    scan[NT] = run;
                }
                __syncthreads();
                const float total = scan[NT];
                const float target = uniform01(p.seed, (unsigned)(row0 + r), (unsigned)t) * total;
                int pick = 0x7fffffff;
                float run = scan[tid];
                for (int v = tid * CH; v < min(V, (tid + 1) * CH); ++v) {
                    const float q = pr[v];
                    if (q > 0.f && run + q > target && pick == 0x7fffffff && run <= target) pick = v;
                    run += q;
                }
                float dummy = pick == 0x7fffffff ? -1.f : 1.f;
                int pk = pick;
                float bv[1] = {dummy};
                int bi[1] = {pk};
                block_argmax<1>(bv, bi, redv, redi, tid);            // the (unique) thread that found the crossing wins
                besti[r] = bi[0] < V ? bi[0] : besti[r];             // rounding corner: fall back to the arg max
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = row0 + r;
                const int sel = besti[r] < V ? besti[r] : 0;
                const int was_fin = fin_s[r];
                if (p.ids && row < B) p.ids[(size_t)row * T + t] = (p.stop == I2L_STOP_STICKY && was_fin) ? -1 : sel;
                tok_s[r] = sel;
                if (sel == p.end_id || row >= B) fin_s[r] = 1;
            }
        }
        __syncthreads();
        par ^= 1;
        if (p.stop == I2L_STOP_STICKY) {
            bool all = true;
#pragma unroll
            for (int r = 0; r < R; ++r) all = all && (fin_s[r] != 0);
            if (all) { ++t; break; }
        }
    }
    // steps never executed (sticky stop): ids = -1
    if (p.ids && t < T) {
        for (int idx = tid; idx < R * (T - t); idx += NT) {
            const int r = idx / (T - t), tt = t + (idx - r * (T - t));
            if (row0 + r < B) p.ids[(size_t)(row0 + r) * T + tt] = -1;
        }
    }
    if (p.h_out || p.c_out) {
        for (int idx = tid; idx < L * R * H; idx += NT) {
            const int l = idx / (R * H);
            const int rem = idx - l * (R * H);
            const int r = rem / H, j = rem - r * H;
            if (row0 + r < B) {
                const size_t g = ((size_t)l * B + row0 + r) * H + j;
                if (p.h_out) p.h_out[g] = hs[(size_t)par * L * R * H + idx];
                if (p.c_out) p.c_out[g] = cs[idx];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Beam search: one workgroup per image, K beams = K rows of the step (seq2seq.py:234-298 at B=1).
// ---------------------------------------------------------------------------------------------
struct BeamParams {
    StepWeights w;
    int images, T, start_id, end_id;
    int32_t* tokhist;    // [images][T][K]  token chosen for slot q at step t
    int32_t* parhist;    // [images][T][K]  slot (of step t-1) it extends
    int32_t* seq_out;    // [images][T+1]
    int32_t* len_out;    // [images]
    double* score_out;   // [images] or null
};

template <int K>
__global__ __launch_bounds__(NT) void beam_kernel(BeamParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const StepWeights& w = p.w;
    const int H = w.H, L = w.L, T = p.T, V = w.V, Vp = w.Vp;
    const int LKH = L * K * H;
    float* hs = smem;                                   // [2][L][K][H]
    float* cs = hs + 2 * LKH;                           // [2][L][K][H]
    float* lg = cs + 2 * LKH;                           // [K][Vp]
    float* psum = lg + (size_t)K * Vp;                  // [K][128][4]
    double* score = reinterpret_cast<double*>(psum + K * 512);   // [K]   (offset is a multiple of 8 bytes)
    double* nscore = score + K;                         // [K]
    float* topv = reinterpret_cast<float*>(nscore + K); // [K][K] log-probs, descending
    int* topi = reinterpret_cast<int*>(topv + K * K);   // [K][K]
    int* last = topi + K * K;                           // [K] last token of each beam
    int* live = last + K;                               // [K]
    int* npar = live + K;                               // [K] parent slot of the new beams
    int* ntok = npar + K;                               // [K]
    int* ctl = ntok + K;                                // [0]=nb [1]=nlive [2]=done

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = blockIdx.x;
    int32_t* tokhist = p.tokhist + (size_t)img * T * K;
    int32_t* parhist = p.parhist + (size_t)img * T * K;

    for (int idx = tid; idx < 2 * LKH; idx += NT) { hs[idx] = 0.f; cs[idx] = 0.f; }
    if (tid == 0) {
        for (int q = 0; q < K; ++q) { score[q] = 0.0; last[q] = p.start_id; live[q] = 0; }
        ctl[0] = 1; ctl[1] = 0; ctl[2] = 0;
    }
    // best completed beam so far (only thread 0 uses these)
    bool has_c = false;
    double best_c = 0.0;
    int best_t = -1, best_q = 0;
    int t_last = -1;                                    // last history row written
    __syncthreads();

    int par = 0, cpar = 0;
    for (int t = 0; t < T; ++t) {
        // (a) beams whose last token is END retire into `completed` (seq2seq.py:258-260), in beam order
        if (tid == 0) {
            const int nb = ctl[0];
            int nl = 0;
            for (int q = 0; q < K; ++q) {
                int lv = 0;
                if (q < nb) {
                    if (last[q] == p.end_id) {
                        if (!has_c || score[q] > best_c) { has_c = true; best_c = score[q]; best_t = t - 1; best_q = q; }
                    } else { lv = 1; ++nl; }
                }
                live[q] = lv;
            }
            ctl[1] = nl;
        }
        __syncthreads();
        if (ctl[1] == 0) break;                          // `if not candidates: break` (:276-277)

        // (b) one decoder step for the K slots (retired / unused slots compute on a clamped token, ignored)
        int tok[K], grow[K];
#pragma unroll
        for (int r = 0; r < K; ++r) { tok[r] = min(max(last[r], 0), V - 1); grow[r] = img; }
        lstm_layers<K>(w, tok, grow, hs, cs + (size_t)cpar * LKH, cs + (size_t)(cpar ^ 1) * LKH, par, tid);
        const float* h_top = hs + ((size_t)(par ^ 1) * L + (L - 1)) * K * H;
        float* lp[K];
#pragma unroll
        for (int r = 0; r < K; ++r) lp[r] = nullptr;
        float best[K];
        int besti[K];
        project<K>(w, h_top, lp, lg, false, 1.f, best, besti, psum, tid);
        __syncthreads();

        // (c) log_softmax (fp32, :266) and top-K (:267: descending, lower index first on ties); wave per row
        for (int r = wave; r < K; r += NT / 64) {
            if (!live[r]) continue;                      // wave-uniform
            const float* x = lg + (size_t)r * Vp;
            float m = -INFINITY;
            for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
            m = wave_max(m);
            float s = 0.f;
            for (int v = lane; v < V; v += 64) s += expf(x[v] - m);
            s = wave_sum(s);
            const float lse = logf(s);
            unsigned taken = 0;                          // bit i: element lane + 64*i already selected
            for (int j = 0; j < K; ++j) {
                float bv = -INFINITY;
                int bi = 0x7fffffff;
                for (int v = lane, i = 0; v < V; v += 64, ++i)
                    if (!((taken >> i) & 1u) && x[v] > bv) { bv = x[v]; bi = v; }
                wave_argmax(bv, bi);
                if (bi < V && (bi & 63) == lane) taken |= 1u << (bi >> 6);
                if (lane == 0) { topv[r * K + j] = (bv - m) - lse; topi[r * K + j] = bi < V ? bi : 0; }
            }
        }
        __syncthreads();

        // (d) candidates in (beam, rank) order, fp64 scores, stable descending selection of K (:273-280)
        if (tid == 0) {
            const int nlive = ctl[1];
            const int ncand = nlive * K;
            const int nnew = ncand < K ? ncand : K;
            unsigned long long used = 0ull;              // K*K <= 64 candidate flags
            for (int q = 0; q < nnew; ++q) {
                double bs = 0.0;
                int bc = -1;
                for (int s0 = 0; s0 < K; ++s0) {
                    if (!live[s0]) continue;
                    for (int j = 0; j < K; ++j) {
                        const int c = s0 * K + j;
                        if ((used >> c) & 1ull) continue;
                        const double sc = score[s0] + (double)topv[c];
                        if (bc < 0 || sc > bs) { bs = sc; bc = c; }
                    }
                }
                used |= 1ull << bc;
                npar[q] = bc / K;
                ntok[q] = topi[bc];
                nscore[q] = bs;
                tokhist[(size_t)t * K + q] = topi[bc];
                parhist[(size_t)t * K + q] = bc / K;
            }
            for (int q = nnew; q < K; ++q) { npar[q] = 0; ntok[q] = 0; nscore[q] = 0.0; }
            ctl[0] = nnew;
            t_last = t;
        }
        __syncthreads();

        // (e) new beam q inherits the fresh state of its parent (hidden.clone(), :272): gather into the other buffers
        {
            const float* hsrc = hs + (size_t)(par ^ 1) * LKH;
            float* hdst = hs + (size_t)par * LKH;
            const float* csrc = cs + (size_t)(cpar ^ 1) * LKH;
            float* cdst = cs + (size_t)cpar * LKH;
            for (int idx = tid; idx < LKH; idx += NT) {
                const int l = idx / (K * H);
                const int rem = idx - l * (K * H);
                const int q = rem / H, j = rem - q * H;
                const size_t src = ((size_t)l * K + npar[q]) * H + j;
                hdst[idx] = hsrc[src];
                cdst[idx] = csrc[src];
            }
        }
        __syncthreads();
        // par / cpar unchanged: the gathered state sits in hs[par], cs[cpar] again
        if (tid == 0) {
            const int nb = ctl[0];
            bool all_end = true;
            for (int q = 0; q < K; ++q) {
                score[q] = nscore[q];
                last[q] = ntok[q];
                if (q < nb && ntok[q] != p.end_id) all_end = false;
            }
            if (all_end) {                               // completed.extend(beams); break   (:282-284)
                for (int q = 0; q < nb; ++q)
                    if (!has_c || score[q] > best_c) { has_c = true; best_c = score[q]; best_t = t; best_q = q; }
                ctl[2] = 1;
            }
        }
        __syncthreads();
        if (ctl[2]) break;
    }

    // result: max(completed) (first on ties) else beams[0]; strip START, cut at END (:286-297)
    if (tid == 0) {
        int tt = has_c ? best_t : t_last, q = has_c ? best_q : 0;
        const double sc = has_c ? best_c : score[0];
        int32_t* out = p.seq_out + (size_t)img * (T + 1);
        int n = tt + 1;                                  // tokens after START
        for (int pos = tt; pos >= 0; --pos) {
            out[pos] = tokhist[(size_t)pos * K + q];
            q = parhist[(size_t)pos * K + q];
        }
        int len = n;
        for (int i = 0; i < n; ++i)
            if (out[i] == p.end_id) { len = i; break; }
        for (int i = len; i < T + 1; ++i) out[i] = -1;
        p.len_out[img] = len;
        if (p.score_out) p.score_out[img] = sc;
    }
}

int check_weights(const i2l_decoder_weights* w) {
    if (!w || !w->embedding || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh || !w->w_out || !w->b_out)
        return I2L_ERR_ARG;
    if (w->vocab <= 0 || w->embed <= 0 || w->hidden <= 0 || w->layers <= 0) return I2L_ERR_ARG;
    if (w->layers > MAXL || w->hidden % 64 != 0 || w->hidden > 2048 || w->embed % 4 != 0) return I2L_ERR_UNSUPPORTED;
    for (int l = 0; l < w->layers; ++l)
        if (!w->w_ih[l] || !w->w_hh[l] || !w->b_ih[l] || !w->b_hh[l]) return I2L_ERR_ARG;
    return I2L_OK;
}

StepWeights step_weights(const Layout& lo, const char* base, int V, int H, int L) {
    auto F = [&](size_t off) { return reinterpret_cast<const float*>(base + off); };
    StepWeights w{};
    w.H = H; w.L = L; w.V = V; w.Vp = lo.Vp;
    w.P = F(lo.P); w.Genc = F(lo.Genc);
    for (int l = 0; l < L; ++l) {
        w.WhhT[l] = F(lo.WhhT[l]);
        w.WihT[l] = l > 0 ? F(lo.WihT[l]) : nullptr;
        w.biasP[l] = l > 0 ? F(lo.biasP[l]) : nullptr;
    }
    w.WoutT = F(lo.WoutT); w.boutP = F(lo.boutP);
    w.wpack16 = lo.wpack16 ? base + lo.wpack16 : nullptr;
    return w;
}

constexpr int RES_KR = 76, RES_KL = 36;   // resident rows of WhhT[0]: registers / LDS (fast path)
constexpr int RES2_KR = 64, RES2_KL = 32; // same with two rows per workgroup (a few more live registers)

size_t decode_lds_bytes(int R, int L, int H, int Vp, int select, int KL = 0) {
    size_t floats = (size_t)3 * L * R * H + 4 * R + 4 * R + R + R + 4 + (size_t)R * 512 + (size_t)KL * 4 * H;
    if (select != I2L_SELECT_LOGITS) floats += (size_t)R * Vp;
    if (select == I2L_SELECT_SAMPLE) floats += (size_t)R * Vp;
    return floats * sizeof(float);
}

}  // namespace

extern "C" size_t i2l_decoder_workspace_bytes(int rows, int vocab, int embed, int hidden, int layers) {
    if (rows <= 0 || vocab <= 0 || embed <= 0 || hidden <= 0 || layers <= 0 || layers > MAXL) return 0;
    return make_layout(rows, vocab, embed, hidden, layers).total;
}

extern "C" int i2l_decoder_prepare(const i2l_decoder_weights* w, const float* enc, int rows, int what,
                                   void* workspace, size_t workspace_bytes, i2l_stream_t stream) {
    int rc = check_weights(w);
    if (rc != I2L_OK) return rc;
    if (rows <= 0 || (what & ~I2L_PREP_ALL) || what == 0) return I2L_ERR_ARG;
    if ((what & I2L_PREP_ROWS) && !enc) return I2L_ERR_ARG;
    const int V = w->vocab, E = w->embed, H = w->hidden, L = w->layers;
    const Layout lo = make_layout(rows, V, E, H, L);
    if (!workspace || workspace_bytes < lo.total) return I2L_ERR_WORKSPACE;
    char* base = static_cast<char*>(workspace);
    hipStream_t s = i2l_s(stream);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    const int G = 4 * H;

    for (int l = 0; l < L && (what & I2L_PREP_WEIGHTS); ++l) {
        dim3 grid(i2l_cdiv(G, 64), i2l_cdiv(H, 64));
        hipLaunchKernelGGL(transpose_perm_kernel, grid, dim3(256), 0, s, w->w_hh[l], G, H, F(lo.WhhT[l]), G, G, H);
        I2L_CHECK_LAUNCH();
        if (l > 0) {
            hipLaunchKernelGGL(transpose_perm_kernel, grid, dim3(256), 0, s, w->w_ih[l], G, H, F(lo.WihT[l]), G, G, H);
            I2L_CHECK_LAUNCH();
            hipLaunchKernelGGL(bias_perm_kernel, dim3(i2l_cdiv(G, 256)), dim3(256), 0, s, w->b_ih[l], w->b_hh[l],
                               F(lo.biasP[l]), H);
            I2L_CHECK_LAUNCH();
        }
    }
    if (what & I2L_PREP_WEIGHTS) {
        dim3 grid(i2l_cdiv(lo.Vp, 64), i2l_cdiv(H, 64));
        hipLaunchKernelGGL(transpose_perm_kernel, grid, dim3(256), 0, s, w->w_out, V, H, F(lo.WoutT), lo.Vp, lo.Vp, 0);
        I2L_CHECK_LAUNCH();
        hipLaunchKernelGGL(bout_pad_kernel, dim3(i2l_cdiv(lo.Vp, 256)), dim3(256), 0, s, w->b_out, F(lo.boutP), V, lo.Vp);
        I2L_CHECK_LAUNCH();
        if (lo.wpack16) {
            hipLaunchKernelGGL(pack16_kernel, dim3(16 * 4), dim3(64), 0, s, F(lo.WhhT[0]), F(lo.WoutT),
                               reinterpret_cast<u32x4_t*>(base + lo.wpack16));
            I2L_CHECK_LAUNCH();
        }
    }
    const size_t gws = lo.total - lo.gemm_ws;
    if (what & I2L_PREP_WEIGHTS) {   // P = Emb @ W_ih_0[:, :E]^T, gate-interleaved columns
        GemmArgs g = gemm_args();
        g.A = w->embedding; g.lda = E;
        g.W = w->w_ih[0]; g.ldw = 2 * E;
        g.C = F(lo.P); g.ldc = G;
        g.M = V; g.N = G; g.K = E; g.perm_h = H;
        rc = i2l_gemm(g, base + lo.gemm_ws, gws, s);
        if (rc != I2L_OK) return rc;
    }
    if (what & I2L_PREP_ROWS) {   // Genc = enc @ W_ih_0[:, E:]^T + b_ih_0 + b_hh_0
        GemmArgs g = gemm_args();
        g.A = enc; g.lda = E;
        g.W = w->w_ih[0] + E; g.ldw = 2 * E;
        g.bias = w->b_ih[0]; g.bias2 = w->b_hh[0];
        g.C = F(lo.Genc); g.ldc = G;
        g.M = rows; g.N = G; g.K = E; g.perm_h = H;
        rc = i2l_gemm(g, base + lo.gemm_ws, gws, s);
        if (rc != I2L_OK) return rc;
    }
    return I2L_OK;
}

namespace {
int launch_decode(const i2l_decoder_weights* w, const void* workspace, int rows, int steps, const int32_t* tok0,
                  const int32_t* forced, const float* h0, const float* c0, float temperature, int select, int stop,
                  int end_id, int top_k, float top_p, unsigned long long seed, int32_t* ids_out, float* logits_out,
                  float* probs_out, float* h_out, float* c_out, i2l_stream_t stream, int rows_per_wg = 0, int flags = 0,
                  uint32_t* resident_flag = nullptr, uint32_t resident_value = 0);
}

extern "C" int i2l_greedy_decode_ex(const i2l_decoder_weights* w, const void* workspace, int rows, int steps,
                                    const int32_t* tok0, const int32_t* forced, const float* h0, const float* c0,
                                    float temperature, int select, int stop, int end_id, int rows_per_workgroup,
                                    int32_t* ids_out, float* logits_out, float* h_out, float* c_out, int flags,
                                    uint32_t* resident_flag, uint32_t resident_value, i2l_stream_t stream) {
    if (select != I2L_SELECT_LOGITS && select != I2L_SELECT_SOFTMAX) return I2L_ERR_ARG;
    if (rows_per_workgroup != 0 && rows_per_workgroup != 1 && rows_per_workgroup != 2 && rows_per_workgroup != 4)
        return I2L_ERR_ARG;
    return launch_decode(w, workspace, rows, steps, tok0, forced, h0, c0, temperature, select, stop, end_id, 0, 0.f, 0ull,
                         ids_out, logits_out, nullptr, h_out, c_out, stream, rows_per_workgroup, flags, resident_flag,
                         resident_value);
}

extern "C" int i2l_greedy_decode(const i2l_decoder_weights* w, const void* workspace, int rows, int steps,
                                 const int32_t* tok0, const int32_t* forced, const float* h0, const float* c0,
                                 float temperature, int select, int stop, int end_id, int32_t* ids_out,
                                 float* logits_out, float* h_out, float* c_out, i2l_stream_t stream) {
    if (select != I2L_SELECT_LOGITS && select != I2L_SELECT_SOFTMAX) return I2L_ERR_ARG;
    return launch_decode(w, workspace, rows, steps, tok0, forced, h0, c0, temperature, select, stop, end_id, 0, 0.f, 0ull,
                         ids_out, logits_out, nullptr, h_out, c_out, stream);
}

extern "C" int i2l_sample_decode(const i2l_decoder_weights* w, const void* workspace, int rows, int steps,
                                 const int32_t* tok0, const float* h0, const float* c0, float temperature, int top_k,
                                 float top_p, uint64_t seed, int stop, int end_id, int32_t* ids_out, float* probs_out,
                                 float* h_out, float* c_out, i2l_stream_t stream) {
    if (!(temperature > 0.f) || top_k < 0 || top_p < 0.f || (top_k == 0 && top_p == 0.f)) return I2L_ERR_ARG;
    if (w && w->vocab > 8 * 256) return I2L_ERR_UNSUPPORTED;
    return launch_decode(w, workspace, rows, steps, tok0, nullptr, h0, c0, temperature, I2L_SELECT_SAMPLE, stop, end_id,
                         top_k, top_p, (unsigned long long)seed, ids_out, nullptr, probs_out, h_out, c_out, stream);
}

namespace {
// the residency signal of a launch that does NOT run a grouped kernel (other dimensions, row-per-workgroup request): its
// workgroups need no partners, so "resident" is said at once -- a waiter never pays its time-out for a kernel choice
__global__ void publish_value32_kernel(unsigned* flag, unsigned value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

int launch_decode(const i2l_decoder_weights* w, const void* workspace, int rows, int steps, const int32_t* tok0,
                  const int32_t* forced, const float* h0, const float* c0, float temperature, int select, int stop,
                  int end_id, int top_k, float top_p, unsigned long long seed, int32_t* ids_out, float* logits_out,
                  float* probs_out, float* h_out, float* c_out, i2l_stream_t stream, int rows_per_wg, int flags,
                  uint32_t* resident_flag, uint32_t resident_value) {
    int rc = check_weights(w);
    if (rc != I2L_OK) return rc;
    if (!workspace || !tok0 || rows <= 0 || steps <= 0) return I2L_ERR_ARG;
    if ((h0 == nullptr) != (c0 == nullptr)) return I2L_ERR_ARG;
    if (stop != I2L_STOP_NONE && stop != I2L_STOP_STICKY) return I2L_ERR_ARG;
    const int V = w->vocab, E = w->embed, H = w->hidden, L = w->layers;
    const Layout lo = make_layout(rows, V, E, H, L);
    const char* base = static_cast<const char*>(workspace);

    DecodeParams p{};
    p.w = step_weights(lo, base, V, H, L);
    p.B = rows; p.T = steps;
    p.tok0 = tok0; p.forced = forced; p.h0 = h0; p.c0 = c0;
    p.h_out = h_out; p.c_out = c_out; p.ids = ids_out; p.logits = logits_out;
    p.temperature = temperature; p.use_temp = (temperature != 1.0f) ? 1 : 0;
    p.select = select; p.stop = stop; p.end_id = end_id;
    p.top_k = top_k; p.top_p = top_p; p.seed = seed; p.probs = probs_out;

    // grouped kernel: 4 workgroups share 4 rows, weights fully on chip (decode_group.inc.h)
    // I2L_SELECT_SOFTMAX takes it too: softmax is monotone, so its arg max is the logits' arg max unless fp32 rounding
    // makes two DIFFERENT scaled logits equal probabilities, which needs the two largest to be adjacent floats AND the
    // largest below 2 in magnitude (exp(x2 - x1) is then within 2 ulp of 1); the row-per-workgroup kernel evaluates
    // the probabilities literally and is one argument away (rows_per_workgroup != 0)
    if (rows_per_wg == 0 && lo.xchg_bytes && (select == I2L_SELECT_LOGITS || select == I2L_SELECT_SOFTMAX) && !h0 &&
        !h_out && !c_out &&
        steps >= 8 && steps <= 65000) {   // the candidate granule carries step + 1 in 16 bits
        if ((flags & I2L_FLAG_DECODE_GROUP16) && !logits_out && !forced && ids_out) {
            // sixteen members x sixteen rows, the per-step products on the matrix cores (decode_group16.inc.h)
            static_assert(GROUP16_XCHG_PER_GROUP == GROUP16_XCHG_BYTES, "exchange region sizing");
            GroupParams gp{};
            gp.w = p.w; gp.B = rows; gp.T = steps; gp.n_groups = i2l_cdiv(rows, 16);
            gp.tok0 = tok0; gp.forced = nullptr; gp.ids = ids_out; gp.logits = nullptr;
            gp.temperature = temperature; gp.use_temp = p.use_temp; gp.stop = stop; gp.end_id = end_id;
            char* xb = const_cast<char*>(base) + lo.xchg;
            gp.status = reinterpret_cast<unsigned*>(xb);
            gp.xchg = reinterpret_cast<u64_t*>(xb + GROUP_STATUS_BYTES);
            gp.opts = group_opts(steps, flags);
            gp.resident_flag = resident_flag; gp.resident_value = resident_value;
            hipStream_t gs = i2l_s(stream);
            static std::atomic<unsigned> attr16{0};
            if (i2l_lds_attr(reinterpret_cast<const void*>(decode_group16_kernel), GRP16_LDS, attr16)) {
                if (!(flags & I2L_FLAG_DECODE_REGION_CLEARED) && hipMemsetAsync(xb, 0, lo.xchg_bytes, gs) != hipSuccess) return I2L_ERR_LAUNCH;
                hipLaunchKernelGGL(decode_group16_kernel, dim3(i2l_cdiv(gp.n_groups, 8) * 128), dim3(G16NT), GRP16_LDS, gs, gp);
                I2L_CHECK_LAUNCH();
                return I2L_OK;
            }
        }
        if ((flags & I2L_FLAG_DECODE_GROUP8) && !logits_out && !forced && ids_out) {
            // eight members x eight rows: one wave per SIMD and ~80 KB of LDS per CU, i.e. room for a conv workgroup beside it
            static_assert(GROUP8_XCHG_PER_GROUP == 2 * GROUP_XCHG_PER_GROUP, "exchange region sizing");
            GroupParams gp{};
            gp.w = p.w; gp.B = rows; gp.T = steps; gp.n_groups = i2l_cdiv(rows, 8);
            gp.tok0 = tok0; gp.forced = nullptr; gp.ids = ids_out; gp.logits = nullptr;
            gp.temperature = temperature; gp.use_temp = p.use_temp; gp.stop = stop; gp.end_id = end_id;
            char* xb = const_cast<char*>(base) + lo.xchg;
            gp.status = reinterpret_cast<unsigned*>(xb);
            gp.xchg = reinterpret_cast<u64_t*>(xb + GROUP_STATUS_BYTES);
            gp.opts = group_opts(steps, flags);
            gp.resident_flag = resident_flag; gp.resident_value = resident_value;
            hipStream_t gs = i2l_s(stream);
            static std::atomic<unsigned> attr8{0};               // per device
            if (i2l_lds_attr(reinterpret_cast<const void*>(decode_group8_kernel), GRP8_LDS, attr8)) {
                if (!(flags & I2L_FLAG_DECODE_REGION_CLEARED) && hipMemsetAsync(xb, 0, lo.xchg_bytes, gs) != hipSuccess) return I2L_ERR_LAUNCH;
                hipLaunchKernelGGL(decode_group8_kernel, dim3(i2l_cdiv(gp.n_groups, 8) * 64), dim3(G8NT), GRP8_LDS, gs, gp);
                I2L_CHECK_LAUNCH();
                return I2L_OK;
            }
        }
        GroupParams gp{};
        gp.w = p.w; gp.B = rows; gp.T = steps; gp.n_groups = lo.n_groups;
        gp.tok0 = tok0; gp.forced = forced; gp.ids = ids_out; gp.logits = logits_out;
        gp.temperature = temperature; gp.use_temp = p.use_temp; gp.stop = stop; gp.end_id = end_id;
        char* xb = const_cast<char*>(base) + lo.xchg;           // scratch region of the workspace
        gp.status = reinterpret_cast<unsigned*>(xb);
        gp.xchg = reinterpret_cast<u64_t*>(xb + GROUP_STATUS_BYTES);
        gp.opts = group_opts(steps, flags);
        gp.resident_flag = resident_flag; gp.resident_value = resident_value;
        hipStream_t gs = i2l_s(stream);
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(decode_group_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRP_LDS) == hipSuccess) {
            if (!(flags & I2L_FLAG_DECODE_REGION_CLEARED) && hipMemsetAsync(xb, 0, lo.xchg_bytes, gs) != hipSuccess) return I2L_ERR_LAUNCH;
            hipLaunchKernelGGL(decode_group_kernel, dim3(i2l_cdiv(lo.n_groups, 8) * 32), dim3(GNT), GRP_LDS, gs, gp);
            I2L_CHECK_LAUNCH();
            return I2L_OK;
        }
    }
    if (resident_flag) {
        hipLaunchKernelGGL(publish_value32_kernel, dim3(1), dim3(1), 0, i2l_s(stream), resident_flag, resident_value);
        I2L_CHECK_LAUNCH();
    }
    // rows per workgroup: fill the 256 CUs first, then stack rows (weights are streamed once per workgroup per step)
    int R = rows <= 256 ? 1 : (rows <= 512 ? 2 : 4);
    if (rows_per_wg > 0) R = rows_per_wg;
    while (R > 1 && decode_lds_bytes(R, L, H, lo.Vp, select) > 64 * 1024) R >>= 1;
    const size_t lds = decode_lds_bytes(R, L, H, lo.Vp, select);
    if (lds > 64 * 1024) return I2L_ERR_UNSUPPORTED;
    const dim3 grid(i2l_cdiv(rows, R));
    hipStream_t s = i2l_s(stream);
    // fast path: weights partly resident on chip; pays a one-off fill, so only for real loops
    if (select == I2L_SELECT_SAMPLE) {
        if (R == 1) hipLaunchKernelGGL((decode_kernel<1, 0, 0, true>), grid, dim3(NT), lds, s, p);
        else if (R == 2) hipLaunchKernelGGL((decode_kernel<2, 0, 0, true>), grid, dim3(NT), lds, s, p);
        else hipLaunchKernelGGL((decode_kernel<4, 0, 0, true>), grid, dim3(NT), lds, s, p);
        I2L_CHECK_LAUNCH();
        return I2L_OK;
    }
    const bool resident = (R == 1 || R == 2) && L == 1 && H == 256 && steps >= 8;
    if (resident) {
        const size_t lds_r = decode_lds_bytes(R, 1, H, lo.Vp, select, R == 1 ? RES_KL : RES2_KL);
        auto kern = R == 1 ? decode_kernel<1, RES_KR, RES_KL> : decode_kernel<2, RES2_KR, RES2_KL>;
        if (lds_r <= 160 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_r) == hipSuccess) {
            hipLaunchKernelGGL(kern, grid, dim3(NT), lds_r, s, p);
            I2L_CHECK_LAUNCH();
            return I2L_OK;
        }
    }
    if (R == 1) hipLaunchKernelGGL((decode_kernel<1, 0, 0>), grid, dim3(NT), lds, s, p);
    else if (R == 2) hipLaunchKernelGGL((decode_kernel<2, 0, 0>), grid, dim3(NT), lds, s, p);
    else hipLaunchKernelGGL((decode_kernel<4, 0, 0>), grid, dim3(NT), lds, s, p);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
}  // namespace

namespace {

size_t beam_lds_bytes(int K, int L, int H, int Vp) {
    size_t b = (size_t)4 * L * K * H * sizeof(float) + (size_t)K * Vp * sizeof(float) + (size_t)K * 512 * sizeof(float);
    b = (b + 7) / 8 * 8;
    b += 2 * (size_t)K * sizeof(double) + (size_t)2 * K * K * 4 + (size_t)4 * K * 4 + 4 * 4;
    return b;
}

template <int K>
int launch_beam(const BeamParams& p, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&beam_kernel<K>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return I2L_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(beam_kernel<K>, dim3(p.images), dim3(NT), lds, s, p);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

}  // namespace

namespace {
// grouped beam search (beam_group.inc.h): shape and beam widths it is built for
inline bool beam_group_ok(int beam, int H, int L) { return L == 1 && H == 256 && beam >= 2 && beam <= 6; }
inline int beam_groups(int images, int beam) { return i2l_cdiv(images, BG_S / beam); }
inline size_t beam_hist_bytes(int images, int beam, int steps) {
    return i2l_align((size_t)2 * images * steps * beam * sizeof(int32_t));
}
template <int K>
int launch_beam_group(const BeamGroupParams& p, hipStream_t s) {
    constexpr size_t lds = bg_lds_bytes();
    static_assert(lds <= 160 * 1024, "beam_group_kernel LDS");
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&beam_group_kernel<K>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return I2L_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(beam_group_kernel<K>, dim3(i2l_cdiv(p.n_groups, 8) * 32), dim3(GNT), lds, s, p);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
}  // namespace

extern "C" size_t i2l_beam_workspace_bytes(int images, int beam, int hidden, int layers, int steps) {
    if (images <= 0 || beam <= 0 || beam > I2L_MAX_BEAM || steps <= 0) return 0;
    size_t b = beam_hist_bytes(images, beam, steps);
    if (beam_group_ok(beam, hidden, layers))
        b += GROUP_STATUS_BYTES + (size_t)i2l_cdiv(beam_groups(images, beam), 8) * 8 * BEAM_XCHG_PER_GROUP;
    return b;
}

extern "C" int i2l_beam_decode(const i2l_decoder_weights* w, const void* workspace, int images, int beam, int steps,
                               int start_id, int end_id, void* beam_workspace, size_t beam_workspace_bytes,
                               int32_t* seq_out, int32_t* len_out, double* score_out, int flags, i2l_stream_t stream) {
    int rc = check_weights(w);
    if (rc != I2L_OK) return rc;
    if (!workspace || !beam_workspace || !seq_out || !len_out || images <= 0 || steps <= 0 || beam <= 0)
        return I2L_ERR_ARG;
    if (beam > I2L_MAX_BEAM || beam > w->vocab) return I2L_ERR_UNSUPPORTED;
    const size_t need = (size_t)2 * images * steps * beam * sizeof(int32_t);
    if (beam_workspace_bytes < need) return I2L_ERR_WORKSPACE;
    const int V = w->vocab, E = w->embed, H = w->hidden, L = w->layers;
    const Layout lo = make_layout(images, V, E, H, L);
    if (lo.Vp > 64 * 32) return I2L_ERR_UNSUPPORTED;      // top-k exclusion mask: 32 elements per lane
    hipStream_t s = i2l_s(stream);
    // grouped kernel: 4 workgroups share 12 beam slots, weights fully on chip (beam_group.inc.h); a caller falls back
    // with I2L_FLAG_NO_GROUP after a -3 result
    const bool group_on = !(flags & I2L_FLAG_NO_GROUP);
    if (group_on && beam_group_ok(beam, H, L) && V <= 512 && steps <= 65000 &&
        beam_workspace_bytes >= i2l_beam_workspace_bytes(images, beam, H, L, steps)) {
        BeamGroupParams gp{};
        gp.w = step_weights(lo, static_cast<const char*>(workspace), V, H, L);
        gp.images = images; gp.T = steps; gp.n_groups = beam_groups(images, beam);
        gp.start_id = start_id; gp.end_id = end_id;
        gp.tokhist = static_cast<int32_t*>(beam_workspace);
        gp.parhist = gp.tokhist + (size_t)images * steps * beam;
        gp.seq_out = seq_out; gp.len_out = len_out; gp.score_out = score_out;
        char* xb = static_cast<char*>(beam_workspace) + beam_hist_bytes(images, beam, steps);
        const size_t xbytes = GROUP_STATUS_BYTES + (size_t)i2l_cdiv(gp.n_groups, 8) * 8 * BEAM_XCHG_PER_GROUP;
        gp.status = reinterpret_cast<unsigned*>(xb);
        gp.xchg = reinterpret_cast<u64_t*>(xb + GROUP_STATUS_BYTES);
        gp.opts = group_opts(steps, flags);
        if (hipMemsetAsync(xb, 0, xbytes, s) != hipSuccess) return I2L_ERR_LAUNCH;
        switch (beam) {
            case 2: return launch_beam_group<2>(gp, s);
            case 3: return launch_beam_group<3>(gp, s);
            case 4: return launch_beam_group<4>(gp, s);
            case 5: return launch_beam_group<5>(gp, s);
            case 6: return launch_beam_group<6>(gp, s);
            default: break;
        }
    }
    BeamParams p{};
    p.w = step_weights(lo, static_cast<const char*>(workspace), V, H, L);
    p.images = images; p.T = steps; p.start_id = start_id; p.end_id = end_id;
    p.tokhist = static_cast<int32_t*>(beam_workspace);
    p.parhist = p.tokhist + (size_t)images * steps * beam;
    p.seq_out = seq_out; p.len_out = len_out; p.score_out = score_out;
    const size_t lds = beam_lds_bytes(beam, L, H, lo.Vp);
    if (lds > 160 * 1024) return I2L_ERR_UNSUPPORTED;
    switch (beam) {
        case 1: return launch_beam<1>(p, lds, s);
        case 2: return launch_beam<2>(p, lds, s);
        case 3: return launch_beam<3>(p, lds, s);
        case 4: return launch_beam<4>(p, lds, s);
        case 5: return launch_beam<5>(p, lds, s);
        case 6: return launch_beam<6>(p, lds, s);
        case 7: return launch_beam<7>(p, lds, s);
        case 8: return launch_beam<8>(p, lds, s);
        default: return I2L_ERR_UNSUPPORTED;
    }
}

extern "C" size_t i2l_decoder_group_status_offset(int rows, int vocab, int embed, int hidden, int layers) {
    const Layout lo = make_layout(rows, vocab, embed, hidden, layers);
    return lo.xchg_bytes ? lo.xchg : 0;
}

// Size of the region that starts there (status words + the groups' exchange granules): what a grouped launch zeroes before its
// kernel -- or the caller, earlier and on another stream, with I2L_FLAG_DECODE_REGION_CLEARED.
extern "C" size_t i2l_decoder_group_region_bytes(int rows, int vocab, int embed, int hidden, int layers) {
    return make_layout(rows, vocab, embed, hidden, layers).xchg_bytes;
}

#ifdef I2L_GROUP_STAMPS
extern "C" size_t i2l_debug_group_status_offset(int rows, int vocab, int embed, int hidden, int layers) {
    return make_layout(rows, vocab, embed, hidden, layers).xchg;
}
#endif
