// Training path of LSTMDecoder: teacher-forced forward (decoder.py:100-195) that keeps what the
// backward pass needs, label-smoothed cross entropy (trainer.py:111-115,335-336) and BPTT.
//
// Structure (DESIGN.md "training"):
//   forward   X = [dropout](cat[emb(tok), enc])                      build_x_kernel
//             GX = X @ W_ih_0^T + b_ih_0 + b_hh_0  (all B*T rows)    MFMA GEMM, gate-interleaved
//             recurrence over t, rows partitioned over workgroups     lstm_train_fwd_kernel (persistent)
//             logits = dropout(h_top) @ W_out^T + b_out               MFMA GEMM
//   loss      per-row logsumexp, loss terms, dlogits                  ce_rows_kernel + ordered reduce
//   backward  dW_out, db_out, dH                                      MFMA GEMMs (A^T B, A B forms)
//             BPTT over t (reverse), rows partitioned                 lstm_train_bwd_kernel (persistent)
//             dW_ih, dW_hh, db (all layers), dX                       MFMA GEMMs over all B*T rows at once
//             dEmb (scatter-add by token), dEnc (sum over t)          small kernels
// Only the two recurrences are sequential; every weight gradient is one large GEMM.
// Dropout masks come from a counter-based hash of (seed, stream, element index), so backward
// regenerates them instead of storing them.  With the reference's Attention over a length-1
// source the context equals enc (see decode.hip), and the attention parameters receive an
// exactly-zero gradient: the host zero-fills them.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int MAXL = I2L_MAX_LSTM_LAYERS;
constexpr int NT = 256;
constexpr int PF = 8;
constexpr int RES_KR = 96, RES_KL = 32;     // rows of W_hh kept in registers / LDS by the recurrence fast paths

// ------------------------------------------------------------------ dropout mask
__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned stream, unsigned long long idx,
                                            float p, float inv_keep) {
    if (p <= 0.f) return 1.f;
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull + (unsigned long long)stream * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.f;
}

enum { DS_X = 1, DS_OUT = 2, DS_LAYER0 = 8 };   // dropout streams: lstm input, lstm output, inter-layer l

// ------------------------------------------------------------------ X = [dropout](cat[emb, enc])
// attn_path == 0: dropout over the whole 2E row (decoder.py:130-133); 1: over the embedding half only (:162).
__global__ __launch_bounds__(256) void build_x_kernel(const float* __restrict__ emb, const float* __restrict__ enc,
                                                      const int32_t* __restrict__ tok, float* __restrict__ X, int B,
                                                      int T, int E, int V, float p, unsigned long long seed,
                                                      int attn_path) {
    const size_t total = (size_t)B * T * 2 * E;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t bt = i / (2 * E);
        const int e = (int)(i - bt * 2 * E);
        const int b = (int)(bt / T);
        float v;
        if (e < E) {
            int tk = tok[bt];
            tk = min(max(tk, 0), V - 1);
            v = emb[(size_t)tk * E + e];
            v *= keep_scale(seed, DS_X, attn_path ? bt * E + e : i, p, inv_keep);
        } else {
            v = enc[(size_t)b * E + (e - E)];
            if (!attn_path) v *= keep_scale(seed, DS_X, i, p, inv_keep);
        }
        X[i] = v;
    }
}

__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           size_t n, float p, unsigned long long seed,
                                                           unsigned stream) {
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = in[i] * keep_scale(seed, stream, i, p, inv_keep);
}

// ------------------------------------------------------------------ pipelined matvec (same scheme as decode.hip)
template <int R>
__device__ __forceinline__ void fma_rows(float4 (&acc)[R], const float4 (&w)[PF], const float* xs, int xstride, int k) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int i4 = 0; i4 < PF; i4 += 4) {
            const float4 xa = *reinterpret_cast<const float4*>(xs + r * xstride + k + i4);
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

// acc[r] += sum_{k<count} W[k][0..3] * xs[r*xstride + k];  count % (2*PF) == 0.
template <int R>
__device__ __forceinline__ void matvec(float4 (&acc)[R], const float* __restrict__ Wcol, size_t ldw, const float* xs,
                                       int xstride, int count) {
    float4 wa[PF], wb[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) wa[i] = *reinterpret_cast<const float4*>(Wcol + (size_t)i * ldw);
    int k = 0;
    for (; k + 2 * PF < count; k += 2 * PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) wb[i] = *reinterpret_cast<const float4*>(Wcol + (size_t)(k + PF + i) * ldw);
        __builtin_amdgcn_sched_barrier(0);
        fma_rows<R>(acc, wa, xs, xstride, k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PF; ++i) wa[i] = *reinterpret_cast<const float4*>(Wcol + (size_t)(k + 2 * PF + i) * ldw);
        __builtin_amdgcn_sched_barrier(0);
        fma_rows<R>(acc, wb, xs, xstride, k + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) wb[i] = *reinterpret_cast<const float4*>(Wcol + (size_t)(k + PF + i) * ldw);
    __builtin_amdgcn_sched_barrier(0);
    fma_rows<R>(acc, wa, xs, xstride, k);
    fma_rows<R>(acc, wb, xs, xstride, k + PF);
}

// One-row matvec whose first KR weight rows sit in this thread's registers and the next KL rows in LDS for the
// whole kernel (loaded once); only the remaining rows are streamed from L2 each step.  Same k order as matvec.
template <int KR, int KL>
__device__ __forceinline__ void matvec_res(float4& acc, const float4 (&wres)[KR > 0 ? KR : 1], const float* wl, int ldl,
                                           const float* __restrict__ Wcol, size_t ldw, const float* xs, int count) {
#pragma unroll
    for (int k4 = 0; k4 < KR; k4 += 4) {
        const float4 xa = *reinterpret_cast<const float4*>(xs + k4);
        const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc.x = fmaf(wres[k4 + i].x, xv[i], acc.x);
            acc.y = fmaf(wres[k4 + i].y, xv[i], acc.y);
            acc.z = fmaf(wres[k4 + i].z, xv[i], acc.z);
            acc.w = fmaf(wres[k4 + i].w, xv[i], acc.w);
        }
    }
#pragma unroll 2
    for (int k4 = 0; k4 < KL; k4 += 4) {
        const float4 xa = *reinterpret_cast<const float4*>(xs + KR + k4);
        const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 wv = *reinterpret_cast<const float4*>(wl + (size_t)(k4 + i) * ldl);
            acc.x = fmaf(wv.x, xv[i], acc.x);
            acc.y = fmaf(wv.y, xv[i], acc.y);
            acc.z = fmaf(wv.z, xv[i], acc.z);
            acc.w = fmaf(wv.w, xv[i], acc.w);
        }
    }
    float4 a1[1] = {acc};
    matvec<1>(a1, Wcol + (size_t)(KR + KL) * ldw, ldw, xs + KR + KL, 0, count - KR - KL);
    acc = a1[0];
}

// gate functions on the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32), as in decode.hip: <= ~2e-7 absolute per call
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

#include "train_group.inc.h"

// ------------------------------------------------------------------ saved tensors
struct TrainBuf {
    int B, T, H, L;
    const float* GX;            // [B*T][4H] gate-interleaved layer-0 input gates (incl. biases)
    const float* WhhT[MAXL];    // [H][4H] gate-interleaved transposes
    const float* WihT[MAXL];    // l > 0
    const float* biasP[MAXL];   // l > 0
    float* ACT[MAXL];           // [B*T][4H] interleaved activated gates (i,f,g,o)
    float* C[MAXL];             // [B*T][H]
    float* Hout[MAXL];          // [B*T][H] layer output h_t (before any dropout)
    float* Hprev[MAXL];         // [B*T][H] h_{t-1} (zeros at t = 0)
    float p;                    // dropout probability (inter-layer, only L > 1)
    unsigned long long seed;
};

// Forward recurrence: workgroup = R batch rows, all T steps, all layers (no inter-workgroup traffic).
// KR / KL > 0: fast path for R == 1, L == 1, H == NT (thread = hidden unit): part of W_hh stays on chip.
template <int R, int KR, int KL>
__global__ __launch_bounds__(NT) void lstm_train_fwd_kernel(TrainBuf p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, L = p.L, B = p.B, T = p.T;
    const size_t G = 4 * (size_t)H;
    float* hs = smem;                    // [2][L][R][H]  raw h (recurrence)
    float* hd = hs + 2 * L * R * H;      // [L][R][H]     inter-layer-dropped h of the current step
    float* cs = hd + L * R * H;          // [L][R][H]
    float* wl = cs + L * R * H;          // [KL][4H] resident rows of WhhT[0]
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * R;
    const float inv_keep = p.p > 0.f ? 1.f / (1.f - p.p) : 1.f;
    for (int idx = tid; idx < 2 * L * R * H; idx += NT) hs[idx] = 0.f;
    for (int idx = tid; idx < L * R * H; idx += NT) { cs[idx] = 0.f; hd[idx] = 0.f; }
    float4 wres[KR > 0 ? KR : 1];
    if (KR > 0) {
#pragma unroll
        for (int k = 0; k < KR; ++k) wres[k] = *reinterpret_cast<const float4*>(p.WhhT[0] + (size_t)k * G + 4 * tid);
    }
    if (KL > 0) {
        const float4* src = reinterpret_cast<const float4*>(p.WhhT[0] + (size_t)KR * G);
        for (int idx = tid; idx < KL * H; idx += NT) reinterpret_cast<float4*>(wl)[idx] = src[idx];
    }
    __syncthreads();
    int par = 0;
    for (int t = 0; t < T; ++t) {
        if (KR + KL > 0) {                       // R == 1, L == 1, H == NT
            const float* h_old = hs + (size_t)par * H;
            float* h_new = hs + (size_t)(par ^ 1) * H;
            const int row = row0;
            const size_t bt = (size_t)row * T + t;
            float4 acc = *reinterpret_cast<const float4*>(p.GX + bt * G + 4 * tid);
            matvec_res<KR, KL>(acc, wres, wl + 4 * tid, (int)G, p.WhhT[0] + 4 * tid, G, h_old, H);
            const float ig = sigmoidf_(acc.x), fg = sigmoidf_(acc.y);
            const float gg = tanhf_(acc.z), og = sigmoidf_(acc.w);
            const float cn = fg * cs[tid] + ig * gg;
            const float hn = og * tanhf_(cn);
            cs[tid] = cn;
            h_new[tid] = hn;
            *reinterpret_cast<float4*>(p.ACT[0] + bt * G + 4 * tid) = make_float4(ig, fg, gg, og);
            p.C[0][bt * H + tid] = cn;
            p.Hout[0][bt * H + tid] = hn;
            if (t == 0) p.Hprev[0][bt * H + tid] = 0.f;
            if (t + 1 < T) p.Hprev[0][(bt + 1) * H + tid] = hn;
            __syncthreads();
            par ^= 1;
            continue;
        }
        for (int l = 0; l < L; ++l) {
            const float* h_old = hs + ((size_t)par * L + l) * R * H;
            float* h_new = hs + ((size_t)(par ^ 1) * L + l) * R * H;
            for (int j = tid; j < H; j += NT) {
                float4 acc[R];
                if (l == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int row = min(row0 + r, B - 1);
                        acc[r] = *reinterpret_cast<const float4*>(p.GX + ((size_t)row * T + t) * G + 4 * j);
                    }
                } else {
                    const float4 bb = *reinterpret_cast<const float4*>(p.biasP[l] + 4 * j);
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = bb;
                    matvec<R>(acc, p.WihT[l] + 4 * j, G, hd + (size_t)(l - 1) * R * H, H, H);
                }
                matvec<R>(acc, p.WhhT[l] + 4 * j, G, h_old, H, H);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const size_t ci = ((size_t)l * R + r) * H + j;
                    const float ig = sigmoidf_(acc[r].x), fg = sigmoidf_(acc[r].y);
                    const float gg = tanhf_(acc[r].z), og = sigmoidf_(acc[r].w);
                    const float cn = fg * cs[ci] + ig * gg;
                    const float hn = og * tanhf_(cn);
                    cs[ci] = cn;
                    h_new[r * H + j] = hn;
                    const int row = row0 + r;
                    if (row < B) {
                        const size_t bt = (size_t)row * T + t;
                        *reinterpret_cast<float4*>(p.ACT[l] + bt * G + 4 * j) = make_float4(ig, fg, gg, og);
                        p.C[l][bt * H + j] = cn;
                        p.Hout[l][bt * H + j] = hn;
                        if (t == 0) p.Hprev[l][bt * H + j] = 0.f;
                        if (t + 1 < T) p.Hprev[l][(bt + 1) * H + j] = hn;
                        if (l + 1 < L)      // nn.LSTM inter-layer dropout (decoder.py:81), training only
                            hd[ci] = hn * keep_scale(p.seed, DS_LAYER0 + l, bt * H + j, p.p, inv_keep);
                    }
                }
            }
            __syncthreads();
        }
        par ^= 1;
    }
}

// Backward recurrence (BPTT).  Per step and layer (top down): gate gradients from (dh, dc), stored in
// the standard gate order n = g*H + j for the weight-gradient GEMMs, then dh_{t-1} = dG @ W_hh and
// (l > 0) the gradient into the layer below = dG @ W_ih.
struct BwdBuf {
    int B, T, H, L;
    const float* ACT[MAXL];
    const float* C[MAXL];
    const float* dHtop;          // [B*T][H] gradient w.r.t. the top layer output (after output-dropout backward)
    const float* Whh[MAXL];      // original layouts: (4H, H)
    const float* Wih[MAXL];      // l > 0: (4H, H)
    float* DG[MAXL];             // [B*T][4H] standard gate order
    float p;
    unsigned long long seed;
};

template <int R, int KR, int KL>
__global__ __launch_bounds__(NT) void lstm_train_bwd_kernel(BwdBuf p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, L = p.L, B = p.B, T = p.T;
    const size_t G = 4 * (size_t)H;
    const int CG = H / 4;                                   // column groups (float4 each)
    int NG = NT / CG;                                       // row groups of the 4H reduction
    if (NG > (int)(G / 32)) NG = (int)(G / 32);
    const int rows_per = (int)(G / NG);                     // multiple of 32
    float* dh_rec = smem;                    // [L][R][H]  dL/dh_{t} arriving from step t+1
    float* dc_next = dh_rec + L * R * H;     // [L][R][H]
    float* dh_low = dc_next + L * R * H;     // [R][H]     gradient handed to the layer below
    float* dgs = dh_low + R * H;             // [R][4H]    gate gradients of the current (layer, step)
    float* part = dgs + R * G;               // [NG][R][H] partial sums of the row-group split
    float* wl = part + (size_t)NG * R * H;   // [NG][KL][H] resident rows of W_hh[0] (fast path)
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * R;
    const float inv_keep = p.p > 0.f ? 1.f / (1.f - p.p) : 1.f;
    for (int idx = tid; idx < 2 * L * R * H; idx += NT) dh_rec[idx] = 0.f;   // dh_rec and dc_next are adjacent
    __syncthreads();
    const int cg = tid % CG, ng = tid / CG;
    // fast path (R == 1, L == 1, H == NT: CG = 64, NG = 4): thread (cg, ng) keeps the first KR rows of its row
    // group of W_hh in registers, the next KL rows of every group sit in LDS
    float4 wres[KR > 0 ? KR : 1];
    if (KR > 0) {
#pragma unroll
        for (int k = 0; k < KR; ++k)
            wres[k] = *reinterpret_cast<const float4*>(p.Whh[0] + ((size_t)ng * rows_per + k) * H + 4 * cg);
    }
    if (KL > 0) {
        for (int idx = tid; idx < NG * KL * (H / 4); idx += NT) {
            const int g2 = idx / (KL * (H / 4)), rem = idx - g2 * (KL * (H / 4));
            const int k = rem / (H / 4), c4 = rem - k * (H / 4);
            reinterpret_cast<float4*>(wl)[idx] =
                *reinterpret_cast<const float4*>(p.Whh[0] + ((size_t)g2 * rows_per + KR + k) * H + 4 * c4);
        }
        __syncthreads();
    }
    for (int t = T - 1; t >= 0; --t) {
        for (int l = L - 1; l >= 0; --l) {
            for (int j = tid; j < H; j += NT) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int row = min(row0 + r, B - 1);
                    const size_t bt = (size_t)row * T + t;
                    const float4 a = *reinterpret_cast<const float4*>(p.ACT[l] + bt * G + 4 * j);
                    const float c = p.C[l][bt * H + j];
                    const float cp = t > 0 ? p.C[l][(bt - 1) * H + j] : 0.f;
                    const size_t si = ((size_t)l * R + r) * H + j;
                    float dh = dh_rec[si];
                    if (l == L - 1) dh += p.dHtop[bt * H + j];
                    else dh += dh_low[r * H + j] * keep_scale(p.seed, DS_LAYER0 + l, bt * H + j, p.p, inv_keep);
                    const float tc = tanhf_(c);
                    const float d_o = dh * tc * a.w * (1.f - a.w);
                    const float dc = dh * a.w * (1.f - tc * tc) + dc_next[si];
                    const float d_i = dc * a.z * a.x * (1.f - a.x);
                    const float d_f = dc * cp * a.y * (1.f - a.y);
                    const float d_g = dc * a.x * (1.f - a.z * a.z);
                    dc_next[si] = dc * a.y;
                    float* dg = dgs + (size_t)r * G;
                    dg[j] = d_i; dg[H + j] = d_f; dg[2 * H + j] = d_g; dg[3 * H + j] = d_o;
                    if (row0 + r < B) {
                        float* o = p.DG[l] + bt * G;
                        o[j] = d_i; o[H + j] = d_f; o[2 * H + j] = d_g; o[3 * H + j] = d_o;
                    }
                }
            }
            __syncthreads();
            // dh_rec[l] = dG @ W_hh_l ; dh_low = dG @ W_ih_l  (row-group partials -> ordered sum)
            for (int which = 0; which < (l > 0 ? 2 : 1); ++which) {
                const float* W = which == 0 ? p.Whh[l] : p.Wih[l];
                if (ng < NG) {
                    float4 acc[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (KR + KL > 0)
                        matvec_res<KR, KL>(acc[0], wres, wl + ((size_t)ng * KL) * H + 4 * cg, H,
                                           W + (size_t)ng * rows_per * H + 4 * cg, (size_t)H, dgs + ng * rows_per, rows_per);
                    else
                        matvec<R>(acc, W + (size_t)ng * rows_per * H + 4 * cg, (size_t)H, dgs + ng * rows_per, (int)G, rows_per);
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        *reinterpret_cast<float4*>(part + ((size_t)ng * R + r) * H + 4 * cg) = acc[r];
                }
                __syncthreads();
                float* dst = which == 0 ? dh_rec + (size_t)l * R * H : dh_low;
                for (int idx = tid; idx < R * H; idx += NT) {
                    float s = 0.f;
                    for (int g2 = 0; g2 < NG; ++g2) s += part[(size_t)g2 * R * H + idx];
                    dst[idx] = s;
                }
                __syncthreads();
            }
        }
    }
}

// ------------------------------------------------------------------ cross entropy with label smoothing
// One workgroup per (b,t) row.  row_loss = keep * ((1-eps)*nll + eps*(-mean_v logp));
// dlogits = keep * (softmax - (1-eps)*onehot - eps/V)   [gradient of the SUM over rows; the mean's 1/count
// is applied later so that data-parallel ranks can all-reduce un-normalised sums].
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits,
                                                      const int32_t* __restrict__ targets, float* __restrict__ dlogits,
                                                      float* __restrict__ row_loss, float* __restrict__ row_keep, int V,
                                                      int pad_id, float eps) {
    __shared__ float red[8];
    const size_t row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + row * V;
    const int tgt = targets[row];
    float m = -INFINITY;
    for (int v = tid; v < V; v += 256) m = fmaxf(m, x[v]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f, sx = 0.f;
    for (int v = tid; v < V; v += 256) { s += expf(x[v] - m); sx += x[v]; }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); sx += __shfl_xor(sx, off, 64); }
    if (lane == 0) { red[wave] = s; red[4 + wave] = sx; }
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    sx = (red[4] + red[5]) + (red[6] + red[7]);
    const float lse = m + logf(s);
    const bool keep = tgt != pad_id;
    const int tg = min(max(tgt, 0), V - 1);
    if (dlogits) {
        float* d = dlogits + row * V;
        const float base = eps / (float)V;
        for (int v = tid; v < V; v += 256) {
            float g = expf(x[v] - lse) - base - (v == tg ? (1.f - eps) : 0.f);
            d[v] = keep ? g : 0.f;
        }
    }
    if (tid == 0) {
        const float nll = lse - x[tg];
        const float smooth = lse - sx / (float)V;
        row_loss[row] = keep ? (1.f - eps) * nll + eps * smooth : 0.f;
        row_keep[row] = keep ? 1.f : 0.f;
    }
}

// out[0] = sum(row_loss), out[1] = sum(row_keep): one workgroup, fixed order -> deterministic.
__global__ __launch_bounds__(256) void ordered_sum2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           size_t n, float* __restrict__ out) {
    __shared__ double ra[256], rb[256];
    double sa = 0.0, sb = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 256) { sa += a[i]; sb += b[i]; }
    ra[threadIdx.x] = sa; rb[threadIdx.x] = sb;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { ra[threadIdx.x] += ra[threadIdx.x + off]; rb[threadIdx.x] += rb[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)ra[0]; out[1] = (float)rb[0]; }
}

// out[n] = sum_m A[m][n]   (bias gradients): one thread per column chunk, rows in order (deterministic).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, size_t M, int N, int rows_per_block,
                                                     float* __restrict__ partial) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const size_t m0 = (size_t)blockIdx.y * rows_per_block;
    const size_t m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    if (n >= N) return;
    float s = 0.f;
    size_t m = m0;
    for (; m + 8 <= m1; m += 8) {                      // 8 loads in flight, added in row order
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = A[(m + j) * N + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += t[j];
    }
    for (; m < m1; ++m) s += A[m * N + n];
    partial[(size_t)blockIdx.y * N + n] = s;
}
// 32 columns x 8 row lanes per workgroup: lane r adds the partial rows r, r + 8, ... (four loads in flight), the eight
// sums meet in LDS and are added in lane order -- a fixed order, so the result is deterministic
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int N,
                                                           float* __restrict__ out, float* __restrict__ out2) {
    __shared__ float sm[8][32];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float s = 0.f;
    if (n < N) {
        int b = r;
        for (; b + 24 < nblk; b += 32) {
            const float v0 = partial[(size_t)b * N + n], v1 = partial[(size_t)(b + 8) * N + n];
            const float v2 = partial[(size_t)(b + 16) * N + n], v3 = partial[(size_t)(b + 24) * N + n];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < nblk; b += 8) s += partial[(size_t)b * N + n];
    }
    sm[r][c] = s;
    __syncthreads();
    if (r == 0 && n < N) {
        float t = sm[0][c];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += sm[i][c];
        out[n] = t;
        if (out2) out2[n] = t;
    }
}

// dEmb[tok[bt]] += dX[bt][:E] * mask ;  dEnc[b] = sum_t dX[b,t][E:] * mask
__global__ __launch_bounds__(256) void emb_scatter_kernel(const float* __restrict__ dX, const int32_t* __restrict__ tok,
                                                          float* __restrict__ dEmb, size_t BT, int E, int V, float p,
                                                          unsigned long long seed, int attn_path) {
    const size_t total = BT * E;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t bt = i / E;
        const int e = (int)(i - bt * E);
        const int tk = min(max(tok[bt], 0), V - 1);
        const float k = keep_scale(seed, DS_X, attn_path ? bt * E + e : bt * 2 * E + e, p, inv_keep);
        const float v = dX[bt * 2 * E + e] * k;
        if (v != 0.f) atomicAdd(dEmb + (size_t)tk * E + e, v);   // rows after a sequence's end carry exact zeros (and hit ONE row: PAD)
    }
}
// 32 (b, e) outputs x 8 time lanes per workgroup: lane r adds the steps r, r + 8, ..., the eight sums are added in
// lane order (fixed order: deterministic)
__global__ __launch_bounds__(256) void denc_reduce_kernel(const float* __restrict__ dX, float* __restrict__ dEnc, int B,
                                                          int T, int E, float p, unsigned long long seed,
                                                          int attn_path) {
    __shared__ float sm[8][32];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + c;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    float s = 0.f;
    if (idx < B * E) {
        const int b = idx / E, e = idx - b * E;
        for (int t = r; t < T; t += 8) {
            const size_t bt = (size_t)b * T + t;
            const size_t i = bt * 2 * E + E + e;
            s += dX[i] * (attn_path ? 1.f : keep_scale(seed, DS_X, i, p, inv_keep));
        }
    }
    sm[r][c] = s;
    __syncthreads();
    if (r == 0 && idx < B * E) {
        float t = sm[0][c];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += sm[i][c];
        dEnc[idx] = t;
    }
}

__global__ void fill_kernel(float* __restrict__ p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// out[k][4j+g] = W[(g*H+j)][k]  (same re-layout as decode.hip's prepare)
__global__ __launch_bounds__(256) void transpose_gate_kernel(const float* __restrict__ W, int H, int K,
                                                             float* __restrict__ out) {
    __shared__ float tile[64][65];
    const int G = 4 * H;
    const int c0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, k = k0 + tx;
        const int n = (c & 3) * H + (c >> 2);
        tile[i][tx] = (c < G && k < K) ? W[(size_t)n * K + k] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int k = k0 + i, c = c0 + tx;
        if (k < K && c < G) out[(size_t)k * G + c] = tile[tx][i];
    }
}
__global__ void bias_gate_kernel(const float* __restrict__ b_ih, const float* __restrict__ b_hh, float* __restrict__ out,
                                 int H) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < 4 * H) { const int n = (c & 3) * H + (c >> 2); out[c] = b_ih[n] + b_hh[n]; }
}

// ------------------------------------------------------------------ workspace layout
struct TLayout {
    size_t X, GX, WhhT[MAXL], WihT[MAXL], biasP[MAXL], ACT[MAXL], C[MAXL], Hout[MAXL], Hprev[MAXL], Hdrop, DG[MAXL],
        dHdrop, dX, row_loss, row_keep, colpart, colpart2, gemm_ws, gemm_ws2, xchg, total;
    size_t gemm_ws_bytes, xchg_bytes;
    int n_groups;
};

TLayout make_tlayout(int B, int T, int V, int E, int H, int L) {
    TLayout o{};
    size_t off = 0;
    auto take = [&](size_t floats) { size_t r = off; off += i2l_align(floats * sizeof(float)); return r; };
    const size_t BT = (size_t)B * T, G = 4 * (size_t)H;
    o.X = take(BT * 2 * E);
    o.GX = take(BT * G);
    for (int l = 0; l < L; ++l) o.WhhT[l] = take((size_t)H * G);
    for (int l = 1; l < L; ++l) { o.WihT[l] = take((size_t)H * G); o.biasP[l] = take(G); }
    for (int l = 0; l < L; ++l) {
        o.ACT[l] = take(BT * G); o.C[l] = take(BT * H); o.Hout[l] = take(BT * H); o.Hprev[l] = take(BT * H);
        o.DG[l] = take(BT * G);
    }
    o.Hdrop = take(BT * H);
    o.dHdrop = take(BT * H);
    o.dX = take(BT * 2 * E);
    o.row_loss = take(BT);
    o.row_keep = take(BT);
    o.colpart = take((size_t)1024 * (G > (size_t)V ? G : V));
    o.colpart2 = take((size_t)1024 * (G > (size_t)V ? G : V));      // the side lane's (i2l_lanes)
    size_t g = 0;
    auto mx = [&](size_t v) { if (v > g) g = v; };
    mx(i2l_gemm_workspace_bytes((int)BT, (int)G, 2 * E));     // GX
    mx(i2l_gemm_workspace_bytes((int)BT, V, H));              // logits
    mx(i2l_gemm_workspace_bytes(V, H, (int)BT));              // dW_out
    mx(i2l_gemm_workspace_bytes((int)BT, H, V));              // dHdrop
    mx(i2l_gemm_workspace_bytes((int)G, 2 * E, (int)BT));     // dW_ih0
    mx(i2l_gemm_workspace_bytes((int)G, H, (int)BT));         // dW_hh / dW_ih_l
    mx(i2l_gemm_workspace_bytes((int)BT, 2 * E, (int)G));     // dX
    mx(i2l_gemm_workspace_bytes((int)BT, E, (int)G));         // dX, one half (embedding / encoder columns)
    o.gemm_ws = off;
    o.gemm_ws_bytes = i2l_align(g);
    off += o.gemm_ws_bytes;
    o.gemm_ws2 = off;                                                // the side stream's
    off += o.gemm_ws_bytes;
    if (L == 1 && H == 256) {           // grouped recurrences (train_group.inc.h): status block + exchange granules
        // rows per group: 1 up to 64 rows (all 256 CUs on a 64-row shard), 2 up to 128 rows, 4 above
        o.n_groups = B <= 64 ? B : (B <= 128 ? i2l_cdiv(B, 2) : i2l_cdiv(B, 4));
        o.xchg = off;
        o.xchg_bytes = 2048 + (size_t)i2l_cdiv(o.n_groups, 8) * 8 * 2 * 4 *
                                  (B <= 64 ? TGB1_GRAN : (B <= 128 ? TGB2_GRAN : TGB_GRAN)) * sizeof(u64_t);
        off += i2l_align(o.xchg_bytes);
    }
    o.total = off;
    return o;
}

int check_w(const i2l_decoder_weights* w) {
    if (!w || !w->embedding || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh || !w->w_out || !w->b_out) return I2L_ERR_ARG;
    if (w->vocab <= 0 || w->embed <= 0 || w->hidden <= 0 || w->layers <= 0) return I2L_ERR_ARG;
    if (w->layers > MAXL || w->hidden % 64 != 0 || w->hidden > 1024 || w->embed % 4 != 0) return I2L_ERR_UNSUPPORTED;
    for (int l = 0; l < w->layers; ++l)
        if (!w->w_ih[l] || !w->w_hh[l] || !w->b_ih[l] || !w->b_hh[l]) return I2L_ERR_ARG;
    return I2L_OK;
}

int rows_per_wg(int B, size_t lds_per_row, size_t lds_fixed) {
    int R = B <= 256 ? 1 : (B <= 512 ? 2 : 4);
    while (R > 1 && lds_fixed + R * lds_per_row > 64 * 1024) R >>= 1;
    return R;
}

int colsum(const float* A, size_t M, int N, float* partial, float* out, float* out2, hipStream_t s) {
    int nblk = (int)((M + 31) / 32);                   // ~32 rows per block: enough blocks to fill the chip
    if (nblk > 1024) nblk = 1024;
    const int rpb = (int)((M + nblk - 1) / nblk);
    nblk = (int)((M + rpb - 1) / rpb);
    hipLaunchKernelGGL(colsum_kernel, dim3(i2l_cdiv(N, 256), nblk), dim3(256), 0, s, A, M, N, rpb, partial);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(i2l_cdiv(N, 32)), dim3(256), 0, s, partial, nblk, N, out, out2);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" size_t i2l_decoder_train_workspace_bytes(int B, int T, int vocab, int embed, int hidden, int layers) {
    if (B <= 0 || T <= 0 || vocab <= 0 || embed <= 0 || hidden <= 0 || layers <= 0 || layers > MAXL) return 0;
    return make_tlayout(B, T, vocab, embed, hidden, layers).total;
}

extern "C" int i2l_decoder_train_fwd(const i2l_decoder_weights* w, const float* enc, const int32_t* tokens, int B, int T,
                                      float dropout_p, uint64_t seed, int attention_path, void* workspace,
                                      size_t workspace_bytes, float* logits_out, int flags, i2l_stream_t stream) {
    int rc = check_w(w);
    if (rc != I2L_OK) return rc;
    const int split = (flags & I2L_FLAG_EXACT_FP32) ? 0 : 1;
    const bool group_on = !(flags & I2L_FLAG_NO_GROUP);
    if (!enc || !tokens || !logits_out || B <= 0 || T <= 0 || dropout_p < 0.f || dropout_p >= 1.f) return I2L_ERR_ARG;
    const int V = w->vocab, E = w->embed, H = w->hidden, L = w->layers;
    const TLayout lo = make_tlayout(B, T, V, E, H, L);
    if (!workspace || workspace_bytes < lo.total) return I2L_ERR_WORKSPACE;
    char* base = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    hipStream_t s = i2l_s(stream);
    const size_t BT = (size_t)B * T;
    const int G = 4 * H;

    hipLaunchKernelGGL(build_x_kernel, dim3(grid_for(BT * 2 * E)), dim3(256), 0, s, w->embedding, enc, tokens, F(lo.X),
                       B, T, E, V, dropout_p, (unsigned long long)seed, attention_path);
    I2L_CHECK_LAUNCH();
    {   // GX = X @ W_ih_0^T + b_ih_0 + b_hh_0, gate-interleaved
        GemmArgs g = gemm_args();
        g.split_bf16 = split;   // training GEMMs: bf16 matrix cores with 3-way split operands (fp32-grade), gemm.hip
        g.A = F(lo.X); g.lda = 2 * E;
        g.W = w->w_ih[0]; g.ldw = 2 * E;
        g.bias = w->b_ih[0]; g.bias2 = w->b_hh[0];
        g.C = F(lo.GX); g.ldc = G;
        g.M = (int)BT; g.N = G; g.K = 2 * E; g.perm_h = H;
        rc = i2l_gemm(g, base + lo.gemm_ws, lo.gemm_ws_bytes, s);
        if (rc != I2L_OK) return rc;
    }
    for (int l = 0; l < L; ++l) {
        dim3 grid(i2l_cdiv(G, 64), i2l_cdiv(H, 64));
        hipLaunchKernelGGL(transpose_gate_kernel, grid, dim3(256), 0, s, w->w_hh[l], H, H, F(lo.WhhT[l]));
        I2L_CHECK_LAUNCH();
        if (l > 0) {
            hipLaunchKernelGGL(transpose_gate_kernel, grid, dim3(256), 0, s, w->w_ih[l], H, H, F(lo.WihT[l]));
            I2L_CHECK_LAUNCH();
            hipLaunchKernelGGL(bias_gate_kernel, dim3(i2l_cdiv(G, 256)), dim3(256), 0, s, w->b_ih[l], w->b_hh[l],
                               F(lo.biasP[l]), H);
            I2L_CHECK_LAUNCH();
        }
    }
    TrainBuf p{};
    p.B = B; p.T = T; p.H = H; p.L = L; p.GX = F(lo.GX); p.p = L > 1 ? dropout_p : 0.f; p.seed = seed;
    for (int l = 0; l < L; ++l) {
        p.WhhT[l] = F(lo.WhhT[l]); p.WihT[l] = l ? F(lo.WihT[l]) : nullptr; p.biasP[l] = l ? F(lo.biasP[l]) : nullptr;
        p.ACT[l] = F(lo.ACT[l]); p.C[l] = F(lo.C[l]); p.Hout[l] = F(lo.Hout[l]); p.Hprev[l] = F(lo.Hprev[l]);
    }
    {
        const size_t per_row = (size_t)4 * L * H * sizeof(float);
        const int R = rows_per_wg(B, per_row, 0);
        const size_t lds = R * per_row;
        if (lds > 64 * 1024) return I2L_ERR_UNSUPPORTED;
        dim3 grid(i2l_cdiv(B, R));
        bool done = false;
        if (lo.xchg_bytes && T >= 8 && group_on) {  // 4 workgroups share 4 rows, W_hh in registers
            TrainGroupFwd gp{};
            gp.B = B; gp.T = T; gp.n_groups = lo.n_groups; gp.GX = p.GX; gp.WhhT = p.WhhT[0];
            gp.ACT = p.ACT[0]; gp.C = p.C[0]; gp.Hout = p.Hout[0]; gp.Hprev = p.Hprev[0];
            gp.status = reinterpret_cast<unsigned*>(base + lo.xchg);
            gp.xchg = reinterpret_cast<u64_t*>(base + lo.xchg + 2048);
            gp.opts = group_opts(T, flags);
            const size_t used = 2048 + (size_t)i2l_cdiv(lo.n_groups, 8) * 8 * 2 * 4 *
                                       (B <= 64 ? TGF1_GRAN : (B <= 128 ? TGF2_GRAN : TGF_GRAN)) * sizeof(u64_t);
            if (hipMemsetAsync(base + lo.xchg, 0, used, s) != hipSuccess) return I2L_ERR_LAUNCH;
            if (B <= 64)
                hipLaunchKernelGGL(lstm_train_fwd_group1_kernel, dim3(i2l_cdiv(lo.n_groups, 8) * 32), dim3(TGT), 0, s, gp);
            else if (B <= 128)
                hipLaunchKernelGGL(lstm_train_fwd_group2_kernel, dim3(i2l_cdiv(lo.n_groups, 8) * 32), dim3(TGT), 0, s, gp);
            else
                hipLaunchKernelGGL(lstm_train_fwd_group_kernel, dim3(i2l_cdiv(lo.n_groups, 8) * 32), dim3(TGT), 0, s, gp);
            done = true;
        }
        if (!done && R == 1 && L == 1 && H == NT && T >= 8) {        // part of W_hh resident on chip
            const size_t lds_r = lds + (size_t)RES_KL * G * sizeof(float);
            auto kern = lstm_train_fwd_kernel<1, RES_KR, RES_KL>;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_r) == hipSuccess) {
                hipLaunchKernelGGL(kern, grid, dim3(NT), lds_r, s, p);
                done = true;
            }
        }
        if (!done) {
            if (R == 1) hipLaunchKernelGGL((lstm_train_fwd_kernel<1, 0, 0>), grid, dim3(NT), lds, s, p);
            else if (R == 2) hipLaunchKernelGGL((lstm_train_fwd_kernel<2, 0, 0>), grid, dim3(NT), lds, s, p);
            else hipLaunchKernelGGL((lstm_train_fwd_kernel<4, 0, 0>), grid, dim3(NT), lds, s, p);
        }
        I2L_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(dropout_rows_kernel, dim3(grid_for(BT * H)), dim3(256), 0, s, (const float*)F(lo.Hout[L - 1]),
                       F(lo.Hdrop), BT * H, dropout_p, (unsigned long long)seed, (unsigned)DS_OUT);
    I2L_CHECK_LAUNCH();
    {   // logits = Hdrop @ W_out^T + b_out
        GemmArgs g = gemm_args();
        g.split_bf16 = split;   // training GEMMs: bf16 matrix cores with 3-way split operands (fp32-grade), gemm.hip
        g.A = F(lo.Hdrop); g.lda = H;
        g.W = w->w_out; g.ldw = H;
        g.bias = w->b_out;
        g.C = logits_out; g.ldc = V;
        g.M = (int)BT; g.N = V; g.K = H;
        rc = i2l_gemm(g, base + lo.gemm_ws, lo.gemm_ws_bytes, s);
        if (rc != I2L_OK) return rc;
    }
    return I2L_OK;
}

extern "C" int i2l_ce_label_smooth_fwd_bwd(const float* logits, const int32_t* targets, int rows, int vocab, int pad_id,
                                            float label_smoothing, void* workspace, size_t workspace_bytes,
                                            float* dlogits_out, float* loss_sum_and_count_out, i2l_stream_t stream) {
    if (!logits || !targets || !loss_sum_and_count_out || rows <= 0 || vocab <= 0) return I2L_ERR_ARG;
    const size_t need = 2 * i2l_align((size_t)rows * sizeof(float));
    if (!workspace || workspace_bytes < need) return I2L_ERR_WORKSPACE;
    float* row_loss = static_cast<float*>(workspace);
    float* row_keep = reinterpret_cast<float*>(static_cast<char*>(workspace) + i2l_align((size_t)rows * sizeof(float)));
    hipStream_t s = i2l_s(stream);
    hipLaunchKernelGGL(ce_rows_kernel, dim3(rows), dim3(256), 0, s, logits, targets, dlogits_out, row_loss, row_keep,
                       vocab, pad_id, label_smoothing);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(ordered_sum2_kernel, dim3(1), dim3(256), 0, s, (const float*)row_loss, (const float*)row_keep,
                       (size_t)rows, loss_sum_and_count_out);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" size_t i2l_ce_workspace_bytes(int rows) {
    return rows > 0 ? 2 * i2l_align((size_t)rows * sizeof(float)) : 0;
}

extern "C" int i2l_decoder_train_bwd(const i2l_decoder_weights* w, const int32_t* tokens, int B, int T, float dropout_p,
                                      uint64_t seed, int attention_path, void* workspace, size_t workspace_bytes,
                                      const float* dlogits, const i2l_decoder_grads* gr, float* denc_out,
                                      int flags, i2l_lanes* lanes, i2l_stream_t stream) {
    int rc = check_w(w);
    if (rc != I2L_OK) return rc;
    const int split = (flags & I2L_FLAG_EXACT_FP32) ? 0 : 1;
    const bool group_on = !(flags & I2L_FLAG_NO_GROUP);
    if (!tokens || !dlogits || !gr || !denc_out || B <= 0 || T <= 0) return I2L_ERR_ARG;
    if (!gr->embedding || !gr->w_ih || !gr->w_hh || !gr->b_ih || !gr->b_hh || !gr->w_out || !gr->b_out) return I2L_ERR_ARG;
    const int V = w->vocab, E = w->embed, H = w->hidden, L = w->layers;
    const TLayout lo = make_tlayout(B, T, V, E, H, L);
    if (!workspace || workspace_bytes < lo.total) return I2L_ERR_WORKSPACE;
    char* base = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    hipStream_t s = i2l_s(stream);
    const size_t BT = (size_t)B * T;
    const int G = 4 * H;
    void* gws = base + lo.gemm_ws;
    // lanes != NULL (single layer): the weight gradients and bias sums run on the caller's side lane 0 with their
    // own GEMM workspace and column-sum scratch -- dW_out / db_out from here on, dW_ih / dW_hh / db once the recurrence's
    // backward (on the caller's stream) has produced the gate gradients.  The caller joins (i2l_lanes_join).
    hipStream_t s_w = s;
    void* gws_w = gws;
    float* colpart_w = F(lo.colpart);
    const bool side = lanes != nullptr && L == 1;
    if (side) {
        hipStream_t f = i2l_side_fork(lanes, s, 0);
        if (f) { s_w = f; gws_w = base + lo.gemm_ws2; colpart_w = F(lo.colpart2); }
    }

    {   // dW_out[v][h] = sum_bt dlogits[bt][v] * Hdrop[bt][h]
        GemmArgs g = gemm_args();
        g.split_bf16 = split;   // training GEMMs: bf16 matrix cores with 3-way split operands (fp32-grade), gemm.hip
        g.A = dlogits; g.lda = V; g.a_kc = 0;
        g.W = F(lo.Hdrop); g.ldw = H; g.w_kc = 0;
        g.C = gr->w_out; g.ldc = H;
        g.M = V; g.N = H; g.K = (int)BT;
        rc = i2l_gemm(g, gws_w, lo.gemm_ws_bytes, s_w);
        if (rc != I2L_OK) return rc;
    }
    rc = colsum(dlogits, BT, V, colpart_w, gr->b_out, nullptr, s_w);
    if (rc != I2L_OK) return rc;
    {   // dHdrop[bt][h] = sum_v dlogits[bt][v] * W_out[v][h]
        GemmArgs g = gemm_args();
        g.split_bf16 = split;   // training GEMMs: bf16 matrix cores with 3-way split operands (fp32-grade), gemm.hip
        g.A = dlogits; g.lda = V;
        g.W = w->w_out; g.ldw = H; g.w_kc = 0;
        g.C = F(lo.dHdrop); g.ldc = H;
        g.M = (int)BT; g.N = H; g.K = V;
        rc = i2l_gemm(g, gws, lo.gemm_ws_bytes, s);
        if (rc != I2L_OK) return rc;
    }
    // output-dropout backward (same mask), in place
    hipLaunchKernelGGL(dropout_rows_kernel, dim3(grid_for(BT * H)), dim3(256), 0, s, (const float*)F(lo.dHdrop),
                       F(lo.dHdrop), BT * H, dropout_p, (unsigned long long)seed, (unsigned)DS_OUT);
    I2L_CHECK_LAUNCH();

    BwdBuf p{};
    p.B = B; p.T = T; p.H = H; p.L = L; p.dHtop = F(lo.dHdrop); p.p = L > 1 ? dropout_p : 0.f; p.seed = seed;
    for (int l = 0; l < L; ++l) {
        p.ACT[l] = F(lo.ACT[l]); p.C[l] = F(lo.C[l]); p.DG[l] = F(lo.DG[l]);
        p.Whh[l] = w->w_hh[l]; p.Wih[l] = l ? w->w_ih[l] : nullptr;
    }
    {
        const int CG = H / 4;
        int NG = NT / CG;
        if (NG > G / 32) NG = G / 32;
        const size_t per_row = ((size_t)2 * L * H + H + G + (size_t)NG * H) * sizeof(float);
        const int R = rows_per_wg(B, per_row, 0);
        const size_t lds = R * per_row;
        if (lds > 64 * 1024) return I2L_ERR_UNSUPPORTED;
        dim3 grid(i2l_cdiv(B, R));
        bool done = false;
        if (lo.xchg_bytes && T >= 8 && group_on) {
            TrainGroupBwd gp{};
            gp.B = B; gp.T = T; gp.n_groups = lo.n_groups; gp.ACT = p.ACT[0]; gp.C = p.C[0]; gp.dHtop = p.dHtop;
            gp.Whh = p.Whh[0]; gp.DG = p.DG[0];
            gp.status = reinterpret_cast<unsigned*>(base + lo.xchg);
            gp.xchg = reinterpret_cast<u64_t*>(base + lo.xchg + 2048);
            gp.opts = group_opts(T, flags);
            if (hipMemsetAsync(base + lo.xchg, 0, lo.xchg_bytes, s) != hipSuccess) return I2L_ERR_LAUNCH;
            if (B <= 64)
                hipLaunchKernelGGL(lstm_train_bwd_group1_kernel, dim3(i2l_cdiv(lo.n_groups, 8) * 32), dim3(TGT), 0, s, gp);
            else if (B <= 128)
                hipLaunchKernelGGL(lstm_train_bwd_group2_kernel, dim3(i2l_cdiv(lo.n_groups, 8) * 32), dim3(TGT), 0, s, gp);
            else
                hipLaunchKernelGGL(lstm_train_bwd_group_kernel, dim3(i2l_cdiv(lo.n_groups, 8) * 32), dim3(TGT), 0, s, gp);
            done = true;
        }
        if (!done && R == 1 && L == 1 && H == NT && T >= 8) {
            const size_t lds_r = lds + (size_t)NG * RES_KL * H * sizeof(float);
            auto kern = lstm_train_bwd_kernel<1, RES_KR, RES_KL>;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_r) == hipSuccess) {
                hipLaunchKernelGGL(kern, grid, dim3(NT), lds_r, s, p);
                done = true;
            }
        }
        if (!done) {
            if (R == 1) hipLaunchKernelGGL((lstm_train_bwd_kernel<1, 0, 0>), grid, dim3(NT), lds, s, p);
            else if (R == 2) hipLaunchKernelGGL((lstm_train_bwd_kernel<2, 0, 0>), grid, dim3(NT), lds, s, p);
            else hipLaunchKernelGGL((lstm_train_bwd_kernel<4, 0, 0>), grid, dim3(NT), lds, s, p);
        }
        I2L_CHECK_LAUNCH();
    }
    // weight gradients: one GEMM each over all B*T rows
    if (side && s_w != s) {                                  // the gate gradients are complete on the caller's stream here
        hipStream_t f = i2l_side_fork(lanes, s, 0);
        if (!f) return I2L_ERR_LAUNCH;
        s_w = f;
    }
    for (int l = 0; l < L; ++l) {
        const float* DGl = F(lo.DG[l]);
        {   // dW_ih_l[n][k] = sum_bt DG[bt][n] * In[bt][k]
            GemmArgs g = gemm_args();
        g.split_bf16 = split;   // training GEMMs: bf16 matrix cores with 3-way split operands (fp32-grade), gemm.hip
            g.A = DGl; g.lda = G; g.a_kc = 0;
            if (l == 0) { g.W = F(lo.X); g.ldw = 2 * E; g.N = 2 * E; }
            else { g.W = F(lo.Hdrop); g.ldw = H; g.N = H; }     // placeholder, replaced below for l > 0
            g.w_kc = 0;
            g.C = gr->w_ih[l]; g.ldc = g.N;
            g.M = G; g.K = (int)BT;
            if (l > 0) {
                // input of layer l = inter-layer-dropped output of layer l-1: rebuild it into dX scratch (BT x H)
                hipLaunchKernelGGL(dropout_rows_kernel, dim3(grid_for(BT * H)), dim3(256), 0, s,
                                   (const float*)F(lo.Hout[l - 1]), F(lo.dX), BT * H, p.p, (unsigned long long)seed,
                                   (unsigned)(DS_LAYER0 + l - 1));
                I2L_CHECK_LAUNCH();
                g.W = F(lo.dX);
            }
            rc = i2l_gemm(g, gws_w, lo.gemm_ws_bytes, s_w);
            if (rc != I2L_OK) return rc;
        }
        {   // dW_hh_l[n][k] = sum_bt DG[bt][n] * h_{t-1}[bt][k]
            GemmArgs g = gemm_args();
        g.split_bf16 = split;   // training GEMMs: bf16 matrix cores with 3-way split operands (fp32-grade), gemm.hip
            g.A = DGl; g.lda = G; g.a_kc = 0;
            g.W = F(lo.Hprev[l]); g.ldw = H; g.w_kc = 0;
            g.C = gr->w_hh[l]; g.ldc = H;
            g.M = G; g.N = H; g.K = (int)BT;
            rc = i2l_gemm(g, gws_w, lo.gemm_ws_bytes, s_w);
            if (rc != I2L_OK) return rc;
        }
        rc = colsum(DGl, BT, G, colpart_w, gr->b_ih[l], gr->b_hh[l], s_w);
        if (rc != I2L_OK) return rc;
    }
    // dX[bt][k] = sum_n DG0[bt][n] * W_ih_0[n][k].  Its ENCODER half (k >= E) feeds dEnc, the head of the encoder's backward
    // pass; its EMBEDDING half only feeds dEmb, which nothing waits for before the optimizer.  With a side lane the halves are
    // two GEMMs: the encoder half here, the embedding half + scatter on the lane behind the weight gradients (r04: the dX
    // GEMM was 95 us on the step's critical path, the half is 47).  Same products and sums either way.
    // (Always two GEMMs, so that the numbers do not depend on whether a lane is there.)
    hipStream_t s_e = (side && s_w != s) ? s_w : s;
    for (int half = 0; half < 2; ++half) {
        GemmArgs g = gemm_args();
        g.split_bf16 = split;   // training GEMMs: bf16 matrix cores with 3-way split operands (fp32-grade), gemm.hip
        g.A = F(lo.DG[0]); g.lda = G;
        g.ldw = 2 * E; g.w_kc = 0;
        g.ldc = 2 * E;
        g.M = (int)BT; g.N = E; g.K = G;
        if (half == 0) {
            g.W = w->w_ih[0] + E; g.C = F(lo.dX) + E;
            rc = i2l_gemm(g, gws, lo.gemm_ws_bytes, s);
        } else {
            g.W = w->w_ih[0]; g.C = F(lo.dX);
            rc = i2l_gemm(g, s_e == s ? gws : gws_w, lo.gemm_ws_bytes, s_e);
        }
        if (rc != I2L_OK) return rc;
    }
    hipLaunchKernelGGL(denc_reduce_kernel, dim3(i2l_cdiv(B * E, 32)), dim3(256), 0, s, (const float*)F(lo.dX),
                       denc_out, B, T, E, dropout_p, (unsigned long long)seed, attention_path);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for((size_t)V * E)), dim3(256), 0, s_e, gr->embedding, (size_t)V * E, 0.f);
    I2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(emb_scatter_kernel, dim3(grid_for(BT * E)), dim3(256), 0, s_e, (const float*)F(lo.dX), tokens,
                       gr->embedding, BT, E, V, dropout_p, (unsigned long long)seed, attention_path);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
