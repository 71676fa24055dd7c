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

#include "common.h"

namespace {

constexpr int MAXL = I2L_MAX_LSTM_LAYERS;
constexpr int NT = 256;        // threads per workgroup
constexpr int VCHUNK = 2 * NT;  // vocab columns per projection pass

struct Layout {
    size_t P, Genc, WhhT[MAXL], WihT[MAXL], biasP[MAXL], WoutT, boutP, gemm_ws, total;
    int Vp;
};

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
    size_t g1 = i2l_gemm_workspace_bytes(V, 4 * H, E), g2 = i2l_gemm_workspace_bytes(rows, 4 * H, E);
    o.gemm_ws = off;
    off += i2l_align(g1 > g2 ? g1 : g2);
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

struct DecodeParams {
    int B, H, L, V, Vp, T;
    const float* P;
    const float* Genc;
    const float* WhhT[MAXL];
    const float* WihT[MAXL];
    const float* biasP[MAXL];
    const float* WoutT;
    const float* boutP;
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
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// acc[r] += sum_k W[k][0..3] * x[r][k], k ascending (one fmaf chain per output).
template <int R>
__device__ __forceinline__ void matvec4(float4 (&acc)[R], const float* __restrict__ Wcol, size_t ldw,
                                        const float* xs, int H) {
    for (int k = 0; k < H; k += 8) {
        float4 w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = *reinterpret_cast<const float4*>(Wcol + (size_t)(k + i) * ldw);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 xa = *reinterpret_cast<const float4*>(xs + r * H + k);
            const float4 xb = *reinterpret_cast<const float4*>(xs + r * H + k + 4);
            const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[r].x = fmaf(w[i].x, xv[i], acc[r].x);
                acc[r].y = fmaf(w[i].y, xv[i], acc[r].y);
                acc[r].z = fmaf(w[i].z, xv[i], acc[r].z);
                acc[r].w = fmaf(w[i].w, xv[i], acc[r].w);
            }
        }
    }
}

template <int R>
__device__ __forceinline__ void matvec2(float2 (&acc)[R], const float* __restrict__ Wcol, size_t ldw,
                                        const float* xs, int H) {
    for (int k = 0; k < H; k += 8) {
        float2 w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = *reinterpret_cast<const float2*>(Wcol + (size_t)(k + i) * ldw);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 xa = *reinterpret_cast<const float4*>(xs + r * H + k);
            const float4 xb = *reinterpret_cast<const float4*>(xs + r * H + k + 4);
            const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[r].x = fmaf(w[i].x, xv[i], acc[r].x);
                acc[r].y = fmaf(w[i].y, xv[i], acc[r].y);
            }
        }
    }
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

template <int R>
__global__ __launch_bounds__(NT) void decode_kernel(DecodeParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, L = p.L, B = p.B, T = p.T, V = p.V;
    const size_t G = 4 * (size_t)H;
    float* hs = smem;                          // [2][L][R][H]  double-buffered h
    float* cs = hs + 2 * L * R * H;            // [L][R][H]
    float* redv = cs + L * R * H;              // [4][R]
    int* redi = reinterpret_cast<int*>(redv + 4 * R);   // [4][R]
    int* tok_s = redi + 4 * R;                 // [R]
    int* fin_s = tok_s + R;                    // [R]
    float* lg = reinterpret_cast<float*>(fin_s + R);    // [R][Vp], only with I2L_SELECT_SOFTMAX

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * R;

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
    __syncthreads();

    int par = 0;
    int t = 0;
    for (; t < T; ++t) {
        int tok[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = min(row0 + r, B - 1);
            int tk = p.forced ? p.forced[(size_t)row * T + t] : tok_s[r];
            tok[r] = min(max(tk, 0), V - 1);
        }
        // ---------------- LSTM layers
        for (int l = 0; l < L; ++l) {
            const float* h_old = hs + ((size_t)par * L + l) * R * H;
            float* h_new = hs + ((size_t)(par ^ 1) * L + l) * R * H;
            for (int j = tid; j < H; j += NT) {
                float4 acc[R];
                if (l == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int row = min(row0 + r, B - 1);
                        const float4 a = *reinterpret_cast<const float4*>(p.P + (size_t)tok[r] * G + 4 * j);
                        const float4 e = *reinterpret_cast<const float4*>(p.Genc + (size_t)row * G + 4 * j);
                        acc[r] = make_float4(a.x + e.x, a.y + e.y, a.z + e.z, a.w + e.w);
                    }
                } else {
                    const float4 bb = *reinterpret_cast<const float4*>(p.biasP[l] + 4 * j);
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = bb;
                    matvec4<R>(acc, p.WihT[l] + 4 * j, G, hs + ((size_t)(par ^ 1) * L + (l - 1)) * R * H, H);
                }
                matvec4<R>(acc, p.WhhT[l] + 4 * j, G, h_old, H);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float* cp = cs + ((size_t)l * R + r) * H + j;
                    const float ig = sigmoidf_(acc[r].x), fg = sigmoidf_(acc[r].y);
                    const float gg = tanhf(acc[r].z), og = sigmoidf_(acc[r].w);
                    const float cn = fg * (*cp) + ig * gg;
                    *cp = cn;
                    h_new[r * H + j] = og * tanhf(cn);
                }
            }
            __syncthreads();
        }
        // ---------------- output projection + token selection
        const float* h_top = hs + ((size_t)(par ^ 1) * L + (L - 1)) * R * H;
        float best[R];
        int besti[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { best[r] = -INFINITY; besti[r] = 0x7fffffff; }
        for (int v0 = 2 * tid; v0 < p.Vp; v0 += VCHUNK) {
            float2 acc[R];
            const float2 bb = *reinterpret_cast<const float2*>(p.boutP + v0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = bb;
            matvec2<R>(acc, p.WoutT + v0, (size_t)p.Vp, h_top, H);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = row0 + r;
                if (p.logits && row < B) {
                    float* lp = p.logits + ((size_t)row * T + t) * V;
                    if (v0 < V) lp[v0] = acc[r].x;
                    if (v0 + 1 < V) lp[v0 + 1] = acc[r].y;
                }
                float a = acc[r].x, b2 = acc[r].y;
                if (p.use_temp) { a = a / p.temperature; b2 = b2 / p.temperature; }
                if (p.select == I2L_SELECT_SOFTMAX) { lg[r * p.Vp + v0] = a; lg[r * p.Vp + v0 + 1] = b2; }
                if (a > best[r]) { best[r] = a; besti[r] = v0; }
                if (b2 > best[r]) { best[r] = b2; besti[r] = v0 + 1; }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            wave_argmax(best[r], besti[r]);
            if (lane == 0) { redv[wave * R + r] = best[r]; redi[wave * R + r] = besti[r]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {           // every thread folds the 4 wave results (same answer everywhere)
            float bv = redv[r];
            int bi = redi[r];
#pragma unroll
            for (int w = 1; w < NT / 64; ++w) {
                const float ov = redv[w * R + r];
                const int oi = redi[w * R + r];
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            best[r] = bv; besti[r] = bi;
        }
        if (p.select == I2L_SELECT_SOFTMAX) {
            // argmax(softmax(x)): probabilities in fp32 as torch.softmax (exp(x - max) / sum), first index wins
            __syncthreads();                     // redv/redi are reused below
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float s = 0.f;
                for (int v = tid; v < V; v += NT) s += expf(lg[r * p.Vp + v] - best[r]);
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
                for (int w = 0; w < NT / 64; ++w) s += redv[w * R + r];
                pbest[r] = -1.f; pbesti[r] = 0x7fffffff;
                for (int v = tid; v < V; v += NT) {
                    const float pr = expf(lg[r * p.Vp + v] - best[r]) / s;
                    if (pr > pbest[r]) { pbest[r] = pr; pbesti[r] = v; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                wave_argmax(pbest[r], pbesti[r]);
                if (lane == 0) { redv[wave * R + r] = pbest[r]; redi[wave * R + r] = pbesti[r]; }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float bv = redv[r];
                int bi = redi[r];
#pragma unroll
                for (int w = 1; w < NT / 64; ++w) {
                    const float ov = redv[w * R + r];
                    const int oi = redi[w * R + r];
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                besti[r] = bi;
            }
        }
        __syncthreads();                          // all reads of redv/redi/tok_s/fin_s done
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

int check_weights(const i2l_decoder_weights* w) {
    if (!w || !w->embedding || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh || !w->w_out || !w->b_out)
        return I2L_ERR_ARG;
    if (w->vocab <= 0 || w->embed <= 0 || w->hidden <= 0 || w->layers <= 0) return I2L_ERR_ARG;
    if (w->layers > MAXL || w->hidden % 64 != 0 || w->hidden > 2048 || w->embed % 4 != 0) return I2L_ERR_UNSUPPORTED;
    for (int l = 0; l < w->layers; ++l)
        if (!w->w_ih[l] || !w->w_hh[l] || !w->b_ih[l] || !w->b_hh[l]) return I2L_ERR_ARG;
    return I2L_OK;
}

size_t decode_lds_bytes(int R, int L, int H, int Vp, int select) {
    size_t floats = (size_t)3 * L * R * H + 4 * R + 4 * R + R + R;
    if (select == I2L_SELECT_SOFTMAX) floats += (size_t)R * Vp;
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
    }
    const size_t gws = lo.total - lo.gemm_ws;
    if (what & I2L_PREP_WEIGHTS) {   // P = Emb @ W_ih_0[:, :E]^T, gate-interleaved columns
        GemmArgs g{};
        g.A = w->embedding; g.lda = E;
        g.W = w->w_ih[0]; g.ldw = 2 * E;
        g.C = F(lo.P); g.ldc = G;
        g.M = V; g.N = G; g.K = E; g.perm_h = H;
        rc = i2l_gemm_nt(g, base + lo.gemm_ws, gws, s);
        if (rc != I2L_OK) return rc;
    }
    if (what & I2L_PREP_ROWS) {   // Genc = enc @ W_ih_0[:, E:]^T + b_ih_0 + b_hh_0
        GemmArgs g{};
        g.A = enc; g.lda = E;
        g.W = w->w_ih[0] + E; g.ldw = 2 * E;
        g.bias = w->b_ih[0]; g.bias2 = w->b_hh[0];
        g.C = F(lo.Genc); g.ldc = G;
        g.M = rows; g.N = G; g.K = E; g.perm_h = H;
        rc = i2l_gemm_nt(g, base + lo.gemm_ws, gws, s);
        if (rc != I2L_OK) return rc;
    }
    return I2L_OK;
}

extern "C" int i2l_greedy_decode(const i2l_decoder_weights* w, const void* workspace, int rows, int steps,
                                 const int32_t* tok0, const int32_t* forced, const float* h0, const float* c0,
                                 float temperature, int select, int stop, int end_id, int32_t* ids_out,
                                 float* logits_out, float* h_out, float* c_out, i2l_stream_t stream) {
    int rc = check_weights(w);
    if (rc != I2L_OK) return rc;
    if (!workspace || !tok0 || rows <= 0 || steps <= 0) return I2L_ERR_ARG;
    if ((h0 == nullptr) != (c0 == nullptr)) return I2L_ERR_ARG;
    if (select != I2L_SELECT_LOGITS && select != I2L_SELECT_SOFTMAX) return I2L_ERR_ARG;
    if (stop != I2L_STOP_NONE && stop != I2L_STOP_STICKY) return I2L_ERR_ARG;
    const int V = w->vocab, E = w->embed, H = w->hidden, L = w->layers;
    const Layout lo = make_layout(rows, V, E, H, L);
    const char* base = static_cast<const char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<const float*>(base + off); };

    DecodeParams p{};
    p.B = rows; p.H = H; p.L = L; p.V = V; p.Vp = lo.Vp; p.T = steps;
    p.P = F(lo.P); p.Genc = F(lo.Genc);
    for (int l = 0; l < L; ++l) {
        p.WhhT[l] = F(lo.WhhT[l]);
        p.WihT[l] = l > 0 ? F(lo.WihT[l]) : nullptr;
        p.biasP[l] = l > 0 ? F(lo.biasP[l]) : nullptr;
    }
    p.WoutT = F(lo.WoutT); p.boutP = F(lo.boutP);
    p.tok0 = tok0; p.forced = forced; p.h0 = h0; p.c0 = c0;
    p.h_out = h_out; p.c_out = c_out; p.ids = ids_out; p.logits = logits_out;
    p.temperature = temperature; p.use_temp = (temperature != 1.0f) ? 1 : 0;
    p.select = select; p.stop = stop; p.end_id = end_id;

    // rows per workgroup: fill the 256 CUs first, then stack rows (weights are streamed once per workgroup per step)
    int R = rows <= 256 ? 1 : (rows <= 512 ? 2 : 4);
    while (R > 1 && decode_lds_bytes(R, L, H, lo.Vp, select) > 64 * 1024) R >>= 1;
    const size_t lds = decode_lds_bytes(R, L, H, lo.Vp, select);
    if (lds > 64 * 1024) return I2L_ERR_UNSUPPORTED;
    const dim3 grid(i2l_cdiv(rows, R));
    hipStream_t s = i2l_s(stream);
    if (R == 1) hipLaunchKernelGGL(decode_kernel<1>, grid, dim3(NT), lds, s, p);
    else if (R == 2) hipLaunchKernelGGL(decode_kernel<2>, grid, dim3(NT), lds, s, p);
    else hipLaunchKernelGGL(decode_kernel<4>, grid, dim3(NT), lds, s, p);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
