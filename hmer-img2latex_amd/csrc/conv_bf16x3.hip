// Fused Conv2d(3x3, pad 1) + bias + ReLU + MaxPool2d(2) on the bf16 matrix cores with fp32-grade accuracy.
//
// Every fp32 operand is split into three bf16 pieces, x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1): 3 x 8 = 24 mantissa bits, the splits are exact in fp32), and a product a*b is
// evaluated as the six partial products a_i * b_j with i + j <= 2 on v_mfma_f32_32x32x16_bf16.  A bf16 x bf16
// product is exact in fp32 and the matrix core accumulates in fp32, so the result carries the same ~2^-24
// relative error class as an fp32 fmaf chain (the dropped terms are <= 2^-24 |a||b|), while the bf16 MFMA
// runs at 16x the fp32 MFMA rate: 6 MFMAs replace 8 for every 16 k, at 1/2 the cycles each -> 2.7x faster
// than the exact-fp32 kernel of conv.hip.  It is NOT bit-identical to an fp32 fmaf chain; results stay within
// the encoder tolerance (1e-5 relative, tests/test_hip_parity.py).  I2L_FLAG_EXACT_FP32 selects conv.hip's
// exact kernel instead.
//
// Tiling is the one of conv.hip: MFMA row i = 4*pp + 2*dy + dx (whole 2x2 pooling quads per lane), column =
// output channel, workgroup = 4 waves x (2 M-tiles x 2 N-tiles) = 8x32 or 16x16 conv positions x 64 channels.
// K order is (tap, 16 input channels): one MFMA k-step = 16 channels at one tap, lane half lh supplies channels
// 8*lh..8*lh+7, so operands are 16-byte ds_read_b128 from channels-last LDS images
//     in_s[split][row][lh][col (padded: row pairs land on disjoint halves of the 256-byte bank row)][8]
//     w_s [tap][split][lh][64 channels][8]
// Input channels go in chunks of 16: the next chunk (fp32 NCHW patch + pre-split weight slab) is prefetched into
// registers while the 216 MFMAs of the current chunk run; the fp32 -> 3 x bf16 split happens on the way to LDS.
#include <stdlib.h>

#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, bf16_t& s0, bf16_t& s1, bf16_t& s2) {
    s0 = f2bf(x);
    const float r1 = x - bf2f(s0);
    s1 = f2bf(r1);
    s2 = f2bf(r1 - bf2f(s1));
}

// two values at once: v_cvt_pk_bf16_f32 (round to nearest even, as f2bf) packs the pair into one dword
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& s0, unsigned& s1, unsigned& s2) {
    f32v2 v = {x0, x1};
    bf16v2 b0 = __builtin_convertvector(v, bf16v2);
    v -= __builtin_convertvector(b0, f32v2);
    bf16v2 b1 = __builtin_convertvector(v, bf16v2);
    v -= __builtin_convertvector(b1, f32v2);
    bf16v2 b2 = __builtin_convertvector(v, bf16v2);
    s0 = *reinterpret_cast<unsigned*>(&b0);
    s1 = *reinterpret_cast<unsigned*>(&b1);
    s2 = *reinterpret_cast<unsigned*>(&b2);
}

constexpr int CH = 16;                        // input channels per chunk (= one MFMA k-step per tap)
constexpr int CO_BLK = 64;
constexpr int W_SLAB_U4 = 9 * 3 * 2 * CO_BLK;   // 16-byte pieces of one weight chunk slab (55,296 B)
constexpr int W_PHASE_U4 = W_SLAB_U4 / 3;      // one filter row (3 taps) of it: what LDS holds at a time

template <int WX> struct Tile3 {
    static constexpr int TR = (4 / WX) * 4, TC = WX * 16;
    static constexpr int IR = TR + 2, IC = TC + 2;
    static constexpr int ICP = WX == 2 ? 36 : (WX == 1 ? 20 : 68);   // padded columns: 2*ICP*16 B == 128 (mod 256)
    static constexpr int ROW16 = 2 * ICP;               // 16-byte pieces per (split, row): [lh][col]
    static constexpr int IN_U4 = 3 * IR * ROW16;        // 16-byte pieces of the input image
};

// Training forward on split products (AM kernels).  The backward pass branches on two discrete facts of every pooling
// window -- which of its four conv outputs is largest (first maximum wins, as ATen) and whether the pooled value
// passes the ReLU -- and the split products (|err| ~ 1e-6 relative) could decide a near-tie differently from an fp32
// evaluation.  So a window whose runner-up is within 2^-13 of the maximum, or whose pre-activation is that close to
// zero, is put on a list (fix[0] = count, entries from fix[FIX_HDR]) and conv_pool_fixup_kernel re-evaluates just
// those windows with fp32 FMAs: the decisions are then those of an fp32 computation (~1e-4 of the windows are
// listed).  Exact ties are not listed: equal inputs give equal outputs in either arithmetic and the first index wins.
constexpr unsigned FIX_HDR = 64;                 // list header, in entries (256 bytes)
constexpr unsigned FIX_CAP = 1u << 20;           // capacity of the list (entries past it stay as computed)
constexpr float FIX_TOL = 0x1p-13f;

__device__ __forceinline__ void pool_window(float a0, float a1, float a2, float a3, float bv, float& outv, int& bi, bool& near) {
    float best = a0;
    bi = 0;
    if (a1 > best) { best = a1; bi = 1; }
    if (a2 > best) { best = a2; bi = 2; }
    if (a3 > best) { best = a3; bi = 3; }
    const float pre = best + bv;
    outv = fmaxf(pre, 0.f);
    const float scale = fmaxf(fmaxf(fabsf(best), fabsf(bv)), 1e-30f);
    const float thr = FIX_TOL * scale;
    const float d0 = best - a0, d1 = best - a1, d2 = best - a2, d3 = best - a3;
    const bool tie = (d0 > 0.f && d0 <= thr) || (d1 > 0.f && d1 <= thr) || (d2 > 0.f && d2 <= thr) || (d3 > 0.f && d3 <= thr);
    near = fabsf(pre) <= thr || (tie && pre > 0.f);
}
__device__ __forceinline__ void fix_append(unsigned* fix, size_t o) {
    const unsigned at = atomicAdd(fix, 1u);
    if (at < FIX_CAP) fix[FIX_HDR + at] = (unsigned)o;
}

// wpack3[cb][chunk][tap][split][lh][co 64][8] = split_s( w[cb*64+co][chunk*16 + 8*lh + j][tap] )
// flip != 0: filter of the DATA-GRADIENT convolution, element = w[ci][co][2-ky][2-kx] of the forward filter w
__global__ void conv_pack3_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp, int Cin, int Cout, int n_chunks,
                                  size_t total, int flip, unsigned* __restrict__ fix) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (.., co, j) of split 0
    if (idx == 0 && fix) fix[0] = 0;                                        // empty near-decision list for the conv launch
    if (idx >= total) return;
    size_t r = idx;
    const int j = (int)(r % 8); r /= 8;
    const int co_l = (int)(r % CO_BLK); r /= CO_BLK;
    const int lh = (int)(r % 2); r /= 2;
    const int tap = (int)(r % 9); r /= 9;
    const int chunk = (int)(r % n_chunks);
    const int cb = (int)(r / n_chunks);
    const int co = cb * CO_BLK + co_l, ci = chunk * CH + 8 * lh + j;
    const float v = (co < Cout && ci < Cin) ? (flip ? w[((size_t)ci * Cout + co) * 9 + (8 - tap)] : w[((size_t)co * Cin + ci) * 9 + tap]) : 0.f;
    bf16_t s[3];
    split3(v, s[0], s[1], s[2]);
    const size_t slab = ((size_t)cb * n_chunks + chunk) * (size_t)W_SLAB_U4 * 8;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp)
        wp[slab + ((((size_t)tap * 3 + sp) * 2 + lh) * CO_BLK + co_l) * 8 + j] = s[sp];
}

// FULL: plain full-resolution convolution (no bias / ReLU / pooling): the data gradient of a block, y is (B,Cout,H,W)
// AM: also write the pooling arg max (training forward); a template parameter so that the inference kernel is unchanged
// NT: 32-channel MFMA column tiles per wave (2 = a 64-channel block; 1 = Cout == 32, FULL only: block 1's data gradient)
template <int WX, bool FULL = false, bool AM = false, int NT = 2>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16x3_kernel(
    const float* __restrict__ x, const bf16_t* __restrict__ wpack, const float* __restrict__ bias,
    float* __restrict__ y, unsigned char* __restrict__ amax, int Cin, int H, int W, int Cout, int Hp, int Wp,
    int tiles_x, int n_chunks, int tiles_per_img, int n_items, int items_per_wg, unsigned* __restrict__ fix) {
    typedef Tile3<WX> TL;
    constexpr int PR = TL::TR / 2, PC = TL::TC / 2;
    __shared__ __attribute__((aligned(16))) uint4 in_s[TL::IN_U4];
    __shared__ __attribute__((aligned(16))) uint4 w_s[W_PHASE_U4];             // the 3 taps of one filter row
    __shared__ float out_s[CO_BLK * 65];                                       // fp32 staging of the pooled tile
    __shared__ unsigned char am_s[AM ? CO_BLK * 68 : 4];                       // pooling arg max of the tile (training)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave / WX, wx = wave % WX;
    // XCD-aware ids: workgroups are placed round-robin over the 8 XCDs (id mod 8), each with its own L2.  The
    // 64-channel blocks of ONE tile walk read the same input patches, so they get ids 8 apart (same XCD, dispatched
    // together, same pace): the second block's reads hit that XCD's L2 instead of going to HBM again.
    const int co_blocks_g = (Cout + CO_BLK - 1) / CO_BLK;
    const int lin = blockIdx.x, within = lin % (8 * co_blocks_g);
    const int cb = within >> 3;
    const int walk = (lin / (8 * co_blocks_g)) * 8 + (within & 7);
    const int i = lane & 31, h = lane >> 5;
    const int dx = i & 1, dy = (i >> 1) & 1, pp = i >> 2;
    // operand bases in 16-byte pieces
    const int a_base = ((wy * 4 + dy) * 2 + h) * TL::ICP + wx * 16 + 2 * pp + dx;
    const int b_base = h * CO_BLK + i;
    const uint4* wp_cb = reinterpret_cast<const uint4*>(wpack) + (size_t)cb * n_chunks * W_SLAB_U4;
    const int HW = H * W;

    // staging plan: piece e of this thread = 8 consecutive channels (half lh) of patch position (r, col)
    constexpr int NP = TL::IR * 2 * TL::IC;
    constexpr int NE = (NP + 255) / 256;
    constexpr int NWV = (W_PHASE_U4 + 255) / 256;
    constexpr int SPLIT_U4 = TL::IR * TL::ROW16;                    // 16-byte pieces between split planes
    int loff[NE], p_r[NE], p_col[NE], p_lh[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * 256;
        p_col[e] = idx % TL::IC;
        p_lh[e] = (idx / TL::IC) & 1;
        p_r[e] = idx / (2 * TL::IC);
        loff[e] = idx < NP ? (p_r[e] * 2 + p_lh[e]) * TL::ICP + p_col[e] : -1;
    }
    // the item whose chunks are being FETCHED (runs one phase ahead of the item being computed)
    const float* f_xb = x;
    int goff[NE];
    unsigned okmask = 0;
    auto setup_fetch = [&](int item) {
        const int fb = item / tiles_per_img, ft = item - fb * tiles_per_img;
        const int fty = ft / tiles_x, ftx = ft - fty * tiles_x;
        const int fy0 = fty * TL::TR, fx0 = ftx * TL::TC;
        f_xb = x + (size_t)fb * Cin * HW;
        okmask = 0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int iy = fy0 - 1 + p_r[e], ix = fx0 - 1 + p_col[e];
            goff[e] = (8 * p_lh[e] * H + iy) * W + ix;
            if (loff[e] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W) okmask |= 1u << e;
        }
    };
    float xin[NE][8];
    uint4 win[NWV];
    auto fetch_x = [&](int chunk) {
        const float* xc = f_xb + (size_t)chunk * CH * HW;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const bool ok = (okmask >> e) & 1u;
#pragma unroll
            for (int j = 0; j < 8; ++j) xin[e][j] = ok ? xc[goff[e] + j * HW] : 0.f;
        }
    };
    auto fetch_w = [&](int phase) {                                 // phase = chunk * 3 + filter row
        const uint4* wsrc = wp_cb + (size_t)phase * W_PHASE_U4;
#pragma unroll
        for (int e = 0; e < NWV; ++e) {
            const int idx = tid + e * 256;
            win[e] = idx < W_PHASE_U4 ? wsrc[idx] : make_uint4(0, 0, 0, 0);
        }
    };
    auto commit_x = [&]() {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (loff[e] >= 0) {
                unsigned s[3][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split3_pair(xin[e][2 * j], xin[e][2 * j + 1], s[0][j], s[1][j], s[2][j]);
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    in_s[loff[e] + sp * SPLIT_U4] = make_uint4(s[sp][0], s[sp][1], s[sp][2], s[sp][3]);
            }
        }
    };
    auto commit_w = [&]() {
#pragma unroll
        for (int e = 0; e < NWV; ++e) {
            const int idx = tid + e * 256;
            if (idx < W_PHASE_U4) w_s[idx] = win[e];
        }
    };

    // Two workgroups share a CU (LDS 2 x ~70 KB, <= 256 registers): while one sits in its barriers / staging the
    // other keeps the matrix cores busy.  A workgroup walks `items_per_wg` consecutive tiles of its 64-channel block
    // (the grid is sized to ONE resident round) and the first patch / filter row of the next tile are fetched
    // during the last phases of the current one, so only the first tile pays a prologue.
    // Per chunk: 3 phases (filter rows) of 3 taps x 24 MFMAs per wave.
    const int n_phases = 3 * n_chunks;
    const int it0 = walk * items_per_wg, it_end = min(it0 + items_per_wg, n_items);
    if (it0 >= n_items) return;
    setup_fetch(it0);
    fetch_x(0);
    fetch_w(0);
    for (int it = it0; it < it_end; ++it) {
        const int b = it / tiles_per_img, tl = it - b * tiles_per_img;
        const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
        const int y0 = ty * TL::TR, x0 = tx * TL::TC;
        f32x16 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

        for (int chunk = 0; chunk < n_chunks; ++chunk) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                __syncthreads();
                if (ky == 0) commit_x();
                commit_w();
                __syncthreads();
                if (chunk * 3 + ky + 1 < n_phases) fetch_w(chunk * 3 + ky + 1);
                else if (it + 1 < it_end) fetch_w(0);
                if (ky == 0) {
                    if (chunk + 1 < n_chunks) fetch_x(chunk + 1);
                    else if (it + 1 < it_end) { setup_fetch(it + 1); fetch_x(0); }
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    bf16x8 a[2][3], bw[NT][3];
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) {
                            const uint4 t = in_s[a_base + (sp * TL::IR + m * 2 + ky) * TL::ROW16 + kx];
                            a[m][sp] = *reinterpret_cast<const bf16x8*>(&t);
                        }
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) {
                            const uint4 t = w_s[b_base + ((kx * 3 + sp) * 2) * CO_BLK + n * 32];
                            bw[n][sp] = *reinterpret_cast<const bf16x8*>(&t);
                        }
                    // smallest partial products first; consecutive MFMAs go to different accumulators
                    constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][TI[t]], bw[n][TJ[t]], acc[m][n], 0, 0, 0);
                }
            }
        }

        // ---- epilogue (same lane layout as conv.hip): registers 4q..4q+3 = the 2x2 quad of pooled column 2q + h
        const int py0 = y0 >> 1, px0 = x0 >> 1;
        if constexpr (FULL) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = cb * CO_BLK + n * 32 + i;
                if (co >= Cout) continue;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int py = py0 + wy * 2 + m;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int px = px0 + wx * 8 + 2 * q + h;
#pragma unroll
                        for (int ry = 0; ry < 2; ++ry) {             // quad element e = 2 ry + rx: two adjacent columns
                            const int Y = 2 * py + ry, X = 2 * px;
                            if (Y < H && X + 1 < W)
                                *reinterpret_cast<float2*>(y + (((size_t)b * Cout + co) * H + Y) * W + X) =
                                    make_float2(acc[m][n][4 * q + 2 * ry], acc[m][n][4 * q + 2 * ry + 1]);
                        }
                    }
                }
            }
            continue;
        }
        if ((Wp & 3) == 0 && py0 + PR <= Hp && px0 + PC <= Wp && (cb + 1) * CO_BLK <= Cout) {
            // out_s was last read before the first barrier of this tile's first phase: free to overwrite
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float bv = bias[cb * CO_BLK + n * 32 + i];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int pos = (wy * 2 + m) * PC + wx * 8 + 2 * q + h;
                        if constexpr (AM) {                  // training forward: first maximum wins, as ATen's max_pool2d
                            float ov;
                            int bi;
                            bool near;
                            pool_window(acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3], bv, ov, bi, near);
                            out_s[(n * 32 + i) * 65 + pos] = ov;
                            am_s[(n * 32 + i) * 68 + pos] = (unsigned char)bi;
                            if (near && fix)
                                fix_append(fix, (((size_t)b * Cout + cb * CO_BLK + n * 32 + i) * Hp + py0 + wy * 2 + m) * Wp + px0 + wx * 8 + 2 * q + h);
                        } else {
                            const float v = fmaxf(fmaxf(acc[m][n][4 * q], acc[m][n][4 * q + 1]),
                                                  fmaxf(acc[m][n][4 * q + 2], acc[m][n][4 * q + 3])) + bv;
                            out_s[(n * 32 + i) * 65 + pos] = fmaxf(v, 0.f);
                        }
                    }
            }
            __syncthreads();
            for (int idx = tid; idx < CO_BLK * 16; idx += 256) {
                constexpr int XG = PC / 4;
                const int x4 = idx % XG, py = (idx / XG) % PR, co_l = idx >> 4;
                const float* sp = &out_s[co_l * 65 + py * PC + 4 * x4];
                const size_t o = (((size_t)b * Cout + cb * CO_BLK + co_l) * Hp + py0 + py) * Wp + px0 + 4 * x4;
                *reinterpret_cast<float4*>(y + o) = make_float4(sp[0], sp[1], sp[2], sp[3]);
                if constexpr (AM) *reinterpret_cast<uchar4*>(amax + o) = *reinterpret_cast<const uchar4*>(&am_s[co_l * 68 + py * PC + 4 * x4]);
            }
            continue;
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = cb * CO_BLK + n * 32 + i;
            if (co >= Cout) continue;
            const float bv = bias[co];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int py = py0 + wy * 2 + m;
                if (py >= Hp) continue;
                const size_t rowoff = (((size_t)b * Cout + co) * Hp + py) * Wp;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int px = px0 + wx * 8 + 2 * q + h;
                    float ov;
                    int bi;
                    bool near;
                    pool_window(acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3], bv, ov, bi, near);
                    if (px < Wp) {
                        y[rowoff + px] = ov;
                        if (amax) amax[rowoff + px] = (unsigned char)bi;
                        if (AM && near && fix) fix_append(fix, rowoff + px);
                    }
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// First conv block (Cin <= 3).  The staged input patch is kept channels-last and already split: one 8-byte pixel
// {c0, c1, c2, 0} per split, so a lane's A operand of one k-step is just TWO pixels (two taps x 4 channels = 8 k)
// read by two ds_read_b64 -- no im2col image, no per-tap splitting.  K = 9 taps x 4 = 36 -> 3 k-steps of 16 (taps
// 9..11 carry zero weights); the filter bank (32 output channels x 48 k x 3 splits) sits in 36 registers per lane
// for the whole kernel.  A workgroup walks the column tiles of an 8-row band (8 x 32 positions per tile, 2 M-tiles
// per wave, 18 MFMAs per M-tile) with the next patch prefetched in registers.  The two M-tiles of a wave run one after
// the other through ONE accumulator (`#pragma unroll 1`): 101-128 registers instead of 166, so FOUR workgroups share a CU
// (~20 KB of LDS each) and one's staging / epilogue hides behind the others' MFMAs: 0.106 -> 0.095 ms at B=256, 78 ->
// 66 us for the 64-image training forward.  The pooled tile goes through LDS so that every store
// instruction writes 64-byte runs; storing 16 bytes per lane straight from the accumulators (v_permlane32_swap to
// gather a lane's four columns) saves the transpose and a barrier but halves the run length: 0.106 -> 0.122 ms.
// ---------------------------------------------------------------------------------------------------------------
template <int CIN, bool AM = false>
__global__ __launch_bounds__(256, 4) void conv3x3_smallk_bf16x3_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
    unsigned char* __restrict__ amax, int H, int W, int Cout, int Hp, int Wp, int tiles_x, int tiles_per_part,
    unsigned* __restrict__ fix) {
    constexpr int RS = 48;                                   // pixels per patch row: rows r, r+1 half a bank row apart
    constexpr int PLANE = 10 * RS;                           // pixels per split plane
    __shared__ __attribute__((aligned(16))) uint2 img[3 * PLANE];
    __shared__ float out_s[32 * 65];
    __shared__ unsigned char am_s[AM ? 32 * 68 : 4];         // pooling arg max of the tile (training forward)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    // blockIdx.x = (part of the band's column tiles, 32-channel block): small batches split the walk so that the grid
    // still fills the chip
    const int cbs = (Cout + 31) / 32;
    const int cb = blockIdx.x % cbs, b = blockIdx.z;
    const int tx_begin = (blockIdx.x / cbs) * tiles_per_part, tx_end = min(tiles_x, tx_begin + tiles_per_part);
    const int y0 = blockIdx.y * 8;

    // B operand: k-step s, lane half h -> taps 4s + 2h, 4s + 2h + 1, channels 0..3 each
    bf16x8 bw[3][3];
    {
        const int co = cb * 32 + i;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            unsigned pk[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tap = 4 * s + 2 * h + (j >> 1), c0 = 2 * (j & 1);
                const bool okt = tap < 9 && co < Cout;
                const float v0 = (okt && c0 < CIN) ? w[((size_t)co * CIN + c0) * 9 + tap] : 0.f;
                const float v1 = (okt && c0 + 1 < CIN) ? w[((size_t)co * CIN + c0 + 1) * 9 + tap] : 0.f;
                split3_pair(v0, v1, pk[0][j], pk[1][j], pk[2][j]);
            }
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                const uint4 t = make_uint4(pk[sp][0], pk[sp][1], pk[sp][2], pk[sp][3]);
                bw[s][sp] = *reinterpret_cast<const bf16x8*>(&t);
            }
        }
    }
    const float bv = (cb * 32 + i < Cout) ? bias[cb * 32 + i] : 0.f;

    // A operand addresses (pixel units): M-tile m covers rows wy*4 + 2m + dy, cols wx*16 + 2pp + dx
    const int pbase = (wy * 4 + ((i >> 1) & 1)) * RS + wx * 16 + 2 * (i >> 2) + (i & 1);
    int aoff[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int tap = min(4 * s + 2 * h + e, 8);       // taps >= 9 have zero weights: read any valid pixel
            aoff[s][e] = pbase + (tap / 3) * RS + (tap % 3);
        }

    // staging: pixel q = tid, tid + 256 of the 10 x 34 patch
    const float* xb = x + (size_t)b * CIN * H * W;
    const size_t HW = (size_t)H * W;
    int q_lds[2], q_col[2];
    long q_goff[2];
    bool q_rowok[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int q = tid + e * 256;
        const int r = q / 34, col = q - r * 34;
        const int iy = y0 - 1 + r;
        q_lds[e] = q < 340 ? r * RS + col : -1;
        q_col[e] = col - 1;
        q_rowok[e] = q < 340 && iy >= 0 && iy < H;
        q_goff[e] = (long)iy * W + col - 1;
    }
    float pin[2][CIN];
    auto fetch = [&](int tx) {
        const int x0 = tx * 32;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ix = x0 + q_col[e];
            const bool ok = q_rowok[e] && ix >= 0 && ix < W;
#pragma unroll
            for (int c = 0; c < CIN; ++c) pin[e][c] = ok ? xb[c * HW + q_goff[e] + x0] : 0.f;
        }
    };
    fetch(tx_begin);
    for (int tx = tx_begin; tx < tx_end; ++tx) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (q_lds[e] >= 0) {
                unsigned p0[3], p1[3];
                split3_pair(pin[e][0], CIN > 1 ? pin[e][CIN > 1 ? 1 : 0] : 0.f, p0[0], p0[1], p0[2]);
                split3_pair(CIN > 2 ? pin[e][CIN > 2 ? 2 : 0] : 0.f, 0.f, p1[0], p1[1], p1[2]);
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) img[sp * PLANE + q_lds[e]] = make_uint2(p0[sp], p1[sp]);
            }
        }
        __syncthreads();                                    // patch complete; out_s of the last tile consumed
        if (tx + 1 < tx_end) fetch(tx + 1);
        // the two M-tiles one after the other (one accumulator, one set of operands live at a time: the register
        // budget of four workgroups per CU); registers 4q..4q+3 = the 2x2 quad of pooled column 2q + h (conv.hip
        // layout), channel = lane & 31
#pragma unroll 1
        for (int m = 0; m < 2; ++m) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                bf16x8 a[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    const uint2 t0 = img[sp * PLANE + m * 2 * RS + aoff[s][0]];
                    const uint2 t1 = img[sp * PLANE + m * 2 * RS + aoff[s][1]];
                    const uint4 t = make_uint4(t0.x, t0.y, t1.x, t1.y);
                    a[sp] = *reinterpret_cast<const bf16x8*>(&t);
                }
                constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TI[t]], bw[s][TJ[t]], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pos = (wy * 2 + m) * 16 + wx * 8 + 2 * q + h;
                if constexpr (AM) {                          // first maximum wins, as ATen's max_pool2d
                    float ov;
                    int bi;
                    bool near;
                    pool_window(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], bv, ov, bi, near);
                    out_s[i * 65 + pos] = ov;
                    am_s[i * 68 + pos] = (unsigned char)bi;
                    const int fpy = (y0 >> 1) + wy * 2 + m, fpx = tx * 16 + wx * 8 + 2 * q + h;
                    if (near && fix && cb * 32 + i < Cout && fpy < Hp && fpx < Wp)
                        fix_append(fix, (((size_t)b * Cout + cb * 32 + i) * Hp + fpy) * Wp + fpx);
                } else {
                    const float v = fmaxf(fmaxf(acc[4 * q], acc[4 * q + 1]), fmaxf(acc[4 * q + 2], acc[4 * q + 3])) + bv;
                    out_s[i * 65 + pos] = fmaxf(v, 0.f);
                }
            }
        }
        __syncthreads();                                    // pooled tile complete (and every wave is past its A reads)
        const int py0 = y0 >> 1, px0 = tx * 16;
        if ((Wp & 3) == 0 && py0 + 4 <= Hp && px0 + 16 <= Wp && (cb + 1) * 32 <= Cout) {
            for (int idx = tid; idx < 32 * 16; idx += 256) {
                const int x4 = idx & 3, py = (idx >> 2) & 3, c = idx >> 4;
                const float* sp = &out_s[c * 65 + py * 16 + 4 * x4];
                const size_t o = (((size_t)b * Cout + cb * 32 + c) * Hp + py0 + py) * Wp + px0 + 4 * x4;
                *reinterpret_cast<float4*>(y + o) = make_float4(sp[0], sp[1], sp[2], sp[3]);
                if constexpr (AM) *reinterpret_cast<uchar4*>(amax + o) = *reinterpret_cast<const uchar4*>(&am_s[c * 68 + py * 16 + 4 * x4]);
            }
        } else {
            for (int idx = tid; idx < 32 * 64; idx += 256) {
                const int px = idx & 15, py = (idx >> 4) & 3, c = idx >> 6;
                if (cb * 32 + c < Cout && py0 + py < Hp && px0 + px < Wp) {
                    const size_t o = (((size_t)b * Cout + cb * 32 + c) * Hp + py0 + py) * Wp + px0 + px;
                    y[o] = out_s[c * 65 + py * 16 + px];
                    if constexpr (AM) amax[o] = am_s[c * 68 + py * 16 + px];
                }
            }
        }
    }
}

// One wave per listed window: the four conv outputs again with fp32 FMAs (lanes over input channels, butterfly sum),
// then bias / ReLU / first maximum exactly as the fused epilogues.
__global__ __launch_bounds__(256) void conv_pool_fixup_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              unsigned char* __restrict__ amax, const unsigned* __restrict__ fix,
                                                              int Cin, int H, int W, int Cout, int Hp, int Wp) {
    const unsigned n = min(fix[0], FIX_CAP);
    const int lane = threadIdx.x & 63;
    for (unsigned e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
        const unsigned o = fix[FIX_HDR + e];
        const int px = (int)(o % Wp), py = (int)((o / Wp) % Hp);
        const int co = (int)((o / ((unsigned)Wp * Hp)) % Cout), b = (int)(o / ((unsigned)Wp * Hp * Cout));
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ci = lane; ci < Cin; ci += 64) {
            const float* xp = x + ((size_t)b * Cin + ci) * H * W;
            const float* wp = w + ((size_t)co * Cin + ci) * 9;
            float win[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int iy = 2 * py - 1 + r, ix = 2 * px - 1 + c;
                    win[r][c] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xp[(size_t)iy * W + ix] : 0.f;
                }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wv = wp[ky * 3 + kx];
#pragma unroll
                    for (int q = 0; q < 4; ++q) s[q] = fmaf(win[(q >> 1) + ky][(q & 1) + kx], wv, s[q]);
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s[q] += __shfl_xor(s[q], off);
        if (lane == 0) {
            float best = s[0];
            int bi = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (s[q] > best) { best = s[q]; bi = q; }
            y[o] = fmaxf(best + bias[co], 0.f);
            amax[o] = (unsigned char)bi;
        }
    }
}

}  // namespace

bool i2l_conv_bf16x3_applicable(int Cin, int Cout) {
    return Cin % CH == 0 && Cout % CO_BLK == 0;
}
// the full-resolution (data gradient) flavour also takes Cout == 32: one MFMA column tile per wave, half a filter block
bool i2l_conv_bf16x3_full_applicable(int Cin, int Cout) {
    return Cin % CH == 0 && (Cout % CO_BLK == 0 || Cout == 32);
}

size_t i2l_conv_bf16x3_workspace_bytes(int Cin, int Cout) {
    return i2l_align((size_t)((Cout + CO_BLK - 1) / CO_BLK) * (Cin / CH) * W_SLAB_U4 * 16);
}
// near-decision list of the training forward (both split kernels): header + FIX_CAP entries
size_t i2l_conv_fixlist_bytes(void) { return i2l_align((size_t)(FIX_HDR + FIX_CAP) * sizeof(unsigned)); }

namespace {
int launch_fixup(const float* x, const float* w, const float* bias, float* y, unsigned char* amax, const unsigned* fix,
                 int B, int Cin, int H, int W, int Cout, hipStream_t s) {
    if ((size_t)B * Cout * (H / 2) * (W / 2) > 0xffffffffull) return I2L_ERR_UNSUPPORTED;     // entries are 32-bit offsets
    hipLaunchKernelGGL(conv_pool_fixup_kernel, dim3(512), dim3(256), 0, s, x, w, bias, y, amax, fix, Cin, H, W, Cout, H / 2, W / 2);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
}  // namespace

int i2l_conv_bf16x3_run(const float* x, const float* w, const float* bias, float* y, unsigned char* amax, int B, int Cin,
                        int H, int W, int Cout, void* workspace, size_t workspace_bytes, hipStream_t s, int full,
                        int weights_packed) {
    if (full && ((H | W) & 1)) return I2L_ERR_UNSUPPORTED;       // full resolution is written quad by quad
    const size_t pack_bytes = i2l_conv_bf16x3_workspace_bytes(Cin, Cout);
    if (workspace_bytes < pack_bytes + (amax ? i2l_conv_fixlist_bytes() : 0) || !workspace) return I2L_ERR_WORKSPACE;
    unsigned* fix = amax ? reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + pack_bytes) : nullptr;
    if (amax && (size_t)B * Cout * (H / 2) * (W / 2) > 0xffffffffull) return I2L_ERR_UNSUPPORTED;
    const int Hp = H / 2, Wp = W / 2;
    const int n_chunks = Cin / CH, co_blocks = (Cout + CO_BLK - 1) / CO_BLK;
    if (Cout % CO_BLK != 0 && !(full && Cout == 32)) return I2L_ERR_UNSUPPORTED;
    bf16_t* wp = static_cast<bf16_t*>(workspace);
    const size_t total = (size_t)co_blocks * n_chunks * 9 * 2 * CO_BLK * 8;
    // weights_packed (I2L_FLAG_WEIGHTS_PACKED, inference only): the caller kept this workspace since an earlier call with
    // the same weights, whose packed image it still holds -- the weight-only work is not repeated per batch
    if (!(weights_packed && !amax && !full))
    hipLaunchKernelGGL(conv_pack3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wp, Cin, Cout,
                       n_chunks, total, full ? 1 : 0, fix);
    I2L_CHECK_LAUNCH();
    const int rows = 2 * Hp, cols = 2 * Wp;
    // tile shape with the fewest wasted positions (8x32, 16x16 or 4x64; ties: 8x32)
    int wx = 2;
    long bestw = -1;
    const int cand[3] = {2, 1, 4};
    for (int c = 0; c < 3; ++c) {
        const int tr = (4 / cand[c]) * 4, tc = cand[c] * 16;
        const long cover = (long)i2l_cdiv(rows, tr) * tr * i2l_cdiv(cols, tc) * tc;
        if (bestw < 0 || cover < bestw) { bestw = cover; wx = cand[c]; }
    }
    const int tr = (4 / wx) * 4, tc = wx * 16;
    const int tiles_x = i2l_cdiv(cols, tc), tiles_y = i2l_cdiv(rows, tr);
    // one resident round: 2 workgroups per CU x 256 CUs, each walking items_per_wg consecutive tiles
    const int tiles_per_img = tiles_x * tiles_y;
    const long long n_items_ll = (long long)tiles_per_img * B;
    if (n_items_ll > 0x7fffffffll) return I2L_ERR_UNSUPPORTED;
    const int n_items = (int)n_items_ll;
    int items_per_wg = (int)((n_items_ll * co_blocks + 511) / 512);
    // a launch with little more than one resident round of single tiles (block 3 of a 64-image training batch: 640):
    // one tile per workgroup fills the CUs more evenly than 320 walks of two (86 -> 73 us); measured worse above that
    if (n_items_ll * co_blocks <= 768) items_per_wg = 1;
    if (items_per_wg < 1) items_per_wg = 1;
    const int walks = i2l_cdiv(n_items, items_per_wg);
    dim3 grid((unsigned)(i2l_cdiv(walks, 8) * 8 * co_blocks));       // ids decoded in the kernel (XCD-aware)
#define I2L_LAUNCH3(WXV)                                                                                              \
    do {                                                                                                              \
        if (full && Cout == 32)                                                                                       \
            hipLaunchKernelGGL((conv3x3_bf16x3_kernel<WXV, true, false, 1>), grid, dim3(256), 0, s, x, (const bf16_t*)wp, bias, y, \
                               amax, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks, tiles_per_img, n_items, items_per_wg, fix); \
        else if (full)                                                                                                \
            hipLaunchKernelGGL((conv3x3_bf16x3_kernel<WXV, true>), grid, dim3(256), 0, s, x, (const bf16_t*)wp, bias, y,  \
                               amax, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks, tiles_per_img, n_items, items_per_wg, fix); \
        else if (amax)                                                                                                \
            hipLaunchKernelGGL((conv3x3_bf16x3_kernel<WXV, false, true>), grid, dim3(256), 0, s, x, (const bf16_t*)wp, bias, y, \
                               amax, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks, tiles_per_img, n_items, items_per_wg, fix); \
        else                                                                                                          \
            hipLaunchKernelGGL((conv3x3_bf16x3_kernel<WXV, false>), grid, dim3(256), 0, s, x, (const bf16_t*)wp, bias, y, \
                               amax, Cin, H, W, Cout, Hp, Wp, tiles_x, n_chunks, tiles_per_img, n_items, items_per_wg, fix); \
    } while (0)
    if (wx == 2) I2L_LAUNCH3(2);
    else if (wx == 1) I2L_LAUNCH3(1);
    else I2L_LAUNCH3(4);
#undef I2L_LAUNCH3
    I2L_CHECK_LAUNCH();
    if (amax) return launch_fixup(x, w, bias, y, amax, fix, B, Cin, H, W, Cout, s);
    return I2L_OK;
}

bool i2l_conv_smallk_applicable(int Cin, int Cout) {
    return Cin >= 1 && Cin <= 3 && Cout % 32 == 0;
}

int i2l_conv_smallk_run(const float* x, const float* w, const float* bias, float* y, unsigned char* amax, int B, int Cin,
                        int H, int W, int Cout, void* workspace, size_t workspace_bytes, hipStream_t s) {
    const int Hp = H / 2, Wp = W / 2;
    unsigned* fix = nullptr;
    if (amax) {                                                 // training forward: near-decision list at the workspace's start
        if (!workspace || workspace_bytes < i2l_conv_fixlist_bytes()) return I2L_ERR_WORKSPACE;
        if ((size_t)B * Cout * Hp * Wp > 0xffffffffull) return I2L_ERR_UNSUPPORTED;
        fix = static_cast<unsigned*>(workspace);
        if (hipMemsetAsync(fix, 0, sizeof(unsigned), s) != hipSuccess) return I2L_ERR_LAUNCH;
    }
    const int tiles_x = i2l_cdiv(2 * Wp, 32), bands = i2l_cdiv(2 * Hp, 8);
    if (bands > 65535 || B > 65535) return I2L_ERR_UNSUPPORTED;
    // at least ~6 workgroups per CU (3 resident): split each band's tile walk into parts when the batch is small
    int parts = 1;
    while (parts < tiles_x && (long)(Cout / 32) * bands * B * parts < 1536) ++parts;
    const int tiles_per_part = i2l_cdiv(tiles_x, parts);
    parts = i2l_cdiv(tiles_x, tiles_per_part);
    dim3 grid((Cout / 32) * parts, bands, B);
#define I2L_LAUNCH_SK(C, A) hipLaunchKernelGGL((conv3x3_smallk_bf16x3_kernel<C, A>), grid, dim3(256), 0, s, x, w, bias, y, amax, H, W, Cout, Hp, Wp, tiles_x, tiles_per_part, fix)
    if (amax) {
        if (Cin == 1) I2L_LAUNCH_SK(1, true);
        else if (Cin == 2) I2L_LAUNCH_SK(2, true);
        else I2L_LAUNCH_SK(3, true);
    } else {
        if (Cin == 1) I2L_LAUNCH_SK(1, false);
        else if (Cin == 2) I2L_LAUNCH_SK(2, false);
        else I2L_LAUNCH_SK(3, false);
    }
#undef I2L_LAUNCH_SK
    I2L_CHECK_LAUNCH();
    if (amax) return launch_fixup(x, w, bias, y, amax, fix, B, Cin, H, W, Cout, s);
    return I2L_OK;
}
