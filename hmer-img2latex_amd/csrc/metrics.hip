// Sequence metrics of the evaluation loop on the device (reference img2latex/training/metrics.py):
// per (prediction, target) pair the INTEGER statistics -- Levenshtein distance table corner (:49-94), clipped n-gram
// match counts of BLEU-n (:97-181), token_list_accuracy counts (:241-277) -- and masked_accuracy's argmax / compare /
// mask counts over a (rows, V) logits matrix (:226-238), so neither the id lists nor the (B,T,V) logits travel to the
// host per step.  The float formulas on top (similarity = 1 - d / max_len, geometric mean, brevity penalty) are a few
// float64 operations per pair and stay in the host wrapper, written exactly as the reference writes them.
#include "common.h"

namespace {

constexpr int MT = 256;

// One workgroup per pair.  LDS: a[R] | b[C] | three anti-diagonals of the distance table (C+1 each) | counters.
__global__ __launch_bounds__(MT) void sequence_metrics_kernel(const int32_t* __restrict__ pred, const int32_t* __restrict__ pred_len,
                                                              int pred_stride, const int32_t* __restrict__ tgt,
                                                              const int32_t* __restrict__ tgt_len, int tgt_stride, int max_len,
                                                              int max_n, int pad_id, int32_t* __restrict__ lev_out,
                                                              int32_t* __restrict__ match_out, int32_t* __restrict__ tla_out) {
    extern __shared__ int sm[];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int R = min(max(pred_len[pair], 0), max_len), C = min(max(tgt_len[pair], 0), max_len);
    int* a = sm;                       // sequence_one = prediction (rows of the table)
    int* b = a + max_len;              // sequence_two = target (columns)
    int* dg = b + max_len;             // 3 x (max_len + 1)
    int* cnt = dg + 3 * (max_len + 1); // [0..3] n-gram matches, [4] correct, [5] non-pad
    for (int i = tid; i < R; i += MT) a[i] = pred[(size_t)pair * pred_stride + i];
    for (int i = tid; i < C; i += MT) b[i] = tgt[(size_t)pair * tgt_stride + i];
    if (tid < 6) cnt[tid] = 0;
    __syncthreads();

    // ---- Levenshtein (metrics.py:63-83): cell (r, c) on anti-diagonal d = r + c; equal tokens copy the diagonal
    // neighbour, otherwise 1 + min(up, left, diagonal) -- the reference's recurrence, not a generic edit distance
    int* p2 = dg;
    int* p1 = dg + (max_len + 1);
    int* cur = dg + 2 * (max_len + 1);
    for (int d = 0; d <= R + C; ++d) {
        for (int c = tid; c <= C; c += MT) {
            const int r = d - c;
            if (r < 0 || r > R) continue;
            int v;
            if (r == 0) v = c;
            else if (c == 0) v = r;
            else if (a[r - 1] == b[c - 1]) v = p2[c - 1];
            else v = 1 + min(min(p1[c], p1[c - 1]), p2[c - 1]);
            cur[c] = v;
        }
        __syncthreads();
        int* t = p2; p2 = p1; p1 = cur; cur = t;
    }
    if (tid == 0) lev_out[pair] = p1[C];           // p1 = the last diagonal written

    // ---- BLEU-n clipped matches (metrics.py:136-160): sum over DISTINCT generated n-grams of min(count in the
    // prediction, count in the target); thread i owns the n-gram starting at i and counts it iff it is its first occurrence
    for (int g = 1; g <= max_n; ++g) {
        const int ng = R - g + 1, nt = C - g + 1;
        int local = 0;
        if (ng > 0 && nt > 0) {
            for (int i = tid; i < ng; i += MT) {
                bool first = true;
                int cg = 0, ct = 0;
                for (int j = 0; j < ng; ++j) {
                    bool eq = true;
                    for (int k = 0; k < g; ++k) eq = eq && a[j + k] == a[i + k];
                    cg += eq ? 1 : 0;
                    if (eq && j < i) first = false;
                }
                for (int j = 0; j < nt; ++j) {
                    bool eq = true;
                    for (int k = 0; k < g; ++k) eq = eq && b[j + k] == a[i + k];
                    ct += eq ? 1 : 0;
                }
                if (first) local += min(cg, ct);
            }
        }
        if (local) atomicAdd(&cnt[g - 1], local);
    }
    // ---- token_list_accuracy (metrics.py:259-274): positions below the shorter length, padding targets ignored
    {
        const int ml = min(R, C);
        int correct = 0, nonpad = 0;
        for (int i = tid; i < ml; i += MT) {
            const bool np = b[i] != pad_id;
            nonpad += np ? 1 : 0;
            correct += (np && a[i] == b[i]) ? 1 : 0;
        }
        if (correct) atomicAdd(&cnt[4], correct);
        if (nonpad) atomicAdd(&cnt[5], nonpad);
    }
    __syncthreads();
    if (tid < 4) match_out[(size_t)pair * 4 + tid] = tid < max_n ? cnt[tid] : 0;
    if (tid < 2 && tla_out) tla_out[(size_t)pair * 2 + tid] = cnt[4 + tid];
}

// One pass of a four-token window over seq[0..len): cnt[n-1] += (the n-gram at j equals own[0..n)) for n = 1..4; an n-gram
// at j exists while j + n <= len.  OWN: the sequence is the one position i lives in -- a match at j < i means i is not
// the first occurrence of its n-gram (each distinct n-gram is counted once, at its first occurrence: metrics.py:136-160).
template <bool OWN>
__device__ __forceinline__ void ngram_scan(const int* __restrict__ seq, int len, int i, const int (&own)[4], int (&cnt)[4],
                                           bool (&first)[4]) {
    int w0 = len > 0 ? seq[0] : 0, w1 = len > 1 ? seq[1] : 0, w2 = len > 2 ? seq[2] : 0, w3 = len > 3 ? seq[3] : 0;
    for (int j = 0; j < len; ++j) {
        const int nx = j + 4 < len ? seq[j + 4] : 0;
        const bool e1 = w0 == own[0];
        const bool e2 = e1 && j + 2 <= len && w1 == own[1];
        const bool e3 = e2 && j + 3 <= len && w2 == own[2];
        const bool e4 = e3 && j + 4 <= len && w3 == own[3];
        cnt[0] += e1 ? 1 : 0;
        cnt[1] += e2 ? 1 : 0;
        cnt[2] += e3 ? 1 : 0;
        cnt[3] += e4 ? 1 : 0;
        if (OWN && j < i) {
            first[0] = first[0] && !e1;
            first[1] = first[1] && !e2;
            first[2] = first[2] && !e3;
            first[3] = first[3] && !e4;
        }
        w0 = w1; w1 = w2; w2 = w3; w3 = nx;
    }
}

// The same statistics with the Levenshtein distance from Myers' bit-vector algorithm (block formulation, J. ACM 46(3)
// 1999; global-distance boundary as in Hyyro 2003) instead of the anti-diagonal table walk: the reference's recurrence
// (metrics.py:73-81: equal tokens copy the diagonal, otherwise 1 + min of the three neighbours) IS the Levenshtein
// recurrence, so the table corner is the edit distance and any exact algorithm gives the same integer.  Wave 0 owns the
// distance: the prediction's tokens sit in registers (lane = position within a 64-token block, W blocks), and per
// target token a ballot yields the match mask of a block, which ~20 scalar bit operations turn into the block's new
// vertical delta vectors and its horizontal carry into the next block -- C x W short dependent steps instead of R + C
// workgroup barriers (150 x 150 tokens: ~30 us instead of ~300).  Waves 1-3 count the n-gram matches and the accuracy
// meanwhile.  W <= 4 (sequences up to 256 tokens); longer ones take the table kernel above.
template <int W>
__global__ __launch_bounds__(MT) void sequence_metrics_bp_kernel(const int32_t* __restrict__ pred, const int32_t* __restrict__ pred_len,
                                                                 int pred_stride, const int32_t* __restrict__ tgt,
                                                                 const int32_t* __restrict__ tgt_len, int tgt_stride, int max_len,
                                                                 int max_n, int pad_id, int32_t* __restrict__ lev_out,
                                                                 int32_t* __restrict__ match_out, int32_t* __restrict__ tla_out) {
    typedef unsigned long long u64;
    extern __shared__ int sm[];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = min(max(pred_len[pair], 0), max_len), C = min(max(tgt_len[pair], 0), max_len);
    int* a = sm;
    int* b = a + max_len;
    int* cnt = b + max_len;
    for (int i = tid; i < R; i += MT) a[i] = pred[(size_t)pair * pred_stride + i];
    for (int i = tid; i < C; i += MT) b[i] = tgt[(size_t)pair * tgt_stride + i];
    if (tid < 6) cnt[tid] = 0;
    __syncthreads();
    if (wave == 0) {
        int score = R;
        if (R == 0) score = C;
        else if (C > 0) {
            int pa[W];
            bool ok[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                ok[w] = 64 * w + lane < R;
                pa[w] = ok[w] ? a[64 * w + lane] : 0;
            }
            u64 Pv[W], Mv[W];
#pragma unroll
            for (int w = 0; w < W; ++w) { Pv[w] = ~0ull; Mv[w] = 0ull; }
            const int last = (R - 1) >> 6;
            const u64 high = 1ull << ((R - 1) & 63);
            for (int j = 0; j < C; ++j) {
                const int t = b[j];
                int hin = 1;                                    // D[0][j] - D[0][j-1] = +1: global distance
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    if (w <= last) {
                        u64 Eq = __ballot(ok[w] && pa[w] == t);
                        const u64 pv = Pv[w], mv = Mv[w];
                        const u64 Xv = Eq | mv;
                        if (hin < 0) Eq |= 1ull;
                        const u64 Xh = (((Eq & pv) + pv) ^ pv) | Eq;
                        u64 Ph = mv | ~(Xh | pv);
                        u64 Mh = pv & Xh;
                        const u64 hb = w == last ? high : (1ull << 63);
                        const int hout = (Ph & hb) ? 1 : ((Mh & hb) ? -1 : 0);
                        Ph <<= 1;
                        Mh <<= 1;
                        if (hin < 0) Mh |= 1ull;
                        else if (hin > 0) Ph |= 1ull;
                        Pv[w] = Mh | ~(Xv | Ph);
                        Mv[w] = Ph & Xv;
                        hin = hout;
                    }
                }
                score += hin;                                   // D[R][j+1] - D[R][j]
            }
        }
        if (lane == 0) lev_out[pair] = score;
    } else {
        const int t3 = tid - 64, NT3 = MT - 64;
        // clipped n-gram matches for n = 1..4 in ONE pass per sequence: the thread that owns position i keeps its four
        // tokens in registers and slides a four-token window over the prediction (count of its n-gram; "is i its first
        // occurrence?") and over the target (count there): one LDS read per window position, where a loop per n with
        // the n-grams compared out of LDS made 2 n dependent reads (0.26 ms per 256 pairs of 150 tokens, r04)
        int local[4] = {0, 0, 0, 0};
        for (int i = t3; i < R; i += NT3) {
            int ai[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ai[k] = i + k < R ? a[i + k] : 0;
            int cg[4] = {0, 0, 0, 0}, ct[4] = {0, 0, 0, 0};
            bool first[4] = {true, true, true, true};
            ngram_scan<true>(a, R, i, ai, cg, first);
            ngram_scan<false>(b, C, i, ai, ct, first);
#pragma unroll
            for (int g = 1; g <= 4; ++g)            // i < R - g + 1: position i starts an n-gram of the prediction
                if (g <= max_n && i < R - g + 1 && C - g + 1 > 0 && first[g - 1]) local[g - 1] += min(cg[g - 1], ct[g - 1]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (local[g]) atomicAdd(&cnt[g], local[g]);
        const int ml = min(R, C);
        int correct = 0, nonpad = 0;
        for (int i = t3; i < ml; i += NT3) {
            const bool np = b[i] != pad_id;
            nonpad += np ? 1 : 0;
            correct += (np && a[i] == b[i]) ? 1 : 0;
        }
        if (correct) atomicAdd(&cnt[4], correct);
        if (nonpad) atomicAdd(&cnt[5], nonpad);
    }
    __syncthreads();
    if (tid < 4) match_out[(size_t)pair * 4 + tid] = tid < max_n ? cnt[tid] : 0;
    if (tid < 2 && tla_out) tla_out[(size_t)pair * 2 + tid] = cnt[4 + tid];
}

// masked_accuracy (metrics.py:226-238): one wave per row: first-index arg max over V, compare, mask; integer counts.
__global__ __launch_bounds__(MT) void masked_accuracy_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                                             long rows, int V, long pad_id,
                                                             unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * (MT / 64) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (MT / 64);
    unsigned long long correct = 0, total = 0;
    for (long row = wave0; row < rows; row += nwaves) {
        const float* p = logits + (size_t)row * V;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int v = lane; v < V; v += 64) {
            const float x = p[v];
            if (x > bv) { bv = x; bi = v; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            const int64_t t = targets[row];
            if (t != pad_id) {
                ++total;
                if ((int64_t)bi == t) ++correct;
            }
        }
    }
    if (lane == 0 && (correct | total)) {
        atomicAdd(&out[0], correct);
        atomicAdd(&out[1], total);
    }
}

// One wave per row: keep the tokens before the first stop position (id == end_id, or id < 0 = the decode kernel's
// "row already finished" filler) whose id is not in drop[0..n_drop), in order.  64 positions per pass: ballot of the
// keep flags, rank = popcount of the lower lanes.
__global__ __launch_bounds__(MT) void compact_ids_kernel(const int32_t* __restrict__ ids, int rows, int width, int stride,
                                                         int end_id, const int* __restrict__ drop, int n_drop,
                                                         int32_t* __restrict__ out, int out_stride,
                                                         int32_t* __restrict__ out_len) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (MT / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int32_t* src = ids + (size_t)row * stride;
    int32_t* dst = out + (size_t)row * out_stride;
    int d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = j < n_drop ? drop[j] : -0x7fffffff;
    int n = 0;
    for (int base = 0; base < width; base += 64) {
        const int pos = base + lane;
        const int v = pos < width ? src[pos] : -1;
        const bool stop = pos >= width || v < 0 || v == end_id;
        const unsigned long long stops = __ballot(stop);
        const int first_stop = stops ? __ffsll((long long)stops) - 1 : 64;
        bool keep = lane < first_stop;
#pragma unroll
        for (int j = 0; j < 8; ++j) keep = keep && v != d[j];
        const unsigned long long km = __ballot(keep);
        if (keep) dst[n + __popcll(km & ((1ull << lane) - 1ull))] = v;
        n += __popcll(km);
        if (first_stop < 64) break;
    }
    if (lane == 0) out_len[row] = n;
}

size_t metrics_lds(int max_len) { return ((size_t)2 * max_len + 3 * ((size_t)max_len + 1) + 8) * sizeof(int); }

}  // namespace

extern "C" int i2l_sequence_metrics(const int32_t* pred, const int32_t* pred_len, int pred_stride, const int32_t* target,
                                    const int32_t* target_len, int target_stride, int pairs, int max_len, int max_n,
                                    int pad_id, int32_t* lev_out, int32_t* match_out, int32_t* tla_out,
                                    i2l_stream_t stream) {
    if (!pred || !pred_len || !target || !target_len || !lev_out || !match_out || pairs <= 0 || max_len < 0 || max_n < 1 ||
        max_n > 4 || pred_stride < max_len || target_stride < max_len)
        return I2L_ERR_ARG;
    if (max_len <= 256) {                                   // bit-vector Levenshtein: W = ceil(max_len / 64) blocks
        const size_t lds_bp = ((size_t)2 * max_len + 8) * sizeof(int);
        const int Wb = max_len <= 64 ? 1 : (max_len <= 128 ? 2 : (max_len <= 192 ? 3 : 4));
        hipStream_t sb = i2l_s(stream);
#define I2L_BP(W_) hipLaunchKernelGGL(sequence_metrics_bp_kernel<W_>, dim3(pairs), dim3(MT), lds_bp, sb, pred, pred_len, pred_stride, \
                                      target, target_len, target_stride, max_len, max_n, pad_id, lev_out, match_out, tla_out)
        if (Wb == 1) I2L_BP(1); else if (Wb == 2) I2L_BP(2); else if (Wb == 3) I2L_BP(3); else I2L_BP(4);
#undef I2L_BP
        I2L_CHECK_LAUNCH();
        return I2L_OK;
    }
    const size_t lds = metrics_lds(max_len);
    if (lds > 160 * 1024) return I2L_ERR_UNSUPPORTED;       // sequences longer than ~8k tokens
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(sequence_metrics_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return I2L_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sequence_metrics_kernel, dim3(pairs), dim3(MT), lds, i2l_s(stream), pred, pred_len, pred_stride, target,
                       target_len, target_stride, max_len, max_n, pad_id, lev_out, match_out, tla_out);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

// Host helper (no GPU work): the float64 tail of calculate_metrics (metrics.py:184-223) on the kernel's integer statistics.
// stats (pairs, stride) int32 rows = [lev, match 1..4, (2 unused), gen_len, true_len] as device_sequence_statistics packs
// them.  Every float64 operation is the reference's, in its order (metrics.py:85-94,113-181): one division per precision,
// logs added left to right from 0.0, one division by n, one exp, the brevity factor multiplied from the left, means as
// left-to-right sums -- libm's log / exp are the functions CPython's math module calls, so the scores are bit-identical to
// the Python formulas (tests) at a hundredth of their cost (0.5 ms per 256 pairs in the interpreter).
extern "C" int i2l_scores_from_statistics(const int32_t* stats, int pairs, int stride, int n, double* bleu_mean_out,
                                          double* lev_mean_out) {
    if (!stats || pairs <= 0 || stride < 9 || n < 1 || n > 4 || !bleu_mean_out || !lev_mean_out) return I2L_ERR_ARG;
    double bleu_sum = 0.0, lev_sum = 0.0;
    for (int i = 0; i < pairs; ++i) {
        const int32_t* r = stats + (size_t)i * stride;
        const int lev = r[0], gen_len = r[7], true_len = r[8];
        double bleu = 0.0;
        const int shorter = gen_len < true_len ? gen_len : true_len;
        if (shorter != 0) {
            double log_sum = 0.0;
            bool zero = false;
            for (int g = 1; g <= n; ++g) {
                const int hits = shorter >= g ? r[g] : 0;
                if (hits == 0) { zero = true; break; }
                log_sum += log((double)hits / (double)(gen_len - g + 1));
            }
            if (!zero) {
                const double score = exp(log_sum / (double)n);
                bleu = gen_len >= true_len ? score : exp(1.0 - (double)true_len / (double)gen_len) * score;
            }
        }
        bleu_sum += bleu;
        const int max_length = gen_len > true_len ? gen_len : true_len;
        lev_sum += max_length == 0 ? 1.0 : 1.0 - ((double)lev / (double)max_length);
    }
    *bleu_mean_out = bleu_sum / (double)pairs;
    *lev_mean_out = lev_sum / (double)pairs;
    return I2L_OK;
}

extern "C" int i2l_masked_accuracy(const float* logits, const int64_t* targets, int64_t rows, int vocab, int64_t pad_id,
                                   uint64_t* correct_total_out, i2l_stream_t stream) {
    if (!logits || !targets || !correct_total_out || rows <= 0 || vocab <= 0) return I2L_ERR_ARG;
    hipStream_t s = i2l_s(stream);
    if (hipMemsetAsync(correct_total_out, 0, 2 * sizeof(uint64_t), s) != hipSuccess) return I2L_ERR_LAUNCH;
    long blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(masked_accuracy_kernel, dim3((unsigned)blocks), dim3(MT), 0, s, logits, targets, (long)rows, vocab,
                       (long)pad_id, reinterpret_cast<unsigned long long*>(correct_total_out));
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

extern "C" int i2l_compact_ids(const int32_t* ids, int rows, int width, int stride, int end_id, const int32_t* drop_ids,
                               int n_drop, int32_t* out_ids, int out_stride, int32_t* out_len, i2l_stream_t stream) {
    if (!ids || !out_ids || !out_len || rows <= 0 || width <= 0 || stride < width || out_stride < width || n_drop < 0 ||
        n_drop > 8 || (n_drop > 0 && !drop_ids))
        return I2L_ERR_ARG;
    hipLaunchKernelGGL(compact_ids_kernel, dim3(i2l_cdiv(rows, MT / 64)), dim3(MT), 0, i2l_s(stream), ids, rows, width,
                       stride, end_id, drop_ids, n_drop, out_ids, out_stride, out_len);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
