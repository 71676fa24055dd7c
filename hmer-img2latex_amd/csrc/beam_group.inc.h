// Grouped beam search (included by decode.hip after decode_group.inc.h, inside its anonymous namespace).
//
// beam_kernel gives every image one workgroup that streams the 1.5 MB of recurrent and projection weights from
// L2 at every step (39 us per step at k = 5).  Here the machinery of the grouped greedy kernel is reused: FOUR
// workgroups (one per CU, 512 threads) keep one quarter of the weights each on chip for the whole search
//     member m:  WhhT[:, gates of hidden units 64m..64m+63]  in registers,   WoutT[:, columns 128m..+127]  in LDS
// and together run BG_S = 12 row slots = the K beams of IPG = 12 / K images (k = 5: two images, 256 workgroups for
// the 128 images of BASELINE configs[2]).  The slots are processed in three passes of four, so the per-thread
// tiles are the greedy kernel's (4 rows x 4 gates / columns) and the resident weights are reused by every pass.
//
// One step (reference seq2seq.py:234-298; the numbering follows beam_kernel's (a)-(f)):
//   A  recurrent part of the gates of the OLD slots (h . Whh^T): independent of the selection, so it runs while the
//      peers' candidates of the previous step are in flight; the results wait in registers
//   B  candidates of the previous step: per slot every member published the top K of its 128 columns and the
//      (max, sum exp) of them; the 4K candidates are merged (value desc, index asc = the order of torch.topk on
//      ties), log_softmax = (v - M) - log(sum_q s_q exp(m_q - M)); then per image the k x k candidates are ranked by
//      fp64 score with ties in (beam, rank) order (d), the history row is written, beams that ended retire (a/f)
//   C  new slot q continues parent slot par[q]: its gates are the parent's recurrent part (through LDS) + P[token]
//      + Genc[image], its cell state the parent's (e); LSTM cell for this member's 64 units
//   D  exchange of h (64 units x 12 slots per member), as in the greedy kernel
//   E  logits of this member's 128 columns for the 12 slots
//   F  per slot: local (max, sum exp) and top K -> published with tag step + 1
// All four members hold the same per-image state and take the same decisions from the same data; member (image & 3)
// writes the history and the result.  Exchange, placement check, progress argument and timeouts: decode_group.inc.h.
// A timed-out poll ends the group with len_out = -3 (i2l_beam_decode is then re-run with I2L_FLAG_NO_GROUP).

constexpr int BG_NP = 3;                         // passes of 4 row slots
constexpr int BG_S = 4 * BG_NP;                  // row slots per group
constexpr int BG_CW = I2L_MAX_BEAM + 2;          // granules per slot: K candidates, local max, local sum
constexpr int BG_GRAN_H = 64 * BG_S;             // h granules per member and step: [slot][unit]
constexpr int BG_GRAN_C = BG_GRAN_H;             // [slot][BG_CW]
constexpr int BG_GRAN_X = BG_GRAN_C + 128;       // placement granule, a line of its own
constexpr int BG_GRAN = BG_GRAN_X + 16;
constexpr size_t BEAM_XCHG_PER_GROUP = (size_t)2 * 4 * BG_GRAN * 8;

struct BeamImageState {                          // one per image of the group, in LDS, identical in all members
    double score[I2L_MAX_BEAM];
    double best_c;
    int last[I2L_MAX_BEAM];
    int live[I2L_MAX_BEAM];
    int nb, done, has_c, best_t, best_q, t_last, pad0, pad1;
};

constexpr size_t bg_lds_bytes() {
    return (size_t)(256 * 128) * 4            // wout_s4
           + (size_t)BG_NP * 256 * 4 * 4      // h_s
           + (size_t)BG_S * 64 * 16           // zs (gates of the old slots) / lg (logits), never live together
           + (size_t)BG_S * 64 * 4            // cs
           + (size_t)BG_S * BG_CW * 8         // own candidates
           + (size_t)BG_S * I2L_MAX_BEAM * 8  // topv, topi
           + (size_t)BG_S * 2 * 4             // par_s, tok_s
           + 16 * 4                           // flags
           + (size_t)BG_S * sizeof(BeamImageState);
}

struct BeamGroupParams {
    StepWeights w;
    int images, T, n_groups, start_id, end_id;
    int32_t* tokhist;
    int32_t* parhist;
    int32_t* seq_out;
    int32_t* len_out;
    double* score_out;
    u64_t* xchg;          // [n_groups][2][4][BG_GRAN]
    unsigned* status;
    GroupOpts opts;       // poll limits, exchange flavour (group_common.inc.h)
};

// all-reduce over the 64 lanes of a wave: 4 rotate steps inside each row of 16, then the 4 row results
__device__ __forceinline__ u64_t wave_umax64(u64_t k) {
    k = umax64(k, dpp_u64<0x128>(k));
    k = umax64(k, dpp_u64<0x124>(k));
    k = umax64(k, dpp_u64<0x122>(k));
    k = umax64(k, dpp_u64<0x121>(k));
    u64_t r = 0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, row * 16);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), row * 16);
        r = umax64(r, ((u64_t)hi << 32) | lo);
    }
    return r;
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = fmaxf(v, dpp_f<0x128>(v));
    v = fmaxf(v, dpp_f<0x124>(v));
    v = fmaxf(v, dpp_f<0x122>(v));
    v = fmaxf(v, dpp_f<0x121>(v));
    float r = -INFINITY;
#pragma unroll
    for (int row = 0; row < 4; ++row) r = fmaxf(r, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), row * 16)));
    return r;
}
__device__ __forceinline__ float wave_sum_f(float v) {      // fixed order: deterministic
    v += dpp_f<0x128>(v);
    v += dpp_f<0x124>(v);
    v += dpp_f<0x122>(v);
    v += dpp_f<0x121>(v);
    float r = 0.f;
#pragma unroll
    for (int row = 0; row < 4; ++row) r += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), row * 16));
    return r;
}
__device__ __forceinline__ u64_t readlane_u64(u64_t k, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), l);
    return ((u64_t)hi << 32) | lo;
}

// The per-lane index arithmetic of each phase is re-derived from this inside the step loop: hoisted out of the loop it
// would occupy dozens of registers for the whole search, next to the 128 that hold the weights, and spill.
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

template <int K>
__global__ __launch_bounds__(GNT) void beam_group_kernel(BeamGroupParams p) {
    constexpr int IPG = BG_S / K;                           // images per group
    constexpr int G = 1024;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* wout_s4 = reinterpret_cast<float4*>(smem);      // [16 j][512 threads]
    float* h_s = smem + 256 * 128;                          // [pass][256 k][4 slots]
    float4* zs = reinterpret_cast<float4*>(h_s + BG_NP * 256 * 4);   // [slot][64 units] (i, f, g, o) recurrent sums
    float* lg = reinterpret_cast<float*>(zs);               // [slot][128 columns]  (E -> F; zs is dead by then)
    float* cs = reinterpret_cast<float*>(zs + BG_S * 64);   // [slot][64 units] cell state
    u64_t* own_c = reinterpret_cast<u64_t*>(cs + BG_S * 64);         // [slot][BG_CW] this member's candidates
    float* topv = reinterpret_cast<float*>(own_c + BG_S * BG_CW);    // [slot][MAX_BEAM] log-probs, descending
    int* topi = reinterpret_cast<int*>(topv + BG_S * I2L_MAX_BEAM);  // [slot][MAX_BEAM]
    int* par_s = topi + BG_S * I2L_MAX_BEAM;                // [slot] parent slot (group numbering)
    int* tok_s = par_s + BG_S;                              // [slot] token fed at this step
    int* flg = tok_s + BG_S;                                // [0] poll timed out  [1] members share an XCD  [2] group done
    BeamImageState* ist = reinterpret_cast<BeamImageState*>(flg + 16);

    const StepWeights& w = p.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int within = blockIdx.x & 31;
    const int group = (blockIdx.x >> 5) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int T = p.T, V = w.V;
    const int img0 = group * IPG;
    const int ul = tid >> 3, ke = tid & 7;
    const int unit = 64 * m + ul;
    const int cq = tid >> 4, ks = tid & 15;
    const int l_v = 128 * m + 4 * cq + ((ks >> 2) & 1) * 2 + (ks >> 3);   // the column whose bias this lane adds in E

    f32x2 wreg[32][2];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float4 t4 = *reinterpret_cast<const float4*>(w.WhhT[0] + (size_t)(8 * j + ke) * G + 4 * unit);
        wreg[j][0] = f32x2{t4.x, t4.y};
        wreg[j][1] = f32x2{t4.z, t4.w};
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
        wout_s4[j * GNT + tid] = *reinterpret_cast<const float4*>(w.WoutT + (size_t)(16 * j + ks) * 512 + 128 * m + 4 * cq);
    for (int idx = tid; idx < BG_NP * 256 * 4; idx += GNT) h_s[idx] = 0.f;
    for (int idx = tid; idx < BG_S * 64; idx += GNT) cs[idx] = 0.f;
    const float l_bias = w.boutP[l_v];
    if (tid < BG_S) {
        const int q = tid, i = q / K, b = q - i * K;
        par_s[q] = q;
        tok_s[q] = (i < IPG && b == 0) ? min(max(p.start_id, 0), V - 1) : 0;
    }
    if (tid < IPG) {
        BeamImageState& s = ist[tid];
        for (int q = 0; q < I2L_MAX_BEAM; ++q) { s.score[q] = 0.0; s.last[q] = p.start_id; s.live[q] = q == 0 ? 1 : 0; }
        s.best_c = 0.0;
        s.nb = 1; s.has_c = 0; s.best_t = -1; s.best_q = 0; s.t_last = -1;
        s.done = (img0 + tid < p.images) ? 0 : 1;
    }
    if (tid == 0) { flg[0] = 0; flg[2] = 0; }
    u64_t* xg = p.xchg + (size_t)group * 2 * GQ * BG_GRAN;

    // placement (see decode_group_kernel)
    if (wave == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        if (lane == 0 && !(p.opts.drop_member && m == 3))      // test hook: member 3 stays silent, its peers time out
            store_granule(xg + (size_t)m * BG_GRAN + BG_GRAN_X, granule(0xC0DEu, __uint_as_float(xcc)), false);
        const int pq = (lane & 3) + ((lane & 3) >= m ? 1 : 0);
        u64_t pv = 0;
        bool bad = false;
        long long t_start = 0;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            if (lane < 3) { pv = load_granule(xg + (size_t)pq * BG_GRAN + BG_GRAN_X); ok = (unsigned)(pv >> 32) == 0xC0DEu; }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
                const long long now = (long long)wall_clock64();
                if (t_start == 0) t_start = now;
                else if (now - t_start > p.opts.limit_first) { bad = true; break; }
            }
        }
        const bool all_same = __all(lane >= 3 || (unsigned)pv == xcc);
        if (lane == 0) { flg[1] = (all_same && !bad) ? 1 : 0; if (bad) flg[0] = 1; }
    }
    __syncthreads();
    const bool local = flg[1] != 0 && !p.opts.agent_scope;

    float c_new[BG_NP];
#pragma unroll
    for (int pp = 0; pp < BG_NP; ++pp) c_new[pp] = 0.f;
    bool failed = false;
    int t = 0;
#ifdef I2L_GROUP_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long st_last = (long long)wall_clock64();
#define I2L_STAMP(i) do { const long long n_ = (long long)wall_clock64(); st_acc[i] += n_ - st_last; st_last = n_; } while (0)
#else
#define I2L_STAMP(i) do { } while (0)
#endif
    for (;; ++t) {
        // candidates of step t-1 (wave w merges slots w and w + 8): requested now, in flight during A
        //   lanes 0..4K-1: candidate j of member qm; lanes 32..35: local max of member lane-32; 36..39: local sum
        const int lane = opaque(tid) & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const bool b_isc = lane < 4 * K, b_ism = lane >= 32 && lane < 36, b_act = b_isc || (lane >= 32 && lane < 40);
        const int b_qm = b_isc ? lane / K : (lane & 3);
        const int b_j = b_isc ? lane - b_qm * K : (b_ism ? K : K + 1);
        const u64_t* b_src = xg + (size_t)((t - 1) & 1) * GQ * BG_GRAN + (size_t)b_qm * BG_GRAN + BG_GRAN_C + b_j;
        u64_t b_gv[2] = {0, 0};
        auto b_fetch = [&]() {                               // peers publish about when this member does: not too early
            if (t > 0 && b_act) {
                b_gv[0] = load_granule(b_src + wave * BG_CW);
                if (wave + 8 < BG_S) b_gv[1] = load_granule(b_src + (wave + 8) * BG_CW);
            }
        };
        // ---- A. recurrent part of the gates of the old slots
        float zr[BG_NP][2];                                 // lanes ke < 4: (i, f) of slot 4p + ke; ke >= 4: (g, o) of slot 4p + ke - 4
#pragma unroll
        for (int pp = 0; pp < BG_NP; ++pp) { zr[pp][0] = 0.f; zr[pp][1] = 0.f; }
        if (t > 0 && t < T) {
            const int ke = opaque(tid) & 7;
#pragma unroll
            for (int pp = 0; pp < BG_NP; ++pp) {
                f32x2 acc[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) { acc[g][0] = splat2(0.f); acc[g][1] = splat2(0.f); }
                const float4* hq4 = reinterpret_cast<const float4*>(h_s) + pp * 256 + ke;
                float4 hb[2][4];                            // h is read 4 k ahead of its use (two register sets)
#pragma unroll
                for (int i = 0; i < 4; ++i) hb[0][i] = hq4[8 * i];
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    if (b + 1 < 8) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) hb[(b + 1) & 1][i] = hq4[8 * ((b + 1) * 4 + i)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) fma_4x4(acc, wreg[b * 4 + i][0], wreg[b * 4 + i][1], hb[b & 1][i]);
                }
                const bool b0 = ke & 1, b1 = ke & 2, b2 = ke & 4;
                float z[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float wv[2];
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        const float ux = rs_level<DPP_HMIRROR>(acc[e][rp].x, acc[2 + e][rp].x, b2);
                        const float uy = rs_level<DPP_HMIRROR>(acc[e][rp].y, acc[2 + e][rp].y, b2);
                        wv[rp] = rs_level<DPP_XOR1>(ux, uy, b0);
                    }
                    z[e] = rs_level<DPP_XOR2>(wv[0], wv[1], b1);
                }
                zr[pp][0] = z[0]; zr[pp][1] = z[1];
                __builtin_amdgcn_sched_barrier(0);
                if (pp == 1) b_fetch();
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            b_fetch();
        }
        I2L_STAMP(0);
        // ---- B. candidates of step t-1 -> per-slot top K with log-probs, per-image selection, bookkeeping
        if (t > 0) {
            const unsigned epoch = (unsigned)t;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int slot = wave + 8 * half;
                if (slot >= BG_S) break;
                u64_t gv = b_gv[half];
                bool bad = false;
                long long t_start = 0;
                unsigned spins = 0;
                for (;;) {
                    const bool ok = !b_act || (unsigned)(gv >> 48) == epoch;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0) {
                        const long long now = (long long)wall_clock64();
                        if (t_start == 0) t_start = now;
                        else if (now - t_start > p.opts.limit_step) { bad = true; break; }
                    }
                    if (b_act) {
                        gv = load_granule(b_src + slot * BG_CW);
                        if (half == 0 && wave + 8 < BG_S) b_gv[1] = load_granule(b_src + (wave + 8) * BG_CW);
                    }
                }
                if (bad) { if (lane == 0) flg[0] = 1; continue; }
                const float val = __uint_as_float((unsigned)gv);
                const int cidx = (int)((unsigned)(gv >> 32) & 0xFFFFu);
                float mq[4], sq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    mq[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), 32 + q));
                    sq[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), 36 + q));
                }
                const float M = fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3]));
                float S = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) S += sq[q] > 0.f ? sq[q] * expf(mq[q] - M) : 0.f;
                const float lse = logf(S);
                const bool valid = b_isc && cidx != 0xFFFF;
                const u64_t key = valid ? am_key(val, cidx) : 0ull;
                int rank = 0;
#pragma unroll
                for (int o = 0; o < 4 * K; ++o) rank += readlane_u64(key, o) > key ? 1 : 0;
                if (valid && rank < K) {
                    topv[slot * I2L_MAX_BEAM + rank] = (val - M) - lse;
                    topi[slot * I2L_MAX_BEAM + rank] = cidx;
                }
            }
            __syncthreads();
            I2L_STAMP(1);
            if (flg[0]) { failed = true; break; }
            // the recurrent sums of the old slots go to LDS now: zs aliases lg, and every wave's F(t-1) lies before
            // the barrier above
            {
                const int tz = opaque(tid), ke = tz & 7, ul = tz >> 3;
                float2* zs2 = reinterpret_cast<float2*>(zs);
#pragma unroll
                for (int pp = 0; pp < BG_NP; ++pp)
                    zs2[((4 * pp + (ke & 3)) * 64 + ul) * 2 + (ke >> 2)] = make_float2(zr[pp][0], zr[pp][1]);
            }
            // (d) wave i ranks the k x k candidates of image i, then (f)/(a) in the same wave: LDS operations of one
            // wave execute in order, so the lanes read back what the winners wrote without a workgroup barrier
            if (wave < IPG && !ist[wave].done) {
                BeamImageState& s = ist[wave];
                const int nb = s.nb;
                const int lq = lane < K ? lane : 0;
                const bool was_live = lane < K && lane < nb && s.live[lq];
                const int nlive = __popcll(__ballot(was_live));
                const int nnew = nlive * K < K ? nlive * K : K;
                const int s0 = lane / K, jj = lane - s0 * K;
                const int s0c = s0 < K ? s0 : 0;
                const bool cand = lane < K * K && s0 < nb && s.live[s0c];
                const int slot = wave * K + s0c;
                const double sc = cand ? s.score[s0c] + (double)topv[slot * I2L_MAX_BEAM + jj] : 0.0;
                const int ctok = cand ? topi[slot * I2L_MAX_BEAM + jj] : 0;
                const unsigned long long cmask = __ballot(cand);
                int rank = 0;
#pragma unroll
                for (int o = 0; o < K * K; ++o) {
                    const double so = __longlong_as_double((long long)readlane_u64((u64_t)__double_as_longlong(sc), o));
                    const bool oc = (cmask >> o) & 1ull;
                    rank += (oc && (so > sc || (so == sc && o < lane))) ? 1 : 0;
                }
                const bool win = cand && rank < nnew;
                const bool all_end = __ballot(win && ctok != p.end_id) == 0ull;
                if (win) {
                    s.score[rank] = sc;
                    s.last[rank] = ctok;
                    par_s[wave * K + rank] = wave * K + s0;
                    tok_s[wave * K + rank] = ctok;
                    if (m == ((img0 + wave) & 3)) {
                        const size_t hrow = ((size_t)(img0 + wave) * T + (t - 1)) * K + rank;
                        p.tokhist[hrow] = ctok;
                        p.parhist[hrow] = s0;
                    }
                }
                if (lane >= nnew && lane < K) {
                    s.score[lane] = 0.0; s.last[lane] = 0;
                    par_s[wave * K + lane] = wave * K; tok_s[wave * K + lane] = 0;
                }
                // (f) every new beam ended -> completed.extend(beams), stop;  else (a) retirement for step t.
                // Lane q looks at beam q; the reference's serial "score > best" scan keeps the FIRST maximum.
                const double myscore = s.score[lq];
                const int mylast = s.last[lq];
                const bool mine = lane < nnew;
                const bool ended = mine && (all_end || mylast == p.end_id);
                const bool consider = ended && (all_end || t < T);
                const unsigned long long emask = __ballot(consider);
                const bool lives = mine && !ended;
                const int nl = __popcll(__ballot(lives));
                if (lane == 0) {
                    int has_c = s.has_c, best_t = s.best_t, best_q = s.best_q;
                    double best_c = s.best_c;
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        const double sq_ = __longlong_as_double((long long)readlane_u64((u64_t)__double_as_longlong(myscore), q));
                        if (((emask >> q) & 1ull) && (!has_c || sq_ > best_c)) { has_c = 1; best_c = sq_; best_t = t - 1; best_q = q; }
                    }
                    s.has_c = has_c; s.best_c = best_c; s.best_t = best_t; s.best_q = best_q;
                    s.nb = nnew; s.t_last = t - 1;
                    if (all_end || (t < T && nl == 0)) s.done = 1;
                }
                if (lane < K && !all_end && t < T) s.live[lane] = lives ? 1 : 0;
            }
            __syncthreads();
            bool all_done = true;
#pragma unroll
            for (int i = 0; i < IPG; ++i) all_done = all_done && ist[i].done != 0;
            if (all_done || t == T) break;
        }
        I2L_STAMP(2);
        // ---- C. LSTM cell of this member's 64 units for the new slots (lanes ke < 4: slots 4p + ke)
        const unsigned epoch = (unsigned)t + 1u;
        u64_t* xslot = xg + (size_t)(t & 1) * GQ * BG_GRAN;
        const int tc = opaque(tid), ke = tc & 7, ul = tc >> 3, unit = 64 * m + ul;
        if (ke < 4) {
            float4 pv[BG_NP], ge[BG_NP];
#pragma unroll
            for (int pp = 0; pp < BG_NP; ++pp) {                // the global loads first: one L2 round trip, not three
                const int q = 4 * pp + ke;
                const int tk = min(max(tok_s[q], 0), V - 1);
                const int img = min(img0 + min(q / K, IPG - 1), p.images - 1);
                pv[pp] = *reinterpret_cast<const float4*>(w.P + (size_t)tk * G + 4 * unit);
                ge[pp] = *reinterpret_cast<const float4*>(w.Genc + (size_t)img * G + 4 * unit);
            }
#pragma unroll
            for (int pp = 0; pp < BG_NP; ++pp) {
                const int q = 4 * pp + ke;
                const int ps = par_s[q];
                const float4 zz = t > 0 ? zs[ps * 64 + ul] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float cp = cs[ps * 64 + ul];
                const float xi = (zz.x + ge[pp].x) + pv[pp].x, xf = (zz.y + ge[pp].y) + pv[pp].y;
                const float xc = (zz.z + ge[pp].z) + pv[pp].z, xo = (zz.w + ge[pp].w) + pv[pp].w;
                const float ig = sigmoidf_(xi), fg = sigmoidf_(xf), gg = tanhf_(xc), og = sigmoidf_(xo);
                c_new[pp] = fg * cp + ig * gg;
                const float hn = og * tanhf_(c_new[pp]);
                store_granule(xslot + (size_t)m * BG_GRAN + q * 64 + ul, granule(epoch, hn), local);
                h_s[pp * 1024 + (64 * m + ul) * 4 + ke] = hn;   // h_s was last read in A, before the barriers of B
            }
        }
        I2L_STAMP(3);
        // ---- D. the other three quarters of h: 3 x 768 granules over 512 threads
        {
            constexpr int NG = 3 * BG_GRAN_H, PER = (NG + GNT - 1) / GNT;
            u64_t gr[PER];
            const int td = opaque(tid);
            auto gsrc = [&](int i) {
                const int idx = min(td + i * GNT, NG - 1);
                const int pi = idx / BG_GRAN_H, gi = idx - pi * BG_GRAN_H;
                return xslot + (size_t)(pi + (pi >= m ? 1 : 0)) * BG_GRAN + gi;
            };
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < PER; ++i) { gr[i] = load_granule(gsrc(i)); ok = ok && (unsigned)(gr[i] >> 32) == epoch; }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > p.opts.limit_step) { failed = true; break; }
                }
            }
            I2L_STAMP(4);
            // h_s was last read in A of this step, before the barriers of B (t > 0) or the start-up barrier (t = 0)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = min(td + i * GNT, NG - 1);
                const int pi = idx / BG_GRAN_H, gi = idx - pi * BG_GRAN_H;
                const int peer = pi + (pi >= m ? 1 : 0);
                h_s[(gi >> 8) * 1024 + (64 * peer + (gi & 63)) * 4 + ((gi >> 6) & 3)] = __uint_as_float((unsigned)gr[i]);
            }
        }
        if (failed) { flg[0] = 1; failed = false; }
        __syncthreads();
        I2L_STAMP(5);
        // every lane has read its parents' cell state (C, before the barrier): the new one may replace it
        if (ke < 4) {
#pragma unroll
            for (int pp = 0; pp < BG_NP; ++pp) cs[(4 * pp + ke) * 64 + ul] = c_new[pp];
        }
        // ---- E. logits of this member's 128 columns, pass by pass; the lane ends with the logit of (l_col, l_row)
        const int te = opaque(tid), cq = te >> 4, ks = te & 15;
        const int l_row = ks & 3, l_col = ((ks >> 2) & 1) * 2 + (ks >> 3);
#pragma unroll
        for (int pp = 0; pp < BG_NP; ++pp) {
            f32x2 pa[4][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) { pa[c][0] = splat2(0.f); pa[c][1] = splat2(0.f); }
            const float4* hq4 = reinterpret_cast<const float4*>(h_s) + pp * 256 + ks;
            float4 wb[2][2], hb[2][2];
            auto fetch = [&](int set, int b) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wb[set][i] = wout_s4[(b * 2 + i) * GNT + te];
                    hb[set][i] = hq4[16 * (b * 2 + i)];
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) fetch((b + 1) & 1, b + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 w4 = wb[b & 1][i];
                    fma_4x4(pa, f32x2{w4.x, w4.y}, f32x2{w4.z, w4.w}, hb[b & 1][i]);
                }
            }
            const bool b0 = ks & 1, b1 = ks & 2, b2 = ks & 4, b3 = ks & 8;
            float z[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float wv[2];
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const float ux = rs_level<DPP_HMIRROR>(pa[e][rp].x, pa[2 + e][rp].x, b2);
                    const float uy = rs_level<DPP_HMIRROR>(pa[e][rp].y, pa[2 + e][rp].y, b2);
                    wv[rp] = rs_level<DPP_XOR1>(ux, uy, b0);
                }
                z[e] = rs_level<DPP_XOR2>(wv[0], wv[1], b1);
            }
            const float lv = rs_level<DPP_ROR8>(z[0], z[1], b3) + l_bias;
            lg[(4 * pp + l_row) * 128 + 4 * cq + l_col] = lv;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        I2L_STAMP(6);
        // ---- F. per slot: local max, sum of exp, top K of the 128 columns; a row of 16 lanes per slot (8 columns per
        //         lane), so every reduction is four DPP rotations inside the row; published with tag t + 1
        if (wave < BG_S / 4) {                               // waves 0..2, the others go on to the next step's A
            const int slot = wave * 4 + (lane >> 4), l16 = lane & 15;
            float x[8];
            float mloc = -INFINITY;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x[i] = lg[slot * 128 + l16 + 16 * i];
                if (128 * m + l16 + 16 * i >= V) x[i] = -INFINITY;
                mloc = fmaxf(mloc, x[i]);
            }
            mloc = fmaxf(mloc, dpp_f<0x128>(mloc));
            mloc = fmaxf(mloc, dpp_f<0x124>(mloc));
            mloc = fmaxf(mloc, dpp_f<0x122>(mloc));
            mloc = fmaxf(mloc, dpp_f<0x121>(mloc));
            float sloc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sloc += x[i] > -INFINITY ? expf(x[i] - mloc) : 0.f;
            sloc += dpp_f<0x128>(sloc);
            sloc += dpp_f<0x124>(sloc);
            sloc += dpp_f<0x122>(sloc);
            sloc += dpp_f<0x121>(sloc);
            // K rounds of (largest value, then lowest column holding it) on order-preserving unsigned images of the
            // values (0 = excluded or already taken): integer maxima / minima over the row, no NaN canonicalisation
            unsigned o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned u = __float_as_uint(x[i] + 0.0f);   // -0 -> +0
                u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
                o[i] = x[i] > -INFINITY ? u : 0u;
            }
            unsigned myo = 0;
            int myi = 0xFFFF;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                unsigned bo = max(max(max(o[0], o[1]), max(o[2], o[3])), max(max(o[4], o[5]), max(o[6], o[7])));
                bo = max(bo, (unsigned)__builtin_amdgcn_mov_dpp((int)bo, 0x128, 0xF, 0xF, true));
                bo = max(bo, (unsigned)__builtin_amdgcn_mov_dpp((int)bo, 0x124, 0xF, 0xF, true));
                bo = max(bo, (unsigned)__builtin_amdgcn_mov_dpp((int)bo, 0x122, 0xF, 0xF, true));
                bo = max(bo, (unsigned)__builtin_amdgcn_mov_dpp((int)bo, 0x121, 0xF, 0xF, true));
                int bi = 0xFFFF;                             // local column 16 i + l16
#pragma unroll
                for (int i = 7; i >= 0; --i) bi = o[i] == bo ? 16 * i + l16 : bi;
                bi = bo != 0u ? bi : 0xFFFF;
                bi = min(bi, __builtin_amdgcn_mov_dpp(bi, 0x128, 0xF, 0xF, true));
                bi = min(bi, __builtin_amdgcn_mov_dpp(bi, 0x124, 0xF, 0xF, true));
                bi = min(bi, __builtin_amdgcn_mov_dpp(bi, 0x122, 0xF, 0xF, true));
                bi = min(bi, __builtin_amdgcn_mov_dpp(bi, 0x121, 0xF, 0xF, true));
                if (l16 == j) { myo = bo; myi = bi; }
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (bi == 16 * i + l16) ? 0u : o[i];
            }
            const unsigned myu = myo ^ ((myo >> 31) ? 0x80000000u : 0xFFFFFFFFu);
            const float myv = __uint_as_float(myu);
            u64_t out = 0;
            if (l16 < K)
                out = myi != 0xFFFF ? granule((epoch << 16) | (unsigned)(128 * m + myi), myv)
                                    : granule((epoch << 16) | 0xFFFFu, -INFINITY);
            else if (l16 == K) out = granule(epoch << 16, mloc);
            else if (l16 == K + 1) out = granule(epoch << 16, sloc);
            if (l16 < K + 2)
                store_granule(xslot + (size_t)m * BG_GRAN + BG_GRAN_C + slot * BG_CW + l16, out, local);
        }
        I2L_STAMP(7);
    }
#ifdef I2L_GROUP_STAMPS
    if (tid == 0 && blockIdx.x < 32) {
        for (int i = 0; i < 8; ++i) p.status[8 + blockIdx.x * 8 + i] = (unsigned)st_acc[i];
        if (blockIdx.x == 0) p.status[4] = (unsigned)t;
    }
#endif
#undef I2L_STAMP
    if (failed) {
        if (tid == 0) atomicOr(p.status, 1u);
        if (tid < IPG && img0 + tid < p.images && m == ((img0 + tid) & 3)) p.len_out[img0 + tid] = -3;
        return;
    }
    // result: max(completed) (first on ties) else beams[0]; strip START, cut at END (seq2seq.py:286-297)
    if (tid < IPG && img0 + tid < p.images && m == ((img0 + tid) & 3)) {
        const BeamImageState& s = ist[tid];
        const int img = img0 + tid;
        int tt = s.has_c ? s.best_t : s.t_last, q = s.has_c ? s.best_q : 0;
        const double sc = s.has_c ? s.best_c : s.score[0];
        int32_t* out = p.seq_out + (size_t)img * (T + 1);
        const int32_t* th = p.tokhist + (size_t)img * T * K;
        const int32_t* ph = p.parhist + (size_t)img * T * K;
        const int n = tt + 1;
        for (int pos = tt; pos >= 0; --pos) {                // history rows were written by other lanes: L1 bypassed
            out[pos] = __hip_atomic_load(th + (size_t)pos * K + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q = __hip_atomic_load(ph + (size_t)pos * K + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int len = n;
        for (int i = 0; i < n; ++i)
            if (out[i] == p.end_id) { len = i; break; }
        for (int i = len; i < T + 1; ++i) out[i] = -1;
        p.len_out[img] = len;
        if (p.score_out) p.score_out[img] = sc;
    }
}
