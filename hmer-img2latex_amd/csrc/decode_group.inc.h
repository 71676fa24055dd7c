// Grouped greedy decode (included by decode.hip inside its anonymous namespace).
//
// The row-per-workgroup kernel streams ~1.1 MB of weights from L2 into every CU at every step (the L2 -> CU path,
// 64 B/clk/CU, bounds it: ~11.5 us/step).  Here FOUR workgroups (one per CU) decode FOUR batch rows together and
// each keeps one QUARTER of the weights on chip for the whole loop, so nothing is streamed:
//     member m holds  WhhT[:, gate columns of hidden units 64m..64m+63]   256 x 256 floats in registers
//                     WoutT[:, vocabulary columns 128m..128m+127]          256 x 128 floats in LDS
// Per step member m computes, for all 4 rows, the gates / LSTM cell of ITS 64 hidden units and the logits of ITS 128
// vocabulary columns; both need the full h of the group, so a step has two small exchanges inside the group:
//     1. every member publishes its 64 x 4 new h values and reads the other three quarters;
//     2. every member publishes, per row, the (max, first index) of its vocabulary quarter; all four combine the four
//        candidates in member (= index) order, so they agree on the token without further talk.
// The recurrent product h . Whh^T of the NEXT step does not depend on the token (only the table row P[token] that is
// added at its end does), so it runs between publishing the candidates and polling for them: exchange 2 is hidden.
//
// Exchange = 8-byte {value, tag} granules, tag = step + 1, each written by ONE agent-scope relaxed atomic store (sc1,
// write-through) and polled by agent-scope relaxed atomic loads (sc1: never served from this CU's L1 or a stale L2
// line): the data is its own flag, no fences, no dependence on XCD placement.  Two buffers by step parity suffice:
// a member overwrites a slot two steps later, after it has consumed from every peer data that the peer published
// after reading that slot.  The granule block is zeroed by a memset node before every launch.
//
// Progress: the members of a group are workgroups 32a + x + 8k (k = 0..3) -- one XCD under round-robin placement
// (speed only) -- and workgroups are dispatched in index order, so whenever a member is resident every lower
// index has been dispatched and some complete group can always run to its end and free its CUs.  Every poll is
// bounded by a wall-clock limit: on expiry the workgroup raises status[0], fills its ids with -3 and exits, its
// peers follow by their own limits, and the host wrapper reports the failure; the GPU is never left spinning.
typedef unsigned long long u64_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int GQ = 4;                    // workgroups (= rows) per group
constexpr int GRAN_H = 256;              // h granules per member and step: [unit 64][row 4]
constexpr int GRAN = GRAN_H + 8;         // + 4 candidate granules (one per row) + 4 unused
constexpr int GRP_HS = 65 * 4;           // floats per quarter of h_s (65: the 4 quarters sit on disjoint banks)
constexpr int GRP_WLD = 144;             // row stride of wout_s: rows k, k+1, k+2, k+3 sit on disjoint banks
constexpr size_t GRP_LDS = (size_t)(256 * GRP_WLD + 4 * GRP_HS + 2 * 128 + 8) * sizeof(float);
constexpr long long GRP_TIMEOUT_TICKS = 300000000ll;   // 3 s of the 100 MHz wall clock

struct GroupParams {
    StepWeights w;
    int B, T, n_groups;
    const int32_t* tok0;
    const int32_t* forced;
    int32_t* ids;
    float* logits;
    float temperature;
    int use_temp, stop, end_id;
    u64_t* xchg;          // [n_groups][2][GQ][GRAN]
    unsigned* status;     // [0] != 0: a poll timed out
};

__device__ __forceinline__ void store_granule(u64_t* g, unsigned epoch, float v) {
    __hip_atomic_store(g, ((u64_t)epoch << 32) | (u64_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64_t load_granule(const u64_t* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float dpp_quad_sum(float v) {
    // all four lanes of a quad end with ((q0+q1)+(q2+q3)) -- the same bits in every lane
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ float dpp_oct_sum(float v) {      // 8 consecutive lanes; the same bits in all of them
    v = dpp_quad_sum(v);
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    return v;
}
__device__ __forceinline__ f32x2 splat2(float x) { return f32x2{x, x}; }
// (value, index) as one unsigned 64-bit key whose order is "larger value first, then smaller index": arg max with
// first-index ties becomes a branch-free integer max
__device__ __forceinline__ u64_t am_key(float v, int i) {
    unsigned u = __float_as_uint(v + 0.0f);                 // -0 -> +0
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((u64_t)u << 32) | (u64_t)(0xFFFFFFFFu - (unsigned)i);
}
__device__ __forceinline__ float am_val(u64_t k) {
    unsigned u = (unsigned)(k >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}
__device__ __forceinline__ int am_idx(u64_t k) { return (int)(0xFFFFFFFFu - (unsigned)k); }
__device__ __forceinline__ u64_t umax64(u64_t a, u64_t b) { return a > b ? a : b; }
template <int CTRL>
__device__ __forceinline__ u64_t dpp_u64(u64_t k) {
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)k, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(k >> 32), CTRL, 0xF, 0xF, true);
    return ((u64_t)hi << 32) | lo;
}
// acc(2 rows) += w.x * h(2 rows) / w.y * h: v_pk_fma_f32 with the scalar picked by op_sel, so a weight PAIR
// occupies one register pair (the compiler's own splat {w, w} would double the resident weights)
__device__ __forceinline__ void pkfma_lo(f32x2& acc, f32x2 wpair, f32x2 h) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(wpair), "v"(h));
}
__device__ __forceinline__ void pkfma_hi(f32x2& acc, f32x2 wpair, f32x2 h) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(wpair), "v"(h));
}

__global__ __launch_bounds__(NT) void decode_group_kernel(GroupParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wout_s = smem;                                   // [256][144] this member's 128 columns of WoutT
    float* h_s = wout_s + 256 * GRP_WLD;                    // [4][65][4] full h of the last step: [quarter][unit][row]
    u64_t* redk = reinterpret_cast<u64_t*>(h_s + 4 * GRP_HS);   // [2][4 rows][16 = wave x lane row] arg-max keys
    int* tok_s = reinterpret_cast<int*>(redk + 128);       // [0..3] tokens of the last step (-1: timed out), [4] h poll timed out                                 // [4] tokens, [4] = failure flag

    const StepWeights& w = p.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int within = blockIdx.x & 31;
    const int group = (blockIdx.x >> 5) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int B = p.B, T = p.T, V = w.V;
    const int row0 = group * GQ;
    const int ul = tid >> 2, kq = tid & 3;                  // gates: hidden unit 64m+ul, k in [64kq, 64kq+64); cell: row kq
    const int unit = 64 * m + ul;
    const int cq = tid >> 3, ke = tid & 7;                  // logits: columns 128m + 4cq.., k = ke (mod 8); row ke & 3
    constexpr int G = 1024;                                 // 4 * H

    f32x2 wreg[64][2];                                      // WhhT[64kq + kk][4 unit .. +3] as (i,f), (g,o)
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) {
        const float4 t4 = *reinterpret_cast<const float4*>(w.WhhT[0] + (size_t)(64 * kq + kk) * G + 4 * unit);
        wreg[kk][0] = f32x2{t4.x, t4.y};
        wreg[kk][1] = f32x2{t4.z, t4.w};
    }
    for (int idx = tid; idx < 256 * 32; idx += NT) {
        const int k = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<float4*>(wout_s + k * GRP_WLD + 4 * c4) =
            *reinterpret_cast<const float4*>(w.WoutT + (size_t)k * 512 + 128 * m + 4 * c4);
    }
    for (int idx = tid; idx < 4 * GRP_HS; idx += NT) h_s[idx] = 0.f;
    if (tid < 8) tok_s[tid] = 0;
    const float4 genc = *reinterpret_cast<const float4*>(w.Genc + (size_t)min(row0 + kq, B - 1) * G + 4 * unit);
    const float2 bo = *reinterpret_cast<const float2*>(w.boutP + 128 * m + 4 * cq + 2 * (ke >> 2));
    float c_own = 0.f, h_own = 0.f;
    int tok[GQ];
    unsigned fin = 0;                                       // bit r: row r has emitted END (or does not exist)
#pragma unroll
    for (int r = 0; r < GQ; ++r) {
        tok[r] = min(max(p.tok0[min(row0 + r, B - 1)], 0), V - 1);
        if (row0 + r >= B) fin |= 1u << r;
    }
    u64_t* xg = p.xchg + (size_t)group * 2 * GQ * GRAN;
    const bool own_row = row0 + m < B;                      // this member writes the ids of row m
    int32_t* ids_row = (p.ids && own_row) ? p.ids + (size_t)(row0 + m) * T : nullptr;
    float* lrow = (p.logits && row0 + (ke & 3) < B) ? p.logits + (size_t)(row0 + (ke & 3)) * T * V : nullptr;
    u64_t cand_k = 0;                                       // wave 0, lanes 0..3: this member's candidate of row = lane
    __syncthreads();

    int t = 0;
    bool failed = false;
#ifdef I2L_GROUP_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long st_last = (long long)wall_clock64();
#define I2L_STAMP(i) do { const long long n_ = (long long)wall_clock64(); st_acc[i] += n_ - st_last; st_last = n_; } while (0)
#else
#define I2L_STAMP(i) do { } while (0)
#endif
    for (;; ++t) {
        // ---- A. token-independent part of the gates of step t: sum_k h[k] Whh[k][.]; thread = (unit, k-quarter)
        f32x2 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) { acc[g][0] = splat2(0.f); acc[g][1] = splat2(0.f); }
        u64_t gv = 0;                                       // wave 0, lanes 0..11: candidate granule (peer, row)
        const int c_qi = lane >> 2;
        const u64_t* cand_src = xg + (size_t)((t - 1) & 1) * GQ * GRAN + (size_t)(c_qi + (c_qi >= m ? 1 : 0)) * GRAN +
                                GRAN_H + (lane & 3);
        const bool c_poll = wave == 0 && lane < 12 && t > 0;
        if (t > 0 && t < T) {
            // h is read 4 k ahead of its use (two register sets): the LDS latency hides behind 32 FMAs
            const float4* hq4 = reinterpret_cast<const float4*>(h_s + kq * GRP_HS);
            float4 hb[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hb[0][i] = hq4[i];
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                if (b + 1 < 16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hb[(b + 1) & 1][i] = hq4[(b + 1) * 4 + i];
                }
                // the peers published their candidates about when this member did: fetch them mid-way, use them after
                if (b == 9 && c_poll) gv = load_granule(cand_src);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kk = b * 4 + i;
                    const float4 hv = hb[b & 1][i];
                    const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
                    pkfma_lo(acc[0][0], wreg[kk][0], h01); pkfma_lo(acc[0][1], wreg[kk][0], h23);
                    pkfma_hi(acc[1][0], wreg[kk][0], h01); pkfma_hi(acc[1][1], wreg[kk][0], h23);
                    pkfma_lo(acc[2][0], wreg[kk][1], h01); pkfma_lo(acc[2][1], wreg[kk][1], h23);
                    pkfma_hi(acc[3][0], wreg[kk][1], h01); pkfma_hi(acc[3][1], wreg[kk][1], h23);
                }
            }
        } else if (c_poll) {
            gv = load_granule(cand_src);
        }
        I2L_STAMP(0);
        // ---- B. tokens chosen at step t-1: combine the four members' candidates (wave 0), tell the workgroup
        if (t > 0) {
            const unsigned epoch = (unsigned)t;             // candidates of step t-1 carry tag t
            if (wave == 0) {
                bool bad = false;
                long long t_start = 0;
                unsigned spins = 0;
                for (;;) {
                    const bool ok = lane >= 12 || (unsigned)(gv >> 48) == epoch;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0) {
                        const long long now = (long long)wall_clock64();
                        if (t_start == 0) t_start = now;
                        else if (now - t_start > GRP_TIMEOUT_TICKS) { bad = true; break; }
                    }
                    if (lane < 12) gv = load_granule(cand_src);
                }
                // lane r < 4 gathers the three peers' granules of row r (lanes r, r+4, r+8: row_shl 4 / 8)
                const int g_lo = (int)(unsigned)gv, g_hi = (int)(unsigned)(gv >> 32);
                const int p_lo[3] = {g_lo, __builtin_amdgcn_mov_dpp(g_lo, 0x104, 0xF, 0xF, true),
                                     __builtin_amdgcn_mov_dpp(g_lo, 0x108, 0xF, 0xF, true)};
                const int p_hi[3] = {g_hi, __builtin_amdgcn_mov_dpp(g_hi, 0x104, 0xF, 0xF, true),
                                     __builtin_amdgcn_mov_dpp(g_hi, 0x108, 0xF, 0xF, true)};
                // the members own disjoint index ranges, so the order of the max does not matter
                u64_t best = cand_k;
#pragma unroll
                for (int j = 0; j < 3; ++j) best = umax64(best, am_key(__int_as_float(p_lo[j]), p_hi[j] & 0xFFFF));
                const int bi = am_idx(best);
                if (lane < 4) tok_s[lane] = bad ? -1 : bi;
            }
            __syncthreads();
            const int4 tk4 = *reinterpret_cast<const int4*>(tok_s);
            const int4 fl4 = *reinterpret_cast<const int4*>(tok_s + 4);
            if (tk4.x < 0 || fl4.x != 0) { failed = true; break; }
            const int tk[4] = {tk4.x, tk4.y, tk4.z, tk4.w};
            bool all_fin = true;
#pragma unroll
            for (int r = 0; r < GQ; ++r) {
                const int bi = tk[r];
                const int sel = bi < V ? bi : 0;
                const bool was_fin = (fin >> r) & 1u;
                if (r == m && tid == 0 && ids_row) ids_row[t - 1] = (p.stop == I2L_STOP_STICKY && was_fin) ? -1 : sel;
                tok[r] = sel;
                if (sel == p.end_id) fin |= 1u << r;
                all_fin = all_fin && ((fin >> r) & 1u);
            }
            if (t == T || (p.stop == I2L_STOP_STICKY && all_fin)) break;
        }
        I2L_STAMP(1);
        if (p.forced) {
#pragma unroll
            for (int r = 0; r < GQ; ++r) tok[r] = min(max(p.forced[(size_t)min(row0 + r, B - 1) * T + t], 0), V - 1);
        }
        // ---- C. LSTM cell of (unit, row kq), publish the new h
        const unsigned epoch = (unsigned)t + 1u;
        u64_t* slot = xg + (size_t)(t & 1) * GQ * GRAN;
        {
            const int mytok = kq == 0 ? tok[0] : (kq == 1 ? tok[1] : (kq == 2 ? tok[2] : tok[3]));
            const float4 pv = *reinterpret_cast<const float4*>(w.P + (size_t)mytok * G + 4 * unit);
            float gs[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float s0 = dpp_quad_sum(acc[g][0].x), s1 = dpp_quad_sum(acc[g][0].y);
                const float s2 = dpp_quad_sum(acc[g][1].x), s3 = dpp_quad_sum(acc[g][1].y);
                gs[g] = kq == 0 ? s0 : (kq == 1 ? s1 : (kq == 2 ? s2 : s3));
            }
            const float xi = (gs[0] + genc.x) + pv.x, xf = (gs[1] + genc.y) + pv.y;
            const float xc = (gs[2] + genc.z) + pv.z, xo = (gs[3] + genc.w) + pv.w;
            const float ig = sigmoidf_(xi), fg = sigmoidf_(xf), gg = tanhf(xc), og = sigmoidf_(xo);
            c_own = fg * c_own + ig * gg;
            h_own = og * tanhf(c_own);
        }
        store_granule(slot + (size_t)m * GRAN + tid, epoch, h_own);
        I2L_STAMP(2);
        // ---- D. the other three quarters of h
        u64_t gr[3];
        {
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int qi = 0; qi < 3; ++qi) {
                    const int q = qi + (qi >= m ? 1 : 0);
                    gr[qi] = load_granule(slot + (size_t)q * GRAN + tid);
                    ok = ok && (unsigned)(gr[qi] >> 32) == epoch;
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > GRP_TIMEOUT_TICKS) { failed = true; break; }
                }
            }
        }
        I2L_STAMP(3);
        // every thread passed the barrier of B (t > 0) after its reads of h_s, so h_s may be rewritten right away
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
            const int q = qi + (qi >= m ? 1 : 0);
            h_s[q * GRP_HS + tid] = __uint_as_float((unsigned)gr[qi]);
        }
        h_s[m * GRP_HS + tid] = h_own;
        if (failed) { tok_s[4] = 1; failed = false; }      // a timed-out poll is reported at the next B (all threads see it)
        __syncthreads();
        I2L_STAMP(4);

        // ---- E. logits of this member's 128 columns: thread = (4 columns 4cq.., k = ke mod 8), 4 rows -> 16 sums
        f32x2 pa[4][2];                                     // [column][row pair]
#pragma unroll
        for (int c = 0; c < 4; ++c) { pa[c][0] = splat2(0.f); pa[c][1] = splat2(0.f); }
        {
            f32x2 wb[2][4][2];
            float4 hb[2][4];
            auto fetch = [&](int set, int b) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 8 * (b * 4 + i) + ke;
                    const float4 w4 = *reinterpret_cast<const float4*>(wout_s + k * GRP_WLD + 4 * cq);
                    wb[set][i][0] = f32x2{w4.x, w4.y};
                    wb[set][i][1] = f32x2{w4.z, w4.w};
                    hb[set][i] = *reinterpret_cast<const float4*>(h_s + (k >> 6) * GRP_HS + (k & 63) * 4);
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) fetch((b + 1) & 1, b + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 hv = hb[b & 1][i];
                    const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
                    pkfma_lo(pa[0][0], wb[b & 1][i][0], h01); pkfma_lo(pa[0][1], wb[b & 1][i][0], h23);
                    pkfma_hi(pa[1][0], wb[b & 1][i][0], h01); pkfma_hi(pa[1][1], wb[b & 1][i][0], h23);
                    pkfma_lo(pa[2][0], wb[b & 1][i][1], h01); pkfma_lo(pa[2][1], wb[b & 1][i][1], h23);
                    pkfma_hi(pa[3][0], wb[b & 1][i][1], h01); pkfma_hi(pa[3][1], wb[b & 1][i][1], h23);
                }
            }
        }
        I2L_STAMP(5);
        // the 8 lanes of a (4-column) group fold their k-slices; lane ke then owns row ke & 3, columns 2 (ke >> 2) + {0,1}
        float lv[2];
        {
            float sm[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sm[c][0] = dpp_oct_sum(pa[c][0].x); sm[c][1] = dpp_oct_sum(pa[c][0].y);
                sm[c][2] = dpp_oct_sum(pa[c][1].x); sm[c][3] = dpp_oct_sum(pa[c][1].y);
            }
            const int rr = ke & 3;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float lo = rr == 0 ? sm[e][0] : (rr == 1 ? sm[e][1] : (rr == 2 ? sm[e][2] : sm[e][3]));
                const float hi = rr == 0 ? sm[2 + e][0] : (rr == 1 ? sm[2 + e][1] : (rr == 2 ? sm[2 + e][2] : sm[2 + e][3]));
                lv[e] = ((ke >> 2) ? hi : lo) + (e == 0 ? bo.x : bo.y);
            }
        }
        const int v0 = 128 * m + 4 * cq + 2 * (ke >> 2);
        if (lrow) {
            if (v0 < V) lrow[(size_t)t * V + v0] = lv[0];
            if (v0 + 1 < V) lrow[(size_t)t * V + v0 + 1] = lv[1];
        }
        if (p.use_temp) { lv[0] = lv[0] / p.temperature; lv[1] = lv[1] / p.temperature; }
        u64_t key = umax64(am_key(lv[0], v0), am_key(lv[1], v0 + 1));
        key = umax64(key, dpp_u64<0x124>(key));             // row_ror:4, row_ror:8: the 4 lanes of a 16-lane row that
        key = umax64(key, dpp_u64<0x128>(key));             // share (lane & 3) now hold the same key
        const int par = (t & 1) * 64;
        if ((lane & 15) < 4) redk[par + (lane & 3) * 16 + wave * 4 + (lane >> 4)] = key;
        __syncthreads();
        if (wave == 0 && lane < 4) {
            const ulonglong2* src = reinterpret_cast<const ulonglong2*>(redk + par + lane * 16);
            u64_t best = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const ulonglong2 kk2 = src[j]; best = umax64(best, umax64(kk2.x, kk2.y)); }
            cand_k = best;
            __hip_atomic_store(slot + (size_t)m * GRAN + GRAN_H + lane,
                               ((u64_t)((epoch << 16) | (unsigned)(am_idx(best) & 0xFFFF)) << 32) |
                                   (u64_t)__float_as_uint(am_val(best)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        I2L_STAMP(6);
    }
#ifdef I2L_GROUP_STAMPS
    if (tid == 0 && blockIdx.x < 32) for (int i = 0; i < 8; ++i) p.status[8 + blockIdx.x * 8 + i] = (unsigned)st_acc[i];
#endif
    if (failed) {
        if (tid == 0) atomicOr(p.status, 1u);
        if (ids_row) for (int tt = tid; tt < T; tt += NT) ids_row[tt] = -3;
        return;
    }
    if (ids_row) for (int tt = t + tid; tt < T; tt += NT) ids_row[tt] = -1;   // steps never executed (sticky stop)
}
