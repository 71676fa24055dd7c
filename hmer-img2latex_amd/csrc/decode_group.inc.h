// Grouped greedy decode (included by decode.hip inside its anonymous namespace).
//
// The row-per-workgroup kernel streams ~1.1 MB of weights from L2 into every CU at every step (the L2 -> CU path,
// 64 B/clk/CU, bounds it: ~11.5 us/step).  Here FOUR workgroups (one per CU) decode FOUR batch rows together and
// each keeps one QUARTER of the weights on chip for the whole loop, so nothing is streamed:
//     member m holds  WhhT[:, gate columns of hidden units 64m..64m+63]   256 x 256 floats in registers
//                     WoutT[:, vocabulary columns 128m..128m+127]          256 x 128 floats in LDS
// with 512 threads per workgroup (two waves per SIMD: one wave's LDS / DPP / transcendental latencies are the
// other's issue slots, and 128 weights per thread stay in architectural registers).
// Per step member m computes, for all 4 rows, the gates / LSTM cell of ITS 64 hidden units and the logits of ITS 128
// vocabulary columns; both need the full h of the group, so a step has two small exchanges inside the group:
//     1. every member publishes its 64 x 4 new h values and reads the other three quarters;
//     2. every member publishes, per row, the (max, first index) of its vocabulary quarter; all four combine the four
//        candidates in member (= index) order, so they agree on the token without further talk.
// The recurrent product h . Whh^T of the NEXT step does not depend on the token (only the table row P[token] that is
// added at its end does), so it runs between publishing the candidates and polling for them: exchange 2 is hidden.
//
// Synchronisation inside a workgroup is ONE barrier per step (after the h exchange).  The two others of the first
// design are gone: (i) every WAVE polls the 16 candidate granules of the previous step itself (its own member's too:
// they are published to L2 like the peers'), merges them in registers and -- because it needs nobody's permission --
// does so in the middle of the recurrent product, so that the gather of the token's table row P[token] is in flight
// while the rest of that product runs; (ii) the per-row arg max over the eight waves is an LDS 64-bit atomic max, and
// the wave that arrives last (an LDS counter tells it) publishes the member's four candidates.  h lives in two LDS
// buffers by step parity, which is what makes the remaining barrier sufficient.
//
// Exchange = 8-byte {value, tag} granules, tag = step + 1, each written by ONE agent-scope relaxed atomic store (sc1,
// write-through) and polled by agent-scope relaxed atomic loads (sc1: never served from this CU's L1 or a stale L2
// line): the data is its own flag, no fences, no dependence on XCD placement.  Two buffers by step parity suffice:
// a member overwrites a slot two steps later, after it has consumed from every peer data that the peer published
// after reading that slot.  The granule block is zeroed by a memset node before every launch.
//
// Progress: the members of a group are workgroups 32a + x + 8k (k = 0..3) -- one XCD under round-robin placement
// (speed only) -- and workgroups are dispatched in index order, so on a GPU with at least the 32 CUs of one grid
// slice free some complete group is resident and runs to its end, freeing its CUs for the next.  That is an
// assumption about the dispatcher and about what else runs on the device, not something HIP promises (the launch
// is not cooperative): with fewer CUs free every resident workgroup may be waiting for a member that is not, so
// EVERY poll is bounded by a wall-clock limit: on expiry the workgroup raises status[0], fills its ids with -3 (and
// its logits with NaN) and exits, its peers follow by their own limits, and the host wrapper reports the failure
// and re-runs on the row-per-workgroup kernel; the GPU is never left spinning.
// Memory model: the granule stores are relaxed atomics at AGENT scope (sc1, write-through) and the polls agent-scope
// loads -- conformant HSA.  When the four members measure themselves on one XCD the stores drop to WORKGROUP scope
// (a plain store that gfx950's write-through L1 forwards to the XCD's L2, where the peers' L1-bypassing polls find
// it: 1.28 -> 0.50 us per exchange); that relies on the gfx950 cache hierarchy, not on the memory model, so the
// conformant flavour stays selectable (I2L_FLAG_AGENT_SCOPE_EXCHANGE) and tested.
#include "group_common.inc.h"

constexpr int GQ = 4;                    // workgroups (= rows) per group
constexpr int GNT = 512;                 // threads per workgroup
constexpr int GRAN_H = 256;              // h granules per member and step: [unit 64][row 4]
constexpr int GRAN_C = GRAN_H;           // 4 candidate granules (one per row), a 128-byte line of their own
constexpr int GRAN_X = GRAN_H + 16;      // 1 placement granule (XCC id), a line of its own
constexpr int GRAN = GRAN_H + 32;
// LDS: W_out quarter | h [2 parities][256 k][4 rows] | per-thread image-side gate constants (Genc, read once per step: four
// registers less across the recurrent product) | arg-max keys [2 parities][4 rows] | arrival counters [2] + flags
constexpr size_t GRP_LDS = (size_t)(256 * 128 + 2 * 256 * 4 + GNT * 4) * sizeof(float) + (size_t)2 * 4 * 8 + 8 * sizeof(int);

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
    GroupOpts opts;       // poll limits, exchange flavour (group_common.inc.h)
    unsigned* resident_flag;   // may be null: receives resident_value once every group of the launch is resident
    unsigned resident_value;
};

constexpr int DPP_SHL12 = 0x10C;

__global__ __launch_bounds__(GNT) void decode_group_kernel(GroupParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* wout_s4 = reinterpret_cast<float4*>(smem);      // [16 j][512 threads]: WoutT[16j + ks][128m + 4cq ..+3]
    float* h_s = smem + 256 * 128;                          // [2][256 k][4 rows]  full h of step t in buffer t & 1
    float4* genc_s = reinterpret_cast<float4*>(h_s + 2 * 256 * 4);   // [512 threads] Genc[row ke & 3][4 unit ..+3]
    u64_t* redk = reinterpret_cast<u64_t*>(genc_s + GNT);   // [2][4 rows] arg-max keys of this member's columns
    int* cnt_s = reinterpret_cast<int*>(redk + 2 * 4);      // [0..1] waves that added their keys (by parity),
                                                            // [2] a poll timed out, [3] members share one XCD

    const StepWeights& w = p.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int within = blockIdx.x & 31;
    const int group = (blockIdx.x >> 5) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int B = p.B, T = p.T, V = w.V;
    const int row0 = group * GQ;
    // gates: hidden unit 64m + ul, k = ke (mod 8) -> 32 k, 4 gates x 4 rows;  cell: lanes ke < 4 own (unit, row ke)
    const int ul = tid >> 3, ke = tid & 7;
    const int unit = 64 * m + ul;
    // logits: columns 128m + 4cq ..+3, k = ks (mod 16) -> 16 k, 4 columns x 4 rows; afterwards the lane owns ONE logit
    const int cq = tid >> 4, ks = tid & 15;
    const int l_row = ks & 3, l_col = ((ks >> 2) & 1) * 2 + (ks >> 3);
    const int l_v = 128 * m + 4 * cq + l_col;
    constexpr int G = 1024;                                 // 4 * H

    f32x2 wreg[32][2];                                      // WhhT[8j + ke][4 unit .. +3] as (i,f), (g,o)
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float4 t4 = *reinterpret_cast<const float4*>(w.WhhT[0] + (size_t)(8 * j + ke) * G + 4 * unit);
        wreg[j][0] = f32x2{t4.x, t4.y};
        wreg[j][1] = f32x2{t4.z, t4.w};
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
        wout_s4[j * GNT + tid] = *reinterpret_cast<const float4*>(w.WoutT + (size_t)(16 * j + ks) * 512 + 128 * m + 4 * cq);
    for (int idx = tid; idx < 2 * 256 * 4; idx += GNT) h_s[idx] = 0.f;
    if (tid < 8) redk[tid] = 0;
    if (tid < 4) cnt_s[tid] = 0;
    genc_s[tid] = *reinterpret_cast<const float4*>(w.Genc + (size_t)min(row0 + (ke & 3), B - 1) * G + 4 * unit);
    const float l_bias = w.boutP[l_v];
    float c_own = 0.f, h_own = 0.f;
    unsigned fin = 0;                                       // bit r: row r has emitted END (or does not exist)
#pragma unroll
    for (int r = 0; r < GQ; ++r)
        if (row0 + r >= B) fin |= 1u << r;
    u64_t* xg = p.xchg + (size_t)group * 2 * GQ * GRAN;
    const bool own_row = row0 + m < B;                      // this member writes the ids of row m
    int32_t* ids_row = (p.ids && own_row) ? p.ids + (size_t)(row0 + m) * T : nullptr;
    float* lrow = (p.logits && row0 + l_row < B && l_v < V) ? p.logits + (size_t)(row0 + l_row) * T * V + l_v : nullptr;
    // Placement: are the four members on one XCD?  Each publishes its XCC id (sc1, seen from anywhere); equal ids
    // switch the granule stores to the L2-local flavour.  Measured, never assumed: correctness does not depend on it
    // (if a member times out here it fails the launch like any other poll).
    __syncthreads();                                        // the LDS initialisation above
    if (wave == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        if (lane == 0 && !(p.opts.drop_member && m == 3))      // test hook: member 3 stays silent, its peers time out
            store_granule(xg + (size_t)m * GRAN + GRAN_X, granule(0xC0DEu, __uint_as_float(xcc)), false);
        const int pq = (lane & 3) + ((lane & 3) >= m ? 1 : 0);
        u64_t pv = 0;
        bool bad = false;
        long long t_start = 0;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            if (lane < 3) { pv = load_granule(xg + (size_t)pq * GRAN + GRAN_X); ok = (unsigned)(pv >> 32) == 0xC0DEu; }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
                const long long now = (long long)wall_clock64();
                if (t_start == 0) t_start = now;
                else if (now - t_start > p.opts.limit_first) { bad = true; break; }
            }
        }
        const bool all_same = __all(lane >= 3 || (unsigned)pv == xcc);
        if (lane == 0) {
            cnt_s[3] = (all_same && !bad) ? 1 : 0;
            if (bad) cnt_s[2] = 1;
            if (m == 0 && !bad) {                            // placement statistics of the launch (read by the host on request)
                count_resident_group(p.status, p.n_groups, p.resident_flag, p.resident_value);
                if (all_same && !p.opts.agent_scope) atomicAdd(p.status + GRP_STAT_LOCAL, 1u);
            }
        }
    }
    __syncthreads();
    const bool local = cnt_s[3] != 0 && !p.opts.agent_scope;

    // token of the row this lane's cell belongs to (row ke & 3) and its table row, gathered as early as the token is known
    const int my_r = ke & 3;
    int mytok = min(max(p.tok0[min(row0 + my_r, B - 1)], 0), V - 1);
    if (p.forced) mytok = min(max(p.forced[(size_t)min(row0 + my_r, B - 1) * T], 0), V - 1);
    float4 pvec = *reinterpret_cast<const float4*>(w.P + (size_t)mytok * G + 4 * unit);

    int t = 0;
    bool failed = cnt_s[2] != 0;
#ifdef I2L_GROUP_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long st_last = (long long)wall_clock64();
#define I2L_STAMP(i) do { const long long n_ = (long long)wall_clock64(); st_acc[i] += n_ - st_last; st_last = n_; } while (0)
#else
#define I2L_STAMP(i) do { } while (0)
#endif
    for (; !failed; ++t) {
        // ---- A. token-independent part of the gates of step t: sum_k h(t-1)[k] Whh[k][.]; on the way, the tokens chosen
        //         at step t-1: every wave polls the 16 candidate granules {member q, row r} = lane 4q + r itself
        f32x2 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) { acc[g][0] = splat2(0.f); acc[g][1] = splat2(0.f); }
        const float* hprev = h_s + ((t - 1) & 1) * 1024;
        const u64_t* cand_src = xg + (size_t)((t - 1) & 1) * GQ * GRAN + (size_t)((lane >> 2) & 3) * GRAN + GRAN_C + (lane & 3);
        const unsigned c_epoch = (unsigned)t;               // candidates of step t-1 carry tag t
        bool have = t == 0;                                 // tokens of step t-1 known (wave-uniform)
        int tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
        // The poll is split in two so that the L2 round trip of the sc1 loads hides behind the product: issue() sends the
        // 16 loads, check() -- three k batches later -- looks at what came back: all 16 granules there -> merge (members
        // own disjoint index ranges: any order), broadcast the four tokens, start the gather of this lane's table row.
        u64_t gv = 0;
        auto issue = [&]() {
            if (lane < 16) gv = load_granule(cand_src);
        };
        auto check = [&]() {
            if (!__all(lane >= 16 || (unsigned)(gv >> 48) == c_epoch)) return;
            const int g_lo = (int)(unsigned)gv, g_hi = (int)(unsigned)(gv >> 32);
            u64_t best = am_key(__int_as_float(g_lo), g_hi & 0xFFFF);
            best = umax64(best, am_key(__int_as_float(__builtin_amdgcn_mov_dpp(g_lo, DPP_SHL4, 0xF, 0xF, true)),
                                       __builtin_amdgcn_mov_dpp(g_hi, DPP_SHL4, 0xF, 0xF, true) & 0xFFFF));
            best = umax64(best, am_key(__int_as_float(__builtin_amdgcn_mov_dpp(g_lo, DPP_SHL8, 0xF, 0xF, true)),
                                       __builtin_amdgcn_mov_dpp(g_hi, DPP_SHL8, 0xF, 0xF, true) & 0xFFFF));
            best = umax64(best, am_key(__int_as_float(__builtin_amdgcn_mov_dpp(g_lo, DPP_SHL12, 0xF, 0xF, true)),
                                       __builtin_amdgcn_mov_dpp(g_hi, DPP_SHL12, 0xF, 0xF, true) & 0xFFFF));
            const int bi = am_idx(best);                    // lanes 0..3: the token of row = lane
            tk0 = __builtin_amdgcn_readlane(bi, 0); tk1 = __builtin_amdgcn_readlane(bi, 1);
            tk2 = __builtin_amdgcn_readlane(bi, 2); tk3 = __builtin_amdgcn_readlane(bi, 3);
            tk0 = tk0 < V ? tk0 : 0; tk1 = tk1 < V ? tk1 : 0; tk2 = tk2 < V ? tk2 : 0; tk3 = tk3 < V ? tk3 : 0;
            have = true;
            if (t < T) {
                mytok = my_r == 0 ? tk0 : (my_r == 1 ? tk1 : (my_r == 2 ? tk2 : tk3));
                if (p.forced) mytok = min(max(p.forced[(size_t)min(row0 + my_r, B - 1) * T + t], 0), V - 1);
                pvec = *reinterpret_cast<const float4*>(w.P + (size_t)mytok * G + 4 * unit);
            }
        };
        if (t > 0 && t < T) {
            // h is read 4 k ahead of its use (two register sets)
            const float4* hq4 = reinterpret_cast<const float4*>(hprev) + ke;
            float4 hb[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hb[0][i] = hq4[8 * i];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hb[(b + 1) & 1][i] = hq4[8 * ((b + 1) * 4 + i)];
                }
                // the peers published their candidates about when this member did: ask early, look three batches later
                if (b == 1) issue();
                if (b == 4) { check(); if (!have) issue(); }
                if (b == 7 && !have) check();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) fma_4x4(acc, wreg[b * 4 + i][0], wreg[b * 4 + i][1], hb[b & 1][i]);
            }
        }
        I2L_STAMP(0);
        // ---- B. wait for the tokens if they are not there yet.  A wave whose poll times out does NOT leave on its own:
        //         it raises cnt_s[2], skips the rest of the step's work and meets the others at the step's barrier, after
        //         which every wave leaves together (uniform barrier use on the failure path too)
        bool bail = false;
        if (!have) {
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                issue();
                check();
                if (have) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > p.opts.limit_step) { bail = true; break; }
                }
            }
            if (bail) cnt_s[2] = 1;
        }
        if (t > 0 && !bail) {
            const int tk[4] = {tk0, tk1, tk2, tk3};
            bool all_fin = true;
#pragma unroll
            for (int r = 0; r < GQ; ++r) {
                const bool was_fin = (fin >> r) & 1u;
                if (r == m && tid == 0 && ids_row) ids_row[t - 1] = (p.stop == I2L_STOP_STICKY && was_fin) ? -1 : tk[r];
                if (tk[r] == p.end_id) fin |= 1u << r;
                all_fin = all_fin && ((fin >> r) & 1u);
            }
            if (t == T || (p.stop == I2L_STOP_STICKY && all_fin)) break;
        }
        I2L_STAMP(1);
        // ---- C. fold the 8 k-slices (reduce-scatter: 16 -> 8 -> 4 -> 2 values per lane), LSTM cell of (unit, row ke)
        const unsigned epoch = (unsigned)t + 1u;
        const int par = t & 1;
        u64_t* slot = xg + (size_t)par * GQ * GRAN;
        float* hcur = h_s + par * 1024;
        if (!bail) {
            const float4 genc = genc_s[tid];
            const bool b0 = ke & 1, b1 = ke & 2, b2 = ke & 4;
            float z[2];                                     // lanes ke < 4: gates (i, f); ke >= 4: (g, o); row ke & 3
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
            const float g2 = dpp_f<DPP_SHL4>(z[0]), g3 = dpp_f<DPP_SHL4>(z[1]);      // (g, o) from lane ke + 4
            const float xi = (z[0] + genc.x) + pvec.x, xf = (z[1] + genc.y) + pvec.y;
            const float xc = (g2 + genc.z) + pvec.z, xo = (g3 + genc.w) + pvec.w;
            const float ig = sigmoidf_(xi), fg = sigmoidf_(xf), gg = tanhf_(xc), og = sigmoidf_(xo);
            c_own = fg * c_own + ig * gg;
            h_own = og * tanhf_(c_own);
        }
        if (ke < 4 && !bail) {
            store_granule(slot + (size_t)m * GRAN + ul * 4 + ke, granule(epoch, h_own), local);
            hcur[m * 256 + ul * 4 + ke] = h_own;
        }
        I2L_STAMP(2);
        // ---- D. the other three quarters of h: threads 0..255 fetch peers 0 and 1, threads 256..511 peer 2
        if (!bail) {
            u64_t gr[2];
            const int gi = tid & 255;
            const int qa = tid < 256 ? 0 : 2;
            const u64_t* pa_ = slot + (size_t)(qa + (qa >= m ? 1 : 0)) * GRAN + gi;
            const u64_t* pb_ = slot + (size_t)(1 + (1 >= m ? 1 : 0)) * GRAN + gi;
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                gr[0] = load_granule(pa_);
                gr[1] = tid < 256 ? load_granule(pb_) : gr[0];
                if ((unsigned)(gr[0] >> 32) == epoch && (unsigned)(gr[1] >> 32) == epoch) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > p.opts.limit_step) { cnt_s[2] = 1; break; }
                }
            }
            I2L_STAMP(3);
            hcur[(qa + (qa >= m ? 1 : 0)) * 256 + gi] = __uint_as_float((unsigned)gr[0]);
            if (tid < 256) hcur[(1 + (1 >= m ? 1 : 0)) * 256 + gi] = __uint_as_float((unsigned)gr[1]);
        }
        __syncthreads();                                    // THE barrier of the step: h(t) complete in hcur
        if (cnt_s[2] != 0) { failed = true; break; }        // some wave's poll timed out: everybody leaves here
        I2L_STAMP(4);

        // ---- E. logits of this member's 128 columns: thread = (4 columns, k = ks mod 16), 4 rows -> 16 sums
        f32x2 pa[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) { pa[c][0] = splat2(0.f); pa[c][1] = splat2(0.f); }
        {
            const float4* hq4 = reinterpret_cast<const float4*>(hcur) + ks;
            float4 wb[2][2], hb[2][2];
            auto fetch = [&](int set, int b) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wb[set][i] = wout_s4[(b * 2 + i) * GNT + tid];
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
        }
        I2L_STAMP(5);
        // fold the 16 k-slices: 16 -> 8 -> 4 -> 2 -> 1; the lane ends with the logit of (column l_col, row l_row)
        float lv;
        {
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
            lv = rs_level<DPP_ROR8>(z[0], z[1], b3) + l_bias;
        }
        if (lrow) lrow[(size_t)t * V] = lv;
        if (p.use_temp) lv = lv / p.temperature;
        u64_t key = am_key(lv, l_v);
        key = umax64(key, dpp_u64<DPP_ROR4>(key));          // the 4 lanes of a 16-lane row that share the row (lane & 3)
        key = umax64(key, dpp_u64<DPP_ROR8>(key));
        // ---- F. arg max of the member's 128 columns per row: LDS atomic max of the 16 (row, column quad) keys of the
        //         wave; the wave that adds last publishes the four candidates and clears the other parity for step t+1
        if (ks < 4) atomicMax(reinterpret_cast<unsigned long long*>(redk + par * 4 + ks), (unsigned long long)key);
        int arrived = 0;
        if (lane == 0) arrived = atomicAdd(cnt_s + par, 1);   // LDS operations of a wave execute in order
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived == GNT / 64 - 1) {
            if (lane < 4) {
                const u64_t best = redk[par * 4 + lane];
                store_granule(slot + (size_t)m * GRAN + GRAN_C + lane,
                              granule((epoch << 16) | (unsigned)(am_idx(best) & 0xFFFF), am_val(best)), local);
                redk[(par ^ 1) * 4 + lane] = 0;
            }
            if (lane == 0) cnt_s[par ^ 1] = 0;
        }
        I2L_STAMP(6);
    }
#ifdef I2L_GROUP_STAMPS
    if (tid == 0 && blockIdx.x < 32) for (int i = 0; i < 8; ++i) p.status[8 + blockIdx.x * 8 + i] = (unsigned)st_acc[i];
#endif
    if (failed) {
        // loud failure: ids -3 (checked by the host wrappers) and NaN logits (a caller that asked for logits only --
        // the validation forward -- gets a NaN loss instead of a partly written tensor)
        if (lane == 0) atomicOr(p.status, 1u);
        if (ids_row) for (int tt = tid; tt < T; tt += GNT) ids_row[tt] = -3;
        if (lrow) for (int tt = 0; tt < T; ++tt) lrow[(size_t)tt * V] = __builtin_nanf("");
        return;
    }
    if (ids_row) for (int tt = t + tid; tt < T; tt += GNT) ids_row[tt] = -1;   // steps never executed (sticky stop)
}
#undef I2L_STAMP
