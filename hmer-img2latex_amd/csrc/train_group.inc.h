// Grouped training recurrences (included by train_decoder.hip inside its anonymous namespace): the forward
// recurrence and the BPTT of the L = 1, H = 256 LSTM with the same 4-workgroups-share-4-rows scheme as the grouped
// decode kernel (decode_group.inc.h) -- W_hh lives in registers for all T steps, one tagged-granule exchange per
// step inside the group, placement measured through HW_REG_XCC_ID, every poll bounded.
//   forward : member m owns hidden units 64m..64m+63: gates = GX[t] + h . Whh^T (its 256 gate columns, 4 rows),
//             cell, stores ACT / C / Hout / Hprev, publishes its 64 x 4 h values.
//   backward: member m owns the same units: gate gradients from (dh, dc) for its units, stores DG, publishes its
//             256 x 4 gate gradients; dh_{t-1}[own units] = sum over ALL 1024 gate rows n of Whh[n][unit] dG[n].
// A timed-out wait raises status[0] and fills the outputs of the affected rows with NaN (the loss turns NaN).
#include "group_common.inc.h"

constexpr int TGT = 512;                       // threads per workgroup
constexpr int TGF_GRAN = 256 + 32;             // forward: h granules + placement line (at 272)
constexpr int TGB_GRAN = 1024 + 16;            // backward: gate-gradient granules + placement line (at 1024)

struct TrainGroupFwd {
    int B, T, n_groups;
    const float* GX;      // [B*T][4H] gate-interleaved input gates (biases included)
    const float* WhhT;    // [H][4H] gate-interleaved transpose
    float* ACT; float* C; float* Hout; float* Hprev;
    u64_t* xchg;          // [n_groups][2][4][TGF_GRAN]
    unsigned* status;
    GroupOpts opts;       // poll limits, exchange flavour (group_common.inc.h)
};
struct TrainGroupBwd {
    int B, T, n_groups;
    const float* ACT; const float* C; const float* dHtop;
    const float* Whh;     // (4H, H) as stored by nn.LSTM
    float* DG;            // [B*T][4H] standard gate order
    u64_t* xchg;          // [n_groups][2][4][TGB_GRAN]
    unsigned* status;
    GroupOpts opts;       // poll limits, exchange flavour (group_common.inc.h)
};

// Are the four members on one XCD?  (decode_group.inc.h: measured, never assumed.)  Returns via LDS word flag[1];
// flag[0] is set when the wait itself timed out.  Called by every thread; contains a barrier.
__device__ __forceinline__ bool group_placement_local(u64_t* xg, int gran, int xslot, int m, int* flag, long long limit, bool silent) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        if (lane == 0 && !silent) store_granule(xg + (size_t)m * gran + xslot, granule(0xC0DEu, __uint_as_float(xcc)), false);
        const int pq = (lane & 3) + ((lane & 3) >= m ? 1 : 0);
        u64_t pv = 0;
        bool bad = false;
        long long t_start = 0;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            if (lane < 3) { pv = load_granule(xg + (size_t)pq * gran + xslot); ok = (unsigned)(pv >> 32) == 0xC0DEu; }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
                const long long now = (long long)wall_clock64();
                if (t_start == 0) t_start = now;
                else if (now - t_start > limit) { bad = true; break; }
            }
        }
        const bool all_same = __all(lane >= 3 || (unsigned)pv == xcc);
        if (lane == 0) { flag[1] = (all_same && !bad) ? 1 : 0; flag[0] = bad ? 1 : 0; }
    }
    __syncthreads();
    return flag[1] != 0;
}

__global__ __launch_bounds__(TGT) void lstm_train_fwd_group_kernel(TrainGroupFwd p) {
    __shared__ __attribute__((aligned(16))) float h_s[2][1024];    // [parity][k][row]: h of the previous step
    __shared__ int flag[4];
    const int tid = threadIdx.x;
    const int within = blockIdx.x & 31;
    const int group = (blockIdx.x >> 5) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int B = p.B, T = p.T;
    const int row0 = group * 4;
    const int ul = tid >> 3, ke = tid & 7, kr = ke & 3;
    const int unit = 64 * m + ul;
    constexpr int G = 1024, H = 256;
    f32x2 wreg[32][2];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float4 t4 = *reinterpret_cast<const float4*>(p.WhhT + (size_t)(8 * j + ke) * G + 4 * unit);
        wreg[j][0] = f32x2{t4.x, t4.y};
        wreg[j][1] = f32x2{t4.z, t4.w};
    }
    for (int idx = tid; idx < 2 * 1024; idx += TGT) (&h_s[0][0])[idx] = 0.f;
    u64_t* xg = p.xchg + (size_t)group * 2 * 4 * TGF_GRAN;
    const bool local = group_placement_local(xg, TGF_GRAN, 272, m, flag, p.opts.limit_first, p.opts.drop_member && m == 3) && !p.opts.agent_scope;      // barrier inside: h_s zeroed
    const int row = min(row0 + kr, B - 1);
    const bool live = ke < 4 && row0 + kr < B;                                  // this lane owns (unit, row kr)
    float c_own = 0.f, h_own = 0.f;
    bool failed = false;
    int t = 0;
    float4 gx_next = *reinterpret_cast<const float4*>(p.GX + (size_t)row * T * G + 4 * unit);     // one step ahead, as in group1
    for (; t < T; ++t) {
        const size_t bt = (size_t)row * T + t;
        const float4 gx = gx_next;
        if (t + 1 < T) gx_next = *reinterpret_cast<const float4*>(p.GX + (bt + 1) * G + 4 * unit);
        f32x2 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) { acc[g][0] = splat2(0.f); acc[g][1] = splat2(0.f); }
        if (t > 0) {
            const float4* hq4 = reinterpret_cast<const float4*>(h_s[t & 1]) + ke;
            float4 hb[2][4];
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
        }
        {
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
            const float g2 = dpp_f<DPP_SHL4>(z[0]), g3 = dpp_f<DPP_SHL4>(z[1]);
            const float ig = sigmoidf_(gx.x + z[0]), fg = sigmoidf_(gx.y + z[1]);
            const float gg = tanhf_(gx.z + g2), og = sigmoidf_(gx.w + g3);
            const float h_prev = h_own;
            c_own = fg * c_own + ig * gg;
            h_own = og * tanhf_(c_own);
            if (live) {
                *reinterpret_cast<float4*>(p.ACT + bt * G + 4 * unit) = make_float4(ig, fg, gg, og);
                p.C[bt * H + unit] = c_own;
                p.Hout[bt * H + unit] = h_own;
                p.Hprev[bt * H + unit] = h_prev;
            }
        }
        if (t + 1 == T) break;
        const unsigned epoch = (unsigned)t + 1u;
        u64_t* slot = xg + (size_t)(t & 1) * 4 * TGF_GRAN;
        if (ke < 4) store_granule(slot + (size_t)m * TGF_GRAN + ul * 4 + ke, granule(epoch, h_own), local);
        {
            const int gi = tid & 255;
            const int qa = tid < 256 ? 0 : 2;
            const int q0 = qa + (qa >= m ? 1 : 0), q1 = 1 + (1 >= m ? 1 : 0);
            const u64_t* pa_ = slot + (size_t)q0 * TGF_GRAN + gi;
            const u64_t* pb_ = slot + (size_t)q1 * TGF_GRAN + gi;
            u64_t g0, g1;
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                g0 = load_granule(pa_);
                g1 = tid < 256 ? load_granule(pb_) : g0;
                if ((unsigned)(g0 >> 32) == epoch && (unsigned)(g1 >> 32) == epoch) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > p.opts.limit_step) { failed = true; break; }
                }
            }
            float* hn = h_s[(t + 1) & 1];
            hn[q0 * 256 + gi] = __uint_as_float((unsigned)g0);
            if (tid < 256) hn[q1 * 256 + gi] = __uint_as_float((unsigned)g1);
            if (ke < 4) hn[m * 256 + ul * 4 + ke] = h_own;
        }
        if (failed) flag[0] = 1;
        __syncthreads();
        if (flag[0]) { failed = true; break; }
    }
    if (failed || flag[0]) {
        if (tid == 0) atomicOr(p.status, 1u);
        if (live) for (int tt = 0; tt < T; ++tt) p.Hout[((size_t)row * T + tt) * H + unit] = __int_as_float(0x7fc00000);
    }
}

__global__ __launch_bounds__(TGT) void lstm_train_bwd_group_kernel(TrainGroupBwd p) {
    __shared__ __attribute__((aligned(16))) float dgs[2][4096];    // [parity][gate row n][row]: gate gradients of a step
    __shared__ int flag[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int within = blockIdx.x & 31;
    const int group = (blockIdx.x >> 5) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int B = p.B, T = p.T;
    const int row0 = group * 4;
    constexpr int G = 1024, H = 256;
    // product: thread = (unit quad jq, n slice ns): dh[4 units][4 rows] += Whh[32 i + ns][units] dG[32 i + ns][rows]
    const int jq = tid >> 5, ns = tid & 31, ks = ns & 15;
    f32x2 wreg[32][2];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float4 t4 = *reinterpret_cast<const float4*>(p.Whh + (size_t)(32 * i + ns) * H + 64 * m + 4 * jq);
        wreg[i][0] = f32x2{t4.x, t4.y};
        wreg[i][1] = f32x2{t4.z, t4.w};
    }
    // after the fold the lane owns dh of (unit 4 jq + o_col, row o_row); lanes ns < 16 run the cell backward for it
    const int o_row = ks & 3, o_col = ((ks >> 2) & 1) * 2 + (ks >> 3);
    const int ul = 4 * jq + o_col, unit = 64 * m + ul;
    const int row = min(row0 + o_row, B - 1);
    const bool owner = ns < 16, live = owner && row0 + o_row < B;
    u64_t* xg = p.xchg + (size_t)group * 2 * 4 * TGB_GRAN;
    const bool local = group_placement_local(xg, TGB_GRAN, 1024, m, flag, p.opts.limit_first, p.opts.drop_member && m == 3) && !p.opts.agent_scope;
    float dh_rec = 0.f, dc_next = 0.f;
    bool failed = false;
    float4 a_nx = make_float4(0.f, 0.f, 0.f, 0.f);                             // step t-1's reads, requested during step t
    float c_nx = 0.f, cp_nx = 0.f, dht_nx = 0.f;
    auto prefetch = [&](int tt) {
        if (!owner || tt < 0) return;
        const size_t b2 = (size_t)row * T + tt;
        a_nx = *reinterpret_cast<const float4*>(p.ACT + b2 * G + 4 * unit);
        c_nx = p.C[b2 * H + unit];
        cp_nx = tt > 0 ? p.C[(b2 - 1) * H + unit] : 0.f;
        dht_nx = p.dHtop[b2 * H + unit];
    };
    prefetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const int par = t & 1;
        const unsigned epoch = (unsigned)(T - t);
        const size_t bt = (size_t)row * T + t;
        u64_t* slot = xg + (size_t)par * 4 * TGB_GRAN;
        float* dcur = dgs[par];
        const float4 a = a_nx;
        const float c = c_nx, cp = cp_nx, dht = dht_nx;
        prefetch(t - 1);
        if (owner) {
            const float dh = dh_rec + dht;
            const float tc = tanhf_(c);
            const float d_o = dh * tc * a.w * (1.f - a.w);
            const float dc = dh * a.w * (1.f - tc * tc) + dc_next;
            const float d_i = dc * a.z * a.x * (1.f - a.x);
            const float d_f = dc * cp * a.y * (1.f - a.y);
            const float d_g = dc * a.x * (1.f - a.z * a.z);
            dc_next = dc * a.y;
            const float dgv[4] = {d_i, d_f, d_g, d_o};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (live) p.DG[bt * G + g * H + unit] = dgv[g];
                dcur[(g * 256 + 64 * m + ul) * 4 + o_row] = dgv[g];
                if (t > 0) store_granule(slot + (size_t)m * TGB_GRAN + (g * 64 + ul) * 4 + o_row, granule(epoch, dgv[g]), local);
            }
        }
        if (t == 0) break;
        {
            u64_t gr[3][2];
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int qi = 0; qi < 3; ++qi) {
                    const int q = qi + (qi >= m ? 1 : 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        gr[qi][j] = load_granule(slot + (size_t)q * TGB_GRAN + tid + TGT * j);
                        ok = ok && (unsigned)(gr[qi][j] >> 32) == epoch;
                    }
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > p.opts.limit_step) { failed = true; break; }
                }
            }
#pragma unroll
            for (int qi = 0; qi < 3; ++qi) {
                const int q = qi + (qi >= m ? 1 : 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = tid + TGT * j;                    // granule (g*64 + ul')*4 + row of member q
                    dcur[(e >> 8) * 1024 + 256 * q + (e & 255)] = __uint_as_float((unsigned)gr[qi][j]);
                }
            }
        }
        if (failed) flag[0] = 1;
        __syncthreads();
        if (flag[0]) { failed = true; break; }
        // dh_{t-1}[own units] = sum_n Whh[n][unit] dG[n]
        f32x2 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) { acc[g][0] = splat2(0.f); acc[g][1] = splat2(0.f); }
        {
            const float4* dq4 = reinterpret_cast<const float4*>(dcur) + ns;
            float4 hb[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hb[0][i] = dq4[32 * i];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hb[(b + 1) & 1][i] = dq4[32 * ((b + 1) * 4 + i)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) fma_4x4(acc, wreg[b * 4 + i][0], wreg[b * 4 + i][1], hb[b & 1][i]);
            }
        }
        {
            const bool b0 = ks & 1, b1 = ks & 2, b2 = ks & 4, b3 = ks & 8;
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
            const float v = rs_level<DPP_ROR8>(z[0], z[1], b3);      // n slices 0..15 (or 16..31) folded
            dh_rec = v + __shfl_xor(v, 16);                          // + the other 16 slices
        }
    }
    if (failed || flag[0]) {
        if (tid == 0) atomicOr(p.status, 1u);
        if (live) for (int tt = 0; tt < T; ++tt) p.DG[((size_t)row * T + tt) * G + unit] = __int_as_float(0x7fc00000);
    }
    (void)lane;
}

// ---------------------------------------------------------------------------------------------------------------
// Two rows per group (B <= 128: BASELINE configs[3] runs 64 rows per GPU).  With four rows per group 64 rows keep only
// 64 of the 256 CUs busy and a step is bound by the in-group exchange plus the 4-row product (0.85 us of packed
// FMAs); with two rows per group the same batch spreads over 128 CUs, the product halves and so does every
// exchange (forward 64 x 2 h values, backward 256 x 2 gate gradients per member).  Same scheme, same granule
// protocol, same numerics (the per-row arithmetic is identical, so results are bit-identical to the 4-row kernels).
// ---------------------------------------------------------------------------------------------------------------
constexpr int TGF2_GRAN = 128 + 32;            // forward: h granules [unit 64][row 2] + placement line (at 144)
constexpr int TGB2_GRAN = 512 + 16;            // backward: gate-gradient granules [gate 4][unit 64][row 2] + placement (at 512)
constexpr int DPP_SHL2 = 0x102, DPP_SHL6 = 0x106;

// acc[4 values] (one row pair each) += w4 (4 gates or units) x h2 (2 rows)
__device__ __forceinline__ void fma_4x2(f32x2 (&acc)[4], f32x2 w01, f32x2 w23, f32x2 h) {
    pkfma_lo(acc[0], w01, h); pkfma_hi(acc[1], w01, h);
    pkfma_lo(acc[2], w23, h); pkfma_hi(acc[3], w23, h);
}

__global__ __launch_bounds__(TGT) void lstm_train_fwd_group2_kernel(TrainGroupFwd p) {
    __shared__ __attribute__((aligned(16))) float h_s[2][512];     // [parity][k][row]: h of the previous step
    __shared__ int flag[4];
    const int tid = threadIdx.x;
    const int within = blockIdx.x & 31;
    const int group = (blockIdx.x >> 5) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int B = p.B, T = p.T;
    const int row0 = group * 2;
    const int ul = tid >> 3, ke = tid & 7, kr = ke & 1;
    const int unit = 64 * m + ul;
    constexpr int G = 1024, H = 256;
    f32x2 wreg[32][2];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float4 t4 = *reinterpret_cast<const float4*>(p.WhhT + (size_t)(8 * j + ke) * G + 4 * unit);
        wreg[j][0] = f32x2{t4.x, t4.y};
        wreg[j][1] = f32x2{t4.z, t4.w};
    }
    for (int idx = tid; idx < 2 * 512; idx += TGT) (&h_s[0][0])[idx] = 0.f;
    u64_t* xg = p.xchg + (size_t)group * 2 * 4 * TGF2_GRAN;
    const bool local = group_placement_local(xg, TGF2_GRAN, 144, m, flag, p.opts.limit_first, p.opts.drop_member && m == 3) && !p.opts.agent_scope;   // barrier inside: h_s zeroed
    const int row = min(row0 + kr, B - 1);
    const bool live = ke < 2 && row0 + kr < B;                                  // this lane owns (unit, row kr)
    float c_own = 0.f, h_own = 0.f;
    bool failed = false;
    int t = 0;
    float4 gx_next = *reinterpret_cast<const float4*>(p.GX + (size_t)row * T * G + 4 * unit);     // one step ahead, as in group1
    for (; t < T; ++t) {
        const size_t bt = (size_t)row * T + t;
        const float4 gx = gx_next;
        if (t + 1 < T) gx_next = *reinterpret_cast<const float4*>(p.GX + (bt + 1) * G + 4 * unit);
        f32x2 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = splat2(0.f);
        if (t > 0) {
            const f32x2* hq2 = reinterpret_cast<const f32x2*>(h_s[t & 1]) + ke;
            f32x2 hb[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hb[0][i] = hq2[8 * i];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hb[(b + 1) & 1][i] = hq2[8 * ((b + 1) * 4 + i)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) fma_4x2(acc, wreg[b * 4 + i][0], wreg[b * 4 + i][1], hb[b & 1][i]);
            }
        }
        {
            // fold the 8 k-slices: 8 -> 4 -> 2 -> 1 value per lane: lane ke holds gate 2*(ke>>2) + (ke>>1 & 1) of row ke & 1
            const bool b0 = ke & 1, b1 = ke & 2, b2 = ke & 4;
            float wv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float ux = rs_level<DPP_HMIRROR>(acc[e].x, acc[2 + e].x, b2);
                const float uy = rs_level<DPP_HMIRROR>(acc[e].y, acc[2 + e].y, b2);
                wv[e] = rs_level<DPP_XOR1>(ux, uy, b0);
            }
            const float zi = rs_level<DPP_XOR2>(wv[0], wv[1], b1);
            const float zf = dpp_f<DPP_SHL2>(zi), zg = dpp_f<DPP_SHL4>(zi), zo = dpp_f<DPP_SHL6>(zi);
            const float ig = sigmoidf_(gx.x + zi), fg = sigmoidf_(gx.y + zf);
            const float gg = tanhf_(gx.z + zg), og = sigmoidf_(gx.w + zo);
            const float h_prev = h_own;
            c_own = fg * c_own + ig * gg;
            h_own = og * tanhf_(c_own);
            if (live) {
                *reinterpret_cast<float4*>(p.ACT + bt * G + 4 * unit) = make_float4(ig, fg, gg, og);
                p.C[bt * H + unit] = c_own;
                p.Hout[bt * H + unit] = h_own;
                p.Hprev[bt * H + unit] = h_prev;
            }
        }
        if (t + 1 == T) break;
        const unsigned epoch = (unsigned)t + 1u;
        u64_t* slot = xg + (size_t)(t & 1) * 4 * TGF2_GRAN;
        if (ke < 2) store_granule(slot + (size_t)m * TGF2_GRAN + ul * 2 + ke, granule(epoch, h_own), local);
        {
            const int gi = tid & 127, pi = min(tid >> 7, 2);
            const int q = pi + (pi >= m ? 1 : 0);
            const u64_t* pa_ = slot + (size_t)q * TGF2_GRAN + gi;
            u64_t g0 = 0;
            long long t_start = 0;
            unsigned spins = 0;
            if (tid < 384) {
                for (;;) {
                    g0 = load_granule(pa_);
                    if ((unsigned)(g0 >> 32) == epoch) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0) {
                        const long long now = (long long)wall_clock64();
                        if (t_start == 0) t_start = now;
                        else if (now - t_start > p.opts.limit_step) { failed = true; break; }
                    }
                }
            }
            float* hn = h_s[(t + 1) & 1];
            if (tid < 384) hn[q * 128 + gi] = __uint_as_float((unsigned)g0);
            if (ke < 2) hn[m * 128 + ul * 2 + ke] = h_own;
        }
        if (failed) flag[0] = 1;
        __syncthreads();
        if (flag[0]) { failed = true; break; }
    }
    if (failed || flag[0]) {
        if (tid == 0) atomicOr(p.status, 1u);
        if (live) for (int tt = 0; tt < T; ++tt) p.Hout[((size_t)row * T + tt) * H + unit] = __int_as_float(0x7fc00000);
    }
}

__global__ __launch_bounds__(TGT) void lstm_train_bwd_group2_kernel(TrainGroupBwd p) {
    __shared__ __attribute__((aligned(16))) float dgs[2][2048];    // [parity][gate row n][row]: gate gradients of a step
    __shared__ int flag[4];
    const int tid = threadIdx.x;
    const int within = blockIdx.x & 31;
    const int group = (blockIdx.x >> 5) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int B = p.B, T = p.T;
    const int row0 = group * 2;
    constexpr int G = 1024, H = 256;
    // product: thread = (unit quad jq, n slice ns): dh[4 units][2 rows] += Whh[32 i + ns][units] dG[32 i + ns][rows]
    const int jq = tid >> 5, ns = tid & 31;
    f32x2 wreg[32][2];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float4 t4 = *reinterpret_cast<const float4*>(p.Whh + (size_t)(32 * i + ns) * H + 64 * m + 4 * jq);
        wreg[i][0] = f32x2{t4.x, t4.y};
        wreg[i][1] = f32x2{t4.z, t4.w};
    }
    // after the fold the lane owns dh of (unit 4 jq + o_col, row o_row); lanes ns < 8 run the cell backward for it
    const int o_row = ns & 1, o_col = ((ns >> 2) & 1) * 2 + ((ns >> 1) & 1);
    const int ul = 4 * jq + o_col, unit = 64 * m + ul;
    const int row = min(row0 + o_row, B - 1);
    const bool owner = ns < 8, live = owner && row0 + o_row < B;
    u64_t* xg = p.xchg + (size_t)group * 2 * 4 * TGB2_GRAN;
    const bool local = group_placement_local(xg, TGB2_GRAN, 512, m, flag, p.opts.limit_first, p.opts.drop_member && m == 3) && !p.opts.agent_scope;
    float dh_rec = 0.f, dc_next = 0.f;
    bool failed = false;
    float4 a_nx = make_float4(0.f, 0.f, 0.f, 0.f);                             // step t-1's reads, requested during step t
    float c_nx = 0.f, cp_nx = 0.f, dht_nx = 0.f;
    auto prefetch = [&](int tt) {
        if (!owner || tt < 0) return;
        const size_t b2 = (size_t)row * T + tt;
        a_nx = *reinterpret_cast<const float4*>(p.ACT + b2 * G + 4 * unit);
        c_nx = p.C[b2 * H + unit];
        cp_nx = tt > 0 ? p.C[(b2 - 1) * H + unit] : 0.f;
        dht_nx = p.dHtop[b2 * H + unit];
    };
    prefetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const int par = t & 1;
        const unsigned epoch = (unsigned)(T - t);
        const size_t bt = (size_t)row * T + t;
        u64_t* slot = xg + (size_t)par * 4 * TGB2_GRAN;
        float* dcur = dgs[par];
        const float4 a = a_nx;
        const float c = c_nx, cp = cp_nx, dht = dht_nx;
        prefetch(t - 1);
        if (owner) {
            const float dh = dh_rec + dht;
            const float tc = tanhf_(c);
            const float d_o = dh * tc * a.w * (1.f - a.w);
            const float dc = dh * a.w * (1.f - tc * tc) + dc_next;
            const float d_i = dc * a.z * a.x * (1.f - a.x);
            const float d_f = dc * cp * a.y * (1.f - a.y);
            const float d_g = dc * a.x * (1.f - a.z * a.z);
            dc_next = dc * a.y;
            const float dgv[4] = {d_i, d_f, d_g, d_o};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (live) p.DG[bt * G + g * H + unit] = dgv[g];
                dcur[(g * 256 + 64 * m + ul) * 2 + o_row] = dgv[g];
                if (t > 0) store_granule(slot + (size_t)m * TGB2_GRAN + (g * 64 + ul) * 2 + o_row, granule(epoch, dgv[g]), local);
            }
        }
        if (t == 0) break;
        {
            u64_t gr[3];
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int qi = 0; qi < 3; ++qi) {
                    const int q = qi + (qi >= m ? 1 : 0);
                    gr[qi] = load_granule(slot + (size_t)q * TGB2_GRAN + tid);
                    ok = ok && (unsigned)(gr[qi] >> 32) == epoch;
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > p.opts.limit_step) { failed = true; break; }
                }
            }
#pragma unroll
            for (int qi = 0; qi < 3; ++qi) {
                const int q = qi + (qi >= m ? 1 : 0);
                // granule tid = (g*64 + ul')*2 + row of member q  ->  gate row n = g*256 + 64 q + ul'
                dcur[(tid >> 7) * 512 + 128 * q + (tid & 127)] = __uint_as_float((unsigned)gr[qi]);
            }
        }
        if (failed) flag[0] = 1;
        __syncthreads();
        if (flag[0]) { failed = true; break; }
        // dh_{t-1}[own units] = sum_n Whh[n][unit] dG[n]
        f32x2 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = splat2(0.f);
        {
            const f32x2* dq2 = reinterpret_cast<const f32x2*>(dcur) + ns;
            f32x2 hb[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hb[0][i] = dq2[32 * i];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hb[(b + 1) & 1][i] = dq2[32 * ((b + 1) * 4 + i)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) fma_4x2(acc, wreg[b * 4 + i][0], wreg[b * 4 + i][1], hb[b & 1][i]);
            }
        }
        {
            const bool b0 = ns & 1, b1 = ns & 2, b2 = ns & 4;
            float wv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float ux = rs_level<DPP_HMIRROR>(acc[e].x, acc[2 + e].x, b2);
                const float uy = rs_level<DPP_HMIRROR>(acc[e].y, acc[2 + e].y, b2);
                wv[e] = rs_level<DPP_XOR1>(ux, uy, b0);
            }
            const float z = rs_level<DPP_XOR2>(wv[0], wv[1], b1);    // the 8 n slices of this lane's group of 8 folded
            const float v = z + dpp_f<DPP_ROR8>(z);                   // + the other group of 8 of the 16-lane row
            dh_rec = v + __shfl_xor(v, 16);                           // + the other 16 slices
        }
    }
    if (failed || flag[0]) {
        if (tid == 0) atomicOr(p.status, 1u);
        if (live) for (int tt = 0; tt < T; ++tt) p.DG[((size_t)row * T + tt) * G + unit] = __int_as_float(0x7fc00000);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ONE row per group (B <= 64: all 256 CUs work on BASELINE configs[3]'s 64 rows per GPU).  With a single row the
// packed FMA has no second row to fill its upper half, so it is filled with a second k instead: a thread's weights are
// stored as pairs {W[k0], W[k1]}, h (or dG) comes from LDS as the matching pair (the LDS image is permuted so that the
// pair is one 8-byte read), an accumulator holds {sum over its k0s, sum over its k1s} and the halves are added before
// the cross-lane fold.  Per step a thread issues 64 packed FMAs (two-row kernel: 128) and the exchanges halve again
// (forward 64 h values, backward 256 gate gradients per member).  Summation order differs from the 2 / 4-row kernels
// (k pairs), so results agree to fp32 rounding, not to the bit.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TGF1_GRAN = 64 + 32;             // forward: h granules [unit 64] + placement line (at 80)
constexpr int TGB1_GRAN = 256 + 16;            // backward: gate-gradient granules [gate 4][unit 64] + placement (at 256)
constexpr int DPP_SHL1 = 0x101, DPP_SHL5 = 0x105;

__device__ __forceinline__ void pkfma(f32x2& acc, f32x2 a, f32x2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__global__ __launch_bounds__(TGT) void lstm_train_fwd_group1_kernel(TrainGroupFwd p) {
    // h image [parity][256]: h[k], k = 16 i + 8 s + ke, lives at ((i * 8 + ke) * 2 + s): a thread's pair (s = 0, 1) is adjacent
    __shared__ __attribute__((aligned(16))) float h_s[2][256];
    __shared__ int flag[4];
    const int tid = threadIdx.x;
    const int within = blockIdx.x & 31;
    const int group = (blockIdx.x >> 5) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int T = p.T;
    const int row = group;                                 // n_groups == B
    const int ul = tid >> 3, ke = tid & 7;
    const int unit = 64 * m + ul;
    constexpr int G = 1024, H = 256;
    f32x2 wp[4][16];                                       // wp[gate][i] = {WhhT[16 i + ke][gate], WhhT[16 i + 8 + ke][gate]}
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 t0 = *reinterpret_cast<const float4*>(p.WhhT + (size_t)(16 * i + ke) * G + 4 * unit);
        const float4 t1 = *reinterpret_cast<const float4*>(p.WhhT + (size_t)(16 * i + 8 + ke) * G + 4 * unit);
        wp[0][i] = f32x2{t0.x, t1.x}; wp[1][i] = f32x2{t0.y, t1.y};
        wp[2][i] = f32x2{t0.z, t1.z}; wp[3][i] = f32x2{t0.w, t1.w};
    }
    for (int idx = tid; idx < 2 * 256; idx += TGT) (&h_s[0][0])[idx] = 0.f;
    u64_t* xg = p.xchg + (size_t)group * 2 * 4 * TGF1_GRAN;
    const bool local = group_placement_local(xg, TGF1_GRAN, 80, m, flag, p.opts.limit_first, p.opts.drop_member && m == 3) && !p.opts.agent_scope;   // barrier inside: h_s zeroed
    const bool live = ke == 0;                                                  // this lane owns the cell of `unit`
    auto hpos = [](int k) { return (((k >> 4) * 8 + (k & 7)) << 1) + ((k >> 3) & 1); };
    float c_own = 0.f, h_own = 0.f;
    bool failed = false;
    int t = 0;
    // the input-side gates of step t+1 are requested at the top of step t: with one row per group the recurrent product is
    // only ~0.2 us, far too short to hide an HBM round trip behind
    float4 gx_next = *reinterpret_cast<const float4*>(p.GX + (size_t)row * T * G + 4 * unit);
    for (; t < T; ++t) {
        const size_t bt = (size_t)row * T + t;
        const float4 gx = gx_next;
        if (t + 1 < T) gx_next = *reinterpret_cast<const float4*>(p.GX + (bt + 1) * G + 4 * unit);
        f32x2 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = splat2(0.f);
        if (t > 0) {
            const f32x2* hq2 = reinterpret_cast<const f32x2*>(h_s[t & 1]) + ke;      // pair i at hq2[8 i]
            f32x2 hb[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hb[0][i] = hq2[8 * i];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hb[(b + 1) & 1][i] = hq2[8 * ((b + 1) * 4 + i)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) pkfma(acc[g], wp[g][b * 4 + i], hb[b & 1][i]);
            }
        }
        {
            // halves, then the 8 k-slices: 4 -> 2 -> 1 value per lane (gate 2*(ke>>2) + (ke & 1)), then the two lane pairs add up
            const bool b0 = ke & 1, b2 = ke & 4;
            const float s0 = acc[0].x + acc[0].y, s1 = acc[1].x + acc[1].y, s2 = acc[2].x + acc[2].y, s3 = acc[3].x + acc[3].y;
            const float u0 = rs_level<DPP_HMIRROR>(s0, s2, b2), u1 = rs_level<DPP_HMIRROR>(s1, s3, b2);
            const float wv = rs_level<DPP_XOR1>(u0, u1, b0);
            const float zi = wv + dpp_f<DPP_XOR2>(wv);
            const float zf = dpp_f<DPP_SHL1>(zi), zg = dpp_f<DPP_SHL4>(zi), zo = dpp_f<DPP_SHL5>(zi);
            const float ig = sigmoidf_(gx.x + zi), fg = sigmoidf_(gx.y + zf);
            const float gg = tanhf_(gx.z + zg), og = sigmoidf_(gx.w + zo);
            const float h_prev = h_own;
            c_own = fg * c_own + ig * gg;
            h_own = og * tanhf_(c_own);
            if (live) {
                *reinterpret_cast<float4*>(p.ACT + bt * G + 4 * unit) = make_float4(ig, fg, gg, og);
                p.C[bt * H + unit] = c_own;
                p.Hout[bt * H + unit] = h_own;
                p.Hprev[bt * H + unit] = h_prev;
            }
        }
        if (t + 1 == T) break;
        const unsigned epoch = (unsigned)t + 1u;
        u64_t* slot = xg + (size_t)(t & 1) * 4 * TGF1_GRAN;
        if (live) store_granule(slot + (size_t)m * TGF1_GRAN + ul, granule(epoch, h_own), local);
        {
            const int gi = tid & 63, pi = min(tid >> 6, 2);
            const int q = pi + (pi >= m ? 1 : 0);
            const u64_t* pa_ = slot + (size_t)q * TGF1_GRAN + gi;
            u64_t g0 = 0;
            long long t_start = 0;
            unsigned spins = 0;
            if (tid < 192) {
                for (;;) {
                    g0 = load_granule(pa_);
                    if ((unsigned)(g0 >> 32) == epoch) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0) {
                        const long long now = (long long)wall_clock64();
                        if (t_start == 0) t_start = now;
                        else if (now - t_start > p.opts.limit_step) { failed = true; break; }
                    }
                }
            }
            float* hn = h_s[(t + 1) & 1];
            if (tid < 192) hn[hpos(64 * q + gi)] = __uint_as_float((unsigned)g0);
            if (live) hn[hpos(64 * m + ul)] = h_own;
        }
        if (failed) flag[0] = 1;
        __syncthreads();
        if (flag[0]) { failed = true; break; }
    }
    if (failed || flag[0]) {
        if (tid == 0) atomicOr(p.status, 1u);
        if (live) for (int tt = 0; tt < T; ++tt) p.Hout[((size_t)row * T + tt) * H + unit] = __int_as_float(0x7fc00000);
    }
}

__global__ __launch_bounds__(TGT) void lstm_train_bwd_group1_kernel(TrainGroupBwd p) {
    // gate-gradient image [parity][1024]: dG[n], n = 64 i + 32 s + ns, lives at ((i * 32 + ns) * 2 + s)
    __shared__ __attribute__((aligned(16))) float dgs[2][1024];
    __shared__ int flag[4];
    const int tid = threadIdx.x;
    const int within = blockIdx.x & 31;
    const int group = (blockIdx.x >> 5) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int T = p.T;
    const int row = group;
    constexpr int G = 1024, H = 256;
    // product: thread = (unit quad jq, n slice ns): dh[4 units] += Whh[64 i + 32 s + ns][units] dG[..], pairs over s
    const int jq = tid >> 5, ns = tid & 31;
    f32x2 wp[4][16];                                       // wp[unit][i] = {Whh[64 i + ns][unit], Whh[64 i + 32 + ns][unit]}
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 t0 = *reinterpret_cast<const float4*>(p.Whh + (size_t)(64 * i + ns) * H + 64 * m + 4 * jq);
        const float4 t1 = *reinterpret_cast<const float4*>(p.Whh + (size_t)(64 * i + 32 + ns) * H + 64 * m + 4 * jq);
        wp[0][i] = f32x2{t0.x, t1.x}; wp[1][i] = f32x2{t0.y, t1.y};
        wp[2][i] = f32x2{t0.z, t1.z}; wp[3][i] = f32x2{t0.w, t1.w};
    }
    // after the fold every lane of a group of 8 holds dh of unit 4 jq + o_col, o_col = 2*(ns>>2 & 1) + (ns & 1); the lanes
    // ns in {0, 1, 4, 5} run the cell backward for their unit
    const int o_col = ((ns >> 2) & 1) * 2 + (ns & 1);
    const int ul = 4 * jq + o_col, unit = 64 * m + ul;
    const bool owner = (ns & ~5) == 0;
    auto dpos = [](int n) { return (((n >> 6) * 32 + (n & 31)) << 1) + ((n >> 5) & 1); };
    u64_t* xg = p.xchg + (size_t)group * 2 * 4 * TGB1_GRAN;
    const bool local = group_placement_local(xg, TGB1_GRAN, 256, m, flag, p.opts.limit_first, p.opts.drop_member && m == 3) && !p.opts.agent_scope;
    float dh_rec = 0.f, dc_next = 0.f;
    bool failed = false;
    // what the cell backward of step t-1 reads (activations, cell states, dh from above) is requested during step t
    float4 a_nx = make_float4(0.f, 0.f, 0.f, 0.f);
    float c_nx = 0.f, cp_nx = 0.f, dht_nx = 0.f;
    auto prefetch = [&](int tt) {
        if (!owner || tt < 0) return;
        const size_t b2 = (size_t)row * T + tt;
        a_nx = *reinterpret_cast<const float4*>(p.ACT + b2 * G + 4 * unit);
        c_nx = p.C[b2 * H + unit];
        cp_nx = tt > 0 ? p.C[(b2 - 1) * H + unit] : 0.f;
        dht_nx = p.dHtop[b2 * H + unit];
    };
    prefetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const int par = t & 1;
        const unsigned epoch = (unsigned)(T - t);
        const size_t bt = (size_t)row * T + t;
        u64_t* slot = xg + (size_t)par * 4 * TGB1_GRAN;
        float* dcur = dgs[par];
        const float4 a = a_nx;
        const float c = c_nx, cp = cp_nx, dht = dht_nx;
        prefetch(t - 1);
        if (owner) {
            const float dh = dh_rec + dht;
            const float tc = tanhf_(c);
            const float d_o = dh * tc * a.w * (1.f - a.w);
            const float dc = dh * a.w * (1.f - tc * tc) + dc_next;
            const float d_i = dc * a.z * a.x * (1.f - a.x);
            const float d_f = dc * cp * a.y * (1.f - a.y);
            const float d_g = dc * a.x * (1.f - a.z * a.z);
            dc_next = dc * a.y;
            const float dgv[4] = {d_i, d_f, d_g, d_o};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                p.DG[bt * G + g * H + unit] = dgv[g];
                dcur[dpos(g * 256 + 64 * m + ul)] = dgv[g];
                if (t > 0) store_granule(slot + (size_t)m * TGB1_GRAN + g * 64 + ul, granule(epoch, dgv[g]), local);
            }
        }
        if (t == 0) break;
        {
            // 3 peers x 256 granules = 768: thread tid fetches granule tid & 255 of peer tid >> 8 (0, 1) and, below 256, of peer 2
            const int e = tid & 255;
            const int qa = (tid >> 8) + ((tid >> 8) >= m ? 1 : 0), qb = 2 + (2 >= m ? 1 : 0);
            u64_t ga, gb = 0;
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                ga = load_granule(slot + (size_t)qa * TGB1_GRAN + e);
                bool ok = (unsigned)(ga >> 32) == epoch;
                if (tid < 256) { gb = load_granule(slot + (size_t)qb * TGB1_GRAN + e); ok = ok && (unsigned)(gb >> 32) == epoch; }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > p.opts.limit_step) { failed = true; break; }
                }
            }
            // granule e = g*64 + ul' of member q  ->  gate row n = g*256 + 64 q + ul'
            dcur[dpos((e >> 6) * 256 + 64 * qa + (e & 63))] = __uint_as_float((unsigned)ga);
            if (tid < 256) dcur[dpos((e >> 6) * 256 + 64 * qb + (e & 63))] = __uint_as_float((unsigned)gb);
        }
        if (failed) flag[0] = 1;
        __syncthreads();
        if (flag[0]) { failed = true; break; }
        // dh_{t-1}[own units] = sum_n Whh[n][unit] dG[n]
        f32x2 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = splat2(0.f);
        {
            const f32x2* dq2 = reinterpret_cast<const f32x2*>(dcur) + ns;        // pair i at dq2[32 i]
            f32x2 hb[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hb[0][i] = dq2[32 * i];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hb[(b + 1) & 1][i] = dq2[32 * ((b + 1) * 4 + i)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int u = 0; u < 4; ++u) pkfma(acc[u], wp[u][b * 4 + i], hb[b & 1][i]);
            }
        }
        {
            const bool b0 = ns & 1, b2 = ns & 4;
            const float s0 = acc[0].x + acc[0].y, s1 = acc[1].x + acc[1].y, s2 = acc[2].x + acc[2].y, s3 = acc[3].x + acc[3].y;
            const float u0 = rs_level<DPP_HMIRROR>(s0, s2, b2), u1 = rs_level<DPP_HMIRROR>(s1, s3, b2);
            const float wv = rs_level<DPP_XOR1>(u0, u1, b0);
            const float z = wv + dpp_f<DPP_XOR2>(wv);                 // the 8 n slices of this lane's group of 8
            const float v = z + dpp_f<DPP_ROR8>(z);                   // + the other group of 8 of the 16-lane row
            dh_rec = v + __shfl_xor(v, 16);                           // + the other 16 slices
        }
    }
    if (failed || flag[0]) {
        if (tid == 0) atomicOr(p.status, 1u);
        if (owner) for (int tt = 0; tt < T; ++tt) p.DG[((size_t)row * T + tt) * G + unit] = __int_as_float(0x7fc00000);
    }
}
