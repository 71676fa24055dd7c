// Grouped greedy decode, EIGHT members x EIGHT rows (included by decode.hip inside its anonymous namespace, after
// decode_group.inc.h).  Same step as decode_group_kernel -- recurrent product, cell, h exchange, logits, arg max,
// candidate exchange, all phases with the same arithmetic per output -- but a member holds an EIGHTH of the weights:
//     WhhT[:, gate columns of hidden units 32m..32m+31]   256 x 128 floats in registers (128 per thread of 256)
//     WoutT[:, vocabulary columns 64m..64m+63]             256 x 64 floats in LDS (64 KB)
// i.e. ONE wave per SIMD, <= 256 registers, ~80 KB of LDS: the footprint that leaves room for one conv workgroup
// (247 registers x 4 waves, 70 KB) on the same CU, so that the encoder of batch i + 1 can run beside the decode of
// batch i (profiles/r03/coresident.txt).  The FMA count per member and step is unchanged (8 rows x 1/8 of
// the columns instead of 4 rows x 1/4); the price is 7 peers to poll instead of 3 and a deeper reduce-scatter.
//   thread (ul = tid >> 3, ke = tid & 7): gates of hidden unit 32m + ul over k = ke (mod 8); after the fold lane ke owns
//                                         ROW ke of that unit (cell, h granule: all 8 lanes busy)
//   thread (cq = tid >> 4, ks = tid & 15): vocabulary columns 64m + 4cq..+3 over k = ks (mod 16); after the fold the lane
//                                         owns columns 2*(ks & 1), +1 of ROW (ks >> 1)
// Exchange granules, tags, time-outs, placement measurement, failure marking: exactly as decode_group_kernel.
// Supported: ids out (both stop rules, temperature, argmax of logits / of softmax); no logits output, no forced tokens,
// no initial / returned state -- those launches take the 4-member kernel.

constexpr int G8Q = 8;                    // members = rows per group
constexpr int G8NT = 256;                 // threads per workgroup
constexpr int G8GRAN_H = 256;             // h granules per member and step: [unit 32][row 8]
constexpr int G8GRAN_C = G8GRAN_H;        // 8 candidate granules, a line of their own
constexpr int G8GRAN_X = G8GRAN_H + 16;   // placement granule
constexpr int G8GRAN = G8GRAN_H + 32;
// LDS: W_out eighth (64 KB) | h [2 parities][256 k][8 rows] | arg-max keys [2][8] | counters
constexpr size_t GRP8_LDS = (size_t)(256 * 64 + 2 * 256 * 8) * sizeof(float) + (size_t)2 * 8 * 8 + 8 * sizeof(int);
constexpr size_t GROUP8_XCHG_PER_GROUP = (size_t)2 * G8Q * G8GRAN * 8;

__global__ __launch_bounds__(G8NT) void decode_group8_kernel(GroupParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* wout_s4 = reinterpret_cast<float4*>(smem);      // [16 j][256 threads]: WoutT[16j + ks][64m + 4cq ..+3]
    float* h_s = smem + 256 * 64;                           // [2][256 k][8 rows]
    u64_t* redk = reinterpret_cast<u64_t*>(h_s + 2 * 256 * 8);   // [2][8 rows]
    int* cnt_s = reinterpret_cast<int*>(redk + 2 * 8);      // [0..1] waves arrived (by parity), [2] time-out, [3] one XCD

    const StepWeights& w = p.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int within = blockIdx.x & 63;
    const int group = (blockIdx.x >> 6) * 8 + (within & 7), m = within >> 3;
    if (group >= p.n_groups) return;
    const int B = p.B, T = p.T, V = w.V;
    const int row0 = group * G8Q;
    const int ul = tid >> 3, ke = tid & 7;
    const int unit = 32 * m + ul;
    const int cq = tid >> 4, ks = tid & 15;
    const int l_row = ks >> 1;                              // row of the two logits this lane ends up with
    const int l_v = 64 * m + 4 * cq + 2 * (ks & 1);         // their vocabulary columns l_v, l_v + 1
    constexpr int G = 1024;

    f32x2 wreg[32][2];                                      // WhhT[8j + ke][4 unit .. +3] as (i,f), (g,o)
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float4 t4 = *reinterpret_cast<const float4*>(w.WhhT[0] + (size_t)(8 * j + ke) * G + 4 * unit);
        wreg[j][0] = f32x2{t4.x, t4.y};
        wreg[j][1] = f32x2{t4.z, t4.w};
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
        wout_s4[j * G8NT + tid] = *reinterpret_cast<const float4*>(w.WoutT + (size_t)(16 * j + ks) * 512 + 64 * m + 4 * cq);
    for (int idx = tid; idx < 2 * 256 * 8; idx += G8NT) h_s[idx] = 0.f;
    if (tid < 16) redk[tid] = 0;
    if (tid < 4) cnt_s[tid] = 0;
    const float4 genc = *reinterpret_cast<const float4*>(w.Genc + (size_t)min(row0 + ke, B - 1) * G + 4 * unit);
    const float l_bias0 = w.boutP[l_v], l_bias1 = w.boutP[l_v + 1];
    float c_own = 0.f, h_own = 0.f;
    unsigned fin = 0;
#pragma unroll
    for (int r = 0; r < G8Q; ++r)
        if (row0 + r >= B) fin |= 1u << r;
    u64_t* xg = p.xchg + (size_t)group * 2 * G8Q * G8GRAN;
    const bool own_row = row0 + m < B;
    int32_t* ids_row = (p.ids && own_row) ? p.ids + (size_t)(row0 + m) * T : nullptr;

    __syncthreads();
    if (wave == 0) {                                        // placement: are the eight members on one XCD?
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        if (lane == 0 && !(p.opts.drop_member && m == 3))
            store_granule(xg + (size_t)m * G8GRAN + G8GRAN_X, granule(0xC0DEu, __uint_as_float(xcc)), false);
        const int pq = (lane & 7) + ((lane & 7) >= m ? 1 : 0);
        u64_t pv = 0;
        bool bad = false;
        long long t_start = 0;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            if (lane < 7) { pv = load_granule(xg + (size_t)pq * G8GRAN + G8GRAN_X); ok = (unsigned)(pv >> 32) == 0xC0DEu; }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
                const long long now = (long long)wall_clock64();
                if (t_start == 0) t_start = now;
                else if (now - t_start > p.opts.limit_first) { bad = true; break; }
            }
        }
        const bool all_same = __all(lane >= 7 || (unsigned)pv == xcc);
        if (lane == 0) {
            cnt_s[3] = (all_same && !bad) ? 1 : 0;
            if (bad) cnt_s[2] = 1;
            if (m == 0 && !bad) {
                count_resident_group(p.status, p.n_groups, p.resident_flag, p.resident_value);
                if (all_same && !p.opts.agent_scope) atomicAdd(p.status + GRP_STAT_LOCAL, 1u);
            }
        }
    }
    __syncthreads();
    const bool local = cnt_s[3] != 0 && !p.opts.agent_scope;

    int mytok = min(max(p.tok0[min(row0 + ke, B - 1)], 0), V - 1);          // token of row ke (this lane's cell)
    float4 pvec = *reinterpret_cast<const float4*>(w.P + (size_t)mytok * G + 4 * unit);

    int t = 0;
    bool failed = cnt_s[2] != 0;
    for (; !failed; ++t) {
        // ---- A. sum_k h(t-1)[k] Whh[k][.] for 8 rows; on the way every wave polls the 64 candidate granules
        //         {member q, row r} = lane 8q + r of step t-1 and merges them into the 8 tokens
        f32x2 acc[2][4][2];                                  // [row half][gate][row pair of the half]
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int g = 0; g < 4; ++g) { acc[hf][g][0] = splat2(0.f); acc[hf][g][1] = splat2(0.f); }
        const float* hprev = h_s + ((t - 1) & 1) * 2048;
        const u64_t* cand_src = xg + (size_t)((t - 1) & 1) * G8Q * G8GRAN + (size_t)(lane >> 3) * G8GRAN + G8GRAN_C + (lane & 7);
        const unsigned c_epoch = (unsigned)t;
        bool have = t == 0;
        int tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        u64_t gv = 0;
        auto issue = [&]() { gv = load_granule(cand_src); };
        auto check = [&]() {
            if (!__all((unsigned)(gv >> 48) == c_epoch)) return;
            u64_t best = am_key(__uint_as_float((unsigned)gv), (int)((unsigned)(gv >> 32) & 0xFFFFu));
            best = umax64(best, dpp_u64<DPP_SHL8>(best));    // lanes 0..7 of a 16-lane row: members 2j, 2j + 1
            best = umax64(best, (u64_t)__shfl_xor((unsigned long long)best, 16));
            best = umax64(best, (u64_t)__shfl_xor((unsigned long long)best, 32));
            const int bi = am_idx(best);                    // lanes 0..7: the token of row = lane
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                tk[r] = __builtin_amdgcn_readlane(bi, r);
                tk[r] = tk[r] < V ? tk[r] : 0;
            }
            have = true;
            if (t < T) {
                mytok = tk[0];
#pragma unroll
                for (int r = 1; r < 8; ++r) mytok = ke == r ? tk[r] : mytok;
                pvec = *reinterpret_cast<const float4*>(w.P + (size_t)mytok * G + 4 * unit);
            }
        };
        if (t > 0 && t < T) {
            const float4* hq4 = reinterpret_cast<const float4*>(hprev) + 2 * ke;     // row [k][8]: two float4 per k
            float4 hb[2][2][2];                                                     // [set][k of the pair][row half]
#pragma unroll
            for (int i = 0; i < 2; ++i) { hb[0][i][0] = hq4[16 * i]; hb[0][i][1] = hq4[16 * i + 1]; }
#pragma unroll
            for (int b = 0; b < 16; ++b) {                  // 16 batches of 2 k (k = 8 (2b + i) + ke)
                if (b + 1 < 16) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        hb[(b + 1) & 1][i][0] = hq4[16 * ((b + 1) * 2 + i)];
                        hb[(b + 1) & 1][i][1] = hq4[16 * ((b + 1) * 2 + i) + 1];
                    }
                }
                if (b == 2) issue();
                if (b == 8) { check(); if (!have) issue(); }
                if (b == 14 && !have) check();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fma_4x4(acc[0], wreg[b * 2 + i][0], wreg[b * 2 + i][1], hb[b & 1][i][0]);
                    fma_4x4(acc[1], wreg[b * 2 + i][0], wreg[b * 2 + i][1], hb[b & 1][i][1]);
                }
            }
        }
        // ---- B. wait for the tokens if they are not there yet (uniform exit on a time-out: see decode_group_kernel)
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
            bool all_fin = true;
#pragma unroll
            for (int r = 0; r < G8Q; ++r) {
                const bool was_fin = (fin >> r) & 1u;
                if (r == m && tid == 0 && ids_row) ids_row[t - 1] = (p.stop == I2L_STOP_STICKY && was_fin) ? -1 : tk[r];
                if (tk[r] == p.end_id) fin |= 1u << r;
                all_fin = all_fin && ((fin >> r) & 1u);
            }
            if (t == T || (p.stop == I2L_STOP_STICKY && all_fin)) break;
        }
        // ---- C. fold the 8 k-slices: 32 sums -> the 4 gates of row ke on lane ke (three reduce-scatter levels), cell
        const unsigned epoch = (unsigned)t + 1u;
        const int par = t & 1;
        u64_t* slot = xg + (size_t)par * G8Q * G8GRAN;
        float* hcur = h_s + par * 2048;
        if (!bail) {
            const bool b0 = ke & 1, b1 = ke & 2, b2 = ke & 4;
            float z[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // level 1 (lane <-> 7 - lane): lanes 0..3 keep rows 0..3, lanes 4..7 rows 4..7
                const float u0x = rs_level<DPP_HMIRROR>(acc[0][g][0].x, acc[1][g][0].x, b2);   // row 0 | 4
                const float u0y = rs_level<DPP_HMIRROR>(acc[0][g][0].y, acc[1][g][0].y, b2);   // row 1 | 5
                const float u1x = rs_level<DPP_HMIRROR>(acc[0][g][1].x, acc[1][g][1].x, b2);   // row 2 | 6
                const float u1y = rs_level<DPP_HMIRROR>(acc[0][g][1].y, acc[1][g][1].y, b2);   // row 3 | 7
                // level 2 (lane ^ 2): keep the row pair b1
                const float vx = rs_level<DPP_XOR2>(u0x, u1x, b1);
                const float vy = rs_level<DPP_XOR2>(u0y, u1y, b1);
                // level 3 (lane ^ 1): keep row b0 of the pair
                z[g] = rs_level<DPP_XOR1>(vx, vy, b0);
            }
            const float xi = (z[0] + genc.x) + pvec.x, xf = (z[1] + genc.y) + pvec.y;
            const float xc = (z[2] + genc.z) + pvec.z, xo = (z[3] + genc.w) + pvec.w;
            const float ig = sigmoidf_(xi), fg = sigmoidf_(xf), gg = tanhf_(xc), og = sigmoidf_(xo);
            c_own = fg * c_own + ig * gg;
            h_own = og * tanhf_(c_own);
            store_granule(slot + (size_t)m * G8GRAN + tid, granule(epoch, h_own), local);      // granule (unit ul, row ke) = tid
            hcur[m * 256 + tid] = h_own;                                                      // h_s[k = 32m + ul][row ke]
        }
        // ---- D. the other seven eighths of h: every thread fetches granule `tid` of each peer
        if (!bail) {
            u64_t gr[7];
            const u64_t* src[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) src[q] = slot + (size_t)(q + (q >= m ? 1 : 0)) * G8GRAN + tid;
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < 7; ++q) gr[q] = load_granule(src[q]);
#pragma unroll
                for (int q = 0; q < 7; ++q) ok = ok && (unsigned)(gr[q] >> 32) == epoch;
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > p.opts.limit_step) { cnt_s[2] = 1; break; }
                }
            }
#pragma unroll
            for (int q = 0; q < 7; ++q) hcur[(q + (q >= m ? 1 : 0)) * 256 + tid] = __uint_as_float((unsigned)gr[q]);
        }
        __syncthreads();                                    // THE barrier of the step: h(t) complete in hcur
        if (cnt_s[2] != 0) { failed = true; break; }

        // ---- E. logits of this member's 64 columns: thread = (4 columns, k = ks mod 16), 8 rows -> 32 sums
        f32x2 pa[2][4][2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int c = 0; c < 4; ++c) { pa[hf][c][0] = splat2(0.f); pa[hf][c][1] = splat2(0.f); }
        {
            const float4* hq4 = reinterpret_cast<const float4*>(hcur) + 2 * ks;
            float4 wb[2], hb[2][2];
            wb[0] = wout_s4[tid];
            hb[0][0] = hq4[0]; hb[0][1] = hq4[1];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j + 1 < 16) {
                    wb[(j + 1) & 1] = wout_s4[(j + 1) * G8NT + tid];
                    hb[(j + 1) & 1][0] = hq4[32 * (j + 1)];
                    hb[(j + 1) & 1][1] = hq4[32 * (j + 1) + 1];
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4 w4 = wb[j & 1];
                fma_4x4(pa[0], f32x2{w4.x, w4.y}, f32x2{w4.z, w4.w}, hb[j & 1][0]);
                fma_4x4(pa[1], f32x2{w4.x, w4.y}, f32x2{w4.z, w4.w}, hb[j & 1][1]);
            }
        }
        // fold the 16 k-slices: 32 -> 16 -> 8 -> 4 -> 2; the lane ends with columns 2 b0, 2 b0 + 1 of row ks >> 1
        float lv0, lv1;
        {
            const bool b0 = ks & 1, b1 = ks & 2, b2 = ks & 4, b3 = ks & 8;
            float v[4];                                      // per column: row (ks >> 1) after three row levels
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // level 1 (lane ^ 8 within 16): lanes ks < 8 keep rows 0..3, the others rows 4..7
                const float u0x = rs_level<DPP_ROR8>(pa[0][c][0].x, pa[1][c][0].x, b3);
                const float u0y = rs_level<DPP_ROR8>(pa[0][c][0].y, pa[1][c][0].y, b3);
                const float u1x = rs_level<DPP_ROR8>(pa[0][c][1].x, pa[1][c][1].x, b3);
                const float u1y = rs_level<DPP_ROR8>(pa[0][c][1].y, pa[1][c][1].y, b3);
                // level 2 (lane <-> 7 - lane within 8): keep the row pair b2
                const float wx = rs_level<DPP_HMIRROR>(u0x, u1x, b2);
                const float wy = rs_level<DPP_HMIRROR>(u0y, u1y, b2);
                // level 3 (lane ^ 2): keep row b1 of the pair
                v[c] = rs_level<DPP_XOR2>(wx, wy, b1);
            }
            // level 4 (lane ^ 1): keep the column pair b0
            lv0 = rs_level<DPP_XOR1>(v[0], v[2], b0) + l_bias0;
            lv1 = rs_level<DPP_XOR1>(v[1], v[3], b0) + l_bias1;
        }
        if (p.use_temp) { lv0 = lv0 / p.temperature; lv1 = lv1 / p.temperature; }
        u64_t key = umax64(am_key(lv0, l_v), am_key(lv1, l_v + 1));
        key = umax64(key, dpp_u64<DPP_XOR1>(key));          // the two lanes (b0 = 0, 1) that share (column quad, row)
        // ---- F. arg max of the member's 64 columns per row: LDS atomic max, the last wave publishes the 8 candidates
        if ((ks & 1) == 0) atomicMax(reinterpret_cast<unsigned long long*>(redk + par * 8 + l_row), (unsigned long long)key);
        int arrived = 0;
        if (lane == 0) arrived = atomicAdd(cnt_s + par, 1);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived == G8NT / 64 - 1) {
            if (lane < 8) {
                const u64_t best = redk[par * 8 + lane];
                store_granule(slot + (size_t)m * G8GRAN + G8GRAN_C + lane,
                              granule((epoch << 16) | (unsigned)(am_idx(best) & 0xFFFF), am_val(best)), local);
                redk[(par ^ 1) * 8 + lane] = 0;
            }
            if (lane == 0) cnt_s[par ^ 1] = 0;
        }
    }
    if (failed) {
        if (lane == 0) atomicOr(p.status, 1u);
        if (ids_row) for (int tt = tid; tt < T; tt += G8NT) ids_row[tt] = -3;
        return;
    }
    if (ids_row) for (int tt = t + tid; tt < T; tt += G8NT) ids_row[tt] = -1;
}
