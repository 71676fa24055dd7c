// Grouped greedy decode on the MATRIX CORES: SIXTEEN members x SIXTEEN rows per group (included by decode.hip inside its
// anonymous namespace, after decode_group8.inc.h).  r04; reference decoder.py:247-250,277-280 (LSTM gates, Linear(H -> V)),
// seq2seq.py:210-221 (arg max loop).
//
// Why: in the co-resident pipeline (decode(i) beside the conv workgroups of encoder(i + 1) on every CU) the 8-member
// kernel's per-step products are v_pk_fma_f32 bursts, and a vector burst beside resident conv waves -- which keep the
// vector ALUs busy splitting and staging their own operands -- takes 2.1x its time alone, while the same products as a
// burst of v_mfma_f32_16x16x32_bf16 take 1.13x (profiles/r04/coresident_mfma.txt).  With 16 rows per group the MFMA's
// M = 16 is full, so the split-bf16 form (3 bf16 pieces per fp32 operand, 6 partial products, fp32 accumulation: the
// error class of an fp32 fmaf chain, section "Split-bf16" of DESIGN.md) costs 6/8 of the packed-FMA time instead of 12/8.
//
// A member holds a SIXTEENTH of the weights, pre-split into bf16 pieces, in registers for the whole loop -- as the B
// operands of the MFMAs, in the matrix cores' own lane layout (lane l: B[k = 32 ks + 8 (l >> 4) + j][column l & 15]).
// Every wave w holds the same amount (144 registers):
//     WhhT[:, the 16 gate columns of hidden units 16m + 4w .. +3]                8 k-steps x 3 pieces x 4 registers = 96
//     WoutT[k half (w >> 1), vocabulary columns 32m + 16 (w & 1) .. +15]         4 k-steps x 3 pieces x 4 registers = 48
// h travels as its three bf16 pieces: an exchange granule is {tag 16 | p0 16 | p1 16 | p2 16} (h = p0 + p1 + p2 exactly),
// so nobody splits anything but the lane that computed the value, and LDS holds h only in operand layout:
//     hp[parity][piece][k block of 8][row 16][8]  (a lane's A fragment = one 16-byte read, conflict-free)
// Per step t (one barrier, as in the other grouped kernels):
//   A  gates(t)[16 rows x 16 columns per wave] = h(t-1) . WhhT  (48 MFMAs per wave) -- token independent, so it runs
//      while the candidates of step t-1 are in flight; every wave polls the 16 x 16 candidate granules itself
//   B  tokens of step t-1 (ids out, stop rule)
//   C  4 x 4 transposition inside lane quads brings the four gates of (unit, row) to one lane: ONE cell per lane, + Genc
//      + P[token], h(t) -> pieces -> granule (exchange 1) and own slice of hp
//   D  all threads: the other fifteen sixteenths of h(t): 64-byte runs (8 units of one row) -> three 16-byte LDS stores
//   -- barrier --
//   E  logits(t)[16 rows x 16 columns] = h(t) . WoutT, each wave half of the reduction of one column tile (24 MFMAs);
//      waves 2, 3 hand their partial sums to waves 0, 1 through LDS (an in-order write + flag, no barrier), which add,
//      take the arg max over the 16 lanes of a row, LDS atomic max across the two waves, and the later one publishes the
//      member's 16 candidates (exchange 2); waves 2, 3 are already in A of step t + 1.
// Tags, two buffers by step parity, bounded polls, placement measurement, failure marking (ids -3), residency signal:
// exactly as decode_group8_kernel.  Supported: ids out (both stop rules, temperature, arg max of logits / of softmax);
// no logits output, no forced tokens, no initial / returned state.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int G16Q = 16;                   // members = rows per group
constexpr int G16NT = 256;                 // threads per workgroup
constexpr int G16GRAN_H = 256;             // h granules per member and step: [k block 2][row 16][unit in block 8]
constexpr int G16GRAN_C = G16GRAN_H;       // 16 candidate granules, a 128-byte line of their own
constexpr int G16GRAN_X = G16GRAN_H + 16;  // placement granule
constexpr int G16GRAN = G16GRAN_H + 32;
constexpr int G16_HP_PIECE = 32 * 16 * 8;  // bf16 elements of one piece image
// LDS: hp [2 parities][3 pieces][32][16][8] bf16 (48 KB) | partial logits [2 tiles][64 lanes][4] | arg-max keys [2][16] | counters
constexpr size_t GRP16_LDS = (size_t)2 * 3 * G16_HP_PIECE * 2 + (size_t)2 * 64 * 16 + (size_t)2 * 16 * 8 + 8 * sizeof(int);
constexpr size_t GROUP16_XCHG_PER_GROUP = (size_t)2 * G16Q * G16GRAN * 8;

__device__ __forceinline__ unsigned short f2bf_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f_(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// x = p0 + p1 + p2 exactly (3 x 8 mantissa bits); every difference below is exact in fp32
__device__ __forceinline__ void split3(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    p0 = f2bf_rne(x);
    const float r1 = x - bf2f_(p0);
    p1 = f2bf_rne(r1);
    p2 = f2bf_rne(r1 - bf2f_(p1));
}
union Frag16 {
    bf16x8_t v;
    u32x4_t q;
    unsigned short s[8];
};
// 16 bytes = two granules, L1 bypassed like load_granule's sc1 polls (each 8-byte half is one peer store: no tearing)
__device__ __forceinline__ void load_run64(const u64_t* src, u32x4_t (&g)[4]) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
        "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
        "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3])
        : "v"(src)
        : "memory");
}

// weight-only preparation (i2l_decoder_prepare, I2L_PREP_WEIGHTS): every (member, wave, lane)'s 36 operand fragments --
// 24 of W_hh^T (8 k-steps x 3 pieces), 12 of W_out^T (4 k-steps x 3 pieces) -- split once and stored lane-contiguously, so
// that the kernel's prologue is 36 coalesced 16-byte loads per lane instead of 192 strided 4-byte loads + 192 splits
// (60 us of every launch, during which the next batch's encoder waits for the decode to report itself resident)
__global__ __launch_bounds__(64) void pack16_kernel(const float* __restrict__ WhhT, const float* __restrict__ WoutT,
                                                    u32x4_t* __restrict__ pack) {
    const int m = blockIdx.x >> 2, wave = blockIdx.x & 3, lane = threadIdx.x;
    const int lhi = lane >> 4, lcol = lane & 15, ltile = wave & 1, lhalf = wave >> 1;
    u32x4_t* dst = pack + (size_t)blockIdx.x * 36 * 64 + lane;
    const float* src = WhhT + 64 * m + 16 * wave + lcol;
    for (int ks = 0; ks < 8; ++ks) {
        Frag16 f0, f1, f2;
#pragma unroll
        for (int j = 0; j < 8; ++j) split3(src[(size_t)(32 * ks + 8 * lhi + j) * 1024], f0.s[j], f1.s[j], f2.s[j]);
        dst[(ks * 3 + 0) * 64] = f0.q; dst[(ks * 3 + 1) * 64] = f1.q; dst[(ks * 3 + 2) * 64] = f2.q;
    }
    const float* srl = WoutT + 32 * m + 16 * ltile + lcol;
    for (int ks = 0; ks < 4; ++ks) {
        Frag16 f0, f1, f2;
#pragma unroll
        for (int j = 0; j < 8; ++j) split3(srl[(size_t)(32 * (4 * lhalf + ks) + 8 * lhi + j) * 512], f0.s[j], f1.s[j], f2.s[j]);
        dst[(24 + ks * 3 + 0) * 64] = f0.q; dst[(24 + ks * 3 + 1) * 64] = f1.q; dst[(24 + ks * 3 + 2) * 64] = f2.q;
    }
}

// two runs at once: both tasks of a thread in flight together (one L2 round trip instead of two)
__device__ __forceinline__ void load_run64x2(const u64_t* s0, const u64_t* s1, u32x4_t (&g)[8]) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
        "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
        "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
        "global_load_dwordx4 %4, %9, off sc1\n\t"
        "global_load_dwordx4 %5, %9, off offset:16 sc1\n\t"
        "global_load_dwordx4 %6, %9, off offset:32 sc1\n\t"
        "global_load_dwordx4 %7, %9, off offset:48 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7])
        : "v"(s0), "v"(s1)
        : "memory");
}
// 8 granules {tag 16 | p0 | p1 | p2} of one (k block, row) -> the three 16-byte operand rows
__device__ __forceinline__ void unpack_run(const u32x4_t* g, u32x4_t* dst) {
    u32x4_t o0, o1, o2;                 // granule = {low word: p1 << 16 | p2, high word: tag << 16 | p0}
    o0.x = (g[0].y & 0xFFFFu) | (g[0].w << 16); o0.y = (g[1].y & 0xFFFFu) | (g[1].w << 16);
    o0.z = (g[2].y & 0xFFFFu) | (g[2].w << 16); o0.w = (g[3].y & 0xFFFFu) | (g[3].w << 16);
    o1.x = (g[0].x >> 16) | (g[0].z & 0xFFFF0000u); o1.y = (g[1].x >> 16) | (g[1].z & 0xFFFF0000u);
    o1.z = (g[2].x >> 16) | (g[2].z & 0xFFFF0000u); o1.w = (g[3].x >> 16) | (g[3].z & 0xFFFF0000u);
    o2.x = (g[0].x & 0xFFFFu) | (g[0].z << 16); o2.y = (g[1].x & 0xFFFFu) | (g[1].z << 16);
    o2.z = (g[2].x & 0xFFFFu) | (g[2].z << 16); o2.w = (g[3].x & 0xFFFFu) | (g[3].z << 16);
    dst[0] = o0;
    dst[G16_HP_PIECE / 8] = o1;
    dst[2 * (G16_HP_PIECE / 8)] = o2;
}

__global__ __launch_bounds__(G16NT, 2) void decode_group16_kernel(GroupParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    unsigned short* hp = reinterpret_cast<unsigned short*>(smem16);                    // [2][3][32][16][8]
    f32x4_t* lpart = reinterpret_cast<f32x4_t*>(smem16 + (size_t)2 * 3 * G16_HP_PIECE * 2);   // [2 tiles][64 lanes]
    u64_t* redk = reinterpret_cast<u64_t*>(lpart + 2 * 64);                           // [2][16 rows]
    int* cnt_s = reinterpret_cast<int*>(redk + 2 * 16);      // [0..1] logit waves arrived (by parity), [2] time-out, [3] one XCD,
                                                             // [4..5] epoch of the partial logits of tile 0 / 1 in lpart
    const StepWeights& w = p.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int within = blockIdx.x & 127;
    const int group = (blockIdx.x >> 7) * 8 + (within & 7), m = within >> 3;      // a group's members: block ids 8 apart = one XCD
    if (group >= p.n_groups) return;
    const int B = p.B, T = p.T, V = w.V;
    const int row0 = group * G16Q;
    constexpr int G = 1024;
    const int lq = lane & 3, uq = (lane >> 2) & 3, lhi = lane >> 4, lcol = lane & 15;
    const int cell_row = 4 * lhi + lq;                       // the row of this lane's cell (after the quad transposition)
    const int unit = 16 * m + 4 * wave + uq;                 // ... and its hidden unit
    const int ltile = wave & 1, lhalf = wave >> 1;           // logits: column tile, half of the reduction

    // ---- placement exchange FIRST (r04): a member reports itself before it loads its 147 KB of weights, so that the launch's
    // residency word (count_resident_group) -- which the next batch's encoder stream is waiting for -- is published ~10 us
    // earlier; waves 1-3 start their weight loads meanwhile
    u64_t* xg = p.xchg + (size_t)group * 2 * G16Q * G16GRAN;
    if (tid < 8) cnt_s[tid] = 0;                             // wave 0: in program order before its writes below
    if (wave == 0) {                                        // placement: are the sixteen members on one XCD?
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        if (lane == 0 && !(p.opts.drop_member && m == 3))
            store_granule(xg + (size_t)m * G16GRAN + G16GRAN_X, granule(0xC0DEu, __uint_as_float(xcc)), false);
        const int pq = (lane & 15) + ((lane & 15) >= m ? 1 : 0);
        u64_t pv = 0;
        bool bad = false;
        long long t_start = 0;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            if (lane < 15) { pv = load_granule(xg + (size_t)pq * G16GRAN + G16GRAN_X); ok = (unsigned)(pv >> 32) == 0xC0DEu; }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
                const long long now = (long long)wall_clock64();
                if (t_start == 0) t_start = now;
                else if (now - t_start > p.opts.limit_first) { bad = true; break; }
            }
        }
        const bool all_same = __all(lane >= 15 || (unsigned)pv == xcc);
        if (lane == 0) {
            cnt_s[3] = (all_same && !bad) ? 1 : 0;
            if (bad) cnt_s[2] = 1;
            if (m == 0 && !bad) {
                count_resident_group(p.status, p.n_groups, p.resident_flag, p.resident_value);
                if (all_same && !p.opts.agent_scope) atomicAdd(p.status + GRP_STAT_LOCAL, 1u);
            }
        }
    }
    // ---- the member's weights, pre-split in B-operand layout by pack16_kernel: 36 coalesced 16-byte loads per lane
    bf16x8_t wg[24], wl[12];                                 // [k-step][piece]
    {
        const u32x4_t* pk = static_cast<const u32x4_t*>(w.wpack16) + (size_t)(m * 4 + wave) * 36 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 24; ++i) { Frag16 f; f.q = pk[i * 64]; wg[i] = f.v; }
#pragma unroll
        for (int i = 0; i < 12; ++i) { Frag16 f; f.q = pk[(24 + i) * 64]; wl[i] = f.v; }
    }
    for (int idx = tid; idx < 2 * 3 * G16_HP_PIECE / 2; idx += G16NT) reinterpret_cast<unsigned*>(hp)[idx] = 0u;   // h(-1) = 0
    if (tid < 32) redk[tid] = 0;
    const float4 genc = *reinterpret_cast<const float4*>(w.Genc + (size_t)min(row0 + cell_row, B - 1) * G + 4 * unit);
    const float l_bias = w.boutP[32 * m + 16 * ltile + lcol];
    float c_own = 0.f, h_own = 0.f;
    unsigned fin = 0;
#pragma unroll
    for (int r = 0; r < G16Q; ++r)
        if (row0 + r >= B) fin |= 1u << r;
    const bool own_row = row0 + m < B;
    int32_t* ids_row = (p.ids && own_row) ? p.ids + (size_t)(row0 + m) * T : nullptr;

    __syncthreads();
    const bool local = cnt_s[3] != 0 && !p.opts.agent_scope;

    int mytok = min(max(p.tok0[min(row0 + cell_row, B - 1)], 0), V - 1);
    float4 pvec = *reinterpret_cast<const float4*>(w.P + (size_t)mytok * G + 4 * unit);

    int t = 0;
    bool failed = cnt_s[2] != 0;
#ifdef I2L_GROUP_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long st_last = (long long)wall_clock64();
#define I2L_STAMP16(i) do { const long long n_ = (long long)wall_clock64(); st_acc[i] += n_ - st_last; st_last = n_; } while (0)
#else
#define I2L_STAMP16(i) do { } while (0)
#endif
    for (; !failed; ++t) {
        // ---- A. gates(t) = h(t-1) . WhhT on the matrix cores; on the way every wave polls the 256 candidate granules of
        //         step t-1 -- lane (row l & 15, members 4 (l >> 4) .. +3) -- and merges them into the 16 tokens
        f32x4_t acc[3];                                     // hi | mid | lo partial products
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const u64_t* cand_src = xg + (size_t)((t - 1) & 1) * G16Q * G16GRAN + (size_t)(4 * lhi) * G16GRAN + G16GRAN_C + lcol;
        const unsigned c_epoch = (unsigned)t & 0xFFFFu;
        bool have = t == 0;
        int tk[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) tk[r] = 0;
        u64_t gv[4] = {0, 0, 0, 0};
        auto issue = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = load_granule(cand_src + (size_t)i * G16GRAN);
        };
        auto check = [&]() {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) ok = ok && (unsigned)(gv[i] >> 48) == c_epoch;
            if (!__all(ok)) return;
            u64_t best = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                best = umax64(best, am_key(__uint_as_float((unsigned)gv[i]), (int)((unsigned)(gv[i] >> 32) & 0xFFFFu)));
            best = umax64(best, (u64_t)__shfl_xor((unsigned long long)best, 16));
            best = umax64(best, (u64_t)__shfl_xor((unsigned long long)best, 32));
            const int bi = am_idx(best);                    // every lane: the token of row l & 15
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tk[r] = __builtin_amdgcn_readlane(bi, r);
                tk[r] = tk[r] < V ? tk[r] : 0;
            }
            have = true;
            if (t < T) {
                mytok = __shfl(bi, cell_row);
                mytok = mytok < V ? mytok : 0;
                pvec = *reinterpret_cast<const float4*>(w.P + (size_t)mytok * G + 4 * unit);
            }
        };
        if (t > 0 && t < T) {
            const u32x4_t* hq = reinterpret_cast<const u32x4_t*>(hp + (size_t)((t - 1) & 1) * 3 * G16_HP_PIECE) + lhi * 16 + lcol;
            issue();
            Frag16 fa[2][3];                                // A fragments (rows l & 15, k = 32 ks + 8 (l >> 4) .. +7), one k-step ahead
#pragma unroll
            for (int c = 0; c < 3; ++c) fa[0][c].q = hq[(c * 32) * 16];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) fa[(ks + 1) & 1][c].q = hq[(c * 32 + 4 * (ks + 1)) * 16];
                }
                if (ks == 6) { check(); if (!have) issue(); }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8_t a0 = fa[ks & 1][0].v, a1 = fa[ks & 1][1].v, a2 = fa[ks & 1][2].v;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wg[ks * 3 + 0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wg[ks * 3 + 1], acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wg[ks * 3 + 0], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wg[ks * 3 + 2], acc[2], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wg[ks * 3 + 1], acc[2], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, wg[ks * 3 + 0], acc[2], 0, 0, 0);
            }
            if (!have) check();
        }
        I2L_STAMP16(0);
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
            for (int r = 0; r < G16Q; ++r) {
                const bool was_fin = (fin >> r) & 1u;
                if (r == m && tid == 0 && ids_row) ids_row[t - 1] = (p.stop == I2L_STOP_STICKY && was_fin) ? -1 : tk[r];
                if (tk[r] == p.end_id) fin |= 1u << r;
                all_fin = all_fin && ((fin >> r) & 1u);
            }
            if (t == T || (p.stop == I2L_STOP_STICKY && all_fin)) break;
        }
        I2L_STAMP16(1);
        // ---- C. 4 x 4 transposition inside lane quads, cell, publish h(t) as its three bf16 pieces
        const unsigned epoch = ((unsigned)t + 1u) & 0xFFFFu;
        const int par = t & 1;
        u64_t* slot = xg + (size_t)par * G16Q * G16GRAN;
        unsigned short* hcur = hp + (size_t)par * 3 * G16_HP_PIECE;
        if (!bail) {
            const bool b0 = lq & 1, b1 = lq & 2;
            // D registers: z[i] = gate (l & 3) of unit uq, row 4 (l >> 4) + i.  small terms first
            float z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = (acc[2][i] + acc[1][i]) + acc[0][i];
            // stage 1 (lane ^ 1): even lanes send z[1], z[3]; odd lanes send z[0], z[2]
            const float r0 = dpp_f<DPP_XOR1>(b0 ? z[0] : z[1]), r1 = dpp_f<DPP_XOR1>(b0 ? z[2] : z[3]);
            const float e0 = b0 ? r0 : z[0], e1 = b0 ? z[1] : r0, e2 = b0 ? r1 : z[2], e3 = b0 ? z[3] : r1;
            // stage 2 (lane ^ 2): lanes 0, 1 send e2, e3; lanes 2, 3 send e0, e1
            const float s0 = dpp_f<DPP_XOR2>(b1 ? e0 : e2), s1 = dpp_f<DPP_XOR2>(b1 ? e1 : e3);
            const float gi = b1 ? s0 : e0, gf = b1 ? s1 : e1, gc = b1 ? e2 : s0, go = b1 ? e3 : s1;   // gates of (unit, row cell_row)
            const float xi = (gi + genc.x) + pvec.x, xf = (gf + genc.y) + pvec.y;
            const float xc = (gc + genc.z) + pvec.z, xo = (go + genc.w) + pvec.w;
            const float ig = sigmoidf_(xi), fg = sigmoidf_(xf), gg = tanhf_(xc), og = sigmoidf_(xo);
            c_own = fg * c_own + ig * gg;
            h_own = og * tanhf_(c_own);
            unsigned short q0, q1, q2;
            split3(h_own, q0, q1, q2);
            const int ul = 4 * wave + uq;                   // unit inside the member: k block ul >> 3, element ul & 7
            const u64_t gran = ((u64_t)epoch << 48) | ((u64_t)q0 << 32) | ((u64_t)q1 << 16) | (u64_t)q2;
            store_granule(slot + (size_t)m * G16GRAN + ((ul >> 3) * 16 + cell_row) * 8 + (ul & 7), gran, local);
            const int o = ((2 * m + (ul >> 3)) * 16 + cell_row) * 8 + (ul & 7);
            hcur[o] = q0;
            hcur[G16_HP_PIECE + o] = q1;
            hcur[2 * G16_HP_PIECE + o] = q2;
        }
        I2L_STAMP16(2);
        // ---- D. the other fifteen sixteenths of h(t): task = (peer, k block, row) = 8 granules = 64 bytes; a thread's two
        //         tasks (tid, tid + 256 < 480) are polled TOGETHER
        if (!bail) {
            const int ta = tid, tb = tid + G16NT < 15 * 32 ? tid + G16NT : tid;         // threads 224 .. 255: one task (twice)
            const int qa_ = ta >> 5, qa = qa_ + (qa_ >= m ? 1 : 0), qb_ = tb >> 5, qb = qb_ + (qb_ >= m ? 1 : 0);
            const u64_t* sa = slot + (size_t)qa * G16GRAN + (((ta >> 4) & 1) * 16 + (ta & 15)) * 8;
            const u64_t* sb = slot + (size_t)qb * G16GRAN + (((tb >> 4) & 1) * 16 + (tb & 15)) * 8;
            u32x4_t g[8];
            long long t_start = 0;
            unsigned spins = 0;
            for (;;) {
                load_run64x2(sa, sb, g);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; ++i) ok = ok && (g[i].y >> 16) == epoch && (g[i].w >> 16) == epoch;
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0) {
                    const long long now = (long long)wall_clock64();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > p.opts.limit_step) { cnt_s[2] = 1; break; }
                }
            }
            unpack_run(g, reinterpret_cast<u32x4_t*>(hcur) + (2 * qa + ((ta >> 4) & 1)) * 16 + (ta & 15));
            unpack_run(g + 4, reinterpret_cast<u32x4_t*>(hcur) + (2 * qb + ((tb >> 4) & 1)) * 16 + (tb & 15));
        }
        I2L_STAMP16(3);
        __syncthreads();                                    // THE barrier of the step: h(t) complete in hcur
        if (cnt_s[2] != 0) { failed = true; break; }
        I2L_STAMP16(4);

        // ---- E. logits(t): each wave one column tile over half of k; waves 2, 3 pass their partial sums to waves 0, 1
        {
            f32x4_t la[3] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
            const u32x4_t* hq = reinterpret_cast<const u32x4_t*>(hcur) + lhi * 16 + lcol;
            Frag16 fa[4][3];                                // all four k-steps' fragments requested at once (12 reads in flight)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int c = 0; c < 3; ++c) fa[ks][c].q = hq[(c * 32 + 4 * (4 * lhalf + ks)) * 16];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t a0 = fa[ks][0].v, a1 = fa[ks][1].v, a2 = fa[ks][2].v;
                la[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wl[ks * 3 + 0], la[0], 0, 0, 0);
                la[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wl[ks * 3 + 1], la[1], 0, 0, 0);
                la[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wl[ks * 3 + 0], la[1], 0, 0, 0);
                la[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wl[ks * 3 + 2], la[2], 0, 0, 0);
                la[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wl[ks * 3 + 1], la[2], 0, 0, 0);
                la[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, wl[ks * 3 + 0], la[2], 0, 0, 0);
            }
            f32x4_t part;
#pragma unroll
            for (int i = 0; i < 4; ++i) part[i] = (la[2][i] + la[1][i]) + la[0][i];
            if (lhalf == 1) {                               // waves 2, 3: LDS write, then the flag (LDS keeps a wave's order)
                lpart[ltile * 64 + lane] = part;
                if (lane == 0) __hip_atomic_store(cnt_s + 4 + ltile, (int)epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                I2L_STAMP16(5);
            } else {                                        // waves 0, 1: the other half, bias, arg max, candidates (exchange 2)
                while (__hip_atomic_load(cnt_s + 4 + ltile, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (int)epoch)
                    __builtin_amdgcn_s_sleep(0);
                const f32x4_t other = lpart[ltile * 64 + lane];
                I2L_STAMP16(5);
                const int l_v = 32 * m + 16 * ltile + lcol; // this lane's vocabulary column; register i = row 4 (l >> 4) + i
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lv = (part[i] + other[i]) + l_bias;
                    if (p.use_temp) lv = lv / p.temperature;
                    // arg max over the 16 lanes (= columns) of a row in 32-bit steps: the value's order-preserving integer image
                    // (am_key's high word) by DPP max, then the FIRST lane that holds it (lowest column wins a tie, as am_key)
                    unsigned vk = __float_as_uint(lv + 0.0f);
                    vk ^= (vk >> 31) ? 0xFFFFFFFFu : 0x80000000u;
                    unsigned mx = vk;
                    mx = max(mx, (unsigned)__builtin_amdgcn_mov_dpp((int)mx, DPP_ROR8, 0xF, 0xF, true));
                    mx = max(mx, (unsigned)__builtin_amdgcn_mov_dpp((int)mx, DPP_ROR4, 0xF, 0xF, true));
                    mx = max(mx, (unsigned)__builtin_amdgcn_mov_dpp((int)mx, DPP_XOR2, 0xF, 0xF, true));
                    mx = max(mx, (unsigned)__builtin_amdgcn_mov_dpp((int)mx, DPP_XOR1, 0xF, 0xF, true));
                    const unsigned long long hit = __ballot(vk == mx);
                    const int first = __ffs((unsigned)(hit >> (16 * lhi)) & 0xFFFFu) - 1;
                    const u64_t key = ((u64_t)mx << 32) | (u64_t)(0xFFFFFFFFu - (unsigned)(l_v - lcol + first));
                    if (lcol == 0) atomicMax(reinterpret_cast<unsigned long long*>(redk + par * 16 + 4 * lhi + i), (unsigned long long)key);
                }
                int arrived = 0;
                if (lane == 0) arrived = atomicAdd(cnt_s + par, 1);
                arrived = __builtin_amdgcn_readfirstlane(arrived);
                if (arrived == 1) {                         // the later of the two publishes
                    if (lane < 16) {
                        const u64_t best = redk[par * 16 + lane];
                        store_granule(slot + (size_t)m * G16GRAN + G16GRAN_C + lane,
                                      granule((epoch << 16) | (unsigned)(am_idx(best) & 0xFFFF), am_val(best)), local);
                        redk[(par ^ 1) * 16 + lane] = 0;
                    }
                    if (lane == 0) cnt_s[par ^ 1] = 0;
                }
                I2L_STAMP16(6);
            }
        }
    }
#ifdef I2L_GROUP_STAMPS
    if (lane == 0 && (wave & 1) == 0 && blockIdx.x < 16)
        for (int i = 0; i < 8; ++i) p.status[8 + blockIdx.x * 16 + (wave >> 1) * 8 + i] = (unsigned)st_acc[i];
#endif
#undef I2L_STAMP16
    if (failed) {
        if (lane == 0) atomicOr(p.status, 1u);
        if (ids_row) for (int tt = tid; tt < T; tt += G16NT) ids_row[tt] = -3;
        return;
    }
    if (ids_row) for (int tt = t + tid; tt < T; tt += G16NT) ids_row[tt] = -1;
}
