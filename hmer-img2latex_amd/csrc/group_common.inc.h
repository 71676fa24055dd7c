// Helpers shared by the grouped persistent kernels (decode_group.inc.h, train_group.inc.h): tagged-granule
// exchange between the workgroups of a group, DPP reduce-scatter, packed-FMA tiles, arg-max keys.
// Included inside each translation unit's anonymous namespace.
#pragma once
typedef unsigned long long u64_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr long long GRP_TIMEOUT_TICKS = 300000000ll;   // 3 s of the 100 MHz wall clock

// Per-launch options of a grouped kernel.  A group only makes progress while its four workgroups are resident
// together; the launch is not cooperative, so every poll has a wall-clock limit.  Two limits: the FIRST poll of a
// workgroup (the placement exchange: "are my three peers there at all?") gets a short one -- 10 ms plus what earlier
// groups of the same launch may legitimately hold the CUs for (50 us per step) -- so that on a GPU shared with other
// work the launch gives up in milliseconds and the host falls back to the row-per-workgroup kernel; every later poll
// (peers known to be resident) keeps the 3 s limit.  drop_member is a TEST hook (I2L_FLAG_TEST_DROP_MEMBER): member 3
// of every group exits at once, which forces the time-out path end to end.
struct GroupOpts {
    long long limit_first, limit_step;
    int agent_scope;      // != 0: every exchange store at agent scope (I2L_FLAG_AGENT_SCOPE_EXCHANGE)
    int drop_member;
};
inline GroupOpts group_opts(int steps, int flags) {
    GroupOpts o;
    o.limit_first = 1000000ll + 5000ll * (long long)steps;
    o.limit_step = GRP_TIMEOUT_TICKS;
    if (flags & I2L_FLAG_TEST_SHORT_TIMEOUT) o.limit_first = o.limit_step = 200000ll;      // 2 ms
    o.agent_scope = (flags & I2L_FLAG_AGENT_SCOPE_EXCHANGE) ? 1 : 0;
    o.drop_member = (flags & I2L_FLAG_TEST_DROP_MEMBER) ? 1 : 0;
    return o;
}
// status block of a grouped launch (zeroed by the launch's memset node): [0] != 0 a poll timed out, [1] groups that
// went through the placement exchange, [2] of those, groups whose four members measured ONE XCD (L2-local stores)
constexpr int GRP_STAT_FAILED = 0, GRP_STAT_GROUPS = 1, GRP_STAT_LOCAL = 2;

// Residency signal of a grouped launch (i2l_greedy_decode_ex's resident_flag / resident_value): member 0 of every group
// counts its group in status[GRP_STAT_GROUPS] once all members have answered the placement exchange, i.e. ARE RESIDENT; the
// group that completes the count publishes resident_value at agent scope.  A stream that must not start before this
// launch owns its CUs waits for the word (i2l_stream_wait_value32) -- a dependency, where r03 guessed with a 30 us delay.
__device__ __forceinline__ void count_resident_group(unsigned* status, int n_groups, unsigned* flag, unsigned value) {
    const unsigned before = atomicAdd(status + GRP_STAT_GROUPS, 1u);
    if (flag && before + 1u == (unsigned)n_groups) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}


// local == false: sc1 store (write-through to memory, seen from every XCD).  local == true (all four members were
// found on ONE XCD): sc0 store, the line stays in that XCD's L2 where the peers' sc1 loads (L1 bypassed) find it --
// an L2 round trip instead of a memory one.
// The L2-local flavour rests on gfx950's write-through L1 (a workgroup-scope store still reaches the XCD's L2, where the
// peers' L1-bypassing polls find it), NOT on the HSA memory model: it is compiled for gfx950 only; any other target gets
// the conformant agent-scope store whatever the caller measured.
__device__ __forceinline__ void store_granule(u64_t* g, u64_t v, bool local) {
#if defined(__gfx950__)
    if (local) { __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
#endif
    __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64_t granule(unsigned tag, float v) { return ((u64_t)tag << 32) | (u64_t)__float_as_uint(v); }
__device__ __forceinline__ u64_t load_granule(const u64_t* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// DPP controls: quad_perm [1,0,3,2] (lane ^ 1), quad_perm [2,3,0,1] (lane ^ 2), row_half_mirror (lane -> 7 - lane
// within 8), row_ror:n (rotate within 16), row_shl:n (lane reads lane + n)
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HMIRROR = 0x141, DPP_ROR4 = 0x124, DPP_ROR8 = 0x128,
              DPP_SHL4 = 0x104, DPP_SHL8 = 0x108;
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// One level of a reduce-scatter between a lane and its DPP partner: the pair holds partial sums of the same two
// values (a, b); the lane with sel == false keeps a, its partner keeps b, each adds the other's partial.
template <int CTRL>
__device__ __forceinline__ float rs_level(float a, float b, bool sel) {
    const float keep = sel ? b : a, send = sel ? a : b;
    return keep + dpp_f<CTRL>(send);
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
// acc[4 values][2 row pairs] += w4 (4 values: gates or columns) x h4 (4 rows)
__device__ __forceinline__ void fma_4x4(f32x2 (&acc)[4][2], f32x2 w01, f32x2 w23, float4 hv) {
    const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
    pkfma_lo(acc[0][0], w01, h01); pkfma_lo(acc[0][1], w01, h23);
    pkfma_hi(acc[1][0], w01, h01); pkfma_hi(acc[1][1], w01, h23);
    pkfma_lo(acc[2][0], w23, h01); pkfma_lo(acc[2][1], w23, h23);
    pkfma_hi(acc[3][0], w23, h01); pkfma_hi(acc[3][1], w23, h23);
}

