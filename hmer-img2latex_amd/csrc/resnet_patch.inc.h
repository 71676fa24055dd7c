// 3x3 / stride 1 / pad 1 convolution of the bf16 ResNet trunk with the INPUT PATCH staged in LDS (included by resnet.hip,
// inside its anonymous namespace).  encoder.py:185-249 (torchvision Bottleneck.conv2 / BasicBlock.conv1, conv2).
//
// Why: the implicit-GEMM ring kernel moves 32 KB (A tile + W tile) from L2 into LDS per 64-deep K step of a 128 x 128
// tile -- 64 B per clock and CU at full matrix rate, which is the whole L2 -> LDS path; it sits at 22 - 25 % matrix-pipe
// busy on every 3x3 layer whatever the ring depth (profiles/r04/resnet_patch.txt).  Nine taps read the same pixels, so
// here a workgroup stages the (R + 2) x (W + 2) pixel patch of its R full output rows ONCE per 32-channel slab
// (LDS-DMA, double buffered: slab s + 1 lands while slab s is multiplied) and only the filter slices stream through
// the ring: 8 KB per (tap, slab) stage instead of 32 KB per 64 k -- 2.4x fewer bytes into LDS per MFMA.
//
//   tile   = R consecutive rows of the (B*H) x W pixel grid (R*W <= 32*WM*MTW output pixels, rows may belong to
//            different images) x 32*WN*NTW output channels; 4 waves as WM x WN, each 32*MTW pixels x 32*NTW channels.
//            A 3x3 layer of the trunk is only ~0.7 k MFMAs per SIMD, so the tile SHAPE is picked per layer for balance
//            (patch_plan: the busiest CU's share of MFMAs): 160 x 128 for 8x40 maps (512 equal tiles), 320 x 64 for
//            16x80, 96 x 128 for 2x10 -- with 128 x 128 everywhere the busiest CU had 3 tiles against a mean of 2.67.
//   K loop = for slab (32 input channels): for tap (ky, kx): 2 MFMA k-steps of 16 channels
//   patch  = [(R + 2) * (W + 2) pixels][32 channels] bf16, 64 B per pixel, 16-byte chunk c of pixel q stored at chunk
//            c ^ ((q >> 2) & 3) (conflict-free ds_read_b128 for 16 consecutive pixels); halo columns, rows outside the
//            grid and the tail of the buffer are DMA'd from a zero chunk.  A patch row that belongs to the neighbouring
//            image (tile rows straddle images when H < R) is masked at READ time: the lane reads the zero tail instead.
//   filter = ring of NSW stages [64*NTW channels][32 k] bf16 (same swizzle by row), NSW - 1 stages ahead, one raw
//            barrier per stage, counted vmcnt waits (the patch DMA of the next slab stays in flight across them).
template <int WM, int WN, int MTW, int NTW, int NLPT, int NSW, bool RES>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_bf16_kernel(BfGemm g, int R, int nb_n, int total_tiles) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int PT = 32 * WM * MTW, BN = 32 * WN * NTW;
    constexpr int WST = BN * 64;                     // bytes of a filter stage: BN rows x 32 k x 2 B
    constexpr int NLW = WST / 4096;                  // 16-byte DMA pieces per thread per filter stage
    constexpr int PBYTES = NLPT * 4096;              // one patch buffer
    constexpr int RING = NSW * WST;
    constexpr int CLD = BN + 8, CHUNKS = BN / 8, NOUT = PT * CHUNKS / 256;
    static_assert(NLW >= 1 && PT * CLD * 2 <= RING + 2 * PBYTES, "epilogue tile must fit");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    int t = blockIdx.x;
    if ((total_tiles & 7) == 0) t = (t & 7) * (total_tiles >> 3) + (t >> 3);   // the N tiles of one patch on one XCD
    const int mb = t / nb_n;
    const int n0 = (t - mb * nb_n) * BN;
    const int W_ = g.iW, H_ = g.iH, C = g.Cin, W2 = W_ + 2;
    const int G = g.M / W_;                          // rows of the pixel grid (B * H)
    const int g0 = mb * R;
    const int nq = (R + 2) * W2;
    const int nslab = C >> 5;
    const bf16_t* const zero_chunk = reinterpret_cast<const bf16_t*>(g_zero_chunk);

    // ---- DMA sources.  patch piece i of this thread: chunk idx = i*256 + tid of the buffer = pixel idx>>2, position idx&3
    int poff[NLPT];                                  // element offset of the piece inside x for slab 0, or -1 (zero)
#pragma unroll
    for (int i = 0; i < NLPT; ++i) {
        const int idx = i * 256 + tid, q = idx >> 2, cpos = idx & 3;
        const int pr = q / W2, pc = q - pr * W2 - 1, grow = g0 - 1 + pr;
        const bool in = q < nq && pc >= 0 && pc < W_ && grow >= 0 && grow < G;
        poff[i] = in ? (grow * W_ + pc) * C + ((cpos ^ ((q >> 2) & 3)) << 3) : -1;
    }
    const bf16_t* srcW[NLW];
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int idx = i * 256 + tid, row = idx >> 2, cpos = idx & 3;
        srcW[i] = g.W + (size_t)(n0 + row) * g.ldw + ((cpos ^ ((row >> 2) & 3)) << 3);
    }
    auto issue_patch = [&](int s) {
        unsigned char* base = lds + RING + (s & 1) * PBYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < NLPT; ++i) glds16(poff[i] >= 0 ? g.A + poff[i] + s * 32 : zero_chunk, base + i * 4096);
    };
    int is_s = 0, is_t = 0, is_slot = 0;             // next filter stage to issue
    auto issue_w = [&]() {
        unsigned char* base = lds + is_slot * WST + wave * 1024;
        const int koff = is_t * C + is_s * 32;
#pragma unroll
        for (int i = 0; i < NLW; ++i) glds16(srcW[i] + koff, base + i * 4096);
        if (++is_t == 9) { is_t = 0; ++is_s; }
        if (++is_slot == NSW) is_slot = 0;
    };

    // ---- per-lane A geometry: pixel m of the tile -> patch pixel q0 (tap 0,0), rows that exist inside the image
    int q0[MTW];
    unsigned rowok[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        int m = wm * 32 * MTW + mt * 32 + li;
        if (m >= R * W_) m = 0;                      // slots past the tile: any valid pixel, never stored
        const int r = m / W_, c = m - r * W_;
        q0[mt] = r * W2 + c;
        const int y = (g0 + r) % H_;
        rowok[mt] = (y > 0 ? 1u : 0u) | 2u | (y < H_ - 1 ? 4u : 0u);
    }
    int rowB[NTW], swB[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int row = wn * 32 * NTW + nt * 32 + li;
        rowB[nt] = row * 64;
        swB[nt] = (row >> 2) & 3;
    }

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int total = nslab * 9;
    issue_patch(0);
#pragma unroll
    for (int s = 0; s < NSW - 1; ++s)
        if (s < total) issue_w();

    int slot = 0, j = 0;
    for (int s = 0; s < nslab; ++s) {
        const unsigned pbase = RING + (s & 1) * PBYTES;
        const unsigned zoff = pbase + PBYTES - 64;   // the buffer's tail: zeros
        const bool more = s + 1 < nslab;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++j) {
            // loads issued after stage j's: NSW - 2 filter stages, and the next slab's patch while it is younger than them
            if (j + NSW - 1 > total) wait_vm<0>();                                   // tail: nothing more was issued
            else if (more && tap >= 1 && tap <= NSW - 1) wait_vm<(NSW - 2) * NLW + NLPT>();
            else wait_vm<(NSW - 2) * NLW>();
            __builtin_amdgcn_s_barrier();            // stage j (and at tap 0 the slab's patch) landed; slot j-1 is free
            if (j + NSW - 1 < total) issue_w();
            if (tap == 0 && more) issue_patch(s + 1);        // its buffer was last read in slab s-1
            const int ky = tap / 3, kx = tap - 3 * (tap / 3);
            const int tq = ky * W2 + kx;
            const unsigned wbase = slot * WST;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[MTW], b[NTW];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const int q = q0[mt] + tq;
                    const unsigned ad = ((rowok[mt] >> ky) & 1u) ? pbase + q * 64 + (((2 * ks + lh) ^ ((q >> 2) & 3)) << 4) : zoff;
                    a[mt] = *reinterpret_cast<const bf16x8*>(lds + ad);
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    b[nt] = *reinterpret_cast<const bf16x8*>(lds + wbase + rowB[nt] + (((2 * ks + lh) ^ swB[nt]) << 4));
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
            if (++slot == NSW) slot = 0;
        }
    }
    __syncthreads();                                 // every wave is done with the ring and the patches
    // epilogue: scaled tile staged in LDS as bf16, then whole pixel rows at 16 bytes per lane
    bf16_t* Cs = reinterpret_cast<bf16_t*>(lds);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int nl = wn * 32 * NTW + nt * 32 + li;
        const float sc = g.scale ? g.scale[n0 + nl] : 1.f, bi = g.bias ? g.bias[n0 + nl] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 32 * MTW + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[mt][nt][r] * sc + bi;
                if (g.relu && !RES) v = fmaxf(v, 0.f);
                Cs[ml * CLD + nl] = f2bf(v);
            }
    }
    __syncthreads();
    const int mvalid = min(R * W_, g.M - g0 * W_);   // pixels of this tile that exist
#pragma unroll
    for (int jo = 0; jo < NOUT; ++jo) {
        const int idx = tid + jo * 256;
        const int ml = idx / CHUNKS, c8 = (idx - ml * CHUNKS) * 8;
        if (ml >= mvalid) continue;
        const size_t pix = (size_t)g0 * W_ + ml;
        uint4 v = *reinterpret_cast<const uint4*>(&Cs[ml * CLD + c8]);
        if constexpr (RES) {
            const uint4 rr = *reinterpret_cast<const uint4*>(g.res + pix * g.ldr + n0 + c8);
            bf16_t* hv = reinterpret_cast<bf16_t*>(&v);
            const bf16_t* hr = reinterpret_cast<const bf16_t*>(&rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = bf2f(hv[e]) + bf2f(hr[e]);
                if (g.relu) f = fmaxf(f, 0.f);
                hv[e] = f2bf(f);
            }
        }
        *reinterpret_cast<uint4*>(g.C + pix * g.ldc + n0 + c8) = v;
    }
}

// ---- host side: which tile shape, how many rows per tile
struct PatchShape { int wm, wn, mtw, ntw; };
constexpr PatchShape PATCH_SHAPES[] = {
    {1, 4, 5, 1},   // 160 pixels x 128 channels
    {1, 4, 3, 1},   //  96 x 128
    {2, 2, 2, 2},   // 128 x 128
    {2, 2, 5, 1},   // 320 x 64
    {2, 2, 4, 1},   // 256 x 64
};
struct PatchPlan { int shape, R, nlpt, nsw; };     // shape < 0: the patch kernel does not apply
// Cost of a shape = MFMAs of the busiest CU (tiles are dealt round-robin, every tile of a layer costs the same): the layers
// are small enough (one to five tiles per CU) that this, not bytes, decides (profiles/r04/resnet_patch.txt).  `force` > 0
// picks shape force-1 (A/B runs and tests: I2L_FLAG_RESNET_PATCH_SHAPE(n)).
inline PatchPlan patch_plan(long M, int H, int W, int Cin, int Cout, int n_cu, int force) {
    PatchPlan best{-1, 0, 0, 0};
    if (H < 1 || W < 1 || Cin % 32 != 0 || Cout % 64 != 0 || M * Cin > 0x7fffffffL || M * Cout > 0x7fffffffL) return best;
    // measured (profiles/r04/resnet_patch.txt): ahead of the implicit-GEMM ring kernel for 64 / 128 input channels (2 - 4 slabs
    // per tile: 44.5 -> 41 us and 44.3 -> 34 - 37 us at B=256), level at 256 and behind at 512 (8 / 16 slabs, one barrier per
    // 6 - 10 MFMAs on a single wave per SIMD) -- there the ring kernel stays unless a shape is forced
    if (force == 0 && Cin > 128) return best;
    const long G = M / W;
    long best_cost = 0, best_tiles = 0;
    for (int i = 0; i < (int)(sizeof(PATCH_SHAPES) / sizeof(PATCH_SHAPES[0])); ++i) {
        if (force > 0 && i != force - 1) continue;
        const PatchShape& sh = PATCH_SHAPES[i];
        const int PT = 32 * sh.wm * sh.mtw, BN = 32 * sh.wn * sh.ntw;
        if (Cout % BN != 0 || W > PT) continue;
        const int R = PT / W;
        const int need = ((R + 2) * (W + 2) * 64 + 64 + 4095) / 4096;      // patch + its zero tail, in 4 KB DMA rounds
        const int nlpt = need <= 4 ? 4 : need <= 8 ? 8 : need <= 12 ? 12 : 0;
        if (!nlpt || (nlpt == 12 && BN != 64) || (nlpt == 4 && sh.mtw == 4)) continue;   // (instantiated combinations)
        // ring depth: 6 stages when two workgroups still fit a CU (80 KB each), else 4
        const int wst = BN * 64;
        const int nsw = (6 * wst + 2 * nlpt * 4096 <= 80 * 1024) ? 6 : 4;
        const long tiles = ((G + R - 1) / R) * (Cout / BN);
        const long cost = ((tiles + n_cu - 1) / n_cu) * sh.mtw * sh.ntw;
        if (best.shape < 0 || cost < best_cost || (cost == best_cost && tiles > best_tiles)) {
            best = PatchPlan{i, R, nlpt, nsw};
            best_cost = cost;
            best_tiles = tiles;
        }
    }
    return best;
}

template <int WM, int WN, int MTW, int NTW, int NLPT, int NSW, bool RES>
int launch_patch4(const BfGemm& g, int R, int nb_n, int total, hipStream_t s) {
    constexpr int lds = NSW * 32 * WN * NTW * 64 + 2 * NLPT * 4096;
    static std::atomic<unsigned> attr{0};
    if (!i2l_lds_attr(reinterpret_cast<const void*>(conv3x3_patch_bf16_kernel<WM, WN, MTW, NTW, NLPT, NSW, RES>), lds, attr))
        return I2L_ERR_LAUNCH;
    hipLaunchKernelGGL((conv3x3_patch_bf16_kernel<WM, WN, MTW, NTW, NLPT, NSW, RES>), dim3(total), dim3(256), lds, s, g, R, nb_n, total);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
template <int WM, int WN, int MTW, int NTW, int NLPT, int NSW>
int launch_patch3(const BfGemm& g, int R, int nb_n, int total, hipStream_t s) {
    return g.res ? launch_patch4<WM, WN, MTW, NTW, NLPT, NSW, true>(g, R, nb_n, total, s)
                 : launch_patch4<WM, WN, MTW, NTW, NLPT, NSW, false>(g, R, nb_n, total, s);
}
inline int launch_patch(const BfGemm& g, const PatchPlan& p, hipStream_t s) {
    const PatchShape& sh = PATCH_SHAPES[p.shape];
    const long G = g.M / g.iW;
    const int nb_n = g.N / (32 * sh.wn * sh.ntw);
    const long total = ((G + p.R - 1) / p.R) * nb_n;
    if (total > 0x7fffffff) return I2L_ERR_UNSUPPORTED;
    const int key = p.shape * 100 + p.nlpt * 10 + p.nsw / 2 - 2;      // nsw 4 / 6 -> 0 / 1
#define I2L_PATCH_CASE(i, wm, wn, mtw, ntw, nl, ns) \
    case (i) * 100 + (nl) * 10 + (ns) / 2 - 2: return launch_patch3<wm, wn, mtw, ntw, nl, ns>(g, p.R, nb_n, (int)total, s);
    switch (key) {
        I2L_PATCH_CASE(0, 1, 4, 5, 1, 4, 6) I2L_PATCH_CASE(0, 1, 4, 5, 1, 8, 4)
        I2L_PATCH_CASE(1, 1, 4, 3, 1, 4, 6) I2L_PATCH_CASE(1, 1, 4, 3, 1, 8, 4)
        I2L_PATCH_CASE(2, 2, 2, 2, 2, 4, 6) I2L_PATCH_CASE(2, 2, 2, 2, 2, 8, 4)
        I2L_PATCH_CASE(3, 2, 2, 5, 1, 4, 6) I2L_PATCH_CASE(3, 2, 2, 5, 1, 8, 4) I2L_PATCH_CASE(3, 2, 2, 5, 1, 12, 4)
        I2L_PATCH_CASE(4, 2, 2, 4, 1, 8, 4) I2L_PATCH_CASE(4, 2, 2, 4, 1, 12, 4)
        default: return I2L_ERR_UNSUPPORTED;
    }
#undef I2L_PATCH_CASE
}
