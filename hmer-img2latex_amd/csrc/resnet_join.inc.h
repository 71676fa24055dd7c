// Bottleneck tail + next head in one launch (included by resnet.hip inside its anonymous namespace; r04):
//   Y = relu( bn3(conv3(o2)) + identity )          1x1, 64 -> 256 channels   (encoder.py:185-249, torchvision Bottleneck.forward)
//   Z = relu( bn1'(conv1'(Y)) )                    1x1, 256 -> 64 | 128       (the NEXT block's first conv)
// Layer1's blocks at B = 256 x 16 x 80 positions are bound by HBM, not by the matrix cores (4.8 - 5.3 TB/s): conv3 reads
// o2 (42 MB) and the identity (168 MB) and writes Y (168 MB), then conv1' reads Y again (168 MB) for a 42 MB output.  Here
// Y's tile is completed in LDS (the same roundings as the two-launch path: scaled tile -> bf16, + identity in fp32, ReLU,
// -> bf16), written out ONCE -- the next block still needs it as its identity -- and multiplied by conv1' straight from
// LDS: 573 -> 410 MB per pair.  The 512 / 1024 / 2048-channel tiles of layers 2 - 4 do not fit beside their filters; those
// blocks keep two launches.
// One PERSISTENT workgroup per CU keeps both filters in LDS (32 + 32 | 64 KB, loaded once) and walks 64-position tiles; the
// next tile's o2 rows (LDS-DMA, double buffered) and identity rows (registers) are requested before the current tile's
// GEMMs, so HBM reads, MFMAs and the Y / Z stores of neighbouring tiles overlap.  (A first version -- one 128-row tile per
// workgroup, everything loaded, then computed, then stored -- ran the pair in ~140 us against 117 for the two launches.)
//   LDS: W1 = conv3 filter [256][64], W2 = conv1' filter as 4 K tiles [N2][64] (16-byte chunks XOR-swizzled by row as in the
//        ring kernel), A1[2] = o2 tiles [64][64], inter = Y tile [64][264] bf16 (row stride 528 B: conflict-free
//        ds_read_b128 down a column of rows; reused to stage Z)
struct JoinArgs {
    const bf16_t* A1; const bf16_t* W1; const float* scale1; const float* bias1;
    const bf16_t* res; bf16_t* Y;
    const bf16_t* W2; const float* scale2; const float* bias2; bf16_t* Z;
    int M, n_tiles;
};
constexpr int JN_ILD = 264;                          // inter row stride in bf16 elements
constexpr int JN_TM = 64;                            // positions per tile
constexpr int JN_W1 = 32 * 1024, JN_A1 = JN_TM * 128, JN_INTER = JN_TM * JN_ILD * 2;
constexpr int join_lds_bytes(int n2) { return JN_W1 + n2 * 512 + 2 * JN_A1 + JN_INTER; }

template <int N2T>                                   // N2 = 64 * N2T output channels of conv1'
__global__ __launch_bounds__(256, 1) void bottleneck_join_kernel(JoinArgs g) {
    constexpr int N2 = 64 * N2T, W2B = N2 * 512;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* const w1s = lds;
    unsigned char* const w2s = lds + JN_W1;
    unsigned char* const a1s = w2s + W2B;            // two buffers of JN_A1 bytes
    bf16_t* const inter = reinterpret_cast<bf16_t*>(a1s + 2 * JN_A1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // ---- the two filters, once
#pragma unroll
    for (int i = 0; i < 8; ++i) {                    // conv3: 256 rows x 8 chunks
        const int s = i * 256 + tid, row = s >> 3, c = (s & 7) ^ ring_swz<64>(row);
        glds16(g.W1 + (size_t)row * 64 + c * 8, w1s + wave * 1024 + i * 4096);
    }
#pragma unroll
    for (int i = 0; i < 8 * N2T; ++i) {              // conv1': K tile kt = [N2 rows][8 chunks]
        const int s = i * 256 + tid, kt = s / (N2 * 8), r = s - kt * (N2 * 8), row = r >> 3, c = (r & 7) ^ ring_swz<64>(row);
        glds16(g.W2 + (size_t)row * 256 + kt * 64 + c * 8, w2s + wave * 1024 + i * 4096);
    }
    uint4 rcur[8], rnext[8];                         // identity rows of the current / next tile: 64 rows x 32 chunks
    auto request = [&](int tile, int buf, uint4 (&rr)[8]) {
        const int m0 = tile * JN_TM;
#pragma unroll
        for (int i = 0; i < 2; ++i) {                // o2 tile: 64 rows x 8 chunks
            const int s = i * 256 + tid, row = s >> 3, c = (s & 7) ^ ring_swz<64>(row);
            glds16(g.A1 + (size_t)min(m0 + row, g.M - 1) * 64 + c * 8, a1s + buf * JN_A1 + wave * 1024 + i * 4096);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + j * 256, ml = idx >> 5, c8 = (idx & 31) * 8;
            rr[j] = *reinterpret_cast<const uint4*>(g.res + (size_t)min(m0 + ml, g.M - 1) * 256 + c8);
        }
    };
    int tile = blockIdx.x, buf = 0;
    if (tile < g.n_tiles) request(tile, 0, rcur);
    // per-wave constants of the two GEMMs
    const int n1base = wave * 64;                    // GEMM 1: wave = all 64 rows x 64 of the 256 columns
    constexpr int NT2 = N2T == 2 ? 2 : 1;            // GEMM 2: 2 x 2 waves, wave = 32 rows x (32 * NT2) columns
    const int r2base = (wave >> 1) * 32, c2base = (wave & 1) * 32 * NT2;
    float sc1[2], bi1[2], sc2[NT2], bi2[NT2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { sc1[nt] = g.scale1[n1base + nt * 32 + li]; bi1[nt] = g.bias1[n1base + nt * 32 + li]; }
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) { sc2[nt] = g.scale2[c2base + nt * 32 + li]; bi2[nt] = g.bias2[c2base + nt * 32 + li]; }

    for (; tile < g.n_tiles; tile += gridDim.x) {
        const int m0 = tile * JN_TM;
        wait_vm<0>();                                // this tile's o2 rows and identity (and, the first time, the filters)
        __syncthreads();                             // ... for every wave; the previous tile is out of `inter`
        const int nxt = tile + gridDim.x;
        if (nxt < g.n_tiles) request(nxt, buf ^ 1, rnext);
        const unsigned char* a1 = a1s + buf * JN_A1;

        // ---- GEMM 1: [64 x 64] x [64 x 256]
        {
            f32x16 acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const int row = x * 32 + li;
                    a[x] = *reinterpret_cast<const bf16x8*>(a1 + row * 128 + (((2 * ks + lh) ^ ring_swz<64>(row)) << 4));
                    const int n = n1base + x * 32 + li;
                    b[x] = *reinterpret_cast<const bf16x8*>(w1s + n * 128 + (((2 * ks + lh) ^ ring_swz<64>(n)) << 4));
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        inter[ml * JN_ILD + n1base + nt * 32 + li] = f2bf(acc[mt][nt][r] * sc1[nt] + bi1[nt]);
                    }
        }
        __syncthreads();
        // ---- + identity, ReLU: the block's output, to HBM once and back into `inter` as GEMM 2's A operand
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + j * 256, ml = idx >> 5, c8 = (idx & 31) * 8;
            uint4 v = *reinterpret_cast<const uint4*>(&inter[ml * JN_ILD + c8]);
            bf16_t* hv = reinterpret_cast<bf16_t*>(&v);
            const bf16_t* hr = reinterpret_cast<const bf16_t*>(&rcur[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = f2bf(fmaxf(bf2f(hv[e]) + bf2f(hr[e]), 0.f));
            *reinterpret_cast<uint4*>(&inter[ml * JN_ILD + c8]) = v;
            if (m0 + ml < g.M) *reinterpret_cast<uint4*>(g.Y + (size_t)(m0 + ml) * 256 + c8) = v;
        }
        __syncthreads();
        // ---- GEMM 2: [64 x 256] (LDS) x [256 x N2]
        f32x16 acc2[NT2];
#pragma unroll
        for (int b = 0; b < NT2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[b][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&inter[(r2base + li) * JN_ILD + kt * 64 + ks * 16 + lh * 8]);
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) {
                    const int n = c2base + nt * 32 + li;
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(w2s + kt * (N2 * 128) + n * 128 + (((2 * ks + lh) ^ ring_swz<64>(n)) << 4));
                    acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2[nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();                             // every wave has read its rows of `inter`: Z is staged there
        constexpr int ZLD = N2 + 8, ZCH = N2 / 8;
        bf16_t* Zs = inter;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = r2base + (r & 3) + 8 * (r >> 2) + 4 * lh;
                Zs[ml * ZLD + c2base + nt * 32 + li] = f2bf(fmaxf(acc2[nt][r] * sc2[nt] + bi2[nt], 0.f));
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < JN_TM * ZCH / 256; ++j) {
            const int idx = tid + j * 256, ml = idx / ZCH, c8 = (idx - ml * ZCH) * 8;
            if (m0 + ml < g.M)
                *reinterpret_cast<uint4*>(g.Z + (size_t)(m0 + ml) * N2 + c8) = *reinterpret_cast<const uint4*>(&Zs[ml * ZLD + c8]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) rcur[j] = rnext[j];
        buf ^= 1;
    }
}
