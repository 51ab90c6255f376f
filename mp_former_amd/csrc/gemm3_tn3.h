// 192 x 256 output tiles for the fp16 x 2 TN products with 256 output columns (round 5) — included by gemm3.hip inside its namespace.
//
//   C[M, 256 t .. 256 t + 255] = A[M, K] . B[N, K]^T (+ the epilogue of g3_epilogue),  K % 32 == 0, N % 256 == 0
//
// gemm3_tn2_kernel (two 4-wave workgroups per CU, 96 / 128 x 256 tiles in two passes over two B stages) pays THREE barriers per
// 32-deep K step of 72-96 MFMAs per wave, and every barrier waits for copies in flight; it runs N = K = 256 at 34-39 us and the
// K = 1024 products at 77-94 us, 0.5-0.75 PFLOP/s issued.  gemm3_nt2 (256 x 256 tiles, ONE 8-wave workgroup per CU, double-buffered
// plane images, one LDS-only barrier per K step) reaches 1.5 on a problem with twice the staging arithmetic.  This is that
// skeleton for the TN products:
//   * tile 192 x 256 (43 008 rows = 224 tiles: one round of the chip at 7/8 of its CUs; 256 rows would be 168), 8 waves as
//     2 (rows of 96) x 4 (columns of 64): 6 x 4 MFMA tiles of v_mfma_f32_16x16x32_f16 per wave, 72 MFMAs per K step, in
//     gemm3_tn2_kernel's order per accumulator (l.h, h.l, h.h, K steps ascending): BIT-IDENTICAL results;
//   * A: each thread loads 3 x 16 bytes per K step, TWO steps ahead (two register sets: the rows come from HBM), and splits them
//     (h = fp16(s x), l = fp16(s x - h)) into the plane image [plane][k-chunk][row][16 B] of stage kt & 1;
//   * B: the pre-split planes come global -> LDS by DMA (4 pieces of 1 KB per wave and K step, image [plane][column][4 chunks x
//     16 B] with the chunk XOR-swizzled on the source side), requested ONE K step ahead and spread over the MFMA phase (one piece
//     per row tile) so that the requests of 8 waves do not queue behind each other;
//   * ONE barrier per K step (LDS-only: ws_barrier) — it publishes A image / B stage kt & 1; the other pair was last read in
//     the MFMA phase of step kt - 1, which every wave has left when it arrives here.
constexpr int kT3BM = 192;
constexpr int kT3T = 512;
constexpr int kT3AKc = kT3BM * 16;                // bytes per (plane, k-chunk) of the A image
constexpr int kT3A = 2 * 4 * kT3AKc;              // 24 KB
constexpr int kT3B = 2 * 256 * 64;                // both planes of 256 columns x 32 k: 32 KB
constexpr int kT3Stage = kT3A + kT3B;             // 56 KB
constexpr int kT3Lds = 2 * kT3Stage;              // 112 KB: one workgroup per CU

// 4 values -> 4 + 4 halves (the per-element arithmetic of split8h)
__device__ __forceinline__ void split4h(const float4 u, const float scale, uint2* h, uint2* l)
{
    const float x[4] = {u.x, u.y, u.z, u.w};
    unsigned hb[2], lb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f32x2 s = {x[2 * i] * scale, x[2 * i + 1] * scale};
        const f16x2 hh = __builtin_convertvector(s, f16x2);
        const f32x2 back = __builtin_convertvector(hh, f32x2);
        const f32x2 r = {s[0] - back[0], s[1] - back[1]};
        const f16x2 ll = __builtin_convertvector(r, f16x2);
        union { f16x2 f; unsigned u; } ch, cl;
        ch.f = hh; cl.f = ll;
        hb[i] = ch.u; lb[i] = cl.u;
    }
    *h = make_uint2(hb[0], hb[1]);
    *l = make_uint2(lb[0], lb[1]);
}

__global__ __launch_bounds__(kT3T, 2) void gemm3_tn3_kernel(G3 p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char t3_lds[];
    const int per_xcd = (p.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    // p.t3_bm = 192, or 176: the second row half (waves 4-7) then owns 80 rows = five MFMA row tiles — the two waves of a SIMD
    // (w, w + 4) share its matrix pipe, so the SIMD's MFMA time per K step goes with 6 + 5 row tiles, the tile's bytes with 176 rows
    const int bm = p.t3_bm;
    const int m0 = tm * bm, n0 = tn * 256, m_end = min(m0 + bm, p.M);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, g = lane >> 4;

    float sc_a, inv_a, sc_b, inv_b;
    h2_scale(amax_read(p.a_amax), &sc_a, &inv_a);
    h2_scale(amax_read(p.b_amax), &sc_b, &inv_b);
    (void)sc_b;

    // A staging: item (row, half chunk hc of 4 values), three per thread: rows tid / 8 + 64 u, values 4 hc .. 4 hc + 3 of the step
    const int srow = tid >> 3, shc = tid & 7;
    const float* ap[3];
    int aoff[3];                                   // byte offset of the item's 8 bytes inside a plane of the A image
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int row = srow + 64 * u;
        ap[u] = p.a + (int64_t)min(m0 + row, m_end - 1) * p.lda + shc * 4;          // (rows past the tile repeat its last row: never used)
        const int kc = shc >> 1;
        aoff[u] = kc * kT3AKc + (row ^ (2 * kc)) * 16 + (shc & 1) * 8;
    }
    // B staging: piece q = wave + 8 i covers columns 16 (q & 15) .. + 15 of plane q >> 4; lane = (column nl, slot), source chunk =
    // slot ^ swizzle(nl) so that the fragment reads below are conflict-free (gemm3_tn2_kernel's image)
    unsigned boff[4];
    int bdst[4];
    {
        const int nl = lane >> 2, kc = (lane & 3) ^ ((0 - (nl >> 2)) & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave + 8 * i;
            const int pl = q >> 4, nb = (q & 15) * 16;
            boff[i] = (unsigned)(((int64_t)pl * p.plane + (int64_t)(n0 + nb + nl) * p.K + kc * 8) * 2);
            bdst[i] = __builtin_amdgcn_readfirstlane(kT3A + pl * (256 * 64) + nb * 64);
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)t3_lds;

    // A values of the next TWO K steps in two register sets (a step's loads have two MFMA phases to arrive: they come from HBM)
    float4 ra[2][3];
#define T3_LOAD_A(set, k0)                                                            \
    {                                                                                 \
        _Pragma("unroll") for (int u = 0; u < 3; ++u) ra[set][u] = *reinterpret_cast<const float4*>(ap[u] + (k0)); \
    }
    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / kBK;
    const int klast = (nk - 1) * kBK;
    const int nrt = wr ? (bm - 96) >> 4 : 6;             // row tiles of this wave (uniform): 6, or 5 in the second half of a 176-row tile
    // fragment addresses: A rows 96 wr + 16 i + r16, chunk g; B columns 64 wc + 16 j + r16, chunk g (swizzled)
    const int a_frag = g * kT3AKc + ((wr * 96 + r16) ^ (2 * g)) * 16;
    const int b_frag = kT3A + (wc * 64 + r16) * 64 + ((g ^ ((0 - (r16 >> 2)) & 3)) * 16);

    T3_LOAD_A(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(p.bp, boff[i], lds0 + (unsigned)bdst[i]);
    __builtin_amdgcn_sched_barrier(0);
    T3_LOAD_A(1, min(kBK, klast));
    __builtin_amdgcn_sched_barrier(0);

    // one K step: split register set `set` into stage kt & 1, reload the set for step kt + 2, barrier, MFMA phase
#define T3_STEP(set, kt)                                                                                                        \
    {                                                                                                                           \
        unsigned char* st = t3_lds + ((kt) & 1) * kT3Stage;                                                                     \
        _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                                                         \
            uint2 h, l;                                                                                                         \
            split4h(ra[set][u], sc_a, &h, &l);                                                                                  \
            *reinterpret_cast<uint2*>(st + aoff[u]) = h;                                                                        \
            *reinterpret_cast<uint2*>(st + 4 * kT3AKc + aoff[u]) = l;                                                           \
        }                                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        T3_LOAD_A(set, min(((kt) + 2) * kBK, klast));                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        /* this step's B pieces (requested during the previous MFMA phase) have landed: the only younger loads are the three above */ \
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                                        \
        ws_barrier();                                                                                                           \
        /* MFMA phase on stage kt & 1; the B pieces of step kt + 1 -> stage (kt + 1) & 1, one per two row tiles */              \
        const int kn = min(((kt) + 1) * kBK, klast);                                                                            \
        const unsigned nst = lds0 + (unsigned)((((kt) + 1) & 1) * kT3Stage);                                                    \
        f16x8 fb[2][4];                                                                                                         \
        _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) _Pragma("unroll") for (int j = 0; j < 4; ++j)                          \
            fb[pl][j] = as_fragh(*reinterpret_cast<const uint4*>(st + b_frag + pl * (256 * 64) + j * 1024));                    \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                         \
            if (i == 5 && nrt < 6) break;                                                                                       \
            const f16x8 fh0 = as_fragh(*reinterpret_cast<const uint4*>(st + a_frag + i * 256));                                 \
            const f16x8 fh1 = as_fragh(*reinterpret_cast<const uint4*>(st + a_frag + 4 * kT3AKc + i * 256));                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][j], fh0, acc[i][j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][j], fh1, acc[i][j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][j], fh0, acc[i][j], 0, 0, 0); \
            if (i < 4) glds16(p.bp + kn, boff[i], nst + (unsigned)bdst[i]);                                                     \
        }                                                                                                                       \
    }
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        T3_STEP(0, kt)
        T3_STEP(1, kt + 1)
    }
    if (kt < nk) T3_STEP(0, kt)
#undef T3_STEP
#undef T3_LOAD_A
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the surplus pieces of the last step must not outlive the LDS allocation)
    // the epilogue in two halves of three row tiles (gemm3_tn2_kernel's instantiation for its 96-row tiles; 36 instead of 72
    // operand registers)
    float omax = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        struct { f32x4 v[3][4]; } out;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) out.v[i][j] = acc[hf * 3 + i][j];
        omax = fmaxf(omax, g3_epilogue<4, 3, decltype(out), true, true>(p, out, lane, m0 + wr * 96 + hf * 48, n0 + wc * 64, inv_a, inv_b, m_end));
    }
    if (p.out_amax) {                    // (uniform) one atomic per workgroup (8 waves: amax_commit is written for 4)
        float m = omax;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float* red = reinterpret_cast<float*>(t3_lds);
        __syncthreads();
        if (lane == 0) red[wave] = m;
        __syncthreads();
        if (tid == 0) {
            m = red[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
            atomicMax(reinterpret_cast<unsigned*>(p.out_amax) + (blockIdx.x % kAmaxSub) * kAmaxStride, __float_as_uint(m));
        }
    }
}
