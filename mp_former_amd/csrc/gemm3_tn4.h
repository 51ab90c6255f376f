// gemm3_tn3_kernel's tile (192 x 256, one 8-wave workgroup per CU, fp16 x 2 TN products with N % 256 == 0) with the two row halves
// half a K step apart (round 6) — included by gemm3.hip inside its namespace, after gemm3_tn3.h.
//
// In gemm3_tn3_kernel all eight waves walk a K step in lockstep: split the A registers into the plane image, barrier, fragment
// reads + 72 MFMAs — so both waves of a SIMD want the matrix pipe at the same time (2 x 1 152 cycles back to back) and both do
// their staging (loads, ~40 VALU, LDS writes, the waits in front of them) at the same time, with the pipe idle.  The measured K
// step is ~4 700-5 500 cycles for 2 304 cycles of MFMA.  Here the workgroup is two GROUPS of four waves — waves 0-3
// own rows 0..95, waves 4-7 rows 96..191; wave w and wave w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs
// cyclically) — and a K step is two SLOTS separated by barriers:
//
//     slot 2k     group A: MFMA phase of step k            | group B: LOAD phase of step k + 1
//     slot 2k + 1 group A: LOAD phase of step k + 1        | group B: MFMA phase of step k
//
// so every SIMD has exactly one wave streaming MFMAs (alone on the pipe) and its partner staging under it.  LOAD(k + 1) = request
// the group's four 1 KB pieces of the B planes of step k + 1 (DMA), split the registers of step k + 1 into the group's own A image,
// re-load that register set for step k + 3.  Buffers (112 KB, as gemm3_tn3): per group two A images of 12 KB (96 rows), two B
// stages of 32 KB shared by both groups.  Hazards: B stage (k + 1) & 1 is written from slot 2k (group B) and 2k + 1 (group A) and was
// last read in slot 2k - 1 (group B's MFMA of step k - 1); group G's A image (k + 1) & 1 is written one slot before G reads it and
// was last read two steps earlier; every slot ends in one LDS-only barrier of all eight waves.
// Same arithmetic in the same order as gemm3_tn3_kernel / gemm3_tn2_kernel: BIT-IDENTICAL results (tests/test_gemm3_gpu.py).
constexpr int kT4AG = 2 * 4 * 96 * 16;            // one group's A image: [plane][k-chunk][96 rows][16 B] = 12 KB
constexpr int kT4A = 2 * 2 * kT4AG;               // [buffer][group]: 48 KB
constexpr int kT4B = 2 * 256 * 64;                // both planes of 256 columns x 32 k: 32 KB
constexpr int kT4Lds = kT4A + 2 * kT4B;           // 112 KB

__global__ __launch_bounds__(kT3T, 2) void gemm3_tn4_kernel(G3 p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char t4_lds[];
    const int per_xcd = (p.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * kT3BM, n0 = tn * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;          // group = row half; wave w + 4 is wave w's SIMD partner
    const int r16 = lane & 15, g = lane >> 4;

    float sc_a, inv_a, sc_b, inv_b;
    h2_scale(amax_read(p.a_amax), &sc_a, &inv_a);
    h2_scale(amax_read(p.b_amax), &sc_b, &inv_b);
    (void)sc_b;

    // A staging of the group's 96 rows by its 256 threads: item (row t / 8 + 32 u, values 4 hc .. 4 hc + 3 of the step), u = 0..2
    const int t = tid & 255;
    const int srow = t >> 3, shc = t & 7;
    const float* ap[3];
    int aoff[3];                                   // byte offset of the item's 8 bytes inside a plane of the group's A image
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int row = srow + 32 * u;
        ap[u] = p.a + (int64_t)min(m0 + grp * 96 + row, p.M - 1) * p.lda + shc * 4;
        const int kc = shc >> 1;
        aoff[u] = kc * (96 * 16) + (row ^ (2 * kc)) * 16 + (shc & 1) * 8;
    }
    // B staging: piece q = wave + 8 i covers columns 16 (q & 15) .. + 15 of plane q >> 4 (gemm3_tn3_kernel's image and pieces)
    unsigned boff[4];
    int bdst[4];
    {
        const int nl = lane >> 2, kc = (lane & 3) ^ ((0 - (nl >> 2)) & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave + 8 * i;
            const int pl = q >> 4, nb = (q & 15) * 16;
            boff[i] = (unsigned)(((int64_t)pl * p.plane + (int64_t)(n0 + nb + nl) * p.K + kc * 8) * 2);
            bdst[i] = __builtin_amdgcn_readfirstlane(kT4A + pl * (256 * 64) + nb * 64);
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)t4_lds;
    unsigned char* a_img = t4_lds + grp * kT4AG;             // + buffer * 2 * kT4AG

    float4 ra[2][3];
#define T4_LOAD_A(set, k0)                                                            \
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
    // fragment addresses: A rows 16 i + r16 of the group's image, chunk g; B columns 64 wc + 16 j + r16, chunk g (swizzled)
    const int a_frag = g * (96 * 16) + (r16 ^ (2 * g)) * 16;
    const int b_frag = kT4A + (wc * 64 + r16) * 64 + ((g ^ ((0 - (r16 >> 2)) & 3)) * 16);

    // LOAD phase of step ks (uniform ks < nk): split register set ks & 1 into A image ks & 1 (the compiler's wait for the set allows
    // the three younger loads it knows of, the other set's reloads: nothing this phase issued), then this wave's four B pieces
    // (DMA; the rest of the partner's MFMA phase to land), then the set's reload for step ks + 2.  At the end: at most the three
    // reloads outstanding = the pieces have landed.
#define T4_LOAD(set, ks)                                                                                                        \
    {                                                                                                                           \
        unsigned char* st = a_img + ((ks) & 1) * (2 * kT4AG);                                                                   \
        _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                                                         \
            uint2 h, l;                                                                                                         \
            split4h(ra[set][u], sc_a, &h, &l);                                                                                  \
            *reinterpret_cast<uint2*>(st + aoff[u]) = h;                                                                        \
            *reinterpret_cast<uint2*>(st + 4 * (96 * 16) + aoff[u]) = l;                                                        \
        }                                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        const unsigned bst = lds0 + (unsigned)(((ks) & 1) * kT4B);                                                              \
        const int kb_ = (ks) * kBK;                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) glds16(p.bp + kb_, boff[i], bst + (unsigned)bdst[i]);                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        T4_LOAD_A(set, min(((ks) + 2) * kBK, klast));                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                                        \
    }
    // MFMA phase of step kt on A image kt & 1 of the group and B stage kt & 1: gemm3_tn3_kernel's product order
#define T4_MFMA(kt)                                                                                                             \
    {                                                                                                                           \
        const unsigned char* sa = a_img + ((kt) & 1) * (2 * kT4AG);                                                             \
        const unsigned char* sb = t4_lds + ((kt) & 1) * kT4B;                                                                   \
        f16x8 fb[2][4];                                                                                                         \
        _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) _Pragma("unroll") for (int j = 0; j < 4; ++j)                          \
            fb[pl][j] = as_fragh(*reinterpret_cast<const uint4*>(sb + b_frag + pl * (256 * 64) + j * 1024));                    \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                         \
            const f16x8 fh0 = as_fragh(*reinterpret_cast<const uint4*>(sa + a_frag + i * 256));                                 \
            const f16x8 fh1 = as_fragh(*reinterpret_cast<const uint4*>(sa + a_frag + 4 * (96 * 16) + i * 256));                 \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][j], fh0, acc[i][j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][j], fh1, acc[i][j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][j], fh0, acc[i][j], 0, 0, 0); \
        }                                                                                                                       \
    }

    T4_LOAD_A(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    T4_LOAD_A(1, min(kBK, klast));
    __builtin_amdgcn_sched_barrier(0);
    T4_LOAD(0, 0)                          // both groups: step 0 (all eight pieces of B stage 0 per group of four waves x 4 = both planes)
    ws_barrier();
    if (grp == 0) {
        // group A: MFMA(k) | LOAD(k + 1)
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            T4_MFMA(kt)
            ws_barrier();
            T4_LOAD(1, kt + 1)
            ws_barrier();
            T4_MFMA(kt + 1)
            ws_barrier();
            if (kt + 2 < nk) T4_LOAD(0, kt + 2)
            ws_barrier();
        }
        if (kt < nk) {
            T4_MFMA(kt)
            ws_barrier();
            ws_barrier();
        }
    } else {
        // group B: LOAD(k + 1) | MFMA(k)
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            T4_LOAD(1, kt + 1)
            ws_barrier();
            T4_MFMA(kt)
            ws_barrier();
            if (kt + 2 < nk) T4_LOAD(0, kt + 2)
            ws_barrier();
            T4_MFMA(kt + 1)
            ws_barrier();
        }
        if (kt < nk) {
            ws_barrier();
            T4_MFMA(kt)
            ws_barrier();
        }
    }
#undef T4_MFMA
#undef T4_LOAD
#undef T4_LOAD_A
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (surplus register reloads of the clamped last steps)
    float omax = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        struct { f32x4 v[3][4]; } out;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) out.v[i][j] = acc[hf * 3 + i][j];
        omax = fmaxf(omax, g3_epilogue<4, 3, decltype(out), true>(p, out, lane, m0 + grp * 96 + hf * 48, n0 + wc * 64, inv_a, inv_b));
    }
    if (p.out_amax) {                    // (uniform) one atomic per workgroup
        float m = omax;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float* red = reinterpret_cast<float*>(t4_lds);
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
