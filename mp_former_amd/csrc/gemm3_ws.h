// Weight-stationary form of the fp16 x 2 TN product for K = 256 (round 5) — included by gemm3.hip inside its namespace.
//
//   C[M, N] = A[M, 256] . B[N, 256]^T (+ bias, addends, ReLU, gates), N a multiple of 256.
//
// The shapes this serves (value / output projections and their input gradients: N = K = 256; linear1 and the hidden
// gradient: N = 1024, K = 256 — msdeformattn.py:116-131, ops/modules/ms_deform_attn.py:98-124) stream 44 MB of activations
// against 256 KB of weights.  The tiled kernels re-read the weight planes from L2 once per 96 / 128 rows (115 MB of L2 -> LDS
// traffic for a 44 MB operand), split every A element once per 256 output columns inside the K loop and pay two or three
// barriers per K step.  Here the weights never move and the split leaves the K loop:
//
//   * one persistent 8-wave workgroup per CU owns a 256-column group; wave w keeps the fp16 h | l fragments of its 32
//     columns over the WHOLE contraction in registers (8 K steps x 2 column tiles x 2 planes x 4 dwords = 128 VGPRs),
//     loaded once per launch;
//   * the workgroup walks a contiguous range of rows in tiles of 32.  A tile of A (32 rows x 1 KB) comes global -> LDS by
//     DMA (global_load_lds_dwordx4: one row per wave instruction, wave w owns rows 4 w .. 4 w + 3 of every tile) into one of
//     two fp32 stages, two tiles ahead of its use;
//   * CONVERT, once per element: the wave that copied a row splits it — h = fp16(s x), l = fp16(s x - h), the same split8h and
//     scale as the tiled kernels — into the tile's fp16 plane image (row = [h(256) | l(256)], 16-byte chunks XOR-swizzled by
//     row & 15).  A wave converts only rows it copied itself, so its own vmcnt wait is all the synchronisation the stage
//     needs, and it refills the rows it has just consumed;
//   * COMPUTE: a K step of a wave is 4 ds_read_b128 (2 row tiles x 2 planes, conflict-free in the 16-lane service groups,
//     requested one step ahead) + 12 v_mfma_f32_16x16x32_f16 in the tiled kernels' order — l.h, h.l, h.h, K steps ascending —
//     so every output element is BIT-IDENTICAL to gemm3_tn2_kernel's (tests/test_gemm3_gpu.py compares the two);
//   * ONE barrier per 32-row tile (plane image t + 1 complete, plane image t released); the loop has no B traffic, no LDS
//     stores beside the 2 KB per wave of the convert, no staging registers;
//   * per iteration a wave converts tile t + 1, stores tile t - 1 (its accumulators live across the barrier) and then runs the
//     MFMA phase of tile t; the epilogue's operands are template flags (no dummy loads, straight-line code).
// What bounds it (tools/ws_phases.py, tools/ws_prologue.py): the two waves of a SIMD do not overlap — an MFMA-streaming wave
// starves its partner — so a tile costs the SUM of its phases (~5 900 cycles per 32 rows: MFMA 2 x 1 536, convert, epilogue,
// barrier skew), and a launch pays ~20 k cycles of prologue for the 256 KB of weight fragments per CU (row-strided 64-byte
// pieces; a fragment-major plane layout would halve it).
//
// Work split: row worker rw (the workgroups that share rows sit on one XCD, so the N / 256 column groups of a row range read
// it from that XCD's L2) owns rows [rw * rpw, (rw + 1) * rpw), rpw a multiple of 16 chosen so that every CU has work:
// 43 008 rows on 256 CUs = 176 rows each (245 workers) instead of 448 tiles on 512 slots.

constexpr int kWsBM = 32;                 // rows per tile
constexpr int kWsK = 256;                 // contraction length (8 K steps of 32)
constexpr int kWsThreads = 512;           // 8 waves x 32 columns
constexpr int kWsStage = kWsBM * kWsK * 4;    // bytes of one fp32 stage = of one plane image (32 KB)
constexpr int kWsLds = 4 * kWsStage;      // two stages + two plane images

// Workgroup barrier for LDS data only.  __syncthreads() is a fence + barrier: with global stores pending hipcc puts
// s_waitcnt vmcnt(0) in front of it, which also drains the row copies requested a moment earlier — every tile then waited for
// a full HBM round trip and for its own stores (measured: 37 us with the stores in the loop, 10 us without them).
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// EPI: 0 = bias only; 1 = + cin (a second addend or an fp32 gate operand take the tiled kernel).  ReLU and the bit-mask output
// are uniform register-only branches as in g3_epilogue.  The operands are template flags because every load of the loop is
// counted by hand (the row copies are inline asm the compiler's vmcnt bookkeeping does not see).
// GB: the ReLU gate arrives as a bit mask (p.gbits).  RELU: max(., 0).  BOUT: the mask of (C > 0) goes to p.gbits_out.
// (Template flags, not uniform branches: the two waves of a SIMD share its issue slots, an MFMA leaves room for about two other
// instructions, and the epilogue of a tile had been ~330 VALU instructions per wave against 96 MFMAs — measured: the phases of
// the wave pair add up instead of overlapping.  Straight-line code, one address computation per row and max3 with |.| source
// modifiers bring it to ~110.)
template <int EPI, bool GB, bool RELU, bool BOUT>
__global__ __launch_bounds__(kWsThreads, 2) void gemm3_ws_kernel(G3 p, int rpw, int ncg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int cg = slot % ncg, rw = (slot / ncg) * 8 + xcd;
    const int r0 = rw * rpw;
    if (r0 >= p.M) return;                                 // (uniform: the whole workgroup leaves)
    const int r1 = min(p.M, r0 + rpw);
    const int ntile = (r1 - r0 + kWsBM - 1) / kWsBM;
    const int n0 = cg * 256 + wave * 32;
    const unsigned lds_stage = (unsigned)(uintptr_t)ws_lds;

    // this wave's 4 rows of tile t -> stage t & 1.  Slot q of a row receives chunk 2 q (q < 32) or 2 (q - 32) + 1: the even
    // 16-byte chunks of the row in its first half, the odd ones in the second, so that the convert's two reads per 8 values
    // (slots kk and 32 + kk) are conflict-free.  Rows past the worker's range repeat its last row (never stored).
    const unsigned src_off = (unsigned)((((2 * lane) & 63) | (lane >> 5)) * 16);
    auto dma_rows = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = wave * 4 + i;
            const int row = min(r0 + t * kWsBM + rl, r1 - 1);
            glds16(p.a + (int64_t)row * p.lda, src_off, lds_stage + (unsigned)((t & 1) * kWsStage + rl * 1024));
        }
    };
    dma_rows(0);
    dma_rows(1);

    // the wave's weight fragments: fb[s][j][pl] = 8 halves k = 32 s + 8 g .. + 7 of column n0 + 16 j + r16, plane pl
    uint4 fb[8][2][2];
    {
        const unsigned short* bw = p.bp + (int64_t)(n0 + r16) * kWsK + g * 8;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    fb[s][j][pl] = *reinterpret_cast<const uint4*>(bw + (int64_t)pl * p.plane + (int64_t)j * 16 * kWsK + s * 32);
    }
    float sc_a, inv_a, sc_b, inv_b;
    h2_scale(amax_read(p.a_amax), &sc_a, &inv_a);
    h2_scale(amax_read(p.b_amax), &sc_b, &inv_b);
    (void)sc_b;
    const int nq = n0 + g * 4;                             // this lane's first column of column tile 0 (tile 1: + 16)
    const float4 bz0 = *reinterpret_cast<const float4*>(p.bias + nq * p.bias_cm);
    const float4 bz1 = *reinterpret_cast<const float4*>(p.bias + (nq + 16) * p.bias_cm);

    // stage t & 1 (this wave's rows) -> plane image t & 1
    auto convert = [&](int t) {
        const unsigned char* st = ws_lds + (t & 1) * kWsStage;
        unsigned char* pb = ws_lds + 2 * kWsStage + (t & 1) * kWsStage;
        const int kk = lane & 31;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int rr = wave * 4 + 2 * a + (lane >> 5);
            const float4 u = *reinterpret_cast<const float4*>(st + rr * 1024 + kk * 16);
            const float4 v = *reinterpret_cast<const float4*>(st + rr * 1024 + (32 + kk) * 16);
            uint4 h, l;
            split8h(u, v, sc_a, &h, &l);
            const int pos = kk ^ (rr & 15);
            *reinterpret_cast<uint4*>(pb + rr * 1024 + pos * 16) = h;
            *reinterpret_cast<uint4*>(pb + rr * 1024 + (32 + pos) * 16) = l;
        }
    };

    // tiles 0 and 1 of this wave's rows and the weight fragments have arrived.  The BUILTIN, not inline asm: hipcc's waitcnt pass
    // has to see that its own 40 loads are done, or it drains them with vmcnt(N) waits inside the loop that count the row
    // copies it does not know of — i.e. waits for copies issued a moment ago.  (vmcnt 0, expcnt 7, lgkmcnt 15: gfx9 encoding)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    convert(0);
    dma_rows(2);
    ws_barrier();

    // fragment of (row tile i, plane pl, K step s): row 16 i + r16, chunk (32 pl + ((4 s + g) ^ r16))
    const int fbase = r16 * 1024 + (g ^ (r16 & 3)) * 16;
    const int fsw = r16 & 12;
    float omax = 0.f;
    f32x4 acc[2][2];

    // ---- compute: plane image t & 1 -> acc -------------------------------------------------------------------------------
    auto compute = [&](int t) {
        const unsigned char* pb = ws_lds + 2 * kWsStage + (t & 1) * kWsStage + fbase;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        uint4 fa[2][2][2];                                 // [buffer][row tile][plane]
#define WS_LOAD_FRAGS(buf, s)                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_)       \
        fa[buf][i_][pl_] = *reinterpret_cast<const uint4*>(pb + i_ * 16384 + pl_ * 512 + ((4 * (s)) ^ fsw) * 16);
        WS_LOAD_FRAGS(0, 0)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) { WS_LOAD_FRAGS((s + 1) & 1, s + 1) }
            if (s & 1) {     // one row copy of tile t + 3 per two K steps (into the rows this wave converted in `middle`): all 32 copies of a
                             // workgroup requested together right after the barrier queued behind each other in the CU's address unit
                             // (~500 cycles per tile for the last wave); spread over the MFMA phase the issue waits hide (N = 1024: 88 -> 85 us)
                const int rl = wave * 4 + (s >> 1);
                const int row = min(r0 + (t + 3) * kWsBM + rl, r1 - 1);
                glds16(p.a + (int64_t)row * p.lda, src_off, lds_stage + (unsigned)(((t + 3) & 1) * kWsStage + rl * 1024));
            }
            // l.h, h.l, h.h — the order of Acc2::pass (smallest first); four independent accumulators per product
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_fragh(fb[s][j][1]), as_fragh(fa[s & 1][i][0]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_fragh(fb[s][j][0]), as_fragh(fa[s & 1][i][1]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_fragh(fb[s][j][0]), as_fragh(fa[s & 1][i][0]), acc[i][j], 0, 0, 0);
        }
#undef WS_LOAD_FRAGS
    };

    // ---- the middle of an iteration: wait for the rows of tile tc, request the addends of tile te, convert tile tc, then the
    // epilogue of tile te from acc (te < 0: none).  Vector-memory operations in flight at the wait, oldest first: the rows of
    // tile tc (two iterations old), stores and — long consumed — addend loads of earlier epilogues, the rows of tile tc + 1
    // (4 loads, the youngest).  Loads return in order among loads, so "at most 4 outstanding" means the rows of tile tc have
    // landed whatever the stores are doing; and no load the compiler knows of is followed by a row copy before its use, so
    // the waits hipcc inserts for the addends never wait for copies it cannot see.
    // The epilogue's operands (addend rows, gate bits) of tile te are requested one iteration AHEAD — right after the row copies
    // of iteration te, before its MFMA phase — so that a whole MFMA phase covers their round trip (requested inside `middle`
    // they were an exposed L2 / HBM latency per tile: the bit-gated product ran 112 us against 86 for the ungated one).
    constexpr int kEpiLoads = (EPI >= 1 ? 4 : 0) + (GB ? 2 : 0);
    float4 ci[2][2];
    unsigned gb[2] = {0xffffffffu, 0xffffffffu};
    auto epi_loads = [&](int te) {
        const int m_tile = r0 + te * kWsBM;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t mr = min(m_tile + i * 16 + r16, r1 - 1);
            if constexpr (EPI >= 1) {
                const float* cr = p.cin + mr * p.ldcin + nq;
                ci[i][0] = *reinterpret_cast<const float4*>(cr);
                ci[i][1] = *reinterpret_cast<const float4*>(cr + 16);
            }
            if constexpr (GB) gb[i] = *reinterpret_cast<const unsigned*>(p.gbits + mr * p.ldgbits + (n0 >> 3));
        }
    };
    auto middle = [&](int tc, int te) {
        // in flight, oldest first: the rows of tile tc, stores of earlier epilogues, the rows of tile tc + 1 (4 loads), the operands
        // of this epilogue (kEpiLoads loads, requested right after those rows): "at most 4 + kEpiLoads outstanding" = rows tc landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + kEpiLoads) : "memory");
        float* crow[2];
        bool mok[2];
        const int m_tile = r0 + max(te, 0) * kWsBM;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m_tile + i * 16 + r16;
            mok[i] = m < r1 && te >= 0;
            const int64_t mr = min(m, r1 - 1);
            crow[i] = p.c + mr * p.ldc + nq;
        }
        __builtin_amdgcn_sched_barrier(0);
        convert(tc);
        __builtin_amdgcn_sched_barrier(0);
        if (te >= 0) {                      // (uniform)
            // ---- epilogue (the arithmetic and its order are g3_epilogue's; an absent addend is not added: x + 0 differs from x
            // only in the sign of a zero) -------------------------------------------------------------------------------------
            unsigned wb[2] = {0u, 0u};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 bz = j ? bz1 : bz0;
                    float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                    o = make_float4(o.x * inv_a * inv_b, o.y * inv_a * inv_b, o.z * inv_a * inv_b, o.w * inv_a * inv_b);
                    o = make_float4(o.x + bz.x, o.y + bz.y, o.z + bz.z, o.w + bz.w);
                    if constexpr (EPI >= 1) o = make_float4(o.x + ci[i][j].x, o.y + ci[i][j].y, o.z + ci[i][j].z, o.w + ci[i][j].w);
                    if constexpr (RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
                    if constexpr (GB) {
                        const unsigned nib = gb[i] >> (j * 16 + g * 4);
                        o = make_float4((nib & 1u) ? o.x : 0.f, (nib & 2u) ? o.y : 0.f, (nib & 4u) ? o.z : 0.f, (nib & 8u) ? o.w : 0.f);
                    }
                    if constexpr (BOUT) {
                        const unsigned pos = (o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u);
                        wb[i] |= pos << (j * 16 + g * 4);
                    }
                    if (mok[i]) {
                        omax = __builtin_fmaxf(__builtin_fmaxf(omax, __builtin_fmaxf(fabsf(o.x), fabsf(o.y))), __builtin_fmaxf(fabsf(o.z), fabsf(o.w)));
                        *reinterpret_cast<float4*>(crow[i] + 16 * j) = o;
                    }
                }
            }
            if constexpr (BOUT) {       // the four lane groups of a row hold disjoint nibbles of its 32 bits
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    unsigned w32 = wb[i];
                    w32 |= __shfl_xor(w32, 16);
                    w32 |= __shfl_xor(w32, 32);
                    if (g == 0 && mok[i])
                        *reinterpret_cast<unsigned*>(p.gbits_out + (int64_t)min(m_tile + i * 16 + r16, r1 - 1) * p.ldgbits_out + (n0 >> 3)) = w32;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // Every wave: convert tile t + 1 and store tile t - 1 (accumulators live across the barrier), refill the rows just consumed,
    // then the MFMA phase of tile t.  One barrier per tile: it publishes plane image t + 1 and retires plane image t.
    // (Measured and rejected: the two waves of a SIMD half a tile apart — one in its MFMA phase while the other converts and
    // stores.  A wave that streams MFMAs leaves its SIMD partner almost no issue slots, whatever s_setprio says: the partner's
    // convert took 2 400 cycles instead of 570 and the pair's phases added up; N = 1024: 100 / 112 us against 86 / 100 in
    // this order.)
    for (int t = 0; t < ntile; ++t) {
        middle(t + 1, t - 1);
        __builtin_amdgcn_sched_barrier(0);
        epi_loads(t);                    // consumed by the epilogue of tile t in the next iteration
        __builtin_amdgcn_sched_barrier(0);
        compute(t);
        __builtin_amdgcn_sched_barrier(0);
        ws_barrier();                    // plane image t + 1 complete; plane image t free for the convert of tile t + 2
    }
    middle(ntile + 1, ntile - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (surplus row copies must not outlive the LDS allocation)
    if (p.out_amax) {                    // (uniform) one atomic per workgroup
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
        float* red = reinterpret_cast<float*>(ws_lds);
        __syncthreads();
        if (lane == 0) red[wave] = omax;
        __syncthreads();
        if (tid == 0) {
            float m = red[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
            atomicMax(reinterpret_cast<unsigned*>(p.out_amax) + (blockIdx.x % kAmaxSub) * kAmaxStride, __float_as_uint(m));
        }
    }
}
