// 256 x 256 output tiles for the grouped weight gradients of an encoder layer (round 5) — included by gemm3.hip inside its namespace.
//
//   Cpart[s][m][n] = sum_{r in split s} A[r, m] * B[r, n]      (fp16 x 2 form, both operands fp32 activations, M % 256 == N % 256 == 0)
//
// The 128 x 128 tiles of gemm3_nt_tile read (128 + 128) columns of the two operands per 16 K outputs: the four gradients of a layer
// (256 x 1024, 1024 x 256, 2 x 256 x 256 over 43 008 rows) pull 1.76 GB through the CUs for 616 MB of operands, and the launch
// runs at that traffic's pace (258 us).  A 256 x 256 tile halves it (0.88 GB): one 8-wave workgroup per CU (wave tile 128 x 64 on
// v_mfma_f32_32x32x16_f16, 128 accumulator registers), the staging of gemm3_nt_tile with twice the columns per thread count —
// column-wise 4-byte buffer loads with a scalar row offset, split in registers, one ds_write_b128 per plane and item — into a
// DOUBLE-buffered plane image (2 x 64 KB), so a K step has ONE barrier.  Same products in the same order per output element as the
// 128 x 128 tile for equal split boundaries.
constexpr int kN2T = 512;                       // threads: 8 waves as 2 (rows of 128) x 4 (columns of 64)
constexpr int kN2Kc = 256 * 16;                 // bytes per (plane, k-chunk) of an operand image
constexpr int kN2Op = 2 * 4 * kN2Kc;            // one operand, both planes: 32 KB
constexpr int kN2Stage = 2 * kN2Op;             // A + B: 64 KB
constexpr int kN2Lds = 2 * kN2Stage;            // two stages

struct AccN2 {
    f32x16 v[4][2];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) v[i][j][e] = 0.f;
    }
    // a_base / b_base: byte offset of the wave's first row / column in the stage
    __device__ __forceinline__ void step(const unsigned char* st, int a_base, int b_base, int lane)
    {
        const int r32 = lane & 31, gh = lane >> 5;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int kc = 2 * kh + gh;                       // this lane's k-chunk (8 values) of the 16-deep MFMA
            const int sw = (r32 ^ (2 * kc)) * 16;
            f16x8 fb[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int t = 0; t < 2; ++t) fb[pl][t] = as_fragh(*reinterpret_cast<const uint4*>(st + kN2Op + b_base + (pl * 4 + kc) * kN2Kc + t * 512 + sw));
            // one 32-row tile of A at a time (8 fragment registers live instead of 32); per accumulator the products keep the order
            // l.h, h.l, h.h of Acc<4, true>::step_h2, the two column tiles alternate between dependent MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 fah = as_fragh(*reinterpret_cast<const uint4*>(st + a_base + (0 * 4 + kc) * kN2Kc + i * 512 + sw));
                const f16x8 fal = as_fragh(*reinterpret_cast<const uint4*>(st + a_base + (1 * 4 + kc) * kN2Kc + i * 512 + sw));
#pragma unroll
                for (int j = 0; j < 2; ++j) v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[1][j], fah, v[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[0][j], fal, v[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[0][j], fah, v[i][j], 0, 0, 0);
            }
        }
    }
};

// CV: the weight gradient of a 3x3 / stride 1 / padding 1 convolution of channel-last images (gemm3_nt_tile's CV mode): the
// column tile fixes the tap (Cin % 256 == 0), B = the input image read at the rows shifted by the tap, taps off the image zeroed
// (W % 8 == 0: the 8 rows of a k-chunk lie in one image row).
template <bool CV>
__device__ __forceinline__ void gemm3_nt2_tile(const G3N& p, const int tile, unsigned char* lds)
{
    float sc_a, sc_b, inv_a, inv_b;
    h2_scale(amax_read(p.a_amax), &sc_a, &inv_a);
    h2_scale(amax_read(p.b_amax), &sc_b, &inv_b);
    const int tn = tile % p.tiles_n, tm = (tile / p.tiles_n) % p.tiles_m, sp = tile / (p.tiles_n * p.tiles_m);
    const int m0 = tm * 256, n0 = tn * 256;
    const int r_begin = sp * p.rows_per_split, r_end = min(p.R, r_begin + p.rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;

    // staging: thread = column (tid & 255) of A and of B, k-chunks kc0 = tid >> 8 and kc0 + 2 (wave-uniform: the row part of every
    // load address is a scalar offset of the buffer load)
    const int col = tid & 255, kc0 = __builtin_amdgcn_readfirstlane(tid >> 8);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, p.b_bytes, 0x00020000);
    const int acolb = (m0 + col) * 4, bcolb = (CV ? (n0 % p.cv_cin) + col : n0 + col) * 4;
    const int ldab = (int)p.lda * 4, ldbb = (int)p.ldb * 4;
    // CV: tap of this column tile; image coordinates of the first row of the thread's two k-chunks (advanced by 32 rows per step)
    int cv_dy = 0, cv_dx = 0, cv_sh = 0, cvx0 = 0, cvy0 = 0, cvx1 = 0, cvy1 = 0;
    if constexpr (CV) {
        const int tap = n0 / p.cv_cin;
        cv_dy = tap / 3 - 1; cv_dx = tap % 3 - 1;
        cv_sh = cv_dy * p.cv_W + cv_dx;
        const int rb0 = r_begin + kc0 * 8, rb1 = rb0 + 16;
        cvx0 = rb0 % p.cv_W; cvy0 = (rb0 / p.cv_W) % p.cv_H;
        cvx1 = rb1 % p.cv_W; cvy1 = (rb1 / p.cv_W) % p.cv_H;
    }
    (void)cv_dy; (void)cv_dx; (void)cv_sh;
    // operand values of the NEXT K step (this thread's 2 + 2 items of 8 rows), requested as soon as the split has consumed the
    // current ones.  (Two alternating register sets — loads two steps ahead — were built: 64 + 128 accumulator registers + fragments do
    // not fit 256, 170-210 spilled registers.)
    float xa0[8], xa1[8], xb0[8], xb1[8];
    float csa = 0.f;
    const bool want_csa = p.csum_a && tn == 0;

#define N2_LOAD_A(r0, TAIL)                                                                                      \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                          \
            const int q0 = (r0) + kc0 * 8 + j, q1 = q0 + 16;                                                     \
            const int c0 = TAIL ? min(q0, r_end - 1) : q0, c1 = TAIL ? min(q1, r_end - 1) : q1;                  \
            float va0 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(ars, acolb, c0 * ldab, 0));          \
            float va1 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(ars, acolb, c1 * ldab, 0));          \
            if (TAIL) { va0 = q0 < r_end ? va0 : 0.f; va1 = q1 < r_end ? va1 : 0.f; }                            \
            xa0[j] = va0; xa1[j] = va1;                                                                          \
        }                                                                                                        \
    }
#define N2_LOAD_B(r0, TAIL)                                                                                      \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                          \
            const int q0 = (r0) + kc0 * 8 + j, q1 = q0 + 16;                                                     \
            int c0 = TAIL ? min(q0, r_end - 1) : q0, c1 = TAIL ? min(q1, r_end - 1) : q1;                        \
            if constexpr (CV) { c0 = min(max(c0 + cv_sh, 0), p.R - 1); c1 = min(max(c1 + cv_sh, 0), p.R - 1); }  \
            float vb0 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, bcolb, c0 * ldbb, 0));          \
            float vb1 = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, bcolb, c1 * ldbb, 0));          \
            if constexpr (CV) {                                                                                  \
                const bool ok0 = (unsigned)(cvy0 + cv_dy) < (unsigned)p.cv_H && (unsigned)(cvx0 + j + cv_dx) < (unsigned)p.cv_W; \
                const bool ok1 = (unsigned)(cvy1 + cv_dy) < (unsigned)p.cv_H && (unsigned)(cvx1 + j + cv_dx) < (unsigned)p.cv_W; \
                vb0 = ok0 ? vb0 : 0.f; vb1 = ok1 ? vb1 : 0.f;                                                    \
            }                                                                                                    \
            if (TAIL) { vb0 = q0 < r_end ? vb0 : 0.f; vb1 = q1 < r_end ? vb1 : 0.f; }                            \
            xb0[j] = vb0; xb1[j] = vb1;                                                                          \
        }                                                                                                        \
        if constexpr (CV) {      /* the next K step is 32 rows further */                                        \
            cvx0 += kBK; while (cvx0 >= p.cv_W) { cvx0 -= p.cv_W; cvy0 = cvy0 + 1 == p.cv_H ? 0 : cvy0 + 1; }    \
            cvx1 += kBK; while (cvx1 >= p.cv_W) { cvx1 -= p.cv_W; cvy1 = cvy1 + 1 == p.cv_H ? 0 : cvy1 + 1; }    \
        }                                                                                                        \
    }
#define N2_LOAD(r0, TAIL) { N2_LOAD_A(r0, TAIL) N2_LOAD_B(r0, TAIL) }

    AccN2 acc;
    acc.zero();
    const int a_base = wr * 128 * 16, b_base = wc * 64 * 16;
    const int sl0 = (col ^ (2 * kc0)) * 16, sl1 = (col ^ (2 * kc0 + 4)) * 16;

    if (r_begin + kBK <= r_end) N2_LOAD(r_begin, false) else N2_LOAD(r_begin, true);
    int s = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += kBK, s ^= 1) {
        unsigned char* st = lds + s * kN2Stage;
        uint4 h, l;
        const bool full_next = r0 + 2 * kBK <= r_end, some_next = r0 + kBK < r_end;     // (uniform)
        // an operand's registers are requested again as soon as its split has consumed them — before the barrier, not after
        split8h(make_float4(xa0[0], xa0[1], xa0[2], xa0[3]), make_float4(xa0[4], xa0[5], xa0[6], xa0[7]), sc_a, &h, &l);
        *reinterpret_cast<uint4*>(st + (0 * 4 + kc0) * kN2Kc + sl0) = h;
        *reinterpret_cast<uint4*>(st + (1 * 4 + kc0) * kN2Kc + sl0) = l;
        split8h(make_float4(xa1[0], xa1[1], xa1[2], xa1[3]), make_float4(xa1[4], xa1[5], xa1[6], xa1[7]), sc_a, &h, &l);
        *reinterpret_cast<uint4*>(st + (0 * 4 + kc0 + 2) * kN2Kc + sl1) = h;
        *reinterpret_cast<uint4*>(st + (1 * 4 + kc0 + 2) * kN2Kc + sl1) = l;
        if (want_csa) {
#pragma unroll
            for (int j = 0; j < 8; ++j) csa += xa0[j] + xa1[j];
        }
        if (full_next) N2_LOAD_A(r0 + kBK, false) else if (some_next) N2_LOAD_A(r0 + kBK, true);
        split8h(make_float4(xb0[0], xb0[1], xb0[2], xb0[3]), make_float4(xb0[4], xb0[5], xb0[6], xb0[7]), sc_b, &h, &l);
        *reinterpret_cast<uint4*>(st + kN2Op + (0 * 4 + kc0) * kN2Kc + sl0) = h;
        *reinterpret_cast<uint4*>(st + kN2Op + (1 * 4 + kc0) * kN2Kc + sl0) = l;
        split8h(make_float4(xb1[0], xb1[1], xb1[2], xb1[3]), make_float4(xb1[4], xb1[5], xb1[6], xb1[7]), sc_b, &h, &l);
        *reinterpret_cast<uint4*>(st + kN2Op + (0 * 4 + kc0 + 2) * kN2Kc + sl1) = h;
        *reinterpret_cast<uint4*>(st + kN2Op + (1 * 4 + kc0 + 2) * kN2Kc + sl1) = l;
        if (full_next) N2_LOAD_B(r0 + kBK, false) else if (some_next) N2_LOAD_B(r0 + kBK, true);
        ws_barrier();                    // (LDS-only barrier: __syncthreads() would drain the loads just requested) stage s complete;
        acc.step(st, a_base, b_base, lane);      // everybody has left stage s (its previous use was two steps ago)
    }
#undef N2_LOAD
#undef N2_LOAD_A
#undef N2_LOAD_B

    // ---- epilogue: the split's partial tile ---------------------------------------------------------------------------------
    float* cp = p.c + (int64_t)sp * p.c_ss;
    {
        const int r32 = lane & 31, gh = lane >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + wr * 128 + i * 32 + r32, n = n0 + wc * 64 + j * 32 + 8 * q + 4 * gh;
                    float4 o = make_float4(acc.v[i][j][4 * q], acc.v[i][j][4 * q + 1], acc.v[i][j][4 * q + 2], acc.v[i][j][4 * q + 3]);
                    o = make_float4(o.x * inv_a * inv_b, o.y * inv_a * inv_b, o.z * inv_a * inv_b, o.w * inv_a * inv_b);
                    *reinterpret_cast<float4*>(cp + (int64_t)m * p.Ndim + n) = o;
                }
    }
    if (want_csa) {                     // column sums of A (the bias gradients): the two k-chunk halves of a column through LDS
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
        if (tid < 256) red[tid] = csa;
        __syncthreads();
        if (tid >= 256) red[col] += csa;
        __syncthreads();
        if (tid < 256) p.csum_a[(int64_t)sp * p.csa_ss + m0 + tid] = red[tid];
    }
}

__global__ __launch_bounds__(kN2T, 2) void gemm3_nt2_group_kernel(G3NG g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char n2_lds[];
    const int per_xcd = (g.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= g.ntiles) return;
    int i = 0;
    while (i + 1 < g.n_items && tile >= g.tile_end[i]) ++i;
    const int first = i ? g.tile_end[i - 1] : 0;
    gemm3_nt2_tile<false>(g.it[i], tile - first, n2_lds);
}

__global__ __launch_bounds__(kN2T, 2) void gemm3_nt2_conv_kernel(G3N p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char n2_lds[];
    const int per_xcd = (p.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    gemm3_nt2_tile<true>(p, tile, n2_lds);
}
