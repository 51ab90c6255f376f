// fp32 GEMM on the bf16 matrix cores by error-free operand splitting ("3xbf16, 6 products").
//
// The pixel decoder's encoder is pinned to fp32 by the reference (msdeformattn.py:314 disables
// autocast), and its Linear layers are the largest block of fp32 work of the step.  gfx950 has no
// reduced-precision path for fp32 operands (no xf32), and the exact fp32 MFMA runs at the vector rate
// (157 TFLOP/s).  Here every fp32 operand x is split into three bf16 pieces x = h + m + l (8 + 8 + 8
// mantissa bits, each piece exact), and the product a*b is evaluated as the six bf16 x bf16 products
// whose weight is >= 2^-16:  h*h + (h*m + m*h) + (h*l + l*h + m*m), each exact in the fp32 accumulator
// of v_mfma_f32_16x16x32_bf16; the dropped terms (m*l, l*m, l*l) are <= 2^-23 |a*b|, i.e. the size of
// the rounding of one fp32 product.  The result is an fp32-accurate GEMM (tests compare its error
// against fp64 with that of the library fp32 GEMM) at up to 6x fewer matrix-core cycles than fp32 MFMA.
//
//   C[M,N] = A[M,K] (+ A2[m % a2_rows, K]) . B[N,K]^T (+ bias[N]) (+ Cin[M,N]);  optional ReLU
//
// B (a weight matrix, small) is given pre-split as three bf16 planes [3][N][K] (mpf_gemm3_split);
// A (activations) is split on the fly while it is staged into LDS.
//
// Block = 128 x BN output tile (BN = 128 or 96), 4 waves as 2x2, wave tile 64 x BN/2, K step 32 (one
// MFMA).  LDS image per operand and plane: [k-chunk of 8][row][16 B], the row's 16-B slot XOR-swizzled
// with 2 * k-chunk: ds_write_b128 is serviced in groups of 8 contiguous lanes over 32 banks (here
// 2 rows x 4 k-chunks -> 8 distinct slots) and ds_read_b128 in the non-contiguous 16-lane groups of
// MI355X_MICROARCH.md (rows {0-3, 12-15} of one k-chunk with rows {4-11} of the next), and the
// swizzle permutes rows only inside aligned groups of 4, so both are bank-conflict free.  MFMA is issued as D^T = B.A^T so a lane
// owns 4 consecutive output columns of one row (16-B stores, bias as float4).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <stdint.h>

#include "mpf_common.h"
#include "amax.h"
#ifndef G3_PRIO
#define G3_PRIO 1
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kBM = 128;
constexpr int kBK = 32;
constexpr int kThreads = 256;
constexpr int kAKc = kBM * 16;                  // bytes per (plane, k-chunk) of the A image

struct G3 {
    const float* a;
    const float* a2;
    const unsigned short* bp;
    const float* bias;
    const float* cin;
    const float* cin2;
    const float* gate;
    float* c;
    int64_t lda, ldc, ldcin, ldcin2, ldgate, plane;
    int M, N, K, a2_rows, relu, tiles_n, ntiles;
    int bias_cm, cin_cm, cin2_cm, gate_cm;   // 1, or 0 when the operand is absent (it then points at g3_const)
    unsigned short* c16;                      // bf16 output instead of c (NULL: fp32)
    int cv_H, cv_W, cv_cin, cv_sign;          // 3x3 convolution mode (gemm3_conv_kernel): image size, channels per tap, +1 / -1
    int tm0, ntiles2, tiles_n2;      // mixed launch: row blocks >= tm0 are cut into ntiles2 tiles of 64 columns (tiles_n2 per row block)
    // fp16 x 2 form (H2 kernels): largest magnitudes of A and of the matrix the B planes were split from (device floats), and
    // an optional slot that receives max |C| (atomic max of the bit patterns: order-independent)
    const float* a_amax;
    const float* b_amax;
    float* out_amax;
    // ReLU gate as a BIT mask (fp16 x 2 entry mpf_gemm3_tn_h2_bits, N % 128 == 0): row m, bit n of [M][ldgbits bytes] = the
    // gate of C[m][n]; an absent mask points at 8 bytes of ones (g3_const) with leading dimension / column multiplier 0.
    // gbits_out (may be NULL): receives the mask of (C > 0) in the same layout.
    const unsigned char* gbits;
    unsigned char* gbits_out;
    int64_t ldgbits, ldgbits_out;
    int gbits_cm;
    int t3_bm;                                // gemm3_tn3_kernel: rows per tile, 192 or 176 (the second row half is then 80 rows)
};

int g_ablate = 0;      // mpf_set_option("gemm3_ablate"): reserved for timing experiments
int g_mixed = 1;       // mpf_set_option("gemm3_mixed_tiles"): 128 x 64 tiles for the last partial round
int g_two_pass = 256;  // mpf_set_option("gemm3_two_pass"): N >= this and N % 256 == 0 -> 128 x 256 / 96 x 256 two-pass tiles (0 = off)
int g_two_pass_rows = 0;   // mpf_set_option("gemm3_two_pass_rows"): 0 = pick 128 or 96 rows per tile by rounds, else force
int g_nt2 = 1;             // mpf_set_option("gemm3_nt2"): grouped fp16 x 2 weight gradients with all dimensions % 256 == 0 on 256 x 256 tiles (0 = 128 x 128)
int g_tn3 = 1;             // mpf_set_option("gemm3_tn3"): fp16 x 2 TN products with N % 256 == 0 on 192 x 256 tiles, one 8-wave workgroup per CU, where the tiles fill the chip (2: wherever M >= 2048; 0: never)
int g_tn3_176 = 1;         // mpf_set_option("gemm3_tn3_176"): gemm3_tn3_kernel on 176-row tiles where that costs no extra round (0 = always 192)
int g_ws = 512;            // mpf_set_option("gemm3_ws"): K = 256, N % 256 == 0, N >= this: fp16 x 2 products on the weight-stationary kernel (0 = never)


__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b)       // {b.hi16, a.hi16}
{
    return __builtin_amdgcn_perm(b, a, 0x07060302u);
}

// 8 floats -> three planes of 8 bf16 (truncating splits: every piece is exact; last piece rounded)
__device__ __forceinline__ void split8(const float4 u, const float4 v, uint4* h, uint4* m, uint4* l)
{
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    unsigned hb[8], mb[8], lb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned xb = __float_as_uint(x[i]);
        hb[i] = xb;
        const float r1 = x[i] - __uint_as_float(xb & 0xFFFF0000u);
        const unsigned rb = __float_as_uint(r1);
        mb[i] = rb;
        const float r2 = r1 - __uint_as_float(rb & 0xFFFF0000u);
        lb[i] = __float_as_uint(r2) + 0x8000u;
    }
    *h = make_uint4(pack_hi16(hb[0], hb[1]), pack_hi16(hb[2], hb[3]), pack_hi16(hb[4], hb[5]), pack_hi16(hb[6], hb[7]));
    *m = make_uint4(pack_hi16(mb[0], mb[1]), pack_hi16(mb[2], mb[3]), pack_hi16(mb[4], mb[5]), pack_hi16(mb[6], mb[7]));
    *l = make_uint4(pack_hi16(lb[0], lb[1]), pack_hi16(lb[2], lb[3]), pack_hi16(lb[4], lb[5]), pack_hi16(lb[6], lb[7]));
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// 8 floats -> two planes of 8 fp16 of (scale * x): h = fp16(s) and l = fp16(s - h), both rounded to nearest — 11 + 11 (+ 1
// from the rounding of h) significand bits, |s - h - l| <= 2^-23 |s| while l stays a normal fp16 number.  scale is a power of
// two chosen from the operand's largest magnitude (so that it lands in [2^14, 2^15)): exact, and undone in the epilogue.
__device__ __forceinline__ void split8h(const float4 u, const float4 v, const float scale, uint4* h, uint4* l)
{
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    unsigned hb[4], lb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 s = {x[2 * i] * scale, x[2 * i + 1] * scale};
        const f16x2 hh = __builtin_convertvector(s, f16x2);
        const f32x2 back = __builtin_convertvector(hh, f32x2);
        const f32x2 r = {s[0] - back[0], s[1] - back[1]};
        const f16x2 ll = __builtin_convertvector(r, f16x2);
        union { f16x2 f; unsigned u; } ch, cl;
        ch.f = hh; cl.f = ll;
        hb[i] = ch.u; lb[i] = cl.u;
    }
    *h = make_uint4(hb[0], hb[1], hb[2], hb[3]);
    *l = make_uint4(lb[0], lb[1], lb[2], lb[3]);
}

// power-of-two scale that puts amax into [2^14, 2^15) (fp16 tops out at 65504) and its inverse, from the exponent field:
// exact, and the same for every kernel that looks at the same amax.  amax below 2^-97 (incl. 0) counts as 2^-97; inf / nan
// give a finite scale (the values themselves stay inf / nan and so does the result).
__device__ __forceinline__ void h2_scale(const unsigned amax_bits, float* scale, float* inv)
{
    int e = (int)((amax_bits >> 23) & 0xffu);
    e = max(e, 30);
    *scale = __uint_as_float((unsigned)(268 - e) << 23);       // 2^(14 - (e - 127))
    *inv = __uint_as_float((unsigned)(e - 14) << 23);          // 2^-(14 - (e - 127))
}

__device__ __forceinline__ f16x8 as_fragh(const uint4 v)
{
    union { uint4 u; f16x8 f; } c;
    c.u = v;
    return c.f;
}

__device__ __forceinline__ float4 add4(float4 x, float4 y) { return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }

__device__ __forceinline__ bf16x8 as_frag(const uint4 v)
{
    union { uint4 u; bf16x8 f; } c;
    c.u = v;
    return c.f;
}

// Accumulators of a wave's 64 x (16 NJ) output tile, one K step (32) of MFMAs from the LDS images, and
// the walk over the lane's results.  D^T = B . A^T (a lane owns 4 consecutive output columns of one
// row), smallest of the six products first.
//   generic: 4 x NJ tiles of v_mfma_f32_16x16x32_bf16 (the TN kernels; NJ = 3 for 96-column tiles);
//   Acc<4, true>: 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 — half the MFMA issues; measured ~4 % faster in
//     the NT kernel and ~8 % slower in the TN kernel, so only the NT kernel uses it.
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NJ, bool W32 = false>
struct Acc {
    f32x4 v[4][NJ];

    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // a_base / b_base: byte offset of the wave's first row in the A / B image
    // BROW: the B image is row-major — [plane][column n][k-chunk slot (kc ^ sigma(n))][16 B], sigma(n) = (-(n >> 2)) & 3 —
    // the layout a DMA piece (64 lanes x 16 B, lane-linear in LDS) fills with 4 consecutive lanes reading the 64
    // contiguous bytes of one weight row; sigma makes the 16-lane service groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...)
    // hit 16 different bank quads.
    struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

    // hook(j) is called after the MFMAs of column tile j (plain six-product form only; tools/ubench/gemm3_pipe.hip)
    template <int AKC, int BKC, bool ONE = false, bool A1 = false, bool BROW = false, typename Hook = NoHook>
    __device__ __forceinline__ void step(const unsigned char* lds, int a_base, int b_base, int lane, Hook hook = Hook())
    {
        const int r16 = lane & 15, g = lane >> 4;
        constexpr int BJ = BROW ? 1024 : 256;        // bytes between the 16-column tiles of the B image
        const int a_frag = a_base + g * AKC + (r16 ^ (2 * g)) * 16;
        const int b_frag = BROW ? b_base + (r16 * 4 + (g ^ ((0 - (r16 >> 2)) & 3))) * 16 : b_base + g * BKC + (r16 ^ (2 * g)) * 16;
        if (A1) {       // A is a bf16 matrix (its plane 0 is exact), B a split fp32 one: three products, smallest first
            bf16x8 fa0[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa0[i] = as_frag(*reinterpret_cast<const uint4*>(lds + a_frag + i * 256));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                bf16x8 fb[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fb[pl] = as_frag(*reinterpret_cast<const uint4*>(lds + b_frag + pl * 4 * BKC + j * BJ));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa0[i], v[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa0[i], v[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa0[i], v[i][j], 0, 0, 0);
            }
            return;
        }
        if (ONE) {      // bf16 operands: plane 0 is the whole value
            bf16x8 fa0[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa0[i] = as_frag(*reinterpret_cast<const uint4*>(lds + a_frag + i * 256));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bf16x8 fb0 = as_frag(*reinterpret_cast<const uint4*>(lds + b_frag + j * BJ));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0, fa0[i], v[i][j], 0, 0, 0);
            }
            return;
        }
        bf16x8 fa[3][4];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[pl][i] = as_frag(*reinterpret_cast<const uint4*>(lds + a_frag + pl * 4 * AKC + i * 256));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bf16x8 fb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb[pl] = as_frag(*reinterpret_cast<const uint4*>(lds + b_frag + pl * 4 * BKC + j * BJ));
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[2][i], v[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[0][i], v[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[1][i], v[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[1][i], v[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[0][i], v[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[0][i], v[i][j], 0, 0, 0);
            hook(j);
        }
    }

    // two fp16 pieces per operand (planes 0, 1 of the images): l*h + h*l + h*h, smallest first
    template <int AKC, int BKC, bool BROW = false>
    __device__ __forceinline__ void step_h2(const unsigned char* lds, int a_base, int b_base, int lane)
    {
        const int r16 = lane & 15, g = lane >> 4;
        constexpr int BJ = BROW ? 1024 : 256;
        const int a_frag = a_base + g * AKC + (r16 ^ (2 * g)) * 16;
        const int b_frag = BROW ? b_base + (r16 * 4 + (g ^ ((0 - (r16 >> 2)) & 3))) * 16 : b_base + g * BKC + (r16 ^ (2 * g)) * 16;
        f16x8 fa[2][4];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[pl][i] = as_fragh(*reinterpret_cast<const uint4*>(lds + a_frag + pl * 4 * AKC + i * 256));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f16x8 fb[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[pl] = as_fragh(*reinterpret_cast<const uint4*>(lds + b_frag + pl * 4 * BKC + j * BJ));
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1], fa[0][i], v[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0], fa[1][i], v[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0], fa[0][i], v[i][j], 0, 0, 0);
        }
    }

    // f(row offset in the wave tile, column offset in the wave tile, the 4 values of that row at columns +0..3)
    template <typename F>
    __device__ __forceinline__ void quads(int lane, F f) const
    {
        const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) f(i * 16 + r16, j * 16 + g * 4, make_float4(v[i][j][0], v[i][j][1], v[i][j][2], v[i][j][3]));
    }
};

template <>
struct Acc<4, true> {
    f32x16 v[2][2];

    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) v[i][j][e] = 0.f;
    }

    // A1 / B1: that operand is a bf16 matrix (only its plane 0 is non-zero): the products with its planes 1, 2 are skipped
    template <int AKC, int BKC, bool ONE = false, bool A1 = false, bool B1 = false>
    __device__ __forceinline__ void step(const unsigned char* lds, int a_base, int b_base, int lane)
    {
        const int r32 = lane & 31, gh = lane >> 5;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int kc = 2 * kh + gh;                       // this lane's k-chunk (8 values) of the 16-deep MFMA
            const int sw = (r32 ^ (2 * kc)) * 16;
            if (ONE) {      // bf16 operands: plane 0 is the whole value
                bf16x8 fa0[2], fb0[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa0[t] = as_frag(*reinterpret_cast<const uint4*>(lds + a_base + kc * AKC + t * 512 + sw));
                    fb0[t] = as_frag(*reinterpret_cast<const uint4*>(lds + b_base + kc * BKC + t * 512 + sw));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0[j], fa0[i], v[i][j], 0, 0, 0);
                continue;
            }
            bf16x8 fa[3][2], fb[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (pl == 0 || !A1) fa[pl][t] = as_frag(*reinterpret_cast<const uint4*>(lds + a_base + (pl * 4 + kc) * AKC + t * 512 + sw));
                    if (pl == 0 || !B1) fb[pl][t] = as_frag(*reinterpret_cast<const uint4*>(lds + b_base + (pl * 4 + kc) * BKC + t * 512 + sw));
                }
#define G3_MMA32(PB, PA)                                                                                          \
    if ((PA == 0 || !A1) && (PB == 0 || !B1)) {                                                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int i = 0; i < 2; ++i)               \
            v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB][j], fa[PA][i], v[i][j], 0, 0, 0);            \
    }
            G3_MMA32(0, 2) G3_MMA32(2, 0) G3_MMA32(1, 1) G3_MMA32(0, 1) G3_MMA32(1, 0) G3_MMA32(0, 0)
#undef G3_MMA32
        }
    }

    template <int AKC, int BKC>
    __device__ __forceinline__ void step_h2(const unsigned char* lds, int a_base, int b_base, int lane)
    {
        const int r32 = lane & 31, gh = lane >> 5;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int kc = 2 * kh + gh;
            const int sw = (r32 ^ (2 * kc)) * 16;
            f16x8 fa[2][2], fb[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[pl][t] = as_fragh(*reinterpret_cast<const uint4*>(lds + a_base + (pl * 4 + kc) * AKC + t * 512 + sw));
                    fb[pl][t] = as_fragh(*reinterpret_cast<const uint4*>(lds + b_base + (pl * 4 + kc) * BKC + t * 512 + sw));
                }
#define G3_MMA32H(PB, PA)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int i = 0; i < 2; ++i)                   \
        v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[PB][j], fa[PA][i], v[i][j], 0, 0, 0);
            G3_MMA32H(1, 0) G3_MMA32H(0, 1) G3_MMA32H(0, 0)
#undef G3_MMA32H
        }
    }

    template <typename F>
    __device__ __forceinline__ void quads(int lane, F f) const
    {
        const int r32 = lane & 31, gh = lane >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    f(i * 32 + r32, j * 32 + 8 * q + 4 * gh,
                      make_float4(v[i][j][4 * q], v[i][j][4 * q + 1], v[i][j][4 * q + 2], v[i][j][4 * q + 3]));
    }
};

// epilogue of the TN kernels: bias / addends / ReLU / gate, 16-byte stores (m_wave, n_wave: first row / column of the wave tile)
// Every operand of the epilogue is read UNCONDITIONALLY: an absent one points at a 16-byte block of zeros (addends, bias)
// or ones (gate) with leading dimension and column multiplier 0, and rows / columns past the edge are clamped (only the
// store is predicated).  With `if (p.cin) load` per quad hipcc branched around each load and waited for it on its own:
// up to 64 dependent memory round trips per wave in the epilogue of a kernel whose main loop takes 8.  Loads of one
// 16-column group (bias + 4 row tiles x 3 operands) are requested together.
// (bit patterns: four zeros, four ones (1.0f), 16 bytes of 0xFF = the absent bit-mask gate)
__device__ __attribute__((aligned(16))) unsigned g3_const[12] = {0u, 0u, 0u, 0u, 0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u,
                                    0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};

// device address of g3_const ON THE CURRENT DEVICE: a __device__ variable has one instance per GPU, so the address is
// cached per device ordinal (a process-wide cache would hand a second GPU a pointer into the first one's memory)
int g3_consts(const float** out, const char* who)
{
    constexpr int kMaxDev = 64;
    static std::atomic<const float*> cache[kMaxDev];
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return mpf::check(e, who);
    const float* c = (dev >= 0 && dev < kMaxDev) ? cache[dev].load(std::memory_order_acquire) : nullptr;
    if (!c) {
        void* sym = nullptr;
        if (hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g3_const)); e != hipSuccess) return mpf::check(e, who);
        c = (const float*)sym;
        if (dev >= 0 && dev < kMaxDev) cache[dev].store(c, std::memory_order_release);     // (racing threads store the same value)
    }
    *out = c;
    return 0;
}

inline void g3_no_bits(G3& p, const float* consts)
{
    p.gbits = reinterpret_cast<const unsigned char*>(consts + 8); p.ldgbits = 0; p.gbits_cm = 0; p.gbits_out = nullptr; p.ldgbits_out = 0;
}

// returns the largest |value| this thread stored (SCALED kernels; 0 otherwise)
// PIPE (round 6): the operands of column tile j + 1 (bias, addends, gate) are REQUESTED BEFORE the stores of column tile j.  On gfx9
// loads and stores share one counter (vmcnt) and complete in issue order, so a load issued behind stores can only be waited for by
// waiting for those stores to be acknowledged by memory — in the plain form every column tile of every epilogue call did that
// (8 drains per wave in gemm3_tn3_kernel), and for the waves that reach their epilogue last, with every CU's stores in flight, a
// drain is microseconds: phase stamps of waves 4-7 showed the epilogue at 45 % (K = 1 024) to 75 % (K = 256) of the wave's life while
// waves 0-3, whose stores drain under the others' MFMAs, spent 16 %.  With the next tile's loads older than this tile's stores the
// wait for them leaves the stores in flight.  Same arithmetic, same order: bit-identical.
template <int NJ, int NI = 4, typename AccT = Acc<NJ, false>, bool SCALED = false, bool PIPE = false>
__device__ __forceinline__ float g3_epilogue(const G3& p, const AccT& acc, int lane, int m_wave, int n_wave, const float inv_a = 1.f,
                                            const float inv_b = 1.f, const int m_end = 0x7fffffff)
{
    float amax = 0.f;
    const int r16 = lane & 15, g = lane >> 4;
    int64_t mrow[NI];
    bool mok[NI];
    uint2 gb[NI];                // the wave tile's 64 (NJ <= 2: 32) gate bits of every row (all ones when there is no bit mask)
    unsigned wb[NI][2];          // (C > 0) of the same 64 columns, this lane's nibbles
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = m_wave + i * 16 + r16;
        mok[i] = m < p.M && m < m_end;           // m_end: rows from there on belong to the next tile (gemm3_tn3_kernel's 176-row tiles)
        mrow[i] = min(m, p.M - 1);
        if constexpr (NJ <= 2) {                 // a 32-column wave tile: 4 bytes of the row's mask
            gb[i].x = *reinterpret_cast<const unsigned*>(p.gbits + mrow[i] * p.ldgbits + (n_wave >> 3) * p.gbits_cm);
            gb[i].y = 0u;
        } else {
            gb[i] = *reinterpret_cast<const uint2*>(p.gbits + mrow[i] * p.ldgbits + (n_wave >> 3) * p.gbits_cm);
        }
        wb[i][0] = wb[i][1] = 0u;
    }
    struct Ops { float4 bz, ci[NI], c2[NI], gt[NI]; };
    auto load_ops = [&](const int j, Ops& q) {
        const int nc = min(n_wave + j * 16 + g * 4, p.N - 4);
        q.bz = *reinterpret_cast<const float4*>(p.bias + nc * p.bias_cm);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            q.ci[i] = *reinterpret_cast<const float4*>(p.cin + mrow[i] * p.ldcin + nc * p.cin_cm);
            q.c2[i] = *reinterpret_cast<const float4*>(p.cin2 + mrow[i] * p.ldcin2 + nc * p.cin2_cm);
            q.gt[i] = *reinterpret_cast<const float4*>(p.gate + mrow[i] * p.ldgate + nc * p.gate_cm);
        }
    };
    Ops ops[2];
    if constexpr (PIPE) load_ops(0, ops[0]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n_wave + j * 16 + g * 4;
        const bool nok = n < p.N;                    // N % 4 == 0: a quad is inside or outside as a whole
        if constexpr (PIPE) {
            if (j + 1 < NJ) load_ops(j + 1, ops[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);       // (keep the requests of tile j + 1 in front of the stores of tile j)
        } else {
            load_ops(j, ops[j & 1]);
        }
        const float4 bz = ops[j & 1].bz;
        const float4(&ci)[NI] = ops[j & 1].ci;
        const float4(&c2)[NI] = ops[j & 1].c2;
        const float4(&gt)[NI] = ops[j & 1].gt;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float4 o = make_float4(acc.v[i][j][0], acc.v[i][j][1], acc.v[i][j][2], acc.v[i][j][3]);
            if constexpr (SCALED)      // two factors: each is a representable power of two, their product need not be
                o = make_float4(o.x * inv_a * inv_b, o.y * inv_a * inv_b, o.z * inv_a * inv_b, o.w * inv_a * inv_b);
            // same order of additions as before: bias, addend 1, addend 2
            o = make_float4(o.x + bz.x, o.y + bz.y, o.z + bz.z, o.w + bz.w);
            o = make_float4(o.x + ci[i].x, o.y + ci[i].y, o.z + ci[i].z, o.w + ci[i].w);
            o = make_float4(o.x + c2[i].x, o.y + c2[i].y, o.z + c2[i].z, o.w + c2[i].w);
            if (p.relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            // ReLU backward: pass the gradient where the saved activation is > 0
            o = make_float4(gt[i].x > 0.f ? o.x : 0.f, gt[i].y > 0.f ? o.y : 0.f, gt[i].z > 0.f ? o.z : 0.f, gt[i].w > 0.f ? o.w : 0.f);
            // the same gate as a bit mask: column n_wave + 16 j + 4 g + t = bit (16 (j & 1) + 4 g + t) of word j >> 1
            // (uniform branches around register arithmetic only: the launches without masks do not pay for it)
            if (p.gbits_cm) {
                const unsigned nib = ((j & 2) ? gb[i].y : gb[i].x) >> ((j & 1) * 16 + g * 4);
                o = make_float4((nib & 1u) ? o.x : 0.f, (nib & 2u) ? o.y : 0.f, (nib & 4u) ? o.z : 0.f, (nib & 8u) ? o.w : 0.f);
            }
            if (p.gbits_out) {
                const unsigned pos = (o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u);
                wb[i][(j >> 1) & 1] |= pos << ((j & 1) * 16 + g * 4);
            }
            if (mok[i] && nok) {
                if constexpr (SCALED) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                if (p.c16) {                 // (uniform; stores only) round to nearest even
                    unsigned w[4];
                    const float e[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        unsigned u = __float_as_uint(e[t]);
                        u += 0x7fffu + ((u >> 16) & 1u);
                        w[t] = u >> 16;
                    }
                    *reinterpret_cast<uint2*>(p.c16 + mrow[i] * p.ldc + n) = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
                } else {
                    *reinterpret_cast<float4*>(p.c + mrow[i] * p.ldc + n) = o;
                }
            }
        }
    }
    if (p.gbits_out) {           // (uniform) the four lane groups of a row hold disjoint nibbles: OR them, group 0 stores 8 bytes
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            unsigned lo = wb[i][0], hi = wb[i][1];
            lo |= __shfl_xor(lo, 16); hi |= __shfl_xor(hi, 16);
            lo |= __shfl_xor(lo, 32); hi |= __shfl_xor(hi, 32);
            if (g == 0 && mok[i] && n_wave < p.N) {
                if constexpr (NJ <= 2) *reinterpret_cast<unsigned*>(p.gbits_out + mrow[i] * p.ldgbits_out + (n_wave >> 3)) = lo;
                else *reinterpret_cast<uint2*>(p.gbits_out + mrow[i] * p.ldgbits_out + (n_wave >> 3)) = make_uint2(lo, hi);
            }
        }
    }
    return amax;
}

// one DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_addr, lds_addr + 1024) (M0 = LDS address
// of lane 0).  Inline asm, not __builtin_amdgcn_global_load_lds: hipcc treats the builtin as an LDS write it cannot
// disambiguate and puts s_waitcnt vmcnt(0) in front of the next ds_read, which would drain the A loads in flight.
// Address = scalar base (SGPR pair) + per-lane 32-bit byte offset: no 64-bit vector arithmetic per piece.
__device__ __forceinline__ void glds16(const void* base, unsigned voff, unsigned lds_addr)
{
    // M0 is a reserved register for inline asm (a clobber on it is "undefined behaviour" to hipcc): the statement saves the
    // compiler's M0 in an SGPR it allocates and restores it, so no clobber is declared and nothing is assumed about M0
    unsigned saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}

#include "gemm3_ws.h"
#include "gemm3_tn3.h"

// one output tile: 128 rows x BN columns starting at (m0, n0)
// CV: A is a channel-last image [N_img*H*W][Cin] and K runs over (tap, channel) of a 3x3 window — K step kt reads the
// rows shifted by the tap's (dy, dx) (sign = -1: the transposed convolution of the input gradient); taps that fall off
// the image contribute zeros (the loads stay unconditional on clamped rows, the registers are zeroed before the split).
// ABF: A is a bf16 matrix (p.a points at 2-byte elements, lda in elements): its rows go to plane 0 of the LDS image as they
// are (one 16-byte load per row and k-chunk, no split) and a K step is three products instead of six.
template <int BN, bool A2, bool CV = false, bool ABF = false, bool H2 = false>
__device__ __forceinline__ void g3_tn_tile(const G3& p, unsigned char* lds, const int m0, const int n0)
{
    static_assert(!(ABF && (A2 || CV)), "bf16 A: no addend, no convolution mode");
    static_assert(!(H2 && (ABF || A2)), "fp16 x 2 form: fp32 A, no addend rows");
    constexpr int NJ = BN / 32;                  // 16-column MFMA tiles per wave
    constexpr int kBKc = BN * 16;                // bytes per (plane, k-chunk) of the B image
    constexpr int kPl = H2 ? 2 : 3;              // planes per operand
    constexpr int kAbytes = kPl * 4 * kAKc;
    constexpr int kBunits = 3 * BN * 4;          // 16-B units of a B stage
    constexpr int kBiter = (kBunits + kThreads - 1) / kThreads;
    // B (the pre-split weight planes, L2-resident) goes global -> LDS by DMA (global_load_lds, 64 lanes x 16 B = 1 KB per
    // wave instruction) into one of TWO stages, no registers and no ds_write: the 16-byte LDS stores were a quarter of this
    // kernel's time (a wave's ds_write_b128 occupies the VGPR -> LDS path for ~13 cycles; A + B were 12 per thread and K step)
    constexpr int kBstage = kPl * 4 * kBKc;      // bytes of one B stage
    constexpr int kPieces = kBstage / 1024;      // DMA pieces per stage (BN = 128: 24, 96: 18, 64: 12; two planes: 16, 12, 8)
    constexpr int kPW = (kPieces + 3) / 4;       // per wave (BN = 96: 5, the two surplus pieces repeat the last one)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;


    // ---- staging maps ---------------------------------------------------------------------------
    const int akc = tid & 3;
    const int arow0 = tid >> 2, arow1 = 64 + (tid >> 2);
    const int aslot0 = arow0 ^ (2 * akc), aslot1 = arow1 ^ (2 * akc);       // LDS slot swizzle (see header)
    // (ABF: the same expressions on 2-byte elements — half the byte offsets)
    const float* ap0 = ABF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p.a) +
                                                              (int64_t)min(m0 + arow0, p.M - 1) * p.lda + akc * 8)
                           : p.a + (int64_t)min(m0 + arow0, p.M - 1) * p.lda + akc * 8;
    const float* ap1 = ABF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p.a) +
                                                              (int64_t)min(m0 + arow1, p.M - 1) * p.lda + akc * 8)
                           : p.a + (int64_t)min(m0 + arow1, p.M - 1) * p.lda + akc * 8;
    const float* a2p0 = nullptr;
    const float* a2p1 = nullptr;
    if (p.a2) {
        a2p0 = p.a2 + (int64_t)(min(m0 + arow0, p.M - 1) % p.a2_rows) * p.K + akc * 8;
        a2p1 = p.a2 + (int64_t)(min(m0 + arow1, p.M - 1) % p.a2_rows) * p.K + akc * 8;
    }
    int cy0 = 0, cx0 = 0, cy1 = 0, cx1 = 0, mr0 = 0, mr1 = 0;
    if constexpr (CV) {
        mr0 = min(m0 + arow0, p.M - 1); mr1 = min(m0 + arow1, p.M - 1);
        cx0 = mr0 % p.cv_W; cy0 = (mr0 / p.cv_W) % p.cv_H;
        cx1 = mr1 % p.cv_W; cy1 = (mr1 / p.cv_W) % p.cv_H;
    }
    (void)ap0; (void)ap1;
    // piece q = wave + 4 i covers LDS units [64 q, 64 q + 64) of a stage: plane q / (BN / 16), columns 16 (q % (BN / 16)) ..+15;
    // lane -> column nl = lane >> 2, LDS slot lane & 3 = k-chunk ^ sigma(nl)
    unsigned boff[kPW];        // byte offset of the lane's 16 bytes from p.bp at k = 0 (the planes span < 4 GiB)
    int bpiece[kPW];           // LDS byte offset of the piece inside a stage (wave-uniform)
    {
        const int nl = lane >> 2, kc = (lane & 3) ^ ((0 - (nl >> 2)) & 3);
#pragma unroll
        for (int i = 0; i < kPW; ++i) {
            const int q = min(wave + 4 * i, kPieces - 1);
            const int pl = q / (BN / 16), nb = (q % (BN / 16)) * 16;
            boff[i] = (unsigned)(((int64_t)pl * p.plane + (int64_t)min(n0 + nb + nl, p.N - 1) * p.K + kc * 8) * 2);
            bpiece[i] = __builtin_amdgcn_readfirstlane(q * 1024);
        }
    }
    const unsigned lds_b = (unsigned)(uintptr_t)(lds + kAbytes);           // LDS address of B stage 0

    // A is prefetched TWO K-steps ahead (HBM latency is longer than one step of MFMAs; measured: with
    // one step of distance the staging and the MFMAs serialise), in two alternating register sets;
    // B (L2-resident planes) one step ahead.
    float4 raE[4], raO[4], ra2[4];
    (void)kBunits; (void)kBiter;
#define G3_LOAD_A(ra, k0)                                                                    \
    {                                                                                        \
        ra[0] = *reinterpret_cast<const float4*>(ap0 + (k0));                                \
        ra[1] = *reinterpret_cast<const float4*>(ap0 + (k0) + 4);                            \
        ra[2] = *reinterpret_cast<const float4*>(ap1 + (k0));                                \
        ra[3] = *reinterpret_cast<const float4*>(ap1 + (k0) + 4);                            \
    }
    // conv mode: loads of K step k0 (tap = k0 / Cin) and the validity of the two rows' taps (bit 0 / bit 1 of vm)
#define G3_LOAD_A_CV(ra, vm, k0)                                                             \
    {                                                                                        \
        const int tap_ = (k0) / p.cv_cin, kk_ = (k0) - tap_ * p.cv_cin;                      \
        const int dy_ = (tap_ / 3 - 1) * p.cv_sign, dx_ = (tap_ % 3 - 1) * p.cv_sign;        \
        const int sh_ = dy_ * p.cv_W + dx_;                                                  \
        const bool v0_ = (unsigned)(cy0 + dy_) < (unsigned)p.cv_H && (unsigned)(cx0 + dx_) < (unsigned)p.cv_W; \
        const bool v1_ = (unsigned)(cy1 + dy_) < (unsigned)p.cv_H && (unsigned)(cx1 + dx_) < (unsigned)p.cv_W; \
        /* 32-bit element offsets from the uniform base (host checks M * lda < 2^31): saddr + voffset addressing */ \
        const unsigned o0_ = (unsigned)(min(max(mr0 + sh_, 0), p.M - 1) * (int)p.lda + akc * 8 + kk_); \
        const unsigned o1_ = (unsigned)(min(max(mr1 + sh_, 0), p.M - 1) * (int)p.lda + akc * 8 + kk_); \
        ra[0] = *reinterpret_cast<const float4*>(p.a + o0_);                                 \
        ra[1] = *reinterpret_cast<const float4*>(p.a + o0_ + 4);                             \
        ra[2] = *reinterpret_cast<const float4*>(p.a + o1_);                                 \
        ra[3] = *reinterpret_cast<const float4*>(p.a + o1_ + 4);                             \
        vm = (v0_ ? 1 : 0) | (v1_ ? 2 : 0);                                                  \
    }
    // B stage `stage` <- K step k0: kPW DMA pieces per wave (inline asm: the compiler neither tracks nor waits for them;
    // G3_WAIT_B is the explicit wait).  The addend rows of A (A2) ride along as before.
#define G3_LOAD_B(k0, stage)                                                                 \
    {                                                                                        \
        if constexpr (A2) {                                                                  \
            ra2[0] = *reinterpret_cast<const float4*>(a2p0 + (k0));                          \
            ra2[1] = *reinterpret_cast<const float4*>(a2p0 + (k0) + 4);                      \
            ra2[2] = *reinterpret_cast<const float4*>(a2p1 + (k0));                          \
            ra2[3] = *reinterpret_cast<const float4*>(a2p1 + (k0) + 4);                      \
        }                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < kPW; ++i_)                                   \
            glds16(p.bp + (k0), boff[i_], lds_b + (stage) * kBstage + bpiece[i_]);           \
    }
#define G3_LA(ra, vm, k0)                                                                    \
    {                                                                                        \
        if constexpr (ABF) {        /* 8 bf16 = 16 bytes per row: (k0) elements = (k0) / 2 floats */ \
            ra[0] = *reinterpret_cast<const float4*>(ap0 + (k0) / 2);                        \
            ra[2] = *reinterpret_cast<const float4*>(ap1 + (k0) / 2);                        \
        } else if constexpr (CV) G3_LOAD_A_CV(ra, vm, k0) else G3_LOAD_A(ra, k0)             \
    }
#define G3_WRITE(ra, vm)                                                                     \
    {                                                                                        \
        uint4 h, m, l;                                                                       \
        /* (temporaries, not writes into ra[]: a float4 array written under a condition goes to scratch) */ \
        float4 w0_ = ra[0], w1_ = ra[1], w2_ = ra[2], w3_ = ra[3];                           \
        if constexpr (CV) {                                                                  \
            const float s0_ = (vm & 1) ? 1.f : 0.f, s1_ = (vm & 2) ? 1.f : 0.f;              \
            w0_ = make_float4(w0_.x * s0_, w0_.y * s0_, w0_.z * s0_, w0_.w * s0_);           \
            w1_ = make_float4(w1_.x * s0_, w1_.y * s0_, w1_.z * s0_, w1_.w * s0_);           \
            w2_ = make_float4(w2_.x * s1_, w2_.y * s1_, w2_.z * s1_, w2_.w * s1_);           \
            w3_ = make_float4(w3_.x * s1_, w3_.y * s1_, w3_.z * s1_, w3_.w * s1_);           \
        }                                                                                    \
        if constexpr (A2) {                                                                  \
            w0_ = add4(w0_, ra2[0]); w1_ = add4(w1_, ra2[1]); w2_ = add4(w2_, ra2[2]); w3_ = add4(w3_, ra2[3]); \
        }                                                                                    \
        if constexpr (ABF) {                                                                 \
            *reinterpret_cast<float4*>(lds + (0 * 4 + akc) * kAKc + aslot0 * 16) = w0_;      \
            *reinterpret_cast<float4*>(lds + (0 * 4 + akc) * kAKc + aslot1 * 16) = w2_;      \
        } else if constexpr (H2) {                                                           \
            split8h(w0_, w1_, sc_a, &h, &l);                                                 \
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKc + aslot0 * 16) = h;         \
            *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKc + aslot0 * 16) = l;         \
            split8h(w2_, w3_, sc_a, &h, &l);                                                 \
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKc + aslot1 * 16) = h;         \
            *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKc + aslot1 * 16) = l;         \
        } else {                                                                             \
        split8(w0_, w1_, &h, &m, &l);                                                        \
        *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKc + aslot0 * 16) = h;             \
        *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKc + aslot0 * 16) = m;             \
        *reinterpret_cast<uint4*>(lds + (2 * 4 + akc) * kAKc + aslot0 * 16) = l;             \
        split8(w2_, w3_, &h, &m, &l);                                                        \
        *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKc + aslot1 * 16) = h;             \
        *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKc + aslot1 * 16) = m;             \
        *reinterpret_cast<uint4*>(lds + (2 * 4 + akc) * kAKc + aslot1 * 16) = l;             \
        }                                                                                    \
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ra2[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    float sc_a = 1.f, inv_a = 1.f, inv_b = 1.f;
    if constexpr (H2) {
        float sc_b;
        h2_scale(amax_read(p.a_amax), &sc_a, &inv_a);
        h2_scale(amax_read(p.b_amax), &sc_b, &inv_b);
    }
    (void)sc_a;
    Acc<NJ> acc;
    acc.zero();

    const int a_frag = wr * 64 * 16;                        // byte offset of the wave's first row in the A image
    const int b_frag = kAbytes + wc * (BN / 2) * 64;        // ... and of its first column in B stage 0 (row-major image)

    const int nk = p.K / kBK;
    // All loads are unconditional (the K index is clamped; a surplus load is never stored): with
    // loads inside branches hipcc cannot count them and falls back to vmcnt(0) before the LDS writes,
    // which would drain the two-steps-ahead A loads every step.
    const int klast = (nk - 1) * kBK;
    int vmE = 3, vmO = 3;
    // order of the vector-memory operations of a step: [B DMA of the next step] [A loads two steps ahead].  Memory
    // operations complete in order, so "at most the 4 (ABF: 2) youngest outstanding" = the DMA has landed.
    constexpr int kAloads = ABF ? 2 : 4;
#define G3_WAIT_B() { if constexpr (kAloads == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    G3_LA(raE, vmE, 0);
    __builtin_amdgcn_sched_barrier(0);
    G3_LOAD_B(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    G3_LA(raO, vmO, min(kBK, klast));
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; kt += 2) {
        __syncthreads();                            // everyone is done with the A image and with B stage 1
        G3_WRITE(raE, vmE);
        G3_WAIT_B();                                // my pieces of B stage 0 (K step kt)
        __syncthreads();
        G3_LOAD_B(min((kt + 1) * kBK, klast), 1);   // B first: the wait above counts on this order
        __builtin_amdgcn_sched_barrier(0);
        G3_LA(raE, vmE, min((kt + 2) * kBK, klast));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(G3_PRIO);
        if constexpr (H2) acc.template step_h2<kAKc, kBKc, true>(lds, a_frag, b_frag, lane);
        else acc.template step<kAKc, kBKc, false, ABF, true>(lds, a_frag, b_frag, lane);
        __builtin_amdgcn_s_setprio(0);
        if (kt + 1 >= nk) break;
        __syncthreads();
        G3_WRITE(raO, vmO);
        G3_WAIT_B();
        __syncthreads();
        G3_LOAD_B(min((kt + 2) * kBK, klast), 0);
        __builtin_amdgcn_sched_barrier(0);
        G3_LA(raO, vmO, min((kt + 3) * kBK, klast));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(G3_PRIO);
        if constexpr (H2) acc.template step_h2<kAKc, kBKc, true>(lds, a_frag, b_frag + kBstage, lane);
        else acc.template step<kAKc, kBKc, false, ABF, true>(lds, a_frag, b_frag + kBstage, lane);
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (surplus DMA pieces of the clamped last steps must not outlive the LDS allocation)
#undef G3_WAIT_B

    const float omax = g3_epilogue<NJ, 4, Acc<NJ, false>, H2>(p, acc, lane, m0 + wr * 64, n0 + wc * (BN / 2), inv_a, inv_b);
    if constexpr (H2) {
        if (p.out_amax) amax_commit(p.out_amax, omax, reinterpret_cast<float*>(lds));
    }
    (void)omax;
}
#undef G3_LOAD_A
#undef G3_LOAD_A_CV
#undef G3_LA
#undef G3_LOAD_B
#undef G3_WRITE

template <int BN, bool A2, bool H2 = false>
__global__ __launch_bounds__(kThreads, 2) void gemm3_tn_kernel(G3 p)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[(H2 ? 8 : 12) * (kAKc + 2 * BN * 16)];       // A image + two B stages
    // XCD-aware tile order: each XCD walks a contiguous run of tiles (column tiles of one row block
    // are neighbours, so the A rows they share stay in that XCD's L2)
    const int per_xcd = (p.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    g3_tn_tile<BN, A2, false, false, H2>(p, lds, tm * kBM, tn * BN);
}

// ------------------------------------------------------------------------------------------------
// 128 x 256 tile as TWO 128-column passes over ONE A image per K step (gemm3_tn2_kernel).
// Measured on the 128 x 128 kernel: the part of a K step that does not depend on the tile width (staging A: loads, split,
// LDS stores) is worth ~91 columns of MFMA time.  Here it is paid once per 256 columns inside the same 72 KB of LDS: B
// stage 0 holds columns 0-127 of the K step, stage 1 columns 128-255; the A fragments are read once and stay in registers
// for both passes; the stage a pass has released is refilled by DMA at once (three barriers per K step):
//   barrier a | split + write A(k) | DMA half 1 of k -> stage 1 | wait half 0 of k | barrier b | A loads of k + 2 |
//   pass 0 (stage 0) | wait half 1 | barrier c | DMA half 0 of k + 1 -> stage 0 | pass 1 (stage 1)
// Same products in the same order per output element as the 128 x 128 tile: bit-identical results.
template <int NI, int AKC, bool F16 = false>
struct Acc2 {
    f32x4 v[2][NI][4];
    bf16x8 fa[F16 ? 1 : 3][NI];
    f16x8 fh[F16 ? 2 : 1][NI];

    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void load_a(const unsigned char* lds, int a_base, int lane)
    {
        const int r16 = lane & 15, g = lane >> 4;
        const int a_frag = a_base + g * AKC + (r16 ^ (2 * g)) * 16;
        if constexpr (F16) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < NI; ++i) fh[pl][i] = as_fragh(*reinterpret_cast<const uint4*>(lds + a_frag + pl * 4 * AKC + i * 256));
            return;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < NI; ++i) fa[pl][i] = as_frag(*reinterpret_cast<const uint4*>(lds + a_frag + pl * 4 * AKC + i * 256));
    }
    template <int H>
    __device__ __forceinline__ void pass(const unsigned char* lds, int b_base, int lane)
    {
        constexpr int BKC = 128 * 16;
        const int r16 = lane & 15, g = lane >> 4;
        const int b_frag = b_base + (r16 * 4 + (g ^ ((0 - (r16 >> 2)) & 3))) * 16;
        if constexpr (F16) {       // two fp16 pieces per operand: l*h + h*l + h*h, smallest first
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f16x8 fb[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fb[pl] = as_fragh(*reinterpret_cast<const uint4*>(lds + b_frag + pl * 4 * BKC + j * 1024));
#pragma unroll
                for (int i = 0; i < NI; ++i) v[H][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1], fh[0][i], v[H][i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NI; ++i) v[H][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0], fh[1][i], v[H][i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NI; ++i) v[H][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0], fh[0][i], v[H][i][j], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x8 fb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[pl] = as_frag(*reinterpret_cast<const uint4*>(lds + b_frag + pl * 4 * BKC + j * 1024));
#pragma unroll
            for (int i = 0; i < NI; ++i) v[H][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[2][i], v[H][i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) v[H][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[0][i], v[H][i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) v[H][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[1][i], v[H][i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) v[H][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[1][i], v[H][i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) v[H][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[0][i], v[H][i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) v[H][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[0][i], v[H][i][j], 0, 0, 0);
        }
    }
};

// BM = 128, or 96: M = 43 008 is 336 row blocks of 128 for the chip's 512 workgroup slots, but 448 of 96 (the rows 64-95 of
// the A tile are staged by the first two waves)
// CV: the 3x3 convolution mode of g3_tn_tile (K step kt belongs to tap kt / (Cin / 32); rows shifted by the tap, taps off
// the image zeroed before the split)
template <int BM, bool CV = false, bool F16 = false>
__global__ __launch_bounds__(kThreads, 2) void gemm3_tn2_kernel(G3 p)
{
    static_assert(BM == 128 || BM == 96, "row blocks of 128 or 96");
    static_assert(!CV || BM == 128, "convolution mode: 128-row blocks");
    constexpr int BN = 128;
    constexpr int NI = BM / 32;                      // 16-row MFMA tiles per wave
    constexpr int kAKcT = BM * 16;                   // bytes per (plane, k-chunk) of the A image
    constexpr int kBKc = BN * 16;
    constexpr int kPl = F16 ? 2 : 3;                 // planes per operand
    constexpr int kAbytes = kPl * 4 * kAKcT;
    constexpr int kBstage = kPl * 4 * kBKc;
    constexpr int kPW = kBstage / 1024 / 4;          // 6 DMA pieces per wave and stage
    __shared__ __attribute__((aligned(16))) unsigned char lds[kAbytes + 2 * kBstage];
    const int per_xcd = (p.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * 256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int akc = tid & 3;
    const int arow0 = tid >> 2, arow1 = 64 + (tid >> 2);
    const int aslot0 = arow0 ^ (2 * akc), aslot1 = arow1 ^ (2 * akc);
    const float* ap0 = p.a + (int64_t)min(m0 + arow0, p.M - 1) * p.lda + akc * 8;
    const float* ap1 = p.a + (int64_t)min(m0 + arow1, p.M - 1) * p.lda + akc * 8;
    const bool second = BM == 128 || wave < 2;       // (wave-uniform) this thread stages a second row
    int cy0 = 0, cx0 = 0, cy1 = 0, cx1 = 0, mr0 = 0, mr1 = 0, vm = 3;
    if constexpr (CV) {
        mr0 = min(m0 + arow0, p.M - 1); mr1 = min(m0 + arow1, p.M - 1);
        cx0 = mr0 % p.cv_W; cy0 = (mr0 / p.cv_W) % p.cv_H;
        cx1 = mr1 % p.cv_W; cy1 = (mr1 / p.cv_W) % p.cv_H;
    }
    (void)cy0; (void)cx0; (void)cy1; (void)cx1; (void)mr0; (void)mr1; (void)vm;
    unsigned boff[kPW];
    int bpiece[kPW];
    {
        const int nl = lane >> 2, kc = (lane & 3) ^ ((0 - (nl >> 2)) & 3);
#pragma unroll
        for (int i = 0; i < kPW; ++i) {
            const int q = wave + 4 * i;
            const int pl = q / (BN / 16), nb = (q % (BN / 16)) * 16;
            boff[i] = (unsigned)(((int64_t)pl * p.plane + (int64_t)(n0 + nb + nl) * p.K + kc * 8) * 2);     // N % 256 == 0: in range
            bpiece[i] = __builtin_amdgcn_readfirstlane(q * 1024);
        }
    }
    const unsigned lds_b = (unsigned)(uintptr_t)(lds + kAbytes);
    const unsigned short* bp1 = p.bp + (int64_t)128 * p.K;         // the second column half

    // ONE register set for A, loaded one step ahead (a step is two passes long, i.e. as far ahead in time as the two-step
    // distance of the 128 x 128 kernel): accumulators 128 + A fragments 48 + B fragments 12 leave no room for a second set
    float4 ra[4];
#define G3_LA2(k0)                                                                           \
    if constexpr (CV) {                                                                      \
        const int tap_ = (k0) / p.cv_cin, kk_ = (k0) - tap_ * p.cv_cin;                      \
        const int dy_ = (tap_ / 3 - 1) * p.cv_sign, dx_ = (tap_ % 3 - 1) * p.cv_sign;        \
        const int sh_ = dy_ * p.cv_W + dx_;                                                  \
        const bool v0_ = (unsigned)(cy0 + dy_) < (unsigned)p.cv_H && (unsigned)(cx0 + dx_) < (unsigned)p.cv_W; \
        const bool v1_ = (unsigned)(cy1 + dy_) < (unsigned)p.cv_H && (unsigned)(cx1 + dx_) < (unsigned)p.cv_W; \
        const unsigned o0_ = (unsigned)(min(max(mr0 + sh_, 0), p.M - 1) * (int)p.lda + akc * 8 + kk_); \
        const unsigned o1_ = (unsigned)(min(max(mr1 + sh_, 0), p.M - 1) * (int)p.lda + akc * 8 + kk_); \
        ra[0] = *reinterpret_cast<const float4*>(p.a + o0_);                                 \
        ra[1] = *reinterpret_cast<const float4*>(p.a + o0_ + 4);                             \
        ra[2] = *reinterpret_cast<const float4*>(p.a + o1_);                                 \
        ra[3] = *reinterpret_cast<const float4*>(p.a + o1_ + 4);                             \
        vm = (v0_ ? 1 : 0) | (v1_ ? 2 : 0);                                                  \
    } else {                                                                                 \
        ra[0] = *reinterpret_cast<const float4*>(ap0 + (k0));                                \
        ra[1] = *reinterpret_cast<const float4*>(ap0 + (k0) + 4);                            \
        if (second) {                                                                        \
            ra[2] = *reinterpret_cast<const float4*>(ap1 + (k0));                            \
            ra[3] = *reinterpret_cast<const float4*>(ap1 + (k0) + 4);                        \
        }                                                                                    \
    }
#define G3_DMA2(base, k0, stage)                                                             \
    {                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < kPW; ++i_)                                   \
            glds16((base) + (k0), boff[i_], lds_b + (stage) * kBstage + bpiece[i_]);         \
    }
#define G3_WRITE2()                                                                          \
    {                                                                                        \
        uint4 h, m, l;                                                                       \
        float4 w0_ = ra[0], w1_ = ra[1], w2_ = ra[2], w3_ = ra[3];                           \
        if constexpr (CV) {                                                                  \
            const float s0_ = (vm & 1) ? 1.f : 0.f, s1_ = (vm & 2) ? 1.f : 0.f;              \
            w0_ = make_float4(w0_.x * s0_, w0_.y * s0_, w0_.z * s0_, w0_.w * s0_);           \
            w1_ = make_float4(w1_.x * s0_, w1_.y * s0_, w1_.z * s0_, w1_.w * s0_);           \
            w2_ = make_float4(w2_.x * s1_, w2_.y * s1_, w2_.z * s1_, w2_.w * s1_);           \
            w3_ = make_float4(w3_.x * s1_, w3_.y * s1_, w3_.z * s1_, w3_.w * s1_);           \
        }                                                                                    \
        if constexpr (F16) {                                                                 \
            split8h(w0_, w1_, sc_a, &h, &l);                                                 \
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKcT + aslot0 * 16) = h;        \
            *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKcT + aslot0 * 16) = l;        \
            if (second) {                                                                    \
                split8h(w2_, w3_, sc_a, &h, &l);                                             \
                *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKcT + aslot1 * 16) = h;    \
                *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKcT + aslot1 * 16) = l;    \
            }                                                                                \
        } else {                                                                             \
        split8(w0_, w1_, &h, &m, &l);                                                        \
        *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKcT + aslot0 * 16) = h;            \
        *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKcT + aslot0 * 16) = m;            \
        *reinterpret_cast<uint4*>(lds + (2 * 4 + akc) * kAKcT + aslot0 * 16) = l;            \
        if (second) {                                                                        \
            split8(w2_, w3_, &h, &m, &l);                                                    \
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKcT + aslot1 * 16) = h;        \
            *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKcT + aslot1 * 16) = m;        \
            *reinterpret_cast<uint4*>(lds + (2 * 4 + akc) * kAKcT + aslot1 * 16) = l;        \
        }                                                                                    \
        }                                                                                    \
    }

    float sc_a = 1.f, inv_a = 1.f, inv_b = 1.f;
    if constexpr (F16) {
        float sc_b;
        h2_scale(amax_read(p.a_amax), &sc_a, &inv_a);
        h2_scale(amax_read(p.b_amax), &sc_b, &inv_b);
    }
    (void)sc_a;
    Acc2<NI, kAKcT, F16> acc;
    acc.zero();
    const int a_frag = wr * (BM / 2) * 16;
    const int b_frag = kAbytes + wc * 64 * 64;
    const int nk = p.K / kBK;
    const int klast = (nk - 1) * kBK;
    G3_LA2(0);
    __builtin_amdgcn_sched_barrier(0);
    G3_DMA2(p.bp, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // Vector-memory operations complete in order.  Issue order of a step: [DMA half 1 of k] [A loads of k + 1] [DMA half 0 of
    // k + 1].  The compiler knows of the A loads only and waits for all of them (vmcnt(0)) before the split, which also
    // covers half 0 of the step (issued one pass earlier); the wait for half 1 is explicit.
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                                   // a: the A image and stage 1 are free
        G3_WRITE2();                                       // (waits for the A loads and, with them, for half 0 of this step)
        __builtin_amdgcn_sched_barrier(0);
        G3_DMA2(bp1, kt * kBK, 1);
        __syncthreads();                                   // b
        G3_LA2(min((kt + 1) * kBK, klast));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(G3_PRIO);
        acc.load_a(lds, a_frag, lane);
        acc.template pass<0>(lds, b_frag, lane);
        __builtin_amdgcn_s_setprio(0);
        // half 1 has landed: only this step's A loads are younger (4 of them, 2 in the waves that stage one row of a 96-row tile)
        if (second) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();                                   // c: stage 0 is free, half 1 visible to all
        G3_DMA2(p.bp, min((kt + 1) * kBK, klast), 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(G3_PRIO);
        acc.template pass<1>(lds, b_frag + kBstage, lane);
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef G3_LA2
#undef G3_DMA2
#undef G3_WRITE2
    float omax = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        struct { f32x4 v[NI][4]; } out;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) out.v[i][j] = acc.v[h][i][j];
        omax = fmaxf(omax, g3_epilogue<4, NI, decltype(out), F16>(p, out, lane, m0 + wr * (BM / 2), n0 + h * 128 + wc * 64, inv_a, inv_b));
    }
    if constexpr (F16) {
        if (p.out_amax) amax_commit(p.out_amax, omax, reinterpret_cast<float*>(lds));
    }
    (void)omax;
}

// A in bf16 (g3_tn_tile<.., ABF>): activations that ARE bf16 (the backbone's feature maps under autocast, the bf16
// gradient of mask_features) enter the fp32 GEMM without a cast pass and at three products per K step
template <int BN>
__global__ __launch_bounds__(kThreads, 2) void gemm3_tn_abf_kernel(G3 p)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[12 * kAKc + 2 * 12 * BN * 16];
    const int per_xcd = (p.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    g3_tn_tile<BN, false, false, true>(p, lds, tm * kBM, tn * BN);
}

// 3x3 convolution (stride 1, zero padding 1) of channel-last images as ONE GEMM with K = 9 * Cin (g3_tn_tile<.., CV>)
__global__ __launch_bounds__(kThreads, 2) void gemm3_conv_kernel(G3 p)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[12 * kAKc + 2 * 12 * 128 * 16];
    const int per_xcd = (p.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    g3_tn_tile<128, false, true>(p, lds, tm * kBM, tn * 128);
}

// Mixed launch against the tail effect: 128 x 128 tiles for as many row blocks as fill WHOLE rounds of the chip's
// 2 x 256 workgroup slots, the remaining row blocks as 128 x 64 tiles (twice as many, half as long).  M = 43 008,
// N = 256: 672 tiles = 1.31 rounds ran as 2; here 512 tiles + 320 half tiles = 1 + 0.5.  The half tiles come last in
// block order, so they start as the full tiles drain.
template <bool A2>
__global__ __launch_bounds__(kThreads, 2) void gemm3_tn_mixed_kernel(G3 p)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[12 * kAKc + 2 * 12 * 128 * 16];
    const int g1 = ((p.ntiles + 7) >> 3) << 3;          // blocks of the first region (multiple of 8)
    if ((int)blockIdx.x < g1) {
        const int per_xcd = g1 >> 3;
        const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
        if (tile >= p.ntiles) return;
        const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
        g3_tn_tile<128, A2>(p, lds, tm * kBM, tn * 128);
    } else {
        const int b = (int)blockIdx.x - g1;
        const int per_xcd = (p.ntiles2 + 7) >> 3;
        const int tile = (b & 7) * per_xcd + (b >> 3);
        if (tile >= p.ntiles2) return;
        const int tm = tile / p.tiles_n2, tn = tile - tm * p.tiles_n2;
        g3_tn_tile<64, A2>(p, lds, (p.tm0 + tm) * kBM, tn * 64);
    }
}

// ------------------------------------------------------------------------------------------------
// NT form (weight gradients): contraction over the ROWS of two row-major activations,
//   Cpart[s][m][n] = sum_{r in split s} A[r, m] * (B[r, n] + B2[r % b2_rows, n]),
// plus the column sums of A and/or B per split (bias gradients; per-level sums for level_embed).
// A thread stages a COLUMN piece (8 consecutive rows of one column: 8 coalesced dword loads), so the
// split pieces go to the same k-contiguous LDS image as in the TN kernel with one b128 write per
// plane, and the column sums are per-thread running sums.
// ------------------------------------------------------------------------------------------------
struct G3N {
    const float* a;
    const float* b;
    const float* b2;
    float* c;
    float* csum_a;
    float* csum_b;
    int64_t lda, ldb, ldb2;
    int R, Mdim, Ndim, b2_rows, rows_per_split, nsplit, tiles_m, tiles_n, ntiles, transpose_out;
    unsigned a_bytes, b_bytes;      // sizes of the operands (for the buffer descriptors; both < 4 GiB)
    int cv_H, cv_W, cv_cin;         // CV: weight gradient of a 3x3 convolution (B = the input image, columns = (tap, channel))
    int64_t c_ss, csa_ss, csb_ss;   // elements between the partial results of consecutive splits (c, csum_a, csum_b)
    const float* a_amax;            // H2 kernels: amax slots of the two operands
    const float* b_amax;
};
// a group of weight-gradient problems over the SAME rows in one launch (mpf_gemm3_nt_grouped): item i owns the tiles
// [tile_end[i-1], tile_end[i])
constexpr int kNtGroupMax = 8;
struct G3NG {
    G3N it[kNtGroupMax];
    int tile_end[kNtGroupMax];
    int n_items, ntiles;
};

// BF: the operands are bf16 matrices (p.a / p.b point to 2-byte elements): one plane, one product —
// the weight gradient of a bf16 Linear with many rows (the K / V projections of the cross-attention).
// CV (BN = 128, fp32): the weight gradient of a 3x3 / stride 1 / padding 1 convolution of channel-last images,
//   dW2[co][tap * Cin + ci] = sum_r dY[r][co] * X[r + dy(tap) * W + dx(tap)][ci]   (taps off the image contribute nothing):
// the column tile fixes the tap (Cin % 128 == 0), so the row shift of the B operand is a per-workgroup scalar, and with
// W % 8 == 0 the 8 rows of a k-chunk lie in one image row, so a row's validity is wave-uniform as well.
// A16 / B16 (BN = 128, fp32 result): that operand is a bf16 matrix (2-byte elements, its lda / ldb in elements): loaded with
// 2-byte reads into plane 0 only, and the products with its planes 1, 2 are skipped (three instead of six).
template <int BN, bool BF, bool CV, bool A16, bool B16, bool H2 = false>
__device__ __forceinline__ void gemm3_nt_tile(const G3N& p, const int tile)
{
    static_assert(!((A16 || B16) && (BF || CV || BN != 128)), "mixed-precision operands: plain 128-column tiles only");
    static_assert(!(H2 && (BF || A16 || B16)), "fp16 x 2 form: fp32 operands only");
    constexpr int NJ = BN / 32;
    constexpr int kBKc = BN * 16;
    constexpr int kPl = H2 ? 2 : 3;
    constexpr int kAbytes = kPl * 4 * kAKc;
    constexpr int kBunits = 4 * BN;                          // (k-chunk, column) units of the B tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[kAbytes + kPl * 4 * kBKc];
    float sc_a = 1.f, sc_b = 1.f, inv_a = 1.f, inv_b = 1.f;
    if constexpr (H2) {
        h2_scale(amax_read(p.a_amax), &sc_a, &inv_a);
        h2_scale(amax_read(p.b_amax), &sc_b, &inv_b);
    }
    (void)sc_a; (void)sc_b;

    // order: column tiles fastest, then row tiles, then splits (neighbouring blocks share the rows)
    const int tn = tile % p.tiles_n, tm = (tile / p.tiles_n) % p.tiles_m, sp = tile / (p.tiles_n * p.tiles_m);
    const int m0 = tm * kBM, n0 = tn * BN;
    const int r_begin = sp * p.rows_per_split, r_end = min(p.R, r_begin + p.rows_per_split);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;


    // staging maps: A units (kc, m) = (tid / 128 + {0, 2}, tid % 128); B units u = tid + {0, 256}.
    // The k-chunk of a thread is the same for its whole wave (kBM = 128 = two waves; BN = 128 likewise), so
    // the row part of every load address is SCALAR: loads become "scalar row pointer + per-lane column"
    // (global_load with an SGPR base), with no per-load 64-bit vector arithmetic.
    const int am = tid & 127, akc = __builtin_amdgcn_readfirstlane(tid >> 7);
    const bool a_col_ok = m0 + am < p.Mdim;
    const int acolx = min(m0 + am, p.Mdim - 1);
    int bn_[2], bkc_[2], bcolx[2];
    bool b_ok[2], b_col_ok[2];
    int b2row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + i * kThreads;
        b_ok[i] = u < kBunits;
        bn_[i] = u % BN;
        bkc_[i] = min(u / BN, 3);
        if (BN == 128) bkc_[i] = __builtin_amdgcn_readfirstlane(bkc_[i]);
        b_col_ok[i] = b_ok[i] && n0 + bn_[i] < p.Ndim;
        bcolx[i] = min(n0 + bn_[i], p.Ndim - 1);
        if constexpr (CV) bcolx[i] = (n0 % p.cv_cin) + bn_[i];          // channel of the input image
        b2row[i] = (!CV && p.b2) ? (r_begin + bkc_[i] * 8) % p.b2_rows : 0;
    }
    // CV: tap of this column tile and the image coordinates of the first row of the thread's two k-chunks
    int cv_dy = 0, cv_dx = 0, cv_sh = 0, cv_x[2] = {0, 0}, cv_y[2] = {0, 0};
    if constexpr (CV) {
        const int tap = n0 / p.cv_cin;
        cv_dy = tap / 3 - 1; cv_dx = tap % 3 - 1;
        cv_sh = cv_dy * p.cv_W + cv_dx;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rb = r_begin + bkc_[i] * 8;
            cv_x[i] = rb % p.cv_W;
            cv_y[i] = (rb / p.cv_W) % p.cv_H;
        }
    }
    // buffer descriptors: voffset = the lane's column (bytes), soffset = the row (bytes, scalar when the k-chunk
    // is wave-uniform)
    constexpr int ESA = (BF || A16) ? 2 : 4, ESB = (BF || B16) ? 2 : 4;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, p.b_bytes, 0x00020000);
    const int acolb = acolx * ESA, bcolb0 = bcolx[0] * ESB, bcolb1 = bcolx[1] * ESB;
    const int ldab = (int)p.lda * ESA, ldbb = (int)p.ldb * ESB;
    auto ld16 = [&](const __amdgpu_buffer_rsrc_t rs, int colb, int rowb, bool uniform) -> float {
        const int vo = uniform ? colb : colb + rowb, so = uniform ? rowb : 0;
        return __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, vo, so, 0) << 16);
    };
    auto ld32 = [&](const __amdgpu_buffer_rsrc_t rs, int colb, int rowb, bool uniform) -> float {
        const int vo = uniform ? colb : colb + rowb, so = uniform ? rowb : 0;
        return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
    };
    auto ld = [&](const __amdgpu_buffer_rsrc_t rs, int colb, int rowb, bool uniform) -> float {     // the B operand (and CV)
        return (BF || B16) ? ld16(rs, colb, rowb, uniform) : ld32(rs, colb, rowb, uniform);
    };
    auto lda_ = [&](const __amdgpu_buffer_rsrc_t rs, int colb, int rowb, bool uniform) -> float {  // the A operand
        return (BF || A16) ? ld16(rs, colb, rowb, uniform) : ld32(rs, colb, rowb, uniform);
    };

    float xa0[8], xa1[8], xb0[8], xb1[8];
    float csa = 0.f, csb0 = 0.f, csb1 = 0.f;
    const bool want_csa = p.csum_a && tn == 0, want_csb = p.csum_b && tm == 0;

// loads are UNCONDITIONAL; TAIL (the last, possibly partial, step of a split) clamps the row (never reads
// past the matrix) and masks the rows that are not this split's
#define G3N_LOAD(r0, TAIL)                                                                             \
    {                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
            const int ra0 = (r0) + akc * 8 + j, ra1 = ra0 + 16;                                        \
            const int rb0 = (r0) + bkc_[0] * 8 + j, rb1 = (r0) + bkc_[1] * 8 + j;                      \
            const int ca0 = TAIL ? min(ra0, r_end - 1) : ra0, ca1 = TAIL ? min(ra1, r_end - 1) : ra1;  \
            const int cb0 = TAIL ? min(rb0, r_end - 1) : rb0, cb1 = TAIL ? min(rb1, r_end - 1) : rb1;  \
            float va0 = lda_(ars, acolb, ca0 * ldab, true), va1 = lda_(ars, acolb, ca1 * ldab, true);  \
            float vb0, vb1;                                                                            \
            if constexpr (CV) {                                                                        \
                const int sb0 = min(max(cb0 + cv_sh, 0), p.R - 1), sb1 = min(max(cb1 + cv_sh, 0), p.R - 1); \
                vb0 = ld(brs, bcolb0, sb0 * ldbb, true); vb1 = ld(brs, bcolb1, sb1 * ldbb, true);      \
                const bool ok0 = (unsigned)(cv_y[0] + cv_dy) < (unsigned)p.cv_H && (unsigned)(cv_x[0] + j + cv_dx) < (unsigned)p.cv_W; \
                const bool ok1 = (unsigned)(cv_y[1] + cv_dy) < (unsigned)p.cv_H && (unsigned)(cv_x[1] + j + cv_dx) < (unsigned)p.cv_W; \
                vb0 = ok0 ? vb0 : 0.f; vb1 = ok1 ? vb1 : 0.f;                                          \
            } else {                                                                                   \
                vb0 = ld(brs, bcolb0, cb0 * ldbb, BN == 128); vb1 = ld(brs, bcolb1, cb1 * ldbb, BN == 128); \
            }                                                                                          \
            if (!BF && !CV && p.b2) {                                                                  \
                int q0 = b2row[0] + j, q1 = b2row[1] + j;                                              \
                q0 = q0 >= p.b2_rows ? q0 - p.b2_rows : q0;                                            \
                q1 = q1 >= p.b2_rows ? q1 - p.b2_rows : q1;                                            \
                vb0 += (p.b2 + (int64_t)q0 * p.ldb2)[bcolx[0]];                                        \
                vb1 += (p.b2 + (int64_t)q1 * p.ldb2)[bcolx[1]];                                        \
            }                                                                                          \
            if (TAIL) {                                                                                \
                va0 = ra0 < r_end ? va0 : 0.f;                                                         \
                va1 = ra1 < r_end ? va1 : 0.f;                                                         \
                vb0 = rb0 < r_end ? vb0 : 0.f;                                                         \
                vb1 = rb1 < r_end ? vb1 : 0.f;                                                         \
            }                                                                                          \
            xa0[j] = va0; xa1[j] = va1; xb0[j] = vb0;                                                  \
            xb1[j] = b_ok[1] ? vb1 : 0.f;                                                              \
        }                                                                                              \
        if (!BF && !CV && p.b2) {                                                                      \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
                b2row[i] += kBK;                                                                       \
                while (b2row[i] >= p.b2_rows) b2row[i] -= p.b2_rows;                                   \
            }                                                                                          \
        }                                                                                              \
        if constexpr (CV) {      /* the next K step is 32 rows further */                              \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
                cv_x[i] += kBK;                                                                        \
                while (cv_x[i] >= p.cv_W) { cv_x[i] -= p.cv_W; cv_y[i] = cv_y[i] + 1 == p.cv_H ? 0 : cv_y[i] + 1; } \
            }                                                                                          \
        }                                                                                              \
    }

    Acc<NJ, NJ == 4> acc;
    acc.zero();

    const int a_frag = wr * 64 * 16;
    const int b_frag = kAbytes + wc * (BN / 2) * 16;

    if (r_begin + kBK <= r_end) G3N_LOAD(r_begin, false) else G3N_LOAD(r_begin, true);
    for (int r0 = r_begin; r0 < r_end; r0 += kBK) {
        __syncthreads();
        if constexpr (H2) {
            uint4 h, l;
            split8h(make_float4(xa0[0], xa0[1], xa0[2], xa0[3]), make_float4(xa0[4], xa0[5], xa0[6], xa0[7]), sc_a, &h, &l);
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKc + (am ^ (2 * akc)) * 16) = h;
            *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKc + (am ^ (2 * akc)) * 16) = l;
            split8h(make_float4(xa1[0], xa1[1], xa1[2], xa1[3]), make_float4(xa1[4], xa1[5], xa1[6], xa1[7]), sc_a, &h, &l);
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc + 2) * kAKc + (am ^ (2 * akc + 4)) * 16) = h;
            *reinterpret_cast<uint4*>(lds + (1 * 4 + akc + 2) * kAKc + (am ^ (2 * akc + 4)) * 16) = l;
            split8h(make_float4(xb0[0], xb0[1], xb0[2], xb0[3]), make_float4(xb0[4], xb0[5], xb0[6], xb0[7]), sc_b, &h, &l);
            *reinterpret_cast<uint4*>(lds + kAbytes + (0 * 4 + bkc_[0]) * kBKc + (bn_[0] ^ (2 * bkc_[0])) * 16) = h;
            *reinterpret_cast<uint4*>(lds + kAbytes + (1 * 4 + bkc_[0]) * kBKc + (bn_[0] ^ (2 * bkc_[0])) * 16) = l;
            if (b_ok[1]) {
                split8h(make_float4(xb1[0], xb1[1], xb1[2], xb1[3]), make_float4(xb1[4], xb1[5], xb1[6], xb1[7]), sc_b, &h, &l);
                *reinterpret_cast<uint4*>(lds + kAbytes + (0 * 4 + bkc_[1]) * kBKc + (bn_[1] ^ (2 * bkc_[1])) * 16) = h;
                *reinterpret_cast<uint4*>(lds + kAbytes + (1 * 4 + bkc_[1]) * kBKc + (bn_[1] ^ (2 * bkc_[1])) * 16) = l;
            }
            if (want_csa) {
#pragma unroll
                for (int j = 0; j < 8; ++j) csa += xa0[j] + xa1[j];
            }
            if (want_csb) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { csb0 += xb0[j]; csb1 += xb1[j]; }
            }
        } else {
            uint4 h, m, l;
            split8(make_float4(xa0[0], xa0[1], xa0[2], xa0[3]), make_float4(xa0[4], xa0[5], xa0[6], xa0[7]), &h, &m, &l);
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKc + (am ^ (2 * akc)) * 16) = h;
            if (!BF && !A16) *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKc + (am ^ (2 * akc)) * 16) = m;
            if (!BF && !A16) *reinterpret_cast<uint4*>(lds + (2 * 4 + akc) * kAKc + (am ^ (2 * akc)) * 16) = l;
            split8(make_float4(xa1[0], xa1[1], xa1[2], xa1[3]), make_float4(xa1[4], xa1[5], xa1[6], xa1[7]), &h, &m, &l);
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc + 2) * kAKc + (am ^ (2 * akc + 4)) * 16) = h;
            if (!BF && !A16) *reinterpret_cast<uint4*>(lds + (1 * 4 + akc + 2) * kAKc + (am ^ (2 * akc + 4)) * 16) = m;
            if (!BF && !A16) *reinterpret_cast<uint4*>(lds + (2 * 4 + akc + 2) * kAKc + (am ^ (2 * akc + 4)) * 16) = l;
            split8(make_float4(xb0[0], xb0[1], xb0[2], xb0[3]), make_float4(xb0[4], xb0[5], xb0[6], xb0[7]), &h, &m, &l);
            *reinterpret_cast<uint4*>(lds + kAbytes + (0 * 4 + bkc_[0]) * kBKc + (bn_[0] ^ (2 * bkc_[0])) * 16) = h;
            if (!BF && !B16) *reinterpret_cast<uint4*>(lds + kAbytes + (1 * 4 + bkc_[0]) * kBKc + (bn_[0] ^ (2 * bkc_[0])) * 16) = m;
            if (!BF && !B16) *reinterpret_cast<uint4*>(lds + kAbytes + (2 * 4 + bkc_[0]) * kBKc + (bn_[0] ^ (2 * bkc_[0])) * 16) = l;
            if (b_ok[1]) {
                split8(make_float4(xb1[0], xb1[1], xb1[2], xb1[3]), make_float4(xb1[4], xb1[5], xb1[6], xb1[7]), &h, &m, &l);
                *reinterpret_cast<uint4*>(lds + kAbytes + (0 * 4 + bkc_[1]) * kBKc + (bn_[1] ^ (2 * bkc_[1])) * 16) = h;
                if (!BF && !B16) *reinterpret_cast<uint4*>(lds + kAbytes + (1 * 4 + bkc_[1]) * kBKc + (bn_[1] ^ (2 * bkc_[1])) * 16) = m;
                if (!BF && !B16) *reinterpret_cast<uint4*>(lds + kAbytes + (2 * 4 + bkc_[1]) * kBKc + (bn_[1] ^ (2 * bkc_[1])) * 16) = l;
            }
            if (want_csa) {
#pragma unroll
                for (int j = 0; j < 8; ++j) csa += xa0[j] + xa1[j];
            }
            if (want_csb) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { csb0 += xb0[j]; csb1 += xb1[j]; }
            }
        }
        __syncthreads();
        if (r0 + 2 * kBK <= r_end) G3N_LOAD(r0 + kBK, false) else if (r0 + kBK < r_end) G3N_LOAD(r0 + kBK, true);
        __builtin_amdgcn_s_setprio(G3_PRIO);        // the MFMA phase outranks the other workgroups' staging VALU on this SIMD
        if constexpr (H2) acc.template step_h2<kAKc, kBKc>(lds, a_frag, b_frag, lane);
        else if constexpr (A16 || B16) acc.template step<kAKc, kBKc, false, A16, B16>(lds, a_frag, b_frag, lane);
        else acc.template step<kAKc, kBKc, BF>(lds, a_frag, b_frag, lane);
        __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue -----------------------------------------------------------------------------------
    float* cp = p.c + (int64_t)sp * p.c_ss;
    acc.quads(lane, [&](int mo, int no, float4 o) {
        const int m = m0 + wr * 64 + mo, n = n0 + wc * (BN / 2) + no;
        if (m >= p.Mdim) return;
        if constexpr (H2) o = make_float4(o.x * inv_a * inv_b, o.y * inv_a * inv_b, o.z * inv_a * inv_b, o.w * inv_a * inv_b);
        const float e4[4] = {o.x, o.y, o.z, o.w};
        if (!p.transpose_out) {
            if (n + 3 < p.Ndim) {
                *reinterpret_cast<float4*>(cp + (int64_t)m * p.Ndim + n) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.Ndim) cp[(int64_t)m * p.Ndim + n + e] = e4[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < p.Ndim) cp[(int64_t)(n + e) * p.Mdim + m] = e4[e];
        }
    });
    // column sums: per-thread running sums -> LDS float atomics (once per block) -> per-split rows
    if (want_csa || want_csb) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
        for (int i = tid; i < kBM + BN; i += kThreads) red[i] = 0.f;
        __syncthreads();
        if (want_csa) atomicAdd(&red[am], csa);
        if (want_csb) {
            atomicAdd(&red[kBM + bn_[0]], csb0);
            if (b_ok[1]) atomicAdd(&red[kBM + bn_[1]], csb1);
        }
        __syncthreads();
        if (want_csa && tid < kBM && m0 + tid < p.Mdim) p.csum_a[(int64_t)sp * p.csa_ss + m0 + tid] = red[tid];
        if (want_csb && tid < BN && n0 + tid < p.Ndim) p.csum_b[(int64_t)sp * p.csb_ss + n0 + tid] = red[kBM + tid];
    }
    (void)a_col_ok; (void)b_col_ok;
}

template <int BN, bool BF = false, bool CV = false, bool A16 = false, bool B16 = false, bool H2 = false>
__global__ __launch_bounds__(kThreads, 2) void gemm3_nt_kernel(G3N p)
{
    const int per_xcd = (p.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    gemm3_nt_tile<BN, BF, CV, A16, B16, H2>(p, tile);
}

// several problems over the same rows: with the tiles of all of them in one launch a workgroup's split is n_items times
// longer at the same number of workgroups (one round of the chip), so the pipeline fill / drain and the partial results
// are paid once per (tile, long split) instead of once per (tile, short split)
template <bool H2>
__global__ __launch_bounds__(kThreads, 2) void gemm3_nt_group_kernel(G3NG g)
{
    const int per_xcd = (g.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= g.ntiles) return;
    int i = 0;
    while (i + 1 < g.n_items && tile >= g.tile_end[i]) ++i;
    const int first = i ? g.tile_end[i - 1] : 0;
    gemm3_nt_tile<128, false, false, false, false, H2>(g.it[i], tile - first);
}

#include "gemm3_nt2.h"

// W[R,C] fp32 -> planes[3][R][C] (transpose = 0) or planes[3][C][R] (transpose = 1), bf16 bits
__global__ __launch_bounds__(256) void gemm3_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                          int R, int C, int transpose)
{
    const int64_t total = (int64_t)R * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT plane (coalesced stores)
        int r, c;
        if (transpose) { c = (int)(i / R); r = (int)(i - (int64_t)c * R); } else { r = (int)(i / C); c = (int)(i - (int64_t)r * C); }
        const float x = w[(int64_t)r * C + c];
        const unsigned xb = __float_as_uint(x);
        const float r1 = x - __uint_as_float(xb & 0xFFFF0000u);
        const unsigned rb = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(rb & 0xFFFF0000u);
        const unsigned lb = __float_as_uint(r2) + 0x8000u;
        out[i] = (unsigned short)(xb >> 16);
        out[total + i] = (unsigned short)(rb >> 16);
        out[2 * total + i] = (unsigned short)(lb >> 16);
    }
}

// the same split for a LIST of weight matrices in one launch: item i owns workgroups [first_block,
// first_block + ceil(rows * cols / 1024)); its three planes are `plane_stride` elements apart and are
// written with leading dimension dst_ld (so two sources can fill row / column ranges of one operand)
__global__ __launch_bounds__(256) void gemm3_split_grouped_kernel(const MpfSplitItem* __restrict__ items, int n_items)
{
    int lo = 0, hi = n_items - 1;
    const int64_t blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const MpfSplitItem it = items[lo];
    const float* __restrict__ w = it.src;
    unsigned short* __restrict__ out = static_cast<unsigned short*>(it.dst);
    const int R = (int)it.rows, C = (int)it.cols;
    const int64_t total = (int64_t)R * C, base = (blk - it.first_block) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;       // index in the (possibly transposed) output matrix
        if (i >= total) continue;
        int r, c;
        int64_t o;
        if (it.transpose) { c = (int)(i / R); r = (int)(i - (int64_t)c * R); o = (int64_t)c * it.dst_ld + r; }
        else { r = (int)(i / C); c = (int)(i - (int64_t)r * C); o = (int64_t)r * it.dst_ld + c; }
        const float x = w[(int64_t)r * C + c];
        const unsigned xb = __float_as_uint(x);
        const float r1 = x - __uint_as_float(xb & 0xFFFF0000u);
        const unsigned rb = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(rb & 0xFFFF0000u);
        const unsigned lb = __float_as_uint(r2) + 0x8000u;
        out[o] = (unsigned short)(xb >> 16);
        out[it.plane_stride + o] = (unsigned short)(rb >> 16);
        out[2 * it.plane_stride + o] = (unsigned short)(lb >> 16);
    }
}

// ---- fp16 x 2 form: largest magnitudes and the weight planes -------------------------------------------------------------
// items[i] (device table): x_i[0 .. n_i) -> atomic max of |x| into *out_i (several items may share a slot: one operand stacked
// from two weights); item i owns workgroups [first_block, first_block + ceil(n_i / 4096))
__global__ __launch_bounds__(256) void amax_grouped_kernel(const MpfAmaxItem* __restrict__ items, int n_items)
{
    int lo = 0, hi = n_items - 1;
    const int64_t blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const MpfAmaxItem it = items[lo];
    const int64_t base = (blk - it.first_block) * 4096;
    float m = 0.f;
    if (((uintptr_t)it.src & 15) == 0 && base + 4096 <= it.numel) {
        const float4* __restrict__ s4 = reinterpret_cast<const float4*>(it.src + base);
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = s4[k * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
    } else {
        for (int64_t i = base + threadIdx.x; i < min(base + 4096, it.numel); i += 256) m = fmaxf(m, fabsf(it.src[i]));
    }
    __shared__ float red[4];
    amax_commit(it.out, m, red);
}

// one tensor, grid-stride (activations: tens of MB)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n4, int64_t n, float* __restrict__ out)
{
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    float m = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = x4[i + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
    }
    for (; i < n4; i += stride) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int64_t j = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) m = fmaxf(m, fabsf(x[j]));
    __shared__ float red[4];
    amax_commit(out, m, red);
}

// the split of gemm3_split_grouped_kernel into TWO fp16 planes of (scale x w), scale from *amax (h2_scale)
__global__ __launch_bounds__(256) void gemm3_split_grouped_h2_kernel(const MpfSplitItemH2* __restrict__ items, int n_items)
{
    int lo = 0, hi = n_items - 1;
    const int64_t blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const MpfSplitItemH2 it = items[lo];
    const float* __restrict__ w = it.src;
    _Float16* __restrict__ out = static_cast<_Float16*>(it.dst);
    float scale, inv;
    h2_scale(amax_read(it.amax), &scale, &inv);
    const int R = (int)it.rows, C = (int)it.cols;
    const int64_t total = (int64_t)R * C, base = (blk - it.first_block) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;       // index in the (possibly transposed) output matrix
        if (i >= total) continue;
        int r, c;
        int64_t o;
        if (it.transpose) { c = (int)(i / R); r = (int)(i - (int64_t)c * R); o = (int64_t)c * it.dst_ld + r; }
        else { r = (int)(i / C); c = (int)(i - (int64_t)r * C); o = (int64_t)r * it.dst_ld + c; }
        const float x = w[(int64_t)r * C + c] * scale;
        const _Float16 h = (_Float16)x;
        out[o] = h;
        out[it.plane_stride + o] = (_Float16)(x - (float)h);
    }
}

}  // namespace

extern "C" int mpf_amax_f32(const float* x, int64_t n, float* amax, void* stream)
{
    if (!x || !amax) return mpf::fail(MPF_E_NULL, "amax_f32: NULL buffer");
    if (n <= 0) return 0;
    if ((uintptr_t)x & 15) return mpf::fail(MPF_E_SHAPE, "amax_f32: x must be 16-byte aligned");
    const int64_t n4 = n / 4;
    const int blocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n4 + 1023) / 1024));
    mpf::prof_begin((hipStream_t)stream);
    mpf::set_kernel("amax_kernel");
    hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n4, n, amax);
    mpf::prof_end("amax_kernel", (hipStream_t)stream, 4.0 * (double)n);
    return mpf::check(hipGetLastError(), "mpf_amax_f32");
}

// Run-time range guard of the fp16 x 2 form.  The split keeps 22 bits of an element while its second piece
// is a normal fp16 number, i.e. for elements within 2^-18 of the operand's amax slot; a ROW whose largest magnitude lies below
// that loses one bit per further binade.  This pass counts, for one operand [rows, cols] and the slot its consumer scales by,
// the non-zero rows and those whose largest magnitude is below 2^-log2_below of the slot: counters[0] += non-zero rows,
// counters[1] += rows below.  One wave per row (cols % 4 == 0), one pair of atomics per workgroup.  mp_former_amd/encoder_fused.py
// runs it on a sampled step over every operand of the encoder's GEMMs and warns when a share exceeds 0.1 %.
__global__ __launch_bounds__(256) void h2_range_stats_kernel(const float* __restrict__ a, int rows, int cols, int64_t lda,
                                                             const float* __restrict__ amax_slot, int log2_below,
                                                             unsigned long long* __restrict__ counters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float ref = __uint_as_float(amax_read(amax_slot));
    const float thr = ldexpf(ref, -log2_below);
    unsigned nz = 0, below = 0;
    for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
        const float4* __restrict__ row = reinterpret_cast<const float4*>(a + (int64_t)r * lda);
        float m = 0.f;
        for (int c = lane; c < cols / 4; c += 64) {
            const float4 v = row[c];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        nz += m > 0.f;
        below += (m > 0.f && m < thr);
    }
    __shared__ unsigned red[8];
    if (lane == 0) { red[wave * 2] = nz; red[wave * 2 + 1] = below; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned v = red[threadIdx.x] + red[2 + threadIdx.x] + red[4 + threadIdx.x] + red[6 + threadIdx.x];
        if (v) atomicAdd(counters + threadIdx.x, (unsigned long long)v);
    }
}

extern "C" int mpf_h2_range_stats(const float* a, int rows, int cols, int64_t lda, const float* amax_slot, int log2_below,
                                  unsigned long long* counters_device, void* stream)
{
    if (!a || !amax_slot || !counters_device) return mpf::fail(MPF_E_NULL, "h2_range_stats: NULL buffer");
    if (rows <= 0 || cols <= 0 || cols % 4 || lda < cols || lda % 4 || ((uintptr_t)a & 15) || log2_below < 0 || log2_below > 120)
        return mpf::fail(MPF_E_SHAPE, "h2_range_stats: cols and lda must be multiples of 4, a 16-byte aligned, 0 <= log2_below <= 120");
    const int blocks = std::min(2048, (rows + 3) / 4);
    mpf::set_kernel("h2_range_stats_kernel");
    hipLaunchKernelGGL(h2_range_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, rows, cols, lda, amax_slot, log2_below,
                       counters_device);
    return mpf::check(hipGetLastError(), "mpf_h2_range_stats");
}

extern "C" int mpf_amax_f32_grouped(const MpfAmaxItem* items_device, int n_items, int64_t total_blocks, void* stream)
{
    if (n_items == 0 || total_blocks == 0) return 0;
    if (!items_device) return mpf::fail(MPF_E_NULL, "amax_f32_grouped: NULL table");
    if (n_items < 0 || total_blocks < 0 || total_blocks > 0x7fffffffLL) return mpf::fail(MPF_E_SHAPE, "amax_f32_grouped: bad sizes");
    mpf::set_kernel("amax_grouped_kernel");
    hipLaunchKernelGGL(amax_grouped_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_device, n_items);
    return mpf::check(hipGetLastError(), "mpf_amax_f32_grouped");
}

extern "C" int mpf_gemm3_split_grouped_h2(const MpfSplitItemH2* items_device, int n_items, int64_t total_blocks, void* stream)
{
    if (n_items == 0 || total_blocks == 0) return 0;
    if (!items_device) return mpf::fail(MPF_E_NULL, "gemm3_split_grouped_h2: NULL table");
    if (n_items < 0 || total_blocks < 0 || total_blocks > 0x7fffffffLL) return mpf::fail(MPF_E_SHAPE, "gemm3_split_grouped_h2: bad sizes");
    mpf::set_kernel("gemm3_split_grouped_h2_kernel");
    hipLaunchKernelGGL(gemm3_split_grouped_h2_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_device,
                       n_items);
    return mpf::check(hipGetLastError(), "mpf_gemm3_split_grouped_h2");
}

int mpf::set_gemm3_option(const char* key, int v)
{
    if (!strcmp(key, "gemm3_mixed_tiles")) { g_mixed = v; return 0; }
    if (!strcmp(key, "gemm3_two_pass")) { g_two_pass = v; return 0; }
    if (!strcmp(key, "gemm3_two_pass_rows")) { g_two_pass_rows = v; return 0; }
    if (!strcmp(key, "gemm3_ws")) { g_ws = v; return 0; }
    if (!strcmp(key, "gemm3_tn3")) { g_tn3 = v; return 0; }
    if (!strcmp(key, "gemm3_tn3_176")) { g_tn3_176 = v; return 0; }
    if (!strcmp(key, "gemm3_nt2")) { g_nt2 = v; return 0; }
    if (strcmp(key, "gemm3_ablate") != 0) return 1;
    g_ablate = v;
    return 0;
}

extern "C" int mpf_gemm3_split(const float* w, int rows, int cols, int transpose, void* planes, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!w || !planes) return mpf::fail(MPF_E_NULL, "gemm3_split: NULL buffer");
    if (rows <= 0 || cols <= 0) return mpf::fail(MPF_E_SHAPE, "gemm3_split: bad sizes");
    const int64_t total = (int64_t)rows * cols;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    mpf::set_kernel("gemm3_split_kernel");
    hipLaunchKernelGGL(gemm3_split_kernel, dim3(blocks), dim3(256), 0, st, w, (unsigned short*)planes, rows, cols, transpose);
    return mpf::check(hipGetLastError(), "mpf_gemm3_split");
}

extern "C" int mpf_gemm3_split_grouped(const MpfSplitItem* items_device, int n_items, int64_t total_blocks, void* stream)
{
    if (n_items == 0 || total_blocks == 0) return 0;
    if (!items_device) return mpf::fail(MPF_E_NULL, "gemm3_split_grouped: NULL table");
    if (n_items < 0 || total_blocks < 0 || total_blocks > 0x7fffffffLL) return mpf::fail(MPF_E_SHAPE, "gemm3_split_grouped: bad sizes");
    mpf::set_kernel("gemm3_split_grouped_kernel");
    hipLaunchKernelGGL(gemm3_split_grouped_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_device,
                       n_items);
    return mpf::check(hipGetLastError(), "mpf_gemm3_split_grouped");
}

// 128 x 256 / 96 x 256 tiles (gemm3_tn2_kernel: two 128-column passes over one A image per K step) for fp32 A without the
// addend rows and N a multiple of 256; row blocks of 96 when they need fewer (rounds x rows) on the chip's workgroup slots
// (M = 43 008, N = 256: 448 tiles instead of 336 for 512 slots).  Starts the launch-log record itself; false = not taken.
static bool g3_launch_two_pass(G3& p, hipStream_t st)
{
    if (g_two_pass <= 0 || p.N % 256 != 0 || p.N < g_two_pass) return false;
    static int slots2 = 0;
    if (!slots2) {
        int dev = 0, cus = 0;
        slots2 = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? 2 * cus : 512;
    }
    p.tiles_n = p.N / 256;
    const int t128 = ((p.M + 127) / 128) * p.tiles_n, t96 = ((p.M + 95) / 96) * p.tiles_n;
    const double c128 = (double)((t128 + slots2 - 1) / slots2) * 128.0, c96 = (double)((t96 + slots2 - 1) / slots2) * 96.0;
    const bool use96r = g_two_pass_rows == 96 || (g_two_pass_rows == 0 && c96 < c128);
    p.ntiles = use96r ? t96 : t128;
    p.tm0 = 0; p.ntiles2 = 0; p.tiles_n2 = 0;
    mpf::prof_begin(st);
    mpf::set_kernel(use96r ? "gemm3_tn_kernel<96x256>" : "gemm3_tn_kernel<128x256>");
    if (p.a_amax) {
        mpf::set_kernel(use96r ? "gemm3_tn_kernel<h2 96x256>" : "gemm3_tn_kernel<h2 128x256>");
        if (use96r) hipLaunchKernelGGL((gemm3_tn2_kernel<96, false, true>), dim3(((p.ntiles + 7) / 8) * 8), dim3(kThreads), 0, st, p);
        else hipLaunchKernelGGL((gemm3_tn2_kernel<128, false, true>), dim3(((p.ntiles + 7) / 8) * 8), dim3(kThreads), 0, st, p);
        return true;
    }
    if (use96r) hipLaunchKernelGGL(gemm3_tn2_kernel<96>, dim3(((p.ntiles + 7) / 8) * 8), dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL(gemm3_tn2_kernel<128>, dim3(((p.ntiles + 7) / 8) * 8), dim3(kThreads), 0, st, p);
    return true;
}

static int g3_tn_impl(const float* a, int64_t lda, const float* a2, int a2_rows, const void* b_planes,
                      const float* bias, const float* c_in, int64_t ldcin, const float* c_in2, int64_t ldcin2,
                      const float* gate, int64_t ldgate, float* c, int64_t ldc, int M, int N, int K,
                      int relu, void* stream, const float* a_amax, const float* b_amax, float* out_amax,
                      const unsigned char* gbits = nullptr, int64_t ldgbits = 0, unsigned char* gbits_out = nullptr,
                      int64_t ldgbits_out = 0);

// weight-stationary kernel (gemm3_ws.h): one persistent workgroup per CU; the N / 256 column groups of a row range on one XCD
static int g3_launch_ws(G3& p, hipStream_t st)
{
    const int ncg = p.N / 256;
    const int cus = mpf::cu_count();
    int per_xcd = cus / 8 / ncg * ncg;                 // workgroups per XCD: a multiple of the column groups
    if (per_xcd < ncg) per_xcd = ncg;
    int workers = per_xcd / ncg * 8;                   // row workers
    const int tiles16 = (p.M + 15) / 16;
    if (workers > tiles16) workers = ((tiles16 + 7) / 8) * 8;
    const int rpw = ((p.M + workers - 1) / workers + 15) / 16 * 16;
    const int grid = workers * ncg;                    // block b: xcd = b & 7, slot = b >> 3: column group slot % ncg, row worker (slot / ncg) * 8 + xcd
    // instantiated epilogues (cin, gate bits, relu, bits out): plain | + addend | ReLU | ReLU + addend | ReLU + mask out | bit-mask gate (+ addend)
    const int key = (p.cin_cm ? 8 : 0) | (p.gbits_cm ? 4 : 0) | (p.relu ? 2 : 0) | (p.gbits_out ? 1 : 0);
    const void* fn = nullptr;
    int slot = 0;
#define WS_CASE(K_, I_, E_, G_, R_, B_) case K_: fn = (const void*)gemm3_ws_kernel<E_, G_, R_, B_>; slot = I_; break;
    switch (key) {
        WS_CASE(0, 0, 0, false, false, false) WS_CASE(8, 1, 1, false, false, false) WS_CASE(2, 2, 0, false, true, false)
        WS_CASE(10, 3, 1, false, true, false) WS_CASE(3, 4, 0, false, true, true) WS_CASE(4, 5, 0, true, false, false)
        WS_CASE(12, 6, 1, true, false, false)
        default: return -1000;          // not instantiated: the tiled kernel
    }
#undef WS_CASE
    static mpf::LdsAttr attr[7];
    if (int e = mpf::ensure_dynamic_lds(fn, kWsLds, attr[slot])) return e;
    mpf::prof_begin(st);
    mpf::set_kernel("gemm3_tn_kernel<h2 ws>");
    {
        void* args[] = {(void*)&p, (void*)&rpw, (void*)&ncg};
        if (hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(kWsThreads), args, kWsLds, st); e != hipSuccess) return mpf::check(e, "gemm3_ws_kernel");
    }
    mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)p.M * p.K + (double)p.M * p.N) + 4.0 * (double)p.N * p.K, 2.0 * p.M * (double)p.N * p.K);
    return mpf::check(hipGetLastError(), "mpf_gemm3_tn_h2(ws)");
}

extern "C" int mpf_gemm3_tn(const float* a, int64_t lda, const float* a2, int a2_rows, const void* b_planes,
                            const float* bias, const float* c_in, int64_t ldcin, const float* c_in2, int64_t ldcin2,
                            const float* gate, int64_t ldgate, float* c, int64_t ldc, int M, int N, int K,
                            int relu, void* stream)
{
    return g3_tn_impl(a, lda, a2, a2_rows, b_planes, bias, c_in, ldcin, c_in2, ldcin2, gate, ldgate, c, ldc, M, N, K, relu, stream,
                      nullptr, nullptr, nullptr);
}

extern "C" int mpf_gemm3_tn_h2(const float* a, int64_t lda, const float* a_amax, const void* b_planes_h2, const float* b_amax,
                               const float* bias, const float* c_in, int64_t ldcin, const float* c_in2, int64_t ldcin2,
                               const float* gate, int64_t ldgate, float* c, int64_t ldc, float* out_amax, int M, int N, int K,
                               int relu, void* stream)
{
    if (!a_amax || !b_amax) return mpf::fail(MPF_E_NULL, "gemm3_tn_h2: NULL amax");
    return g3_tn_impl(a, lda, nullptr, 0, b_planes_h2, bias, c_in, ldcin, c_in2, ldcin2, gate, ldgate, c, ldc, M, N, K, relu, stream,
                      a_amax, b_amax, out_amax);
}

extern "C" int mpf_gemm3_tn_h2_bits(const float* a, int64_t lda, const float* a_amax, const void* b_planes_h2, const float* b_amax,
                                    const float* bias, const float* c_in, int64_t ldcin, const float* c_in2, int64_t ldcin2,
                                    const unsigned char* gate_bits, int64_t ldgbits, float* c, int64_t ldc, float* out_amax,
                                    unsigned char* gate_bits_out, int64_t ldgbits_out, int M, int N, int K, int relu, void* stream)
{
    if (!a_amax || !b_amax) return mpf::fail(MPF_E_NULL, "gemm3_tn_h2_bits: NULL amax");
    if (N % 128 != 0 || (gate_bits && (ldgbits < N / 8 || ldgbits % 8 != 0 || ((uintptr_t)gate_bits & 7))) ||
        (gate_bits_out && (ldgbits_out < N / 8 || ldgbits_out % 8 != 0 || ((uintptr_t)gate_bits_out & 7))))
        return mpf::fail(MPF_E_SHAPE, "gemm3_tn_h2_bits: N must be a multiple of 128; mask rows of >= N / 8 bytes, 8-byte aligned");
    return g3_tn_impl(a, lda, nullptr, 0, b_planes_h2, bias, c_in, ldcin, c_in2, ldcin2, nullptr, 0, c, ldc, M, N, K, relu, stream,
                      a_amax, b_amax, out_amax, gate_bits, ldgbits, gate_bits_out, ldgbits_out);
}

static int g3_tn_impl(const float* a, int64_t lda, const float* a2, int a2_rows, const void* b_planes,
                      const float* bias, const float* c_in, int64_t ldcin, const float* c_in2, int64_t ldcin2,
                      const float* gate, int64_t ldgate, float* c, int64_t ldc, int M, int N, int K,
                      int relu, void* stream, const float* a_amax, const float* b_amax, float* out_amax,
                      const unsigned char* gbits, int64_t ldgbits, unsigned char* gbits_out, int64_t ldgbits_out)
{
    hipStream_t st = (hipStream_t)stream;
    if (!a || !b_planes || !c) return mpf::fail(MPF_E_NULL, "gemm3_tn: NULL buffer");
    if (M <= 0 || N <= 0 || K <= 0) return mpf::fail(MPF_E_SHAPE, "gemm3_tn: bad sizes");
    if (K % kBK != 0 || N % 4 != 0 || lda % 4 != 0 || ldc % 4 != 0 || (c_in && ldcin % 4 != 0) || (c_in2 && ldcin2 % 4 != 0) ||
        (gate && ldgate % 4 != 0))
        return mpf::fail(MPF_E_SHAPE, "gemm3_tn: K must be a multiple of 32; N, lda, ldc multiples of 4");
    if (a2 && a2_rows <= 0) return mpf::fail(MPF_E_SHAPE, "gemm3_tn: a2_rows must be positive");
    G3 p;
    p.a_amax = nullptr; p.b_amax = nullptr; p.out_amax = nullptr;
    p.a = a; p.a2 = a2; p.bp = (const unsigned short*)b_planes; p.bias = bias; p.cin = c_in; p.c = c;
    p.cin2 = c_in2; p.gate = gate; p.ldcin2 = ldcin2; p.ldgate = ldgate;
    p.lda = lda; p.ldc = ldc; p.ldcin = ldcin; p.plane = (int64_t)N * K;
    p.M = M; p.N = N; p.K = K; p.a2_rows = a2_rows; p.relu = relu; p.c16 = nullptr;
    p.a_amax = a_amax; p.b_amax = b_amax; p.out_amax = out_amax;
    {
        const float* consts = nullptr;
        if (int rc = g3_consts(&consts, "gemm3_tn: constants")) return rc;
        p.bias_cm = bias ? 1 : 0; p.cin_cm = c_in ? 1 : 0; p.cin2_cm = c_in2 ? 1 : 0; p.gate_cm = gate ? 1 : 0;
        if (!bias) p.bias = consts;
        if (!c_in) { p.cin = consts; p.ldcin = 0; }
        if (!c_in2) { p.cin2 = consts; p.ldcin2 = 0; }
        if (!gate) { p.gate = consts + 4; p.ldgate = 0; }
        g3_no_bits(p, consts);
        if (gbits) { p.gbits = gbits; p.ldgbits = ldgbits; p.gbits_cm = 1; }
        p.gbits_out = gbits_out; p.ldgbits_out = ldgbits_out;
    }
    const int tiles_m = (M + kBM - 1) / kBM;
    // 96-wide column tiles when they waste fewer columns (e.g. N = 288 = 3 x 96)
    const int waste128 = ((N + 127) / 128) * 128 - N, waste96 = ((N + 95) / 96) * 96 - N;
    const bool use96 = waste96 < waste128;
    p.tiles_n = use96 ? (N + 95) / 96 : (N + 127) / 128;
    p.ntiles = tiles_m * p.tiles_n;
    p.tm0 = 0; p.ntiles2 = 0; p.tiles_n2 = 0;
    // (N >= g_ws: at N = 256 the 64 MB of weight fragments the workgroups fetch once per launch cost what the tiled kernel's re-reads do)
    if (a_amax && !a2 && !gate && !c_in2 && g_ws > 0 && N >= g_ws && K == kWsK && N % 256 == 0 && ((uintptr_t)a & 15) == 0) {
        const int r = g3_launch_ws(p, st);
        if (r != -1000) return r;
    }
    // 192 x 256 tiles, ONE workgroup per CU: only where they fill the chip — the last (or only) round of tiles must cover at least
    // 3/4 of the CUs (43 008 rows: 224 tiles on 256 CUs; 16 800 rows would be 88 tiles against 175 half-size tiles of the
    // two-pass kernel on two workgroup slots per CU: config D's head measured 11.62 -> 11.79 ms per step with them)
    bool tn3 = a_amax && !a2 && g_tn3 && N % 256 == 0 && M >= 2048 && ((uintptr_t)a & 15) == 0 && lda % 4 == 0;
    int t3_bm = kT3BM;
    if (tn3) {
        const int cus = mpf::cu_count(), nt3 = ((M + kT3BM - 1) / kT3BM) * (N / 256), rem = nt3 % cus;
        tn3 = g_tn3 == 2 || (nt3 >= cus * 3 / 4 && (rem == 0 || 4 * rem >= 3 * cus || nt3 >= 4 * cus));
        // 176-row tiles (96 + 80: the second row half one MFMA row tile shorter) when they need no more rounds of the chip than
        // 192-row tiles do: a workgroup's time goes with its rows, and 43 008 rows are 245 tiles of 176 on 256 CUs (224 of 192)
        const int nt176 = ((M + 175) / 176) * (N / 256);
        if (g_tn3_176 && (nt176 + cus - 1) / cus <= (nt3 + cus - 1) / cus) t3_bm = 176;
    }
    if (tn3) {
        static mpf::LdsAttr attr;
        if (int e = mpf::ensure_dynamic_lds((const void*)gemm3_tn3_kernel, kT3Lds, attr)) return e;
        p.tiles_n = N / 256;
        p.t3_bm = t3_bm;
        p.ntiles = ((M + t3_bm - 1) / t3_bm) * p.tiles_n;
        mpf::prof_begin(st);
        mpf::set_kernel(t3_bm == kT3BM ? "gemm3_tn_kernel<h2 192x256>" : "gemm3_tn_kernel<h2 192x256:176>");
        {
            void* args[] = {(void*)&p};
            if (hipError_t e = hipLaunchKernel((const void*)gemm3_tn3_kernel, dim3(((p.ntiles + 7) / 8) * 8), dim3(kT3T), args, kT3Lds, st); e != hipSuccess)
                return mpf::check(e, "gemm3_tn3_kernel");
        }
        mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)M * K + (double)M * N) + 4.0 * (double)N * K, 2.0 * M * (double)N * K);
        return mpf::check(hipGetLastError(), "mpf_gemm3_tn_h2(192x256)");
    }
    if (!a2 && g3_launch_two_pass(p, st)) {
        mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)M * K + (double)M * N) + (a_amax ? 4.0 : 6.0) * (double)N * K, 2.0 * M * (double)N * K);
        return mpf::check(hipGetLastError(), "mpf_gemm3_tn");
    }
    if (a_amax) {       // N % 256 != 0: the one-pass 128- / 96-column tiles
        const int grid_h = ((p.ntiles + 7) / 8) * 8;
        mpf::prof_begin(st);
        if (use96) {
            mpf::set_kernel("gemm3_tn_kernel<h2 96>");
            hipLaunchKernelGGL((gemm3_tn_kernel<96, false, true>), dim3(grid_h), dim3(kThreads), 0, st, p);
        } else {
            mpf::set_kernel("gemm3_tn_kernel<h2 128>");
            hipLaunchKernelGGL((gemm3_tn_kernel<128, false, true>), dim3(grid_h), dim3(kThreads), 0, st, p);
        }
        mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)M * K + (double)M * N) + 4.0 * (double)N * K, 2.0 * M * (double)N * K);
        return mpf::check(hipGetLastError(), "mpf_gemm3_tn_h2");
    }
    // tail effect: when the last round of 128 x 128 tiles would fill at most half of the chip's workgroup slots, its row
    // blocks are cut into 128 x 64 tiles instead (gemm3_tn_mixed_kernel)
    if (!use96 && N % 64 == 0 && g_mixed) {
        static int slots = 0;
        if (!slots) {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
                slots = 2 * cus;
            else
                slots = 512;
        }
        const int rounds = p.ntiles / slots, rem = p.ntiles - rounds * slots;
        if (rounds >= 1 && rem > 0 && 2 * rem <= slots && slots % p.tiles_n == 0) {
            const int main_blocks = rounds * (slots / p.tiles_n);
            p.tm0 = main_blocks;
            p.tiles_n2 = N / 64;
            p.ntiles2 = (tiles_m - main_blocks) * p.tiles_n2;
            p.ntiles = main_blocks * p.tiles_n;
            const int grid_mixed = ((p.ntiles + 7) / 8) * 8 + ((p.ntiles2 + 7) / 8) * 8;
            mpf::prof_begin(st);
            mpf::set_kernel("gemm3_tn_kernel<128+64>");
            if (a2) hipLaunchKernelGGL(gemm3_tn_mixed_kernel<true>, dim3(grid_mixed), dim3(kThreads), 0, st, p);
            else hipLaunchKernelGGL(gemm3_tn_mixed_kernel<false>, dim3(grid_mixed), dim3(kThreads), 0, st, p);
            mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)M * K + (double)M * N) + 6.0 * (double)N * K, 2.0 * M * (double)N * K);
            return mpf::check(hipGetLastError(), "mpf_gemm3_tn");
        }
    }
    const int grid = ((p.ntiles + 7) / 8) * 8;
    mpf::prof_begin(st);
    if (use96) {
        mpf::set_kernel("gemm3_tn_kernel<96>");
        if (a2) hipLaunchKernelGGL((gemm3_tn_kernel<96, true>), dim3(grid), dim3(kThreads), 0, st, p);
        else hipLaunchKernelGGL((gemm3_tn_kernel<96, false>), dim3(grid), dim3(kThreads), 0, st, p);
    } else {
        mpf::set_kernel("gemm3_tn_kernel<128>");
        if (a2) hipLaunchKernelGGL((gemm3_tn_kernel<128, true>), dim3(grid), dim3(kThreads), 0, st, p);
        else hipLaunchKernelGGL((gemm3_tn_kernel<128, false>), dim3(grid), dim3(kThreads), 0, st, p);
    }
    mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)M * K + (double)M * N) + 6.0 * (double)N * K, 2.0 * M * (double)N * K);
    return mpf::check(hipGetLastError(), "mpf_gemm3_tn");
}

extern "C" int mpf_gemm3_tn_ex(const void* a, int a_dtype, int64_t lda, const void* b_planes, const float* bias, const float* c_in,
                               int64_t ldcin, void* c, int c_dtype, int64_t ldc, int M, int N, int K, int relu, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!a || !b_planes || !c) return mpf::fail(MPF_E_NULL, "gemm3_tn_ex: NULL buffer");
    if ((a_dtype != MPF_F32 && a_dtype != MPF_BF16) || (c_dtype != MPF_F32 && c_dtype != MPF_BF16))
        return mpf::fail(MPF_E_DTYPE, "gemm3_tn_ex: a / c must be MPF_F32 or MPF_BF16");
    const int aal = a_dtype == MPF_BF16 ? 8 : 4;
    if (M <= 0 || N <= 0 || K <= 0 || K % kBK != 0 || N % 4 != 0 || lda % aal != 0 || ldc % 4 != 0 || (c_in && ldcin % 4 != 0) ||
        ((uintptr_t)a & 15))
        return mpf::fail(MPF_E_SHAPE, "gemm3_tn_ex: K must be a multiple of 32, N / ldc of 4, 16-byte aligned A rows");
    if (a_dtype == MPF_F32 && c_dtype == MPF_F32)
        return mpf_gemm3_tn((const float*)a, lda, nullptr, 0, b_planes, bias, c_in, ldcin, nullptr, 0, nullptr, 0, (float*)c, ldc, M, N, K, relu, stream);
    const float* consts = nullptr;
    if (int rc = g3_consts(&consts, "gemm3_tn_ex: constants")) return rc;
    G3 p;
    p.a_amax = nullptr; p.b_amax = nullptr; p.out_amax = nullptr;
    p.a = (const float*)a; p.a2 = nullptr; p.bp = (const unsigned short*)b_planes;
    p.bias = bias ? bias : consts; p.bias_cm = bias ? 1 : 0;
    p.cin = c_in ? c_in : consts; p.ldcin = c_in ? ldcin : 0; p.cin_cm = c_in ? 1 : 0;
    p.cin2 = consts; p.ldcin2 = 0; p.cin2_cm = 0; p.gate = consts + 4; p.ldgate = 0; p.gate_cm = 0;
    g3_no_bits(p, consts);
    p.c = c_dtype == MPF_F32 ? (float*)c : nullptr; p.c16 = c_dtype == MPF_BF16 ? (unsigned short*)c : nullptr;
    p.lda = lda; p.ldc = ldc; p.plane = (int64_t)N * K;
    p.M = M; p.N = N; p.K = K; p.a2_rows = 0; p.relu = relu;
    p.cv_H = p.cv_W = p.cv_cin = 0; p.cv_sign = 1;
    p.tm0 = 0; p.ntiles2 = 0; p.tiles_n2 = 0;
    const int tiles_m = (M + kBM - 1) / kBM;
    const int waste128 = ((N + 127) / 128) * 128 - N, waste96 = ((N + 95) / 96) * 96 - N;
    const bool use96 = waste96 < waste128;
    p.tiles_n = use96 ? (N + 95) / 96 : (N + 127) / 128;
    p.ntiles = tiles_m * p.tiles_n;
    if (a_dtype == MPF_F32 && g3_launch_two_pass(p, st)) {
        mpf::prof_end(mpf_last_kernel(), st, 4.0 * (double)M * K + 2.0 * (double)M * N + 6.0 * (double)N * K, 2.0 * M * (double)N * K);
        return mpf::check(hipGetLastError(), "mpf_gemm3_tn_ex");
    }
    const dim3 grid(((p.ntiles + 7) / 8) * 8);
    mpf::prof_begin(st);
    if (a_dtype == MPF_BF16) {
        mpf::set_kernel("gemm3_tn_kernel<a16>");
        if (use96) hipLaunchKernelGGL(gemm3_tn_abf_kernel<96>, grid, dim3(kThreads), 0, st, p);
        else hipLaunchKernelGGL(gemm3_tn_abf_kernel<128>, grid, dim3(kThreads), 0, st, p);
    } else {
        mpf::set_kernel("gemm3_tn_kernel<c16>");
        if (use96) hipLaunchKernelGGL((gemm3_tn_kernel<96, false>), grid, dim3(kThreads), 0, st, p);
        else hipLaunchKernelGGL((gemm3_tn_kernel<128, false>), grid, dim3(kThreads), 0, st, p);
    }
    // flops: the fp32 GEMM it stands for (a bf16 A needs 3 of the 6 products)
    mpf::prof_end(mpf_last_kernel(), st, (a_dtype == MPF_BF16 ? 2.0 : 4.0) * (double)M * K + (c_dtype == MPF_BF16 ? 2.0 : 4.0) * (double)M * N + 6.0 * (double)N * K,
                  2.0 * M * (double)N * K);
    return mpf::check(hipGetLastError(), "mpf_gemm3_tn_ex");
}

static int g3_conv_impl(const float* x, const void* w_planes, const float* bias, float* y, int n_img, int H, int W, int Cin,
                        int Cout, int transposed, void* stream, const float* x_amax, const float* w_amax, float* out_amax);

extern "C" int mpf_gemm3_conv3x3(const float* x, const void* w_planes, const float* bias, float* y, int n_img, int H, int W, int Cin,
                                 int Cout, int transposed, void* stream)
{
    return g3_conv_impl(x, w_planes, bias, y, n_img, H, W, Cin, Cout, transposed, stream, nullptr, nullptr, nullptr);
}

extern "C" int mpf_gemm3_conv3x3_h2(const float* x, const float* x_amax, const void* w_planes_h2, const float* w_amax, const float* bias,
                                    float* y, float* out_amax, int n_img, int H, int W, int Cin, int Cout, int transposed, void* stream)
{
    if (!x_amax || !w_amax) return mpf::fail(MPF_E_NULL, "gemm3_conv3x3_h2: NULL amax");
    if (Cout % 256 != 0) return mpf::fail(MPF_E_SHAPE, "gemm3_conv3x3_h2: Cout must be a multiple of 256");
    return g3_conv_impl(x, w_planes_h2, bias, y, n_img, H, W, Cin, Cout, transposed, stream, x_amax, w_amax, out_amax);
}

static int g3_conv_impl(const float* x, const void* w_planes, const float* bias, float* y, int n_img, int H, int W, int Cin,
                        int Cout, int transposed, void* stream, const float* x_amax, const float* w_amax, float* out_amax)
{
    hipStream_t st = (hipStream_t)stream;
    if (!x || !w_planes || !y) return mpf::fail(MPF_E_NULL, "gemm3_conv3x3: NULL buffer");
    if (n_img <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 != 0 || Cout % 4 != 0)
        return mpf::fail(MPF_E_SHAPE, "gemm3_conv3x3: Cin must be a multiple of 32, Cout of 4");
    const int64_t M64 = (int64_t)n_img * H * W;
    if (M64 >= (1ll << 31) / 4 || M64 * Cin >= (1ll << 40)) return mpf::fail(MPF_E_TOO_LARGE, "gemm3_conv3x3: image too large");
    G3 p;
    p.a_amax = nullptr; p.b_amax = nullptr; p.out_amax = nullptr;
    const float* consts = nullptr;
    if (int rc = g3_consts(&consts, "gemm3_conv3x3: constants")) return rc;
    p.a = x; p.a2 = nullptr; p.bp = (const unsigned short*)w_planes; p.c = y;
    p.bias = bias ? bias : consts; p.bias_cm = bias ? 1 : 0;
    p.cin = consts; p.ldcin = 0; p.cin_cm = 0; p.cin2 = consts; p.ldcin2 = 0; p.cin2_cm = 0; p.gate = consts + 4; p.ldgate = 0; p.gate_cm = 0;
    g3_no_bits(p, consts);
    p.M = (int)M64; p.N = Cout; p.K = 9 * Cin; p.lda = Cin; p.ldc = Cout; p.plane = (int64_t)Cout * p.K;
    p.a2_rows = 0; p.relu = 0; p.c16 = nullptr;
    p.cv_H = H; p.cv_W = W; p.cv_cin = Cin; p.cv_sign = transposed ? -1 : 1;
    p.tiles_n = (Cout + 127) / 128;
    p.ntiles = ((p.M + kBM - 1) / kBM) * p.tiles_n;
    p.tm0 = 0; p.ntiles2 = 0; p.tiles_n2 = 0;
    mpf::prof_begin(st);
    mpf::set_kernel("gemm3_conv_kernel");
    p.a_amax = x_amax; p.b_amax = w_amax; p.out_amax = out_amax;
    if (x_amax) mpf::set_kernel("gemm3_conv_kernel<h2>");
    if (x_amax) {
        p.tiles_n = Cout / 256;
        p.ntiles = ((p.M + kBM - 1) / kBM) * p.tiles_n;
        hipLaunchKernelGGL((gemm3_tn2_kernel<128, true, true>), dim3(((p.ntiles + 7) / 8) * 8), dim3(kThreads), 0, st, p);
    } else if (g_two_pass > 0 && Cout % 256 == 0) {       // 128 x 256 tiles, two passes over one A image per K step
        p.tiles_n = Cout / 256;
        p.ntiles = ((p.M + kBM - 1) / kBM) * p.tiles_n;
        hipLaunchKernelGGL((gemm3_tn2_kernel<128, true>), dim3(((p.ntiles + 7) / 8) * 8), dim3(kThreads), 0, st, p);
    } else
    hipLaunchKernelGGL(gemm3_conv_kernel, dim3(((p.ntiles + 7) / 8) * 8), dim3(kThreads), 0, st, p);
    mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)p.M * Cin + (double)p.M * Cout) + (x_amax ? 4.0 : 6.0) * (double)Cout * p.K,
                  2.0 * p.M * (double)Cout * p.K);
    return mpf::check(hipGetLastError(), "mpf_gemm3_conv3x3");
}

static int g3_nt_impl(const float* a, int64_t lda, const float* b, int64_t ldb, const float* b2, int64_t ldb2, int b2_rows,
                      float* c_part, float* csum_a, float* csum_b, int R, int Mdim, int Ndim, int rows_per_split,
                      int transpose_out, void* stream, const float* a_amax, const float* b_amax);

extern "C" int mpf_gemm3_nt(const float* a, int64_t lda, const float* b, int64_t ldb, const float* b2, int64_t ldb2, int b2_rows,
                            float* c_part, float* csum_a, float* csum_b, int R, int Mdim, int Ndim, int rows_per_split,
                            int transpose_out, void* stream)
{
    return g3_nt_impl(a, lda, b, ldb, b2, ldb2, b2_rows, c_part, csum_a, csum_b, R, Mdim, Ndim, rows_per_split, transpose_out, stream,
                      nullptr, nullptr);
}

extern "C" int mpf_gemm3_nt_h2(const float* a, int64_t lda, const float* a_amax, const float* b, int64_t ldb, const float* b_amax,
                               float* c_part, float* csum_a, float* csum_b, int R, int Mdim, int Ndim, int rows_per_split,
                               int transpose_out, void* stream)
{
    if (!a_amax || !b_amax) return mpf::fail(MPF_E_NULL, "gemm3_nt_h2: NULL amax");
    return g3_nt_impl(a, lda, b, ldb, nullptr, 0, 0, c_part, csum_a, csum_b, R, Mdim, Ndim, rows_per_split, transpose_out, stream,
                      a_amax, b_amax);
}

static int g3_nt_impl(const float* a, int64_t lda, const float* b, int64_t ldb, const float* b2, int64_t ldb2, int b2_rows,
                      float* c_part, float* csum_a, float* csum_b, int R, int Mdim, int Ndim, int rows_per_split,
                      int transpose_out, void* stream, const float* a_amax, const float* b_amax)
{
    hipStream_t st = (hipStream_t)stream;
    if (!a || !b || !c_part) return mpf::fail(MPF_E_NULL, "gemm3_nt: NULL buffer");
    if (R <= 0 || Mdim <= 0 || Ndim <= 0 || rows_per_split <= 0 || rows_per_split % kBK != 0)
        return mpf::fail(MPF_E_SHAPE, "gemm3_nt: bad sizes (rows_per_split must be a positive multiple of 32)");
    if (b2 && b2_rows <= 0) return mpf::fail(MPF_E_SHAPE, "gemm3_nt: b2_rows must be positive");
    if (!transpose_out && Ndim % 4 != 0) return mpf::fail(MPF_E_SHAPE, "gemm3_nt: Ndim must be a multiple of 4 unless transpose_out");
    G3N p;
    p.a_amax = nullptr; p.b_amax = nullptr;
    p.a = a; p.b = b; p.b2 = b2; p.c = c_part; p.csum_a = csum_a; p.csum_b = csum_b;
    p.lda = lda; p.ldb = ldb; p.ldb2 = ldb2;
    p.R = R; p.Mdim = Mdim; p.Ndim = Ndim; p.b2_rows = b2_rows; p.rows_per_split = rows_per_split;
    p.nsplit = (R + rows_per_split - 1) / rows_per_split;
    p.transpose_out = transpose_out;
    {
        const uint64_t ab = ((uint64_t)(R - 1) * lda + Mdim) * 4, bb = ((uint64_t)(R - 1) * ldb + Ndim) * 4;
        if (ab >= (1ull << 32) || bb >= (1ull << 32) || lda * 4 >= (1ll << 31) || ldb * 4 >= (1ll << 31))
            return mpf::fail(MPF_E_TOO_LARGE, "gemm3_nt: an operand spans 4 GiB or more");
        p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    }
    p.tiles_m = (Mdim + kBM - 1) / kBM;
    const int waste128 = ((Ndim + 127) / 128) * 128 - Ndim, waste96 = ((Ndim + 95) / 96) * 96 - Ndim;
    const bool use96 = waste96 < waste128;
    p.tiles_n = use96 ? (Ndim + 95) / 96 : (Ndim + 127) / 128;
    p.ntiles = p.tiles_m * p.tiles_n * p.nsplit;
    p.c_ss = (int64_t)p.Mdim * p.Ndim; p.csa_ss = p.Mdim; p.csb_ss = p.Ndim;
    const int grid = ((p.ntiles + 7) / 8) * 8;
    p.a_amax = a_amax; p.b_amax = b_amax;
    mpf::prof_begin(st);
    if (a_amax) {
        if (use96) {
            mpf::set_kernel("gemm3_nt_kernel<96>h2");
            hipLaunchKernelGGL((gemm3_nt_kernel<96, false, false, false, false, true>), dim3(grid), dim3(kThreads), 0, st, p);
        } else {
            mpf::set_kernel("gemm3_nt_kernel<128>h2");
            hipLaunchKernelGGL((gemm3_nt_kernel<128, false, false, false, false, true>), dim3(grid), dim3(kThreads), 0, st, p);
        }
    } else if (use96) {
        mpf::set_kernel("gemm3_nt_kernel<96>");
        hipLaunchKernelGGL(gemm3_nt_kernel<96>, dim3(grid), dim3(kThreads), 0, st, p);
    } else {
        mpf::set_kernel("gemm3_nt_kernel<128>");
        hipLaunchKernelGGL(gemm3_nt_kernel<128>, dim3(grid), dim3(kThreads), 0, st, p);
    }
    mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)R * Mdim + (double)R * Ndim + (double)p.nsplit * Mdim * Ndim),
                  2.0 * R * (double)Mdim * Ndim);
    return mpf::check(hipGetLastError(), "mpf_gemm3_nt");
}

template <typename Item, bool H2>
static int g3_nt_grouped_impl(const Item* items, int n_items, int R, int rows_per_split, int64_t split_stride, void* stream);

extern "C" int mpf_gemm3_nt_grouped(const MpfNtItem* items, int n_items, int R, int rows_per_split, int64_t split_stride,
                                    void* stream)
{
    return g3_nt_grouped_impl<MpfNtItem, false>(items, n_items, R, rows_per_split, split_stride, stream);
}

extern "C" int mpf_gemm3_nt_grouped_h2(const MpfNtItemH2* items, int n_items, int R, int rows_per_split, int64_t split_stride,
                                       void* stream)
{
    return g3_nt_grouped_impl<MpfNtItemH2, true>(items, n_items, R, rows_per_split, split_stride, stream);
}

template <typename Item, bool H2>
static int g3_nt_grouped_impl(const Item* items, int n_items, int R, int rows_per_split, int64_t split_stride, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!items) return mpf::fail(MPF_E_NULL, "gemm3_nt_grouped: NULL items");
    if (n_items <= 0 || n_items > kNtGroupMax) return mpf::fail(MPF_E_SHAPE, "gemm3_nt_grouped: 1..8 items");
    if (R <= 0 || rows_per_split <= 0 || rows_per_split % kBK != 0 || split_stride <= 0)
        return mpf::fail(MPF_E_SHAPE, "gemm3_nt_grouped: bad sizes (rows_per_split must be a positive multiple of 32)");
    G3NG g;
    memset(&g, 0, sizeof(g));
    const int nsplit = (R + rows_per_split - 1) / rows_per_split;
    int tiles = 0;
    double bytes = 0.0, flops = 0.0;
    bool big = H2 && g_nt2;                          // 256 x 256 tiles (gemm3_nt2.h): every dimension a multiple of 256
    for (int i = 0; i < n_items && big; ++i) big = items[i].Mdim % 256 == 0 && items[i].Ndim % 256 == 0 && items[i].Mdim > 0 && items[i].Ndim > 0;
    for (int i = 0; i < n_items; ++i) {
        const Item& it = items[i];
        if (!it.a || !it.b || !it.c_part) return mpf::fail(MPF_E_NULL, "gemm3_nt_grouped: NULL buffer");
        if (it.Mdim <= 0 || it.Ndim <= 0 || it.Ndim % 4 != 0 || it.Mdim > (1 << 20) || it.Ndim > (1 << 20))
            return mpf::fail(MPF_E_SHAPE, "gemm3_nt_grouped: Ndim must be a positive multiple of 4");
        G3N& p = g.it[i];
        p.a = it.a; p.b = it.b; p.b2 = nullptr; p.c = it.c_part; p.csum_a = it.csum_a; p.csum_b = nullptr;
        p.lda = it.lda; p.ldb = it.ldb; p.ldb2 = 0;
        p.R = R; p.Mdim = (int)it.Mdim; p.Ndim = (int)it.Ndim; p.b2_rows = 0; p.rows_per_split = rows_per_split;
        p.nsplit = nsplit;
        p.transpose_out = 0;
        const uint64_t ab = ((uint64_t)(R - 1) * it.lda + it.Mdim) * 4, bb = ((uint64_t)(R - 1) * it.ldb + it.Ndim) * 4;
        if (ab >= (1ull << 32) || bb >= (1ull << 32) || it.lda * 4 >= (1ll << 31) || it.ldb * 4 >= (1ll << 31))
            return mpf::fail(MPF_E_TOO_LARGE, "gemm3_nt_grouped: an operand spans 4 GiB or more");
        p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
        p.tiles_m = big ? p.Mdim / 256 : (p.Mdim + kBM - 1) / kBM;
        p.tiles_n = big ? p.Ndim / 256 : (p.Ndim + 127) / 128;
        p.ntiles = p.tiles_m * p.tiles_n * nsplit;
        p.c_ss = p.csa_ss = split_stride; p.csb_ss = 0;
        p.a_amax = nullptr; p.b_amax = nullptr;
        if constexpr (H2) {
            if (!it.a_amax || !it.b_amax) return mpf::fail(MPF_E_NULL, "gemm3_nt_grouped_h2: NULL amax");
            p.a_amax = it.a_amax; p.b_amax = it.b_amax;
        }
        tiles += p.ntiles;
        g.tile_end[i] = tiles;
        bytes += 4.0 * ((double)R * p.Mdim + (double)R * p.Ndim + (double)nsplit * p.Mdim * p.Ndim);
        flops += 2.0 * R * (double)p.Mdim * p.Ndim;
    }
    g.n_items = n_items; g.ntiles = tiles;
    if (big) {
        static mpf::LdsAttr attr;
        if (int e = mpf::ensure_dynamic_lds((const void*)gemm3_nt2_group_kernel, kN2Lds, attr)) return e;
        mpf::prof_begin(st);
        mpf::set_kernel("gemm3_nt_group_kernel<h2 256x256>");
        hipLaunchKernelGGL(gemm3_nt2_group_kernel, dim3(((tiles + 7) / 8) * 8), dim3(kN2T), kN2Lds, st, g);
        mpf::prof_end(mpf_last_kernel(), st, bytes, flops);
        return mpf::check(hipGetLastError(), "mpf_gemm3_nt_grouped(256x256)");
    }
    mpf::prof_begin(st);
    mpf::set_kernel(H2 ? "gemm3_nt_group_kernel<h2>" : "gemm3_nt_group_kernel");
    hipLaunchKernelGGL(gemm3_nt_group_kernel<H2>, dim3(((tiles + 7) / 8) * 8), dim3(kThreads), 0, st, g);
    mpf::prof_end(mpf_last_kernel(), st, bytes, flops);
    return mpf::check(hipGetLastError(), "mpf_gemm3_nt_grouped");
}

namespace {

// out[j] = bf16(sum_s part[s][j]): the split partials of the bf16 weight-gradient GEMM and of its column
// sums, both in one launch (float4 per thread per split, coalesced; summation order fixed)
template <bool F32OUT>
__global__ __launch_bounds__(256) void nt_reduce_kernel(const float* __restrict__ c_part, int64_t cn,
                                                        const float* __restrict__ s_part, int64_t sn, int nsplit,
                                                        void* __restrict__ c_out_, void* __restrict__ s_out_)
{
    // one float2 per thread (a wave reads 512 contiguous bytes per split), 8 splits in flight
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t cq = cn >> 1, sq = sn >> 1;
    if (q >= cq + sq) return;
    const bool is_c = q < cq;
    const float* src = is_c ? c_part + 2 * q : s_part + 2 * (q - cq);
    const int64_t stride = is_c ? cn : sn;
    float2 acc = make_float2(0.f, 0.f);
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {
        float2 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const float2*>(src + (int64_t)(sp + k) * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc.x += t[k].x; acc.y += t[k].y; }
    }
    for (; sp < nsplit; ++sp) {
        const float2 t = *reinterpret_cast<const float2*>(src + (int64_t)sp * stride);
        acc.x += t.x; acc.y += t.y;
    }
    if (F32OUT) {
        float* dst = is_c ? static_cast<float*>(c_out_) + 2 * q : static_cast<float*>(s_out_) + 2 * (q - cq);
        *reinterpret_cast<float2*>(dst) = acc;
        return;
    }
    auto rne = [](float f) -> unsigned {
        unsigned u = __float_as_uint(f);
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    };
    unsigned short* dst = is_c ? static_cast<unsigned short*>(c_out_) + 2 * q : static_cast<unsigned short*>(s_out_) + 2 * (q - cq);
    *reinterpret_cast<unsigned*>(dst) = rne(acc.x) | (rne(acc.y) << 16);
}

}  // namespace

extern "C" int mpf_gemm3_nt_ex(const void* a, int a_dtype, int64_t lda, const void* b, int b_dtype, int64_t ldb, float* c_part,
                               float* csum_a, int R, int Mdim, int Ndim, int rows_per_split, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!a || !b || !c_part) return mpf::fail(MPF_E_NULL, "gemm3_nt_ex: NULL buffer");
    const bool a16 = a_dtype == MPF_BF16, b16 = b_dtype == MPF_BF16;
    if ((!a16 && a_dtype != MPF_F32) || (!b16 && b_dtype != MPF_F32) || (a16 && b16))
        return mpf::fail(MPF_E_DTYPE, "gemm3_nt_ex: operands are MPF_F32 or MPF_BF16, at most one of them bf16 (both: mpf_gemm_nt_bf16)");
    if (!a16 && !b16)
        return mpf_gemm3_nt((const float*)a, lda, (const float*)b, ldb, nullptr, 0, 0, c_part, csum_a, nullptr, R, Mdim, Ndim, rows_per_split, 0, stream);
    if (R <= 0 || Mdim <= 0 || Ndim <= 0 || Ndim % 128 != 0 || rows_per_split <= 0 || rows_per_split % kBK != 0)
        return mpf::fail(MPF_E_SHAPE, "gemm3_nt_ex: Ndim must be a multiple of 128, rows_per_split a positive multiple of 32");
    G3N p;
    p.a_amax = nullptr; p.b_amax = nullptr;
    p.a = (const float*)a; p.b = (const float*)b; p.b2 = nullptr; p.c = c_part; p.csum_a = csum_a; p.csum_b = nullptr;
    p.lda = lda; p.ldb = ldb; p.ldb2 = 0;
    p.R = R; p.Mdim = Mdim; p.Ndim = Ndim; p.b2_rows = 0; p.rows_per_split = rows_per_split;
    p.nsplit = (R + rows_per_split - 1) / rows_per_split;
    p.transpose_out = 0;
    p.cv_H = p.cv_W = p.cv_cin = 0;
    {
        const uint64_t ab = ((uint64_t)(R - 1) * lda + Mdim) * (a16 ? 2 : 4), bb = ((uint64_t)(R - 1) * ldb + Ndim) * (b16 ? 2 : 4);
        if (ab >= (1ull << 32) || bb >= (1ull << 32)) return mpf::fail(MPF_E_TOO_LARGE, "gemm3_nt_ex: an operand spans 4 GiB or more");
        p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    }
    p.tiles_m = (Mdim + kBM - 1) / kBM;
    p.tiles_n = Ndim / 128;
    p.ntiles = p.tiles_m * p.tiles_n * p.nsplit;
    p.c_ss = (int64_t)p.Mdim * p.Ndim; p.csa_ss = p.Mdim; p.csb_ss = p.Ndim;
    const dim3 grid(((p.ntiles + 7) / 8) * 8);
    mpf::prof_begin(st);
    mpf::set_kernel(a16 ? "gemm3_nt_kernel<a16>" : "gemm3_nt_kernel<b16>");
    if (a16) hipLaunchKernelGGL((gemm3_nt_kernel<128, false, false, true, false>), grid, dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL((gemm3_nt_kernel<128, false, false, false, true>), grid, dim3(kThreads), 0, st, p);
    mpf::prof_end(mpf_last_kernel(), st, (a16 ? 2.0 : 4.0) * (double)R * Mdim + (b16 ? 2.0 : 4.0) * (double)R * Ndim + 4.0 * (double)p.nsplit * Mdim * Ndim,
                  2.0 * R * (double)Mdim * Ndim);
    return mpf::check(hipGetLastError(), "mpf_gemm3_nt_ex");
}

static int g3_conv_wgrad_impl(const float* dy, const float* x, float* c_part, float* csum_dy, int n_img, int H, int W, int Cin,
                              int Cout, int rows_per_split, void* stream, const float* dy_amax, const float* x_amax);

extern "C" int mpf_gemm3_conv3x3_wgrad(const float* dy, const float* x, float* c_part, float* csum_dy, int n_img, int H, int W, int Cin,
                                       int Cout, int rows_per_split, void* stream)
{
    return g3_conv_wgrad_impl(dy, x, c_part, csum_dy, n_img, H, W, Cin, Cout, rows_per_split, stream, nullptr, nullptr);
}

extern "C" int mpf_gemm3_conv3x3_wgrad_h2(const float* dy, const float* dy_amax, const float* x, const float* x_amax, float* c_part,
                                          float* csum_dy, int n_img, int H, int W, int Cin, int Cout, int rows_per_split, void* stream)
{
    if (!dy_amax || !x_amax) return mpf::fail(MPF_E_NULL, "gemm3_conv3x3_wgrad_h2: NULL amax");
    return g3_conv_wgrad_impl(dy, x, c_part, csum_dy, n_img, H, W, Cin, Cout, rows_per_split, stream, dy_amax, x_amax);
}

static int g3_conv_wgrad_impl(const float* dy, const float* x, float* c_part, float* csum_dy, int n_img, int H, int W, int Cin,
                              int Cout, int rows_per_split, void* stream, const float* dy_amax, const float* x_amax)
{
    hipStream_t st = (hipStream_t)stream;
    if (!dy || !x || !c_part) return mpf::fail(MPF_E_NULL, "gemm3_conv3x3_wgrad: NULL buffer");
    if (n_img <= 0 || H <= 0 || W <= 0 || W % 8 != 0 || Cin <= 0 || Cin % 128 != 0 || Cout <= 0 || rows_per_split <= 0 ||
        rows_per_split % kBK != 0)
        return mpf::fail(MPF_E_SHAPE, "gemm3_conv3x3_wgrad: needs W % 8 == 0, Cin % 128 == 0, rows_per_split % 32 == 0");
    const int64_t R64 = (int64_t)n_img * H * W;
    if (R64 * (Cin > Cout ? Cin : Cout) * 4 >= (1ll << 32)) return mpf::fail(MPF_E_TOO_LARGE, "gemm3_conv3x3_wgrad: an operand spans 4 GiB or more");
    G3N p;
    p.a_amax = nullptr; p.b_amax = nullptr;
    p.a = dy; p.b = x; p.b2 = nullptr; p.c = c_part; p.csum_a = csum_dy; p.csum_b = nullptr;
    p.lda = Cout; p.ldb = Cin; p.ldb2 = 0;
    p.R = (int)R64; p.Mdim = Cout; p.Ndim = 9 * Cin; p.b2_rows = 0; p.rows_per_split = rows_per_split;
    p.nsplit = (p.R + rows_per_split - 1) / rows_per_split;
    p.transpose_out = 0;
    p.a_bytes = (unsigned)(R64 * Cout * 4); p.b_bytes = (unsigned)(R64 * Cin * 4);
    p.cv_H = H; p.cv_W = W; p.cv_cin = Cin;
    p.tiles_m = (Cout + kBM - 1) / kBM;
    p.tiles_n = 9 * Cin / 128;
    p.ntiles = p.tiles_m * p.tiles_n * p.nsplit;
    p.c_ss = (int64_t)p.Mdim * p.Ndim; p.csa_ss = p.Mdim; p.csb_ss = p.Ndim;
    p.a_amax = dy_amax; p.b_amax = x_amax;
    if (dy_amax && g_nt2 && Cout % 256 == 0 && Cin % 256 == 0) {       // 256 x 256 tiles (gemm3_nt2.h): half the operand traffic
        p.tiles_m = Cout / 256; p.tiles_n = 9 * Cin / 256;
        p.ntiles = p.tiles_m * p.tiles_n * p.nsplit;
        static mpf::LdsAttr attr;
        if (int e = mpf::ensure_dynamic_lds((const void*)gemm3_nt2_conv_kernel, kN2Lds, attr)) return e;
        mpf::prof_begin(st);
        mpf::set_kernel("gemm3_nt_kernel<conv3x3 h2 256x256>");
        hipLaunchKernelGGL(gemm3_nt2_conv_kernel, dim3(((p.ntiles + 7) / 8) * 8), dim3(kN2T), kN2Lds, st, p);
        mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)p.R * Cout + (double)p.R * Cin + (double)p.nsplit * Cout * 9 * Cin),
                      2.0 * p.R * (double)Cout * 9 * Cin);
        return mpf::check(hipGetLastError(), "mpf_gemm3_conv3x3_wgrad(256x256)");
    }
    mpf::prof_begin(st);
    mpf::set_kernel(dy_amax ? "gemm3_nt_kernel<conv3x3 h2>" : "gemm3_nt_kernel<conv3x3>");
    if (dy_amax) hipLaunchKernelGGL((gemm3_nt_kernel<128, false, true, false, false, true>), dim3(((p.ntiles + 7) / 8) * 8), dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL((gemm3_nt_kernel<128, false, true>), dim3(((p.ntiles + 7) / 8) * 8), dim3(kThreads), 0, st, p);
    mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)p.R * Cout + (double)p.R * Cin + (double)p.nsplit * Cout * 9 * Cin),
                  2.0 * p.R * (double)Cout * 9 * Cin);
    return mpf::check(hipGetLastError(), "mpf_gemm3_conv3x3_wgrad");
}

extern "C" size_t mpf_gemm_nt_bf16_workspace_bytes(int R, int Mdim, int Ndim, int rows_per_split)
{
    if (R <= 0 || Mdim <= 0 || Ndim <= 0 || rows_per_split <= 0) return 0;
    const size_t ns = (size_t)(R + rows_per_split - 1) / rows_per_split;
    return ns * ((size_t)Mdim * Ndim + Mdim) * sizeof(float);
}

extern "C" int mpf_gemm_nt_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* c_out, void* csum_out, int R,
                                int Mdim, int Ndim, int rows_per_split, void* workspace, size_t workspace_bytes, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!a || !b || !c_out || !workspace) return mpf::fail(MPF_E_NULL, "gemm_nt_bf16: NULL buffer");
    if (R <= 0 || Mdim <= 0 || Ndim <= 0 || rows_per_split <= 0 || rows_per_split % kBK != 0 || Mdim % 4 || Ndim % 4)
        return mpf::fail(MPF_E_SHAPE, "gemm_nt_bf16: bad sizes (rows_per_split % 32, Mdim % 4, Ndim % 4 must be 0)");
    if (workspace_bytes < mpf_gemm_nt_bf16_workspace_bytes(R, Mdim, Ndim, rows_per_split))
        return mpf::fail(MPF_E_SHAPE, "gemm_nt_bf16: workspace too small");
    G3N p;
    p.a_amax = nullptr; p.b_amax = nullptr;
    p.a = static_cast<const float*>(a); p.b = static_cast<const float*>(b); p.b2 = nullptr;
    p.nsplit = (R + rows_per_split - 1) / rows_per_split;
    p.c = static_cast<float*>(workspace);
    p.csum_a = p.c + (size_t)p.nsplit * Mdim * Ndim;
    p.csum_b = nullptr;
    p.lda = lda; p.ldb = ldb; p.ldb2 = 0;
    p.R = R; p.Mdim = Mdim; p.Ndim = Ndim; p.b2_rows = 0; p.rows_per_split = rows_per_split;
    p.transpose_out = 0;
    {
        const uint64_t ab = ((uint64_t)(R - 1) * lda + Mdim) * 2, bb = ((uint64_t)(R - 1) * ldb + Ndim) * 2;
        if (ab >= (1ull << 32) || bb >= (1ull << 32)) return mpf::fail(MPF_E_TOO_LARGE, "gemm_nt_bf16: an operand spans 4 GiB or more");
        p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    }
    p.tiles_m = (Mdim + kBM - 1) / kBM;
    p.tiles_n = (Ndim + 127) / 128;
    p.ntiles = p.tiles_m * p.tiles_n * p.nsplit;
    p.c_ss = (int64_t)p.Mdim * p.Ndim; p.csa_ss = p.Mdim; p.csb_ss = p.Ndim;
    const int grid = ((p.ntiles + 7) / 8) * 8;
    mpf::prof_begin(st);
    mpf::set_kernel("gemm3_nt_kernel<128, bf16>");
    hipLaunchKernelGGL((gemm3_nt_kernel<128, true>), dim3(grid), dim3(kThreads), 0, st, p);
    mpf::prof_end("gemm3_nt_kernel<128, bf16>", st, 2.0 * ((double)R * Mdim * p.tiles_n + (double)R * Ndim * p.tiles_m) +
                                                       4.0 * (double)p.nsplit * Mdim * Ndim);
    const int64_t cn = (int64_t)Mdim * Ndim, sn = csum_out ? Mdim : 0;
    const int64_t quads = (cn + sn) / 2;
    hipLaunchKernelGGL(nt_reduce_kernel<false>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, p.c, cn, p.csum_a, sn,
                       p.nsplit, c_out, csum_out);
    return mpf::check(hipGetLastError(), "mpf_gemm_nt_bf16");
}

// the same sum with the column sums ALSO grouped by a per-split key (the feature level of a split's rows):
// lvl_out[l][j] = sum over the splits of level l of s_part[s][j], s_out = sum over l.  Replaces zeros + index_add_ (float
// atomics) + two torch reductions per encoder layer; fixed order.
__global__ __launch_bounds__(256) void nt_reduce_levels_kernel(const float* __restrict__ c_part, int64_t cn, const float* __restrict__ s_part,
                                                               int64_t sn, int nsplit, const int64_t* __restrict__ level_of_split, int L,
                                                               float* __restrict__ c_out, float* __restrict__ lvl_out,
                                                               float* __restrict__ s_out)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t cq = cn >> 1, sq = sn >> 1;
    if (q >= cq + sq) return;
    if (q < cq) {
        const float* src = c_part + 2 * q;
        float2 acc = make_float2(0.f, 0.f);
        int sp = 0;
        for (; sp + 8 <= nsplit; sp += 8) {
            float2 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const float2*>(src + (int64_t)(sp + k) * cn);
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc.x += t[k].x; acc.y += t[k].y; }
        }
        for (; sp < nsplit; ++sp) {
            const float2 t = *reinterpret_cast<const float2*>(src + (int64_t)sp * cn);
            acc.x += t.x; acc.y += t.y;
        }
        *reinterpret_cast<float2*>(c_out + 2 * q) = acc;
        return;
    }
    const int64_t j = 2 * (q - cq);
    float2 acc[4] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {                       // 8 splits in flight (loads first, then the selects)
        float2 t[8];
        int lv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            t[k] = *reinterpret_cast<const float2*>(s_part + (int64_t)(sp + k) * sn + j);
            lv[k] = (int)level_of_split[sp + k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                acc[l].x += lv[k] == l ? t[k].x : 0.f;
                acc[l].y += lv[k] == l ? t[k].y : 0.f;
            }
    }
    for (; sp < nsplit; ++sp) {
        const float2 t = *reinterpret_cast<const float2*>(s_part + (int64_t)sp * sn + j);
        const int lv = (int)level_of_split[sp];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            acc[l].x += lv == l ? t.x : 0.f;
            acc[l].y += lv == l ? t.y : 0.f;
        }
    }
    float2 tot = make_float2(0.f, 0.f);
#pragma unroll
    for (int l = 0; l < 4; ++l)
        if (l < L) {
            *reinterpret_cast<float2*>(lvl_out + (int64_t)l * sn + j) = acc[l];
            tot.x += acc[l].x; tot.y += acc[l].y;
        }
    *reinterpret_cast<float2*>(s_out + j) = tot;
}

extern "C" int mpf_gemm3_nt_reduce_levels(const float* c_part, int64_t c_numel, const float* s_part, int64_t s_numel, int nsplit,
                                          const int64_t* level_of_split, int n_levels, float* c_out, float* lvl_out, float* s_out,
                                          void* stream)
{
    if (!c_part || !c_out || !s_part || !s_out || !lvl_out || !level_of_split) return mpf::fail(MPF_E_NULL, "gemm3_nt_reduce_levels: NULL buffer");
    if (nsplit <= 0 || c_numel <= 0 || s_numel <= 0 || c_numel % 4 || s_numel % 4 || n_levels < 1 || n_levels > 4)
        return mpf::fail(MPF_E_SHAPE, "gemm3_nt_reduce_levels: sizes must be positive multiples of 4, 1..4 levels");
    const int64_t quads = (c_numel + s_numel) / 2;
    mpf::set_kernel("nt_reduce_levels_kernel");
    hipLaunchKernelGGL(nt_reduce_levels_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c_part, c_numel,
                       s_part, s_numel, nsplit, level_of_split, n_levels, c_out, lvl_out, s_out);
    return mpf::check(hipGetLastError(), "mpf_gemm3_nt_reduce_levels");
}

extern "C" int mpf_gemm3_nt_reduce(const float* c_part, int64_t c_numel, const float* s_part, int64_t s_numel, int nsplit,
                                   float* c_out, float* s_out, void* stream)
{
    if (!c_part || !c_out || (s_numel > 0 && (!s_part || !s_out))) return mpf::fail(MPF_E_NULL, "gemm3_nt_reduce: NULL buffer");
    if (nsplit <= 0 || c_numel <= 0 || s_numel < 0 || c_numel % 4 || s_numel % 4)
        return mpf::fail(MPF_E_SHAPE, "gemm3_nt_reduce: sizes must be positive multiples of 4");
    const int64_t quads = (c_numel + s_numel) / 2;
    mpf::set_kernel("nt_reduce_kernel");
    hipLaunchKernelGGL(nt_reduce_kernel<true>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c_part,
                       c_numel, s_part, s_numel, nsplit, c_out, s_out);
    return mpf::check(hipGetLastError(), "mpf_gemm3_nt_reduce");
}

