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
// MFMA).  LDS image per operand and plane: [k-chunk of 8][row][16 B] with 64 B of padding per k-chunk,
// so that both the staging writes (4 lanes = 4 k-chunks of one row) and the fragment reads (16 lanes =
// 16 rows of one k-chunk) are bank-conflict free b128 accesses.  MFMA is issued as D^T = B.A^T so a lane
// owns 4 consecutive output columns of one row (16-B stores, bias as float4).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kBM = 128;
constexpr int kBK = 32;
constexpr int kThreads = 256;
constexpr int kAKc = kBM * 16 + 64;             // bytes per (plane, k-chunk) of the A image

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
};

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

__device__ __forceinline__ float4 add4(float4 x, float4 y) { return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }

__device__ __forceinline__ bf16x8 as_frag(const uint4 v)
{
    union { uint4 u; bf16x8 f; } c;
    c.u = v;
    return c.f;
}

template <int BN>
__global__ __launch_bounds__(kThreads, 2) void gemm3_tn_kernel(G3 p)
{
    constexpr int NJ = BN / 32;                  // 16-column MFMA tiles per wave
    constexpr int kBKc = BN * 16 + 64;           // bytes per (plane, k-chunk) of the B image
    constexpr int kAbytes = 12 * kAKc;
    constexpr int kBunits = 3 * BN * 4;          // 16-B units of a B stage
    constexpr int kBiter = (kBunits + kThreads - 1) / kThreads;
    __shared__ __attribute__((aligned(16))) unsigned char lds[kAbytes + 12 * kBKc];

    // XCD-aware tile order: each XCD walks a contiguous run of tiles (column tiles of one row block
    // are neighbours, so the A rows they share stay in that XCD's L2)
    const int per_xcd = (p.ntiles + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * kBM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;

    // ---- staging maps ---------------------------------------------------------------------------
    const int akc = tid & 3;
    const int arow0 = tid >> 2, arow1 = 64 + (tid >> 2);
    const float* ap0 = p.a + (int64_t)min(m0 + arow0, p.M - 1) * p.lda + akc * 8;
    const float* ap1 = p.a + (int64_t)min(m0 + arow1, p.M - 1) * p.lda + akc * 8;
    const float* a2p0 = nullptr;
    const float* a2p1 = nullptr;
    if (p.a2) {
        a2p0 = p.a2 + (int64_t)(min(m0 + arow0, p.M - 1) % p.a2_rows) * p.K + akc * 8;
        a2p1 = p.a2 + (int64_t)(min(m0 + arow1, p.M - 1) % p.a2_rows) * p.K + akc * 8;
    }
    const unsigned short* bsrc[kBiter];
    int bdst[kBiter];
#pragma unroll
    for (int i = 0; i < kBiter; ++i) {
        const int u = tid + i * kThreads;
        const int kc = u & 3, n = (u >> 2) % BN, pl = min(u / (4 * BN), 2);      // (units past the end: clamped, not stored)
        bsrc[i] = p.bp + (int64_t)pl * p.plane + (int64_t)min(n0 + n, p.N - 1) * p.K + kc * 8;
        bdst[i] = kAbytes + (pl * 4 + kc) * kBKc + n * 16;
    }

    float4 ra[4], ra2[4];
    uint4 rb0, rb1, rb2, rb3, rb4 = make_uint4(0, 0, 0, 0), rb5 = make_uint4(0, 0, 0, 0);
    static_assert(kBiter >= 4 && kBiter <= 6, "B staging assumes 4..6 units per thread");
#define G3_LOAD_STAGE(k0)                                                                    \
    {                                                                                        \
        ra[0] = *reinterpret_cast<const float4*>(ap0 + (k0));                                \
        ra[1] = *reinterpret_cast<const float4*>(ap0 + (k0) + 4);                            \
        ra[2] = *reinterpret_cast<const float4*>(ap1 + (k0));                                \
        ra[3] = *reinterpret_cast<const float4*>(ap1 + (k0) + 4);                            \
        if (p.a2) {                                                                          \
            ra2[0] = *reinterpret_cast<const float4*>(a2p0 + (k0));                          \
            ra2[1] = *reinterpret_cast<const float4*>(a2p0 + (k0) + 4);                      \
            ra2[2] = *reinterpret_cast<const float4*>(a2p1 + (k0));                          \
            ra2[3] = *reinterpret_cast<const float4*>(a2p1 + (k0) + 4);                      \
        }                                                                                    \
        rb0 = *reinterpret_cast<const uint4*>(bsrc[0] + (k0));                               \
        rb1 = *reinterpret_cast<const uint4*>(bsrc[1] + (k0));                               \
        rb2 = *reinterpret_cast<const uint4*>(bsrc[2] + (k0));                               \
        rb3 = *reinterpret_cast<const uint4*>(bsrc[3] + (k0));                               \
        if constexpr (kBiter > 4) rb4 = *reinterpret_cast<const uint4*>(bsrc[4] + (k0));     \
        if constexpr (kBiter > 5) rb5 = *reinterpret_cast<const uint4*>(bsrc[5] + (k0));     \
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ra2[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_frag = g * kAKc + (wr * 64 + r16) * 16;                       // + pl*4*kAKc + i*256
    const int b_frag = kAbytes + g * kBKc + (wc * (BN / 2) + r16) * 16;       // + pl*4*kBKc + j*256

    const int nk = p.K / kBK;
    G3_LOAD_STAGE(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        {
            uint4 h, m, l;
            split8(add4(ra[0], ra2[0]), add4(ra[1], ra2[1]), &h, &m, &l);
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKc + arow0 * 16) = h;
            *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKc + arow0 * 16) = m;
            *reinterpret_cast<uint4*>(lds + (2 * 4 + akc) * kAKc + arow0 * 16) = l;
            split8(add4(ra[2], ra2[2]), add4(ra[3], ra2[3]), &h, &m, &l);
            *reinterpret_cast<uint4*>(lds + (0 * 4 + akc) * kAKc + arow1 * 16) = h;
            *reinterpret_cast<uint4*>(lds + (1 * 4 + akc) * kAKc + arow1 * 16) = m;
            *reinterpret_cast<uint4*>(lds + (2 * 4 + akc) * kAKc + arow1 * 16) = l;
            *reinterpret_cast<uint4*>(lds + bdst[0]) = rb0;
            *reinterpret_cast<uint4*>(lds + bdst[1]) = rb1;
            *reinterpret_cast<uint4*>(lds + bdst[2]) = rb2;
            *reinterpret_cast<uint4*>(lds + bdst[3]) = rb3;
            if constexpr (kBiter > 4)
                if (kBunits >= 5 * kThreads || tid + 4 * kThreads < kBunits) *reinterpret_cast<uint4*>(lds + bdst[4]) = rb4;
            if constexpr (kBiter > 5)
                if (kBunits >= 6 * kThreads || tid + 5 * kThreads < kBunits) *reinterpret_cast<uint4*>(lds + bdst[5]) = rb5;
        }
        __syncthreads();
        if (kt + 1 < nk) G3_LOAD_STAGE((kt + 1) * kBK);
        bf16x8 fa[3][4];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[pl][i] = as_frag(*reinterpret_cast<const uint4*>(lds + a_frag + pl * 4 * kAKc + i * 256));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bf16x8 fb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb[pl] = as_frag(*reinterpret_cast<const uint4*>(lds + b_frag + pl * 4 * kBKc + j * 256));
            // smallest terms first; D^T = B . A^T
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[2][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[0][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[1][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[1][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[0][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[0][i], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns row m, columns n..n+3 of each tile -----------------------------------
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wc * (BN / 2) + j * 16 + g * 4;
        if (n >= p.N) continue;
        float4 bz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bz = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wr * 64 + i * 16 + r16;
            if (m >= p.M) continue;
            float4 o = make_float4(acc[i][j][0] + bz.x, acc[i][j][1] + bz.y, acc[i][j][2] + bz.z, acc[i][j][3] + bz.w);
            if (p.cin) {
                const float4 ci = *reinterpret_cast<const float4*>(p.cin + (int64_t)m * p.ldcin + n);
                o = make_float4(o.x + ci.x, o.y + ci.y, o.z + ci.z, o.w + ci.w);
            }
            if (p.cin2) {
                const float4 ci = *reinterpret_cast<const float4*>(p.cin2 + (int64_t)m * p.ldcin2 + n);
                o = make_float4(o.x + ci.x, o.y + ci.y, o.z + ci.z, o.w + ci.w);
            }
            if (p.relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            if (p.gate) {               // ReLU backward: pass the gradient where the saved activation is > 0
                const float4 gt = *reinterpret_cast<const float4*>(p.gate + (int64_t)m * p.ldgate + n);
                o = make_float4(gt.x > 0.f ? o.x : 0.f, gt.y > 0.f ? o.y : 0.f, gt.z > 0.f ? o.z : 0.f, gt.w > 0.f ? o.w : 0.f);
            }
            *reinterpret_cast<float4*>(p.c + (int64_t)m * p.ldc + n) = o;
        }
    }
}

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

}  // namespace

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

extern "C" int mpf_gemm3_tn(const float* a, int64_t lda, const float* a2, int a2_rows, const void* b_planes,
                            const float* bias, const float* c_in, int64_t ldcin, const float* c_in2, int64_t ldcin2,
                            const float* gate, int64_t ldgate, float* c, int64_t ldc, int M, int N, int K,
                            int relu, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!a || !b_planes || !c) return mpf::fail(MPF_E_NULL, "gemm3_tn: NULL buffer");
    if (M <= 0 || N <= 0 || K <= 0) return mpf::fail(MPF_E_SHAPE, "gemm3_tn: bad sizes");
    if (K % kBK != 0 || N % 4 != 0 || lda % 4 != 0 || ldc % 4 != 0 || (c_in && ldcin % 4 != 0) || (c_in2 && ldcin2 % 4 != 0) ||
        (gate && ldgate % 4 != 0))
        return mpf::fail(MPF_E_SHAPE, "gemm3_tn: K must be a multiple of 32; N, lda, ldc multiples of 4");
    if (a2 && a2_rows <= 0) return mpf::fail(MPF_E_SHAPE, "gemm3_tn: a2_rows must be positive");
    G3 p;
    p.a = a; p.a2 = a2; p.bp = (const unsigned short*)b_planes; p.bias = bias; p.cin = c_in; p.c = c;
    p.cin2 = c_in2; p.gate = gate; p.ldcin2 = ldcin2; p.ldgate = ldgate;
    p.lda = lda; p.ldc = ldc; p.ldcin = ldcin; p.plane = (int64_t)N * K;
    p.M = M; p.N = N; p.K = K; p.a2_rows = a2_rows; p.relu = relu;
    const int tiles_m = (M + kBM - 1) / kBM;
    // 96-wide column tiles when they waste fewer columns (e.g. N = 288 = 3 x 96)
    const int waste128 = ((N + 127) / 128) * 128 - N, waste96 = ((N + 95) / 96) * 96 - N;
    const bool use96 = waste96 < waste128;
    p.tiles_n = use96 ? (N + 95) / 96 : (N + 127) / 128;
    p.ntiles = tiles_m * p.tiles_n;
    const int grid = ((p.ntiles + 7) / 8) * 8;
    mpf::prof_begin(st);
    if (use96) {
        mpf::set_kernel("gemm3_tn_kernel<96>");
        hipLaunchKernelGGL(gemm3_tn_kernel<96>, dim3(grid), dim3(kThreads), 0, st, p);
    } else {
        mpf::set_kernel("gemm3_tn_kernel<128>");
        hipLaunchKernelGGL(gemm3_tn_kernel<128>, dim3(grid), dim3(kThreads), 0, st, p);
    }
    mpf::prof_end(mpf_last_kernel(), st, 4.0 * ((double)M * K + (double)M * N) + 6.0 * (double)N * K);
    return mpf::check(hipGetLastError(), "mpf_gemm3_tn");
}
