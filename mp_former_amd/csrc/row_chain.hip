// Row-local chains of the transformer decoder's query side (a few hundred rows of 256 channels): consecutive steps that need
// nothing but their own rows run inside ONE workgroup per 16 rows instead of one launch per step —
//   * output projection + residual + LayerNorm (mask2former_transformer_decoder.py:42-52, :100-112: tgt = norm(tgt + out_proj(..)))
//   * decoder_norm + the three Linear layers of mask_embed (:1859-1866, MLP :190-206)
// Every step of these chains was a 4-5 us launch doing < 1 us of work (a kernel boundary costs the drain of the previous kernel,
// the dispatch and one cold round trip to L2 per operand); inside a workgroup the hand-over is an LDS barrier.
// A workgroup = 8 waves; wave w owns output columns [32 w, 32 w + 32) of the 16 rows over the whole 256-deep contraction
// (v_mfma_f32_16x16x32_bf16, issued as D^T = W . A^T so that a lane owns 4 consecutive columns of one row); the 128 KB of a
// weight matrix come from L2 (every workgroup reads the same bytes), 32 fragments per lane requested together: one latency per GEMM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

constexpr int kE = 256;
constexpr int kNT = 2;             // 16-column tiles per wave
constexpr int kNW = 16 / kNT;       // waves per workgroup
constexpr int kRW = 16 / kNW;       // rows per wave in the LayerNorm phases
constexpr int kRowB = 528;          // bytes of a 256-channel bf16 row in LDS (+16: the 16 rows of a fragment read hit 16 x 4 different banks)

union Frag {
    uint4 q;
    bf16x8 v;
};

__device__ __forceinline__ unsigned f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// the wave's 4 x 8 weight fragments: rows (= output columns) 64 wv + 16 n + li, contraction elements 32 s + 8 g .. + 7
__device__ __forceinline__ void load_w(const u16* __restrict__ w, int wv, int li, int g, Frag (&fb)[kNT][8])
{
#pragma unroll
    for (int n = 0; n < kNT; ++n) {
        const u16* row = w + (int64_t)(16 * kNT * wv + 16 * n + li) * kE + 8 * g;
#pragma unroll
        for (int s = 0; s < 8; ++s) fb[n][s].q = *reinterpret_cast<const uint4*>(row + 32 * s);
    }
}

// The 256-deep contraction in the order of small_gemm_kernel<1> (small_gemm.hip: wave w of its workgroup takes the 32-deep steps
// w and w + 4, the four partial tiles are then summed as ((p0 + p1) + p2) + p3): the results are bit-identical to the launches
// this file replaces, so the sign of a mask logit near zero — which decides an attention-mask bit — does not depend on the route.
__device__ __forceinline__ void mma(const Frag (&fa)[8], const Frag (&fb)[kNT][8], f32x4 (&acc)[kNT])
{
#pragma unroll
    for (int n = 0; n < kNT; ++n) {
        f32x4 part[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            part[w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[n][w].v, fa[w].v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            part[w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[n][w + 4].v, fa[w + 4].v, part[w], 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n][e] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
    }
}

// acc[n][e] (+ bias) of row li, column 64 wv + 16 n + 4 g + e
__device__ __forceinline__ void add_bias(f32x4 (&acc)[kNT], const u16* __restrict__ bias, int wv, int g)
{
#pragma unroll
    for (int n = 0; n < kNT; ++n) {
        const uint2 bb = *reinterpret_cast<const uint2*>(bias + 16 * kNT * wv + 16 * n + 4 * g);
        acc[n][0] += bf_lo(bb.x); acc[n][1] += bf_hi(bb.x); acc[n][2] += bf_lo(bb.y); acc[n][3] += bf_hi(bb.y);
    }
}

struct LinLn {
    const u16* a;              // [R][256] bf16: the attention output
    const u16* w;              // [256][256] bf16 (out_proj.weight)
    const u16* bias;           // [256] bf16
    const float* x;            // [R][256] fp32 residual
    const float* gamma;
    const float* beta;
    float* s_out;              // x + bf16(a W^T + b)  (what the LayerNorm backward reads)
    float* y32;                // may be NULL
    u16* y16;                  // may be NULL
    float* mean;
    float* rstd;
    int R;
    float eps;
};

// tgt = LayerNorm(tgt + out_proj(attention output)): the projection's result is rounded to bf16 before the residual add, as the
// separate kernels (and the reference's autocast Linear) round it
__global__ __launch_bounds__(64 * kNW) void lin256_res_ln_kernel(const LinLn p)
{
    __shared__ __attribute__((aligned(16))) float tbuf[16][kE];        // bf16(a W^T + b) of the 16 rows, as fp32
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    const int r0 = (int)blockIdx.x * 16;
    const int ic = min(r0 + li, p.R - 1);
    Frag fa[8], fb[kNT][8];
    {
        const u16* arow = p.a + (int64_t)ic * kE + 8 * g;
#pragma unroll
        for (int s = 0; s < 8; ++s) fa[s].q = *reinterpret_cast<const uint4*>(arow + 32 * s);
    }
    load_w(p.w, wv, li, g, fb);
    // the LayerNorm phase is res_ln256_fwd_kernel's (elementwise.hip): one wave per row, lane l owns channels 4 l .. 4 l + 3
    const int c = lane * 4;
    float4 xv[kRW];
#pragma unroll
    for (int k = 0; k < kRW; ++k) xv[k] = *reinterpret_cast<const float4*>(p.x + (int64_t)min(r0 + kRW * wv + k, p.R - 1) * kE + c);
    const float4 gm = *reinterpret_cast<const float4*>(p.gamma + c), bt = *reinterpret_cast<const float4*>(p.beta + c);
    f32x4 acc[kNT];
    mma(fa, fb, acc);
    add_bias(acc, p.bias, wv, g);
#pragma unroll
    for (int n = 0; n < kNT; ++n)
        *reinterpret_cast<float4*>(&tbuf[li][16 * kNT * wv + 16 * n + 4 * g]) =
            make_float4(__uint_as_float(f2bf(acc[n][0]) << 16), __uint_as_float(f2bf(acc[n][1]) << 16), __uint_as_float(f2bf(acc[n][2]) << 16),
                        __uint_as_float(f2bf(acc[n][3]) << 16));
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRW; ++k) {
        const int row = r0 + kRW * wv + k;
        if (row >= p.R) break;
        const float4 u = *reinterpret_cast<const float4*>(&tbuf[kRW * wv + k][c]);
        const float4 v = make_float4(xv[k].x + u.x, xv[k].y + u.y, xv[k].z + u.z, xv[k].w + u.w);
        *reinterpret_cast<float4*>(p.s_out + (int64_t)row * kE + c) = v;
        const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
        const float4 d = make_float4(v.x - mu, v.y - mu, v.z - mu, v.w - mu);
        const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.f / 256.f);
        const float rs = rsqrtf(var + p.eps);
        const float4 o = make_float4(d.x * rs * gm.x + bt.x, d.y * rs * gm.y + bt.y, d.z * rs * gm.z + bt.z, d.w * rs * gm.w + bt.w);
        if (p.y32) *reinterpret_cast<float4*>(p.y32 + (int64_t)row * kE + c) = o;
        if (p.y16) {
            uint2 o16;
            o16.x = f2bf(o.x) | (f2bf(o.y) << 16);
            o16.y = f2bf(o.z) | (f2bf(o.w) << 16);
            *reinterpret_cast<uint2*>(p.y16 + (int64_t)row * kE + c) = o16;
        }
        if (lane == 0) { p.mean[row] = mu; p.rstd[row] = rs; }
    }
}

struct LnMlp3 {
    const float* x;            // [R][256] fp32
    const float* gamma;
    const float* beta;
    const u16 *w0, *b0, *w1, *b1, *w2, *b2;
    u16* out;                  // [R][256] bf16
    int R;
    float eps;
};

__device__ __forceinline__ void read_a(const unsigned char* buf, int li, int g, Frag (&fa)[8])
{
#pragma unroll
    for (int s = 0; s < 8; ++s) fa[s].q = *reinterpret_cast<const uint4*>(buf + li * kRowB + (32 * s + 8 * g) * 2);
}

// bf16(relu(acc)) of the wave's 64 columns -> the LDS row image of the next layer's operand
__device__ __forceinline__ void put_a(unsigned char* buf, int li, int g, int wv, const f32x4 (&acc)[kNT])
{
#pragma unroll
    for (int n = 0; n < kNT; ++n) {
        uint2 o;
        o.x = f2bf(fmaxf(acc[n][0], 0.f)) | (f2bf(fmaxf(acc[n][1], 0.f)) << 16);
        o.y = f2bf(fmaxf(acc[n][2], 0.f)) | (f2bf(fmaxf(acc[n][3], 0.f)) << 16);
        *reinterpret_cast<uint2*>(buf + li * kRowB + (16 * kNT * wv + 16 * n + 4 * g) * 2) = o;
    }
}

// mask_embed(decoder_norm(x)): LayerNorm (the arithmetic of res_ln256_fwd_kernel: one wave per row) -> Linear + ReLU -> Linear + ReLU
// -> Linear; every intermediate is rounded to bf16 where the separate kernels stored it
__global__ __launch_bounds__(64 * kNW) void ln_mlp3_kernel(const LnMlp3 p)
{
    __shared__ __attribute__((aligned(16))) unsigned char abuf[2][16 * kRowB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    const int r0 = (int)blockIdx.x * 16;
    Frag fa[8], fb[kNT][8];
    load_w(p.w0, wv, li, g, fb);
    {
        const int c = lane * 4;
        const float4 gm = *reinterpret_cast<const float4*>(p.gamma + c), bt = *reinterpret_cast<const float4*>(p.beta + c);
        float4 xv[kRW];
#pragma unroll
        for (int k = 0; k < kRW; ++k) xv[k] = *reinterpret_cast<const float4*>(p.x + (int64_t)min(r0 + kRW * wv + k, p.R - 1) * kE + c);
#pragma unroll
        for (int k = 0; k < kRW; ++k) {
            const float4 v = xv[k];
            const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
            const float4 d = make_float4(v.x - mu, v.y - mu, v.z - mu, v.w - mu);
            const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.f / 256.f);
            const float rs = rsqrtf(var + p.eps);
            uint2 o;
            o.x = f2bf(d.x * rs * gm.x + bt.x) | (f2bf(d.y * rs * gm.y + bt.y) << 16);
            o.y = f2bf(d.z * rs * gm.z + bt.z) | (f2bf(d.w * rs * gm.w + bt.w) << 16);
            *reinterpret_cast<uint2*>(abuf[0] + (kRW * wv + k) * kRowB + c * 2) = o;
        }
    }
    __syncthreads();
    f32x4 acc[kNT];
    read_a(abuf[0], li, g, fa);
    mma(fa, fb, acc);
    load_w(p.w1, wv, li, g, fb);              // (requested before this layer's epilogue and barrier)
    add_bias(acc, p.b0, wv, g);
    put_a(abuf[1], li, g, wv, acc);
    __syncthreads();
    read_a(abuf[1], li, g, fa);
    mma(fa, fb, acc);
    load_w(p.w2, wv, li, g, fb);
    add_bias(acc, p.b1, wv, g);
    put_a(abuf[0], li, g, wv, acc);
    __syncthreads();
    read_a(abuf[0], li, g, fa);
    mma(fa, fb, acc);
    add_bias(acc, p.b2, wv, g);
    const int i = r0 + li;
    if (i >= p.R) return;
#pragma unroll
    for (int n = 0; n < kNT; ++n) {
        uint2 o;
        o.x = f2bf(acc[n][0]) | (f2bf(acc[n][1]) << 16);
        o.y = f2bf(acc[n][2]) | (f2bf(acc[n][3]) << 16);
        *reinterpret_cast<uint2*>(p.out + (int64_t)i * kE + 16 * kNT * wv + 16 * n + 4 * g) = o;
    }
}

bool misaligned16(const void* q) { return ((uintptr_t)q & 15) != 0; }

}  // namespace

extern "C" int mpf_lin256_res_ln_forward(const void* a, const void* w, const void* bias, const float* x, const float* gamma, const float* beta,
                                         float* s_out, float* y32, void* y16, float* mean, float* rstd, int rows, float eps, void* stream)
{
    if (rows == 0) return 0;
    if (!a || !w || !bias || !x || !gamma || !beta || !s_out || !mean || !rstd || (!y32 && !y16))
        return mpf::fail(MPF_E_NULL, "lin256_res_ln_forward: NULL buffer");
    if (rows < 0) return mpf::fail(MPF_E_SHAPE, "lin256_res_ln_forward: bad rows");
    if (misaligned16(a) || misaligned16(w) || misaligned16(x) || misaligned16(gamma) || misaligned16(beta) || misaligned16(s_out) ||
        (y32 && misaligned16(y32)) || (y16 && ((uintptr_t)y16 & 7)) || ((uintptr_t)bias & 7))
        return mpf::fail(MPF_E_SHAPE, "lin256_res_ln_forward: operands must be 16-byte aligned (bias, y16: 8)");
    LinLn p{static_cast<const u16*>(a), static_cast<const u16*>(w), static_cast<const u16*>(bias), x, gamma, beta, s_out, y32,
            static_cast<u16*>(y16), mean, rstd, rows, eps};
    mpf::prof_begin((hipStream_t)stream);
    mpf::set_kernel("lin256_res_ln_kernel");
    hipLaunchKernelGGL(lin256_res_ln_kernel, dim3((rows + 15) / 16), dim3(64 * kNW), 0, (hipStream_t)stream, p);
    mpf::prof_end("lin256_res_ln_kernel", (hipStream_t)stream, 2.0 * kE * kE + (double)rows * kE * (2 + 4 + 4 + (y32 ? 4 : 0) + (y16 ? 2 : 0)));
    return mpf::check(hipGetLastError(), "mpf_lin256_res_ln_forward");
}

extern "C" int mpf_ln256_mlp3_forward(const float* x, const float* gamma, const float* beta, const void* w0, const void* b0, const void* w1,
                                      const void* b1, const void* w2, const void* b2, void* out, int rows, float eps, void* stream)
{
    if (rows == 0) return 0;
    if (!x || !gamma || !beta || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !out) return mpf::fail(MPF_E_NULL, "ln256_mlp3_forward: NULL buffer");
    if (rows < 0) return mpf::fail(MPF_E_SHAPE, "ln256_mlp3_forward: bad rows");
    if (misaligned16(x) || misaligned16(gamma) || misaligned16(beta) || misaligned16(w0) || misaligned16(w1) || misaligned16(w2) ||
        (((uintptr_t)b0 | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)out) & 7))
        return mpf::fail(MPF_E_SHAPE, "ln256_mlp3_forward: operands must be 16-byte aligned (biases, out: 8)");
    LnMlp3 p{x, gamma, beta, static_cast<const u16*>(w0), static_cast<const u16*>(b0), static_cast<const u16*>(w1),
             static_cast<const u16*>(b1), static_cast<const u16*>(w2), static_cast<const u16*>(b2), static_cast<u16*>(out), rows, eps};
    mpf::prof_begin((hipStream_t)stream);
    mpf::set_kernel("ln_mlp3_kernel");
    hipLaunchKernelGGL(ln_mlp3_kernel, dim3((rows + 15) / 16), dim3(64 * kNW), 0, (hipStream_t)stream, p);
    mpf::prof_end("ln_mlp3_kernel", (hipStream_t)stream, 6.0 * kE * kE + (double)rows * kE * (4 + 2));
    return mpf::check(hipGetLastError(), "mpf_ln256_mlp3_forward");
}
