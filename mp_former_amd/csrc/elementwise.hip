// Fused per-channel bias (+ residual) + ReLU epilogue for channel-last activations (the folded
// FrozenBN shift of the bench backbone): y = relu(x + bias[c] + res), one pass instead of three
// (MIOpen's separate bias kernel, the residual add, the ReLU).  bf16 or fp32 activations, fp32 bias.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpf_common.h"
#include "amax.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <bool RES>
__global__ __launch_bounds__(256) void bias_act_bf16_kernel(const bf16x8* __restrict__ x, const float* __restrict__ bias,
                                                            const bf16x8* __restrict__ res, bf16x8* __restrict__ y,
                                                            int64_t n8, int C, int relu)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 v = x[i];
        const int c0 = (int)((i * 8) % C);
        const float4 b0 = *reinterpret_cast<const float4*>(bias + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        bf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (RES) r = res[i];            // (compile-time: a load behind `if (res)` is waited for on its own)
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // same rounding points as the unfused ops: (x + b) -> bf16, (+ res) -> bf16
            float t = (float)(__bf16)((float)v[j] + b[j]);
            if constexpr (RES) t = (float)(__bf16)(t + (float)r[j]);
            o[j] = (__bf16)(relu ? fmaxf(t, 0.f) : t);
        }
        y[i] = o;
    }
}

__global__ __launch_bounds__(256) void bias_act_f32_kernel(const float4* __restrict__ x, const float* __restrict__ bias,
                                                           const float4* __restrict__ res, float4* __restrict__ y,
                                                           int64_t n4, int C, int relu)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = x[i];
        const float4 b = *reinterpret_cast<const float4*>(bias + (int)((i * 4) % C));
        v = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
        if (res) {
            const float4 r = res[i];
            v = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
        }
        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        y[i] = v;
    }
}

// backward of the block-output ReLU of a residual block with the sum of the two gradients that reach it (next block's
// conv1 / shortcut input gradient and its identity-skip gradient) folded in: out = y > 0 ? bf16(ga + gb) : 0 — the rounding
// points of aten's add followed by threshold_backward, one pass instead of two
template <bool ADD>
__global__ __launch_bounds__(256) void relu_bwd_add_bf16_kernel(const bf16x8* __restrict__ ga, const bf16x8* __restrict__ gb,
                                                                const bf16x8* __restrict__ y, bf16x8* __restrict__ out, int64_t n8)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 a = ga[i], yy = y[i];
        bf16x8 b = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (ADD) b = gb[i];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 sum = ADD ? (__bf16)((float)a[j] + (float)b[j]) : a[j];
            o[j] = (float)yy[j] > 0.f ? sum : (__bf16)0.f;
        }
        out[i] = o;
    }
}

}  // namespace

namespace {
// 3 x 3 / stride 2 / padding 1 max pooling of the ResNet stem on a channel-last bf16 activation (torch.nn.functional.max_pool2d's
// semantics: the FIRST maximum of the window in (kh, kw) scan order wins, NaN propagates).  The forward keeps the winner's
// window position (0..8) as one byte per output element; the backward is a gather over the <= 4 windows that contain an input
// pixel (no atomics, fp32 sum in aten's (oh, ow) order).  8 channels (16 B) per thread.
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ y,
                                                               uint2* __restrict__ code, int N, int H, int W, int C8, int OH, int OW)
{
    // block = (32 * 8 / C8... ) output pixels of one output row: thread -> (pixel, 8-channel group), no 64-bit divisions
    const int per = 256 / C8;                      // output pixels per block (C8 divides 256: checked by the caller)
    const int c = threadIdx.x % C8, ow = blockIdx.x * per + threadIdx.x / C8;
    const int oh = blockIdx.y, n = blockIdx.z;
    if (ow < OW) {
        const int64_t i = (((int64_t)n * OH + oh) * OW + ow) * C8 + c;
        // aten: maxval = -inf, maxidx = the window's first in-range position, then `val > maxval || isnan(val)` in scan order.
        // All nine loads are issued unconditionally on clamped coordinates (a load behind a branch is waited for on its own)
        // and positions outside the image are masked out of the update.
        float best[8];
        unsigned arg[8];
        const unsigned first = (unsigned)((oh == 0 ? 3 : 0) + (ow == 0 ? 1 : 0));
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = first; }
        bf16x8 v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int h = min(max(2 * oh - 1 + k / 3, 0), H - 1), w = min(max(2 * ow - 1 + k % 3, 0), W - 1);
            v[k] = x[(((int64_t)n * H + h) * W + w) * C8 + c];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int h = 2 * oh - 1 + k / 3, w = 2 * ow - 1 + k % 3;
            const bool valid = h >= 0 && h < H && w >= 0 && w < W;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = (float)v[k][j];
                const bool upd = valid && (f > best[j] || f != f);
                best[j] = upd ? f : best[j];
                arg[j] = upd ? (unsigned)k : arg[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)best[j];
        y[i] = o;
        code[i] = make_uint2(arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((unsigned)arg[3] << 24),
                             arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((unsigned)arg[7] << 24));
    }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const bf16x8* __restrict__ gy, const uint2* __restrict__ code,
                                                               bf16x8* __restrict__ gx, int N, int H, int W, int C8, int OH, int OW)
{
    const int per = 256 / C8;
    const int c = threadIdx.x % C8, w = blockIdx.x * per + threadIdx.x / C8;
    const int h = blockIdx.y, n = blockIdx.z;
    if (w < W) {
        const int64_t i = (((int64_t)n * H + h) * W + w) * C8 + c;
        // windows oh with 2 oh - 1 <= h <= 2 oh + 1 (and likewise ow): h / 2 and, for odd h, (h + 1) / 2 — requested together
        // on clamped indices, the absent ones masked (aten's accumulation order: oh outer, ow inner, fp32)
        const int ohs[2] = {h >> 1, min((h + 1) >> 1, OH - 1)}, ows[2] = {w >> 1, min((w + 1) >> 1, OW - 1)};
        const bool vh[2] = {(h >> 1) < OH, (h & 1) && ((h + 1) >> 1) < OH}, vw[2] = {(w >> 1) < OW, (w & 1) && ((w + 1) >> 1) < OW};
        uint2 cd[4];
        bf16x8 g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t o = (((int64_t)n * OH + min(ohs[k >> 1], OH - 1)) * OW + min(ows[k & 1], OW - 1)) * C8 + c;
            cd[k] = code[o];
            g[k] = gy[o];
        }
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oh = ohs[k >> 1], ow = ows[k & 1];
            const bool valid = vh[k >> 1] && vw[k & 1];
            const unsigned me = (unsigned)((h - (2 * oh - 1)) * 3 + (w - (2 * ow - 1)));      // this pixel's position in that window
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned a = ((j < 4 ? cd[k].x : cd[k].y) >> (8 * (j & 3))) & 0xFFu;
                if (valid && a == me) acc[j] += (float)g[k][j];
            }
        }
        bf16x8 o8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = (__bf16)acc[j];
        gx[i] = o8;
    }
}
}  // namespace

extern "C" int mpf_maxpool3x3s2_forward(const void* x, void* y, void* code, int N, int H, int W, int C, void* stream)
{
    if (!x || !y || !code) return mpf::fail(MPF_E_NULL, "maxpool3x3s2_forward: NULL buffer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return mpf::fail(MPF_E_SHAPE, "maxpool3x3s2_forward: C must be a positive multiple of 8");
    if (((uintptr_t)x | (uintptr_t)y) & 15 || ((uintptr_t)code & 7)) return mpf::fail(MPF_E_SHAPE, "maxpool3x3s2_forward: 16-byte aligned buffers");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    if (256 % (C / 8) || OH > 65535 || N > 65535) return mpf::fail(MPF_E_SHAPE, "maxpool3x3s2_forward: C / 8 must divide 256; OH, N <= 65535");
    const int per = 256 / (C / 8);
    mpf::set_kernel("maxpool3x3s2_fwd_kernel");
    hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3((OW + per - 1) / per, OH, N), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, (bf16x8*)y, (uint2*)code,
                       N, H, W, C / 8, OH, OW);
    return mpf::check(hipGetLastError(), "mpf_maxpool3x3s2_forward");
}

extern "C" int mpf_maxpool3x3s2_backward(const void* gy, const void* code, void* gx, int N, int H, int W, int C, void* stream)
{
    if (!gy || !gx || !code) return mpf::fail(MPF_E_NULL, "maxpool3x3s2_backward: NULL buffer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return mpf::fail(MPF_E_SHAPE, "maxpool3x3s2_backward: C must be a positive multiple of 8");
    if (((uintptr_t)gx | (uintptr_t)gy) & 15 || ((uintptr_t)code & 7)) return mpf::fail(MPF_E_SHAPE, "maxpool3x3s2_backward: 16-byte aligned buffers");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    if (256 % (C / 8) || H > 65535 || N > 65535) return mpf::fail(MPF_E_SHAPE, "maxpool3x3s2_backward: C / 8 must divide 256; H, N <= 65535");
    const int per = 256 / (C / 8);
    mpf::set_kernel("maxpool3x3s2_bwd_kernel");
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3((W + per - 1) / per, H, N), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)gy, (const uint2*)code,
                       (bf16x8*)gx, N, H, W, C / 8, OH, OW);
    return mpf::check(hipGetLastError(), "mpf_maxpool3x3s2_backward");
}

extern "C" int mpf_relu_bwd_add(const void* ga, const void* gb, const void* y, void* out, int64_t numel, int dtype, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!ga || !y || !out) return mpf::fail(MPF_E_NULL, "relu_bwd_add: NULL buffer");
    if (numel <= 0 || numel % 8 != 0) return mpf::fail(MPF_E_SHAPE, "relu_bwd_add: numel must be a positive multiple of 8");
    if (dtype != MPF_BF16) return mpf::fail(MPF_E_DTYPE, "relu_bwd_add: dtype must be MPF_BF16");
    if (((uintptr_t)ga | (uintptr_t)gb | (uintptr_t)y | (uintptr_t)out) & 15) return mpf::fail(MPF_E_SHAPE, "relu_bwd_add: 16-byte aligned buffers");
    const int64_t n8 = numel / 8;
    const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    mpf::set_kernel("relu_bwd_add_bf16_kernel");
    if (gb)
        hipLaunchKernelGGL(relu_bwd_add_bf16_kernel<true>, dim3(blocks), dim3(256), 0, st, (const bf16x8*)ga, (const bf16x8*)gb,
                           (const bf16x8*)y, (bf16x8*)out, n8);
    else
        hipLaunchKernelGGL(relu_bwd_add_bf16_kernel<false>, dim3(blocks), dim3(256), 0, st, (const bf16x8*)ga, (const bf16x8*)gb,
                           (const bf16x8*)y, (bf16x8*)out, n8);
    return mpf::check(hipGetLastError(), "mpf_relu_bwd_add");
}

extern "C" int mpf_bias_act(const void* x, const float* bias, const void* res, void* y, int64_t numel, int C, int dtype,
                            int relu, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!x || !bias || !y) return mpf::fail(MPF_E_NULL, "bias_act: NULL buffer");
    if (numel <= 0 || C <= 0 || C % 8 != 0 || numel % C != 0) return mpf::fail(MPF_E_SHAPE, "bias_act: C must be a multiple of 8 dividing numel");
    if (dtype == MPF_BF16) {
        const int64_t n8 = numel / 8;
        const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
        mpf::set_kernel("bias_act_bf16_kernel");
        if (res)
            hipLaunchKernelGGL(bias_act_bf16_kernel<true>, dim3(blocks), dim3(256), 0, st, (const bf16x8*)x, bias, (const bf16x8*)res,
                               (bf16x8*)y, n8, C, relu);
        else
            hipLaunchKernelGGL(bias_act_bf16_kernel<false>, dim3(blocks), dim3(256), 0, st, (const bf16x8*)x, bias, (const bf16x8*)res,
                               (bf16x8*)y, n8, C, relu);
    } else if (dtype == MPF_F32) {
        const int64_t n4 = numel / 4;
        const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
        mpf::set_kernel("bias_act_f32_kernel");
        hipLaunchKernelGGL(bias_act_f32_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)x, bias, (const float4*)res,
                           (float4*)y, n4, C, relu);
    } else {
        return mpf::fail(MPF_E_DTYPE, "bias_act: dtype must be MPF_F32 or MPF_BF16");
    }
    return mpf::check(hipGetLastError(), "mpf_bias_act");
}

// ------------------------------------------------------------------------------------------------
// Post-norm residual block of the decoder (mask2former_transformer_decoder.py:42-52, :100-112,
// :165-169: tgt = LayerNorm(tgt + tgt2)), C = 256:  s = x + t (x fp32 residual stream, t bf16 or fp32
// branch output, may be NULL), y = LN(s) * gamma + beta, written as fp32 (next residual) and/or bf16
// (next GEMM operand) in ONE pass; mean / rstd saved for the backward.  One wave per row, 4 columns
// per lane.  Backward: g = gy32 + gy16 -> ds (fp32 and/or bf16 copies) and per-block partial sums of
// dgamma / dbeta accumulated with float atomics into zero-initialised vectors.
// ------------------------------------------------------------------------------------------------
namespace {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename TT>
__device__ __forceinline__ float4 load4(const TT* p)
{
    if constexpr (sizeof(TT) == 4) {
        return *reinterpret_cast<const float4*>(p);
    } else {
        const bf16x4v v = *reinterpret_cast<const bf16x4v*>(p);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
}

// HAS_T / PLUS are compile-time: as run-time pointer tests every optional load sat behind its own branch and was waited
// for on its own (x, then t, then — after both reductions — the positional addend: three dependent round trips per row)
template <typename TT, bool HAS_T, bool PLUS>
__global__ __launch_bounds__(256) void res_ln256_fwd_kernel(const float* __restrict__ x, const TT* __restrict__ t,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ s_out, float* __restrict__ y32, __bf16* __restrict__ y16,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int rows, float eps,
                                                            const float* __restrict__ padd, int padd_rows, float* __restrict__ y_plus,
                                                            float* __restrict__ y_bound, const float* __restrict__ padd_amax,
                                                            float* __restrict__ yplus_bound)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int c = (threadIdx.x & 63) * 4;
    // amax slots of the outputs as an UPPER BOUND from the parameters (no pass over the data, no atomics): a normalised row of
    // 256 values has |x^| <= sqrt(255) < 16, so |y| <= 16 max|gamma| + max|beta| (typically 4-5x the true maximum: two of the
    // 18 binades the fp16 x 2 split carries at full precision), and |y + padd| <= that + max|padd|.  Wave 0 of block 0 writes.
    if (y_bound && blockIdx.x == 0 && threadIdx.x < 64) {
        const float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
        float gm = fmaxf(fmaxf(fabsf(g4.x), fabsf(g4.y)), fmaxf(fabsf(g4.z), fabsf(g4.w)));
        float bm = fmaxf(fmaxf(fabsf(b4.x), fabsf(b4.y)), fmaxf(fabsf(b4.z), fabsf(b4.w)));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { gm = fmaxf(gm, __shfl_xor(gm, o)); bm = fmaxf(bm, __shfl_xor(bm, o)); }
        if (threadIdx.x == 0) {
            const float bound = 16.f * gm + bm;
            y_bound[0] = bound;
            if (yplus_bound) yplus_bound[0] = bound + __uint_as_float(amax_read(padd_amax));
        }
    }
    float4 v = *reinterpret_cast<const float4*>(x + (int64_t)row * 256 + c);
    float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (PLUS) pp = *reinterpret_cast<const float4*>(padd + (int64_t)(row % padd_rows) * 256 + c);
    if constexpr (HAS_T) {
        const float4 u = load4(t + (int64_t)row * 256 + c);
        v = make_float4(v.x + u.x, v.y + u.y, v.z + u.z, v.w + u.w);
    }
    if (s_out) *reinterpret_cast<float4*>(s_out + (int64_t)row * 256 + c) = v;
    const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float4 d = make_float4(v.x - mu, v.y - mu, v.z - mu, v.w - mu);
    const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.f / 256.f);
    const float rs = rsqrtf(var + eps);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    const float4 o = make_float4(d.x * rs * g.x + b.x, d.y * rs * g.y + b.y, d.z * rs * g.z + b.z, d.w * rs * g.w + b.w);
    if (y32) *reinterpret_cast<float4*>(y32 + (int64_t)row * 256 + c) = o;
    if (y16) *reinterpret_cast<bf16x4v*>(y16 + (int64_t)row * 256 + c) = bf16x4v{(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
    if constexpr (PLUS)   // y + positional term (row-periodic): the query input of the next deformable-attention layer
        *reinterpret_cast<float4*>(y_plus + (int64_t)row * 256 + c) = make_float4(o.x + pp.x, o.y + pp.y, o.z + pp.z, o.w + pp.w);
    if ((threadIdx.x & 63) == 0) { mean[row] = mu; rstd[row] = rs; }
}

// PART: the block's dgamma / dbeta sums go to partial[block][2][256] (summed in a fixed order by ln_partial_reduce_kernel)
// instead of float atomics into zero-initialised vectors: deterministic, no pre-zeroing, no contended atomics
// PART == 2: the same partials, summed in block order by the workgroup that ARRIVES LAST (ticket from one integer atomic per
// workgroup): one launch, no floating-point atomics, bit-reproducible.  dgamma = the workspace: [0] ticket counter (zero on
// entry, reset to zero on exit), partials from byte 256; dbeta = the output pair [2][256] (dgamma row, dbeta row).
template <bool G32, bool G16, bool GP, int PART = 0>
__global__ __launch_bounds__(256) void res_ln256_bwd_kernel(const float* __restrict__ s, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ gy32, const __bf16* __restrict__ gy16,
                                                            const float* __restrict__ gy_plus,
                                                            float* __restrict__ ds32, __bf16* __restrict__ ds16,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int rows_per_block,
                                                            float* __restrict__ ds_amax = nullptr)
{
    __shared__ float red[2][4][256];
    float omax = 0.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int row = r0 + wave; row < r1; row += 4) {
        // all operands of the row requested together (G32 / G16 / GP compile-time: no load behind a branch)
        float4 dy = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 sv = *reinterpret_cast<const float4*>(s + (int64_t)row * 256 + c);
        const float mu = mean[row], rs = rstd[row];
        if constexpr (G32) dy = *reinterpret_cast<const float4*>(gy32 + (int64_t)row * 256 + c);
        if constexpr (G16) {
            const float4 u = load4(gy16 + (int64_t)row * 256 + c);
            dy = make_float4(dy.x + u.x, dy.y + u.y, dy.z + u.z, dy.w + u.w);
        }
        if constexpr (GP) {
            const float4 u = *reinterpret_cast<const float4*>(gy_plus + (int64_t)row * 256 + c);
            dy = make_float4(dy.x + u.x, dy.y + u.y, dy.z + u.z, dy.w + u.w);
        }
        const float4 xh = make_float4((sv.x - mu) * rs, (sv.y - mu) * rs, (sv.z - mu) * rs, (sv.w - mu) * rs);
        const float4 dg = make_float4(dy.x * g.x, dy.y * g.y, dy.z * g.z, dy.w * g.w);
        const float s1 = wave_sum(dg.x + dg.y + dg.z + dg.w) * (1.f / 256.f);
        const float s2 = wave_sum(dg.x * xh.x + dg.y * xh.y + dg.z * xh.z + dg.w * xh.w) * (1.f / 256.f);
        const float4 o = make_float4(rs * (dg.x - s1 - xh.x * s2), rs * (dg.y - s1 - xh.y * s2), rs * (dg.z - s1 - xh.z * s2),
                                     rs * (dg.w - s1 - xh.w * s2));
        if (ds32) *reinterpret_cast<float4*>(ds32 + (int64_t)row * 256 + c) = o;
        if (ds16) *reinterpret_cast<bf16x4v*>(ds16 + (int64_t)row * 256 + c) = bf16x4v{(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        ag = make_float4(ag.x + dy.x * xh.x, ag.y + dy.y * xh.y, ag.z + dy.z * xh.z, ag.w + dy.w * xh.w);
        ab = make_float4(ab.x + dy.x, ab.y + dy.y, ab.z + dy.z, ab.w + dy.w);
    }
    if (ds_amax) {      // (uniform) the largest |ds| of this block's rows -> one atomic max (amax.h)
        __shared__ float ared[4];
        amax_commit(ds_amax, omax, ared);
    }
    *reinterpret_cast<float4*>(&red[0][wave][c]) = ag;
    *reinterpret_cast<float4*>(&red[1][wave][c]) = ab;
    __syncthreads();
    const int col = threadIdx.x;
    const float sg = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
    const float sb = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
    if constexpr (PART == 1) {       // dgamma doubles as the partial buffer pointer
        dgamma[(int64_t)blockIdx.x * 512 + col] = sg;
        dgamma[(int64_t)blockIdx.x * 512 + 256 + col] = sb;
    } else if constexpr (PART == 2) {
        float* part = dgamma + 64;
        part[(int64_t)blockIdx.x * 512 + col] = sg;
        part[(int64_t)blockIdx.x * 512 + 256 + col] = sb;
        __threadfence();                                      // partials visible device-wide before the ticket
        __shared__ int s_last;
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(reinterpret_cast<int*>(dgamma), 1) == (int)gridDim.x - 1;
        __syncthreads();
        if (s_last) {
            // 128 float4 columns of [dgamma | dbeta] x 2 slices of the block list, 16 independent loads in flight per thread
            // (a `volatile` walk over the blocks serialised one L2 round trip per block: 35 us for 58 blocks); the lines were
            // never read by this CU in this launch, so plain loads after the fence see the other workgroups' writes
            __threadfence();
            const int vec = threadIdx.x & 127, slice = threadIdx.x >> 7, nb = (int)gridDim.x;
            const float4* pv = reinterpret_cast<const float4*>(part) + vec;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int b = slice;
            for (; b + 30 < nb; b += 32) {
                float4 t[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) t[k] = pv[(int64_t)(b + 2 * k) * 128];
#pragma unroll
                for (int k = 0; k < 16; ++k) { acc.x += t[k].x; acc.y += t[k].y; acc.z += t[k].z; acc.w += t[k].w; }
            }
            for (; b < nb; b += 2) { const float4 t = pv[(int64_t)b * 128]; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
            float4* sm = reinterpret_cast<float4*>(&red[0][0][0]);
            __syncthreads();
            if (slice == 1) sm[vec] = acc;
            __syncthreads();
            if (slice == 0) {
                const float4 o = sm[vec];
                reinterpret_cast<float4*>(dbeta)[vec] = make_float4(acc.x + o.x, acc.y + o.y, acc.z + o.z, acc.w + o.w);
            }
            if (threadIdx.x == 0) *reinterpret_cast<int*>(dgamma) = 0;
        }
    } else {
        atomicAdd(dgamma + col, sg);
        atomicAdd(dbeta + col, sb);
    }
}

// out[which][col] = sum over blocks of partial[block][which][col] in a fixed order; grid = (2 "which", 8 groups of 32 columns),
// 1024 threads = 32 columns x 32 block slices (a wave reads two 128-byte row pieces), slices merged through LDS.  (Two
// workgroups walking all blocks were latency-bound: 245 dependent-ish reads per thread, 29 us for 2 MB.)
// blockIdx.z = which LayerNorm of a group (partials `zstride` floats apart, outputs 512 floats apart: dgamma row, dbeta row)
__global__ __launch_bounds__(1024) void ln_partial_reduce_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int64_t zstride = 0)
{
    __shared__ float red[32][33];
    const int which = blockIdx.x, c = threadIdx.x & 31, col = blockIdx.y * 32 + c, sl = threadIdx.x >> 5;
    partial += (int64_t)blockIdx.z * zstride;
    dgamma += (int64_t)blockIdx.z * 512;
    dbeta += (int64_t)blockIdx.z * 512;
    const float* src = partial + which * 256 + col;
    float acc = 0.f;
    int b = sl;
    for (; b + 224 < blocks; b += 256) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = src[(int64_t)(b + 32 * k) * 512];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += t[k];
    }
    for (; b < blocks; b += 32) acc += src[(int64_t)b * 512];
    red[sl][c] = acc;
    __syncthreads();
    if (sl == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][c];
        (which ? dbeta : dgamma)[col] = t;
    }
}

}  // namespace

static int res_ln256_forward_impl(const float* x, const void* t, int t_dtype, const float* gamma, const float* beta,
                                  float* s_out, float* y32, void* y16, float* mean, float* rstd, int rows, float eps,
                                  const float* padd, int padd_rows, float* y_plus, void* stream, float* y_bound,
                                  const float* padd_amax, float* yplus_bound);

extern "C" int mpf_res_ln256_forward(const float* x, const void* t, int t_dtype, const float* gamma, const float* beta,
                                     float* s_out, float* y32, void* y16, float* mean, float* rstd, int rows, float eps,
                                     const float* padd, int padd_rows, float* y_plus, void* stream)
{
    return res_ln256_forward_impl(x, t, t_dtype, gamma, beta, s_out, y32, y16, mean, rstd, rows, eps, padd, padd_rows, y_plus, stream,
                                  nullptr, nullptr, nullptr);
}

extern "C" int mpf_res_ln256_forward_b(const float* x, const void* t, int t_dtype, const float* gamma, const float* beta,
                                       float* s_out, float* y32, void* y16, float* mean, float* rstd, int rows, float eps,
                                       const float* padd, int padd_rows, float* y_plus, float* y_bound, const float* padd_amax,
                                       float* yplus_bound, void* stream)
{
    if (!y_bound || (yplus_bound && (!padd_amax || !y_plus))) return mpf::fail(MPF_E_NULL, "res_ln256_forward_b: NULL amax slot");
    return res_ln256_forward_impl(x, t, t_dtype, gamma, beta, s_out, y32, y16, mean, rstd, rows, eps, padd, padd_rows, y_plus, stream,
                                  y_bound, padd_amax, yplus_bound);
}

static int res_ln256_forward_impl(const float* x, const void* t, int t_dtype, const float* gamma, const float* beta,
                                  float* s_out, float* y32, void* y16, float* mean, float* rstd, int rows, float eps,
                                  const float* padd, int padd_rows, float* y_plus, void* stream, float* y_bound,
                                  const float* padd_amax, float* yplus_bound)
{
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return 0;
    if (!x || !gamma || !beta || !mean || !rstd || (!y32 && !y16)) return mpf::fail(MPF_E_NULL, "res_ln256_forward: NULL buffer");
    if (rows < 0 || (y_plus && (!padd || padd_rows <= 0))) return mpf::fail(MPF_E_SHAPE, "res_ln256_forward: bad rows / missing addend");
    const dim3 grid((rows + 3) / 4);
    mpf::set_kernel("res_ln256_fwd_kernel");
#define RLN_FWD(TT, HT, PL)                                                                                                        \
    hipLaunchKernelGGL((res_ln256_fwd_kernel<TT, HT, PL>), grid, dim3(256), 0, st, x, (const TT*)t, gamma, beta, s_out, y32, (__bf16*)y16, \
                       mean, rstd, rows, eps, padd, padd_rows, y_plus, y_bound, padd_amax, yplus_bound)
    const bool plus = y_plus != nullptr;
    if (t && t_dtype == MPF_BF16) { if (plus) RLN_FWD(__bf16, true, true); else RLN_FWD(__bf16, true, false); }
    else if (t && t_dtype == MPF_F32) { if (plus) RLN_FWD(float, true, true); else RLN_FWD(float, true, false); }
    else if (!t) { if (plus) RLN_FWD(float, false, true); else RLN_FWD(float, false, false); }
    else
        return mpf::fail(MPF_E_DTYPE, "res_ln256_forward: t dtype must be MPF_F32 or MPF_BF16");
#undef RLN_FWD
    return mpf::check(hipGetLastError(), "mpf_res_ln256_forward");
}

extern "C" int mpf_res_ln256_backward(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                      const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma, float* dbeta,
                                      int rows, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return 0;
    if (!s || !mean || !rstd || !gamma || (!gy32 && !gy16 && !gy_plus) || (!ds32 && !ds16) || !dgamma || !dbeta)
        return mpf::fail(MPF_E_NULL, "res_ln256_backward: NULL buffer");
    if (rows < 0) return mpf::fail(MPF_E_SHAPE, "res_ln256_backward: bad rows");
    // ~1024 blocks at most; each block reduces its rows' dgamma / dbeta before the atomics
    int rpb = (rows + 1023) / 1024;
    rpb = ((rpb + 3) / 4) * 4;
    const dim3 grid((rows + rpb - 1) / rpb);
    mpf::set_kernel("res_ln256_bwd_kernel");
#define RLN_BWD(A, B, C)                                                                                                          \
    hipLaunchKernelGGL((res_ln256_bwd_kernel<A, B, C>), grid, dim3(256), 0, st, s, mean, rstd, gamma, gy32, (const __bf16*)gy16, gy_plus, \
                       ds32, (__bf16*)ds16, dgamma, dbeta, rows, rpb)
    switch ((gy32 ? 4 : 0) | (gy16 ? 2 : 0) | (gy_plus ? 1 : 0)) {
        case 1: RLN_BWD(false, false, true); break;
        case 2: RLN_BWD(false, true, false); break;
        case 3: RLN_BWD(false, true, true); break;
        case 4: RLN_BWD(true, false, false); break;
        case 5: RLN_BWD(true, false, true); break;
        case 6: RLN_BWD(true, true, false); break;
        default: RLN_BWD(true, true, true); break;
    }
#undef RLN_BWD
    return mpf::check(hipGetLastError(), "mpf_res_ln256_backward");
}

extern "C" size_t mpf_res_ln256_backward_workspace_bytes(int rows)
{
    if (rows <= 0) return 0;
    int rpb = (rows + 1023) / 1024;
    rpb = ((rpb + 3) / 4) * 4;
    return (size_t)((rows + rpb - 1) / rpb) * 512 * sizeof(float);
}

// the same backward with the parameter gradients reduced WITHOUT atomics (deterministic; dgamma / dbeta need no zeroing)
static int res_ln256_backward_ws_impl(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                      const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma, float* dbeta,
                                      int rows, void* workspace, size_t workspace_bytes, void* stream, float* ds_amax);

extern "C" int mpf_res_ln256_backward_ws(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                         const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma, float* dbeta,
                                         int rows, void* workspace, size_t workspace_bytes, void* stream)
{
    return res_ln256_backward_ws_impl(s, mean, rstd, gamma, gy32, gy16, gy_plus, ds32, ds16, dgamma, dbeta, rows, workspace, workspace_bytes,
                                      stream, nullptr);
}

// ... recording the largest |ds| in an amax slot (zeroed by the caller) for the fp16 x 2 GEMMs that consume ds
extern "C" int mpf_res_ln256_backward_ws_amax(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                              const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma, float* dbeta,
                                              int rows, void* workspace, size_t workspace_bytes, float* ds_amax, void* stream)
{
    if (!ds_amax) return mpf::fail(MPF_E_NULL, "res_ln256_backward_ws_amax: NULL amax slot");
    return res_ln256_backward_ws_impl(s, mean, rstd, gamma, gy32, gy16, gy_plus, ds32, ds16, dgamma, dbeta, rows, workspace, workspace_bytes,
                                      stream, ds_amax);
}

static int res_ln256_backward_ws_impl(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                      const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma, float* dbeta,
                                      int rows, void* workspace, size_t workspace_bytes, void* stream, float* ds_amax)
{
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return 0;
    if (!s || !mean || !rstd || !gamma || (!gy32 && !gy16 && !gy_plus) || (!ds32 && !ds16) || !dgamma || !dbeta || !workspace)
        return mpf::fail(MPF_E_NULL, "res_ln256_backward_ws: NULL buffer");
    if (rows < 0 || workspace_bytes < mpf_res_ln256_backward_workspace_bytes(rows))
        return mpf::fail(MPF_E_SHAPE, "res_ln256_backward_ws: bad rows / workspace too small");
    int rpb = (rows + 1023) / 1024;
    rpb = ((rpb + 3) / 4) * 4;
    const dim3 grid((rows + rpb - 1) / rpb);
    float* part = (float*)workspace;
    mpf::set_kernel("res_ln256_bwd_kernel");
#define RLN_BWDP(A, B, C)                                                                                                         \
    hipLaunchKernelGGL((res_ln256_bwd_kernel<A, B, C, 1>), grid, dim3(256), 0, st, s, mean, rstd, gamma, gy32, (const __bf16*)gy16,    \
                       gy_plus, ds32, (__bf16*)ds16, part, part, rows, rpb, ds_amax)
    switch ((gy32 ? 4 : 0) | (gy16 ? 2 : 0) | (gy_plus ? 1 : 0)) {
        case 1: RLN_BWDP(false, false, true); break;
        case 2: RLN_BWDP(false, true, false); break;
        case 3: RLN_BWDP(false, true, true); break;
        case 4: RLN_BWDP(true, false, false); break;
        case 5: RLN_BWDP(true, false, true); break;
        case 6: RLN_BWDP(true, true, false); break;
        default: RLN_BWDP(true, true, true); break;
    }
#undef RLN_BWDP
    hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3(2, 8), dim3(1024), 0, st, (const float*)part, (int)grid.x, dgamma, dbeta);
    return mpf::check(hipGetLastError(), "mpf_res_ln256_backward_ws");
}

// The two halves of mpf_res_ln256_backward_ws as separate calls, for callers that run SEVERAL LayerNorm backwards and reduce
// their parameter gradients together (a decoder layer: three LayerNorms, one reduce launch at the end of the layer):
//   mpf_res_ln256_backward_partial: ds + the per-workgroup partial sums into `partials`
//                                   (mpf_res_ln256_backward_workspace_bytes(rows) bytes);
//   mpf_ln_partial_reduce: out[z][2][256] = fixed-order sums of the partials of LayerNorm z (z < n_ln, `stride_bytes` apart).
extern "C" int mpf_res_ln256_backward_partial(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                              const void* gy16, const float* gy_plus, float* ds32, void* ds16, int rows, void* partials,
                                              size_t partials_bytes, void* stream)
{
    return mpf_res_ln256_backward_partial_amax(s, mean, rstd, gamma, gy32, gy16, gy_plus, ds32, ds16, rows, partials, partials_bytes, nullptr,
                                               stream);
}

// ... recording the largest |ds| in an amax slot (zeroed by the caller; may be NULL)
extern "C" int mpf_res_ln256_backward_partial_amax(const float* s, const float* mean, const float* rstd, const float* gamma,
                                                   const float* gy32, const void* gy16, const float* gy_plus, float* ds32, void* ds16,
                                                   int rows, void* partials, size_t partials_bytes, float* ds_amax, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return 0;
    if (!s || !mean || !rstd || !gamma || (!gy32 && !gy16 && !gy_plus) || (!ds32 && !ds16) || !partials)
        return mpf::fail(MPF_E_NULL, "res_ln256_backward_partial: NULL buffer");
    if (rows < 0 || partials_bytes < mpf_res_ln256_backward_workspace_bytes(rows))
        return mpf::fail(MPF_E_SHAPE, "res_ln256_backward_partial: bad rows / buffer too small");
    int rpb = (rows + 1023) / 1024;
    rpb = ((rpb + 3) / 4) * 4;
    const dim3 grid((rows + rpb - 1) / rpb);
    float* part = (float*)partials;
    mpf::set_kernel("res_ln256_bwd_kernel");
#define RLN_BWDQ(A, B, C)                                                                                                         \
    hipLaunchKernelGGL((res_ln256_bwd_kernel<A, B, C, 1>), grid, dim3(256), 0, st, s, mean, rstd, gamma, gy32, (const __bf16*)gy16,       \
                       gy_plus, ds32, (__bf16*)ds16, part, part, rows, rpb, ds_amax)
    switch ((gy32 ? 4 : 0) | (gy16 ? 2 : 0) | (gy_plus ? 1 : 0)) {
        case 1: RLN_BWDQ(false, false, true); break;
        case 2: RLN_BWDQ(false, true, false); break;
        case 3: RLN_BWDQ(false, true, true); break;
        case 4: RLN_BWDQ(true, false, false); break;
        case 5: RLN_BWDQ(true, false, true); break;
        case 6: RLN_BWDQ(true, true, false); break;
        default: RLN_BWDQ(true, true, true); break;
    }
#undef RLN_BWDQ
    return mpf::check(hipGetLastError(), "mpf_res_ln256_backward_partial");
}

extern "C" int mpf_ln_partial_reduce(const void* partials, size_t stride_bytes, int rows, int n_ln, float* out, void* stream)
{
    if (!partials || !out) return mpf::fail(MPF_E_NULL, "ln_partial_reduce: NULL buffer");
    if (rows <= 0 || n_ln <= 0 || n_ln > 65535 || (stride_bytes & 3)) return mpf::fail(MPF_E_SHAPE, "ln_partial_reduce: bad sizes");
    int rpb = (rows + 1023) / 1024;
    rpb = ((rpb + 3) / 4) * 4;
    const int blocks = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3(2, 8, n_ln), dim3(1024), 0, (hipStream_t)stream, (const float*)partials, blocks, out,
                       out + 256, (int64_t)(stride_bytes / 4));
    return mpf::check(hipGetLastError(), "mpf_ln_partial_reduce");
}

// few rows (the decoder's query side): ONE launch, the workgroup that arrives last sums the partials in block order
extern "C" size_t mpf_res_ln256_backward_det_workspace_bytes(int rows)
{
    return rows <= 0 ? 0 : 256 + mpf_res_ln256_backward_workspace_bytes(rows);
}

extern "C" int mpf_res_ln256_backward_det(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                          const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma_dbeta, int rows,
                                          void* workspace, size_t workspace_bytes, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return 0;
    if (!s || !mean || !rstd || !gamma || (!gy32 && !gy16 && !gy_plus) || (!ds32 && !ds16) || !dgamma_dbeta || !workspace)
        return mpf::fail(MPF_E_NULL, "res_ln256_backward_det: NULL buffer");
    if (rows < 0 || workspace_bytes < mpf_res_ln256_backward_det_workspace_bytes(rows))
        return mpf::fail(MPF_E_SHAPE, "res_ln256_backward_det: bad rows / workspace too small");
    int rpb = (rows + 1023) / 1024;
    rpb = ((rpb + 3) / 4) * 4;
    const dim3 grid((rows + rpb - 1) / rpb);
    float* ws = (float*)workspace;
    mpf::set_kernel("res_ln256_bwd_kernel");
#define RLN_BWDD(A, B, C)                                                                                                         \
    hipLaunchKernelGGL((res_ln256_bwd_kernel<A, B, C, 2>), grid, dim3(256), 0, st, s, mean, rstd, gamma, gy32, (const __bf16*)gy16,       \
                       gy_plus, ds32, (__bf16*)ds16, ws, dgamma_dbeta, rows, rpb)
    switch ((gy32 ? 4 : 0) | (gy16 ? 2 : 0) | (gy_plus ? 1 : 0)) {
        case 1: RLN_BWDD(false, false, true); break;
        case 2: RLN_BWDD(false, true, false); break;
        case 3: RLN_BWDD(false, true, true); break;
        case 4: RLN_BWDD(true, false, false); break;
        case 5: RLN_BWDD(true, false, true); break;
        case 6: RLN_BWDD(true, true, false); break;
        default: RLN_BWDD(true, true, true); break;
    }
#undef RLN_BWDD
    return mpf::check(hipGetLastError(), "mpf_res_ln256_backward_det");
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics (pixel decoder: nn.GroupNorm(32, 256) after the 1x1 / 3x3 convolutions,
// msdeformattn.py:245-281): mean and 1/sqrt(var + eps) of every (image, group) = one contiguous run of
// (C/G)*H*W floats in NCHW.  There are only N*G = 64 such rows at batch 2, and a one-workgroup-per-row
// kernel leaves three quarters of the chip idle (216 us for 134 MB); here every row is cut into
// chunks, one workgroup per chunk computes (count, mean, M2) of its chunk in two passes over registers,
// and the chunks are merged with Chan's formula by a second tiny launch.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kGnThreads = 256, kGnPer = 32;          // 8192 elements per chunk

__global__ __launch_bounds__(kGnThreads) void gn_chunk_stats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                                    int64_t row_len, int chunks)
{
    __shared__ float red[kGnThreads / 64];
    const int row = blockIdx.y, ch = blockIdx.x;
    const int64_t c0 = (int64_t)ch * kGnThreads * kGnPer, c1 = min(row_len, c0 + (int64_t)kGnThreads * kGnPer);
    const float* p = x + (int64_t)row * row_len;
    float v[kGnPer];
    float s = 0.f;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kGnPer / 4; ++j) {
        const int64_t i = c0 + ((int64_t)j * kGnThreads + threadIdx.x) * 4;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i + 3 < c1) { t = *reinterpret_cast<const float4*>(p + i); cnt += 4; }
        else {
            float e[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < 4; ++k) if (i + k < c1) { e[k] = p[i + k]; ++cnt; }
            t = make_float4(e[0], e[1], e[2], e[3]);
        }
        v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
        s += (t.x + t.y) + (t.z + t.w);
    }
    auto block_sum = [&](float a) {
        a = wave_sum(a);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    const float n = (float)(c1 - c0);
    const float mean = block_sum(s) / n;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < kGnPer; ++j) {
        const int64_t i = c0 + ((int64_t)(j / 4) * kGnThreads + threadIdx.x) * 4 + (j & 3);
        const float d = v[j] - mean;
        m2 += i < c1 ? d * d : 0.f;
    }
    m2 = block_sum(m2);
    if (threadIdx.x == 0) {
        float* o = part + ((int64_t)row * chunks + ch) * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
    (void)cnt;
}

__global__ __launch_bounds__(64) void gn_merge_kernel(const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd,
                                                      int rows, int chunks, float eps)
{
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    double n = 0.0, mu = 0.0, m2 = 0.0;
    for (int c = 0; c < chunks; ++c) {
        const float* o = part + ((int64_t)row * chunks + c) * 3;
        const double nb = o[0], mb = o[1], qb = o[2];
        const double d = mb - mu, nn = n + nb;
        mu += d * nb / nn;
        m2 += qb + d * d * n * nb / nn;
        n = nn;
    }
    mean[row] = (float)mu;
    rstd[row] = (float)(1.0 / sqrt(m2 / n + (double)eps));
}

}  // namespace

extern "C" size_t mpf_group_stats_workspace_bytes(int rows, int64_t row_len)
{
    if (rows <= 0 || row_len <= 0) return 0;
    const int64_t chunks = (row_len + kGnThreads * kGnPer - 1) / (kGnThreads * kGnPer);
    return (size_t)rows * chunks * 3 * sizeof(float);
}

extern "C" int mpf_group_stats(const float* x, int rows, int64_t row_len, float eps, float* mean, float* rstd,
                               void* workspace, size_t workspace_bytes, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!x || !mean || !rstd || !workspace) return mpf::fail(MPF_E_NULL, "group_stats: NULL buffer");
    if (rows <= 0 || row_len <= 0 || row_len % 4 != 0) return mpf::fail(MPF_E_SHAPE, "group_stats: row length must be a positive multiple of 4");
    if (workspace_bytes < mpf_group_stats_workspace_bytes(rows, row_len)) return mpf::fail(MPF_E_SHAPE, "group_stats: workspace too small");
    const int chunks = (int)((row_len + kGnThreads * kGnPer - 1) / (kGnThreads * kGnPer));
    mpf::prof_begin(st);
    mpf::set_kernel("gn_chunk_stats_kernel");
    hipLaunchKernelGGL(gn_chunk_stats_kernel, dim3(chunks, rows), dim3(kGnThreads), 0, st, x, (float*)workspace, row_len, chunks);
    mpf::prof_end(mpf_last_kernel(), st, 4.0 * (double)rows * (double)row_len);
    hipLaunchKernelGGL(gn_merge_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, (const float*)workspace, mean, rstd, rows, chunks, eps);
    return mpf::check(hipGetLastError(), "mpf_group_stats");
}

// ------------------------------------------------------------------------------------------------
// Grouped per-channel scale + cast over a LIST of tensors in one launch: dst_i[c, j] = cast(src_i[c, j] *
// scale_i[c]).  Folding the FrozenBatchNorm scale into the 53 convolution weights of the ResNet
// (w' = w * scale, cast to the autocast dtype) and, in the backward, turning the bf16 weight gradients
// back into fp32 parameter gradients (g = g' * scale) were 53 + 53 broadcast multiplies and 10 grouped
// casts per step (torch._foreach_mul has no fast path for a [C,1,1,1] operand).
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kGsPer = 8;               // elements per thread
constexpr int kGsBlock = 256 * kGsPer;  // elements per workgroup

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void grouped_scale_cast_kernel(const MpfScaleCastItem* __restrict__ items, int n_items)
{
    // items[i].first_block is ascending: find the item of this workgroup
    int lo = 0, hi = n_items - 1;
    const int64_t blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const MpfScaleCastItem it = items[lo];
    const TS* __restrict__ src = static_cast<const TS*>(it.src);
    TD* __restrict__ dst = static_cast<TD*>(it.dst);
    const float* __restrict__ scale = it.scale;
    const int64_t base = (blk - it.first_block) * kGsBlock;
#pragma unroll
    for (int k = 0; k < kGsPer; ++k) {
        const int64_t e = base + (int64_t)k * 256 + threadIdx.x;
        if (e < it.numel) {
            const float v = (float)src[e] * scale[e / it.inner];
            dst[e] = (TD)v;
        }
    }
}

}  // namespace

extern "C" int mpf_grouped_scale_cast(const MpfScaleCastItem* items_device, int n_items, int64_t total_blocks, int src_dtype,
                                      int dst_dtype, void* stream)
{
    if (n_items == 0 || total_blocks == 0) return 0;
    if (!items_device) return mpf::fail(MPF_E_NULL, "grouped_scale_cast: NULL table");
    if (n_items < 0 || total_blocks < 0 || total_blocks > 0x7fffffffLL) return mpf::fail(MPF_E_SHAPE, "grouped_scale_cast: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)total_blocks);
    mpf::set_kernel("grouped_scale_cast_kernel");
    if (src_dtype == MPF_F32 && dst_dtype == MPF_BF16)
        hipLaunchKernelGGL((grouped_scale_cast_kernel<float, __bf16>), grid, dim3(256), 0, st, items_device, n_items);
    else if (src_dtype == MPF_BF16 && dst_dtype == MPF_F32)
        hipLaunchKernelGGL((grouped_scale_cast_kernel<__bf16, float>), grid, dim3(256), 0, st, items_device, n_items);
    else if (src_dtype == MPF_F32 && dst_dtype == MPF_F32)
        hipLaunchKernelGGL((grouped_scale_cast_kernel<float, float>), grid, dim3(256), 0, st, items_device, n_items);
    else
        return mpf::fail(MPF_E_DTYPE, "grouped_scale_cast: (src, dst) must be (f32, bf16), (bf16, f32) or (f32, f32)");
    return mpf::check(hipGetLastError(), "mpf_grouped_scale_cast");
}

// ------------------------------------------------------------------------------------------------
// Optimizer tail: full-model gradient-norm clipping + AdamW of the reference's
// FullModelGradientClippingOptimizer (train_net.py:316-320 around torch.optim.AdamW, :259-337) in three
// launches over ALL parameters: squared-norm partials per 2048-element block, a one-workgroup
// fixed-order reduction that also derives clip = min(1, max_norm / (norm + 1e-6)), and the AdamW update
// reading `clip` from device memory (so nothing synchronises and the clipped gradients are never written).
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kOptBlock = 2048;

__device__ __forceinline__ const MpfOptItem& opt_item_of(const MpfOptItem* __restrict__ items, int n_items, int64_t blk)
{
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    return items[lo];
}

__global__ __launch_bounds__(256) void opt_sqnorm_kernel(const MpfOptItem* __restrict__ items, int n_items, float* __restrict__ partial)
{
    __shared__ float red[4];
    const MpfOptItem& it = opt_item_of(items, n_items, blockIdx.x);
    const float* __restrict__ g = it.grad;
    const int64_t base = ((int64_t)blockIdx.x - it.first_block) * kOptBlock;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kOptBlock / 256; ++k) {
        const int64_t e = base + (int64_t)k * 256 + threadIdx.x;
        const float v = e < it.numel ? g[e] : 0.f;
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = total gradient norm, out[1] = clip coefficient
__global__ __launch_bounds__(1024) void opt_clip_coef_kernel(const float* __restrict__ partial, int64_t n, float max_norm, float* __restrict__ out)
{
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        const float norm = (float)sqrt(t);
        out[0] = norm;
        out[1] = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
    }
}

__global__ __launch_bounds__(256) void opt_adamw_kernel(const MpfOptItem* __restrict__ items, int n_items, const float* __restrict__ clip,
                                                        float one_minus_beta1, float beta2, float one_minus_beta2, float eps)
{
    const MpfOptItem& it = opt_item_of(items, n_items, blockIdx.x);
    float* __restrict__ p = it.param;
    const float* __restrict__ g = it.grad;
    float* __restrict__ m = it.exp_avg;
    float* __restrict__ v = it.exp_avg_sq;
    const float c = clip ? clip[1] : 1.0f;
    const float decay = 1.0f - it.lr * it.weight_decay, step_size = it.lr / it.bc1, bc2_sqrt = it.bc2_sqrt;
    const int64_t base = ((int64_t)blockIdx.x - it.first_block) * kOptBlock;
#pragma unroll
    for (int k = 0; k < kOptBlock / 256; ++k) {
        const int64_t e = base + (int64_t)k * 256 + threadIdx.x;
        if (e < it.numel) {
            const float gr = g[e] * c;
            float pe = p[e] * decay;
            const float me = m[e] + (gr - m[e]) * one_minus_beta1;         // lerp, as torch's fused kernel
            const float ve = beta2 * v[e] + one_minus_beta2 * gr * gr;
            const float denom = sqrtf(ve) / bc2_sqrt + eps;
            pe -= step_size * (me / denom);
            p[e] = pe; m[e] = me; v[e] = ve;
        }
    }
}

}  // namespace

extern "C" int mpf_clip_adamw_step(const MpfOptItem* items_device, int n_items, int64_t total_blocks, float max_norm, double beta1,
                                   double beta2, double eps, float* partial, float* norm_clip, void* stream)
{
    if (n_items == 0 || total_blocks == 0) return 0;
    if (!items_device || !partial || !norm_clip) return mpf::fail(MPF_E_NULL, "clip_adamw_step: NULL buffer");
    if (n_items < 0 || total_blocks < 0 || total_blocks > 0x7fffffffLL) return mpf::fail(MPF_E_SHAPE, "clip_adamw_step: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    mpf::set_kernel("opt_adamw_kernel");
    hipLaunchKernelGGL(opt_sqnorm_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, items_device, n_items, partial);
    hipLaunchKernelGGL(opt_clip_coef_kernel, dim3(1), dim3(1024), 0, st, partial, total_blocks, max_norm, norm_clip);
    // the coefficients are formed in double like torch's (1 - 0.999f would be off by 5e-5 relative)
    hipLaunchKernelGGL(opt_adamw_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, items_device, n_items, norm_clip,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps);
    return mpf::check(hipGetLastError(), "mpf_clip_adamw_step");
}

// ------------------------------------------------------------------------------------------------
// [B, R, C] -> [B, C, R] fp32 transpose through LDS (64 x 64 tiles, float4 on both sides): the
// channels-last <-> NCHW relayout of the pixel decoder's conv outputs around GroupNorm (R = H*W, C = 256).
// aten's strided copy moves these 134 MB maps at 1.8 TB/s; this runs at the HBM rate.
// ------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, int64_t in_bs, float* __restrict__ out,
                                                            int64_t out_bs, int R, int C)
{
    __shared__ float tile[64][65];
    const int b = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const float* src = in + (int64_t)b * in_bs;
    float* dst = out + (int64_t)b * out_bs;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 float4 columns x 16 rows per pass
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 16 * k, c = c0 + 4 * tx;
        if (r < R && c + 3 < C) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)r * C + c);
            tile[ty + 16 * k][4 * tx] = v.x; tile[ty + 16 * k][4 * tx + 1] = v.y;
            tile[ty + 16 * k][4 * tx + 2] = v.z; tile[ty + 16 * k][4 * tx + 3] = v.w;
        } else {
            for (int e = 0; e < 4; ++e)
                tile[ty + 16 * k][4 * tx + e] = (r < R && c + e < C) ? src[(int64_t)r * C + c + e] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 16 * k, r = r0 + 4 * tx;
        if (c >= C) continue;
        const float4 v = make_float4(tile[4 * tx][ty + 16 * k], tile[4 * tx + 1][ty + 16 * k], tile[4 * tx + 2][ty + 16 * k],
                                     tile[4 * tx + 3][ty + 16 * k]);
        if (r + 3 < R) {
            *reinterpret_cast<float4*>(dst + (int64_t)c * R + r) = v;
        } else {
            const float e4[4] = {v.x, v.y, v.z, v.w};
            for (int e = 0; e < 4; ++e)
                if (r + e < R) dst[(int64_t)c * R + r + e] = e4[e];
        }
    }
}

}  // namespace

extern "C" int mpf_transpose_f32(const float* in, int64_t in_batch_stride, float* out, int64_t out_batch_stride, int B, int R, int C,
                                 void* stream)
{
    if (B == 0 || R == 0 || C == 0) return 0;
    if (!in || !out) return mpf::fail(MPF_E_NULL, "transpose_f32: NULL buffer");
    if (B < 0 || R < 0 || C < 0 || B > 65535 || R % 4 || C % 4 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || in_batch_stride % 4 ||
        out_batch_stride % 4)
        return mpf::fail(MPF_E_SHAPE, "transpose_f32: R and C must be multiples of 4, buffers 16-B aligned, B <= 65535");
    mpf::set_kernel("transpose_f32_kernel");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((C + 63) / 64, (R + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, in, in_batch_stride,
                       out, out_batch_stride, R, C);
    return mpf::check(hipGetLastError(), "mpf_transpose_f32");
}
