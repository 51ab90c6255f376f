// Decoder-side kernels for MI355X.
//
// attn_mask_kernel: the attention mask of the next decoder layer from the mask logits
// (mask2former_transformer_decoder.py:1869-1875: bilinear resize of outputs_mask to the level size,
// sigmoid < 0.5, repeated over heads) fused with the mask-piloted row overwrite (:1814-1816, rows of
// the MP queries come from the ground-truth masks) and the "a fully masked row attends everywhere"
// rule (:1780).  One [N, Qtot, hl*wl] byte mask shared by the heads is produced directly; the
// reference's float copy of the logits, the resized float tensor, the sigmoid, the 8x repeat and the
// separate all()/where passes are gone.  sigmoid(x) < 0.5  <=>  x < 0.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float ldf(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const __hip_bfloat16* p, int64_t i) { return __bfloat162float(p[i]); }

// one workgroup per (image, query) row
template <typename T>
__global__ __launch_bounds__(kThreads) void attn_mask_kernel(
    const T* __restrict__ masks, int64_t stride_n, int64_t stride_q, int h, int w,
    const uint8_t* __restrict__ mp_rows, int pad, uint8_t* __restrict__ out, int Q, int hl, int wl)
{
    extern __shared__ uint8_t bits[];              // [hl*wl]
    __shared__ int any_open;
    const int n = blockIdx.x / Q, q = blockIdx.x % Q;
    const int HW = hl * wl;
    if (threadIdx.x == 0) any_open = 0;
    __syncthreads();
    int open = 0;
    if (q < pad) {
        const uint8_t* src = mp_rows + ((int64_t)n * pad + q) * HW;
        for (int i = threadIdx.x; i < HW; i += kThreads) {
            const uint8_t b = src[i] ? 1 : 0;
            bits[i] = b;
            open |= !b;
        }
    } else {
        const T* m = masks + n * stride_n + q * stride_q;
        // F.interpolate(mode="bilinear", align_corners=False): src = max(0, (dst + 0.5) * in/out - 0.5)
        const float sy = (float)h / (float)hl, sx = (float)w / (float)wl;
        for (int i = threadIdx.x; i < HW; i += kThreads) {
            const int oy = i / wl, ox = i - oy * wl;
            const float fy = fmaxf(0.f, ((float)oy + 0.5f) * sy - 0.5f);
            const float fx = fmaxf(0.f, ((float)ox + 0.5f) * sx - 0.5f);
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float v = (1.f - ly) * ((1.f - lx) * ldf(m, (int64_t)y0 * w + x0) + lx * ldf(m, (int64_t)y0 * w + x1)) +
                            ly * ((1.f - lx) * ldf(m, (int64_t)y1 * w + x0) + lx * ldf(m, (int64_t)y1 * w + x1));
            const uint8_t b = v < 0.f ? 1 : 0;        // True = do not attend
            bits[i] = b;
            open |= !b;
        }
    }
    if (__any(open) && (threadIdx.x & 63) == 0) atomicOr(&any_open, 1);
    __syncthreads();
    const bool keep = any_open != 0;                  // fully masked row -> attend everywhere (:1780)
    uint8_t* dst = out + ((int64_t)n * Q + q) * HW;
    for (int i = threadIdx.x; i < HW; i += kThreads) dst[i] = keep ? bits[i] : 0;
}

}  // namespace

extern "C" int mpf_attn_mask(const void* masks, int dtype, int64_t stride_n, int64_t stride_q, int h, int w,
                             const uint8_t* mp_rows, int pad, uint8_t* out, int N, int Q, int hl, int wl,
                             void* stream)
{
    if (!masks || !out || (pad > 0 && !mp_rows)) return mpf::fail(MPF_E_NULL, "attn_mask: NULL buffer");
    if (N <= 0 || Q <= 0 || h <= 0 || w <= 0 || hl <= 0 || wl <= 0 || pad < 0 || pad > Q)
        return mpf::fail(MPF_E_SHAPE, "attn_mask: bad sizes");
    if ((size_t)hl * wl > 96 * 1024) return mpf::fail(MPF_E_TOO_LARGE, "attn_mask: level larger than 96K positions");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)hl * wl;
    if (lds > 64 * 1024) {
        // more dynamic LDS than the default per-kernel limit: opt in explicitly (160 KB per CU on gfx950)
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mask_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mask_kernel<__hip_bfloat16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess || e2 != hipSuccess) return mpf::check(e1 != hipSuccess ? e1 : e2, "mpf_attn_mask(hipFuncSetAttribute)");
    }
    mpf::prof_begin(st);
    if (dtype == MPF_F32) {
        mpf::set_kernel("attn_mask_kernel<float>");
        hipLaunchKernelGGL(attn_mask_kernel<float>, dim3(N * Q), dim3(kThreads), lds, st, (const float*)masks, stride_n,
                           stride_q, h, w, mp_rows, pad, out, Q, hl, wl);
    } else if (dtype == MPF_BF16) {
        mpf::set_kernel("attn_mask_kernel<bf16>");
        hipLaunchKernelGGL(attn_mask_kernel<__hip_bfloat16>, dim3(N * Q), dim3(kThreads), lds, st,
                           (const __hip_bfloat16*)masks, stride_n, stride_q, h, w, mp_rows, pad, out, Q, hl, wl);
    } else {
        return mpf::fail(MPF_E_DTYPE, "mpf_attn_mask: dtype must be MPF_F32 or MPF_BF16");
    }
    mpf::prof_end(mpf_last_kernel(), st, (double)N * Q * hl * wl * (1.0 + 4.0 * (dtype == MPF_F32 ? 4.0 : 2.0)));
    return mpf::check(hipGetLastError(), "mpf_attn_mask");
}

// ------------------------------------------------------------------------------------------------
// 'masked' rows of the mask-piloted queries: out[t, y, x] = 1 iff ground-truth mask t has NO pixel in
// the (H/h) x (W/w) block (y, x) — F.interpolate(mode='area') <= 1e-8 of prepare_for_dn_v5
// (mask2former_transformer_decoder.py:986-987) for H % h == 0, W % w == 0.  One thread per block,
// adjacent threads own adjacent blocks of a row, so a wave reads whole mask rows.
// ------------------------------------------------------------------------------------------------
namespace {

template <typename WORD>
__global__ __launch_bounds__(256) void block_empty_kernel(const uint8_t* __restrict__ masks, uint8_t* __restrict__ out, int T, int H,
                                                          int W, int h, int w)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)T * h * w;
    if (idx >= total) return;
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const int64_t t = idx / ((int64_t)w * h);
    const int bh = H / h, bw = W / w;
    const uint8_t* p = masks + (t * H + (int64_t)y * bh) * W + (int64_t)x * bw;
    constexpr int WB = (int)sizeof(WORD);
    WORD acc = 0;
    for (int r = 0; r < bh; ++r) {
        const WORD* row = reinterpret_cast<const WORD*>(p + (int64_t)r * W);
        for (int c = 0; c < bw / WB; ++c) acc |= row[c];
    }
    out[idx] = acc == 0 ? 1 : 0;
}

}  // namespace

extern "C" int mpf_mask_block_empty(const uint8_t* masks, uint8_t* out, int T, int H, int W, int h, int w, void* stream)
{
    if (T == 0) return 0;
    if (!masks || !out) return mpf::fail(MPF_E_NULL, "mask_block_empty: NULL buffer");
    if (T < 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || H % h || W % w)
        return mpf::fail(MPF_E_SHAPE, "mask_block_empty: the level size must divide the mask size");
    hipStream_t st = (hipStream_t)stream;
    const int bw = W / w;
    const int64_t total = (int64_t)T * h * w;
    const dim3 grid((unsigned)((total + 255) / 256));
    mpf::prof_begin(st);
    mpf::set_kernel("block_empty_kernel");
    if (bw % 8 == 0 && W % 8 == 0 && ((uintptr_t)masks & 7) == 0)
        hipLaunchKernelGGL(block_empty_kernel<uint64_t>, grid, dim3(256), 0, st, masks, out, T, H, W, h, w);
    else if (bw % 4 == 0 && W % 4 == 0 && ((uintptr_t)masks & 3) == 0)
        hipLaunchKernelGGL(block_empty_kernel<uint32_t>, grid, dim3(256), 0, st, masks, out, T, H, W, h, w);
    else
        hipLaunchKernelGGL(block_empty_kernel<uint8_t>, grid, dim3(256), 0, st, masks, out, T, H, W, h, w);
    mpf::prof_end("block_empty_kernel", st, (double)T * H * W + (double)total);
    return mpf::check(hipGetLastError(), "mpf_mask_block_empty");
}

// ------------------------------------------------------------------------------------------------
// Decoder inputs of one feature level (mask2former_transformer_decoder.py:1756-1764): from the level's
// feature map x (n, c, s) — here a channel-last VIEW of the encoder memory —
//   src[s, n, c]  = x[n, c, s] + level_embed[c]               (value input of the cross-attention)
//   kin[s, n, c]  = src[s, n, c] + pos[s, c]                   (key input: + sine position embedding)
// both written sequence-first, row-contiguous, in the autocast dtype (bf16) or fp32: one pass instead of
// two strided adds and two strided casts.  Backward: dx(n, c, s) = g_src[s, n, c] + g_kin[s, n, c].
// Thread = 8 consecutive channels of one (s, n); x needs unit channel stride.
// ------------------------------------------------------------------------------------------------
namespace {

template <typename TO>
__device__ __forceinline__ void store8(TO* dst, const float (&v)[8]);

template <>
__device__ __forceinline__ void store8<float>(float* dst, const float (&v)[8])
{
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

template <>
__device__ __forceinline__ void store8<__hip_bfloat16>(__hip_bfloat16* dst, const float (&v)[8])
{
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __hip_bfloat16 lo = __float2bfloat16(v[2 * j]), hi = __float2bfloat16(v[2 * j + 1]);
        w[j] = (unsigned)(*reinterpret_cast<const unsigned short*>(&lo)) | ((unsigned)(*reinterpret_cast<const unsigned short*>(&hi)) << 16);
    }
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename TO>
__global__ __launch_bounds__(256) void decoder_inputs_fwd_kernel(const float* __restrict__ x, int64_t sx_n, int64_t sx_s,
                                                                 const float* __restrict__ level_embed, const float* __restrict__ pos,
                                                                 TO* __restrict__ src, TO* __restrict__ kin, int S, int N, int C)
{
    const int c8 = C >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)S * N * c8) return;
    const int cg = (int)(idx % c8);
    const int n = (int)((idx / c8) % N);
    const int s = (int)(idx / ((int64_t)c8 * N));
    const float* xp = x + (int64_t)n * sx_n + (int64_t)s * sx_s + 8 * cg;
    const float4 a0 = *reinterpret_cast<const float4*>(xp), a1 = *reinterpret_cast<const float4*>(xp + 4);
    const float4 e0 = *reinterpret_cast<const float4*>(level_embed + 8 * cg), e1 = *reinterpret_cast<const float4*>(level_embed + 8 * cg + 4);
    const float4 p0 = *reinterpret_cast<const float4*>(pos + (int64_t)s * C + 8 * cg), p1 = *reinterpret_cast<const float4*>(pos + (int64_t)s * C + 8 * cg + 4);
    const float v[8] = {a0.x + e0.x, a0.y + e0.y, a0.z + e0.z, a0.w + e0.w, a1.x + e1.x, a1.y + e1.y, a1.z + e1.z, a1.w + e1.w};
    const float k[8] = {v[0] + p0.x, v[1] + p0.y, v[2] + p0.z, v[3] + p0.w, v[4] + p1.x, v[5] + p1.y, v[6] + p1.z, v[7] + p1.w};
    const int64_t o = ((int64_t)s * N + n) * C + 8 * cg;
    store8<TO>(src + o, v);
    store8<TO>(kin + o, k);
}

__device__ __forceinline__ void load8(const float* p, float (&v)[8])
{
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

__device__ __forceinline__ void load8(const __hip_bfloat16* p, float (&v)[8])
{
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(w[j] << 16);
        v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
}

template <typename TG>
__global__ __launch_bounds__(256) void decoder_inputs_bwd_kernel(const TG* __restrict__ g_src, const TG* __restrict__ g_kin,
                                                                 float* __restrict__ dx, int64_t sx_n, int64_t sx_s, int S, int N, int C)
{
    const int c8 = C >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)S * N * c8) return;
    const int cg = (int)(idx % c8);
    const int n = (int)((idx / c8) % N);
    const int s = (int)(idx / ((int64_t)c8 * N));
    const int64_t o = ((int64_t)s * N + n) * C + 8 * cg;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g_src) load8(g_src + o, a);
    if (g_kin) load8(g_kin + o, b);
    float* d = dx + (int64_t)n * sx_n + (int64_t)s * sx_s + 8 * cg;
    *reinterpret_cast<float4*>(d) = make_float4(a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]);
    *reinterpret_cast<float4*>(d + 4) = make_float4(a[4] + b[4], a[5] + b[5], a[6] + b[6], a[7] + b[7]);
}

}  // namespace

extern "C" int mpf_decoder_inputs_forward(const float* x, int64_t sx_n, int64_t sx_s, const float* level_embed, const float* pos,
                                          void* src, void* kin, int out_dtype, int S, int N, int C, void* stream)
{
    if (S == 0 || N == 0) return 0;
    if (!x || !level_embed || !pos || !src || !kin) return mpf::fail(MPF_E_NULL, "decoder_inputs_forward: NULL buffer");
    if (S < 0 || N < 0 || C <= 0 || C % 8 || sx_n % 4 || sx_s % 4 || ((uintptr_t)x & 15))
        return mpf::fail(MPF_E_SHAPE, "decoder_inputs_forward: C % 8 == 0 and 16-B aligned rows (unit channel stride) required");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)S * N * (C / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    mpf::set_kernel("decoder_inputs_fwd_kernel");
    if (out_dtype == MPF_BF16)
        hipLaunchKernelGGL(decoder_inputs_fwd_kernel<__hip_bfloat16>, grid, dim3(256), 0, st, x, sx_n, sx_s, level_embed, pos,
                           (__hip_bfloat16*)src, (__hip_bfloat16*)kin, S, N, C);
    else if (out_dtype == MPF_F32)
        hipLaunchKernelGGL(decoder_inputs_fwd_kernel<float>, grid, dim3(256), 0, st, x, sx_n, sx_s, level_embed, pos, (float*)src,
                           (float*)kin, S, N, C);
    else
        return mpf::fail(MPF_E_DTYPE, "decoder_inputs_forward: out dtype must be MPF_F32 or MPF_BF16");
    return mpf::check(hipGetLastError(), "mpf_decoder_inputs_forward");
}

extern "C" int mpf_decoder_inputs_backward(const void* g_src, const void* g_kin, int g_dtype, float* dx, int64_t sx_n, int64_t sx_s,
                                           int S, int N, int C, void* stream)
{
    if (S == 0 || N == 0) return 0;
    if (!dx || (!g_src && !g_kin)) return mpf::fail(MPF_E_NULL, "decoder_inputs_backward: NULL buffer");
    if (S < 0 || N < 0 || C <= 0 || C % 8 || sx_n % 4 || sx_s % 4 || ((uintptr_t)dx & 15))
        return mpf::fail(MPF_E_SHAPE, "decoder_inputs_backward: C % 8 == 0 and 16-B aligned rows required");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)S * N * (C / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    mpf::set_kernel("decoder_inputs_bwd_kernel");
    if (g_dtype == MPF_BF16)
        hipLaunchKernelGGL(decoder_inputs_bwd_kernel<__hip_bfloat16>, grid, dim3(256), 0, st, (const __hip_bfloat16*)g_src,
                           (const __hip_bfloat16*)g_kin, dx, sx_n, sx_s, S, N, C);
    else if (g_dtype == MPF_F32)
        hipLaunchKernelGGL(decoder_inputs_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)g_src, (const float*)g_kin, dx,
                           sx_n, sx_s, S, N, C);
    else
        return mpf::fail(MPF_E_DTYPE, "decoder_inputs_backward: gradient dtype must be MPF_F32 or MPF_BF16");
    return mpf::check(hipGetLastError(), "mpf_decoder_inputs_backward");
}
