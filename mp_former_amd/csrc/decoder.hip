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
