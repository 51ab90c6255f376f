// Point-sampled mask loss kernels (matcher cost sampling, importance sampling, BCE + dice sums and
// their backward) for MI355X.
//
// Reference semantics: detectron2 point_sample(x, c) = grid_sample(x, 2c-1, bilinear, zeros,
// align_corners=False), used at mask2former/modeling/criterion.py:164-182 (loss) and
// mask2former/modeling/matcher.py:122-132 (matching cost); sigmoid_ce_loss / dice_loss at
// criterion.py:21-65.
//
// The reference materialises, per loss call: a gathered copy of the matched prediction maps, a
// float copy of the full-resolution ground-truth masks, three grid_sample outputs, and the
// element-wise BCE / sigmoid tensors.  Here one kernel reads the prediction map (f32 or bf16) and
// the BYTE ground-truth mask directly at the sampled corners and emits four sums per (prediction,
// target) pair; the backward scatters d(loss)/d(logit) to the four corners with fp32 atomics.
// All of it is HBM/L2-latency bound gather work: one thread per point, coalesced coordinate reads.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float ld(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float ld(const __hip_bfloat16* p, int64_t i) { return __bfloat162float(p[i]); }
__device__ __forceinline__ float ld(const uint8_t* p, int64_t i) { return p[i] ? 1.f : 0.f; }

// ground-truth masks packed 32 pixels per word (mpf_pack_mask_bits): 1 MiB -> 128 KiB per 1024^2 mask,
// so a step's masks stay L2-resident under the random bilinear gathers of the loss kernels
struct BitPlane {
    const uint32_t* words;      // base of ALL planes
    int64_t first;              // pixel index of this plane's (0, 0)
};
__device__ __forceinline__ float ld(const BitPlane& m, int64_t i)
{
    const int64_t j = m.first + i;
    return ((m.words[j >> 5] >> (unsigned)(j & 31)) & 1u) ? 1.f : 0.f;
}

struct Bilin {
    int x0, y0;
    float lx, ly;
};

// align_corners=False: pixel coordinate = c * size - 0.5
__device__ __forceinline__ Bilin bilin(float cx, float cy, int h, int w)
{
    const float x = cx * (float)w - 0.5f, y = cy * (float)h - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    Bilin b;
    b.x0 = (int)xf; b.y0 = (int)yf; b.lx = x - xf; b.ly = y - yf;
    return b;
}

// the four corner loads are UNCONDITIONAL (indices clamped into the plane, out-of-range corners zeroed
// afterwards): a load behind a run-time condition makes hipcc branch around it and wait for it separately,
// i.e. four dependent round trips per sample instead of four loads in flight
template <typename MAP>
__device__ __forceinline__ float sample(const MAP map, int h, int w, const Bilin& b)
{
    const bool x0v = b.x0 >= 0 && b.x0 < w, x1v = b.x0 + 1 >= 0 && b.x0 + 1 < w;
    const bool y0v = b.y0 >= 0 && b.y0 < h, y1v = b.y0 + 1 >= 0 && b.y0 + 1 < h;
    const int xa = min(max(b.x0, 0), w - 1), xb = min(max(b.x0 + 1, 0), w - 1);
    const int64_t ra = (int64_t)min(max(b.y0, 0), h - 1) * w, rb = (int64_t)min(max(b.y0 + 1, 0), h - 1) * w;
    const float l00 = ld(map, ra + xa), l01 = ld(map, ra + xb), l10 = ld(map, rb + xa), l11 = ld(map, rb + xb);
    const float v00 = (y0v && x0v) ? l00 : 0.f, v01 = (y0v && x1v) ? l01 : 0.f;
    const float v10 = (y1v && x0v) ? l10 : 0.f, v11 = (y1v && x1v) ? l11 : 0.f;
    const float hx = 1.f - b.lx, hy = 1.f - b.ly;
    return hy * (hx * v00 + b.lx * v01) + b.ly * (hx * v10 + b.lx * v11);
}

// the ground-truth plane of row r as a sampling source: byte mask or bit-packed
template <bool BITS>
struct GtPlane;
template <>
struct GtPlane<false> {
    const uint8_t* p;
    __device__ __forceinline__ GtPlane(const void* gt, int64_t row, int H, int W) : p((const uint8_t*)gt + row * H * W) {}
    __device__ __forceinline__ float at(int H, int W, const Bilin& b) const { return sample(p, H, W, b); }
};
template <>
struct GtPlane<true> {
    BitPlane p;
    __device__ __forceinline__ GtPlane(const void* gt, int64_t row, int H, int W) : p{(const uint32_t*)gt, row * H * W} {}
    __device__ __forceinline__ float at(int H, int W, const Bilin& b) const { return sample(p, H, W, b); }
};

// bits[i] = 32 consecutive pixels of the byte masks (pixel count a multiple of 32)
__global__ __launch_bounds__(kThreads) void pack_mask_bits_kernel(const uint8_t* __restrict__ m, uint32_t* __restrict__ bits, int64_t nwords)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * kThreads) {
        const uint4 a = reinterpret_cast<const uint4*>(m)[2 * i], b = reinterpret_cast<const uint4*>(m)[2 * i + 1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) o |= (((w[k] >> (8 * j)) & 0xFFu) ? 1u : 0u) << (4 * k + j);
        bits[i] = o;
    }
}

// out[i, p] = bilinear(bit plane offs[i] (pixel index), coords[coord_rows ? coord_rows[i] : i, p])
__global__ __launch_bounds__(kThreads) void point_sample_bits_kernel(
    const uint32_t* __restrict__ src, int h, int w, const int64_t* __restrict__ offs,
    const float* __restrict__ coords, const int32_t* __restrict__ coord_rows, float* __restrict__ out, int n, int P)
{
    const int i = blockIdx.y;
    const BitPlane map{src, offs[i]};
    const float2* c = reinterpret_cast<const float2*>(coords) + (int64_t)(coord_rows ? coord_rows[i] : i) * P;
    for (int p = blockIdx.x * kThreads + threadIdx.x; p < P; p += gridDim.x * kThreads) {
        const float2 xy = c[p];
        out[(int64_t)i * P + p] = sample(map, h, w, bilin(xy.x, xy.y, h, w));
    }
}

// out[i, p] = bilinear(map at src + offs[i], coords[coord_rows ? coord_rows[i] : i, p])
template <typename T>
__global__ __launch_bounds__(kThreads) void point_sample_kernel(
    const T* __restrict__ src, int h, int w, const int64_t* __restrict__ offs,
    const float* __restrict__ coords, const int32_t* __restrict__ coord_rows, float* __restrict__ out, int n, int P)
{
    const int i = blockIdx.y;
    const T* map = src + offs[i];
    const float2* c = reinterpret_cast<const float2*>(coords) + (int64_t)(coord_rows ? coord_rows[i] : i) * P;
    for (int p = blockIdx.x * kThreads + threadIdx.x; p < P; p += gridDim.x * kThreads) {
        const float2 xy = c[p];
        out[(int64_t)i * P + p] = sample(map, h, w, bilin(xy.x, xy.y, h, w));
    }
}

__device__ __forceinline__ float block_sum(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kThreads / 64; ++k) s += red[k];
    return s;
}

// partial[i, chunk, 0..3] = sum over the chunk's points of {bce(x,t), sigmoid(x)*t, sigmoid(x), t}
template <typename T, bool BITS>
__global__ __launch_bounds__(kThreads) void mask_loss_fwd_kernel(
    const T* __restrict__ pred, int h, int w, const int64_t* __restrict__ pred_offs,
    const void* __restrict__ gt, int H, int W, const int32_t* __restrict__ gt_rows,
    const float* __restrict__ coords, float* __restrict__ partial, int n, int P, int chunks)
{
    __shared__ float red[kThreads / 64];
    const int i = blockIdx.y, ch = blockIdx.x;
    const T* pm = pred + pred_offs[i];
    const GtPlane<BITS> gm(gt, gt_rows[i], H, W);
    const float2* c = reinterpret_cast<const float2*>(coords) + (int64_t)i * P;
    const int per = (P + chunks - 1) / chunks;
    const int p0 = ch * per, p1 = min(P, p0 + per);
    float s_bce = 0.f, s_pt = 0.f, s_p = 0.f, s_t = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += kThreads) {
        const float2 xy = c[p];
        const float x = sample(pm, h, w, bilin(xy.x, xy.y, h, w));
        const float t = gm.at(H, W, bilin(xy.x, xy.y, H, W));
        const float sg = 1.f / (1.f + __expf(-x));
        s_bce += fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));   // BCE-with-logits, stable form
        s_pt += sg * t;
        s_p += sg;
        s_t += t;
    }
    s_bce = block_sum(s_bce, red);
    s_pt = block_sum(s_pt, red);
    s_p = block_sum(s_p, red);
    s_t = block_sum(s_t, red);
    if (threadIdx.x == 0) {
        float* o = partial + ((int64_t)i * chunks + ch) * 4;
        o[0] = s_bce; o[1] = s_pt; o[2] = s_p; o[3] = s_t;
    }
}

// grad map (fp32, at grad_pred + grad_offs[i]) += d/dx of  g[i,0]*bce + g[i,1]*sum(s*t) + g[i,2]*sum(s)
template <typename T>
__global__ __launch_bounds__(kThreads) void mask_loss_bwd_kernel(
    const T* __restrict__ pred, int h, int w, const int64_t* __restrict__ pred_offs,
    const uint8_t* __restrict__ gt, int H, int W, const int32_t* __restrict__ gt_rows,
    const float* __restrict__ coords, const float* __restrict__ gsum, float* __restrict__ grad_pred,
    const int64_t* __restrict__ grad_offs, int n, int P)
{
    const int i = blockIdx.y;
    const T* pm = pred + pred_offs[i];
    float* gp = grad_pred + grad_offs[i];
    const uint8_t* gm = gt + (int64_t)gt_rows[i] * H * W;
    const float2* c = reinterpret_cast<const float2*>(coords) + (int64_t)i * P;
    const float g_bce = gsum[4 * i], g_pt = gsum[4 * i + 1], g_p = gsum[4 * i + 2];
    for (int p = blockIdx.x * kThreads + threadIdx.x; p < P; p += gridDim.x * kThreads) {
        const float2 xy = c[p];
        const Bilin b = bilin(xy.x, xy.y, h, w);
        const float x = sample(pm, h, w, b);
        const float t = sample(gm, H, W, bilin(xy.x, xy.y, H, W));
        const float sg = 1.f / (1.f + __expf(-x));
        const float dx = g_bce * (sg - t) + (g_pt * t + g_p) * sg * (1.f - sg);
        const bool x0v = b.x0 >= 0 && b.x0 < w, x1v = b.x0 + 1 >= 0 && b.x0 + 1 < w;
        const bool y0v = b.y0 >= 0 && b.y0 < h, y1v = b.y0 + 1 >= 0 && b.y0 + 1 < h;
        const int64_t o = (int64_t)b.y0 * w + b.x0;
        const float hx = 1.f - b.lx, hy = 1.f - b.ly;
        if (y0v && x0v) atomicAdd(gp + o, dx * hy * hx);
        if (y0v && x1v) atomicAdd(gp + o + 1, dx * hy * b.lx);
        if (y1v && x0v) atomicAdd(gp + o + w, dx * b.ly * hx);
        if (y1v && x1v) atomicAdd(gp + o + w + 1, dx * b.ly * b.lx);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward without global atomics: one workgroup per (pair, horizontal band of the prediction plane).
// The band lives in LDS as 32-bit FIXED-POINT accumulators (per-lane LDS integer atomics run >10x
// faster than fp32 LDS or global atomics on this chip, tools/ubench/lds_atomics.hip, and make the sum
// order-independent = deterministic); every point of the pair whose bilinear footprint touches the
// band is re-evaluated and scattered into it; the band is then written ONCE, converted to the
// gradient dtype, to its place in the dense gradient of the prediction maps.  Scale: |d loss/d x| <=
// B = |g_bce| + (|g_pt| + |g_p|)/4 per point, so 2^31 / (256 B) leaves room for 256 full-magnitude
// hits on one pixel at a resolution of B * 2^-23 (fp32-grade).
// ------------------------------------------------------------------------------------------------
template <typename T, typename TG, bool BITS>
__global__ __launch_bounds__(1024) void mask_loss_bwd_band_kernel(
    const T* __restrict__ pred, int h, int w, const int64_t* __restrict__ pred_offs,
    const void* __restrict__ gt, int H, int W, const int32_t* __restrict__ gt_rows,
    const float* __restrict__ coords, const float* __restrict__ gsum, TG* __restrict__ grad,
    const int64_t* __restrict__ grad_offs, int n, int P, int band_rows)
{
    extern __shared__ int band[];
    const int i = blockIdx.y;
    const int y_lo = blockIdx.x * band_rows, y_hi = min(h, y_lo + band_rows);
    const int cells = (y_hi - y_lo) * w;
    for (int k = threadIdx.x; k < cells; k += 1024) band[k] = 0;
    const T* pm = pred + pred_offs[i];
    const GtPlane<BITS> gm(gt, gt_rows[i], H, W);
    const float2* c = reinterpret_cast<const float2*>(coords) + (int64_t)i * P;
    const float g_bce = gsum[4 * i], g_pt = gsum[4 * i + 1], g_p = gsum[4 * i + 2];
    const float B = fabsf(g_bce) + 0.25f * (fabsf(g_pt) + fabsf(g_p));
    const float scale = B > 0.f ? 8388608.f / B : 0.f;             // 2^23 / B
    const float inv = B > 0.f ? B * (1.f / 8388608.f) : 0.f;
    __syncthreads();
    if (B > 0.f) {
        for (int p = threadIdx.x; p < P; p += 1024) {
            const float2 xy = c[p];
            const Bilin b = bilin(xy.x, xy.y, h, w);
            if (b.y0 + 1 < y_lo || b.y0 >= y_hi) continue;       // footprint rows y0, y0+1 miss the band
            const float x = sample(pm, h, w, b);
            const float t = gm.at(H, W, bilin(xy.x, xy.y, H, W));
            const float sg = 1.f / (1.f + __expf(-x));
            const float dx = (g_bce * (sg - t) + (g_pt * t + g_p) * sg * (1.f - sg)) * scale;
            const bool x0v = b.x0 >= 0 && b.x0 < w, x1v = b.x0 + 1 >= 0 && b.x0 + 1 < w;
            const bool y0v = b.y0 >= y_lo && b.y0 < y_hi, y1v = b.y0 + 1 >= y_lo && b.y0 + 1 < y_hi;
            const int o = (b.y0 - y_lo) * w + b.x0;
            const float hx = 1.f - b.lx, hy = 1.f - b.ly;
            if (y0v && x0v) atomicAdd(&band[o], __float2int_rn(dx * hy * hx));
            if (y0v && x1v) atomicAdd(&band[o + 1], __float2int_rn(dx * hy * b.lx));
            if (y1v && x0v) atomicAdd(&band[o + w], __float2int_rn(dx * b.ly * hx));
            if (y1v && x1v) atomicAdd(&band[o + w + 1], __float2int_rn(dx * b.ly * b.lx));
        }
    }
    __syncthreads();
    TG* dst = grad + grad_offs[i] + (int64_t)y_lo * w;
    for (int k = threadIdx.x; k < cells; k += 1024) dst[k] = (TG)((float)band[k] * inv);
}

int check_common(const void* a, const void* b, const void* c, const void* d, int n, int P, int h, int w)
{
    if (!a || !b || !c || !d) return mpf::fail(MPF_E_NULL, "point-sample: NULL buffer");
    if (n < 0 || P <= 0 || h <= 0 || w <= 0) return mpf::fail(MPF_E_SHAPE, "point-sample: bad sizes");
    return 0;
}

}  // namespace

extern "C" int mpf_point_sample(const void* src, int src_dtype, int h, int w, const int64_t* rows,
                                const float* coords, const int32_t* coord_rows, float* out, int n, int P,
                                void* stream)
{
    if (int e = check_common(src, rows, coords, out, n, P, h, w)) return e;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (n > 65535) return mpf::fail(MPF_E_TOO_LARGE, "loss kernels: more than 65535 rows in one launch (grid.y)");
    dim3 grid((P + kThreads - 1) / kThreads, n);
    mpf::prof_begin(st);
    if (src_dtype == MPF_F32) {
        mpf::set_kernel("point_sample_kernel<float>");
        hipLaunchKernelGGL(point_sample_kernel<float>, grid, dim3(kThreads), 0, st, (const float*)src, h, w,
                           rows, coords, coord_rows, out, n, P);
    } else if (src_dtype == MPF_BF16) {
        mpf::set_kernel("point_sample_kernel<bf16>");
        hipLaunchKernelGGL(point_sample_kernel<__hip_bfloat16>, grid, dim3(kThreads), 0, st,
                           (const __hip_bfloat16*)src, h, w, rows, coords, coord_rows, out, n, P);
    } else if (src_dtype == MPF_U8) {
        mpf::set_kernel("point_sample_kernel<u8>");
        hipLaunchKernelGGL(point_sample_kernel<uint8_t>, grid, dim3(kThreads), 0, st, (const uint8_t*)src, h, w,
                           rows, coords, coord_rows, out, n, P);
    } else if (src_dtype == MPF_BITS) {        // rows[] are PIXEL indices of the planes' first pixel
        mpf::set_kernel("point_sample_bits_kernel");
        hipLaunchKernelGGL(point_sample_bits_kernel, grid, dim3(kThreads), 0, st, (const uint32_t*)src, h, w,
                           rows, coords, coord_rows, out, n, P);
    } else {
        return mpf::fail(MPF_E_DTYPE, "mpf_point_sample: dtype must be MPF_F32, MPF_BF16, MPF_U8 or MPF_BITS");
    }
    mpf::prof_end(mpf_last_kernel(), st, (double)n * P * (8.0 + 4.0 + 16.0));
    return mpf::check(hipGetLastError(), "mpf_point_sample");
}

// defined further down (needs the LDS sampling helpers): returns false if the LDS variant does not apply
static bool launch_mask_loss_fwd_lds(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_rows, const void* gt, bool bits,
                                     int H, int W, const int32_t* gt_rows, const float* coords, float* partial, int n, int P, int chunks,
                                     hipStream_t st, int* err);

extern "C" int mpf_pack_mask_bits(const uint8_t* masks, void* bits, int64_t n_pixels, void* stream)
{
    if (!masks || !bits) return mpf::fail(MPF_E_NULL, "pack_mask_bits: NULL buffer");
    if (n_pixels < 0 || n_pixels % 32 != 0) return mpf::fail(MPF_E_SHAPE, "pack_mask_bits: pixel count must be a multiple of 32");
    if (n_pixels == 0) return 0;
    const int64_t nwords = n_pixels / 32;
    const int blocks = (int)((nwords + kThreads - 1) / kThreads < 4096 ? (nwords + kThreads - 1) / kThreads : 4096);
    mpf::set_kernel("pack_mask_bits_kernel");
    hipLaunchKernelGGL(pack_mask_bits_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, masks, (uint32_t*)bits, nwords);
    return mpf::check(hipGetLastError(), "mpf_pack_mask_bits");
}

extern "C" int mpf_mask_loss_forward(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_rows,
                                     const void* gt, int gt_dtype, int H, int W, const int32_t* gt_rows,
                                     const float* coords, float* partial, int n, int P, int chunks,
                                     void* stream)
{
    if (int e = check_common(pred, pred_rows, coords, partial, n, P, h, w)) return e;
    if (!gt || !gt_rows) return mpf::fail(MPF_E_NULL, "mask_loss_forward: NULL buffer");
    if (H <= 0 || W <= 0 || chunks <= 0) return mpf::fail(MPF_E_SHAPE, "mask_loss_forward: bad sizes");
    if (gt_dtype != MPF_U8 && gt_dtype != MPF_BITS) return mpf::fail(MPF_E_DTYPE, "mask_loss_forward: gt dtype must be MPF_U8 or MPF_BITS");
    if (gt_dtype == MPF_BITS && ((int64_t)H * W) % 32 != 0) return mpf::fail(MPF_E_SHAPE, "mask_loss_forward: bit-packed masks need H*W % 32 == 0");
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (n > 65535) return mpf::fail(MPF_E_TOO_LARGE, "loss kernels: more than 65535 rows in one launch (grid.y)");
    dim3 grid(chunks, n);
    const bool bits = gt_dtype == MPF_BITS;
    mpf::prof_begin(st);
    {
        int err = 0;
        if (launch_mask_loss_fwd_lds(pred, pred_dtype, h, w, pred_rows, gt, bits, H, W, gt_rows, coords, partial, n, P, chunks, st, &err)) {
            if (err) return err;
            mpf::prof_end(mpf_last_kernel(), st, (double)n * ((double)h * w * 2.0 + P * (8.0 + 4.0)));
            return mpf::check(hipGetLastError(), "mpf_mask_loss_forward");
        }
    }
    if (pred_dtype == MPF_F32) {
        mpf::set_kernel("mask_loss_fwd_kernel<float>");
        if (bits) hipLaunchKernelGGL((mask_loss_fwd_kernel<float, true>), grid, dim3(kThreads), 0, st, (const float*)pred, h, w,
                                     pred_rows, gt, H, W, gt_rows, coords, partial, n, P, chunks);
        else hipLaunchKernelGGL((mask_loss_fwd_kernel<float, false>), grid, dim3(kThreads), 0, st, (const float*)pred, h, w,
                                pred_rows, gt, H, W, gt_rows, coords, partial, n, P, chunks);
    } else if (pred_dtype == MPF_BF16) {
        mpf::set_kernel("mask_loss_fwd_kernel<bf16>");
        if (bits) hipLaunchKernelGGL((mask_loss_fwd_kernel<__hip_bfloat16, true>), grid, dim3(kThreads), 0, st,
                                     (const __hip_bfloat16*)pred, h, w, pred_rows, gt, H, W, gt_rows, coords, partial, n, P, chunks);
        else hipLaunchKernelGGL((mask_loss_fwd_kernel<__hip_bfloat16, false>), grid, dim3(kThreads), 0, st,
                                (const __hip_bfloat16*)pred, h, w, pred_rows, gt, H, W, gt_rows, coords, partial, n, P, chunks);
    } else {
        return mpf::fail(MPF_E_DTYPE, "mpf_mask_loss_forward: pred dtype must be MPF_F32 or MPF_BF16");
    }
    mpf::prof_end(mpf_last_kernel(), st, (double)n * P * (8.0 + 16.0 + 4.0));
    return mpf::check(hipGetLastError(), "mpf_mask_loss_forward");
}

extern "C" int mpf_mask_loss_backward(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_rows,
                                      const uint8_t* gt, int H, int W, const int32_t* gt_rows,
                                      const float* coords, const float* grad_sums, float* grad_pred,
                                      const int64_t* grad_offs, int n, int P, void* stream)
{
    if (int e = check_common(pred, pred_rows, coords, grad_pred, n, P, h, w)) return e;
    if (!gt || !gt_rows || !grad_sums || !grad_offs) return mpf::fail(MPF_E_NULL, "mask_loss_backward: NULL buffer");
    if (H <= 0 || W <= 0) return mpf::fail(MPF_E_SHAPE, "mask_loss_backward: bad sizes");
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (n > 65535) return mpf::fail(MPF_E_TOO_LARGE, "loss kernels: more than 65535 rows in one launch (grid.y)");
    dim3 grid((P + kThreads * 4 - 1) / (kThreads * 4), n);
    mpf::prof_begin(st);
    if (pred_dtype == MPF_F32) {
        mpf::set_kernel("mask_loss_bwd_kernel<float>");
        hipLaunchKernelGGL(mask_loss_bwd_kernel<float>, grid, dim3(kThreads), 0, st, (const float*)pred, h, w,
                           pred_rows, gt, H, W, gt_rows, coords, grad_sums, grad_pred, grad_offs, n, P);
    } else if (pred_dtype == MPF_BF16) {
        mpf::set_kernel("mask_loss_bwd_kernel<bf16>");
        hipLaunchKernelGGL(mask_loss_bwd_kernel<__hip_bfloat16>, grid, dim3(kThreads), 0, st,
                           (const __hip_bfloat16*)pred, h, w, pred_rows, gt, H, W, gt_rows, coords, grad_sums,
                           grad_pred, grad_offs, n, P);
    } else {
        return mpf::fail(MPF_E_DTYPE, "mpf_mask_loss_backward: pred dtype must be MPF_F32 or MPF_BF16");
    }
    mpf::prof_end(mpf_last_kernel(), st, (double)n * P * (8.0 + 16.0 + 4.0 + 32.0));
    return mpf::check(hipGetLastError(), "mpf_mask_loss_backward");
}

extern "C" int mpf_mask_loss_backward_dense(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_rows,
                                            const void* gt, int gt_dtype, int H, int W, const int32_t* gt_rows, const float* coords,
                                            const float* grad_sums, void* grad, int grad_dtype, const int64_t* grad_offs,
                                            int n, int P, void* stream)
{
    if (int e = check_common(pred, pred_rows, coords, grad, n, P, h, w)) return e;
    if (!gt || !gt_rows || !grad_sums || !grad_offs) return mpf::fail(MPF_E_NULL, "mask_loss_backward_dense: NULL buffer");
    if (n == 0) return 0;
    if (grad_dtype != pred_dtype) return mpf::fail(MPF_E_DTYPE, "mask_loss_backward_dense: gradient dtype must equal the map dtype");
    if (gt_dtype != MPF_U8 && gt_dtype != MPF_BITS) return mpf::fail(MPF_E_DTYPE, "mask_loss_backward_dense: gt dtype must be MPF_U8 or MPF_BITS");
    if (gt_dtype == MPF_BITS && ((int64_t)H * W) % 32 != 0) return mpf::fail(MPF_E_SHAPE, "mask_loss_backward_dense: bit-packed masks need H*W % 32 == 0");
    const bool bits = gt_dtype == MPF_BITS;
    hipStream_t st = (hipStream_t)stream;
    // horizontal bands of at most 128 KiB of 32-bit accumulators
    const int max_rows = (128 * 1024) / (4 * w);
    if (max_rows < 2) return mpf::fail(MPF_E_TOO_LARGE, "mask_loss_backward_dense: plane rows wider than 16384");
    const int nb = (h + max_rows - 1) / max_rows;
    const int band_rows = (h + nb - 1) / nb;
    const size_t lds = (size_t)band_rows * w * 4;
    if (n > 65535) return mpf::fail(MPF_E_TOO_LARGE, "loss kernels: more than 65535 rows in one launch (grid.y)");
    const dim3 grid(nb, n);
    mpf::prof_begin(st);
    if (pred_dtype == MPF_F32) {
        mpf::set_kernel("mask_loss_bwd_band_kernel<float>");
        const void* fn = bits ? (const void*)mask_loss_bwd_band_kernel<float, float, true> : (const void*)mask_loss_bwd_band_kernel<float, float, false>;
        if (int e = mpf::check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute")) return e;
        if (bits) hipLaunchKernelGGL((mask_loss_bwd_band_kernel<float, float, true>), grid, dim3(1024), lds, st, (const float*)pred, h, w,
                                     pred_rows, gt, H, W, gt_rows, coords, grad_sums, (float*)grad, grad_offs, n, P, band_rows);
        else hipLaunchKernelGGL((mask_loss_bwd_band_kernel<float, float, false>), grid, dim3(1024), lds, st, (const float*)pred, h, w,
                                pred_rows, gt, H, W, gt_rows, coords, grad_sums, (float*)grad, grad_offs, n, P, band_rows);
    } else if (pred_dtype == MPF_BF16) {
        mpf::set_kernel("mask_loss_bwd_band_kernel<bf16>");
        const void* fn = bits ? (const void*)mask_loss_bwd_band_kernel<__hip_bfloat16, __hip_bfloat16, true>
                              : (const void*)mask_loss_bwd_band_kernel<__hip_bfloat16, __hip_bfloat16, false>;
        if (int e = mpf::check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute")) return e;
        if (bits) hipLaunchKernelGGL((mask_loss_bwd_band_kernel<__hip_bfloat16, __hip_bfloat16, true>), grid, dim3(1024), lds, st,
                                     (const __hip_bfloat16*)pred, h, w, pred_rows, gt, H, W, gt_rows, coords, grad_sums,
                                     (__hip_bfloat16*)grad, grad_offs, n, P, band_rows);
        else hipLaunchKernelGGL((mask_loss_bwd_band_kernel<__hip_bfloat16, __hip_bfloat16, false>), grid, dim3(1024), lds, st,
                                (const __hip_bfloat16*)pred, h, w, pred_rows, gt, H, W, gt_rows, coords, grad_sums,
                                (__hip_bfloat16*)grad, grad_offs, n, P, band_rows);
    } else {
        return mpf::fail(MPF_E_DTYPE, "mpf_mask_loss_backward_dense: pred dtype must be MPF_F32 or MPF_BF16");
    }
    mpf::prof_end(mpf_last_kernel(), st, (double)n * P * (8.0 + 16.0 + 4.0) + (double)n * h * w * 2.0);
    return mpf::check(hipGetLastError(), "mpf_mask_loss_backward_dense");
}

// ================================================================================================
// Importance selection: per row keep the k points with the SMALLEST |logit| (= the most uncertain,
// uncertainty = -|logit|, criterion.py:73-87 + detectron2 get_uncertain_point_coords_with_randomness)
// and emit their coordinates.  MSB-first 8-bit radix select on the bit pattern of |x| (monotonic for
// non-negative floats) with an LDS histogram, then an order-preserving compaction — replaces
// torch.topk (a full sort of 37632 keys per row) + gather.
// ================================================================================================
namespace {

constexpr int kSelThreads = 1024;

__device__ __forceinline__ int block_excl_scan_1024(int v, int* red, int* total)
{
    // exclusive scan of one int per thread over 1024 threads (16 waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kSelThreads / 64; ++k) {
        const int r = red[k];
        if (k < wave) base += r;
        tot += r;
    }
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(kSelThreads) void select_uncertain_kernel(
    const float* __restrict__ vals, const float* __restrict__ coords_in, float* __restrict__ coords_out,
    int M, int k, int P_out)
{
    __shared__ int hist[256];
    __shared__ int red[kSelThreads / 64];
    __shared__ unsigned s_prefix;
    __shared__ int s_krem;
    const int row = blockIdx.x;
    const float* v = vals + (int64_t)row * M;
    const float2* cin = reinterpret_cast<const float2*>(coords_in) + (int64_t)row * M;
    float2* cout = reinterpret_cast<float2*>(coords_out) + (int64_t)row * P_out;
    const int tid = threadIdx.x;
    if (tid == 0) { s_prefix = 0u; s_krem = k; }
    unsigned prefix = 0u, mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        prefix = s_prefix;
        for (int i = tid; i < M; i += kSelThreads) {
            const unsigned key = __float_as_uint(fabsf(v[i]));
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int krem = s_krem, d = 0;
            for (; d < 256; ++d) {
                const int c = hist[d];
                if (krem <= c) break;
                krem -= c;
            }
            if (d > 255) d = 255;
            s_krem = krem;                       // how many of the threshold digit's bucket are needed
            s_prefix = prefix | ((unsigned)d << shift);
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned thr = s_prefix;               // exact k-th smallest key
    const int need_eq = s_krem;                  // number of == thr elements to take
    // order-preserving compaction: contiguous slices per thread
    const int per = (M + kSelThreads - 1) / kSelThreads;
    const int i0 = tid * per, i1 = min(M, i0 + per);
    int n_less = 0, n_eq = 0;
    for (int i = i0; i < i1; ++i) {
        const unsigned key = __float_as_uint(fabsf(v[i]));
        n_less += key < thr;
        n_eq += key == thr;
    }
    int tot_less, tot_eq;
    const int off_less = block_excl_scan_1024(n_less, red, &tot_less);
    const int off_eq = block_excl_scan_1024(n_eq, red, &tot_eq);
    int pl = off_less, pe = off_eq;
    for (int i = i0; i < i1; ++i) {
        const unsigned key = __float_as_uint(fabsf(v[i]));
        if (key < thr) {
            cout[pl++] = cin[i];
        } else if (key == thr) {
            if (pe < need_eq) cout[tot_less + pe] = cin[i];
            ++pe;
        }
    }
}

// ================================================================================================
// Matching cost (matcher.py:105-148, mask + dice part): for every (layer, image, query) row and every
// ground-truth mask t of that image,
//   cost[row, t] = w_mask * (sum_p softplus(x_p) - sum_p x_p t_p) / P
//                + w_dice * (1 - (2 sum_p s_p t_p + 1) / (sum_p s_p + sum_p t_p + 1))
// x sampled on the fly from the prediction map, t from the pre-sampled [.., P] ground-truth points.
// One workgroup per row; TT targets at a time in registers.
// ================================================================================================
constexpr int kTT = 8;

template <typename T>
__global__ __launch_bounds__(kThreads) void match_cost_kernel(
    const T* __restrict__ pred, int h, int w, const int64_t* __restrict__ pred_offs,
    const float* __restrict__ coords, const int32_t* __restrict__ coord_rows,
    const float* __restrict__ tsamp, const int32_t* __restrict__ t_first, const int32_t* __restrict__ t_count,
    float* __restrict__ cost, int Tmax, int P, float w_mask, float w_dice)
{
    extern __shared__ float xs[];                    // [P] sampled logits of this row (gathered ONCE)
    __shared__ float red[kThreads / 64];
    const int row = blockIdx.x;
    const int Tn = t_count[row];
    if (Tn == 0) return;
    const T* pm = pred + pred_offs[row];
    const float2* c = reinterpret_cast<const float2*>(coords) + (int64_t)coord_rows[row] * P;
    const int T0 = t_first[row];
    float* out = cost + (int64_t)row * Tmax;
    float sp = 0.f, sg_sum = 0.f;
    for (int p = threadIdx.x; p < P; p += kThreads) {
        const float2 xy = c[p];
        const float x = sample(pm, h, w, bilin(xy.x, xy.y, h, w));
        xs[p] = x;
        sp += fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x)));          // softplus(x)
        sg_sum += 1.f / (1.f + __expf(-x));
    }
    sp = block_sum(sp, red);
    sg_sum = block_sum(sg_sum, red);                               // (block_sum's barriers publish xs)
    for (int tb = 0; tb < Tn; tb += kTT) {
        float ax[kTT], as[kTT], at[kTT];
#pragma unroll
        for (int j = 0; j < kTT; ++j) { ax[j] = 0.f; as[j] = 0.f; at[j] = 0.f; }
        for (int p = threadIdx.x; p < P; p += kThreads) {
            const float x = xs[p];
            const float sg = 1.f / (1.f + __expf(-x));
#pragma unroll
            for (int j = 0; j < kTT; ++j) {
                if (tb + j < Tn) {
                    const float tv = tsamp[(int64_t)(T0 + tb + j) * P + p];
                    ax[j] += x * tv; as[j] += sg * tv; at[j] += tv;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kTT; ++j) {
            const float sx = block_sum(ax[j], red), ss = block_sum(as[j], red), st = block_sum(at[j], red);
            if (threadIdx.x == 0 && tb + j < Tn) {
                const float cm = (sp - sx) / (float)P;
                const float cd = 1.f - (2.f * ss + 1.f) / (sg_sum + st + 1.f);
                out[tb + j] = w_mask * cm + w_dice * cd;
            }
        }
    }
}

}  // namespace

extern "C" int mpf_select_uncertain(const float* vals, const float* coords_in, float* coords_out,
                                    int n, int M, int k, int P_out, void* stream)
{
    if (!vals || !coords_in || !coords_out) return mpf::fail(MPF_E_NULL, "select_uncertain: NULL buffer");
    if (n < 0 || M <= 0 || k < 0 || k > M || P_out < k) return mpf::fail(MPF_E_SHAPE, "select_uncertain: bad sizes");
    if (n == 0 || k == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    mpf::prof_begin(st);
    mpf::set_kernel("select_uncertain_kernel");
    hipLaunchKernelGGL(select_uncertain_kernel, dim3(n), dim3(kSelThreads), 0, st, vals, coords_in, coords_out, M, k, P_out);
    mpf::prof_end(mpf_last_kernel(), st, (double)n * M * 12.0 * 2 + (double)n * k * 8.0);
    return mpf::check(hipGetLastError(), "mpf_select_uncertain");
}

// ------------------------------------------------------------------------------------------------
// Matching cost with the prediction plane staged in LDS.  The samples of a row are 4 x P random 2-byte
// gathers from a 128 KB plane — from L2 that is the whole cost of the kernel above; from LDS it is
// noise.  One workgroup takes G consecutive rows that share their point set and targets (the queries of
// one (output, image)): for each row the plane is copied into LDS with 16-byte loads and sampled (the
// samples and their sigmoids stay in registers, PT points per thread), then the target samples are
// streamed ONCE for the G rows.  fp32 throughout; block reductions by wave shuffles + one LDS pass.
// ------------------------------------------------------------------------------------------------
constexpr int kMcThreads = 1024;

// global -> LDS copy of a plane by a 1024-thread workgroup: all loads of a thread are issued before its
// first LDS store (a load-store-load-store loop exposes one memory latency per 16 KiB)
template <int THREADS = 1024>
__device__ __forceinline__ void copy_plane_to_lds(void* lds_plane, const void* src_plane, int plane_bytes, int tid)
{
    const uint4* src = reinterpret_cast<const uint4*>(src_plane);
    uint4* dst = reinterpret_cast<uint4*>(lds_plane);
    const int n16 = plane_bytes / 16;
    for (int base = 0; base < n16; base += 8 * THREADS) {
        uint4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = base + tid + k * THREADS;
            t[k] = idx < n16 ? src[idx] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = base + tid + k * THREADS;
            if (idx < n16) dst[idx] = t[k];
        }
    }
}

template <typename T>
__device__ __forceinline__ float sample_lds(const T* plane, int h, int w, const Bilin& b)
{
    const bool x0v = b.x0 >= 0 && b.x0 < w, x1v = b.x0 + 1 >= 0 && b.x0 + 1 < w;
    const bool y0v = b.y0 >= 0 && b.y0 < h, y1v = b.y0 + 1 >= 0 && b.y0 + 1 < h;
    const int o = b.y0 * w + b.x0;
    const float v00 = (y0v && x0v) ? ld(plane, o) : 0.f;
    const float v01 = (y0v && x1v) ? ld(plane, o + 1) : 0.f;
    const float v10 = (y1v && x0v) ? ld(plane, o + w) : 0.f;
    const float v11 = (y1v && x1v) ? ld(plane, o + w + 1) : 0.f;
    const float hx = 1.f - b.lx, hy = 1.f - b.ly;
    return hy * (hx * v00 + b.lx * v01) + b.ly * (hx * v10 + b.lx * v11);
}

// sum over the 64 lanes with DPP row operations (VALU cross-lane moves; __shfl_xor lowers to ds_bpermute
// = an LDS instruction + a wait per step) and two v_readlane
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add_(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum64(float v)
{
    v = dpp_add_<0xB1>(v);          // quad_perm [1,0,3,2]
    v = dpp_add_<0x4E>(v);          // quad_perm [2,3,0,1]
    v = dpp_add_<0x141>(v);         // row_half_mirror
    v = dpp_add_<0x140>(v);         // row_mirror: every lane = sum of its row of 16
    v = dpp_add_<0x142, 0xa>(v);    // row_bcast15 into rows 1 and 3: lanes 16..31 / 48..63 = sums of the 32-lane halves
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <typename T, int G, int PT, int THREADS>
__global__ __launch_bounds__(THREADS) void match_cost_lds_kernel(
    const T* __restrict__ pred, int h, int w, const int64_t* __restrict__ pred_offs,
    const float* __restrict__ coords, const int32_t* __restrict__ coord_rows,
    const float* __restrict__ tsamp, const int32_t* __restrict__ t_first, const int32_t* __restrict__ t_count,
    float* __restrict__ cost, int n_rows, int Tmax, int P, float w_mask, float w_dice, int plane_bytes)
{
    constexpr int TT = 4;                       // targets per pass
    constexpr int NV = 2 * G * TT + TT;         // values reduced per pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* plane = reinterpret_cast<T*>(smem);
    float* red = reinterpret_cast<float*>(smem + plane_bytes);       // [16 waves][NV]
    const int row0 = blockIdx.x * G;
    const int Tn = t_count[row0];
    if (Tn == 0) return;
    const int T0 = t_first[row0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float2* c = reinterpret_cast<const float2*>(coords) + (int64_t)coord_rows[row0] * P;
    const int ng = min(G, n_rows - row0);

    float2 xy[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int p = tid + j * THREADS;
        xy[j] = p < P ? c[p] : make_float2(-4.f, -4.f);          // far outside: samples 0
    }
    float xs[G][PT];            // (the sigmoids are recomputed per target pass: 1024 threads leave 128 VGPRs per lane)
    float sp[G], sgs[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        sp[g] = 0.f; sgs[g] = 0.f;
        __syncthreads();
        if (g < ng) copy_plane_to_lds<THREADS>(plane, pred + pred_offs[row0 + g], plane_bytes, tid);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const bool valid = (tid + j * THREADS < P) && g < ng;
            const float x = valid ? sample_lds(plane, h, w, bilin(xy[j].x, xy[j].y, h, w)) : 0.f;
            const float s_ = 1.f / (1.f + __expf(-x));
            xs[g][j] = x;                                        // (0 for padding points)
            if (valid) {
                sp[g] += fmaxf(x, 0.f) + __logf(1.f + __expf(-fabsf(x)));       // softplus; |error| <= 6e-8 per point
                sgs[g] += s_;
            }
        }
    }
    // softplus / sigmoid sums of each row
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float a = wave_sum64(sp[g]), b = wave_sum64(sgs[g]);
        if (lane == 0) { red[wave * NV + 2 * g] = a; red[wave * NV + 2 * g + 1] = b; }
    }
    __syncthreads();
    float sp_tot = 0.f, sg_tot = 0.f;          // needed by the threads that finalise (tid < G*TT): row g = tid / TT
    {
        const int g = min(tid / TT, G - 1);
#pragma unroll
        for (int k = 0; k < THREADS / 64; ++k) { sp_tot += red[k * NV + 2 * g]; sg_tot += red[k * NV + 2 * g + 1]; }
    }
    for (int tb = 0; tb < Tn; tb += TT) {
        float ax[G][TT], as_[G][TT], at[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            at[tt] = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) { ax[g][tt] = 0.f; as_[g][tt] = 0.f; }
        }
        // 5 points x TT targets of loads in flight at a time (all PT x TT at once would spill ~200 registers)
        constexpr int JC = 5;
#pragma unroll
        for (int jc = 0; jc < PT; jc += JC) {
            float tv[JC][TT];
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                const int j = jc + jj;
                const int p = min(tid + j * THREADS, P - 1);
                // unconditional loads (indices clamped, values masked below): a load behind a run-time condition
                // makes hipcc wait for each one separately — 100 dependent L2 round trips per pass
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) tv[jj][tt] = tsamp[(int64_t)(T0 + min(tb + tt, Tn - 1)) * P + p];
            }
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                const int j = jc + jj;
                if (j < PT) {
                    const bool pv = tid + j * THREADS < P;         // padding point: its target value counts as 0
                    float sgv[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) sgv[g] = __frcp_rn(1.f + __expf(-xs[g][j]));
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) {
                        const float t_ = pv ? tv[jj][tt] : 0.f;
                        at[tt] += t_;
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            ax[g][tt] += xs[g][j] * t_;
                            as_[g][tt] += sgv[g] * t_;
                        }
                    }
                }
            }
            asm volatile("" ::: "memory");       // keep the next chunk's loads behind this chunk's arithmetic
        }
        __syncthreads();                        // previous pass's red[] fully consumed
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const float v = wave_sum64(at[tt]);
            if (lane == 0) red[wave * NV + 2 * G * TT + tt] = v;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float a = wave_sum64(ax[g][tt]), b = wave_sum64(as_[g][tt]);
                if (lane == 0) { red[wave * NV + (g * TT + tt) * 2] = a; red[wave * NV + (g * TT + tt) * 2 + 1] = b; }
            }
        }
        __syncthreads();
        if (tid < G * TT) {
            const int g = tid / TT, tt = tid - g * TT;
            if (g < ng && tb + tt < Tn) {
                float sx = 0.f, ss = 0.f, st = 0.f;
#pragma unroll
                for (int k = 0; k < THREADS / 64; ++k) {
                    sx += red[k * NV + (g * TT + tt) * 2];
                    ss += red[k * NV + (g * TT + tt) * 2 + 1];
                    st += red[k * NV + 2 * G * TT + tt];
                }
                const float cm = (sp_tot - sx) / (float)P;
                const float cd = 1.f - (2.f * ss + 1.f) / (sg_tot + st + 1.f);
                cost[(int64_t)(row0 + g) * Tmax + tb + tt] = w_mask * cm + w_dice * cd;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Importance sampling in one kernel (criterion.py:164-170 + detectron2
// get_uncertain_point_coords_with_randomness): sample the M = 3P candidate points of a prediction
// plane and keep the k with the smallest |logit|.  One workgroup per (prediction, target) pair: the
// 128 KB bf16 plane is copied into LDS and sampled from there, the |logit| keys never leave the
// registers (PT per thread, point p = tid + 1024 j), the k-th smallest key is found by an MSB-first
// 8-bit radix select on an LDS histogram, and the selected coordinates are emitted in index order
// (ties at the threshold broken by index) from wave ballots + one scan of the per-(j, wave) counts.
// Replaces point_sample (60 MB of logits written and read back) + select_uncertain.
// ------------------------------------------------------------------------------------------------
template <typename T, int PT>
__global__ __launch_bounds__(kMcThreads) void sample_select_kernel(
    const T* __restrict__ pred, int h, int w, const int64_t* __restrict__ pred_offs,
    const float* __restrict__ coords_in, float* __restrict__ coords_out, int M, int k, int P_out, int plane_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* plane = reinterpret_cast<T*>(smem);
    int* hist = reinterpret_cast<int*>(smem + plane_bytes);           // [256]
    int* cnt = hist + 256;                                             // [2][PT][16] less / equal counts per (j, wave)
    int* misc = cnt + 2 * PT * 16;                                     // [0] prefix, [1] krem, [2] total less
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float2* cin = reinterpret_cast<const float2*>(coords_in) + (int64_t)row * M;
    float2* cout = reinterpret_cast<float2*>(coords_out) + (int64_t)row * P_out;
    copy_plane_to_lds(plane, pred + pred_offs[row], plane_bytes, tid);
    if (tid == 0) { misc[0] = 0; misc[1] = k; }
    __syncthreads();
    unsigned key[PT];
    // (The coordinate loads stay behind `if (p < M)`, one dependent round trip each: every unconditional form measured in round
    // 3 — batches of 2..8 on clamped indices, buffer loads with the point stride as scalar offset, scheduling barriers between
    // batches — made hipcc spill ~240 registers at the 128 VGPRs of a 1024-thread workgroup: 172 -> 280 us.)
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int p = tid + j * kMcThreads;
        key[j] = 0xFFFFFFFFu;                                          // past the end: never selected (k <= M)
        if (p < M) {
            const float2 xy = cin[p];
            key[j] = __float_as_uint(fabsf(sample_lds(plane, h, w, bilin(xy.x, xy.y, h, w))));
        }
    }
    unsigned prefix = 0u, mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        prefix = (unsigned)misc[0];
#pragma unroll
        for (int j = 0; j < PT; ++j)
            if (tid + j * kMcThreads < M && (key[j] & mask) == prefix) atomicAdd(&hist[(key[j] >> shift) & 255u], 1);
        __syncthreads();
        const int krem = misc[1];
        // exclusive prefix of the 256 digit counts: wave scan (threads 0..255 = 4 waves) + the sums of the lower waves
        // (a per-thread loop `for d < tid` was up to 255 dependent LDS reads on the critical path of every pass)
        int below = 0, mine = 0;
        if (tid < 256) {
            mine = hist[tid];
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) misc[4 + wave] = incl;
            below = incl - mine;
        }
        __syncthreads();                                               // everyone has read misc[1] / hist; wave sums published
        if (tid < 256) {
            for (int wv = 0; wv < wave; ++wv) below += misc[4 + wv];
            if (below < krem && krem <= below + mine) {                // exactly one digit holds the k-th key
                misc[1] = krem - below;
                misc[0] = (int)(prefix | ((unsigned)tid << shift));
            }
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned thr = (unsigned)misc[0];
    const int need_eq = misc[1];
    // per-(j, wave) counts -> exclusive offsets in index order p = tid + 1024 j  (j major, then wave, then lane)
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const bool in = tid + j * kMcThreads < M;
        const unsigned long long bl = __ballot(in && key[j] < thr), be = __ballot(in && key[j] == thr);
        if (lane == 0) { cnt[j * 16 + wave] = __popcll(bl); cnt[PT * 16 + j * 16 + wave] = __popcll(be); }
    }
    __syncthreads();
    if (wave < 2) {                                                    // wave 0 scans the "less" counts, wave 1 the "equal" counts
        int* c = cnt + wave * PT * 16;
        constexpr int N = PT * 16, PER = (N + 63) / 64;
        int loc[PER], tot = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = lane * PER + i;
            loc[i] = idx < N ? c[idx] : 0;
            tot += loc[i];
        }
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        int run = incl - tot;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = lane * PER + i;
            if (idx < N) c[idx] = run;
            run += loc[i];
        }
        if (wave == 0 && lane == 63) misc[2] = incl;                   // total number of keys below the threshold
    }
    __syncthreads();
    const int tot_less = misc[2];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int p = tid + j * kMcThreads;
        const bool less = p < M && key[j] < thr, eq = p < M && key[j] == thr;
        const unsigned long long bl = __ballot(less), be = __ballot(eq);
        if (less) {
            cout[cnt[j * 16 + wave] + __popcll(bl & lt)] = cin[p];
        } else if (eq) {
            const int rank = cnt[PT * 16 + j * 16 + wave] + __popcll(be & lt);
            if (rank < need_eq) cout[tot_less + rank] = cin[p];
        }
    }
}

// mask_loss_fwd with the prediction plane staged in LDS (bf16 planes of at most 128 KiB): one workgroup per
// pair, all P points; partial[i, 0, :] receives the sums, the other chunks zeros
template <bool BITS>
__global__ __launch_bounds__(kMcThreads) void mask_loss_fwd_lds_kernel(
    const __hip_bfloat16* __restrict__ pred, int h, int w, const int64_t* __restrict__ pred_offs,
    const void* __restrict__ gt, int H, int W, const int32_t* __restrict__ gt_rows,
    const float* __restrict__ coords, float* __restrict__ partial, int P, int chunks, int plane_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __hip_bfloat16* plane = reinterpret_cast<__hip_bfloat16*>(smem);
    float* red = reinterpret_cast<float*>(smem + plane_bytes);          // [16 waves][4]
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    copy_plane_to_lds(plane, pred + pred_offs[i], plane_bytes, tid);
    const GtPlane<BITS> gm(gt, gt_rows[i], H, W);
    const float2* c = reinterpret_cast<const float2*>(coords) + (int64_t)i * P;
    __syncthreads();
    float s_bce = 0.f, s_pt = 0.f, s_p = 0.f, s_t = 0.f;
    for (int p = tid; p < P; p += kMcThreads) {
        const float2 xy = c[p];
        const float x = sample_lds(plane, h, w, bilin(xy.x, xy.y, h, w));
        const float t = gm.at(H, W, bilin(xy.x, xy.y, H, W));
        const float sg = 1.f / (1.f + __expf(-x));
        s_bce += fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
        s_pt += sg * t;
        s_p += sg;
        s_t += t;
    }
    s_bce = wave_sum64(s_bce); s_pt = wave_sum64(s_pt); s_p = wave_sum64(s_p); s_t = wave_sum64(s_t);
    if (lane == 0) { red[wave * 4] = s_bce; red[wave * 4 + 1] = s_pt; red[wave * 4 + 2] = s_p; red[wave * 4 + 3] = s_t; }
    __syncthreads();
    if (tid < 4 * chunks) {
        float v = 0.f;
        if (tid < 4)
            for (int k = 0; k < kMcThreads / 64; ++k) v += red[k * 4 + tid];
        partial[(int64_t)i * chunks * 4 + tid] = v;
    }
}

static bool launch_mask_loss_fwd_lds(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_rows, const void* gt, bool bits,
                                     int H, int W, const int32_t* gt_rows, const float* coords, float* partial, int n, int P, int chunks,
                                     hipStream_t st, int* err)
{
    const int plane_bytes = h * w * 2;
    if (pred_dtype != MPF_BF16 || plane_bytes % 16 != 0 || plane_bytes > 128 * 1024 || chunks > 256) return false;
    const size_t lds = (size_t)plane_bytes + (kMcThreads / 64) * 4 * sizeof(float);
    const void* fn = bits ? (const void*)mask_loss_fwd_lds_kernel<true> : (const void*)mask_loss_fwd_lds_kernel<false>;
    *err = mpf::check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    if (*err) return true;
    mpf::set_kernel("mask_loss_fwd_lds_kernel");
    if (bits) hipLaunchKernelGGL(mask_loss_fwd_lds_kernel<true>, dim3(n), dim3(kMcThreads), lds, st, (const __hip_bfloat16*)pred, h, w,
                                 pred_rows, gt, H, W, gt_rows, coords, partial, P, chunks, plane_bytes);
    else hipLaunchKernelGGL(mask_loss_fwd_lds_kernel<false>, dim3(n), dim3(kMcThreads), lds, st, (const __hip_bfloat16*)pred, h, w,
                            pred_rows, gt, H, W, gt_rows, coords, partial, P, chunks, plane_bytes);
    return true;
}

// 512 threads x 25 points: the samples of 4 rows stay in registers without spilling (1024 threads leave
// 128 VGPRs per lane: 180 spilled registers and 2.6x the time)
template <typename T>
static bool launch_match_cost_lds(const T* pred, int h, int w, const int64_t* pred_offs, const float* coords,
                                  const int32_t* coord_rows, const float* tsamp, const int32_t* t_first, const int32_t* t_count,
                                  float* cost, int n_rows, int Tmax, int P, float w_mask, float w_dice, int G, hipStream_t st,
                                  int* err)
{
    constexpr int THREADS = 512, PT = 25;
    const int plane_bytes = h * w * (int)sizeof(T);
    constexpr int GQ = 2;                      // rows per workgroup (4 would need ~300 registers per lane)
    if (G % GQ != 0 || plane_bytes % 16 != 0 || plane_bytes > 136 * 1024 || P > PT * THREADS || n_rows % GQ != 0) return false;
    constexpr int NV = 2 * GQ * 4 + 4;
    const size_t lds = (size_t)plane_bytes + (THREADS / 64) * NV * sizeof(float);
    auto kfn = match_cost_lds_kernel<T, GQ, PT, THREADS>;
    *err = mpf::check(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    if (*err) return true;
    hipLaunchKernelGGL(kfn, dim3(n_rows / GQ), dim3(THREADS), lds, st, pred, h, w, pred_offs, coords, coord_rows, tsamp, t_first,
                       t_count, cost, n_rows, Tmax, P, w_mask, w_dice, plane_bytes);
    return true;
}

extern "C" int mpf_match_cost(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_offs,
                              const float* coords, const int32_t* coord_rows, const float* tsamp,
                              const int32_t* t_first, const int32_t* t_count, float* cost,
                              int n_rows, int Tmax, int P, float w_mask, float w_dice, int rows_per_group, void* stream)
{
    if (!pred || !pred_offs || !coords || !coord_rows || !tsamp || !t_first || !t_count || !cost)
        return mpf::fail(MPF_E_NULL, "match_cost: NULL buffer");
    if (n_rows < 0 || Tmax <= 0 || P <= 0 || h <= 0 || w <= 0) return mpf::fail(MPF_E_SHAPE, "match_cost: bad sizes");
    if ((size_t)P * 4 > 150 * 1024) return mpf::fail(MPF_E_TOO_LARGE, "match_cost: more than 38400 points per row");
    if (n_rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    mpf::prof_begin(st);
    if (pred_dtype == MPF_BF16 && rows_per_group >= 2 && rows_per_group % 2 == 0) {
        int err = 0;
        mpf::set_kernel("match_cost_lds_kernel<bf16>");
        if (launch_match_cost_lds((const __hip_bfloat16*)pred, h, w, pred_offs, coords, coord_rows, tsamp, t_first, t_count, cost,
                                  n_rows, Tmax, P, w_mask, w_dice, rows_per_group, st, &err)) {
            if (err) return err;
            mpf::prof_end(mpf_last_kernel(), st, (double)n_rows * ((double)h * w * 2.0 + P * 8.0 / 4 + P * 4.0 * Tmax / 4));
            return mpf::check(hipGetLastError(), "mpf_match_cost");
        }
    }
    if (pred_dtype == MPF_F32) {
        mpf::set_kernel("match_cost_kernel<float>");
        hipLaunchKernelGGL(match_cost_kernel<float>, dim3(n_rows), dim3(kThreads), (size_t)P * 4, st, (const float*)pred, h, w,
                           pred_offs, coords, coord_rows, tsamp, t_first, t_count, cost, Tmax, P, w_mask, w_dice);
    } else if (pred_dtype == MPF_BF16) {
        mpf::set_kernel("match_cost_kernel<bf16>");
        hipLaunchKernelGGL(match_cost_kernel<__hip_bfloat16>, dim3(n_rows), dim3(kThreads), (size_t)P * 4, st,
                           (const __hip_bfloat16*)pred, h, w, pred_offs, coords, coord_rows, tsamp, t_first, t_count,
                           cost, Tmax, P, w_mask, w_dice);
    } else {
        return mpf::fail(MPF_E_DTYPE, "mpf_match_cost: pred dtype must be MPF_F32 or MPF_BF16");
    }
    mpf::prof_end(mpf_last_kernel(), st, (double)n_rows * P * (8.0 + 16.0 + 4.0 * Tmax));
    return mpf::check(hipGetLastError(), "mpf_match_cost");
}

extern "C" int mpf_sample_select_uncertain(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_offs,
                                           const float* coords_in, float* coords_out, int n, int M, int k, int P_out,
                                           void* stream)
{
    if (!pred || !pred_offs || !coords_in || !coords_out) return mpf::fail(MPF_E_NULL, "sample_select_uncertain: NULL buffer");
    if (n < 0 || M <= 0 || k < 0 || k > M || k > P_out || h <= 0 || w <= 0) return mpf::fail(MPF_E_SHAPE, "sample_select_uncertain: bad sizes");
    if (n == 0 || k == 0) return 0;
    const int plane_bytes = h * w * 2;
    if (pred_dtype != MPF_BF16) return mpf::fail(MPF_E_DTYPE, "sample_select_uncertain: bf16 maps only (use mpf_point_sample + mpf_select_uncertain)");
    if (plane_bytes % 16 != 0 || plane_bytes > 128 * 1024 || M > 40 * kMcThreads)
        return mpf::fail(MPF_E_TOO_LARGE, "sample_select_uncertain: plane larger than 128 KiB or more than 40960 candidates");
    hipStream_t st = (hipStream_t)stream;
    // keys per thread: 37 covers the shipped 3 x 12544 candidates without register spills; 40 is the general cap
    const int PT = M <= 16 * kMcThreads ? 16 : (M <= 37 * kMcThreads ? 37 : 40);
    const size_t lds = (size_t)plane_bytes + (256 + 2 * PT * 16 + 8) * sizeof(int);
    const void* fn = PT == 16 ? (const void*)sample_select_kernel<__hip_bfloat16, 16>
                              : (PT == 37 ? (const void*)sample_select_kernel<__hip_bfloat16, 37> : (const void*)sample_select_kernel<__hip_bfloat16, 40>);
    if (int e = mpf::check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute")) return e;
    mpf::prof_begin(st);
    mpf::set_kernel("sample_select_kernel<bf16>");
#define MPF_SS_LAUNCH(PTV)                                                                                                        \
    hipLaunchKernelGGL((sample_select_kernel<__hip_bfloat16, PTV>), dim3(n), dim3(kMcThreads), lds, st, (const __hip_bfloat16*)pred, h, w, \
                       pred_offs, coords_in, coords_out, M, k, P_out, plane_bytes)
    if (PT == 16) MPF_SS_LAUNCH(16); else if (PT == 37) MPF_SS_LAUNCH(37); else MPF_SS_LAUNCH(40);
#undef MPF_SS_LAUNCH
    mpf::prof_end(mpf_last_kernel(), st, (double)n * ((double)plane_bytes + M * 8.0 + k * 16.0));
    return mpf::check(hipGetLastError(), "mpf_sample_select_uncertain");
}
