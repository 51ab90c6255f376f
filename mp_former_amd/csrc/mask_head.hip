// Fused mask head -> next-layer attention mask for MI355X (bf16 MFMA).
//
// Reference (mask2former_transformer_decoder.py:1859-1877, forward_prediction_heads): per decoder layer
//     outputs_mask = einsum("bqc,bchw->bqhw", mask_embed, mask_features)          [N, Qtot, H/4, W/4]
//     attn_mask    = F.interpolate(outputs_mask, size=level, mode="bilinear", align_corners=False)
//     attn_mask    = (attn_mask.sigmoid() < 0.5)   (repeated over the 8 heads, detached)
// i.e. ten times per step a 3.4 GF product whose 31 MB result is written, re-read by the resize and thrown away.
//
// Bilinear resizing is linear and acts on the pixel axis only, so it commutes with the channel contraction:
//     resize(mask_embed . mask_features) = mask_embed . resize(mask_features)
// The pixel-decoder features are therefore resized ONCE per step to each of the three level grids
// (mpf_pool_features: [N, 256, H/4, W/4] -> [N, h_l*w_l, 256] bf16, pixel-major so that a pixel's 256 channels are the
// contraction-contiguous B operand), and the per-layer work shrinks to a [Qtot x 256] x [256 x h_l*w_l] product —
// 4 / 16 / 64 times fewer flops than the full-resolution map — whose result never leaves the registers:
// mpf_mask_head_bits forms 16 x 16 tiles with v_mfma_f32_16x16x32_bf16, takes the sign (sigmoid(x) < 0.5 <=> x < 0),
// overwrites the mask-piloted rows with their ground-truth rows (:1814-1816), stages the bytes of a
// [<=128 queries x 128 pixels] block in LDS and writes 128-byte rows.  The "a fully masked row attends everywhere"
// rule (:1780) needs a whole row: the product kernel ORs a per-row "has an open pixel" flag, a second tiny launch
// clears the rows whose flag stayed 0 (and resets the flags).  The full-resolution map is never formed.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int kC = 256;           // mask dimension (contraction length)
constexpr int kT = 256;

__device__ __forceinline__ float ldf(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const __hip_bfloat16* p, int64_t i) { return __bfloat162float(p[i]); }

// ----------------------------------------------------------------------------------------------------------------
// resize(mask_features) to one level grid, transposed to pixel-major bf16
// workgroup = 32 consecutive output pixels of one output row x all 256 channels
// ----------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kT) void pool_features_kernel(const T* __restrict__ mf, __hip_bfloat16* __restrict__ out, int h, int w,
                                                           int hl, int wl, int xblocks)
{
    __shared__ __hip_bfloat16 tile[32][kC + 8];          // +8: rows 528 B apart (bank spread for the column writes)
    const int bx = blockIdx.x % xblocks, oy = (blockIdx.x / xblocks) % hl, n = blockIdx.x / (xblocks * hl);
    const int tid = threadIdx.x, px = tid & 31, cg = tid >> 5;        // 8 channels in flight, 32 pixels
    const int ox = bx * 32 + px;
    // F.interpolate(mode="bilinear", align_corners=False): src = max(0, (dst + 0.5) * in/out - 0.5)
    const float sy = (float)h / (float)hl, sx = (float)w / (float)wl;
    const float fy = fmaxf(0.f, ((float)oy + 0.5f) * sy - 0.5f);
    const float fx = fmaxf(0.f, ((float)min(ox, wl - 1) + 0.5f) * sx - 0.5f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const int64_t o00 = (int64_t)y0 * w + x0, o01 = (int64_t)y0 * w + x1, o10 = (int64_t)y1 * w + x0, o11 = (int64_t)y1 * w + x1;
    const T* base = mf + (int64_t)n * kC * h * w;
#pragma unroll 4
    for (int c = cg; c < kC; c += 8) {
        const T* m = base + (int64_t)c * h * w;
        const float v = (1.f - ly) * ((1.f - lx) * ldf(m, o00) + lx * ldf(m, o01)) + ly * ((1.f - lx) * ldf(m, o10) + lx * ldf(m, o11));
        tile[px][c] = __float2bfloat16(v);
    }
    __syncthreads();
    // 32 pixels x 512 B: 16-byte pieces, pixel-major
    for (int i = tid; i < 32 * (kC / 8); i += kT) {
        const int p = i / (kC / 8), k8 = i - p * (kC / 8);
        const int x = bx * 32 + p;
        if (x < wl)
            *reinterpret_cast<uint4*>(out + ((int64_t)(n * hl + oy) * wl + x) * kC + k8 * 8) = *reinterpret_cast<const uint4*>(&tile[p][k8 * 8]);
    }
}

// the same resize for channel-last features [N, h*w, 256] (what the pixel decoder's last convolution leaves): a pixel's
// channels are contiguous on both sides, so no transpose — 32 lanes x 8 channels per output pixel, 8 pixels per workgroup
__device__ __forceinline__ void ld8(const float* p, float v[8])
{
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const __hip_bfloat16* p, float v[8])
{
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

template <typename T>
__global__ __launch_bounds__(kT) void pool_features_cl_kernel(const T* __restrict__ mf, int64_t mf_bs, __hip_bfloat16* __restrict__ out,
                                                              int h, int w, int hl, int wl)
{
    const int n = blockIdx.y, p = blockIdx.x * 8 + (threadIdx.x >> 5), c = (threadIdx.x & 31) * 8;
    if (p >= hl * wl) return;
    const int oy = p / wl, ox = p - oy * wl;
    const float sy = (float)h / (float)hl, sx = (float)w / (float)wl;
    const float fy = fmaxf(0.f, ((float)oy + 0.5f) * sy - 0.5f);
    const float fx = fmaxf(0.f, ((float)ox + 0.5f) * sx - 0.5f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const T* base = mf + (int64_t)n * mf_bs + c;
    float a[8], b[8], d[8], e[8];
    ld8(base + ((int64_t)y0 * w + x0) * kC, a);
    ld8(base + ((int64_t)y0 * w + x1) * kC, b);
    ld8(base + ((int64_t)y1 * w + x0) * kC, d);
    ld8(base + ((int64_t)y1 * w + x1) * kC, e);
    __hip_bfloat16 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        r[i] = __float2bfloat16((1.f - ly) * ((1.f - lx) * a[i] + lx * b[i]) + ly * ((1.f - lx) * d[i] + lx * e[i]));
    *reinterpret_cast<uint4*>(out + ((int64_t)n * hl * wl + p) * kC + c) = *reinterpret_cast<const uint4*>(r);
}

// ----------------------------------------------------------------------------------------------------------------
// sign(mask_embed . pooled features) -> byte mask, MP rows, per-row open flags
// grid (pixel blocks of PB, N); 4 waves; wave w owns query tiles w, w + 4 of the current 128-query group.
// PB = 32 (round 3; was 128): a workgroup's critical path is PB / 16 dependent (8 loads -> 16 MFMAs -> LDS bytes) steps per wave
// and a level has only HW / PB x N workgroups — with 128-pixel blocks every launch took ~20 us whatever the level (256 / 64 / 16
// workgroups of one wave per SIMD walking 8 steps); 32-pixel blocks give 4x the workgroups and a quarter of the chain.
// ----------------------------------------------------------------------------------------------------------------
template <int PB>
__global__ __launch_bounds__(kT) void mask_head_bits_kernel(const __hip_bfloat16* __restrict__ me, int64_t me_stride_n,
                                                            int64_t me_stride_q, const __hip_bfloat16* __restrict__ pooled,
                                                            const uint8_t* __restrict__ mp_rows, int pad, uint8_t* __restrict__ out,
                                                            int* __restrict__ flags, int Q, int HW)
{
    constexpr int NPT = PB / 16, NPC = PB / 16;                       // pixel tiles / 16-byte output pieces per row
    __shared__ __attribute__((aligned(16))) uint8_t s_out[128][PB + 16];       // [query][pixel]
    __shared__ int s_any[128];
    const int n = blockIdx.y, p0 = blockIdx.x * PB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const __hip_bfloat16* men = me + n * me_stride_n;
    const __hip_bfloat16* pn = pooled + (int64_t)n * HW * kC;
    for (int q0 = 0; q0 < Q; q0 += 128) {
        if (tid < 128) s_any[tid] = 0;
        // A fragments of this wave's two query tiles: lane (query li, k group kg) holds 8 consecutive channels per k step
        bf16x8 a[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = min(q0 + (wave + 4 * t) * 16 + li, Q - 1);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                a[t][ks] = *reinterpret_cast<const bf16x8*>(men + q * me_stride_q + ks * 32 + kg * 8);
        }
        __syncthreads();
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int pos = min(p0 + pt * 16 + li, HW - 1);
            bf16x8 b[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) b[ks] = *reinterpret_cast<const bf16x8*>(pn + (int64_t)pos * kC + ks * 32 + kg * 8);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][ks], b[ks], acc, 0, 0, 0);
                // D: lane holds (query 4 * kg + r, pixel li) of the tile
#pragma unroll
                for (int r = 0; r < 4; ++r) s_out[(wave + 4 * t) * 16 + kg * 4 + r][pt * 16 + li] = acc[r] < 0.f ? 1 : 0;
            }
        }
        __syncthreads();
        // rows out: 128 queries x NPC pieces of 16 bytes; MP rows take their ground-truth bytes instead
        for (int i = tid; i < 128 * NPC; i += kT) {
            const int ql = i / NPC, pc = i % NPC, q = q0 + ql, p = p0 + pc * 16;
            if (q >= Q || p >= HW) continue;
            uint4 v = *reinterpret_cast<const uint4*>(&s_out[ql][pc * 16]);
            if (q < pad) {
                const uint4 g = *reinterpret_cast<const uint4*>(mp_rows + ((int64_t)n * pad + q) * HW + p);
                // bool bytes are 0 / 1 already; normalise anyway
                v.x = g.x & 0x01010101u; v.y = g.y & 0x01010101u; v.z = g.z & 0x01010101u; v.w = g.w & 0x01010101u;
                v.x |= (g.x >> 1 | g.x >> 2 | g.x >> 3 | g.x >> 4 | g.x >> 5 | g.x >> 6 | g.x >> 7) & 0x01010101u;
                v.y |= (g.y >> 1 | g.y >> 2 | g.y >> 3 | g.y >> 4 | g.y >> 5 | g.y >> 6 | g.y >> 7) & 0x01010101u;
                v.z |= (g.z >> 1 | g.z >> 2 | g.z >> 3 | g.z >> 4 | g.z >> 5 | g.z >> 6 | g.z >> 7) & 0x01010101u;
                v.w |= (g.w >> 1 | g.w >> 2 | g.w >> 3 | g.w >> 4 | g.w >> 5 | g.w >> 6 | g.w >> 7) & 0x01010101u;
            }
            *reinterpret_cast<uint4*>(out + ((int64_t)n * Q + q) * HW + p) = v;
            const bool open = v.x != 0x01010101u || v.y != 0x01010101u || v.z != 0x01010101u || v.w != 0x01010101u;
            if (open) s_any[ql] = 1;                                   // benign race: every writer stores 1
        }
        __syncthreads();
        if (tid < 128 && q0 + tid < Q && s_any[tid]) atomicOr(&flags[n * Q + q0 + tid], 1);
        __syncthreads();
    }
}

// rows whose flag stayed 0 are fully masked: attend everywhere (:1780); flags are reset for the next call
__global__ __launch_bounds__(kT) void mask_head_fix_kernel(uint8_t* __restrict__ out, int* __restrict__ flags, int HW)
{
    const int row = blockIdx.x;
    const int f = flags[row];
    __syncthreads();
    if (threadIdx.x == 0) flags[row] = 0;
    if (f) return;
    uint4* dst = reinterpret_cast<uint4*>(out + (int64_t)row * HW);
    for (int i = threadIdx.x; i < HW / 16; i += kT) dst[i] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace

extern "C" int mpf_pool_features(const void* mask_features, int dtype, void* out_bf16, int N, int C, int h, int w, int hl, int wl,
                                 void* stream)
{
    if (!mask_features || !out_bf16) return mpf::fail(MPF_E_NULL, "pool_features: NULL buffer");
    if (C != kC) return mpf::fail(MPF_E_SHAPE, "pool_features: 256 channels only");
    if (N <= 0 || h <= 0 || w <= 0 || hl <= 0 || wl <= 0) return mpf::fail(MPF_E_SHAPE, "pool_features: bad sizes");
    if (dtype != MPF_F32 && dtype != MPF_BF16) return mpf::fail(MPF_E_DTYPE, "pool_features: f32 or bf16 input");
    hipStream_t st = (hipStream_t)stream;
    const int xblocks = (wl + 31) / 32;
    const int64_t grid = (int64_t)N * hl * xblocks;
    if (grid >= (1ll << 31)) return mpf::fail(MPF_E_TOO_LARGE, "pool_features: grid too large");
    mpf::prof_begin(st);
    mpf::set_kernel("pool_features_kernel");
    if (dtype == MPF_F32)
        hipLaunchKernelGGL(pool_features_kernel<float>, dim3((unsigned)grid), dim3(kT), 0, st, (const float*)mask_features,
                           (__hip_bfloat16*)out_bf16, h, w, hl, wl, xblocks);
    else
        hipLaunchKernelGGL(pool_features_kernel<__hip_bfloat16>, dim3((unsigned)grid), dim3(kT), 0, st, (const __hip_bfloat16*)mask_features,
                           (__hip_bfloat16*)out_bf16, h, w, hl, wl, xblocks);
    mpf::prof_end("pool_features_kernel", st, (double)N * kC * ((double)h * w * (dtype == MPF_F32 ? 4 : 2) + (double)hl * wl * 2));
    return mpf::check(hipGetLastError(), "mpf_pool_features");
}

extern "C" int mpf_pool_features_cl(const void* mask_features, int64_t batch_stride, int dtype, void* out_bf16, int N, int C, int h,
                                    int w, int hl, int wl, void* stream)
{
    if (!mask_features || !out_bf16) return mpf::fail(MPF_E_NULL, "pool_features_cl: NULL buffer");
    if (C != kC) return mpf::fail(MPF_E_SHAPE, "pool_features_cl: 256 channels only");
    if (N <= 0 || N > 65535 || h <= 0 || w <= 0 || hl <= 0 || wl <= 0 || batch_stride % 8 != 0)
        return mpf::fail(MPF_E_SHAPE, "pool_features_cl: bad sizes");
    if (dtype != MPF_F32 && dtype != MPF_BF16) return mpf::fail(MPF_E_DTYPE, "pool_features_cl: f32 or bf16 input");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((hl * wl + 7) / 8, N);
    mpf::prof_begin(st);
    mpf::set_kernel("pool_features_cl_kernel");
    if (dtype == MPF_F32)
        hipLaunchKernelGGL(pool_features_cl_kernel<float>, grid, dim3(kT), 0, st, (const float*)mask_features, batch_stride,
                           (__hip_bfloat16*)out_bf16, h, w, hl, wl);
    else
        hipLaunchKernelGGL(pool_features_cl_kernel<__hip_bfloat16>, grid, dim3(kT), 0, st, (const __hip_bfloat16*)mask_features,
                           batch_stride, (__hip_bfloat16*)out_bf16, h, w, hl, wl);
    mpf::prof_end("pool_features_cl_kernel", st, (double)N * kC * ((double)h * w * (dtype == MPF_F32 ? 4 : 2) + (double)hl * wl * 2));
    return mpf::check(hipGetLastError(), "mpf_pool_features_cl");
}

extern "C" int mpf_mask_head_bits(const void* mask_embed, int64_t stride_n, int64_t stride_q, const void* pooled,
                                  const uint8_t* mp_rows, int pad, uint8_t* out, int32_t* flags, int N, int Q, int HW, void* stream)
{
    if (!mask_embed || !pooled || !out || !flags || (pad > 0 && !mp_rows)) return mpf::fail(MPF_E_NULL, "mask_head_bits: NULL buffer");
    if (N <= 0 || Q <= 0 || HW <= 0 || pad < 0 || pad > Q) return mpf::fail(MPF_E_SHAPE, "mask_head_bits: bad sizes");
    if (HW % 16) return mpf::fail(MPF_E_SHAPE, "mask_head_bits: level size must be a multiple of 16 pixels");
    if ((stride_q % 8) || (stride_n % 8)) return mpf::fail(MPF_E_SHAPE, "mask_head_bits: mask_embed rows must be 16-byte aligned");
    if (N > 65535) return mpf::fail(MPF_E_TOO_LARGE, "mask_head_bits: batch > 65535");
    hipStream_t st = (hipStream_t)stream;
    mpf::prof_begin(st);
    mpf::set_kernel("mask_head_bits_kernel");
    constexpr int kPB = 32;
    hipLaunchKernelGGL(mask_head_bits_kernel<kPB>, dim3((HW + kPB - 1) / kPB, N), dim3(kT), 0, st, (const __hip_bfloat16*)mask_embed, stride_n,
                       stride_q, (const __hip_bfloat16*)pooled, mp_rows, pad, out, (int*)flags, Q, HW);
    mpf::prof_end("mask_head_bits_kernel", st, 2.0 * ((double)N * Q * kC + (double)N * HW * kC) + (double)N * Q * HW,
                  2.0 * N * (double)Q * HW * kC);
    hipLaunchKernelGGL(mask_head_fix_kernel, dim3(N * Q), dim3(kT), 0, st, out, (int*)flags, HW);
    return mpf::check(hipGetLastError(), "mpf_mask_head_bits");
}
