// Fused per-channel bias (+ residual) + ReLU epilogue for channel-last activations (the folded
// FrozenBN shift of the bench backbone): y = relu(x + bias[c] + res), one pass instead of three
// (MIOpen's separate bias kernel, the residual add, the ReLU).  bf16 or fp32 activations, fp32 bias.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

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
        if (res) r = res[i];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // same rounding points as the unfused ops: (x + b) -> bf16, (+ res) -> bf16
            float t = (float)(__bf16)((float)v[j] + b[j]);
            if (res) t = (float)(__bf16)(t + (float)r[j]);
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

}  // namespace

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
        hipLaunchKernelGGL(bias_act_bf16_kernel, dim3(blocks), dim3(256), 0, st, (const bf16x8*)x, bias, (const bf16x8*)res,
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
// dst[dst_offs[i] + p] = (T) src[i * plane + p]: the fp32 gradient planes that the point-sampled mask
// loss accumulated (one scratch plane per (map, image, query) pair) -> their places in the dense
// gradient of the prediction maps, cast to the maps' dtype.  Planes not named stay as the caller
// initialised them (zero).
// ------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void planes_scatter_kernel(const float4* __restrict__ src, const int64_t* __restrict__ dst_offs,
                                                             T* __restrict__ dst, int plane4)
{
    const int i = blockIdx.y;
    const float4* s = src + (int64_t)i * plane4;
    T* d = dst + dst_offs[i];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < plane4; p += gridDim.x * 256) {
        const float4 v = s[p];
        if constexpr (sizeof(T) == 4) {
            reinterpret_cast<float4*>(d)[p] = v;
        } else {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            reinterpret_cast<bf16x4*>(d)[p] = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        }
    }
}

}  // namespace

extern "C" int mpf_planes_scatter(const float* src, const int64_t* dst_offs, void* dst, int dst_dtype, int n, int plane,
                                  void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    if (!src || !dst_offs || !dst) return mpf::fail(MPF_E_NULL, "planes_scatter: NULL buffer");
    if (n < 0 || plane <= 0 || plane % 4 != 0) return mpf::fail(MPF_E_SHAPE, "planes_scatter: plane must be a positive multiple of 4");
    const int bx = (plane / 4 + 255) / 256 < 64 ? (plane / 4 + 255) / 256 : 64;
    if (dst_dtype == MPF_BF16) {
        mpf::set_kernel("planes_scatter_kernel<bf16>");
        hipLaunchKernelGGL(planes_scatter_kernel<__bf16>, dim3(bx, n), dim3(256), 0, st, (const float4*)src, dst_offs, (__bf16*)dst, plane / 4);
    } else if (dst_dtype == MPF_F32) {
        mpf::set_kernel("planes_scatter_kernel<float>");
        hipLaunchKernelGGL(planes_scatter_kernel<float>, dim3(bx, n), dim3(256), 0, st, (const float4*)src, dst_offs, (float*)dst, plane / 4);
    } else {
        return mpf::fail(MPF_E_DTYPE, "planes_scatter: dst dtype must be MPF_F32 or MPF_BF16");
    }
    return mpf::check(hipGetLastError(), "mpf_planes_scatter");
}
