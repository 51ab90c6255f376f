// The small arithmetic at the end of SetCriterion (mask2former/modeling/criterion.py): the weighted cross-entropy of the
// class logits of every output (loss_labels, :123-139: F.cross_entropy(logits^T, target_classes, empty_weight) =
// sum_r w[t_r] * nll_r / sum_r w[t_r]) and the reduction of the per-pair point sums to the per-output mask / dice losses
// (dice_loss :21-40, sigmoid_ce_loss :48-65, the / num_masks of :189-190).
//
// Why native: as tensor expressions these are ~100 launches of 2-5 us over vectors of 10-500 elements per step
// (log_softmax, gather, index, mul, sum, div, index_add, ... and the same again in the backward) — at the end of the forward,
// where nothing else can run.  Here: one forward and one backward launch for the class losses of ALL outputs, one of each
// for the mask losses.  Sums run in a fixed order (a wave walks its rows in order, waves are merged in order): the results
// are reproducible run to run.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_add(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldf<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }

struct ClsArgs {
    const void* logits;            // [L, N, Q, C] with element strides sl, sn, sq, 1
    const int64_t* target;         // [L or 1, N, Q] contiguous (tl = N * Q or 0)
    const float* weight;           // [C]
    float* lse;                    // [L, N * Q]   log-sum-exp of every row (forward -> backward)
    float* ce;                     // [L]
    float* wsum;                   // [L]          sum of the rows' class weights
    int64_t sl, sn, sq, tl;
    int L, N, Q, C;
};

// lanes cover the classes (C <= 64 * kCls)
constexpr int kCls = 4;            // classes per lane: C <= 256

// one wave per row: log-sum-exp, and the row's weighted nll / weight into rowv [L * rows][2] (a per-output loop over its
// 228 rows in one wave was a chain of dependent loads: 37 us)
template <typename T>
__global__ __launch_bounds__(256) void class_loss_rows_kernel(ClsArgs a, float* __restrict__ rowv)
{
    const int lane = threadIdx.x & 63;
    const int rows = a.N * a.Q;
    const int64_t gr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gr >= (int64_t)a.L * rows) return;
    const int l = (int)(gr / rows), r = (int)(gr - (int64_t)l * rows);
    const int n = r / a.Q, q = r - n * a.Q;
    const T* row = static_cast<const T*>(a.logits) + (int64_t)l * a.sl + (int64_t)n * a.sn + (int64_t)q * a.sq;
    float x[kCls];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < kCls; ++k) {
        const int c = lane + 64 * k;
        x[k] = c < a.C ? ldf<T>(row + min(c, a.C - 1)) : -INFINITY;
        mx = fmaxf(mx, x[k]);
    }
    // a label outside [0, C) (F.cross_entropy device-asserts on it): reads stay in range and the row's loss is NaN, so the
    // step fails visibly instead of reading out of bounds
    const int64_t t_raw = a.target[(int64_t)l * a.tl + r];
    const bool t_ok = t_raw >= 0 && t_raw < a.C;
    const int64_t t = t_ok ? t_raw : 0;
    const float xt = ldf<T>(row + t);
    const float w = t_ok ? a.weight[t] : __builtin_nanf("");
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < kCls; ++k) se += (lane + 64 * k < a.C) ? __expf(x[k] - mx) : 0.f;
    se = wave_add(se);
    const float lse = mx + __logf(se);
    if (lane == 0) {
        a.lse[gr] = lse;
        rowv[2 * gr] = w * (lse - xt);
        rowv[2 * gr + 1] = w;
    }
}

// one wave per output: lane j adds rows j, j + 64, ... in order, then the butterfly (the same order every run)
__global__ __launch_bounds__(64) void class_loss_reduce_kernel(const float* __restrict__ rowv, int rows, float* __restrict__ ce,
                                                               float* __restrict__ wsum)
{
    const int l = blockIdx.x, lane = threadIdx.x;
    float s = 0.f, w = 0.f;
    for (int r = lane; r < rows; r += 64) {
        const float2 v = *reinterpret_cast<const float2*>(rowv + 2 * ((int64_t)l * rows + r));
        s += v.x;
        w += v.y;
    }
    s = wave_add(s);
    w = wave_add(w);
    if (lane == 0) { ce[l] = s / w; wsum[l] = w; }
}

// d logits[l, n, q, c] = g[l] / wsum[l] * w[t] * (softmax_c - [c == t]); one wave per row, dense [L, N, Q, C] output in T
template <typename T>
__global__ __launch_bounds__(256) void class_loss_bwd_kernel(ClsArgs a, const float* __restrict__ g, T* __restrict__ dlogits)
{
    const int lane = threadIdx.x & 63;
    const int rows = a.N * a.Q;
    const int64_t gr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gr >= (int64_t)a.L * rows) return;
    const int l = (int)(gr / rows), r = (int)(gr - (int64_t)l * rows);
    const int n = r / a.Q, q = r - n * a.Q;
    const T* row = static_cast<const T*>(a.logits) + (int64_t)l * a.sl + (int64_t)n * a.sn + (int64_t)q * a.sq;
    const int64_t t_raw = a.target[(int64_t)l * a.tl + r];
    const bool t_ok = t_raw >= 0 && t_raw < a.C;
    const int64_t t = t_ok ? t_raw : 0;
    const float lse = a.lse[gr];
    const float s = t_ok ? g[l] / a.wsum[l] * a.weight[t] : __builtin_nanf("");
    T* out = dlogits + gr * a.C;
#pragma unroll
    for (int k = 0; k < kCls; ++k) {
        const int c = lane + 64 * k;
        if (c < a.C) {
            const float p = __expf(ldf<T>(row + c) - lse);
            const float v = s * (p - (c == (int)t ? 1.f : 0.f));
            if constexpr (sizeof(T) == 2) out[c] = __float2bfloat16(v);
            else out[c] = v;
        }
    }
}

// ---- mask / dice: sums[i] = (sum BCE, sum sig * t, sum sig, sum t) of pair i; the pairs of group g are the run
// [run[2g], run[2g] + run[2g + 1]) (the criterion lays the pairs out per output: matched pairs, then MP pairs) ------------
struct MaskFin {
    const float* sums;      // [n, 4]
    const int64_t* run;     // [G, 2] (first pair, count)
    const float* norm;      // [G]
    int n, G;
    float inv_p;            // 1 / points per pair
};

// one wave per group: lane j takes pairs first + j, first + j + 64, ...; butterfly sum (same order every run)
__global__ __launch_bounds__(64) void mask_finalize_fwd_kernel(MaskFin a, float* __restrict__ out /* [2, G]: mask, dice */)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    const int first = (int)a.run[2 * g], cnt = (int)a.run[2 * g + 1];
    float m = 0.f, d = 0.f;
    for (int i = lane; i < cnt; i += 64) {
        const float4 s = *reinterpret_cast<const float4*>(a.sums + 4 * (int64_t)(first + i));
        m += s.x * a.inv_p;
        d += 1.f - (2.f * s.y + 1.f) / (s.z + s.w + 1.f);
    }
    m = wave_add(m);
    d = wave_add(d);
    if (lane == 0) {
        out[g] = m / a.norm[g];
        out[a.G + g] = d / a.norm[g];
    }
}

// one wave per group again (the group of a pair is its block)
__global__ __launch_bounds__(64) void mask_finalize_bwd_kernel(MaskFin a, const float* __restrict__ gout /* [2, G] */,
                                                               float* __restrict__ dsums /* [n, 4] */)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    const int first = (int)a.run[2 * g], cnt = (int)a.run[2 * g + 1];
    const float gm = gout[g] / a.norm[g], gd = gout[a.G + g] / a.norm[g];
    for (int i = lane; i < cnt; i += 64) {
        const float4 s = *reinterpret_cast<const float4*>(a.sums + 4 * (int64_t)(first + i));
        const float den = s.z + s.w + 1.f, num = 2.f * s.y + 1.f;
        float4 o;
        o.x = gm * a.inv_p;
        o.y = gd * (-2.f / den);
        o.z = gd * (num / (den * den));
        o.w = o.z;                      // (the targets carry no gradient; kept for the formula's symmetry)
        *reinterpret_cast<float4*>(dsums + 4 * (int64_t)(first + i)) = o;
    }
}

}  // namespace

extern "C" int mpf_class_loss_forward(const void* logits, int dtype, int64_t sl, int64_t sn, int64_t sq, const int64_t* target,
                                      int target_per_output, const float* weight, int L, int N, int Q, int C, float* lse, float* ce,
                                      float* wsum, void* stream)
{
    if (L == 0) return 0;
    if (!logits || !target || !weight || !lse || !ce || !wsum) return mpf::fail(MPF_E_NULL, "class_loss_forward: NULL buffer");
    if (L < 0 || N <= 0 || Q <= 0 || C <= 0 || C > 64 * kCls) return mpf::fail(MPF_E_SHAPE, "class_loss_forward: needs 1 <= C <= 256");
    ClsArgs a{logits, target, weight, lse, ce, wsum, sl, sn, sq, target_per_output ? (int64_t)N * Q : 0, L, N, Q, C};
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)L * N * Q;
    float* rowv = lse + rows;                      // lse is [3, L, N * Q]: log-sum-exp, then (w * nll, w) per row
    const dim3 grid((unsigned)((rows + 3) / 4));
    mpf::set_kernel("class_loss_rows_kernel");
    if (dtype == MPF_BF16) hipLaunchKernelGGL(class_loss_rows_kernel<__hip_bfloat16>, grid, dim3(256), 0, st, a, rowv);
    else if (dtype == MPF_F32) hipLaunchKernelGGL(class_loss_rows_kernel<float>, grid, dim3(256), 0, st, a, rowv);
    else return mpf::fail(MPF_E_DTYPE, "class_loss_forward: logits must be MPF_F32 or MPF_BF16");
    hipLaunchKernelGGL(class_loss_reduce_kernel, dim3(L), dim3(64), 0, st, (const float*)rowv, N * Q, ce, wsum);
    return mpf::check(hipGetLastError(), "mpf_class_loss_forward");
}

extern "C" int mpf_class_loss_backward(const void* logits, int dtype, int64_t sl, int64_t sn, int64_t sq, const int64_t* target,
                                       int target_per_output, const float* weight, int L, int N, int Q, int C, const float* lse,
                                       const float* wsum, const float* grad_ce, void* dlogits, void* stream)
{
    if (L == 0) return 0;
    if (!logits || !target || !weight || !lse || !wsum || !grad_ce || !dlogits)
        return mpf::fail(MPF_E_NULL, "class_loss_backward: NULL buffer");
    if (L < 0 || N <= 0 || Q <= 0 || C <= 0 || C > 64 * kCls) return mpf::fail(MPF_E_SHAPE, "class_loss_backward: needs 1 <= C <= 256");
    ClsArgs a{logits, target, weight, const_cast<float*>(lse), nullptr, const_cast<float*>(wsum), sl, sn, sq,
              target_per_output ? (int64_t)N * Q : 0, L, N, Q, C};
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)L * N * Q;
    const dim3 grid((unsigned)((rows + 3) / 4));
    mpf::set_kernel("class_loss_bwd_kernel");
    if (dtype == MPF_BF16)
        hipLaunchKernelGGL(class_loss_bwd_kernel<__hip_bfloat16>, grid, dim3(256), 0, st, a, grad_ce, (__hip_bfloat16*)dlogits);
    else if (dtype == MPF_F32) hipLaunchKernelGGL(class_loss_bwd_kernel<float>, grid, dim3(256), 0, st, a, grad_ce, (float*)dlogits);
    else return mpf::fail(MPF_E_DTYPE, "class_loss_backward: logits must be MPF_F32 or MPF_BF16");
    return mpf::check(hipGetLastError(), "mpf_class_loss_backward");
}

extern "C" int mpf_mask_loss_finalize(const float* sums, const int64_t* runs, const float* norm, int n, int G, int points,
                                      float* out, void* stream)
{
    if (G == 0) return 0;
    if (!norm || !out || !runs || (n > 0 && !sums)) return mpf::fail(MPF_E_NULL, "mask_loss_finalize: NULL buffer");
    if (n < 0 || G < 0 || points <= 0) return mpf::fail(MPF_E_SHAPE, "mask_loss_finalize: bad sizes");
    MaskFin a{sums, runs, norm, n, G, 1.f / (float)points};
    mpf::set_kernel("mask_finalize_fwd_kernel");
    hipLaunchKernelGGL(mask_finalize_fwd_kernel, dim3(G), dim3(64), 0, (hipStream_t)stream, a, out);
    return mpf::check(hipGetLastError(), "mpf_mask_loss_finalize");
}

extern "C" int mpf_mask_loss_finalize_backward(const float* sums, const int64_t* runs, const float* norm, int n, int G, int points,
                                               const float* grad_out, float* dsums, void* stream)
{
    if (n == 0) return 0;
    if (!sums || !runs || !norm || !grad_out || !dsums) return mpf::fail(MPF_E_NULL, "mask_loss_finalize_backward: NULL buffer");
    if (n < 0 || G <= 0 || points <= 0) return mpf::fail(MPF_E_SHAPE, "mask_loss_finalize_backward: bad sizes");
    MaskFin a{sums, runs, norm, n, G, 1.f / (float)points};
    mpf::set_kernel("mask_finalize_bwd_kernel");
    hipLaunchKernelGGL(mask_finalize_bwd_kernel, dim3(G), dim3(64), 0, (hipStream_t)stream, a, grad_out, dsums);
    return mpf::check(hipGetLastError(), "mpf_mask_loss_finalize_backward");
}
