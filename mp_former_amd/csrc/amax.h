// "amax slots": the largest magnitude of a tensor, kept on the device for the fp16 x 2 GEMMs (csrc/gemm3.hip, include/mpformer_hip.h
// MPF_AMAX_SLOT_FLOATS).  Shared by the kernels that produce a slot (GEMM epilogues, LayerNorm, the amax passes) and read it.
#pragma once
#include <hip/hip_runtime.h>

namespace {

// An "amax slot" is kAmaxSub sub-slots kAmaxStride floats apart (different cache lines): producers max into the sub-slot
// blockIdx.x % kAmaxSub with ONE atomic per workgroup (device-scope atomics on one address serialise at ~11 ns each: 8 192 of
// them cost a 44 MB reduction 90 us), consumers take the largest of the 16.  Non-negative floats order like their bit
// patterns and a NaN pattern is larger than inf, so the integer max is order-independent and keeps a NaN visible.
constexpr int kAmaxSub = 16, kAmaxStride = 32;

__device__ __forceinline__ unsigned amax_read(const float* slot)
{
    const unsigned* s = reinterpret_cast<const unsigned*>(slot);
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < kAmaxSub; ++i) m = max(m, s[i * kAmaxStride]);
    return m;
}

// running max |x| of the threads of a workgroup (256 threads) -> one atomic max; red: 4 floats of LDS nobody else is using
// (called by all threads, contains a barrier)
__device__ __forceinline__ void amax_commit(float* slot, float m, float* red)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(reinterpret_cast<unsigned*>(slot) + (blockIdx.x % kAmaxSub) * kAmaxStride, __float_as_uint(m));
    }
}

}  // namespace
