// Shared host-side helpers of libmpformer_hip.so (error reporting, last-kernel tracking).
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>

#include <atomic>

#include "../../include/mpformer_hip.h"

namespace mpf {
// record an argument error (negative code) and return it
int fail(int code, const char* msg);
// map a hipError_t to the ABI return value (0 on success), recording the message
int check(hipError_t err, const char* where);
void set_kernel(const char* name);
// launch profiler (mpf_profile_enable): events on the launch stream around the kernel only
void prof_begin(hipStream_t st);
void prof_end(const char* name, hipStream_t st, double algorithmic_bytes, double flops = 0.0);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) and size class instead of before every launch: the
// call takes a runtime lock (~10-30 us of launch-thread time each, 55 per training step when issued per launch).  The attribute
// is per device, so the high-water mark is kept per device ordinal; `slots` is the call site's own static array.
constexpr int kMaxDevices = 32;
struct LdsAttr { std::atomic<int> bytes[kMaxDevices] = {}; };     // (racing first launches from two host threads set the same attribute twice: harmless)
inline int ensure_dynamic_lds(const void* fn, size_t bytes, LdsAttr& slots)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
    if (dev >= 0 && slots.bytes[dev].load(std::memory_order_relaxed) >= (int)bytes) return 0;
    if (int e = check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), "hipFuncSetAttribute")) return e;
    if (dev >= 0) slots.bytes[dev].store((int)bytes, std::memory_order_relaxed);
    return 0;
}
// compute units of the current device (cached per device ordinal; 256 when the query fails)
inline int cu_count()
{
    static std::atomic<int> cache[kMaxDevices];
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    cus = cache[dev].load(std::memory_order_relaxed);
    if (cus > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cache[dev].store(cus, std::memory_order_relaxed);
    return cus;
}
// per-subsystem option hooks: return 0 if handled, 1 if the key is not theirs, <0 on bad value
int set_msda_option(const char* key, int v);
int set_binned_option(const char* key, int v);
int set_gemm3_option(const char* key, int v);
int set_block_option(const char* key, int v);
int set_decoder_option(const char* key, int v);
int set_attn_option(const char* key, int v);
int set_small_gemm_option(const char* key, int v);
// spatially blocked MSDA (msda_block.hip); return -1000 when the problem is outside their shapes
// raw != NULL: loc / attn are outputs computed from the raw projection + reference points (msda_prep fused in)
int msda_block_forward(const void* value, const int64_t* host_shapes, const void* loc, const void* attn, void* out, int N, int S, int M,
                       int D, int L, int Lq, int P, int dtype, hipStream_t st, const void* raw = nullptr, const void* ref = nullptr);
size_t msda_block_workspace_bytes(const int64_t* host_shapes, int N, int M, int L, int Lq, int P);
int msda_block_backward(const void* value, const int64_t* host_shapes, const void* loc, const void* attn, const void* go, void* gv,
                        void* gl, void* ga, void* graw, int N, int S, int M, int D, int L, int Lq, int P, int dtype, void* workspace,
                        size_t workspace_bytes, hipStream_t st, const void* fwd_out = nullptr, float* graw_amax = nullptr,
                        float* gv_amax = nullptr, bool* amax_recorded = nullptr);
}  // namespace mpf
