// Shared host-side helpers of libmpformer_hip.so (error reporting, last-kernel tracking).
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/mpformer_hip.h"

namespace mpf {
// record an argument error (negative code) and return it
int fail(int code, const char* msg);
// map a hipError_t to the ABI return value (0 on success), recording the message
int check(hipError_t err, const char* where);
void set_kernel(const char* name);
// launch profiler (mpf_profile_enable): events on the launch stream around the kernel only
void prof_begin(hipStream_t st);
void prof_end(const char* name, hipStream_t st, double algorithmic_bytes);
// per-subsystem option hooks: return 0 if handled, 1 if the key is not theirs, <0 on bad value
int set_msda_option(const char* key, int v);
int set_binned_option(const char* key, int v);
int set_gemm3_option(const char* key, int v);
}  // namespace mpf
