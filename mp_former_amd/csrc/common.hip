// Error reporting and option plumbing of the C ABI (include/mpformer_hip.h).
#include "mpf_common.h"

#include <stdio.h>

#include <atomic>

namespace {
thread_local char t_err[256] = "";
// process-wide (autograd runs backward on its own thread)
std::atomic<const char*> g_kernel{""};
}  // namespace

namespace mpf {
int fail(int code, const char* msg)
{
    snprintf(t_err, sizeof(t_err), "%s", msg);
    return code;
}

int check(hipError_t err, const char* where)
{
    if (err == hipSuccess) return 0;
    snprintf(t_err, sizeof(t_err), "%s: %s (%d)", where, hipGetErrorString(err), (int)err);
    return (int)err;
}

void set_kernel(const char* name) { g_kernel.store(name, std::memory_order_relaxed); }
}  // namespace mpf

extern "C" int mpf_abi_version(void) { return 1; }
extern "C" const char* mpf_last_error(void) { return t_err; }
extern "C" const char* mpf_last_kernel(void) { return g_kernel.load(std::memory_order_relaxed); }

extern "C" int mpf_set_option(const char* key, int value)
{
    if (!key) return MPF_E_NULL;
    int r = mpf::set_msda_option(key, value);
    if (r <= 0) return r;
    return mpf::fail(MPF_E_SHAPE, "mpf_set_option: unknown key");
}
