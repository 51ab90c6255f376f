// Error reporting and option plumbing of the C ABI (include/mpformer_hip.h).
#include "mpf_common.h"

#include <stdio.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

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

// ---- in-library launch profiler: HIP events recorded on the launch stream around each kernel ----
namespace {
struct ProfRec { const char* name; hipEvent_t e0, e1; double bytes, flops; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
std::atomic<int> g_prof_on{0};
thread_local hipEvent_t t_e0 = nullptr;
}  // namespace

bool prof_enabled() { return g_prof_on.load(std::memory_order_relaxed) != 0; }

void prof_begin(hipStream_t st)
{
    if (!prof_enabled()) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { t_e0 = nullptr; return; }
    (void)hipEventRecord(e, st);
    t_e0 = e;
}

void prof_end(const char* name, hipStream_t st, double algorithmic_bytes, double flops)
{
    if (!prof_enabled() || !t_e0) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({name, t_e0, e, algorithmic_bytes, flops});
    t_e0 = nullptr;
}
}  // namespace mpf

extern "C" int mpf_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(mpf::g_prof_mu);
    for (auto& r : mpf::g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    mpf::g_prof.clear();
    mpf::g_prof_on.store(on ? 1 : 0);
    return 0;
}

extern "C" int mpf_profile_get(const char* name_substr, int* count, double* total_ms, double* total_bytes)
{
    if (!name_substr || !count || !total_ms || !total_bytes) return MPF_E_NULL;
    std::lock_guard<std::mutex> lk(mpf::g_prof_mu);
    int n = 0; double ms = 0, by = 0;
    for (auto& r : mpf::g_prof) {
        if (!strstr(r.name, name_substr)) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
        ms += t; by += r.bytes; ++n;
    }
    *count = n; *total_ms = ms; *total_bytes = by;
    return 0;
}

extern "C" int mpf_profile_get_flops(const char* name_substr, double* total_flops)
{
    if (!name_substr || !total_flops) return MPF_E_NULL;
    std::lock_guard<std::mutex> lk(mpf::g_prof_mu);
    double fl = 0;
    for (auto& r : mpf::g_prof)
        if (strstr(r.name, name_substr)) fl += r.flops;
    *total_flops = fl;
    return 0;
}

// ---- small host -> device tables through the KERNEL ARGUMENT segment ------------------------------------------
// The grouped launches take device tables of (pointer, size, ...) items.  Uploading one through pinned staging memory
// costs the launch thread an allocator round trip + an asynchronous copy and cannot be captured in a HIP graph (the
// staging block is recycled).  Here the bytes travel BY VALUE as a kernel argument (<= kUpPayload per launch) and a
// one-workgroup kernel stores them: one launch, no staging memory, and a captured graph node carries the table itself.
namespace {
constexpr int kUpPayload = 3968;          // bytes per launch (the kernel argument segment is 4 KB)
struct UpPayload { uint32_t w[kUpPayload / 4]; };
__global__ __launch_bounds__(256) void upload_small_kernel(UpPayload p, uint32_t* __restrict__ dst, int nwords)
{
    for (int i = threadIdx.x; i < nwords; i += 256) dst[i] = p.w[i];
}
}  // namespace

extern "C" int mpf_upload_small(const void* host_src, void* device_dst, int64_t nbytes, void* stream)
{
    if (nbytes == 0) return 0;
    if (!host_src || !device_dst) return mpf::fail(MPF_E_NULL, "upload_small: NULL buffer");
    if (nbytes < 0 || (nbytes & 3) || ((uintptr_t)device_dst & 3)) return mpf::fail(MPF_E_SHAPE, "upload_small: size / destination must be multiples of 4 bytes");
    const unsigned char* src = static_cast<const unsigned char*>(host_src);
    unsigned char* dst = static_cast<unsigned char*>(device_dst);
    for (int64_t off = 0; off < nbytes; off += kUpPayload) {
        const int n = (int)((nbytes - off) < kUpPayload ? (nbytes - off) : kUpPayload);
        UpPayload p;
        memcpy(p.w, src + off, (size_t)n);
        hipLaunchKernelGGL(upload_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, reinterpret_cast<uint32_t*>(dst + off), n / 4);
    }
    return mpf::check(hipGetLastError(), "mpf_upload_small");
}

extern "C" int mpf_abi_version(void) { return 1; }
extern "C" const char* mpf_last_error(void) { return t_err; }
extern "C" const char* mpf_last_kernel(void) { return g_kernel.load(std::memory_order_relaxed); }

extern "C" int mpf_set_option(const char* key, int value)
{
    if (!key) return MPF_E_NULL;
    int r = mpf::set_msda_option(key, value);
    if (r <= 0) return r;
    r = mpf::set_binned_option(key, value);
    if (r <= 0) return r;
    r = mpf::set_gemm3_option(key, value);
    if (r <= 0) return r;
    r = mpf::set_decoder_option(key, value);
    if (r <= 0) return r;
    r = mpf::set_block_option(key, value);
    if (r <= 0) return r;
    r = mpf::set_attn_option(key, value);
    if (r <= 0) return r;
    r = mpf::set_small_gemm_option(key, value);
    if (r <= 0) return r;
    return mpf::fail(MPF_E_SHAPE, "mpf_set_option: unknown key");
}
