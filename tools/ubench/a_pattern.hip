// What does the A-operand access pattern of gemm3_tn3_kernel cost by itself?  (round 6)
// The kernel's workgroup (512 threads) owns 192 rows of A [M, K] fp32 and per 32-deep K step reads a 128-byte piece of each row:
// thread (row t / 8 + 64 u, 16-byte part t % 8), u = 0..2, two steps ahead.  Variants, all reading the same M x K x 4 bytes once:
//   0  that pattern                                   (rows 4 K bytes apart, 128 B per row and step)
//   1  the same rows, FOUR steps per burst             (512 contiguous bytes per row every fourth step)
//   2  the workgroup's rows as one linear stream       (its 192 x K floats are contiguous in memory)
//   3  pattern 0 with 176 rows per workgroup (245 workgroups)
// prints GB/s per variant for K = 1024 and K = 256, M = 43008.   hipcc -O3 --offload-arch=gfx950 a_pattern.hip -o a_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int V>
__global__ __launch_bounds__(512) void read_kernel(const float* __restrict__ a, int M, int K, int rows_per_wg, float* __restrict__ out)
{
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * rows_per_wg;
    if (m0 >= M) return;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    auto add = [&](const float4 v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; };
    if (V == 0 || V == 3) {
        const int srow = tid >> 3, shc = tid & 7;
        const float* ap[3];
        for (int u = 0; u < 3; ++u) ap[u] = a + (size_t)min(m0 + min(srow + 64 * u, rows_per_wg - 1), M - 1) * K + shc * 4;
        const int nk = K / 32;
        float4 r0[3], r1[3];
        for (int u = 0; u < 3; ++u) r0[u] = *reinterpret_cast<const float4*>(ap[u]);
        for (int u = 0; u < 3; ++u) r1[u] = *reinterpret_cast<const float4*>(ap[u] + min(32, (nk - 1) * 32));
        for (int k = 0; k < nk; k += 2) {
            for (int u = 0; u < 3; ++u) add(r0[u]);
            for (int u = 0; u < 3; ++u) r0[u] = *reinterpret_cast<const float4*>(ap[u] + min((k + 2) * 32, (nk - 1) * 32));
            __builtin_amdgcn_s_barrier();
            for (int u = 0; u < 3; ++u) add(r1[u]);
            for (int u = 0; u < 3; ++u) r1[u] = *reinterpret_cast<const float4*>(ap[u] + min((k + 3) * 32, (nk - 1) * 32));
            __builtin_amdgcn_s_barrier();
        }
    } else if (V == 1) {
        // 192 rows x 512 B per burst = 6144 float4 = 12 per thread: thread (row t / 32 + 16 u, part t % 32)
        const int prow = tid >> 5, part = tid & 31;
        const int nb = K / 128;
        float4 r[12], rn[12];
        for (int u = 0; u < 12; ++u) r[u] = *reinterpret_cast<const float4*>(a + (size_t)min(m0 + prow + 16 * u, M - 1) * K + part * 4);
        for (int b = 0; b < nb; ++b) {
            const int bn = min(b + 1, nb - 1);
            for (int u = 0; u < 12; ++u) rn[u] = *reinterpret_cast<const float4*>(a + (size_t)min(m0 + prow + 16 * u, M - 1) * K + bn * 128 + part * 4);
            for (int u = 0; u < 12; ++u) add(r[u]);
            for (int s = 0; s < 4; ++s) __builtin_amdgcn_s_barrier();
            for (int u = 0; u < 12; ++u) r[u] = rn[u];
        }
    } else {
        const size_t total4 = (size_t)min(rows_per_wg, M - m0) * K / 4;
        const float4* p = reinterpret_cast<const float4*>(a + (size_t)m0 * K);
        for (size_t i = tid; i < total4; i += 512 * 4) {
            float4 v[4];
            for (int j = 0; j < 4; ++j) v[j] = p[min(i + (size_t)j * 512, total4 - 1)];
            for (int j = 0; j < 4; ++j) add(v[j]);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

template <int V>
float run(const float* a, int M, int K, int rows, float* out)
{
    const int grid = (M + rows - 1) / rows;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(read_kernel<V>, dim3(grid), dim3(512), 0, 0, a, M, K, rows, out);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(read_kernel<V>, dim3(grid), dim3(512), 0, 0, a, M, K, rows, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}

int main()
{
    const int M = 43008;
    float *a, *out;
    hipMalloc(&a, (size_t)M * 1024 * 4); hipMalloc(&out, 4096);
    hipMemset(a, 0, (size_t)M * 1024 * 4);
    for (int K : {1024, 256}) {
        const double mb = (double)M * K * 4 / 1e6;
        const float t0 = run<0>(a, M, K, 192, out), t1 = run<1>(a, M, K, 192, out), t2 = run<2>(a, M, K, 192, out), t3 = run<3>(a, M, K, 176, out);
        const float t2b = run<2>(a, M, K, 168, out);
        printf("K=%4d  %.1f MB   tn3 pattern %.1f us (%.2f TB/s) | 4-step bursts %.1f us (%.2f) | linear stream %.1f us (%.2f) | tn3 pattern, 176 rows %.1f us (%.2f) | linear, 168 rows (256 workgroups) %.1f us (%.2f)\n",
               K, mb, t0, mb / t0 / 1e0 * 1e-6 * 1e6 / 1e6, t1, mb / t1, t2, mb / t2, t3, mb / t3, t2b, mb / t2b);
    }
    return 0;
}
