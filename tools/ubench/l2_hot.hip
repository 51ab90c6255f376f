// How fast can every CU re-read the same small region (weight planes) from L2, against private regions?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/l2_hot.hip -o tools/ubench/l2_hot && tools/ubench/l2_hot
// region = 256 KB; each workgroup (256 threads, 1 or 2 per CU) reads it `reps` times with 16-byte loads, 8 in flight per lane.
// mode 0: all workgroups the same region; 1: one region per XCD (blockIdx & 7); 2: one region per workgroup;
// mode 3: same region, each workgroup starting at a different offset (rotated walk)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ base, int region_vec, int reps, int mode, uint4* sink)
{
    const int b = blockIdx.x;
    const uint4* p = base + (size_t)(mode == 1 ? (b & 7) : mode == 2 ? b : 0) * region_vec;
    const int nblk = region_vec / 256;
    const int start = mode == 3 ? (b * 1031) % nblk : 0;
    if (mode >= 4) {
        // the weight-plane access of the GEMM kernels: a wave instruction = 16 rows x 64 B (4 lanes per row), row pitch `pitch`
        // bytes (mode 4: 512 = K 256 fp16; mode 5: 2048 = K 1024); the region is walked as [row group of 16][64-byte column]
        const int pitch = mode == 4 ? 512 : 2048;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int rows = region_vec * 16 / pitch;                 // rows of the region
        const int cols64 = pitch / 64;
        const char* pc = (const char*)base;
        uint4 acc2 = make_uint4(0, 0, 0, 0);
        for (int r = 0; r < reps; ++r) {
            for (int rg = wave; rg < rows / 16; rg += 4) {
                for (int c0 = 0; c0 < cols64; c0 += 8) {
                    uint4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        v[u] = *(const uint4*)(pc + (size_t)(rg * 16 + (lane >> 2)) * pitch + (c0 + u) * 64 + (lane & 3) * 16);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { acc2.x ^= v[u].x + r; acc2.y ^= v[u].y; acc2.z ^= v[u].z; acc2.w ^= v[u].w; }
                }
            }
            asm volatile("" ::: "memory");
        }
        if (acc2.x == 0x12345678u) sink[b] = acc2;
        return;
    }
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r) {
        for (int i0 = 0; i0 < nblk; i0 += 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int blk = i0 + u + start;
                blk = blk >= nblk ? blk - nblk : blk;
                v[u] = p[(size_t)blk * 256 + threadIdx.x];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x + r; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
        asm volatile("" ::: "memory");
    }
    if (acc.x == 0x12345678u) sink[b] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const int region = 256 * 1024, region_vec = region / 16;
    const int nwg_max = 512;
    uint4 *buf, *sink;
    CK(hipMalloc(&buf, (size_t)region * nwg_max));
    CK(hipMemset(buf, 1, (size_t)region * nwg_max));
    CK(hipMalloc(&sink, nwg_max * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nwg : {32, 64, 128, 256, 512}) for (int mode = (nwg < 256 ? 2 : 0); mode < (nwg < 256 ? 3 : 6); ++mode) {
        const int reps = 8;
        rd<<<nwg, 256>>>(buf, region_vec, 2, mode, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        rd<<<nwg, 256>>>(buf, region_vec, reps, mode, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)nwg * region * reps;
        printf("wgs %d mode %d: %.1f us  %.2f TB/s  %.1f B/clk/CU @2.4GHz\n", nwg, mode, ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / (nwg < 256 ? nwg : 256) / 2.4e9);
    }
    return 0;
}
