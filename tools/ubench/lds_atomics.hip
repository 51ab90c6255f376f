// Micro-benchmark: LDS atomic-add throughput (f32 vs u32, conflict patterns) on MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE 0: f32 atomic, random rows (2 rows per wave instr, 32 lanes each)
// MODE 1: u32 atomic, random rows
// MODE 2: f32 atomic, all 64 lanes distinct banks?? (row pair chosen so halves hit different bank halves: impossible with 32 banks -> same as 0 but rows equal => same address both halves? no: lanes j and j+32 same addr => 2 adds same address)
// MODE 3: f32 plain read-modify-write (non-atomic) random rows
// MODE 4: f32 atomic, 64 lanes cover ONE 256-B span (row pair adjacent: rows 2k,2k+1) -> 64 distinct addresses, banks mod 32 collide 2-way
template <int MODE>
__global__ void k(float* out, int iters, int nwin_rows)
{
    extern __shared__ float win[];
    for (int i = threadIdx.x; i < nwin_rows * 32; i += blockDim.x) win[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 5, j = lane & 31;
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) / 64 * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;                 // wave-uniform LCG
        unsigned r0 = (s >> 8) % (unsigned)nwin_rows;
        unsigned r1 = (s >> 18) % (unsigned)nwin_rows;
        unsigned row = sub ? r1 : r0;
        if (MODE == 4) row = ((r0 & ~1u) + sub) % (unsigned)nwin_rows;
        float* p = &win[row * 32 + j];
        if (MODE == 0 || MODE == 4) atomicAdd(p, 1.0f);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(p), 1u);
        else if (MODE == 3) *p += 1.0f;
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = threadIdx.x; i < nwin_rows * 32; i += blockDim.x) acc += win[i];
    if (acc == -1.f) out[0] = acc;
}

template <int MODE>
void run(const char* name, int threads, int rows)
{
    float* out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096, blocks = 256 * (1024 / threads);
    k<MODE><<<blocks, threads, rows * 128>>>(out, 16, rows); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, threads, rows * 128>>>(out, iters, rows);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double lane_ops = (double)blocks * threads * iters;
    printf("%-44s thr=%4d rows=%4d %8.3f ms %8.1f Glane-ops/s  %.2f lanes/clk/CU@2.4GHz\n", name, threads, rows, ms,
           lane_ops / ms / 1e6, lane_ops / ms / 1e6 / 256 / 2.4);
}

int main()
{
    run<0>("ds_add_f32 random rows", 1024, 1024);
    run<1>("ds_add_u32 random rows", 1024, 1024);
    run<3>("plain RMW random rows", 1024, 1024);
    run<4>("ds_add_f32 adjacent row pair", 1024, 1024);
    run<0>("ds_add_f32 random rows", 256, 256);
    run<1>("ds_add_u32 random rows", 256, 256);
    return 0;
}
