// Micro-benchmark: the MFMA + fragment-read skeleton of the split-bf16 GEMM (gemm3.hip Acc::step) on an LDS-resident tile,
// with / without the two barriers of a K step, at 1-3 workgroups per CU, and with the fragment reads of the next column tile
// issued before the MFMAs of the current one.  Answers: what is the ceiling of ds_read_b128 + 96 MFMAs per wave and K step?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include tools/ubench/gemm3_skeleton.hip -o tools/ubench/gemm3_skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../mp_former_amd/csrc/gemm3.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

namespace {

// MODE bit 0: barrier before the step, bit 1: a second barrier, bit 2: software-pipelined fragment reads (variant step)
template <int MODE, int OCC>
__global__ __launch_bounds__(kThreads, OCC) void skel_kernel(float* out, int nk)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[12 * kAKc + 2 * 12 * 128 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (int)sizeof(lds) / 4; i += kThreads) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const int wr = wave >> 1, wc = wave & 1;
    Acc<4> acc;
    acc.zero();
    const int a_frag = wr * 64 * 16, b_frag = 12 * kAKc + wc * 64 * 16;
    for (int kt = 0; kt < nk; ++kt) {
        if (MODE & 1) __syncthreads();
        if (MODE & 2) __syncthreads();
        if (MODE & 4) {
            // variant: all 12 A fragments, then B fragments of tile j + 1 requested BEFORE the MFMAs of tile j
            const int r16 = lane & 15, g = lane >> 4;
            const int af = a_frag + g * kAKc + (r16 ^ (2 * g)) * 16, bf = b_frag + g * (128 * 16) + (r16 ^ (2 * g)) * 16;
            bf16x8 fa[3][4], fb[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[pl][i] = as_frag(*reinterpret_cast<const uint4*>(lds + af + pl * 4 * kAKc + i * 256));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[0][pl] = as_frag(*reinterpret_cast<const uint4*>(lds + bf + pl * 4 * (128 * 16)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j + 1 < 4) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fb[(j + 1) & 1][pl] = as_frag(*reinterpret_cast<const uint4*>(lds + bf + pl * 4 * (128 * 16) + (j + 1) * 256));
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8* b = fb[j & 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc.v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], fa[2][i], acc.v[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc.v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2], fa[0][i], acc.v[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc.v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], fa[1][i], acc.v[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc.v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], fa[1][i], acc.v[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc.v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], fa[0][i], acc.v[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc.v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], fa[0][i], acc.v[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE & 8) {
            // MFMA only: fragments read once, outside the loop (registers)
            bf16x8 f = as_frag(*reinterpret_cast<const uint4*>(lds + a_frag + lane * 16));
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc.v[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, acc.v[i][j], 0, 0, 0);
        } else if (MODE & 16) {     // row-major B image (the DMA layout), alternating stages
            acc.template step<kAKc, 128 * 16, false, false, true>(lds, a_frag, 12 * kAKc + (kt & 1) * 12 * 2048 + wc * 64 * 64, lane);
        } else {
            acc.template step<kAKc, 128 * 16>(lds, a_frag, b_frag, lane);
        }
    }
    acc.quads(lane, [&](int mo, int no, float4 o) {
        if (o.x == 12345.678f) out[mo * 128 + no] = o.x + o.y + o.z + o.w;
    });
}

template <int MODE, int OCC>
void run(const char* name, int wgs_per_cu)
{
    float* out;
    CK(hipMalloc(&out, 1 << 20));
    const int nk = 512, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((skel_kernel<MODE, OCC>), dim3(grid), dim3(kThreads), 0, 0, out, 8);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((skel_kernel<MODE, OCC>), dim3(grid), dim3(kThreads), 0, 0, out, nk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid * nk * 384.0 * 16384.0;
    printf("%-58s %d WG/CU: %8.1f us  %6.3f PFLOP/s bf16  (%4.1f %% of 2.5)\n", name, wgs_per_cu, best * 1e3, flops / (best * 1e-3) / 1e15,
           flops / (best * 1e-3) / 2.5e15 * 100);
    CK(hipFree(out));
}

}  // namespace

int main()
{
    run<8, 2>("MFMA only (registers)", 2);
    run<8, 2>("MFMA only (registers)", 1);
    run<0, 2>("Acc::step (ds_read_b128 + 96 MFMA), no barrier", 2);
    run<0, 2>("Acc::step, no barrier", 1);
    run<0, 3>("Acc::step, no barrier, launch bound 3", 3);
    run<3, 2>("Acc::step, two barriers per step", 2);
    run<1, 2>("Acc::step, one barrier per step", 2);
    run<4, 2>("pipelined B fragments, no barrier", 2);
    run<7, 2>("pipelined B fragments, two barriers", 2);
    run<4, 2>("pipelined B fragments, no barrier", 1);
    run<16, 2>("Acc::step with the row-major (DMA) B image, no barrier", 2);
    run<19, 2>("Acc::step with the row-major (DMA) B image, two barriers", 2);
    return 0;
}
