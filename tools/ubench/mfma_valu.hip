// Can VALU / LDS work hide under an MFMA stream on gfx950 — inside ONE wave (software interleave) and between the TWO waves of a SIMD?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu.hip -o tools/ubench/mfma_valu && tools/ubench/mfma_valu
// One workgroup per CU.  Modes (cycles per loop iteration of ONE wave, s_memtime):
//   0  MFMA only: 12 independent v_mfma_f32_16x16x32_f16 per iteration, 1 wave per SIMD
//   1  VALU only: 36 dependent-free v_fma_f32 per iteration, 1 wave per SIMD
//   2  both interleaved in one wave (3 v_fma after every MFMA), 1 wave per SIMD
//   3  two waves per SIMD: waves 0-3 MFMA only, waves 4-7 VALU only (reports both)
//   4  two waves per SIMD, both run the interleaved stream of mode 2
//   5  mode 2 + 4 ds_read_b128 per iteration (the fragment reads of a K step)
//   6-9  wave pairs with separate loops: MFMA | VALU + LDS reads, MFMA | fp16 x 2 split LDS -> LDS, the split alone, MFMA with its
//        fragment reads | split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(0.5f - i * 0.1f); }
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = lane + i;
    const float m = 1.0001f, c = 0.5f;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = i;
    __syncthreads();
    uint4 fr[4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 3 || MODE == 6 || MODE == 7 || MODE == 8 || MODE == 9) {
        // wave pair, separate loops (wave-uniform branch outside): waves 0-3 stream MFMAs, waves 4-7 run VALU (3), VALU + LDS (6),
        // or the fp16 x 2 split of 8 floats read from and written back to LDS (7)
        if (wave < 4) {
            if (MODE == 8) {
                // (no MFMA partner: waves 0-3 idle)
            } else if (MODE == 9) {
                f16x8 fa[4];
                for (int q = 0; q < 4; ++q) fa[q] = a;
                for (int it = 0; it < iters; ++it) {
                    uint4 nx[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) nx[q] = *reinterpret_cast<const uint4*>(lds + ((it & 7) * 4096 + q * 1024 + lane * 16) % 32768);
#pragma unroll
                    for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i & 3], b, acc[i], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) fa[q] = __builtin_bit_cast(f16x8, nx[q]);
                }
            } else
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            }
        } else if (MODE == 3) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    v[i] = __builtin_fmaf(v[i], m, c);
                    v[(i + 4) % 12] = __builtin_fmaf(v[(i + 4) % 12], m, c);
                    v[(i + 8) % 12] = __builtin_fmaf(v[(i + 8) % 12], m, c);
                }
            }
        } else if (MODE == 6) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 4; ++q) fr[q] = *reinterpret_cast<const uint4*>(lds + ((it & 7) * 4096 + q * 1024 + lane * 16) % 32768);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    v[i] = __builtin_fmaf(v[i], m, c);
                    v[(i + 4) % 12] = __builtin_fmaf(v[(i + 4) % 12], m, c);
                    v[(i + 8) % 12] = __builtin_fmaf(v[(i + 8) % 12], m, c);
                }
                v[0] += __uint_as_float(fr[0].x ^ fr[1].y ^ fr[2].z ^ fr[3].w);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                unsigned char* rowp = lds + ((wave - 4) * 8192 + (it & 3) * 2048);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 u = *reinterpret_cast<const float4*>(rowp + q * 1024 + (lane & 31) * 16 + (lane >> 5) * 512);
                    const float sc = m;
                    _Float16 hh[4], ll[4];
                    const float xs[4] = {u.x * sc, u.y * sc, u.z * sc, u.w * sc};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hh[e] = (_Float16)xs[e]; ll[e] = (_Float16)(xs[e] - (float)hh[e]); }
                    uint2 ph, pl;
                    ph.x = (unsigned)__builtin_bit_cast(unsigned short, hh[0]) | ((unsigned)__builtin_bit_cast(unsigned short, hh[1]) << 16);
                    ph.y = (unsigned)__builtin_bit_cast(unsigned short, hh[2]) | ((unsigned)__builtin_bit_cast(unsigned short, hh[3]) << 16);
                    pl.x = (unsigned)__builtin_bit_cast(unsigned short, ll[0]) | ((unsigned)__builtin_bit_cast(unsigned short, ll[1]) << 16);
                    pl.y = (unsigned)__builtin_bit_cast(unsigned short, ll[2]) | ((unsigned)__builtin_bit_cast(unsigned short, ll[3]) << 16);
                    *reinterpret_cast<uint2*>(rowp + q * 1024 + lane * 8) = ph;
                    *reinterpret_cast<uint2*>(rowp + q * 1024 + 512 + lane * 8) = pl;
                }
            }
        }
    } else {
    const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5;
    const bool do_valu = MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 5) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fr[q] = *reinterpret_cast<const uint4*>(lds + ((it & 7) * 4096 + q * 1024 + lane * 16) % 32768);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (do_mfma) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            if (do_valu) {
                v[i] = __builtin_fmaf(v[i], m, c);
                v[(i + 4) % 12] = __builtin_fmaf(v[(i + 4) % 12], m, c);
                v[(i + 8) % 12] = __builtin_fmaf(v[(i + 8) % 12], m, c);
            }
            if (do_mfma && do_valu) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // 3 VALU
            }
        }
        if (MODE == 5) { a[0] += (_Float16)__uint_as_float(fr[0].x ^ fr[1].y ^ fr[2].z ^ fr[3].w); }
    }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3] + v[i];
    if (s == 1234.5f) sink[threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int MODE>
void run(int threads, const char* what)
{
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 4096);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(out, sink, 10);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, sink, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("mode %d (%s): wave 0 %.1f ticks/iter", MODE, what, (double)h[0] / iters);
    if (threads == 512) printf(", wave 4 %.1f", (double)h[4] / iters);
    printf("; kernel %.1f us -> %.1f ns/iter\n", ms * 1e3, ms * 1e6 / iters);
}

int main()
{
    run<0>(256, "MFMA only, 12 per iter");
    run<1>(256, "VALU only, 36 fma per iter");
    run<2>(256, "interleaved in one wave");
    run<3>(512, "wave pair: MFMA | VALU");
    run<4>(512, "wave pair: both interleaved");
    run<5>(256, "interleaved + 4 ds_read_b128");
    run<6>(512, "wave pair: MFMA | VALU + 4 ds_read_b128");
    run<7>(512, "wave pair: MFMA | split of 2 x 256 floats LDS -> LDS");
    run<8>(512, "waves 4-7 alone: split of 2 x 256 floats LDS -> LDS");
    run<9>(512, "wave pair: MFMA + 4 ds_read_b128 per 12 | split");
    return 0;
}
