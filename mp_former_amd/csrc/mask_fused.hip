// Fused mask product + point-sampled matching cost / loss planes for MI355X (bf16 MFMA, gfx950).
//
// Reference: mask2former_transformer_decoder.py:1865-1870
//     outputs_mask = einsum("bqc,bchw->bqhw", mask_embed, mask_features)            [N, Qtot, H/4, W/4] per decoder output
// consumed ONLY through point samples: matcher.py:120-132 (P points per (output, image), shared by its queries) and
// criterion.py:141-191 (3 P candidate + P final points per matched / mask-piloted (prediction, target) pair).
// The ten [N, Qtot, H/4, W/4] maps (300 MB per step at config B) are never formed here:
//
//   * matching cost (mpf_match_cost_fused).  Bilinear sampling is linear in the map and the map is linear in the features:
//         logit(q, p) = sum_c E[q, c] * bilinear(F[:, :, c])(p)
//     so a workgroup gathers the four corner rows (512 B, channel-last) of 32 points, interpolates them in fp32 and keeps the
//     result as TWO bf16 planes (value and rounding remainder: 16 significand bits) in LDS; the [128 queries x 256] x
//     [256 x 32 points] product runs on v_mfma_f32_16x16x32_bf16 with the embeddings as register-resident A fragments; the
//     logits go through LDS to a (query, target-half) thread mapping that accumulates sum_p x t, sum_p sigmoid(x) t,
//     sum_p softplus(x), sum_p sigmoid(x) over the workgroup's tiles in registers (fp32, matcher.py:20-62).  Partial sums
//     per workgroup are reduced in a FIXED order by a second launch: no atomics, bit-reproducible costs.
//
//   * loss planes (mpf_pair_planes_forward).  Importance sampling draws 3 P = 37 632 DIFFERENT points per pair — more point
//     evaluations than the plane has pixels — so for the ~500 (prediction, target) pairs of a step (of 2 280 rows) the plane
//     IS the cheaper form: out[slot] = E[row(slot)] . F^T as one MFMA launch over gathered embedding rows, 64 MB instead of
//     300 MB, produced after the assignment so that only paired rows exist.  The same kernel with the identity row list is
//     the full product for callers of the reference interface that want a pred_masks tensor.
//
//   * backward of the planes (mpf_pair_planes_backward): d F = G^T E (K = slots) and d E = G F (K = pixels, split over
//     pixel ranges with a fixed-order reduction).  Both have one operand whose contraction index is the slow one in memory
//     (G [slot][pixel] for d F, F [pixel][channel] for d E); it is transposed on its way into LDS — 16-byte global loads of
//     two adjacent rows, v_perm_b32 pairs, 4-byte LDS stores into a [column][64 k] image whose 16-byte slots are XOR-swizzled
//     so that both the stores (2 x 32 lanes, 32 banks) and the ds_read_b128 fragment reads (MI355X_MICROARCH.md: four
//     non-contiguous 16-lane groups, 64 banks) are conflict-free.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include <algorithm>

#include "mpf_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int kC = 256;           // mask dimension (contraction length of the product)
constexpr int kT = 256;

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);
    return (unsigned)(*reinterpret_cast<const unsigned short*>(&x)) | ((unsigned)(*reinterpret_cast<const unsigned short*>(&y)) << 16);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// sum over the 64 lanes (DPP row operations + two readlanes); result wave-uniform
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add_(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum64(float v)
{
    v = dpp_add_<0xB1>(v);
    v = dpp_add_<0x4E>(v);
    v = dpp_add_<0x141>(v);
    v = dpp_add_<0x140>(v);
    v = dpp_add_<0x142, 0xa>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ====================================================================================================================
// 1. plane product over gathered embedding rows
//    out[slot][px] = sum_c E[row_off[pair_of_slot[slot]] + c] * F[b][px][c]          bf16 in, fp32 accumulate, bf16 out
//    grid (ceil(HW / 128), N), 4 waves x 32 pixels; the wave's pixel rows are its A fragments for the whole launch
// ====================================================================================================================
__global__ __launch_bounds__(kT) void pair_planes_fwd_kernel(const __hip_bfloat16* __restrict__ embed, const int64_t* __restrict__ row_off,
                                                             const int32_t* __restrict__ pair_of_slot, const int32_t* __restrict__ slot_first,
                                                             const int32_t* __restrict__ slot_count, const __hip_bfloat16* __restrict__ feat,
                                                             int64_t feat_bs, __hip_bfloat16* __restrict__ out, int HW)
{
    const int b = blockIdx.y, p0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int first = slot_first[b], cnt = slot_count[b];
    if (cnt <= 0) return;
    const __hip_bfloat16* fb = feat + (int64_t)b * feat_bs;
    bf16x8 a[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int px = min(p0 + wave * 32 + t * 16 + li, HW - 1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) a[t][ks] = *reinterpret_cast<const bf16x8*>(fb + (int64_t)px * kC + ks * 32 + kg * 8);
    }
    auto load = [&](int s0, bf16x8 (&dst)[8]) {
        const int s = first + min(s0 + li, cnt - 1);
        const int pr = pair_of_slot ? pair_of_slot[s] : s;
        const __hip_bfloat16* e = embed + row_off[pr] + kg * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(e + ks * 32);
    };
    auto compute = [&](int s0, const bf16x8 (&src)[8]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][ks], src[ks], acc, 0, 0, 0);
            // D: lane holds (pixel 4 kg + r, slot li): four consecutive pixels of one plane -> one 8-byte store
            const int p = p0 + wave * 32 + t * 16 + kg * 4;
            if (s0 + li < cnt && p < HW)
                *reinterpret_cast<uint2*>(out + (int64_t)(first + s0 + li) * HW + p) = make_uint2(pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3]));
        }
    };
    // two named register sets, the next slot tile's rows in flight under the current tile's MFMAs
    bf16x8 X[8], Y[8];
    load(0, X);
    for (int s0 = 0; s0 < cnt; s0 += 32) {
        load(s0 + 16, Y);
        compute(s0, X);
        load(s0 + 32, X);
        if (s0 + 16 < cnt) compute(s0 + 16, Y);
    }
}

// ====================================================================================================================
// 2. transposing stage: T[j][k] <- X[k0 + k][j0 + j],  k = 0..63, j = 0..J-1  (bf16), 256 threads
//    LDS image: row j = 128 B (64 k), its eight 16-byte slots XOR-swizzled by hs(j) = ((j >> 1) ^ (j >> 4)) & 7.
//    A task = (k pair kp, 8-column chunk cc): two 16-byte row pieces -> eight dwords (k, k + 1) for columns 8 cc + i.
//    Lane order inside a wave: kp low bits fastest (4 pairs), then 8 / 16 chunks: the 32 lanes of a store group hit 32
//    different banks ((4 (kp_hi ^ hs) + kp_lo) is a bijection of (kp_lo, cc & 7)); a global instruction reads 256-byte runs.
// ====================================================================================================================
__device__ __forceinline__ int hs_of(int j) { return ((j >> 1) ^ (j >> 4)) & 7; }

template <int J>
struct StageT {
    static constexpr int kChunks = J / 8;                  // 16 or 32
    static constexpr int kTasks = 32 * kChunks;            // 512 or 1024
    static constexpr int kPer = kTasks / kT;               // 2 or 4 tasks per thread
    uint4 ra[kPer], rb[kPer];

    // rows k0 + 2 kp, k0 + 2 kp + 1 of X (leading dimension ld elements), columns j0 + 8 cc ..; rows are clamped to max_row
    __device__ __forceinline__ void load(const __hip_bfloat16* __restrict__ X, int64_t ld, int k0, int max_row, int j0, int tid)
    {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int t = tid + u * kT;
            const int kp = (t & 3) + 4 * ((t >> 2) / kChunks), cc = (t >> 2) % kChunks;
            const int r0 = min(k0 + 2 * kp, max_row), r1 = min(k0 + 2 * kp + 1, max_row);
            ra[u] = *reinterpret_cast<const uint4*>(X + (int64_t)r0 * ld + j0 + cc * 8);
            rb[u] = *reinterpret_cast<const uint4*>(X + (int64_t)r1 * ld + j0 + cc * 8);
        }
    }
    __device__ __forceinline__ void store(unsigned char* T, int tid) const
    {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int t = tid + u * kT;
            const int kp = (t & 3) + 4 * ((t >> 2) / kChunks), cc = (t >> 2) % kChunks;
            const unsigned a[4] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w}, b[4] = {rb[u].x, rb[u].y, rb[u].z, rb[u].w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // element i of both rows -> one dword (k even in the low half)
                const unsigned v = (i & 1) ? __builtin_amdgcn_perm(b[i >> 1], a[i >> 1], 0x07060302u)
                                           : __builtin_amdgcn_perm(b[i >> 1], a[i >> 1], 0x05040100u);
                const int j = cc * 8 + i;
                *reinterpret_cast<unsigned*>(T + j * 128 + (((kp >> 2) ^ hs_of(j)) << 4) + ((kp & 3) << 2)) = v;
            }
        }
    }
};
// fragment (row j, contraction k = 32 ks + 8 kg .. + 7) of the image
__device__ __forceinline__ bf16x8 frag_T(const unsigned char* T, int j, int ks, int kg)
{
    return *reinterpret_cast<const bf16x8*>(T + j * 128 + (((ks * 4 + kg) ^ hs_of(j)) << 4));
}

// Et[b][c][k] = E[row of slot first_b + k][c] (zero for k >= count_b), k pitch KP (a multiple of 64): the K-contiguous
// copy of the gathered embedding rows that d F contracts over.  grid (KP / 32, N), thread = channel
__global__ __launch_bounds__(kT) void pair_embed_transpose_kernel(const __hip_bfloat16* __restrict__ embed, const int64_t* __restrict__ row_off,
                                                                  const int32_t* __restrict__ pair_of_slot, const int32_t* __restrict__ slot_first,
                                                                  const int32_t* __restrict__ slot_count, __hip_bfloat16* __restrict__ Et, int KP)
{
    const int b = blockIdx.y, k0 = blockIdx.x * 32, c = threadIdx.x;
    const int first = slot_first[b], cnt = slot_count[b];
    unsigned w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + 2 * i + h;
            const int s = cnt > 0 ? first + min(k, cnt - 1) : 0;             // (an image without pairs reads slot 0: always valid)
            const int pr = pair_of_slot ? pair_of_slot[s] : s;
            const float x = cnt > 0 ? __bfloat162float(embed[row_off[pr] + c]) : 0.f;
            v[h] = k < cnt ? x : 0.f;
        }
        w[i] = pack_bf16(v[0], v[1]);            // (values are bf16 already: exact)
    }
    uint4* dst = reinterpret_cast<uint4*>(Et + ((int64_t)b * kC + c) * KP + k0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ====================================================================================================================
// 3. d F[b][px][c] = sum_slot G[slot][px] * E[row(slot)][c]           grid (HW / 128, N)
//    A = Et (rows = channels, from global / L2), B = transposed G tile (rows = pixels, LDS); wave w owns channels
//    64 w .. 64 w + 63 (4 row tiles) x the 128 pixels (8 column tiles); D lane = (channel 4 kg + r, pixel li)
// ====================================================================================================================
__global__ __launch_bounds__(kT) void pair_planes_dfeat_kernel(const __hip_bfloat16* __restrict__ G, const __hip_bfloat16* __restrict__ Et, int KP,
                                                               const int32_t* __restrict__ slot_first, const int32_t* __restrict__ slot_count,
                                                               __hip_bfloat16* __restrict__ dF, int64_t df_bs, int HW)
{
    __shared__ __attribute__((aligned(16))) unsigned char sT[2][128 * 128];
    const int b = blockIdx.y, p0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int first = slot_first[b], cnt = slot_count[b];
    __hip_bfloat16* out = dF + (int64_t)b * df_bs;
    f32x4 acc[4][8];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = (cnt + 63) >> 6;
    StageT<128> st;
    const __hip_bfloat16* Gb = G + (int64_t)first * HW;
    if (nk > 0) st.load(Gb, HW, 0, cnt - 1, p0, tid);
    for (int kt = 0; kt < nk; ++kt) {
        unsigned char* T = sT[kt & 1];
        st.store(T, tid);
        __syncthreads();
        if (kt + 1 < nk) st.load(Gb, HW, (kt + 1) * 64, cnt - 1, p0, tid);
        const __hip_bfloat16* ea = Et + ((int64_t)b * kC + wave * 64 + li) * KP + kt * 64 + kg * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(ea + (int64_t)mt * 16 * KP + ks * 32);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const bf16x8 bfr = frag_T(T, nt * 16 + li, ks, kg);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bfr, acc[mt][nt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const int px = p0 + nt * 16 + li;
        if (px < HW) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                *reinterpret_cast<uint2*>(out + (int64_t)px * kC + wave * 64 + mt * 16 + kg * 4) =
                    make_uint2(pack_bf16(acc[mt][nt][0], acc[mt][nt][1]), pack_bf16(acc[mt][nt][2], acc[mt][nt][3]));
        }
    }
}

// ====================================================================================================================
// 4. d E partial sums:  part[kc][slot][c] = sum_{px in chunk kc} G[slot][px] * F[b][px][c]
//    grid (KS pixel chunks, ceil(max count / 128), N); A = G rows (global, contraction contiguous), B = transposed F tile
//    (LDS); wave w owns slots 32 w .. 32 w + 31 of the group (2 row tiles) x 256 channels (16 column tiles)
// ====================================================================================================================
__global__ __launch_bounds__(kT) void pair_planes_dembed_kernel(const __hip_bfloat16* __restrict__ G, const __hip_bfloat16* __restrict__ feat,
                                                                int64_t feat_bs, const int32_t* __restrict__ slot_first,
                                                                const int32_t* __restrict__ slot_count, float* __restrict__ part,
                                                                int total_slots, int HW, int px_per_chunk)
{
    __shared__ __attribute__((aligned(16))) unsigned char sT[2][256 * 128];
    const int kc = blockIdx.x, mg = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int first = slot_first[b], cnt = slot_count[b];
    if (mg * 128 >= cnt) return;
    const __hip_bfloat16* fb = feat + (int64_t)b * feat_bs;
    const int px0 = kc * px_per_chunk, nk = px_per_chunk >> 6;
    f32x4 acc[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __hip_bfloat16* ga[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int s = min(mg * 128 + wave * 32 + mt * 16 + li, cnt - 1);
        ga[mt] = G + (int64_t)(first + s) * HW + px0 + kg * 8;
    }
    StageT<256> st;
    st.load(fb, kC, px0, HW - 1, 0, tid);
    for (int kt = 0; kt < nk; ++kt) {
        unsigned char* T = sT[kt & 1];
        st.store(T, tid);
        __syncthreads();
        if (kt + 1 < nk) st.load(fb, kC, px0 + (kt + 1) * 64, HW - 1, 0, tid);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(ga[mt] + kt * 64 + ks * 32);
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
                const bf16x8 bfr = frag_T(T, nt * 16 + li, ks, kg);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bfr, acc[mt][nt], 0, 0, 0);
            }
        }
    }
    // D lane = (slot 4 kg + r, channel li)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = mg * 128 + wave * 32 + mt * 16 + kg * 4 + r;
            if (s < cnt) {
                float* dst = part + ((int64_t)kc * total_slots + first + s) * kC + li;
#pragma unroll
                for (int nt = 0; nt < 16; ++nt) dst[nt * 16] = acc[mt][nt][r];
            }
        }
}

// d E rows: fixed-order sum of the pixel-chunk partials, written (bf16 or fp32) to the embedding row of the slot's pair.
// One workgroup per slot, thread = channel.  PRECONDITION (include/mpformer_hip.h, mpf_pair_planes_backward): the embedding
// rows of the valid slots are DISTINCT — a query is matched at most once per output, so this holds by construction on the
// training path and the python wrapper asserts it for host-built pair lists.  A row paired twice would be overwritten, not summed.
template <typename OT>
__global__ __launch_bounds__(kT) void pair_dembed_reduce_kernel(const float* __restrict__ part, int KS, int total_slots,
                                                                const int64_t* __restrict__ row_off, const int32_t* __restrict__ pair_of_slot,
                                                                const int32_t* __restrict__ slot_valid, OT* __restrict__ d_embed)
{
    const int s = blockIdx.x, c = threadIdx.x;
    if (!slot_valid[s]) return;
    float v = 0.f;
    for (int k = 0; k < KS; ++k) v += part[((int64_t)k * total_slots + s) * kC + c];
    const int pr = pair_of_slot ? pair_of_slot[s] : s;
    d_embed[row_off[pr] + c] = (OT)v;
}
template <>
__global__ __launch_bounds__(kT) void pair_dembed_reduce_kernel<__hip_bfloat16>(const float* __restrict__ part, int KS, int total_slots,
                                                                                const int64_t* __restrict__ row_off,
                                                                                const int32_t* __restrict__ pair_of_slot,
                                                                                const int32_t* __restrict__ slot_valid,
                                                                                __hip_bfloat16* __restrict__ d_embed)
{
    const int s = blockIdx.x, c = threadIdx.x;
    if (!slot_valid[s]) return;
    float v = 0.f;
    for (int k = 0; k < KS; ++k) v += part[((int64_t)k * total_slots + s) * kC + c];
    const int pr = pair_of_slot ? pair_of_slot[s] : s;
    d_embed[row_off[pr] + c] = __float2bfloat16(v);
}

// slot_valid[s] = 1 for the slots [first_b, first_b + count_b) of every image (the blocks may be padded)
__global__ void slot_valid_kernel(const int32_t* __restrict__ slot_first, const int32_t* __restrict__ slot_count, int N, int total_slots,
                                  int32_t* __restrict__ valid)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total_slots) return;
    int v = 0;
    for (int b = 0; b < N; ++b) v |= (s >= slot_first[b] && s < slot_first[b] + slot_count[b]) ? 1 : 0;
    valid[s] = v;
}

// ====================================================================================================================
// 5. matching cost from the factors
// ====================================================================================================================
// row sums of the pre-sampled ground-truth points: tsum[row] = sum_p tsamp[row][p]; one workgroup per row, fixed order
__global__ __launch_bounds__(kT) void tsamp_rowsum_kernel(const float* __restrict__ tsamp, float* __restrict__ tsum, int rows, int P)
{
    __shared__ float red[kT / 64];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* src = tsamp + (int64_t)row * P;
    float v = 0.f;
    if ((P & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        for (int i = tid; i < (P >> 2); i += kT) { const float4 t = s4[i]; v += (t.x + t.y) + (t.z + t.w); }
    } else {
        for (int p = tid; p < P; p += kT) v += src[p];
    }
    v = wave_sum64(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) tsum[row] = (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int kMP = 32;           // points per tile
constexpr int kMQ = 128;          // queries per workgroup (8 row tiles)
// LDS: [0, 16K) interpolated features, high planes [32 pt][512 B]; [16K, 32K) low planes; then X [128][33] f32; then TV
constexpr int kOffLo = kMP * 512, kOffX = 2 * kMP * 512, kOffTV = kOffX + kMQ * 33 * 4;

template <int TCH>
__global__ __launch_bounds__(kT) void match_cost_fused_kernel(
    const __hip_bfloat16* __restrict__ embed, const int64_t* __restrict__ embed_first, int64_t embed_row_stride,
    const __hip_bfloat16* __restrict__ feat, int64_t feat_bs, const int32_t* __restrict__ group_image, int h, int w,
    const float* __restrict__ coords, const float* __restrict__ tsamp, const int32_t* __restrict__ t_first,
    const int32_t* __restrict__ t_count, float* __restrict__ part_qt, float* __restrict__ part_q, int G, int Q, int Tmax, int P,
    int tiles_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sX = reinterpret_cast<float*>(smem + kOffX);
    float* sTV = reinterpret_cast<float*>(smem + kOffTV);          // [32 points][2 TCH]
    const int g = blockIdx.y, qg = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int Tn = t_count[g];
    if (Tn <= 0) return;
    const int T0 = t_first[g];
    const __hip_bfloat16* fb = feat + (int64_t)group_image[g] * feat_bs;
    const float2* cg = reinterpret_cast<const float2*>(coords) + (int64_t)g * P;
    const int ntiles = (P + kMP - 1) / kMP;
    const int tile0 = blockIdx.x * tiles_per_wg, tile1 = min(tile0 + tiles_per_wg, ntiles);

    // A fragments: the wave's two query tiles, all 8 contraction steps (embedding rows, register-resident)
    bf16x8 a[2][8];
    {
        const __hip_bfloat16* e0 = embed + embed_first[g];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = min(qg * kMQ + (wave * 2 + t) * 16 + li, Q - 1);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) a[t][ks] = *reinterpret_cast<const bf16x8*>(e0 + (int64_t)q * embed_row_stride + ks * 32 + kg * 8);
        }
    }
    // accumulation role: thread = (query ql, target half th)
    const int ql = tid & (kMQ - 1), th = tid >> 7;
    float ax[TCH], as_[TCH], sp = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < TCH; ++j) { ax[j] = 0.f; as_[j] = 0.f; }

    for (int tile = tile0; tile < tile1; ++tile) {
        const int pbase = tile * kMP;
        // ---- gather role: thread = (point tid >> 3, 16-byte slots (tid & 7) + 8 j): four corner rows, fp32 interpolation
        {
            const int pt = tid >> 3, cl = tid & 7;
            const int p = pbase + pt;
            const float2 xy = cg[min(p, P - 1)];
            const float x = xy.x * (float)w - 0.5f, y = xy.y * (float)h - 0.5f;
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = (int)xf, y0 = (int)yf;
            const float lx = x - xf, ly = y - yf;
            const bool live = p < P;
            const bool x0v = live && x0 >= 0 && x0 < w, x1v = live && x0 + 1 >= 0 && x0 + 1 < w;
            const bool y0v = y0 >= 0 && y0 < h, y1v = y0 + 1 >= 0 && y0 + 1 < h;
            const float w00 = (y0v && x0v) ? (1.f - ly) * (1.f - lx) : 0.f, w01 = (y0v && x1v) ? (1.f - ly) * lx : 0.f;
            const float w10 = (y1v && x0v) ? ly * (1.f - lx) : 0.f, w11 = (y1v && x1v) ? ly * lx : 0.f;
            const int xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1);
            const int ya = min(max(y0, 0), h - 1), yb = min(max(y0 + 1, 0), h - 1);
            const __hip_bfloat16* r00 = fb + ((int64_t)ya * w + xa) * kC + cl * 8;
            const __hip_bfloat16* r01 = fb + ((int64_t)ya * w + xb) * kC + cl * 8;
            const __hip_bfloat16* r10 = fb + ((int64_t)yb * w + xa) * kC + cl * 8;
            const __hip_bfloat16* r11 = fb + ((int64_t)yb * w + xb) * kC + cl * 8;
            uint4 v00[4], v01[4], v10[4], v11[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v00[j] = *reinterpret_cast<const uint4*>(r00 + j * 64); v01[j] = *reinterpret_cast<const uint4*>(r01 + j * 64);
                v10[j] = *reinterpret_cast<const uint4*>(r10 + j * 64); v11[j] = *reinterpret_cast<const uint4*>(r11 + j * 64);
            }
            // the tile's target samples [32 points][2 TCH] (zero beyond the group's targets / the last point)
            // (2 TCH * 32 values = TCH / 4 whole passes of the 256 threads: unconditional loads on clamped indices)
#pragma unroll
            for (int u = 0; u < TCH / 4; ++u) {
                const int i = tid + u * kT;
                const int pp = i & (kMP - 1), tt = i >> 5;
                const float v = tsamp[(int64_t)(T0 + min(tt, Tn - 1)) * P + min(pbase + pp, P - 1)];
                sTV[pp * (2 * TCH) + tt] = (tt < Tn && pbase + pp < P) ? v : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned c00[4] = {v00[j].x, v00[j].y, v00[j].z, v00[j].w}, c01[4] = {v01[j].x, v01[j].y, v01[j].z, v01[j].w};
                const unsigned c10[4] = {v10[j].x, v10[j].y, v10[j].z, v10[j].w}, c11[4] = {v11[j].x, v11[j].y, v11[j].z, v11[j].w};
                unsigned hi[4], lo[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float e = w00 * bf_lo(c00[k]) + w01 * bf_lo(c01[k]) + w10 * bf_lo(c10[k]) + w11 * bf_lo(c11[k]);
                    const float o = w00 * bf_hi(c00[k]) + w01 * bf_hi(c01[k]) + w10 * bf_hi(c10[k]) + w11 * bf_hi(c11[k]);
                    hi[k] = pack_bf16(e, o);
                    lo[k] = pack_bf16(e - bf_lo(hi[k]), o - bf_hi(hi[k]));
                }
                const int slot = (cl + 8 * j) ^ (pt & 15);
                *reinterpret_cast<uint4*>(smem + pt * 512 + slot * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                *reinterpret_cast<uint4*>(smem + kOffLo + pt * 512 + slot * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
        }
        __syncthreads();
        // ---- product role: logits of the wave's 32 queries at the 32 points -> LDS [query][point]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            const unsigned char* row = smem + (nt * 16 + li) * 512;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int so = ((ks * 4 + kg) ^ li) << 4;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(row + so);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(row + kOffLo + so);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][ks], bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][ks], bl, acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sX[((wave * 2 + t) * 16 + kg * 4 + r) * 33 + nt * 16 + li] = acc[t][r];
        }
        __syncthreads();
        // ---- accumulation role
        const int pn = min(kMP, P - pbase);
        for (int p = 0; p < pn; ++p) {
            const float x = sX[ql * 33 + p];
            const float e = __expf(-fabsf(x));
            const float s_ = (x >= 0.f ? 1.f : e) * __frcp_rn(1.f + e);
            if (th == 0) { sp += fmaxf(x, 0.f) + __logf(1.f + e); ss += s_; }
            const float4* tv = reinterpret_cast<const float4*>(sTV + p * (2 * TCH) + th * TCH);
#pragma unroll
            for (int j4 = 0; j4 < TCH / 4; ++j4) {
                const float4 t4 = tv[j4];
                ax[j4 * 4 + 0] += x * t4.x; as_[j4 * 4 + 0] += s_ * t4.x;
                ax[j4 * 4 + 1] += x * t4.y; as_[j4 * 4 + 1] += s_ * t4.y;
                ax[j4 * 4 + 2] += x * t4.z; as_[j4 * 4 + 2] += s_ * t4.z;
                ax[j4 * 4 + 3] += x * t4.w; as_[j4 * 4 + 3] += s_ * t4.w;
            }
        }
        __syncthreads();          // X / TV / feature planes are rewritten by the next tile
    }
    const int q = qg * kMQ + ql;
    if (q < Q) {
        float* dst = part_qt + (((int64_t)blockIdx.x * G + g) * Q + q) * Tmax * 2;
#pragma unroll
        for (int j = 0; j < TCH; ++j) {
            const int t = th * TCH + j;
            if (t < Tn) { dst[t * 2] = ax[j]; dst[t * 2 + 1] = as_[j]; }
        }
        if (th == 0) {
            float* dq = part_q + (((int64_t)blockIdx.x * G + g) * Q + q) * 2;
            dq[0] = sp; dq[1] = ss;
        }
    }
}

// cost[g][q][t] = w_mask (sum softplus(x) - sum x t) / P + w_dice (1 - (2 sum s t + 1) / (sum s + sum t + 1))   (matcher.py:20-62)
__global__ __launch_bounds__(kT) void match_cost_reduce_kernel(const float* __restrict__ part_qt, const float* __restrict__ part_q,
                                                               const float* __restrict__ tsum, const int32_t* __restrict__ t_first,
                                                               const int32_t* __restrict__ t_count, float* __restrict__ cost, int nwg, int G,
                                                               int Q, int Tmax, int P, float w_mask, float w_dice)
{
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (int64_t)G * Q * Tmax) return;
    const int t = (int)(i % Tmax), q = (int)((i / Tmax) % Q), g = (int)(i / ((int64_t)Tmax * Q));
    if (t >= t_count[g]) return;
    float sx = 0.f, sst = 0.f, sp = 0.f, ss = 0.f;
    for (int k = 0; k < nwg; ++k) {
        const int64_t r = ((int64_t)k * G + g) * Q + q;
        sx += part_qt[(r * Tmax + t) * 2]; sst += part_qt[(r * Tmax + t) * 2 + 1];
        sp += part_q[r * 2]; ss += part_q[r * 2 + 1];
    }
    const float st = tsum[t_first[g] + t];
    cost[i] = w_mask * (sp - sx) / (float)P + w_dice * (1.f - (2.f * sst + 1.f) / (ss + st + 1.f));
}

struct McGeom {
    int ntiles, tiles_per_wg, nwg, qgroups;
};
McGeom mc_geom(int G, int Q, int P)
{
    McGeom m;
    m.ntiles = (P + kMP - 1) / kMP;
    m.qgroups = (Q + kMQ - 1) / kMQ;
    // two workgroups fit a CU (208 VGPRs): ONE full round of 512 (768 workgroups = 1.5 rounds measured 8 % slower)
    const int want = std::max(1, 512 / std::max(1, G * m.qgroups));
    m.tiles_per_wg = (m.ntiles + want - 1) / want;
    m.nwg = (m.ntiles + m.tiles_per_wg - 1) / m.tiles_per_wg;
    return m;
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <int TCH>
void launch_mc(dim3 grid, hipStream_t st, const __hip_bfloat16* embed, const int64_t* embed_first, int64_t ers, const __hip_bfloat16* feat,
               int64_t feat_bs, const int32_t* gi, int h, int w, const float* coords, const float* tsamp, const int32_t* t_first,
               const int32_t* t_count, float* pqt, float* pq, int G, int Q, int Tmax, int P, int tpw)
{
    const size_t lds = kOffTV + (size_t)kMP * 2 * TCH * 4;
    // > 64 KB of dynamic LDS needs the attribute; it is PER DEVICE, so it is set (and checked) on every launch like loss.hip does
    if (lds > 48 * 1024 &&
        mpf::check(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_cost_fused_kernel<TCH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                   "match_cost_fused: hipFuncSetAttribute") != 0)
        return;
    hipLaunchKernelGGL(match_cost_fused_kernel<TCH>, grid, dim3(kT), lds, st, embed, embed_first, ers, feat, feat_bs, gi, h, w, coords, tsamp,
                       t_first, t_count, pqt, pq, G, Q, Tmax, P, tpw);
}

}  // namespace

// --------------------------------------------------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------------------------------------------------
extern "C" size_t mpf_match_cost_fused_workspace_bytes(int G, int Q, int Tmax, int P, int tsamp_rows)
{
    if (G <= 0 || Q <= 0 || Tmax <= 0 || P <= 0 || tsamp_rows < 0) return 0;
    const McGeom m = mc_geom(G, Q, P);
    return align256((size_t)m.nwg * G * Q * Tmax * 2 * 4) + align256((size_t)m.nwg * G * Q * 2 * 4) + align256((size_t)tsamp_rows * 4);
}

extern "C" int mpf_match_cost_fused(const void* embed, const int64_t* embed_first, int64_t embed_row_stride, const void* features,
                                    int64_t feat_img_stride, const int32_t* group_image, int h, int w, int channels, const float* coords,
                                    const float* tsamp, int tsamp_rows, const int32_t* t_first, const int32_t* t_count, float* cost, int G,
                                    int Q, int Tmax, int P, float w_mask, float w_dice, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!embed || !embed_first || !features || !group_image || !coords || !tsamp || !t_first || !t_count || !cost || !workspace)
        return mpf::fail(MPF_E_NULL, "match_cost_fused: NULL buffer");
    if (channels != kC) return mpf::fail(MPF_E_SHAPE, "match_cost_fused: 256 channels only");
    if (G <= 0 || Q <= 0 || Tmax <= 0 || P <= 0 || h <= 0 || w <= 0 || tsamp_rows <= 0) return mpf::fail(MPF_E_SHAPE, "match_cost_fused: bad sizes");
    if (Tmax > 128) return mpf::fail(MPF_E_SHAPE, "match_cost_fused: at most 128 targets per image");
    if ((embed_row_stride % 8) || (feat_img_stride % 8)) return mpf::fail(MPF_E_SHAPE, "match_cost_fused: rows must be 16-byte aligned");
    if (G > 65535 || (int64_t)G * Q * Tmax >= (1ll << 31)) return mpf::fail(MPF_E_TOO_LARGE, "match_cost_fused: too many groups");
    if (workspace_bytes < mpf_match_cost_fused_workspace_bytes(G, Q, Tmax, P, tsamp_rows))
        return mpf::fail(MPF_E_SHAPE, "match_cost_fused: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const McGeom m = mc_geom(G, Q, P);
    char* ws = (char*)workspace;
    float* pqt = (float*)ws;
    float* pq = (float*)(ws + align256((size_t)m.nwg * G * Q * Tmax * 2 * 4));
    float* tsum = (float*)((char*)pq + align256((size_t)m.nwg * G * Q * 2 * 4));
    const __hip_bfloat16* e = (const __hip_bfloat16*)embed;
    const __hip_bfloat16* f = (const __hip_bfloat16*)features;
    hipLaunchKernelGGL(tsamp_rowsum_kernel, dim3(tsamp_rows), dim3(kT), 0, st, tsamp, tsum, tsamp_rows, P);
    const dim3 grid(m.nwg, G, m.qgroups);
    const int need = (Tmax + 1) / 2;
    mpf::prof_begin(st);
    mpf::set_kernel("match_cost_fused_kernel");
#define MPF_MC(TCH) launch_mc<TCH>(grid, st, e, embed_first, embed_row_stride, f, feat_img_stride, group_image, h, w, coords, tsamp, t_first, \
                                   t_count, pqt, pq, G, Q, Tmax, P, m.tiles_per_wg)
    if (need <= 4) MPF_MC(4);
    else if (need <= 8) MPF_MC(8);
    else if (need <= 12) MPF_MC(12);
    else if (need <= 16) MPF_MC(16);
    else if (need <= 24) MPF_MC(24);
    else if (need <= 32) MPF_MC(32);
    else MPF_MC(64);
#undef MPF_MC
    // gathers: 4 corner rows of 512 B per point; flops: the two-plane product
    mpf::prof_end("match_cost_fused_kernel", st, (double)G * m.qgroups * P * 2048.0 + (double)tsamp_rows * P * 4.0,
                  2.0 * 2.0 * (double)G * Q * P * kC);
    const int64_t n = (int64_t)G * Q * Tmax;
    hipLaunchKernelGGL(match_cost_reduce_kernel, dim3((unsigned)((n + kT - 1) / kT)), dim3(kT), 0, st, pqt, pq, tsum, t_first, t_count, cost,
                       m.nwg, G, Q, Tmax, P, w_mask, w_dice);
    return mpf::check(hipGetLastError(), "mpf_match_cost_fused");
}

extern "C" int mpf_pair_planes_forward(const void* embed, const int64_t* row_off, const int32_t* pair_of_slot, const int32_t* slot_first,
                                       const int32_t* slot_count, const void* features, int64_t feat_img_stride, void* out, int N, int HW,
                                       int channels, void* stream)
{
    if (!embed || !row_off || !slot_first || !slot_count || !features || !out) return mpf::fail(MPF_E_NULL, "pair_planes_forward: NULL buffer");
    if (channels != kC) return mpf::fail(MPF_E_SHAPE, "pair_planes_forward: 256 channels only");
    if (N <= 0 || N > 65535 || HW <= 0 || (HW % 4) || (feat_img_stride % 8)) return mpf::fail(MPF_E_SHAPE, "pair_planes_forward: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    mpf::prof_begin(st);
    mpf::set_kernel("pair_planes_fwd_kernel");
    hipLaunchKernelGGL(pair_planes_fwd_kernel, dim3((HW + 127) / 128, N), dim3(kT), 0, st, (const __hip_bfloat16*)embed, row_off, pair_of_slot,
                       slot_first, slot_count, (const __hip_bfloat16*)features, feat_img_stride, (__hip_bfloat16*)out, HW);
    mpf::prof_end("pair_planes_fwd_kernel", st, 2.0 * (double)N * HW * kC);
    return mpf::check(hipGetLastError(), "mpf_pair_planes_forward");
}

namespace {
struct PbGeom {
    int KP, KS, px_per_chunk, mgroups;
};
PbGeom pb_geom(int N, int HW, int max_count)
{
    PbGeom p;
    p.KP = std::max(64, (max_count + 63) / 64 * 64);
    p.mgroups = std::max(1, (max_count + 127) / 128);
    // pixel chunks: enough workgroups to fill the chip, at most 64 partial planes; a chunk is a multiple of 64 pixels
    int ks = std::max(1, std::min(64, 512 / std::max(1, N * p.mgroups)));
    while (ks > 1 && (HW % (ks * 64))) --ks;
    p.KS = ks;
    p.px_per_chunk = HW / ks;
    return p;
}
}  // namespace

extern "C" size_t mpf_pair_planes_backward_workspace_bytes(int N, int HW, int total_slots, int max_count)
{
    if (N <= 0 || HW <= 0 || total_slots <= 0 || max_count <= 0) return 0;
    const PbGeom p = pb_geom(N, HW, max_count);
    return align256((size_t)N * kC * p.KP * 2) + align256((size_t)p.KS * total_slots * kC * 4) + align256((size_t)total_slots * 4);
}

// grad_planes [total_slots, HW] bf16 (rows outside the images' slot ranges are ignored) ->
//   d_features [N][HW][256] bf16 (fully written), d_embed rows at the same offsets as `embed` (embed_dtype MPF_BF16 / MPF_F32; only
//   the paired rows are written: the caller zero-fills)
extern "C" int mpf_pair_planes_backward(const void* grad_planes, const void* embed, const int64_t* row_off, const int32_t* pair_of_slot,
                                        const int32_t* slot_first, const int32_t* slot_count, const void* features, int64_t feat_img_stride,
                                        void* d_features, int64_t dfeat_img_stride, void* d_embed, int d_embed_dtype, int N, int HW, int channels,
                                        int total_slots, int max_count, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!grad_planes || !embed || !row_off || !slot_first || !slot_count || !features || !workspace)
        return mpf::fail(MPF_E_NULL, "pair_planes_backward: NULL buffer");
    if (channels != kC) return mpf::fail(MPF_E_SHAPE, "pair_planes_backward: 256 channels only");
    if (N <= 0 || N > 65535 || HW <= 0 || (HW % 128) || total_slots <= 0 || max_count <= 0 || (feat_img_stride % 8) || (dfeat_img_stride % 8))
        return mpf::fail(MPF_E_SHAPE, "pair_planes_backward: bad sizes (HW must be a multiple of 128)");
    if (d_embed && d_embed_dtype != MPF_BF16 && d_embed_dtype != MPF_F32) return mpf::fail(MPF_E_DTYPE, "pair_planes_backward: d_embed bf16 or f32");
    if (workspace_bytes < mpf_pair_planes_backward_workspace_bytes(N, HW, total_slots, max_count))
        return mpf::fail(MPF_E_SHAPE, "pair_planes_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const PbGeom p = pb_geom(N, HW, max_count);
    char* ws = (char*)workspace;
    __hip_bfloat16* Et = (__hip_bfloat16*)ws;
    float* part = (float*)(ws + align256((size_t)N * kC * p.KP * 2));
    int32_t* valid = (int32_t*)((char*)part + align256((size_t)p.KS * total_slots * kC * 4));
    const __hip_bfloat16* G = (const __hip_bfloat16*)grad_planes;
    const __hip_bfloat16* e = (const __hip_bfloat16*)embed;
    const __hip_bfloat16* f = (const __hip_bfloat16*)features;
    if (d_features) {
        hipLaunchKernelGGL(pair_embed_transpose_kernel, dim3(p.KP / 32, N), dim3(kT), 0, st, e, row_off, pair_of_slot, slot_first, slot_count, Et, p.KP);
        mpf::prof_begin(st);
        mpf::set_kernel("pair_planes_dfeat_kernel");
        hipLaunchKernelGGL(pair_planes_dfeat_kernel, dim3(HW / 128, N), dim3(kT), 0, st, G, Et, p.KP, slot_first, slot_count,
                           (__hip_bfloat16*)d_features, dfeat_img_stride, HW);
        mpf::prof_end("pair_planes_dfeat_kernel", st, 2.0 * (double)N * HW * kC + 2.0 * (double)total_slots * HW, 2.0 * (double)total_slots * HW * kC);
    }
    if (d_embed) {
        hipLaunchKernelGGL(slot_valid_kernel, dim3((total_slots + kT - 1) / kT), dim3(kT), 0, st, slot_first, slot_count, N, total_slots, valid);
        mpf::prof_begin(st);
        mpf::set_kernel("pair_planes_dembed_kernel");
        hipLaunchKernelGGL(pair_planes_dembed_kernel, dim3(p.KS, p.mgroups, N), dim3(kT), 0, st, G, f, feat_img_stride, slot_first, slot_count, part,
                           total_slots, HW, p.px_per_chunk);
        mpf::prof_end("pair_planes_dembed_kernel", st, 2.0 * (double)N * HW * kC * p.mgroups + 2.0 * (double)total_slots * HW,
                      2.0 * (double)total_slots * HW * kC);
        if (d_embed_dtype == MPF_BF16)
            hipLaunchKernelGGL(pair_dembed_reduce_kernel<__hip_bfloat16>, dim3(total_slots), dim3(kT), 0, st, part, p.KS, total_slots, row_off,
                               pair_of_slot, valid, (__hip_bfloat16*)d_embed);
        else
            hipLaunchKernelGGL(pair_dembed_reduce_kernel<float>, dim3(total_slots), dim3(kT), 0, st, part, p.KS, total_slots, row_off, pair_of_slot,
                               valid, (float*)d_embed);
    }
    return mpf::check(hipGetLastError(), "mpf_pair_planes_backward");
}
