// Masked multi-head attention (cross- and self-attention of the decoder) as bf16 MFMA tiles for
// MI355X (gfx950): softmax(mask(Q K^T / sqrt(hd))) V per (image, head), head dim 32.
// Reference semantics: nn.MultiheadAttention inside CrossAttentionLayer / SelfAttentionLayer
// (mask2former_transformer_decoder.py:42-52, :100-112) with a boolean attn_mask (True = -inf).
//
// Shapes of this workload: few queries (Qtot = 100..300), many keys (HW up to 16 384 per level),
// 8 heads x 32 dims.  The work is bound by streaming K / V and the byte mask, not by MFMA, so the
// design is flash-decoding-like:
//   * one WAVE per (16*QS query rows, head, image, key split); keys are split across waves so that the
//     chip is filled although there are only ~14 query tiles.  Up to 16 splits of a query tile are the
//     waves of ONE workgroup and merge their partial (max, sum, O) through LDS — for key counts up to
//     4 096 the kernel writes the final bf16 output itself; longer key axes add a second level of
//     splits across workgroups, merged by a small combine kernel;
//   * S^T = K Q^T with v_mfma_f32_16x16x32_bf16: K = head dim = 32 in ONE instruction per 16 keys x 16
//     queries.  Operands come straight from global memory in MFMA layout: a lane loads 16 contiguous
//     bytes of a K row (A operand) / of a Q row (B operand) — no LDS staging, every K byte is read once
//     per query tile;
//   * the transposed product puts the queries on the lane axis (C layout: col = lane&15 = query), so a
//     lane owns ONE query row of P: the online-softmax scale factors are per-lane scalars and the row
//     max needs two cross-lane steps (xor 16, 32) per 32 keys;
//   * O^T = V^T P^T: the rows of the two S^T tiles of a step are taken in the order key_of_row (tile t,
//     row i = key 8 (i >> 2) + 4 t + (i & 3)), so the 8 probabilities a lane holds after the two tiles
//     belong to 8 CONSECUTIVE keys and ARE its B fragment in natural key order; V is consumed as V^T
//     [N, 256, Lk] (what the 1x1 projection of the NCHW feature map produces anyway): the A fragment is
//     one 16-byte load per lane;
//   * the byte attention mask [N, Lq, Lk] (one copy for all heads) is read as one 8-byte word per lane
//     and query sub-tile; the dK / dV kernel (keys on the lane axis) reads it transposed (attn_bwd_aux);
//   * every key loop is software-pipelined by hand (next step requested before the current step's
//     arithmetic, scheduling barriers between the phases): hipcc otherwise sinks each load to its use.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <string.h>

#include "mpf_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Loads are issued UNCONDITIONALLY on clamped indices and zeroed afterwards where out of range: a load
// behind a run-time condition makes hipcc branch around it and wait for it on its own, which turned every
// 32-key step of these kernels into ~8 dependent L2 round trips.
__device__ __forceinline__ bf16x8 ld8_rows(const __hip_bfloat16* base, int row, int limit, int64_t row_stride, int col)
{
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + (int64_t)min(row, limit - 1) * row_stride + col);
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return row < limit ? v : z;
}
// The same loads WITHOUT the zeroing select, for the software-pipelined key loops: a select on the loaded value inside the
// request phase makes the wave wait for the request it has just issued.  Keys at or past the range's end are dead in the
// score (probability exactly 0), so the finite stand-in data of the clamped address never reaches a result.
__device__ __forceinline__ bf16x8 ld8_rows_raw(const __hip_bfloat16* base, int row, int limit, int64_t row_stride, int col)
{
    return *reinterpret_cast<const bf16x8*>(base + (int64_t)min(row, limit - 1) * row_stride + col);
}
// 8 consecutive elements of a row starting at i; AL: i and limit multiples of 8, 16-byte aligned rows
template <bool AL>
__device__ __forceinline__ bf16x8 ld8_raw(const __hip_bfloat16* row, int i, int limit)
{
    if constexpr (AL) {
        return *reinterpret_cast<const bf16x8*>(row + max(min(i, limit - 8), 0));
    } else {
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const __bf16*>(row + min(i + j, limit - 1));
        return r;
    }
}
// the 8 mask bytes of keys key0 .. key0 + 7 of one query row (nonzero byte = masked), clamped reads
template <bool AL>
__device__ __forceinline__ uint2 ld_mask8(const uint8_t* mrow, int key0, int limit)
{
    if constexpr (AL) {
        return *reinterpret_cast<const uint2*>(mrow + max(min(key0, limit - 8), 0));
    } else {
        uint2 mw = make_uint2(0u, 0u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mw.x |= (mrow[min(key0 + r, limit - 1)] ? 0xFFu : 0u) << (8 * r);
            mw.y |= (mrow[min(key0 + 4 + r, limit - 1)] ? 0xFFu : 0u) << (8 * r);
        }
        return mw;
    }
}
// Key order inside a 32-key step of the forward / dQ kernels: row i of S^T tile t is key 8 (i >> 2) + 4 t + (i & 3), so that
// the 8 scores a lane holds after the two tiles (C layout: rows 4 g + r) are the CONSECUTIVE keys 8 g .. 8 g + 7 — its
// B fragment of the P V product in natural key order.  V^T / K^T fragments are then ONE 16-byte load per lane and the mask
// bytes of a query row one 8-byte load (was two 8-byte + two 4-byte loads on 32-byte pieces of twice as many cache lines).
__device__ __forceinline__ int key_of_row(int t, int i) { return ((i >> 2) << 3) + 4 * t + (i & 3); }

constexpr int kHD = 32;          // head dim
constexpr float kNegInf = -INFINITY;

__device__ __forceinline__ float xmax(float v)
{
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float xsum(float v)
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

struct AttnParams {
    const __hip_bfloat16* q;      // [Lq, N, E]   (E = heads*32)
    const __hip_bfloat16* k;      // [Lk, N, E]
    const __hip_bfloat16* vt;     // [N, E, Lk]
    const uint8_t* mask;          // [N or 1, Lq, Lk] bytes (1 = masked) or nullptr
    int64_t mask_stride_n;        // 0 for a mask shared by all images
    float* part_o;                // [splits, N, H, Lq, 32]   unnormalised partial O
    float* part_ml;               // [splits, N, H, Lq, 2]    (max, sum)
    int Lq, Lk, N, H, E, splits, keys_per_split;    // splits = workgroup-level splits (partials that reach global memory)
    int nw;                       // waves per workgroup = key splits merged inside it
    __hip_bfloat16* out;          // [Lq, N, E] final output and
    float* lse;                   // [N, H, Lq] log-sum-exp, written by the kernel itself when splits == 1
    float scale;
    int64_t k_row, k_img;         // element strides of k between sequence positions / images (N*E, E when dense)
};

constexpr int kMaxNW = 8;         // waves per workgroup of the forward / dQ kernels
constexpr int kMergePitch = 36;   // floats per query row of a wave's partial in LDS (32 + 4: 16-byte aligned, rows 4 banks apart)
constexpr int kMergeWave = 32 * kMergePitch + 64;   // floats per wave: 32 rows of O (or dQ) + (max, sum) per row

// QS = number of 16-row query sub-tiles per wave
// OCC (round 6): minimum waves per SIMD the register allocation must allow.  4 = two 8-wave workgroups per CU (<= 128 registers; the
// aligned forms fit without spills): these kernels are bound by the dependent instruction stream of a (query tile, key step) unit —
// their time goes with the number of queries, not with bytes or prefetch depth — so a second workgroup per CU is what hides it.
template <int QS, bool AL, bool MK, int OCC = 2>
__global__ __launch_bounds__(64 * kMaxNW, OCC) void attn_fwd_kernel(AttnParams p)
{
    static_assert(QS == 2, "the LDS merge is laid out for 32 query rows per wave");
    extern __shared__ __attribute__((aligned(16))) float s_merge[];
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qtiles = (p.Lq + 16 * QS - 1) / (16 * QS);
    const int qt = blockIdx.x % qtiles, wgs = blockIdx.x / qtiles;
    const int h = blockIdx.y, n = blockIdx.z;
    const int q0 = qt * 16 * QS;
    const int kb0 = min(p.Lk, (wgs * p.nw + wave) * p.keys_per_split);      // (a wave past the end of the keys has an empty range)
    const int kb1 = min(p.Lk, kb0 + p.keys_per_split);
    const int64_t rowE = (int64_t)p.N * p.E;                       // stride between sequence positions
    const __hip_bfloat16* qb = p.q + (int64_t)n * p.E + h * kHD;
    const __hip_bfloat16* kb = p.k + (int64_t)n * p.k_img + h * kHD;
    const __hip_bfloat16* vb = p.vt + ((int64_t)n * p.E + h * kHD) * p.Lk;
    bf16x8 bq[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        bq[s] = ld8_rows(qb, q0 + 16 * s + c16, p.Lq, rowE, 8 * g);
    }
    f32x4 o[QS][2];
    float m[QS], l[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        o[s][0] = f32x4{0, 0, 0, 0}; o[s][1] = f32x4{0, 0, 0, 0};
        m[s] = kNegInf; l[s] = 0.f;
    }

    // Operands of one 32-key step, ALL requested before the first MFMA (clamped addresses, no branches)
    struct Step {
        bf16x8 ak[2], av[2];
        uint2 mw[QS];
    };
    auto load_step = [&](Step& st, const int kk) {
        // S^T tiles: row c16 of tile t = key kk + key_of_row(t, c16)
#pragma unroll
        for (int t = 0; t < 2; ++t) st.ak[t] = ld8_rows_raw(kb, kk + key_of_row(t, c16), kb1, p.k_row, 8 * g);
        // V^T fragments: rows d = 16*dt + c16, keys kk + 8g .. + 7  (AL: Lk % 8 == 0, 16-byte aligned rows)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) st.av[dt] = ld8_raw<AL>(vb + (int64_t)(16 * dt + c16) * p.Lk, kk + 8 * g, kb1);
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            st.mw[s] = make_uint2(0u, 0u);
            if constexpr (MK) {
                const int qi = q0 + 16 * s + c16;
                const uint8_t* mrow = p.mask + (int64_t)n * p.mask_stride_n + (int64_t)min(qi, p.Lq - 1) * p.Lk;
                st.mw[s] = ld_mask8<AL>(mrow, kk + 8 * g, kb1);
            }
        }
    };
    auto compute_step = [&](const Step& st, const int kk) {
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            float sc[8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.ak[t], bq[s], acc, 0, 0, 0);
                // lane (query c16, group g) holds keys kk + 8g + 4t + r
                const int key0 = kk + 8 * g + 4 * t;
                const uint32_t mw = t ? st.mw[s].y : st.mw[s].x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dead = (key0 + r >= kb1) || ((mw >> (8 * r)) & 0xFFu);
                    sc[4 * t + r] = dead ? kNegInf : acc[r] * p.scale;
                }
            }
            float mx = sc[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sc[j]);
            mx = xmax(mx);
            const float m_new = fmaxf(m[s], mx);
            const float alpha = (m_new == kNegInf) ? 1.f : __expf(m[s] - m_new);
            float ps = 0.f;
            bf16x8 bp;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float e = (m_new == kNegInf) ? 0.f : __expf(sc[j] - m_new);
                ps += e;
                bp[j] = (__bf16)e;
            }
            l[s] = l[s] * alpha + ps;
            m[s] = m_new;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                o[s][dt] *= alpha;
                o[s][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.av[dt], bp, o[s][dt], 0, 0, 0);
            }
        }
    };
    // Step i + 1 is requested before the arithmetic of step i (two alternating Step sets; the loads are unconditional on
    // clamped addresses, so a request past the wave's range is harmless): with the splits of a query tile as the waves of
    // one workgroup the short key axes run at <= 2 waves per SIMD, where nothing else covers a step's round trip.  (At the
    // fine level, four resident waves per SIMD, it changes nothing: there the step is bound by its ~300 VALU instructions
    // of softmax bookkeeping against 8 MFMAs.)
    // (straight-line body, two steps per trip: a step past the wave's range has every key dead and changes nothing — an exit
    // in the middle of the body made hipcc wait vmcnt(0) at the merge, i.e. for the request it had just issued)
    // The scheduling barriers keep hipcc from sinking a step's loads down to their first use (it did: four
    // "global_load; s_waitcnt vmcnt(0)" pairs per trip, i.e. every operand's round trip exposed).
    Step sa, sb;
    load_step(sa, kb0);
    __builtin_amdgcn_sched_barrier(0);
    for (int kk = kb0; kk < kb1; kk += 64) {
        load_step(sb, kk + 32);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(sa, kk);
        __builtin_amdgcn_sched_barrier(0);
        load_step(sa, kk + 64);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(sb, kk + 32);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- merge the waves of the workgroup through LDS ---------------------------------------------------
    {
        float* so = s_merge + wave * kMergeWave;
        float* sml = so + 32 * kMergePitch;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            const int row = 16 * s + c16;
            const float lt = xsum(l[s]);
            // O^T C layout: lane (query c16, g) holds d = 16*dt + 4g + r
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) *reinterpret_cast<f32x4*>(so + row * kMergePitch + 16 * dt + 4 * g) = o[s][dt];
            if (g == 0) { sml[row * 2] = m[s]; sml[row * 2 + 1] = lt; }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * kHD; idx += blockDim.x) {
        const int q = idx >> 5, d = idx & 31;
        float M = kNegInf;
        for (int w = 0; w < p.nw; ++w) M = fmaxf(M, s_merge[w * kMergeWave + 32 * kMergePitch + q * 2]);
        float L = 0.f, O = 0.f;
        for (int w = 0; w < p.nw; ++w) {
            const float* so = s_merge + w * kMergeWave;
            const float ms = so[32 * kMergePitch + q * 2];
            const float wgt = (ms == kNegInf) ? 0.f : __expf(ms - M);
            L += wgt * so[32 * kMergePitch + q * 2 + 1];
            O += wgt * so[q * kMergePitch + d];
        }
        const int qi = q0 + q;
        if (qi < p.Lq) {
            if (p.splits == 1) {
                p.out[((int64_t)qi * p.N + n) * p.E + h * kHD + d] = __float2bfloat16(L > 0.f ? O / L : 0.f);
                if (d == 0 && p.lse) p.lse[((int64_t)n * p.H + h) * p.Lq + qi] = (L > 0.f) ? M + __logf(L) : kNegInf;
            } else {
                const int64_t row = (((int64_t)wgs * p.N + n) * p.H + h) * p.Lq + qi;
                p.part_o[row * kHD + d] = O;
                if (d == 0) { p.part_ml[row * 2] = M; p.part_ml[row * 2 + 1] = L; }
            }
        }
    }
}

// merge the key splits: out[q, n, h*32+d] (bf16) and lse[n, h, q]
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                            __hip_bfloat16* __restrict__ out, float* __restrict__ lse,
                                                            int Lq, int N, int H, int E, int splits)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;       // over N*H*Lq*32
    const int total = N * H * Lq * kHD;
    if (idx >= total) return;
    const int d = idx & 31;
    const int row = idx >> 5;                              // (n*H + h)*Lq + q
    const int q = row % Lq, nh = row / Lq, h = nh % H, n = nh / H;
    const int64_t sstride = (int64_t)N * H * Lq;
    float M = kNegInf, L = 0.f, O = 0.f;
    if (splits <= 8) {
        // every operand requested at once on clamped split indices (one round trip instead of one per unrolled group)
        float2 ml[8];
        float po[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int sc = min(s, splits - 1);
            ml[s] = *reinterpret_cast<const float2*>(part_ml + (sc * sstride + row) * 2);
            po[s] = part_o[(sc * sstride + row) * kHD + d];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) M = fmaxf(M, ml[s].x);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float w = (s >= splits || ml[s].x == kNegInf) ? 0.f : __expf(ml[s].x - M);
            L += w * ml[s].y;
            O += w * po[s];
        }
    } else {
#pragma unroll 8
        for (int s = 0; s < splits; ++s) M = fmaxf(M, part_ml[(s * sstride + row) * 2]);
#pragma unroll 8
        for (int s = 0; s < splits; ++s) {
            const float ms = part_ml[(s * sstride + row) * 2];
            const float w = (ms == kNegInf) ? 0.f : __expf(ms - M);
            L += w * part_ml[(s * sstride + row) * 2 + 1];
            O += w * part_o[(s * sstride + row) * kHD + d];
        }
    }
    out[((int64_t)q * N + n) * E + h * kHD + d] = __float2bfloat16(L > 0.f ? O / L : 0.f);
    if (d == 0 && lse) lse[row] = (L > 0.f) ? M + __logf(L) : kNegInf;
}


// ================================================================================================
// Backward.  dV = P^T dO, dP = dO V^T, dS = P o (dP - delta), dQ = scale dS K, dK = scale dS^T Q with
// P recomputed from Q, K and the saved log-sum-exp, delta[q] = sum_d dO[q,d] O[q,d].
//   attn_bwd_kv_kernel: one wave per (32 keys, head, image) loops over ALL queries and owns its
//                       dK / dV rows (no atomics).  S = Q K^T puts the keys on the lane axis, so the
//                       C tiles of dV^T / dK^T (rows = d, cols = keys) are stored key-major directly.
//   attn_bwd_q_kernel : one wave per (32 queries, head, image, key split) loops over its keys like
//                       the forward (queries on the lane axis) and accumulates dQ^T; splits are summed
//                       by attn_sum_splits_kernel.
// The contraction index of every MFMA must be contiguous in a lane's fragment, hence the transposed
// companions Q^T, dO^T [N, E, LqP] (tiny) and K^T [N, E, Lk] next to the key-major K, V.
// ================================================================================================
struct AttnBwdParams {
    const __hip_bfloat16 *q, *k, *v, *dout;        // [Lq|Lk, N, E]
    const __hip_bfloat16 *qT, *doT;                // [N, E, LqP]  (LqP = Lq rounded up to 32, zero padded)
    const __hip_bfloat16* kT;                      // [N, E, Lk]
    const uint8_t* mask; int64_t mask_stride_n;
    const float *lse, *delta;                      // [N, H, Lq]
    const float2* ld2;                             // [N, H, LqP] (lse, delta) pairs and
    const uint8_t* maskT;                          // [N or 1, Lk, LqP] the mask transposed: the dK / dV kernel's operands (aux)
    int64_t maskT_stride_n;
    __hip_bfloat16 *dk, *dv;                       // [Lk, N, E]
    float* part_dq;                                // [splits, N, H, Lq, 32]  (splits = workgroup-level splits)
    __hip_bfloat16* dq;                            // [Lq, N, E]: written by the dQ kernel itself when splits == 1
    int nw;                                        // waves per workgroup of the dQ kernel = key splits summed inside it
    int Lq, LqP, Lk, N, H, E, splits, keys_per_split;
    float scale;
    int64_t kv_row, kv_img;        // element strides of k, v between sequence positions / images (N*E, E when dense)
    int64_t dkv_row, dkv_img;      // ... and of dk, dv
};

// dK / dV.  Keys live on the lane axis (C layout of S = Q K^T: col = key, rows = queries), so per (key, query) the kernel needs
// the mask byte, lse and delta with the QUERY index running inside a lane: they come from the aux buffers of
// attn_bwd_aux (mask transposed to [Lk, LqP], (lse, delta) pairs padded to LqP), and the rows of the two S tiles of a
// 32-query step are taken in the order key_of_row (tile qt, row i = query 8 (i >> 2) + 4 qt + (i & 3)), which makes a
// lane's 8 queries CONSECUTIVE: per step and lane one 8-byte mask load per key tile, four 16-byte loads of (lse, delta)
// and ONE 16-byte load per Q^T / dO^T fragment — 14 loads instead of 44 (16 of them single bytes).  Step i + 1 is requested
// before the arithmetic of step i (scheduling barriers: see the forward).
template <bool MK>
__global__ __launch_bounds__(64) void attn_bwd_kv_kernel(AttnBwdParams p)
{
    const int lane = threadIdx.x, c16 = lane & 15, g = lane >> 4;
    const int kb = blockIdx.x * 32, h = blockIdx.y, n = blockIdx.z;
    const int64_t rowE = (int64_t)p.N * p.E;
    const int64_t hoff = (int64_t)n * p.E + h * kHD;
    const int64_t kvoff = (int64_t)n * p.kv_img + h * kHD, dkvoff = (int64_t)n * p.dkv_img + h * kHD;
    bf16x8 bk[2], bv[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = kb + 16 * kt + c16;
        bk[kt] = ld8_rows(p.k + kvoff, key, p.Lk, p.kv_row, 8 * g);
        bv[kt] = ld8_rows(p.v + kvoff, key, p.Lk, p.kv_row, 8 * g);
    }
    f32x4 dkt[2][2], dvt[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { dkt[a][b] = f32x4{0, 0, 0, 0}; dvt[a][b] = f32x4{0, 0, 0, 0}; }
    const float2* ld2 = p.ld2 + ((int64_t)n * p.H + h) * p.LqP;
    const __hip_bfloat16* qTb = p.qT + ((int64_t)n * p.E + h * kHD) * p.LqP;
    const __hip_bfloat16* doTb = p.doT + ((int64_t)n * p.E + h * kHD) * p.LqP;
    const uint8_t* mT[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
        mT[kt] = MK ? p.maskT + (int64_t)n * p.maskT_stride_n + (int64_t)min(kb + 16 * kt + c16, p.Lk - 1) * p.LqP : nullptr;

    struct Step {
        bf16x8 aq[2], ado[2], aqT[2], adoT[2];
        f32x4 ld[4];             // (lse, delta) of queries qq + 8g .. + 7
        uint2 mb[2];             // mask bytes of those queries for key tile kt
    };
    auto load_step = [&](Step& st, const int qq_) {
        const int qq = min(qq_, p.LqP - 32);              // (the padding step of an odd step count: every query dead)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qi = qq + key_of_row(qt, c16);
            st.aq[qt] = ld8_rows_raw(p.q + hoff, qi, p.Lq, rowE, 8 * g);
            st.ado[qt] = ld8_rows_raw(p.dout + hoff, qi, p.Lq, rowE, 8 * g);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) st.ld[u] = *reinterpret_cast<const f32x4*>(ld2 + qq + 8 * g + 2 * u);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            st.mb[kt] = make_uint2(0u, 0u);
            if constexpr (MK) st.mb[kt] = *reinterpret_cast<const uint2*>(mT[kt] + qq + 8 * g);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int64_t ro = (int64_t)(16 * dt + c16) * p.LqP + qq + 8 * g;
            st.adoT[dt] = *reinterpret_cast<const bf16x8*>(doTb + ro);
            st.aqT[dt] = *reinterpret_cast<const bf16x8*>(qTb + ro);
        }
    };
    auto compute_step = [&](const Step& st, const int qq) {
        bf16x8 bp[2], bds[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = kb + 16 * kt + c16;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 z = {0, 0, 0, 0};
                const f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.aq[qt], bk[kt], z, 0, 0, 0);
                const f32x4 dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.ado[qt], bv[kt], z, 0, 0, 0);
                const uint32_t mw = qt ? st.mb[kt].y : st.mb[kt].x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // C row 4g + r of tile qt = query qq + 8g + 4qt + r = the lane's query j = 4qt + r
                    const int jq = 4 * qt + r, qi = qq + 8 * g + jq;
                    const float ls = st.ld[jq >> 1][(jq & 1) * 2], de = st.ld[jq >> 1][(jq & 1) * 2 + 1];
                    const bool dead = (qi >= p.Lq) || (key >= p.Lk) || (ls == kNegInf) || ((mw >> (8 * r)) & 0xFFu);
                    const float pr = dead ? 0.f : __expf(sacc[r] * p.scale - ls);
                    const float ds = dead ? 0.f : pr * (dpacc[r] - de) * p.scale;
                    bp[kt][jq] = (__bf16)pr;
                    bds[kt][jq] = (__bf16)ds;
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                dvt[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.adoT[dt], bp[kt], dvt[kt][dt], 0, 0, 0);
                dkt[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.aqT[dt], bds[kt], dkt[kt][dt], 0, 0, 0);
            }
        }
    };
    Step sa, sb;
    load_step(sa, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int qq = 0; qq < p.Lq; qq += 64) {
        load_step(sb, qq + 32);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(sa, qq);
        __builtin_amdgcn_sched_barrier(0);
        load_step(sa, qq + 64);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(sb, qq + 32);
        __builtin_amdgcn_sched_barrier(0);
    }
    // C tiles: rows d = 16dt + 4g + r, cols = key c16  ->  key-major 8-byte stores
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = kb + 16 * kt + c16;
        if (key < p.Lk) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x4 kk = {(__bf16)dkt[kt][dt][0], (__bf16)dkt[kt][dt][1], (__bf16)dkt[kt][dt][2], (__bf16)dkt[kt][dt][3]};
                const bf16x4 vv = {(__bf16)dvt[kt][dt][0], (__bf16)dvt[kt][dt][1], (__bf16)dvt[kt][dt][2], (__bf16)dvt[kt][dt][3]};
                *reinterpret_cast<bf16x4*>(p.dk + dkvoff + (int64_t)key * p.dkv_row + 16 * dt + 4 * g) = kk;
                *reinterpret_cast<bf16x4*>(p.dv + dkvoff + (int64_t)key * p.dkv_row + 16 * dt + 4 * g) = vv;
            }
        }
    }
}

template <int QS, bool AL, bool MK>
__global__ __launch_bounds__(64 * kMaxNW) void attn_bwd_q_kernel(AttnBwdParams p)
{
    static_assert(QS == 2, "the LDS merge is laid out for 32 query rows per wave");
    extern __shared__ __attribute__((aligned(16))) float s_merge[];
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qtiles = (p.Lq + 16 * QS - 1) / (16 * QS);
    const int qt = blockIdx.x % qtiles, wgs = blockIdx.x / qtiles;
    const int h = blockIdx.y, n = blockIdx.z;
    const int q0 = qt * 16 * QS;
    const int kb0 = min(p.Lk, (wgs * p.nw + wave) * p.keys_per_split);
    const int kb1 = min(p.Lk, kb0 + p.keys_per_split);
    const int64_t rowE = (int64_t)p.N * p.E;
    const int64_t hoff = (int64_t)n * p.E + h * kHD;
    const int64_t kvoff = (int64_t)n * p.kv_img + h * kHD;
    const __hip_bfloat16* kTb = p.kT + ((int64_t)n * p.E + h * kHD) * p.Lk;
    bf16x8 bq[QS], bdo[QS];
    float ls[QS], de[QS];
    f32x4 dq[QS][2];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        const int qi = q0 + 16 * s + c16;
        bq[s] = ld8_rows(p.q + hoff, qi, p.Lq, rowE, 8 * g);
        bdo[s] = ld8_rows(p.dout + hoff, qi, p.Lq, rowE, 8 * g);
        ls[s] = qi < p.Lq ? p.lse[((int64_t)n * p.H + h) * p.Lq + qi] : kNegInf;
        de[s] = qi < p.Lq ? p.delta[((int64_t)n * p.H + h) * p.Lq + qi] : 0.f;
        dq[s][0] = f32x4{0, 0, 0, 0}; dq[s][1] = f32x4{0, 0, 0, 0};
    }
    // operands of one 32-key step, all requested before the first MFMA (as in the forward)
    struct Step {
        bf16x8 ak[2], av[2], akT[2];
        uint2 mw[QS];
    };
    auto load_step = [&](Step& st, const int kk) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = kk + key_of_row(t, c16);
            st.ak[t] = ld8_rows_raw(p.k + kvoff, key, kb1, p.kv_row, 8 * g);
            st.av[t] = ld8_rows_raw(p.v + kvoff, key, kb1, p.kv_row, 8 * g);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) st.akT[dt] = ld8_raw<AL>(kTb + (int64_t)(16 * dt + c16) * p.Lk, kk + 8 * g, kb1);
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            st.mw[s] = make_uint2(0u, 0u);
            if constexpr (MK) {
                const int qi = q0 + 16 * s + c16;
                const uint8_t* mrow = p.mask + (int64_t)n * p.mask_stride_n + (int64_t)min(qi, p.Lq - 1) * p.Lk;
                st.mw[s] = ld_mask8<AL>(mrow, kk + 8 * g, kb1);
            }
        }
    };
    auto compute_step = [&](const Step& st, const int kk) {
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            bf16x8 bds;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 z = {0, 0, 0, 0};
                const f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.ak[t], bq[s], z, 0, 0, 0);
                const f32x4 dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.av[t], bdo[s], z, 0, 0, 0);
                const int key0 = kk + 8 * g + 4 * t;                      // (key order of a step: key_of_row)
                const uint32_t mw = t ? st.mw[s].y : st.mw[s].x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dead = (key0 + r >= kb1) || ((mw >> (8 * r)) & 0xFFu) || (ls[s] == kNegInf);
                    const float pr = dead ? 0.f : __expf(sacc[r] * p.scale - ls[s]);
                    bds[4 * t + r] = (__bf16)(pr * (dpacc[r] - de[s]) * p.scale);
                }
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                dq[s][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.akT[dt], bds, dq[s][dt], 0, 0, 0);
        }
    };
    // Step i + 1 is requested before the arithmetic of step i (two alternating Step sets; the loads are unconditional on
    // clamped addresses, so a request past the wave's range is harmless): with the splits of a query tile as the waves of
    // one workgroup the short key axes run at <= 2 waves per SIMD, where nothing else covers a step's round trip.  (At the
    // fine level, four resident waves per SIMD, it changes nothing: there the step is bound by its ~300 VALU instructions
    // of softmax bookkeeping against 8 MFMAs.)
    // (straight-line body, two steps per trip: a step past the wave's range has every key dead and changes nothing — an exit
    // in the middle of the body made hipcc wait vmcnt(0) at the merge, i.e. for the request it had just issued)
    // The scheduling barriers keep hipcc from sinking a step's loads down to their first use (it did: four
    // "global_load; s_waitcnt vmcnt(0)" pairs per trip, i.e. every operand's round trip exposed).
    Step sa, sb;
    load_step(sa, kb0);
    __builtin_amdgcn_sched_barrier(0);
    for (int kk = kb0; kk < kb1; kk += 64) {
        load_step(sb, kk + 32);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(sa, kk);
        __builtin_amdgcn_sched_barrier(0);
        load_step(sa, kk + 64);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(sb, kk + 32);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the waves of the workgroup (key splits of one query tile) are summed through LDS, in wave order
    {
        float* so = s_merge + wave * (32 * kMergePitch);
#pragma unroll
        for (int s = 0; s < QS; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) *reinterpret_cast<f32x4*>(so + (16 * s + c16) * kMergePitch + 16 * dt + 4 * g) = dq[s][dt];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * kHD; idx += blockDim.x) {
        const int q = idx >> 5, d = idx & 31;
        float acc = 0.f;
        for (int w = 0; w < p.nw; ++w) acc += s_merge[w * (32 * kMergePitch) + q * kMergePitch + d];
        const int qi = q0 + q;
        if (qi < p.Lq) {
            if (p.splits == 1) p.dq[((int64_t)qi * p.N + n) * p.E + h * kHD + d] = __float2bfloat16(acc);
            else p.part_dq[((((int64_t)wgs * p.N + n) * p.H + h) * p.Lq + qi) * kHD + d] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void attn_sum_splits_kernel(const float* __restrict__ part, __hip_bfloat16* __restrict__ out,
                                                               int Lq, int N, int H, int E, int splits)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = N * H * Lq * kHD;
    if (idx >= total) return;
    const int d = idx & 31, row = idx >> 5;
    const int q = row % Lq, nh = row / Lq, h = nh % H, n = nh / H;
    float acc = 0.f;
#pragma unroll 8
    for (int s = 0; s < splits; ++s) acc += part[((int64_t)s * N * H * Lq + row) * kHD + d];
    out[((int64_t)q * N + n) * E + h * kHD + d] = __float2bfloat16(acc);
}


// ------------------------------------------------------------------------------------------------
// Layout helpers (one launch each instead of a handful of permute / pad / reduce kernels):
//   transpose2: [L, N, E] -> [N, E, LP] for two tensors at once (K and V in the forward; Q and dO in
//               the backward), zero padded to LP >= L;
//   delta     : delta[n, h, q] = sum_d dO[q, n, h*32+d] * O[q, n, h*32+d].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void transpose2_tile(const __hip_bfloat16* __restrict__ a, const __hip_bfloat16* __restrict__ b,
                                                __hip_bfloat16* __restrict__ aT, __hip_bfloat16* __restrict__ bT,
                                                int L, int LP, int NE, const int bx, const int by, const int E = 0,
                                                const int64_t in_row = 0, const int64_t in_img = 0)
{
    // input element (position l, column c = n * E + e) at l * in_row + n * in_img + e; E == 0: dense rows of NE columns
    auto src = [&](int l, int c) -> int64_t {
        return E ? (int64_t)l * in_row + (int64_t)(c / E) * in_img + (c % E) : (int64_t)l * NE + c;
    };
    // tile of 64 sequence positions x 64 columns through LDS: 16-byte global loads (8 columns of a row) and stores (8
    // positions of a column); the transpose itself is 2-byte LDS writes into [column][position] rows of 72 elements
    // (144 B: the 16-byte reads of consecutive columns start 36 banks apart)
    __shared__ __attribute__((aligned(16))) unsigned short ta[64][72], tb[64][72];
    const int l0 = bx * 64, c0 = by * 64;
    const bool vec = (NE % 8 == 0) && (LP % 8 == 0) && (E % 8 == 0) && (in_row % 8 == 0) && (in_img % 8 == 0) &&
                     (((uintptr_t)a | (uintptr_t)b | (uintptr_t)aT | (uintptr_t)bT) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int u = threadIdx.x + it * 256;                 // 512 pieces: (row r, 8-column group cg)
            const int r = u >> 3, cg = u & 7;
            const int l = l0 + r, c = c0 + cg * 8;
            uint4 va = make_uint4(0u, 0u, 0u, 0u), vb = va;
            if (l < L && c < NE) {
                va = *reinterpret_cast<const uint4*>(a + src(l, c));
                vb = *reinterpret_cast<const uint4*>(b + src(l, c));
            }
            const unsigned wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ta[cg * 8 + 2 * k][r] = (unsigned short)(wa[k] & 0xffffu);
                ta[cg * 8 + 2 * k + 1][r] = (unsigned short)(wa[k] >> 16);
                tb[cg * 8 + 2 * k][r] = (unsigned short)(wb[k] & 0xffffu);
                tb[cg * 8 + 2 * k + 1][r] = (unsigned short)(wb[k] >> 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int u = threadIdx.x + it * 256;                 // (column cc, 8-position group lg)
            const int cc = u >> 3, lg = u & 7;
            const int c = c0 + cc, l = l0 + lg * 8;
            if (c < NE && l < LP) {
                *reinterpret_cast<uint4*>(aT + (int64_t)c * LP + l) = *reinterpret_cast<const uint4*>(&ta[cc][lg * 8]);
                *reinterpret_cast<uint4*>(bT + (int64_t)c * LP + l) = *reinterpret_cast<const uint4*>(&tb[cc][lg * 8]);
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 4 rows per pass
    {
        // all 32 loads of a thread requested together on clamped indices (a load behind `ok ? p[i] : 0` is a branch + a wait of
        // its own: 16 dependent round trips made this path 12 us for the 114-query self-attention operands)
        const int c = min(c0 + tx, NE - 1);
        unsigned short va[16], vb[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int l = min(l0 + ty + 4 * k, L - 1);
            va[k] = reinterpret_cast<const unsigned short*>(a)[src(l, c)];
            vb[k] = reinterpret_cast<const unsigned short*>(b)[src(l, c)];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int r = ty + 4 * k;
            const bool ok = l0 + r < L && c0 + tx < NE;
            ta[tx][r] = ok ? va[k] : (unsigned short)0;
            tb[tx][r] = ok ? vb[k] : (unsigned short)0;
        }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, l = l0 + tx;
        if (c < NE && l < LP) {
            reinterpret_cast<unsigned short*>(aT)[(int64_t)c * LP + l] = ta[r][tx];
            reinterpret_cast<unsigned short*>(bT)[(int64_t)c * LP + l] = tb[r][tx];
        }
    }
}

__global__ __launch_bounds__(256) void attn_transpose2_kernel(const __hip_bfloat16* __restrict__ a, const __hip_bfloat16* __restrict__ b,
                                                               __hip_bfloat16* __restrict__ aT, __hip_bfloat16* __restrict__ bT,
                                                               int L, int LP, int NE, int E, int64_t in_row, int64_t in_img)
{
    transpose2_tile(a, b, aT, bT, L, LP, NE, (int)blockIdx.x, (int)blockIdx.y, E, in_row, in_img);
}

__device__ __forceinline__ void delta_block(const __hip_bfloat16* __restrict__ dout, const __hip_bfloat16* __restrict__ out,
                                            float* __restrict__ delta, int Lq, int N, int H, const int block,
                                            const float* __restrict__ lse = nullptr, float2* __restrict__ ld2 = nullptr, int LqP = 0)
{
    const int idx = block * 256 + threadIdx.x;                    // over Lq*N*H*32, 32 lanes per (q,n,h)
    const int d = idx & 31, row = idx >> 5;
    const int total = Lq * N * H;
    float v = 0.f;
    if (row < total) v = __bfloat162float(dout[(int64_t)row * kHD + d]) * __bfloat162float(out[(int64_t)row * kHD + d]);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
    if (row < total && d == 0) {
        const int h = row % H, n = (row / H) % N, q = row / (H * N);
        delta[((int64_t)n * H + h) * Lq + q] = v;
        if (ld2) ld2[((int64_t)n * H + h) * LqP + q] = make_float2(lse[((int64_t)n * H + h) * Lq + q], v);
    }
}

// mask [Nm, Lq, Lk] bytes -> maskT [Nm, Lk, LqP] (queries past Lq: 1 = masked), 64 x 64 tiles through LDS; block = (image, query
// tile, key tile).  16-byte loads / stores when the rows allow it.
__device__ __forceinline__ void mask_transpose_block(const uint8_t* __restrict__ mask, uint8_t* __restrict__ maskT, int Lq, int LqP,
                                                     int Lk, const int block, const int tiles_q, const int tiles_k)
{
    __shared__ __attribute__((aligned(16))) uint8_t tm[64][80];          // [key][query], 80-byte rows
    const int img = block / (tiles_q * tiles_k), rem = block - img * (tiles_q * tiles_k);
    const int q0 = (rem / tiles_k) * 64, k0 = (rem % tiles_k) * 64;
    const uint8_t* src = mask + (int64_t)img * Lq * Lk;
    uint8_t* dst = maskT + (int64_t)img * Lk * LqP;
    const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 16;         // query row r of the tile, keys c0 .. c0 + 15
    const int q = q0 + r;
    uint8_t b[16];
    if ((Lk & 15) == 0 && ((uintptr_t)mask & 15) == 0 && k0 + c0 + 16 <= Lk) {
        const uint4 w = *reinterpret_cast<const uint4*>(src + (int64_t)min(q, Lq - 1) * Lk + k0 + c0);
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = (uint8_t)(ws[i >> 2] >> (8 * (i & 3)));
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = src[(int64_t)min(q, Lq - 1) * Lk + min(k0 + c0 + i, Lk - 1)];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) tm[c0 + i][r] = q < Lq ? b[i] : (uint8_t)1;
    __syncthreads();
    const int kr = threadIdx.x >> 2, qc = (threadIdx.x & 3) * 16;        // key row kr of the tile, queries qc .. qc + 15
    if (k0 + kr < Lk && q0 + qc < LqP)                                   // (LqP % 32 == 0: whole 16-query pieces)
        *reinterpret_cast<uint4*>(dst + (int64_t)(k0 + kr) * LqP + q0 + qc) = *reinterpret_cast<const uint4*>(&tm[kr][qc]);
}

__global__ __launch_bounds__(256) void attn_delta_kernel(const __hip_bfloat16* __restrict__ dout, const __hip_bfloat16* __restrict__ out,
                                                          float* __restrict__ delta, int Lq, int N, int H)
{
    delta_block(dout, out, delta, Lq, N, H, (int)blockIdx.x);
}

// What the attention backward needs from the query side, in one launch: Q^T and dO^T (blocks < n_tiles) and delta (the rest)
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const __hip_bfloat16* __restrict__ q, const __hip_bfloat16* __restrict__ dout,
                                                             const __hip_bfloat16* __restrict__ out, __hip_bfloat16* __restrict__ qT,
                                                             __hip_bfloat16* __restrict__ doT, float* __restrict__ delta, int Lq,
                                                             int LqP, int N, int H, int tiles_x, int n_tiles)
{
    const int b = (int)blockIdx.x;
    if (b < n_tiles) transpose2_tile(q, dout, qT, doT, Lq, LqP, N * H * kHD, b % tiles_x, b / tiles_x);
    else delta_block(dout, out, delta, Lq, N, H, b - n_tiles);
}

// attn_bwd_prep + the aux operands of the dK / dV kernel in the same launch: (lse, delta) pairs from the delta blocks, then the
// transposed mask
__global__ __launch_bounds__(256) void attn_bwd_prep_aux_kernel(const __hip_bfloat16* __restrict__ q, const __hip_bfloat16* __restrict__ dout,
                                                                 const __hip_bfloat16* __restrict__ out, const float* __restrict__ lse,
                                                                 const uint8_t* __restrict__ mask, __hip_bfloat16* __restrict__ qT,
                                                                 __hip_bfloat16* __restrict__ doT, float* __restrict__ delta,
                                                                 float2* __restrict__ ld2, uint8_t* __restrict__ maskT, int Lq, int LqP,
                                                                 int Lk, int N, int H, int tiles_x, int n_tiles, int delta_blocks,
                                                                 int mtiles_q, int mtiles_k)
{
    const int b = (int)blockIdx.x;
    if (b < n_tiles) transpose2_tile(q, dout, qT, doT, Lq, LqP, N * H * kHD, b % tiles_x, b / tiles_x);
    else if (b < n_tiles + delta_blocks) delta_block(dout, out, delta, Lq, N, H, b - n_tiles, lse, ld2, LqP);
    else mask_transpose_block(mask, maskT, Lq, LqP, Lk, b - n_tiles - delta_blocks, mtiles_q, mtiles_k);
}

// the aux operands alone, from an existing delta (callers of the entry points without aux)
__global__ __launch_bounds__(256) void attn_bwd_aux_kernel(const float* __restrict__ lse, const float* __restrict__ delta,
                                                            const uint8_t* __restrict__ mask, float2* __restrict__ ld2,
                                                            uint8_t* __restrict__ maskT, int Lq, int LqP, int Lk, int NH, int ld_blocks,
                                                            int mtiles_q, int mtiles_k)
{
    const int b = (int)blockIdx.x;
    if (b < ld_blocks) {
        const int idx = b * 256 + threadIdx.x;
        if (idx < NH * Lq) {
            const int nh = idx / Lq, qi = idx - nh * Lq;
            ld2[(int64_t)nh * LqP + qi] = make_float2(lse[idx], delta[idx]);
        }
    } else {
        mask_transpose_block(mask, maskT, Lq, LqP, Lk, b - ld_blocks, mtiles_q, mtiles_k);
    }
}

}  // namespace

// Key splits of the forward / dQ kernels: one WAVE per (32 queries, head, image, split) and each wave walks its keys in
// 32-key steps of ~2 us, so short waves are what makes these kernels fast: 128 keys per wave (64 for key axes up to 256),
// doubled while that would mean more than ~4 096 waves (measured at config B, Lq = 120: level 1 (Lk = 4 096) 16 + 8 us with
// 128 keys per wave against 24 + 5 with 256, level 2 (Lk = 16 384) 58 + 8 against 49 + 5).  Up to kMaxNW = 8 waves form the
// workgroup of a query tile and merge through LDS — key axes up to 1 024 need no second pass; beyond that the workgroups'
// partials go through the workspace and the combine / sum kernels.
struct AttnSplit { int keys_per_wave, nw, wg_splits; };
static int g_attn_occ = 4;      // mpf_set_option attn_occ (2 | 4): forward, aligned forms
static int g_attn_nw = kMaxNW, g_attn_kpw = 0;     // A/B switches (mpf_set_option attn_nw / attn_kpw; 0 = the policy above)
static AttnSplit attn_split(int Lq, int Lk, int N, int H)
{
    AttnSplit a;
    int kpw = g_attn_kpw;
    if (!kpw) {
        const int64_t tiles = (int64_t)((Lq + 31) / 32) * N * H;
        kpw = Lk <= 256 ? 64 : 128;
        while (kpw < 512 && tiles * ((Lk + kpw - 1) / kpw) > 4096) kpw *= 2;
    }
    const int waves = (Lk + kpw - 1) / kpw;
    a.keys_per_wave = kpw;
    a.nw = waves < g_attn_nw ? waves : g_attn_nw;
    a.wg_splits = (waves + a.nw - 1) / a.nw;
    return a;
}
namespace mpf {
int set_attn_option(const char* key, int v)
{
    if (!strcmp(key, "attn_occ")) { if (v != 2 && v != 4) return -1; g_attn_occ = v; return 0; }
    if (!strcmp(key, "attn_nw")) { if (v < 1 || v > kMaxNW) return -1; g_attn_nw = v; return 0; }
    if (!strcmp(key, "attn_kpw")) { if (v && (v < 64 || (v & 63))) return -1; g_attn_kpw = v; return 0; }
    return 1;
}
}  // namespace mpf

static size_t attn_part_bytes(int Lq, int Lk, int N, int H)
{
    const int splits = attn_split(Lq, Lk, N, H).wg_splits;
    return (((size_t)splits * N * H * Lq * (kHD + 2) * sizeof(float)) + 255) & ~(size_t)255;
}
static size_t attn_ld2_bytes(int LqP, int N, int H) { return (((size_t)N * H * LqP * sizeof(float2)) + 255) & ~(size_t)255; }

extern "C" size_t mpf_attn_bwd_aux_bytes(int Lq, int Lk, int N, int H, int mask_images)
{
    if (Lq <= 0 || Lk <= 0 || N <= 0 || H <= 0 || mask_images < 0) return 0;
    const int LqP = (Lq + 31) / 32 * 32;
    return attn_ld2_bytes(LqP, N, H) + (size_t)mask_images * Lk * LqP;
}

// (covers the split partials AND the aux operands of a backward called without aux: mask per image assumed)
extern "C" size_t mpf_attn_workspace_bytes(int Lq, int Lk, int N, int H)
{
    if (Lq <= 0 || Lk <= 0 || N <= 0 || H <= 0) return 0;
    return attn_part_bytes(Lq, Lk, N, H) + mpf_attn_bwd_aux_bytes(Lq, Lk, N, H, N);
}

extern "C" int mpf_attn_forward(const void* q, const void* k, const void* vt, const uint8_t* mask, int mask_per_image,
                                void* out, float* lse, int Lq, int Lk, int N, int H, int head_dim, float scale,
                                void* workspace, size_t workspace_bytes, void* stream)
{
    return mpf_attn_forward_kv(q, k, 0, 0, vt, mask, mask_per_image, out, lse, Lq, Lk, N, H, head_dim, scale, workspace,
                               workspace_bytes, stream);
}

extern "C" int mpf_attn_forward_kv(const void* q, const void* k, int64_t k_row_stride, int64_t k_img_stride, const void* vt,
                                   const uint8_t* mask, int mask_per_image, void* out, float* lse, int Lq, int Lk, int N, int H,
                                   int head_dim, float scale, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!q || !k || !vt || !out || !workspace) return mpf::fail(MPF_E_NULL, "attn_forward: NULL buffer");
    if (k_row_stride < 0 || k_img_stride < 0 || (k_row_stride | k_img_stride) % 8 || ((uintptr_t)k & 15))
        return mpf::fail(MPF_E_SHAPE, "attn_forward: K strides must be non-negative multiples of 8 elements, K 16-byte aligned");
    if (head_dim != kHD) return mpf::fail(MPF_E_SHAPE, "attn_forward: head_dim must be 32");
    if (Lq <= 0 || Lk <= 0 || N <= 0 || H <= 0) return mpf::fail(MPF_E_SHAPE, "attn_forward: bad sizes");
    if (workspace_bytes < mpf_attn_workspace_bytes(Lq, Lk, N, H)) return mpf::fail(MPF_E_SHAPE, "attn_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    AttnParams p;
    p.q = (const __hip_bfloat16*)q; p.k = (const __hip_bfloat16*)k; p.vt = (const __hip_bfloat16*)vt;
    p.mask = mask; p.mask_stride_n = mask_per_image ? (int64_t)Lq * Lk : 0;
    p.Lq = Lq; p.Lk = Lk; p.N = N; p.H = H; p.E = H * kHD; p.scale = scale;
    p.k_row = k_row_stride ? k_row_stride : (int64_t)N * p.E;
    p.k_img = k_row_stride ? k_img_stride : p.E;
    const AttnSplit sp = attn_split(Lq, Lk, N, H);
    p.splits = sp.wg_splits; p.nw = sp.nw; p.keys_per_split = sp.keys_per_wave;
    p.out = (__hip_bfloat16*)out; p.lse = lse;
    p.part_o = (float*)workspace;
    p.part_ml = p.part_o + (size_t)p.splits * N * H * Lq * kHD;
    constexpr int QS = 2;
    const int qtiles = (Lq + 16 * QS - 1) / (16 * QS);
    const size_t lds = (size_t)p.nw * kMergeWave * sizeof(float);
    mpf::prof_begin(st);
    mpf::set_kernel("attn_fwd_kernel<2>");
    {
        const dim3 grid(qtiles * p.splits, H, N), block(64 * p.nw);
        const bool al = (Lk & 7) == 0 && ((uintptr_t)p.vt & 15) == 0 && (!mask || ((uintptr_t)mask & 7) == 0);
        static_assert(kMaxNW * kMergeWave * sizeof(float) <= 64 * 1024, "the merge buffer fits the default dynamic-LDS limit");
        auto launch = [&](auto kfn) -> int {
            hipLaunchKernelGGL(kfn, grid, block, lds, st, p);
            return 0;
        };
        int e;
        if (al && mask && g_attn_occ == 4) e = launch(attn_fwd_kernel<QS, true, true, 4>);
        else if (al && g_attn_occ == 4) e = launch(attn_fwd_kernel<QS, true, false, 4>);
        else if (al && mask) e = launch(attn_fwd_kernel<QS, true, true>);
        else if (al) e = launch(attn_fwd_kernel<QS, true, false>);
        else if (mask) e = launch(attn_fwd_kernel<QS, false, true>);
        else e = launch(attn_fwd_kernel<QS, false, false>);
        if (e) return e;
    }
    mpf::prof_end("attn_fwd_kernel<2>", st, 2.0 * ((double)Lk * N * p.E * 2 + (double)Lq * N * p.E) + (mask ? (double)N * Lq * Lk : 0.0),
                  4.0 * Lq * (double)Lk * p.E * N);      // QK^T + PV
    if (p.splits > 1) {
        const int total = N * H * Lq * kHD;
        hipLaunchKernelGGL(attn_combine_kernel, dim3((total + 255) / 256), dim3(256), 0, st, p.part_o, p.part_ml,
                           (__hip_bfloat16*)out, lse, Lq, N, H, p.E, p.splits);
    }
    return mpf::check(hipGetLastError(), "mpf_attn_forward");
}

extern "C" int mpf_attn_backward(const void* q, const void* k, const void* v, const void* kT, const void* qT,
                                 const void* dout, const void* doutT, const uint8_t* mask, int mask_per_image,
                                 const float* lse, const float* delta, void* dq, void* dk, void* dv,
                                 int Lq, int LqP, int Lk, int N, int H, int head_dim, float scale,
                                 void* workspace, size_t workspace_bytes, void* stream)
{
    return mpf_attn_backward_kv(q, k, v, 0, 0, kT, qT, dout, doutT, mask, mask_per_image, lse, delta, dq, dk, dv, 0, 0, Lq, LqP, Lk, N, H,
                                head_dim, scale, workspace, workspace_bytes, stream);
}

extern "C" int mpf_attn_backward_kv(const void* q, const void* k, const void* v, int64_t kv_row_stride, int64_t kv_img_stride,
                                    const void* kT, const void* qT, const void* dout, const void* doutT, const uint8_t* mask,
                                    int mask_per_image, const float* lse, const float* delta, void* dq, void* dk, void* dv,
                                    int64_t dkv_row_stride, int64_t dkv_img_stride, int Lq, int LqP, int Lk, int N, int H,
                                    int head_dim, float scale, void* workspace, size_t workspace_bytes, void* stream)
{
    return mpf_attn_backward_kv_aux(q, k, v, kv_row_stride, kv_img_stride, kT, qT, dout, doutT, mask, mask_per_image, lse, delta, dq, dk,
                                    dv, dkv_row_stride, dkv_img_stride, Lq, LqP, Lk, N, H, head_dim, scale, workspace, workspace_bytes,
                                    nullptr, stream);
}

extern "C" int mpf_attn_backward_kv_aux(const void* q, const void* k, const void* v, int64_t kv_row_stride, int64_t kv_img_stride,
                                        const void* kT, const void* qT, const void* dout, const void* doutT, const uint8_t* mask,
                                        int mask_per_image, const float* lse, const float* delta, void* dq, void* dk, void* dv,
                                        int64_t dkv_row_stride, int64_t dkv_img_stride, int Lq, int LqP, int Lk, int N, int H,
                                        int head_dim, float scale, void* workspace, size_t workspace_bytes, const void* aux,
                                        void* stream)
{
    if (!q || !k || !v || !kT || !qT || !dout || !doutT || !lse || !delta || !dq || !dk || !dv || !workspace)
        return mpf::fail(MPF_E_NULL, "attn_backward: NULL buffer");
    if (LqP != (Lq + 31) / 32 * 32) return mpf::fail(MPF_E_SHAPE, "attn_backward: LqP must be Lq rounded up to a multiple of 32");
    if (kv_row_stride < 0 || kv_img_stride < 0 || dkv_row_stride < 0 || dkv_img_stride < 0 ||
        (kv_row_stride | kv_img_stride | dkv_row_stride | dkv_img_stride) % 8 ||
        (((uintptr_t)k | (uintptr_t)v | (uintptr_t)dk | (uintptr_t)dv) & 15))
        return mpf::fail(MPF_E_SHAPE, "attn_backward: K / V strides must be non-negative multiples of 8 elements, buffers 16-byte aligned");
    if (head_dim != kHD) return mpf::fail(MPF_E_SHAPE, "attn_backward: head_dim must be 32");
    if (Lq <= 0 || Lk <= 0 || N <= 0 || H <= 0 || LqP < Lq || (LqP & 31)) return mpf::fail(MPF_E_SHAPE, "attn_backward: bad sizes");
    const int mimgs = mask ? (mask_per_image ? N : 1) : 0;
    if (workspace_bytes < attn_part_bytes(Lq, Lk, N, H) + (aux ? 0 : mpf_attn_bwd_aux_bytes(Lq, Lk, N, H, mimgs)))
        return mpf::fail(MPF_E_SHAPE, "attn_backward: workspace too small");
    if ((uintptr_t)workspace & 15 || (aux && ((uintptr_t)aux & 15))) return mpf::fail(MPF_E_SHAPE, "attn_backward: workspace / aux must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    // aux operands of the dK / dV kernel: (lse, delta) pairs padded to LqP | the mask transposed (mpf_attn_bwd_prep_aux makes
    // them in the launch that also makes qT / doT / delta; without aux they are made here, in the workspace)
    const char* auxp = (const char*)aux;
    if (!aux) {
        char* a = (char*)workspace + attn_part_bytes(Lq, Lk, N, H);
        const int ld_blocks = (N * H * Lq + 255) / 256, mq = (LqP + 63) / 64, mk = (Lk + 63) / 64;
        mpf::set_kernel("attn_bwd_aux_kernel");
        hipLaunchKernelGGL(attn_bwd_aux_kernel, dim3(ld_blocks + mimgs * mq * mk), dim3(256), 0, st, lse, delta, mask, (float2*)a,
                           (uint8_t*)(a + attn_ld2_bytes(LqP, N, H)), Lq, LqP, Lk, N * H, ld_blocks, mq, mk);
        auxp = a;
    }
    AttnBwdParams p;
    p.ld2 = (const float2*)auxp;
    p.maskT = mask ? (const uint8_t*)(auxp + attn_ld2_bytes(LqP, N, H)) : nullptr;
    p.maskT_stride_n = mask_per_image ? (int64_t)Lk * LqP : 0;
    p.q = (const __hip_bfloat16*)q; p.k = (const __hip_bfloat16*)k; p.v = (const __hip_bfloat16*)v;
    p.kT = (const __hip_bfloat16*)kT; p.qT = (const __hip_bfloat16*)qT;
    p.dout = (const __hip_bfloat16*)dout; p.doT = (const __hip_bfloat16*)doutT;
    p.mask = mask; p.mask_stride_n = mask_per_image ? (int64_t)Lq * Lk : 0;
    p.lse = lse; p.delta = delta;
    p.dk = (__hip_bfloat16*)dk; p.dv = (__hip_bfloat16*)dv; p.part_dq = (float*)workspace;
    p.Lq = Lq; p.LqP = LqP; p.Lk = Lk; p.N = N; p.H = H; p.E = H * kHD; p.scale = scale;
    p.kv_row = kv_row_stride ? kv_row_stride : (int64_t)N * p.E;
    p.kv_img = kv_row_stride ? kv_img_stride : p.E;
    p.dkv_row = dkv_row_stride ? dkv_row_stride : (int64_t)N * p.E;
    p.dkv_img = dkv_row_stride ? dkv_img_stride : p.E;
    const AttnSplit sp = attn_split(Lq, Lk, N, H);
    p.splits = sp.wg_splits; p.nw = sp.nw; p.keys_per_split = sp.keys_per_wave;
    p.dq = (__hip_bfloat16*)dq;
    const double bytes = 2.0 * (4.0 * Lk * N * p.E + 4.0 * Lq * N * p.E) + (mask ? 2.0 * N * Lq * Lk : 0.0);
    mpf::prof_begin(st);
    mpf::set_kernel("attn_bwd_kv_kernel");
    if (mask) hipLaunchKernelGGL(attn_bwd_kv_kernel<true>, dim3((Lk + 31) / 32, H, N), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(attn_bwd_kv_kernel<false>, dim3((Lk + 31) / 32, H, N), dim3(64), 0, st, p);
    mpf::prof_end("attn_bwd_kv_kernel", st, bytes * 0.5, 8.0 * Lq * (double)Lk * p.E * N);   // S, dP, dV, dK
    constexpr int QS = 2;
    const int qtiles = (Lq + 16 * QS - 1) / (16 * QS);
    mpf::prof_begin(st);
    mpf::set_kernel("attn_bwd_q_kernel<2>");
    {
        const dim3 grid(qtiles * p.splits, H, N), block(64 * p.nw);
        const size_t lds = (size_t)p.nw * 32 * kMergePitch * sizeof(float);
        const bool al = (Lk & 7) == 0 && ((uintptr_t)p.kT & 15) == 0 && (!mask || ((uintptr_t)mask & 7) == 0);
        static_assert(kMaxNW * kMergeWave * sizeof(float) <= 64 * 1024, "the merge buffer fits the default dynamic-LDS limit");
        auto launch = [&](auto kfn) -> int {
            hipLaunchKernelGGL(kfn, grid, block, lds, st, p);
            return 0;
        };
        int e;
        if (al && mask) e = launch(attn_bwd_q_kernel<QS, true, true>);
        else if (al) e = launch(attn_bwd_q_kernel<QS, true, false>);
        else if (mask) e = launch(attn_bwd_q_kernel<QS, false, true>);
        else e = launch(attn_bwd_q_kernel<QS, false, false>);
        if (e) return e;
    }
    mpf::prof_end("attn_bwd_q_kernel<2>", st, bytes * 0.5, 6.0 * Lq * (double)Lk * p.E * N);  // S, dP, dQ
    if (p.splits > 1) {
        const int total = N * H * Lq * kHD;
        hipLaunchKernelGGL(attn_sum_splits_kernel, dim3((total + 255) / 256), dim3(256), 0, st, p.part_dq,
                           (__hip_bfloat16*)dq, Lq, N, H, p.E, p.splits);
    }
    return mpf::check(hipGetLastError(), "mpf_attn_backward");
}

extern "C" int mpf_attn_transpose2(const void* a, const void* b, void* aT, void* bT, int L, int LP, int N, int E, void* stream)
{
    return mpf_attn_transpose2_strided(a, b, 0, 0, aT, bT, L, LP, N, E, stream);
}

extern "C" int mpf_attn_transpose2_strided(const void* a, const void* b, int64_t in_row_stride, int64_t in_img_stride, void* aT,
                                           void* bT, int L, int LP, int N, int E, void* stream)
{
    if (!a || !b || !aT || !bT) return mpf::fail(MPF_E_NULL, "attn_transpose2: NULL buffer");
    if (L <= 0 || LP < L || N <= 0 || E <= 0 || in_row_stride < 0 || in_img_stride < 0)
        return mpf::fail(MPF_E_SHAPE, "attn_transpose2: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int NE = N * E;
    mpf::set_kernel("attn_transpose2_kernel");
    hipLaunchKernelGGL(attn_transpose2_kernel, dim3((LP + 63) / 64, (NE + 63) / 64), dim3(256), 0, st,
                       (const __hip_bfloat16*)a, (const __hip_bfloat16*)b, (__hip_bfloat16*)aT, (__hip_bfloat16*)bT, L, LP, NE,
                       in_row_stride ? E : 0, in_row_stride, in_img_stride);
    return mpf::check(hipGetLastError(), "mpf_attn_transpose2");
}

extern "C" int mpf_attn_bwd_prep(const void* q, const void* dout, const void* out, void* qT, void* doT, float* delta, int Lq, int LqP,
                                 int N, int H, void* stream)
{
    if (!q || !dout || !out || !qT || !doT || !delta) return mpf::fail(MPF_E_NULL, "attn_bwd_prep: NULL buffer");
    if (Lq <= 0 || LqP < Lq || N <= 0 || H <= 0) return mpf::fail(MPF_E_SHAPE, "attn_bwd_prep: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int NE = N * H * kHD;
    const int tiles_x = (LqP + 63) / 64, n_tiles = tiles_x * ((NE + 63) / 64);
    const int delta_blocks = (Lq * N * H * kHD + 255) / 256;
    mpf::set_kernel("attn_bwd_prep_kernel");
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(n_tiles + delta_blocks), dim3(256), 0, st, (const __hip_bfloat16*)q,
                       (const __hip_bfloat16*)dout, (const __hip_bfloat16*)out, (__hip_bfloat16*)qT, (__hip_bfloat16*)doT, delta, Lq,
                       LqP, N, H, tiles_x, n_tiles);
    return mpf::check(hipGetLastError(), "mpf_attn_bwd_prep");
}

extern "C" int mpf_attn_bwd_prep_aux(const void* q, const void* dout, const void* out, const float* lse, const uint8_t* mask,
                                     int mask_per_image, int Lk, void* qT, void* doT, float* delta, void* aux, size_t aux_bytes, int Lq,
                                     int LqP, int N, int H, void* stream)
{
    if (!q || !dout || !out || !lse || !qT || !doT || !delta || !aux) return mpf::fail(MPF_E_NULL, "attn_bwd_prep_aux: NULL buffer");
    if (Lq <= 0 || Lk <= 0 || LqP != (Lq + 31) / 32 * 32 || N <= 0 || H <= 0) return mpf::fail(MPF_E_SHAPE, "attn_bwd_prep_aux: bad sizes");
    const int mimgs = mask ? (mask_per_image ? N : 1) : 0;
    if (aux_bytes < mpf_attn_bwd_aux_bytes(Lq, Lk, N, H, mimgs) || ((uintptr_t)aux & 15))
        return mpf::fail(MPF_E_SHAPE, "attn_bwd_prep_aux: aux too small or not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int NE = N * H * kHD;
    const int tiles_x = (LqP + 63) / 64, n_tiles = tiles_x * ((NE + 63) / 64);
    const int delta_blocks = (Lq * N * H * kHD + 255) / 256;
    const int mq = (LqP + 63) / 64, mk = (Lk + 63) / 64;
    mpf::set_kernel("attn_bwd_prep_aux_kernel");
    hipLaunchKernelGGL(attn_bwd_prep_aux_kernel, dim3(n_tiles + delta_blocks + mimgs * mq * mk), dim3(256), 0, st,
                       (const __hip_bfloat16*)q, (const __hip_bfloat16*)dout, (const __hip_bfloat16*)out, lse, mask, (__hip_bfloat16*)qT,
                       (__hip_bfloat16*)doT, delta, (float2*)aux, (uint8_t*)aux + attn_ld2_bytes(LqP, N, H), Lq, LqP, Lk, N, H, tiles_x,
                       n_tiles, delta_blocks, mq, mk);
    return mpf::check(hipGetLastError(), "mpf_attn_bwd_prep_aux");
}

extern "C" int mpf_attn_delta(const void* dout, const void* out, float* delta, int Lq, int N, int H, void* stream)
{
    if (!dout || !out || !delta) return mpf::fail(MPF_E_NULL, "attn_delta: NULL buffer");
    if (Lq <= 0 || N <= 0 || H <= 0) return mpf::fail(MPF_E_SHAPE, "attn_delta: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int total = Lq * N * H * kHD;
    mpf::set_kernel("attn_delta_kernel");
    hipLaunchKernelGGL(attn_delta_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const __hip_bfloat16*)dout,
                       (const __hip_bfloat16*)out, delta, Lq, N, H);
    return mpf::check(hipGetLastError(), "mpf_attn_delta");
}
