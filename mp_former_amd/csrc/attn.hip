// Masked multi-head attention (cross- and self-attention of the decoder) as bf16 MFMA tiles for
// MI355X (gfx950): softmax(mask(Q K^T / sqrt(hd))) V per (image, head), head dim 32.
// Reference semantics: nn.MultiheadAttention inside CrossAttentionLayer / SelfAttentionLayer
// (mask2former_transformer_decoder.py:42-52, :100-112) with a boolean attn_mask (True = -inf).
//
// Shapes of this workload: few queries (Qtot = 100..300), many keys (HW up to 16 384 per level),
// 8 heads x 32 dims.  The work is bound by streaming K / V and the byte mask, not by MFMA, so the
// design is flash-decoding-like:
//   * one WAVE per (16*QS query rows, head, image, key split); keys are split across waves so that the
//     chip is filled although there are only ~14 query tiles; partial (max, sum, O) per split are merged
//     by a small combine kernel;
//   * S^T = K Q^T with v_mfma_f32_16x16x32_bf16: K = head dim = 32 in ONE instruction per 16 keys x 16
//     queries.  Operands come straight from global memory in MFMA layout: a lane loads 16 contiguous
//     bytes of a K row (A operand) / of a Q row (B operand) — no LDS staging, every K byte is read once
//     per query tile;
//   * the transposed product puts the queries on the lane axis (C layout: col = lane&15 = query), so a
//     lane owns ONE query row of P: the online-softmax scale factors are per-lane scalars and the row
//     max needs two cross-lane steps (xor 16, 32) per 32 keys;
//   * O^T = V^T P^T: the 8 probabilities a lane holds after two 16-key tiles ARE its B fragment once
//     the 32 keys of the block are relabelled (k = 8g+j <-> key 4g+j / 16+4g+j-4; sums do not care),
//     and V is consumed as V^T [N, 256, Lk] (what the 1x1 projection of the NCHW feature map produces
//     anyway), so the A fragment is two 8-byte loads per lane;
//   * the byte attention mask [N, Lq, Lk] (one copy for all heads) is read as one dword per lane per
//     16-key tile.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

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
// 4 consecutive elements of a row starting at i (zero past `limit`); AL: i and limit multiples of 4.  AL is a
// COMPILE-TIME switch: as a run-time flag it put every one of these loads behind a branch, i.e. exactly the serialised
// round trips the comment above is about (9 per 32-key step in the forward kernel).
template <bool AL>
__device__ __forceinline__ bf16x4 ld4_clamped(const __hip_bfloat16* row, int i, int limit)
{
    const bf16x4 z = {0, 0, 0, 0};
    if constexpr (AL) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(row + max(min(i, limit - 4), 0));
        return i < limit ? v : z;
    } else {
        bf16x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 e = *reinterpret_cast<const __bf16*>(row + min(i + j, limit - 1));
            r[j] = i + j < limit ? e : (__bf16)0.f;
        }
        return r;
    }
}

// the 4 mask bytes of keys key0 .. key0 + 3 of one query row as a word (nonzero byte = masked), clamped reads
template <bool AL>
__device__ __forceinline__ uint32_t ld_mask4(const uint8_t* mrow, int key0, int limit)
{
    if constexpr (AL) {
        return *reinterpret_cast<const uint32_t*>(mrow + max(min(key0, limit - 4), 0));
    } else {
        uint32_t mw = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) mw |= (mrow[min(key0 + r, limit - 1)] ? 0xFFu : 0u) << (8 * r);
        return mw;
    }
}

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
    int Lq, Lk, N, H, E, splits, keys_per_split;
    float scale;
    int64_t k_row, k_img;         // element strides of k between sequence positions / images (N*E, E when dense)
};

// QS = number of 16-row query sub-tiles per wave
template <int QS, bool AL, bool MK>
__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnParams p)
{
    const int lane = threadIdx.x, c16 = lane & 15, g = lane >> 4;
    const int qtiles = (p.Lq + 16 * QS - 1) / (16 * QS);
    const int qt = blockIdx.x % qtiles, split = blockIdx.x / qtiles;
    const int h = blockIdx.y, n = blockIdx.z;
    const int q0 = qt * 16 * QS;
    const int kb0 = split * p.keys_per_split;
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
        uint32_t mw[QS][2];
    };
    auto load_step = [&](Step& st, const int kk) {
        // S^T tiles: keys kk+16t .. +15
#pragma unroll
        for (int t = 0; t < 2; ++t) st.ak[t] = ld8_rows(kb, kk + 16 * t + c16, kb1, p.k_row, 8 * g);
        // V^T fragments: rows d = 16*dt + c16, keys {kk+4g..+3} and {kk+16+4g..+3}
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const __hip_bfloat16* vr = vb + (int64_t)(16 * dt + c16) * p.Lk;
            // (Lk % 4 == 0: 8-byte aligned, whole quads in range or not)
            const bf16x4 lo = ld4_clamped<AL>(vr, kk + 4 * g, kb1), hi = ld4_clamped<AL>(vr, kk + 16 + 4 * g, kb1);
            st.av[dt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int s = 0; s < QS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                st.mw[s][t] = 0;
                if constexpr (MK) {
                    const int qi = q0 + 16 * s + c16;
                    const uint8_t* mrow = p.mask + (int64_t)n * p.mask_stride_n + (int64_t)min(qi, p.Lq - 1) * p.Lk;
                    st.mw[s][t] = ld_mask4<AL>(mrow, kk + 16 * t + 4 * g, kb1);
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
                // lane (query c16, group g) holds keys kk + 16t + 4g + r
                const int key0 = kk + 16 * t + 4 * g;
                const uint32_t mw = st.mw[s][t];
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
    // (requesting step i + 1 before the MFMAs of step i — two alternating Step sets — was measured: no gain, the four
    // resident waves per SIMD already cover the round trip; the step is bound by its ~300 VALU instructions of softmax
    // bookkeeping against 8 MFMAs)
    Step st;
    for (int kk = kb0; kk < kb1; kk += 32) {
        load_step(st, kk);
        compute_step(st, kk);
    }
    // ---- write the partial result of this split -------------------------------------------------------
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        const int qi = q0 + 16 * s + c16;
        const float lt = xsum(l[s]);
        if (qi < p.Lq) {
            const int64_t row = (((int64_t)split * p.N + n) * p.H + h) * p.Lq + qi;
            // O^T C layout: lane (query c16, g) holds d = 16*dt + 4g + r
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                *reinterpret_cast<f32x4*>(p.part_o + row * kHD + 16 * dt + 4 * g) = o[s][dt];
            if (g == 0) { p.part_ml[row * 2] = m[s]; p.part_ml[row * 2 + 1] = lt; }
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
    float M = kNegInf;
#pragma unroll 8
    for (int s = 0; s < splits; ++s) M = fmaxf(M, part_ml[(s * sstride + row) * 2]);
    float L = 0.f, O = 0.f;
#pragma unroll 8
    for (int s = 0; s < splits; ++s) {
        const float ms = part_ml[(s * sstride + row) * 2];
        const float w = (ms == kNegInf) ? 0.f : __expf(ms - M);
        L += w * part_ml[(s * sstride + row) * 2 + 1];
        O += w * part_o[(s * sstride + row) * kHD + d];
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
    __hip_bfloat16 *dk, *dv;                       // [Lk, N, E]
    float* part_dq;                                // [splits, N, H, Lq, 32]
    int Lq, LqP, Lk, N, H, E, splits, keys_per_split;
    float scale;
    int64_t kv_row, kv_img;        // element strides of k, v between sequence positions / images (N*E, E when dense)
    int64_t dkv_row, dkv_img;      // ... and of dk, dv
};

__device__ __forceinline__ bf16x8 load8(const __hip_bfloat16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero8() { return bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }

// two 8-byte pieces (4 + 4 consecutive elements) of a row -> one MFMA fragment
template <bool AL>
__device__ __forceinline__ bf16x8 load4x2(const __hip_bfloat16* row, int i0, int i1, int limit)
{
    const bf16x4 lo = ld4_clamped<AL>(row, i0, limit), hi = ld4_clamped<AL>(row, i1, limit);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

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
    const float* lse = p.lse + ((int64_t)n * p.H + h) * p.Lq;
    const float* dl = p.delta + ((int64_t)n * p.H + h) * p.Lq;
    const __hip_bfloat16* qTb = p.qT + ((int64_t)n * p.E + h * kHD) * p.LqP;
    const __hip_bfloat16* doTb = p.doT + ((int64_t)n * p.E + h * kHD) * p.LqP;

    for (int qq = 0; qq < p.Lq; qq += 32) {
        bf16x8 aq[2], ado[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qi = qq + 16 * qt + c16;
            aq[qt] = ld8_rows(p.q + hoff, qi, p.Lq, rowE, 8 * g);
            ado[qt] = ld8_rows(p.dout + hoff, qi, p.Lq, rowE, 8 * g);
        }
        float ls[2][4], de[2][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = qq + 16 * qt + 4 * g + r;
                const float lv = lse[min(qi, p.Lq - 1)], dv_ = dl[min(qi, p.Lq - 1)];
                ls[qt][r] = qi < p.Lq ? lv : 0.f;
                de[qt][r] = qi < p.Lq ? dv_ : 0.f;
            }
        // everything this step reads is requested before its first MFMA: the mask bytes (keys on the lane axis: one
        // byte per (query row, key)) and the transposed Q / dO fragments of the dV / dK products
        uint8_t mb[2][2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mb[kt][qt][r] = 0;
                    if constexpr (MK) {
                        const int qi = qq + 16 * qt + 4 * g + r, key = kb + 16 * kt + c16;
                        mb[kt][qt][r] = p.mask[(int64_t)n * p.mask_stride_n + (int64_t)min(qi, p.Lq - 1) * p.Lk + min(key, p.Lk - 1)];
                    }
                }
        bf16x8 adoT[2], aqT[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int64_t ro = (int64_t)(16 * dt + c16) * p.LqP;
            adoT[dt] = load4x2<true>(doTb + ro, qq + 4 * g, qq + 16 + 4 * g, p.LqP);
            aqT[dt] = load4x2<true>(qTb + ro, qq + 4 * g, qq + 16 + 4 * g, p.LqP);
        }
        bf16x8 bp[2], bds[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = kb + 16 * kt + c16;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 z = {0, 0, 0, 0};
                const f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[qt], bk[kt], z, 0, 0, 0);
                const f32x4 dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ado[qt], bv[kt], z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = qq + 16 * qt + 4 * g + r;
                    const bool dead = (qi >= p.Lq) || (key >= p.Lk) || (ls[qt][r] == kNegInf) || (mb[kt][qt][r] != 0);
                    const float pr = dead ? 0.f : __expf(sacc[r] * p.scale - ls[qt][r]);
                    const float ds = pr * (dpacc[r] - de[qt][r]) * p.scale;
                    bp[kt][4 * qt + r] = (__bf16)pr;
                    bds[kt][4 * qt + r] = (__bf16)ds;
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                dvt[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adoT[dt], bp[kt], dvt[kt][dt], 0, 0, 0);
                dkt[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aqT[dt], bds[kt], dkt[kt][dt], 0, 0, 0);
            }
        }
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
__global__ __launch_bounds__(64) void attn_bwd_q_kernel(AttnBwdParams p)
{
    const int lane = threadIdx.x, c16 = lane & 15, g = lane >> 4;
    const int qtiles = (p.Lq + 16 * QS - 1) / (16 * QS);
    const int qt = blockIdx.x % qtiles, split = blockIdx.x / qtiles;
    const int h = blockIdx.y, n = blockIdx.z;
    const int q0 = qt * 16 * QS;
    const int kb0 = split * p.keys_per_split;
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
        uint32_t mw[QS][2];
    };
    auto load_step = [&](Step& st, const int kk) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = kk + 16 * t + c16;
            st.ak[t] = ld8_rows(p.k + kvoff, key, kb1, p.kv_row, 8 * g);
            st.av[t] = ld8_rows(p.v + kvoff, key, kb1, p.kv_row, 8 * g);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            st.akT[dt] = load4x2<AL>(kTb + (int64_t)(16 * dt + c16) * p.Lk, kk + 4 * g, kk + 16 + 4 * g, kb1);
#pragma unroll
        for (int s = 0; s < QS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                st.mw[s][t] = 0;
                if constexpr (MK) {
                    const int qi = q0 + 16 * s + c16;
                    const uint8_t* mrow = p.mask + (int64_t)n * p.mask_stride_n + (int64_t)min(qi, p.Lq - 1) * p.Lk;
                    st.mw[s][t] = ld_mask4<AL>(mrow, kk + 16 * t + 4 * g, kb1);
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
                const int key0 = kk + 16 * t + 4 * g;
                const uint32_t mw = st.mw[s][t];
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
    // (requesting step i + 1 before the MFMAs of step i — two alternating Step sets — was measured: no gain, the four
    // resident waves per SIMD already cover the round trip; the step is bound by its ~300 VALU instructions of softmax
    // bookkeeping against 8 MFMAs)
    Step st;
    for (int kk = kb0; kk < kb1; kk += 32) {
        load_step(st, kk);
        compute_step(st, kk);
    }
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        const int qi = q0 + 16 * s + c16;
        if (qi < p.Lq) {
            const int64_t row = (((int64_t)split * p.N + n) * p.H + h) * p.Lq + qi;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                *reinterpret_cast<f32x4*>(p.part_dq + row * kHD + 16 * dt + 4 * g) = dq[s][dt];
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
                                            float* __restrict__ delta, int Lq, int N, int H, const int block)
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
    }
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

}  // namespace

// Key splits of the forward / dQ kernels: one WAVE per (32 queries, head, image, split) and each wave
// walks its keys in dependent 32-key steps, so the chip needs several thousand waves to hide the load
// latency of a step: aim at ~2048 waves (measured at config B: 1024 / 2048 / 4096 / 8192 waves -> 39.3 / 38.95 / 39.15 /
// 39.2 ms per step; more splits also mean more partials for the combine kernels), at least 64 keys and at most 1024
// keys per split.
static int attn_splits(int Lq, int Lk, int N, int H)
{
    const int qtiles = (Lq + 31) / 32;
    int splits = (2048 + qtiles * H * N - 1) / (qtiles * H * N);
    const int max_splits = (Lk + 63) / 64, min_splits = (Lk + 1023) / 1024;
    splits = splits > max_splits ? max_splits : splits;
    splits = splits < min_splits ? min_splits : splits;
    return splits < 1 ? 1 : splits;
}

extern "C" size_t mpf_attn_workspace_bytes(int Lq, int Lk, int N, int H)
{
    if (Lq <= 0 || Lk <= 0 || N <= 0 || H <= 0) return 0;
    const int splits = attn_splits(Lq, Lk, N, H);
    return (size_t)splits * N * H * Lq * (kHD + 2) * sizeof(float);
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
    p.splits = attn_splits(Lq, Lk, N, H);
    p.keys_per_split = (((Lk + p.splits - 1) / p.splits) + 31) & ~31;
    p.splits = (Lk + p.keys_per_split - 1) / p.keys_per_split;          // (rounding up to 32 keys may need fewer)
    p.part_o = (float*)workspace;
    p.part_ml = p.part_o + (size_t)p.splits * N * H * Lq * kHD;
    constexpr int QS = 2;
    const int qtiles = (Lq + 16 * QS - 1) / (16 * QS);
    mpf::prof_begin(st);
    mpf::set_kernel("attn_fwd_kernel<2>");
    {
        const dim3 grid(qtiles * p.splits, H, N);
        const bool al = (Lk & 3) == 0 && ((uintptr_t)p.vt & 7) == 0 && (!mask || ((uintptr_t)mask & 3) == 0);
        if (al && mask) hipLaunchKernelGGL((attn_fwd_kernel<QS, true, true>), grid, dim3(64), 0, st, p);
        else if (al) hipLaunchKernelGGL((attn_fwd_kernel<QS, true, false>), grid, dim3(64), 0, st, p);
        else if (mask) hipLaunchKernelGGL((attn_fwd_kernel<QS, false, true>), grid, dim3(64), 0, st, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<QS, false, false>), grid, dim3(64), 0, st, p);
    }
    mpf::prof_end("attn_fwd_kernel<2>", st, 2.0 * ((double)Lk * N * p.E * 2 + (double)Lq * N * p.E) + (mask ? (double)N * Lq * Lk : 0.0),
                  4.0 * Lq * (double)Lk * p.E * N);      // QK^T + PV
    const int total = N * H * Lq * kHD;
    hipLaunchKernelGGL(attn_combine_kernel, dim3((total + 255) / 256), dim3(256), 0, st, p.part_o, p.part_ml,
                       (__hip_bfloat16*)out, lse, Lq, N, H, p.E, p.splits);
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
    if (!q || !k || !v || !kT || !qT || !dout || !doutT || !lse || !delta || !dq || !dk || !dv || !workspace)
        return mpf::fail(MPF_E_NULL, "attn_backward: NULL buffer");
    if (kv_row_stride < 0 || kv_img_stride < 0 || dkv_row_stride < 0 || dkv_img_stride < 0 ||
        (kv_row_stride | kv_img_stride | dkv_row_stride | dkv_img_stride) % 8 ||
        (((uintptr_t)k | (uintptr_t)v | (uintptr_t)dk | (uintptr_t)dv) & 15))
        return mpf::fail(MPF_E_SHAPE, "attn_backward: K / V strides must be non-negative multiples of 8 elements, buffers 16-byte aligned");
    if (head_dim != kHD) return mpf::fail(MPF_E_SHAPE, "attn_backward: head_dim must be 32");
    if (Lq <= 0 || Lk <= 0 || N <= 0 || H <= 0 || LqP < Lq || (LqP & 31)) return mpf::fail(MPF_E_SHAPE, "attn_backward: bad sizes");
    if (workspace_bytes < mpf_attn_workspace_bytes(Lq, Lk, N, H)) return mpf::fail(MPF_E_SHAPE, "attn_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    AttnBwdParams p;
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
    p.splits = attn_splits(Lq, Lk, N, H);
    p.keys_per_split = (((Lk + p.splits - 1) / p.splits) + 31) & ~31;
    p.splits = (Lk + p.keys_per_split - 1) / p.keys_per_split;          // (rounding up to 32 keys may need fewer)
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
        const dim3 grid(qtiles * p.splits, H, N);
        const bool al = (Lk & 3) == 0 && ((uintptr_t)p.kT & 7) == 0 && (!mask || ((uintptr_t)mask & 3) == 0);
        if (al && mask) hipLaunchKernelGGL((attn_bwd_q_kernel<QS, true, true>), grid, dim3(64), 0, st, p);
        else if (al) hipLaunchKernelGGL((attn_bwd_q_kernel<QS, true, false>), grid, dim3(64), 0, st, p);
        else if (mask) hipLaunchKernelGGL((attn_bwd_q_kernel<QS, false, true>), grid, dim3(64), 0, st, p);
        else hipLaunchKernelGGL((attn_bwd_q_kernel<QS, false, false>), grid, dim3(64), 0, st, p);
    }
    mpf::prof_end("attn_bwd_q_kernel<2>", st, bytes * 0.5, 6.0 * Lq * (double)Lk * p.E * N);  // S, dP, dQ
    const int total = N * H * Lq * kHD;
    hipLaunchKernelGGL(attn_sum_splits_kernel, dim3((total + 255) / 256), dim3(256), 0, st, p.part_dq,
                       (__hip_bfloat16*)dq, Lq, N, H, p.E, p.splits);
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
