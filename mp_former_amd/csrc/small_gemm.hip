// bf16 GEMMs of the transformer decoder's query-side Linear layers (Qtot * N ~ 200-300 rows):
// forward, input gradient and weight gradient (+ bias gradient) of y = x W^T + b, optional ReLU.
//
// Reference: the nn.Linear / nn.MultiheadAttention projections of SelfAttentionLayer /
// CrossAttentionLayer / FFNLayer / MLP (mask2former_transformer_decoder.py:19-206) under autocast.
// These problems are 30-240 MFLOP: the library picks one or two 256x256 macro-tiles for them (13 us
// on one CU); here a block of 4 waves owns one 16 x (16*NJ) output tile, splits the contraction between its
// waves and feeds v_mfma_f32_16x16x32_bf16 straight from global memory (the operands are L2 resident),
// so a 228x256x256 problem is 240 blocks of 4 waves x 2 MFMA steps.
//
//   C[i][j] = sum_k A(i,k) * B(j,k) (+ bias[j]) (+ Cin[i][j]) (ReLU),   i < I, j < J, k < Kc,   fp32 accumulation
//
// Each operand is addressed through (row stride, contraction stride), one of which must be 1:
//   contraction-contiguous: a lane's 8 consecutive k are one 16-B load         (x, W in the forward)
//   row-contiguous:         8 two-byte loads, 16 lanes cover 32 contiguous B   (W in dX; dY and x in dW)
// so the three GEMMs of a Linear need no transposed copies.  `gate` (same addressing as A) applies
// the ReLU backward to A on the fly: A(i,k) is used only where gate(i,k) > 0.  `rowsum_a` returns
// sum_k A(i,k) after the gate — the bias gradient in the dW form (A(n,m) = dY[m,n]).
// MFMA is issued as D^T = B . A^T so a lane owns 4 consecutive output columns (8-B bf16 stores).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "mpf_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

struct SG {
    const u16* a;
    const u16* gate;
    const u16* b;
    const u16* bias;
    const u16* cin;
    u16* c;
    u16* rowsum;
    int64_t a_rs, a_ks, b_rs, b_ks, ldc, ldcin;
    int64_t a_bs, c_bs;              // block strides (elements) of the blocked forms, see mpf_small_gemm_bf16_blocked
    int a_blk, c_blk;
    int I, J, Kc, relu, n_it, n_waves;
};

union Frag {
    uint4 q;
    unsigned w[4];
    bf16x8 v;
};

__device__ __forceinline__ unsigned f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// 8 consecutive-k elements of one operand row for this lane.  CONTIG: one 16-B load; else 8 u16 loads
// at stride ks.  MASK: elements with k >= Kc read as zero (loads stay unconditional: clamped address).
template <bool CONTIG, bool MASK>
__device__ __forceinline__ Frag load_frag(const u16* __restrict__ row, int64_t ks, int k, int Kc)
{
    Frag f;
    if (CONTIG) {
        const int kc = MASK ? min(k, Kc - 8) : k;
        f.q = *reinterpret_cast<const uint4*>(row + kc);
        if (MASK && k >= Kc) f.q = make_uint4(0u, 0u, 0u, 0u);
    } else {
        unsigned e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = MASK ? min(k + j, Kc - 1) : k + j;
            e[j] = row[(int64_t)kk * ks];
            if (MASK && k + j >= Kc) e[j] = 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) f.w[j] = e[2 * j] | (e[2 * j + 1] << 16);
    }
    return f;
}

__device__ __forceinline__ unsigned gate_word(unsigned v, unsigned g)
{
    const unsigned lo = ((int)(g << 16) > 0) ? 0x0000ffffu : 0u;
    const unsigned hi = ((int)(g & 0xffff0000u) > 0) ? 0xffff0000u : 0u;
    return v & (lo | hi);
}

__device__ __forceinline__ float sum_words(const Frag& f)
{
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += __uint_as_float(f.w[j] << 16) + __uint_as_float(f.w[j] & 0xffff0000u);
    return s;
}

// One block = one 16 x (16*NJ) output tile; its 4 waves take the 32-wide contraction steps round-robin
// (step s belongs to wave s % 4, so at any time the block reads 4 adjacent 64-B pieces of a row) and
// U = 4 / NJ of a wave's steps are loaded together before their MFMAs: a 256-deep contraction is ONE
// memory latency per wave, a 2048-deep one four.  The partial tiles are summed through LDS by wave 0.
template <int NJ, bool AC, bool BC, bool GATE, bool MASK>
__device__ __forceinline__ void sg_tile(const SG& p, const int block, float4 (*red)[NJ][64], float (*red_rs)[64])
{
    constexpr int U = 4 / NJ;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int it = block % p.n_it, jt = block / p.n_it;
    const int li = lane & 15, g = lane >> 4;
    const int i = it * 16 + li;
    const int ic = min(i, p.I - 1);
    // row-contiguous A with blocked rows: row i lives in block i / a_blk (the packed q | k | v gradient of the
    // self-attention in-projection, three [R, 256] buffers a_bs apart)
    const int64_t arow_off = (!AC && p.a_blk) ? (int64_t)(ic / p.a_blk) * p.a_bs + (ic % p.a_blk) : (int64_t)ic * p.a_rs;
    const u16* __restrict__ arow = p.a + arow_off;
    const u16* __restrict__ grow = GATE ? p.gate + arow_off : nullptr;
    const u16* __restrict__ brow[NJ];
    const int j0 = jt * 16 * NJ;
#pragma unroll
    for (int n = 0; n < NJ; ++n) brow[n] = p.b + (int64_t)min(j0 + 16 * n + li, p.J - 1) * p.b_rs;
    f32x4 acc[NJ];
#pragma unroll
    for (int n = 0; n < NJ; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rs = 0.f;
    const int nsteps = (p.Kc + 31) >> 5;
    for (int s0 = wv; s0 < nsteps; s0 += 4 * U) {
        Frag fa[U], fg[U], fb[U][NJ];
#pragma unroll
        for (int u = 0; u < U; ++u) {       // steps past the end re-read the last one (discarded below)
            const int k = min(s0 + 4 * u, nsteps - 1) * 32 + 8 * g;
#pragma unroll
            for (int n = 0; n < NJ; ++n) fb[u][n] = load_frag<BC, MASK>(brow[n], p.b_ks, k, p.Kc);
            // contraction-contiguous A with a blocked contraction index (blocks of a_blk, a_bs apart)
            const int ka = (AC && p.a_blk) ? (int)((k / p.a_blk) * p.a_bs) + (k % p.a_blk) : k;
            fa[u] = load_frag<AC, MASK>(arow, p.a_ks, ka, (AC && p.a_blk) ? 0x7fffffff : p.Kc);
            if (GATE) fg[u] = load_frag<AC, MASK>(grow, p.a_ks, ka, (AC && p.a_blk) ? 0x7fffffff : p.Kc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (GATE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) fa[u].w[j] = gate_word(fa[u].w[j], fg[u].w[j]);
            }
            if (u > 0 && s0 + 4 * u >= nsteps) fa[u].q = make_uint4(0u, 0u, 0u, 0u);
            rs += sum_words(fa[u]);
#pragma unroll
            for (int n = 0; n < NJ; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u][n].v, fa[u].v, acc[n], 0, 0, 0);
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int n = 0; n < NJ; ++n) red[wv - 1][n][lane] = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
        red_rs[wv - 1][lane] = rs;
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
#pragma unroll
        for (int n = 0; n < NJ; ++n) {
            const float4 t = red[o][n][lane];
            acc[n][0] += t.x; acc[n][1] += t.y; acc[n][2] += t.z; acc[n][3] += t.w;
        }
        rs += red_rs[o][lane];
    }
    if (p.rowsum && jt == 0) {
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        if (g == 0 && i < p.I) p.rowsum[i] = (u16)f2bf(rs);
    }
    if (i >= p.I) return;
#pragma unroll
    for (int n = 0; n < NJ; ++n) {
        const int j = j0 + 16 * n + 4 * g;
        if (j >= p.J) continue;
        float v[4] = {acc[n][0], acc[n][1], acc[n][2], acc[n][3]};
        if (p.bias) {
            const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + j);
            v[0] += __uint_as_float(bb.x << 16);
            v[1] += __uint_as_float(bb.x & 0xffff0000u);
            v[2] += __uint_as_float(bb.y << 16);
            v[3] += __uint_as_float(bb.y & 0xffff0000u);
        }
        if (p.cin) {
            const uint2 cc = *reinterpret_cast<const uint2*>(p.cin + (int64_t)i * p.ldcin + j);
            v[0] += __uint_as_float(cc.x << 16);
            v[1] += __uint_as_float(cc.x & 0xffff0000u);
            v[2] += __uint_as_float(cc.y << 16);
            v[3] += __uint_as_float(cc.y & 0xffff0000u);
        }
        if (p.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        uint2 o;
        o.x = f2bf(v[0]) | (f2bf(v[1]) << 16);
        o.y = f2bf(v[2]) | (f2bf(v[3]) << 16);
        const int64_t co = p.c_blk ? (int64_t)(j / p.c_blk) * p.c_bs + (j % p.c_blk) : j;      // column-blocked output
        *reinterpret_cast<uint2*>(p.c + (int64_t)i * p.ldc + co) = o;
    }
}

template <int NJ, bool AC, bool BC, bool GATE, bool MASK>
__global__ __launch_bounds__(256) void small_gemm_kernel(const SG p)
{
    __shared__ float4 red[3][NJ][64];
    __shared__ float red_rs[3][64];
    sg_tile<NJ, AC, BC, GATE, MASK>(p, (int)blockIdx.x, red, red_rs);
}

// Several independent weight-gradient problems (both operands row-contiguous: dY and x read along their rows) in ONE
// launch: the six dW GEMMs of a decoder layer's backward are 6-9 us each as launches of 16-128 blocks, mostly latency.
// A block finds its problem from the running block counts; the tile width and the gate are per problem.
constexpr int kGroupMax = 8;
struct SGGroup {
    SG it[kGroupMax];
    int first[kGroupMax + 1];        // first block of problem g (first[n] = grid size)
    int nj[kGroupMax];
    int n;
};

// CT: every problem's operands are contraction-contiguous (transposed copies made by mpf_transpose_group_bf16: 16-byte fragment
// loads instead of eight 2-byte ones — the row-contiguous form of a decoder layer's six weight gradients was 30 us of mostly
// vector-memory instruction issue)
template <bool MASK, bool CT>
__global__ __launch_bounds__(256) void small_gemm_group_kernel(const SGGroup g)
{
    __shared__ float4 red[3 * 4 * 64];
    __shared__ float red_rs[3][64];
    int k = 0;
#pragma unroll
    for (int t = 1; t < kGroupMax; ++t)
        if (t < g.n && (int)blockIdx.x >= g.first[t]) k = t;
    const SG& p = g.it[k];
    const int block = (int)blockIdx.x - g.first[k];
    const bool gate = p.gate != nullptr;
    const int nj = g.nj[k];
#define SG_CASE(NJ_, G_) sg_tile<NJ_, CT, CT, G_, MASK>(p, block, reinterpret_cast<float4(*)[NJ_][64]>(red), red_rs)
    if (nj == 4) { if (gate) SG_CASE(4, true); else SG_CASE(4, false); }
    else if (nj == 2) { if (gate) SG_CASE(2, true); else SG_CASE(2, false); }
    else { if (gate) SG_CASE(1, true); else SG_CASE(1, false); }
#undef SG_CASE
}

// Transposed, zero-padded copies of several bf16 matrices in one launch: src [R, C] (row stride ld) -> dst [C, Rp] (Rp >= R, a
// multiple of 8; rows R .. Rp - 1 of the source read as zero), optionally gated: elements whose `gate` element (same addressing)
// is <= 0 become zero (the ReLU backward of the FFN's hidden gradient).  64 x 64 tiles through LDS, 16-byte loads and stores.
constexpr int kTrMax = 16;
struct TrItem {
    const u16* src;
    const u16* gate;
    u16* dst;
    int64_t ld;
    int R, C, first;             // first block of this item
};
struct TrGroup {
    TrItem it[kTrMax];
    int n, Rp, tiles_r;
};

__global__ __launch_bounds__(256) void transpose_group_kernel(const TrGroup g)
{
    __shared__ __attribute__((aligned(16))) u16 tl[64][72];       // [column][row], 144-byte rows
    int k = 0;
#pragma unroll
    for (int t = 1; t < kTrMax; ++t)
        if (t < g.n && (int)blockIdx.x >= g.it[t].first) k = t;
    const TrItem& p = g.it[k];
    const int b = (int)blockIdx.x - p.first;
    const int tr = b % g.tiles_r, tc = b / g.tiles_r;
    const int r0 = tr * 64, c0 = tc * 64;
#pragma unroll
    for (int itx = 0; itx < 2; ++itx) {
        const int u = threadIdx.x + itx * 256;
        const int r = u >> 3, cg = u & 7;
        const int row = r0 + r, col = c0 + cg * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u), gt = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        if (row < p.R && col < p.C) {
            v = *reinterpret_cast<const uint4*>(p.src + (int64_t)row * p.ld + col);
            if (p.gate) gt = *reinterpret_cast<const uint4*>(p.gate + (int64_t)row * p.ld + col);
        }
        const unsigned w[4] = {gate_word(v.x, gt.x), gate_word(v.y, gt.y), gate_word(v.z, gt.z), gate_word(v.w, gt.w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tl[cg * 8 + 2 * q][r] = (u16)(w[q] & 0xffffu);
            tl[cg * 8 + 2 * q + 1][r] = (u16)(w[q] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int itx = 0; itx < 2; ++itx) {
        const int u = threadIdx.x + itx * 256;
        const int cc = u >> 3, lg = u & 7;
        const int col = c0 + cc, row = r0 + lg * 8;
        if (col < p.C && row < g.Rp)
            *reinterpret_cast<uint4*>(p.dst + (int64_t)col * g.Rp + row) = *reinterpret_cast<const uint4*>(&tl[cc][lg * 8]);
    }
}

template <int NJ, bool AC, bool BC>
void launch2(const SG& p, bool gate, bool mask, int blocks, hipStream_t st)
{
    if (gate && mask)
        small_gemm_kernel<NJ, AC, BC, true, true><<<blocks, 256, 0, st>>>(p);
    else if (gate)
        small_gemm_kernel<NJ, AC, BC, true, false><<<blocks, 256, 0, st>>>(p);
    else if (mask)
        small_gemm_kernel<NJ, AC, BC, false, true><<<blocks, 256, 0, st>>>(p);
    else
        small_gemm_kernel<NJ, AC, BC, false, false><<<blocks, 256, 0, st>>>(p);
}

template <int NJ>
void launch(const SG& p, bool ac, bool bc, int blocks, hipStream_t st)
{
    const bool gate = p.gate != nullptr, mask = (p.Kc & 31) != 0;
    if (ac && bc)
        launch2<NJ, true, true>(p, gate, mask, blocks, st);
    else if (ac)
        launch2<NJ, true, false>(p, gate, mask, blocks, st);
    else if (bc)
        launch2<NJ, false, true>(p, gate, mask, blocks, st);
    else
        launch2<NJ, false, false>(p, gate, mask, blocks, st);
}

}  // namespace

namespace {

// validates one problem and fills its descriptor; *nj_out = the tile width chosen for it.  Returns 1 for an empty problem.
int sg_fill(SG& p, int* nj_out, const void* a, int64_t a_rs, int64_t a_ks, int a_blk, int64_t a_bs, const void* gate, const void* b,
            int64_t b_rs, int64_t b_ks, const void* bias, const void* c_in, int64_t ldcin, void* c, int64_t ldc, int c_blk,
            int64_t c_bs, void* rowsum_a, int I, int J, int Kc, int relu, int* err)
{
    *err = 0;
#define SG_FAIL(code, msg) { *err = mpf::fail(code, msg); return 0; }
    if (a_blk < 0 || c_blk < 0 || (a_blk && a_blk % 32) || (c_blk && c_blk % 64) || (a_blk && a_ks == 1 && Kc % 32) ||
        (c_blk && c_in))
        SG_FAIL(MPF_E_SHAPE, "small_gemm_blocked: a_blk % 32, c_blk % 64 (and Kc % 32 for a blocked contraction) must be 0");
    if (I < 0 || J < 0 || Kc < 0) SG_FAIL(MPF_E_SHAPE, "small_gemm: negative size");
    if (I == 0 || J == 0) return 1;
    if (!a || !b || !c) SG_FAIL(MPF_E_NULL, "small_gemm: a, b, c must not be null");
    if (Kc == 0) SG_FAIL(MPF_E_SHAPE, "small_gemm: empty contraction");
    const bool ac = a_ks == 1, bc = b_ks == 1;
    if ((!ac && a_rs != 1) || (!bc && b_rs != 1))
        SG_FAIL(MPF_E_SHAPE, "small_gemm: each operand needs a unit row or contraction stride");
    if ((ac && (Kc % 8 || a_rs % 8 || ((uintptr_t)a & 15) || (gate && ((uintptr_t)gate & 15)))) ||
        (bc && (Kc % 8 || b_rs % 8 || ((uintptr_t)b & 15))))
        SG_FAIL(MPF_E_SHAPE, "small_gemm: contraction-contiguous operands need 16-B aligned rows, Kc % 8 == 0");
    if (J % 4 || ldc % 4 || ((uintptr_t)c & 7) || (bias && ((uintptr_t)bias & 7)) ||
        (c_in && (ldcin % 4 || ((uintptr_t)c_in & 7))))
        SG_FAIL(MPF_E_SHAPE, "small_gemm: J and ldc must be multiples of 4 (8-B stores)");
#undef SG_FAIL
    p.a = static_cast<const u16*>(a);
    p.gate = static_cast<const u16*>(gate);
    p.b = static_cast<const u16*>(b);
    p.bias = static_cast<const u16*>(bias);
    p.cin = static_cast<const u16*>(c_in);
    p.ldcin = ldcin;
    p.c = static_cast<u16*>(c);
    p.rowsum = static_cast<u16*>(rowsum_a);
    p.a_rs = a_rs; p.a_ks = a_ks; p.b_rs = b_rs; p.b_ks = b_ks; p.ldc = ldc;
    p.a_blk = a_blk; p.a_bs = a_bs; p.c_blk = c_blk; p.c_bs = c_bs;
    p.I = I; p.J = J; p.Kc = Kc; p.relu = relu;
    p.n_it = (I + 15) / 16;
    // widest tile that still gives the chip >= 256 blocks
    int nj = 4;
    while (nj > 1 && (int64_t)p.n_it * ((J + 16 * nj - 1) / (16 * nj)) < 256) nj >>= 1;
    p.n_waves = p.n_it * ((J + 16 * nj - 1) / (16 * nj));
    *nj_out = nj;
    return 0;
}

}  // namespace

extern "C" int mpf_small_gemm_bf16_blocked(const void* a, int64_t a_rs, int64_t a_ks, int a_blk, int64_t a_bs, const void* gate,
                                           const void* b, int64_t b_rs, int64_t b_ks, const void* bias, const void* c_in,
                                           int64_t ldcin, void* c, int64_t ldc, int c_blk, int64_t c_bs, void* rowsum_a, int I,
                                           int J, int Kc, int relu, void* stream)
{
    SG p;
    int nj = 1, err = 0;
    const int empty = sg_fill(p, &nj, a, a_rs, a_ks, a_blk, a_bs, gate, b, b_rs, b_ks, bias, c_in, ldcin, c, ldc, c_blk, c_bs, rowsum_a,
                              I, J, Kc, relu, &err);
    if (err) return err;
    if (empty) return 0;
    const bool ac = a_ks == 1, bc = b_ks == 1;
    const int blocks = p.n_waves;
    hipStream_t st = static_cast<hipStream_t>(stream);
    mpf::prof_begin(st);
    if (nj == 4) launch<4>(p, ac, bc, blocks, st);
    else if (nj == 2) launch<2>(p, ac, bc, blocks, st);
    else launch<1>(p, ac, bc, blocks, st);
    mpf::set_kernel("small_gemm_kernel");
    const double bytes = 2.0 * ((double)I * Kc * (gate ? 2 : 1) + (double)J * Kc + (double)I * J);
    mpf::prof_end("small_gemm_kernel", st, bytes);
    return mpf::check(hipGetLastError(), "small_gemm launch");
}

extern "C" int mpf_small_gemm_bf16(const void* a, int64_t a_rs, int64_t a_ks, const void* gate, const void* b,
                                   int64_t b_rs, int64_t b_ks, const void* bias, const void* c_in, int64_t ldcin, void* c,
                                   int64_t ldc, void* rowsum_a, int I, int J, int Kc, int relu, void* stream)
{
    return mpf_small_gemm_bf16_blocked(a, a_rs, a_ks, 0, 0, gate, b, b_rs, b_ks, bias, c_in, ldcin, c, ldc, 0, 0, rowsum_a, I, J, Kc,
                                       relu, stream);
}

extern "C" int mpf_small_gemm_bf16_group(const MpfSmallGemmItem* items, int n_items, void* stream)
{
    if (n_items < 0 || n_items > kGroupMax) return mpf::fail(MPF_E_SHAPE, "small_gemm_group: at most 8 problems per launch");
    if (n_items == 0) return 0;
    if (!items) return mpf::fail(MPF_E_NULL, "small_gemm_group: NULL items");
    SGGroup g;
    g.n = 0;
    g.first[0] = 0;
    bool mask = false, contig = false;
    double bytes = 0.0;
    for (int t = 0; t < n_items; ++t) {
        const MpfSmallGemmItem& m = items[t];
        const bool ct_item = m.a_ks == 1 && m.b_ks == 1;
        if (!ct_item && (m.a_rs != 1 || m.b_rs != 1))
            return mpf::fail(MPF_E_SHAPE, "small_gemm_group: both operands row-contiguous (the weight-gradient form) or both contraction-contiguous");
        if (t > 0 && ct_item != contig) return mpf::fail(MPF_E_SHAPE, "small_gemm_group: one operand form per group");
        contig = ct_item;
        if (ct_item && m.gate) return mpf::fail(MPF_E_SHAPE, "small_gemm_group: the gate belongs into the transposed copy (mpf_transpose_group_bf16)");
        int nj = 1, err = 0;
        const int empty = sg_fill(g.it[g.n], &nj, m.a, m.a_rs, m.a_ks, m.a_blk, m.a_bs, m.gate, m.b, m.b_rs, m.b_ks, nullptr, nullptr, 0,
                                  m.c, m.ldc, 0, 0, m.rowsum_a, m.I, m.J, m.Kc, 0, &err);
        if (err) return err;
        if (empty) continue;
        if (g.n > 0 && ((m.Kc & 31) != 0) != mask) return mpf::fail(MPF_E_SHAPE, "small_gemm_group: Kc % 32 must agree across the group");
        mask = (m.Kc & 31) != 0;
        g.nj[g.n] = nj;
        g.first[g.n + 1] = g.first[g.n] + g.it[g.n].n_waves;
        bytes += 2.0 * ((double)m.I * m.Kc * (m.gate ? 2 : 1) + (double)m.J * m.Kc + (double)m.I * m.J);
        ++g.n;
    }
    if (g.n == 0) return 0;
    for (int t = g.n; t < kGroupMax; ++t) { g.first[t + 1] = g.first[g.n]; g.nj[t] = 1; g.it[t] = g.it[0]; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    mpf::prof_begin(st);
    if (contig) {
        if (mask) small_gemm_group_kernel<true, true><<<g.first[g.n], 256, 0, st>>>(g);
        else small_gemm_group_kernel<false, true><<<g.first[g.n], 256, 0, st>>>(g);
    } else {
        if (mask) small_gemm_group_kernel<true, false><<<g.first[g.n], 256, 0, st>>>(g);
        else small_gemm_group_kernel<false, false><<<g.first[g.n], 256, 0, st>>>(g);
    }
    mpf::set_kernel("small_gemm_group_kernel");
    mpf::prof_end("small_gemm_group_kernel", st, bytes);
    return mpf::check(hipGetLastError(), "small_gemm_group launch");
}

extern "C" int mpf_transpose_group_bf16(const MpfTransposeItem* items, int n_items, int Rp, void* stream)
{
    if (n_items < 0 || n_items > kTrMax) return mpf::fail(MPF_E_SHAPE, "transpose_group: at most 16 matrices per launch");
    if (n_items == 0) return 0;
    if (!items) return mpf::fail(MPF_E_NULL, "transpose_group: NULL items");
    if (Rp <= 0 || Rp % 8) return mpf::fail(MPF_E_SHAPE, "transpose_group: Rp must be a positive multiple of 8");
    TrGroup g;
    g.n = n_items; g.Rp = Rp; g.tiles_r = (Rp + 63) / 64;
    int blocks = 0;
    for (int t = 0; t < n_items; ++t) {
        const MpfTransposeItem& m = items[t];
        if (!m.src || !m.dst) return mpf::fail(MPF_E_NULL, "transpose_group: NULL matrix");
        if (m.R <= 0 || m.R > Rp || m.C <= 0 || m.C % 8 || m.ld % 8 || ((uintptr_t)m.src & 15) || ((uintptr_t)m.dst & 15) ||
            (m.gate && ((uintptr_t)m.gate & 15)))
            return mpf::fail(MPF_E_SHAPE, "transpose_group: 0 < R <= Rp, C % 8 == 0, 16-byte aligned rows");
        g.it[t].src = static_cast<const u16*>(m.src); g.it[t].gate = static_cast<const u16*>(m.gate); g.it[t].dst = static_cast<u16*>(m.dst);
        g.it[t].ld = m.ld; g.it[t].R = m.R; g.it[t].C = m.C; g.it[t].first = blocks;
        blocks += g.tiles_r * ((m.C + 63) / 64);
    }
    for (int t = n_items; t < kTrMax; ++t) { g.it[t] = g.it[0]; g.it[t].first = blocks; }
    mpf::set_kernel("transpose_group_kernel");
    transpose_group_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(g);
    return mpf::check(hipGetLastError(), "transpose_group launch");
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 Linear with MANY rows: the key / value in-projections of the cross-attention (nn.MultiheadAttention in_proj of
// mask2former_transformer_decoder.py:100-112; rows = S_l * N = 2 048 .. 32 768 per level, the three layers of a level side
// by side: 768 outputs) and their input gradients, the batched prediction heads (:1859-1870, ~2 400 rows).  Round 4: these
// were the last library GEMMs of the head (hipBLASLt Cijk_* / CK batched_gemm_xdl: 0.3 ms/step in 21 launches).
//   C[m][n] = sum_k A[m][k] * B[n][k] (+ bias[n]),   A [M, K] bf16 (row stride lda), B [N, K] bf16 (row stride ldb),
//   C [M, N] bf16 (row stride ldc), fp32 accumulation, one rounding of the result — the library's rounding points.
// Both operands are contraction-contiguous, i.e. ARE v_mfma_f32_16x16x32_bf16 fragments as they lie in memory (a lane's 16
// bytes = 8 consecutive k of one row): no LDS, no staging — the weights (<= 400 KB) stay in L2, an A row block is read by
// the two waves that share it.  The product is formed transposed (D = B-tile x A-tile^T) so that a lane ends up with FOUR
// CONSECUTIVE output columns of one row: 8-byte stores.  Workgroup = 4 waves = 128 rows x 128 columns, wave = 64 x 64
// (4 x 4 MFMA tiles), fragments of K step k + 1 requested before the 16 MFMAs of step k.
// ------------------------------------------------------------------------------------------------------------------
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 tg_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 tg_bf16x4;
typedef float tg_f32x4 __attribute__((ext_vector_type(4)));

template <bool BIAS>
__global__ __launch_bounds__(256) void tall_gemm_bf16_kernel(const __bf16* __restrict__ A, int64_t lda, const __bf16* __restrict__ B,
                                                             int64_t ldb, const __bf16* __restrict__ bias, __bf16* __restrict__ C,
                                                             int64_t ldc, int M, int N, int K)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.y * 128 + (wave >> 1) * 64, col0 = blockIdx.x * 128 + (wave & 1) * 64;
    if (row0 >= M || col0 >= N) return;
    const int li = lane & 15, kb = (lane >> 4) * 8;
    // clamped rows / columns: out-of-range tiles compute on the last row / column and are not stored
    const __bf16* ap[4];
    const __bf16* bp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ap[t] = A + (int64_t)min(row0 + t * 16 + li, M - 1) * lda + kb;
        bp[t] = B + (int64_t)min(col0 + t * 16 + li, N - 1) * ldb + kb;
    }
    tg_f32x4 acc[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = tg_f32x4{0.f, 0.f, 0.f, 0.f};
    tg_bf16x8 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { a0[t] = *reinterpret_cast<const tg_bf16x8*>(ap[t]); b0[t] = *reinterpret_cast<const tg_bf16x8*>(bp[t]); }
    const int nk = K >> 5;
    for (int ks = 0; ks < nk; ks += 2) {
        const int k1 = min(ks + 1, nk - 1) * 32, k2 = min(ks + 2, nk - 1) * 32;
#pragma unroll
        for (int t = 0; t < 4; ++t) { a1[t] = *reinterpret_cast<const tg_bf16x8*>(ap[t] + k1); b1[t] = *reinterpret_cast<const tg_bf16x8*>(bp[t] + k1); }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[n], a0[m], acc[n][m], 0, 0, 0);
        if (ks + 1 < nk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { a0[t] = *reinterpret_cast<const tg_bf16x8*>(ap[t] + k2); b0[t] = *reinterpret_cast<const tg_bf16x8*>(bp[t] + k2); }
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[n], a1[m], acc[n][m], 0, 0, 0);
        }
    }
    // D[n-tile][m-tile]: lane holds output columns col0 + n * 16 + 4 * (lane >> 4) + r (r = 0..3) of row row0 + m * 16 + (lane & 15)
    const int cq = (lane >> 4) * 4;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int col = col0 + n * 16 + cq;
        float bz[4] = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bz[r] = (float)bias[min(col + r, N - 1)];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = row0 + m * 16 + li;
            if (row >= M || col >= N) continue;
            __bf16* dst = C + (int64_t)row * ldc + col;
            const tg_bf16x4 v = {(__bf16)(acc[n][m][0] + bz[0]), (__bf16)(acc[n][m][1] + bz[1]), (__bf16)(acc[n][m][2] + bz[2]),
                                 (__bf16)(acc[n][m][3] + bz[3])};
            if (col + 3 < N) {
                *reinterpret_cast<tg_bf16x4*>(dst) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (col + r < N) dst[r] = v[r];
            }
        }
    }
}

// Weight-stationary form for K = 256 / 768 (the key / value projections of the decoder and their input gradients): what bounded
// the kernel above was L1 <- L2 traffic of fragments that every wave fetched for itself (A six times per column tile and twice
// per workgroup, W once per row tile: ~400 MB for a 67 MB problem).  Here a wave keeps the W fragments of its 16 NB columns over
// the WHOLE contraction in registers (16 NB KS dwords) for the lifetime of the workgroup, which walks row tiles; a row tile of X
// (16 MB rows x K) comes global -> LDS by DMA once per workgroup in full 128-byte lines and is read as ds_read_b128 fragments by
// the four waves.  LDS image = rows of K elements, 16-byte slot XOR-swizzled by (row & 15) on the SOURCE side of the DMA (the
// destination of a DMA piece is lane-linear): the 16-lane service groups of ds_read_b128 then touch 16 different slots.  One LDS
// buffer: the copy of tile t + 1 is requested after the barrier that ends the reads of tile t and lands under tile t's epilogue
// stores; several workgroups per CU cover the rest.
// one DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_addr, lds_addr + 1024).  Inline asm, not
// __builtin_amdgcn_global_load_lds: hipcc puts s_waitcnt vmcnt(0) in front of the next LDS access after the builtin, which
// would wait for the tile that has just been requested (M0 saved / restored: see gemm3.hip glds16).
__device__ __forceinline__ void tall_glds16(const void* gaddr, unsigned lds_addr)
{
    unsigned saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved) : "v"(gaddr), "s"(lds_addr) : "memory");
}

template <int KS, int NB, int MB, int NBUF, bool BIAS>
__global__ __launch_bounds__(256) void tall_ws_bf16_kernel(const __bf16* __restrict__ A, int64_t lda, const __bf16* __restrict__ B,
                                                           int64_t ldb, const __bf16* __restrict__ bias, __bf16* __restrict__ C,
                                                           int64_t ldc, int M, int N, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];
    constexpr int CPR = KS * 4;                   // 16-byte chunks per row
    constexpr int ROWS = MB * 16;
    constexpr int PIECES = ROWS * CPR / 64;       // DMA pieces (1 KB) per tile
    static_assert(CPR % 16 == 0 && PIECES % 4 == 0, "tile shape");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, cg = lane >> 4;
    const int col0 = blockIdx.x * (64 * NB) + wave * (16 * NB);
    // W fragments of this wave's columns: the A operand of D = W . X^T (clamped rows: columns past N are computed and not stored)
    tg_bf16x8 wf[NB][KS];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const __bf16* wr = B + (int64_t)min(col0 + n * 16 + li, N - 1) * ldb + cg * 8;
#pragma unroll
        for (int j = 0; j < KS; ++j) wf[n][j] = *reinterpret_cast<const tg_bf16x8*>(wr + j * 32);
    }
    float bz[NB][4];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) bz[n][r] = BIAS ? (float)bias[min(col0 + n * 16 + cg * 4 + r, N - 1)] : 0.f;
    constexpr int TILE_BYTES = ROWS * CPR * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)ws_lds;          // (LDS byte address of the dynamic segment)
    auto request = [&](int t, int buf) {
        const int row0 = t * ROWS;
#pragma unroll
        for (int i = 0; i < PIECES / 4; ++i) {
            const int piece = wave * (PIECES / 4) + i;
            const int slot = piece * 64 + lane, row = slot / CPR, pos = slot - row * CPR;
            const __bf16* src = A + (int64_t)min(row0 + row, M - 1) * lda + (pos ^ (row & 15)) * 8;
            tall_glds16(src, lds0 + buf * TILE_BYTES + piece * 1024);
        }
    };
    // NBUF buffers, tiles requested D = NBUF - 1 ahead, ONE barrier per tile.  At the top of tile t the wave waits for
    // everything it has in flight — its pieces of tile t AND the result stores of the tile before.  (A count that lets those
    // stores stay in flight — vmcnt(#stores) — is NOT safe: the counter is shared by loads and stores, which complete in order
    // only among themselves, so a satisfied count does not prove that the older DMA pieces have landed.  That form shipped for a
    // few hours in round 4 and showed as a rare 0.3 % deviation of a whole forward pass in one of three full test runs.)
    // The barrier then says every wave's pieces have landed AND every wave is done with tile t - 1, whose buffer receives
    // tile t + D.
    constexpr int D = NBUF - 1;
    const int g = gridDim.y;
    int t = blockIdx.y, it = 0;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (t + d * g < ntiles) request(t + d * g, d);
    for (; t < ntiles; t += g, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + D * g < ntiles) request(t + D * g, (it + D) % NBUF);
        const unsigned char* tile = ws_lds + (it % NBUF) * TILE_BYTES;
        tg_f32x4 acc[NB][MB];
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[n][m] = tg_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            tg_bf16x8 xf[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int row = m * 16 + li;
                xf[m] = *reinterpret_cast<const tg_bf16x8*>(tile + (row * CPR + ((4 * j + cg) ^ li)) * 16);
            }
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][j], xf[m], acc[n][m], 0, 0, 0);
        }
        // D[n][m]: lane holds columns col0 + 16 n + 4 cg + r (r = 0..3) of row t ROWS + 16 m + li
        const int row0 = t * ROWS;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int col = col0 + n * 16 + cg * 4;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int row = row0 + m * 16 + li;
                if (row >= M || col >= N) continue;
                __bf16* dst = C + (int64_t)row * ldc + col;
                const tg_bf16x4 v = {(__bf16)(acc[n][m][0] + bz[n][0]), (__bf16)(acc[n][m][1] + bz[n][1]),
                                     (__bf16)(acc[n][m][2] + bz[n][2]), (__bf16)(acc[n][m][3] + bz[n][3])};
                if (col + 3 < N) {
                    *reinterpret_cast<tg_bf16x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (col + r < N) dst[r] = v[r];
                }
            }
        }
    }
}

int g_tall_ws_wgs = 0;      // mpf_set_option("tall_ws_wgs"): workgroups of the weight-stationary kernel (0 = one residency)

template <int KS, int NB, int MB, int NBUF>
int launch_tall_ws(const void* a, int64_t lda, const void* b, int64_t ldb, const void* bias, void* c, int64_t ldc, int M, int N,
                   hipStream_t st)
{
    const int ntiles = (M + MB * 16 - 1) / (MB * 16);
    const unsigned gx = (unsigned)((N + 64 * NB - 1) / (64 * NB));
    const size_t lds = (size_t)MB * 16 * KS * 64 * NBUF;
    // the grid is exactly one residency (LDS: 160 KB per CU; registers: two workgroups), so that no workgroup starts after the
    // others have walked their tiles
    const unsigned per_cu = lds > 80 * 1024 ? 1u : 2u;
    unsigned gy = (unsigned)(g_tall_ws_wgs ? g_tall_ws_wgs : mpf::cu_count() * (int)per_cu) / gx;
    gy = gy < 1u ? 1u : (gy > (unsigned)ntiles ? (unsigned)ntiles : gy);
    {
        const void* fn = bias ? (const void*)tall_ws_bf16_kernel<KS, NB, MB, NBUF, true> : (const void*)tall_ws_bf16_kernel<KS, NB, MB, NBUF, false>;
        static mpf::LdsAttr attr[2];       // (per instantiation of this launcher: with / without bias)
        if (int e = mpf::ensure_dynamic_lds(fn, lds, attr[bias ? 1 : 0])) return e;
    }
    if (bias)
        hipLaunchKernelGGL((tall_ws_bf16_kernel<KS, NB, MB, NBUF, true>), dim3(gx, gy), dim3(256), lds, st, (const __bf16*)a, lda,
                           (const __bf16*)b, ldb, (const __bf16*)bias, (__bf16*)c, ldc, M, N, ntiles);
    else
        hipLaunchKernelGGL((tall_ws_bf16_kernel<KS, NB, MB, NBUF, false>), dim3(gx, gy), dim3(256), lds, st, (const __bf16*)a, lda,
                           (const __bf16*)b, ldb, (const __bf16*)bias, (__bf16*)c, ldc, M, N, ntiles);
    return 0;
}

int g_tall_ws = 1;          // mpf_set_option("tall_ws", 0): the fragment-per-wave kernel for every shape (A/B)
}  // namespace

namespace mpf {
int set_small_gemm_option(const char* key, int v)
{
    if (!strcmp(key, "tall_ws")) { g_tall_ws = v != 0; return 0; }
    if (!strcmp(key, "tall_ws_wgs")) { g_tall_ws_wgs = v; return 0; }

    return 1;
}
}  // namespace mpf

extern "C" int mpf_tall_gemm_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const void* bias, void* c, int64_t ldc, int M,
                                  int N, int K, void* stream)
{
    if (M == 0 || N == 0) return 0;
    if (!a || !b || !c) return mpf::fail(MPF_E_NULL, "tall_gemm_bf16: NULL buffer");
    if (M < 0 || N < 0 || K <= 0 || (K & 31)) return mpf::fail(MPF_E_SHAPE, "tall_gemm_bf16: K must be a positive multiple of 32");
    if ((lda & 7) || (ldb & 7) || (ldc & 3) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)c & 7))
        return mpf::fail(MPF_E_SHAPE, "tall_gemm_bf16: 16-byte aligned operand rows, 8-byte aligned result rows");
    const dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128));
    if (grid.y > 65535u) return mpf::fail(MPF_E_TOO_LARGE, "tall_gemm_bf16: more than 65 535 row tiles");
    hipStream_t st = (hipStream_t)stream;
    if (g_tall_ws && M >= 1024 && (K == 256 || K == 768)) {
        mpf::prof_begin(st);
        mpf::set_kernel("tall_ws_bf16_kernel");
        int e;
        // (measured at 32 768 rows: 64-row tiles x 2 buffers, 32-row tiles x 3 or 4 buffers all give 29-30 us for K = 256 and
        // 35-38 us for K = 768: the copies are not what the kernel waits for — ablations: result stores 9 us of HBM write time,
        // copies 3-8, MFMA phase 3 (K = 256) / 15 (K = 768: one 16-column block per wave reads the whole X tile from LDS))
        if (K == 256) e = launch_tall_ws<8, 2, 4, 2>(a, lda, b, ldb, bias, c, ldc, M, N, st);
        else e = launch_tall_ws<24, 1, 2, 2>(a, lda, b, ldb, bias, c, ldc, M, N, st);
        if (e) return e;
        mpf::prof_end("tall_gemm_bf16_kernel", st, 2.0 * ((double)M * K + (double)N * K + (double)M * N), 2.0 * M * (double)N * K);
        return mpf::check(hipGetLastError(), "mpf_tall_gemm_bf16");
    }
    mpf::prof_begin(st);
    mpf::set_kernel("tall_gemm_bf16_kernel");
    if (bias)
        hipLaunchKernelGGL((tall_gemm_bf16_kernel<true>), grid, dim3(256), 0, st, (const __bf16*)a, lda, (const __bf16*)b, ldb,
                           (const __bf16*)bias, (__bf16*)c, ldc, M, N, K);
    else
        hipLaunchKernelGGL((tall_gemm_bf16_kernel<false>), grid, dim3(256), 0, st, (const __bf16*)a, lda, (const __bf16*)b, ldb,
                           (const __bf16*)bias, (__bf16*)c, ldc, M, N, K);
    mpf::prof_end("tall_gemm_bf16_kernel", st, 2.0 * ((double)M * K + (double)N * K + (double)M * N), 2.0 * M * (double)N * K);
    return mpf::check(hipGetLastError(), "mpf_tall_gemm_bf16");
}
