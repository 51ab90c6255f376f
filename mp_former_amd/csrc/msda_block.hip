// Multi-scale deformable attention for MI355X, second generation: spatially blocked forward / backward.
// fp32, 32 channels per head, 4 points per level (the shapes every Mask2Former / MP-Former config uses);
// anything else takes the kernels of msda.hip.
//
// Reference semantics: mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh
// :242-304 (forward), :306-408 + :92-164 (backward).
//
// Why a second design.  The first-generation kernels gather every corner row of every sample from L2
// (48 rows of 128 B per (query, head): 12 288*e*S bytes against 800*e*S algorithmic) and the backward
// streams one 16-byte entry per (sample, pixel row) through HBM.  Both ran at L2-gather / issue rates:
// 0.16 and 0.044 of the HBM roofline (profiles/r01r_*).  Here:
//
//   forward:  a workgroup owns ONE head x a 2-D block of 8x8 spatially adjacent queries (queries of the pixel decoder ARE
//     the pixels of the levels).  Per level it decodes its 256 samples (one per thread), reduces their bounding box, and —
//     if the box fits the LDS budget, which it does whenever the offsets are a few pixels — stages the box's value rows
//     ONCE with direct global->LDS loads (1 KB per wave instruction) and samples from LDS (ds_read_b128, 8 lanes x 16 B
//     per row, 8 rows per wave instruction).  A level whose box does not fit (coarse queries looking into the finest map,
//     adversarial offsets) takes buffer-load gathers from L2 exactly like the first generation — a per-(workgroup, level)
//     uniform decision, no per-sample divergence.  Corners outside the image read a zero row (LDS) / an out-of-range
//     buffer offset (global): no branches, no masking.
//
//   backward ("bin" + "tile", round 4; the query-centric push + pull pair of rounds 2-3 was deleted in round 5 — HISTORY.md):
//     the scatter-add is re-stated as a tiny dense product per destination tile.  A tile is 4x4 pixels of one (image, head,
//     level).  For the samples s whose 2x2 footprint touches the tile,
//         grad_value[pixel, :] = sum_s  hat(px - x_s) * hat(py - y_s) * a_s  *  grad_out[q_s, :]
//     with hat(t) = max(0, 1 - |t|) — the bilinear weight of ANY pixel in closed form (zero outside the footprint, so no
//     corner bookkeeping and the image border needs no special case).  That is D[16 px x 32 ch] += A[16 px x 4 samples] *
//     B[4 samples x 32 ch] on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, cdna_hip_programming.md §3): accumulators stay
//     in registers — no LDS accumulators, no atomics, every grad_value element is written once.  The lists "samples per tile"
//     are 4-byte entries (query, point) appended by the bin kernel into fixed-capacity per-tile runs (one returning integer
//     add per (workgroup, tile)); run overflow goes to a spill list that a small atomic kernel applies afterwards (never
//     taken with pixel-decoder-like offsets; exercised by the tests).  See the section comment above msda_bwd_bin_kernel.
#include <hip/hip_runtime.h>
#include <limits.h>

#include <algorithm>
#include <stdint.h>

#include "amax.h"
#include "mpf_common.h"

namespace {

constexpr int kT = 256;
constexpr int kD = 32;
constexpr int kP = 4;
constexpr int kMaxL = 4;          // templated level counts 1..4
constexpr int kMaxBand = 16;      // horizontal bands of the pull kernel's workgroup order
constexpr int kSlots = 256;       // LDS hash slots of the push kernel (distinct destination tiles per workgroup)
constexpr unsigned kEmpty = 0xFFFFFFFFu;
constexpr int kOobOff = (int)0x80000000u;

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

struct GeomB {
    int L, M, Lq, S, N;
    int H[kMaxL], W[kMaxL], start[kMaxL];
    // query blocks: nql "query levels" (the value levels when the queries are the pixels, else one strip)
    int nql, bw_log2, bh;
    int qH[kMaxL], qW[kMaxL], qstart[kMaxL], qnbx[kMaxL], qblk_base[kMaxL];
    int blocks_per_b;
    // destination tiles (4 x 4 pixels) and their entry runs
    int ntx[kMaxL], tile_base[kMaxL], cap[kMaxL], ent_base[kMaxL];
    int tiles_per_bm, ent_per_bm;
    // pull launch geometry: waves per tile; workgroups (16 waves) per (image, head) ordered band-major, level-minor
    int wpt[kMaxL], nband, wg_per_bm;
    int band_wg_base[kMaxBand * kMaxL];
    // every band has the same number of tile rows of every level (level heights multiples of 4 * nband: the power-of-two maps
    // of the 1024^2 / 512 x 1024 configs): workgroup -> (band, level) is arithmetic, no table lookup in the kernel prologue
    int band_uniform, wg_per_band, lvl_wg_base[kMaxL];
    // geometry built on the DEVICE (mpf_msda_*_dev: the reference's all-device signature, no host copy of the shapes):
    //   ok          0 = the shapes cannot be served (sizes out of range, level ranges outside value or overlapping): every
    //               blocked kernel returns at once and msda_dev_guard_kernel fills the outputs with NaN
    //   contiguous  level_start_index is the running sum of H*W (else grad_value has rows no level owns: zero-filled first)
    //   nblk, nwg   workgroups of the query-block kernels / of the tile kernel (the launch is an upper bound; the kernels stride)
    int ok, contiguous, nblk, nwg;
};

// a[i] for a run-time i without a run-time kernarg offset: every element is read at its constant offset (hipcc batches
// those into one or two wide scalar loads) and selected; g.x[i] with a run-time i is one DEPENDENT scalar load per use —
// a dozen serialized round trips (~5000 cycles) in a kernel prologue
template <int N>
__device__ __forceinline__ int sel(const int (&a)[N], int i)
{
    int v = a[0];
#pragma unroll
    for (int k = 1; k < N; ++k) v = i == k ? a[k] : v;
    return v;
}

// pixel coordinate of a sampling location; explicitly rounded (no FMA contraction) so that every kernel
// derives the same footprint for a sample
__device__ __forceinline__ float pix(float loc, int size) { return __fsub_rn(__fmul_rn(loc, (float)size), 0.5f); }

__device__ __forceinline__ int xcd_index(int n)
{
    const int per_xcd = (n + 7) >> 3;
    return ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
}

// Which block / workgroup-slot does iteration `it` of this workgroup serve?  Host geometry (!DG): exactly one, the XCD-aware
// remap of blockIdx.x.  Device geometry: workgroup ids stride by the launch size through the same remap of the REAL count n.
template <bool DG>
__device__ __forceinline__ bool dg_next(int it, int n, int& blk)
{
    if (!DG) {
        if (it) return false;
        blk = xcd_index(n);
        return blk < n;
    }
    const int per_xcd = (n + 7) >> 3;
    const int64_t bid = (int64_t)blockIdx.x + (int64_t)it * gridDim.x;          // (gridDim.x is a multiple of 8: the XCD stays)
    if ((bid >> 3) >= per_xcd) return false;
    blk = (int)(bid & 7) * per_xcd + (int)(bid >> 3);
    return blk < n;            // (false = a tail slot of the last XCD's share: later ids of this workgroup are larger still)
}

// Inside the strided loop of a DG kernel the compiler hoists everything that does not depend on the iteration (lane / wave
// arithmetic, LDS addresses) out of the loop and keeps it live across the whole body: +13 VGPRs and scratch spills in the tile
// kernel.  A zero the compiler cannot see through, added to threadIdx.x per iteration, keeps those values where they are used.
template <bool DG>
__device__ __forceinline__ int dg_zero()
{
    if (!DG) return 0;
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

template <int CTRL, int RM>
__device__ __forceinline__ int dpp_keep(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, RM, 0xf, false); }

// wave-wide min / max through DPP (VALU only); result is wave-uniform
__device__ __forceinline__ int wave_min(int v)
{
    v = min(v, dpp_keep<0xB1, 0xf>(v));
    v = min(v, dpp_keep<0x4E, 0xf>(v));
    v = min(v, dpp_keep<0x141, 0xf>(v));
    v = min(v, dpp_keep<0x140, 0xf>(v));
    v = min(v, dpp_keep<0x142, 0xa>(v));
    v = min(v, dpp_keep<0x143, 0xc>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max(int v)
{
    v = max(v, dpp_keep<0xB1, 0xf>(v));
    v = max(v, dpp_keep<0x4E, 0xf>(v));
    v = max(v, dpp_keep<0x141, 0xf>(v));
    v = max(v, dpp_keep<0x140, 0xf>(v));
    v = max(v, dpp_keep<0x142, 0xa>(v));
    v = max(v, dpp_keep<0x143, 0xc>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

template <int CTRL>
__device__ __forceinline__ float dpp_addf(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(t);
}
// sum over the 8 consecutive lanes that share one sample; valid in every lane of the group
__device__ __forceinline__ float sum8(float v)
{
    v = dpp_addf<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_addf<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_addf<0x141>(v);     // row_half_mirror: lane i <-> 7 - i inside each group of 8
    return v;
}

struct Dec {
    int x0, y0;
    float lx, ly;
    bool in;
};

__device__ __forceinline__ Dec decode(float2 xy, int H, int W, bool valid)
{
    Dec d;
    const float x = pix(xy.x, W), y = pix(xy.y, H);
    d.in = valid && (y > -1.f && x > -1.f && y < (float)H && x < (float)W);
    const float xf = floorf(x), yf = floorf(y);
    d.x0 = (int)xf; d.y0 = (int)yf; d.lx = x - xf; d.ly = y - yf;
    if (!d.in) { d.x0 = 0; d.y0 = 0; d.lx = 0.f; d.ly = 0.f; }
    return d;
}

// LDS layout shared by the forward and the push kernel
//   [0, 4096)      float4 s_f[256]   per-sample floats (forward: 4 corner weights; push: lx, ly, a, -)
//   [4096, 8192)   int4   s_o[256]   per-sample corner byte offsets (LDS offsets or value-buffer offsets)
//   [8192, 8320)   int    s_bb[kMaxL*4] bounding boxes   (+ padding)
//   [8320, 8448)   128 B of zeros (the row every out-of-image corner reads)
//   [8448, ...)    region rows
constexpr int kOffBB = 8192, kOffZero = 8320, kOffReg = 8448;

struct BlockCtx {
    int b, m, ql, by, bx, qH, qW, qstart;
};

// stage `rows` (= rw * rh) value rows of the box (ymin.., xmin..) of level l into the region buffer with
// direct global->LDS loads: a wave instruction moves 8 rows (lane = (row, 16-byte piece)).
__device__ __forceinline__ void stage_region(const float* __restrict__ value, unsigned char* smem, const GeomB& g, int b, int m,
                                             int l, int xmin, int ymin, int rw, int rows, int tid)
{
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float inv_rw = 1.0f / (float)rw;
    const int nchunk = (rows + 31) >> 5;
    const int W = g.W[l];
    const float* base = value + ((int64_t)(b * g.S + g.start[l]) * g.M + m) * kD + (lane & 7) * 4;
    for (int c = 0; c < nchunk; ++c) {
        const int r = min(c * 32 + wave * 8 + (lane >> 3), rows - 1);
        const int ry = (int)(((float)r + 0.5f) * inv_rw);
        const int rx = r - ry * rw;
        const float* src = base + (int64_t)((ymin + ry) * W + (xmin + rx)) * (g.M * kD);
        unsigned char* dst = smem + kOffReg + (c * 32 + wave * 8) * 128;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

// corner byte offsets of a sample for the two gather paths
__device__ __forceinline__ int4 corner_offsets(const GeomB& g, const Dec& d, int l, bool lds_path, int xmin, int ymin, int rw,
                                               int b, int m)
{
    const int H = g.H[l], W = g.W[l];
    const bool y0v = d.in && d.y0 >= 0, y1v = d.in && d.y0 + 1 <= H - 1, x0v = d.in && d.x0 >= 0, x1v = d.in && d.x0 + 1 <= W - 1;
    int4 o;
    if (lds_path) {
        const int base = kOffReg + ((d.y0 - ymin) * rw + (d.x0 - xmin)) * 128;
        o.x = (y0v && x0v) ? base : kOffZero;
        o.y = (y0v && x1v) ? base + 128 : kOffZero;
        o.z = (y1v && x0v) ? base + rw * 128 : kOffZero;
        o.w = (y1v && x1v) ? base + rw * 128 + 128 : kOffZero;
    } else {
        const int sx = g.M * 128, sy = W * sx;
        const int base = ((b * g.S + g.start[l]) * g.M + m) * 128 + d.y0 * sy + d.x0 * sx;
        o.x = (y0v && x0v) ? base : kOobOff;
        o.y = (y0v && x1v) ? base + sx : kOobOff;
        o.z = (y1v && x0v) ? base + sy : kOobOff;
        o.w = (y1v && x1v) ? base + sy + sx : kOobOff;
    }
    return o;
}

__device__ __forceinline__ float4 lds_row(const unsigned char* smem, int off) { return *reinterpret_cast<const float4*>(smem + off); }
__device__ __forceinline__ float4 buf_row(__amdgpu_buffer_rsrc_t rs, int off)
{
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    return make_float4(__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3]));
}

__device__ __forceinline__ void block_of(const GeomB& g, int blk, BlockCtx& c)
{
    const int qb = blk % g.blocks_per_b, bm = blk / g.blocks_per_b;
    c.m = bm % g.M; c.b = bm / g.M;
    int ql = 0;
#pragma unroll
    for (int k = 1; k < kMaxL; ++k) if (k < g.nql && qb >= g.qblk_base[k]) ql = k;
    const int r = qb - sel(g.qblk_base, ql), nbx = sel(g.qnbx, ql);
    c.ql = ql; c.by = r / nbx; c.bx = r - c.by * nbx;
    c.qH = sel(g.qH, ql); c.qW = sel(g.qW, ql); c.qstart = sel(g.qstart, ql);
}

// query index of block-local query qi (or -1 when the block sticks out of its level)
__device__ __forceinline__ int query_of(const GeomB& g, const BlockCtx& c, int qi)
{
    const int qy = qi >> g.bw_log2, qx = qi & ((1 << g.bw_log2) - 1);
    const int gy = c.by * g.bh + qy, gx = (c.bx << g.bw_log2) + qx;
    return (gy < c.qH && gx < c.qW) ? c.qstart + gy * c.qW + gx : -1;
}

// --------------------------------------------------------------------------------------------------
// forward
// --------------------------------------------------------------------------------------------------
// RAW: the kernel also does the job of msda_prep_kernel (msda.hip) — loc / attn are OUTPUTS (the backward reads them)
// computed from the 288-wide projection `raw` [N*Lq][M*LP*2 offsets | M*LP logits] and the reference points `ref`
// [Lq][2]: attn = softmax over the L*P logits of a (query, head) (the thread holds its point's logit of every level;
// the 4 points of a query are a DPP quad), loc = ref + offset / (W_l, H_l)   (ms_deform_attn.py:106-119)
__device__ __forceinline__ float quad_max(float v)
{
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)));
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true)));
}
__device__ __forceinline__ float quad_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
}

// DG ("device geometry", the mpf_msda_*_dev entry points): the geometry is read from `gd` (built by msda_geom_kernel from the
// device-side spatial_shapes / level_start_index) instead of the kernel argument `gk`, and the workgroup count is only known
// on the device — the launch is an upper-bound estimate, workgroups past the real count leave at once and, should the
// estimate have been low (odd level sizes), the others stride over the remainder.
template <int NL, bool RAW, bool DG = false>
__global__ __launch_bounds__(kT) void msda_fwd_block_kernel(const float* __restrict__ value, const float* loc_,
                                                            const float* attn_, float* __restrict__ out, GeomB gk,
                                                            int nblocks_k, int region_cap, unsigned value_bytes,
                                                            const float* __restrict__ raw, const float* __restrict__ ref,
                                                            unsigned* __restrict__ stats, const GeomB* __restrict__ gd = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* s_f = reinterpret_cast<float4*>(smem);
    int4* s_o = reinterpret_cast<int4*>(smem + 4096);
    int* s_bb = reinterpret_cast<int*>(smem + kOffBB);
    const GeomB& g = DG ? *gd : gk;
    if (DG && !g.ok) return;
    const int nblocks = DG ? g.nblk : nblocks_k;

  for (int it_ = 0;; ++it_) {
    int blk;
    if (!dg_next<DG>(it_, nblocks, blk)) break;
    BlockCtx c;
    block_of(g, blk, c);
    const int tid = threadIdx.x + dg_zero<DG>(), lane = tid & 63;
    if (tid < kMaxL * 4) s_bb[tid] = (tid & 2) ? INT_MIN : INT_MAX;
    if (tid >= 32 && tid < 64) reinterpret_cast<float*>(smem + kOffZero)[tid - 32] = 0.f;

    // decode role: thread = (query tid >> 2, point tid & 3), one sample per level
    const int q_d = query_of(g, c, tid >> 2);
    const int LP = NL * kP;
    float2 xy[NL];
    float at[NL];
    {
        const int64_t gi0 = ((int64_t)(c.b * g.Lq + max(q_d, 0)) * g.M + c.m) * LP + (tid & 3);
        if (RAW) {
            const float* r = raw + (int64_t)(c.b * g.Lq + max(q_d, 0)) * (g.M * LP * 3);
            const float2 rp = reinterpret_cast<const float2*>(ref)[max(q_d, 0)];
            float lg[NL];
            float2 of[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                of[l] = reinterpret_cast<const float2*>(r + c.m * LP * 2)[l * kP + (tid & 3)];
                lg[l] = r[g.M * LP * 2 + c.m * LP + l * kP + (tid & 3)];
            }
            float mx = lg[0];
#pragma unroll
            for (int l = 1; l < NL; ++l) mx = fmaxf(mx, lg[l]);
            mx = quad_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int l = 0; l < NL; ++l) { at[l] = expf(lg[l] - mx); sum += at[l]; }
            sum = quad_sum(sum);
            float* loc_w = const_cast<float*>(loc_);
            float* attn_w = const_cast<float*>(attn_);
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                at[l] = at[l] / sum;
                xy[l] = make_float2(rp.x + of[l].x / (float)g.W[l], rp.y + of[l].y / (float)g.H[l]);
                if (q_d >= 0) {
                    reinterpret_cast<float2*>(loc_w)[gi0 + l * kP] = xy[l];
                    attn_w[gi0 + l * kP] = at[l];
                }
            }
        } else {
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                xy[l] = reinterpret_cast<const float2*>(loc_)[gi0 + l * kP];
                at[l] = attn_[gi0 + l * kP];
            }
        }
    }
    __syncthreads();
    Dec dec[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        dec[l] = decode(xy[l], g.H[l], g.W[l], q_d >= 0);
        const Dec& d = dec[l];
        const int H = g.H[l], W = g.W[l];
        // box over the corners that exist (clamped to the image)
        const int xa = d.in ? max(d.x0, 0) : INT_MAX, ya = d.in ? max(d.y0, 0) : INT_MAX;
        const int xb = d.in ? min(d.x0 + 1, W - 1) : INT_MIN, yb = d.in ? min(d.y0 + 1, H - 1) : INT_MIN;
        const int x_lo = wave_min(xa), y_lo = wave_min(ya), x_hi = wave_max(xb), y_hi = wave_max(yb);
        if (lane == 0) {
            atomicMin(&s_bb[l * 4 + 0], x_lo); atomicMin(&s_bb[l * 4 + 1], y_lo);
            atomicMax(&s_bb[l * 4 + 2], x_hi); atomicMax(&s_bb[l * 4 + 3], y_hi);
        }
    }
    __syncthreads();

    // sampling role: 8 lanes per sample (16 B of the 128-B row each); lane group grp owns queries grp, grp + 32
    const int grp = tid >> 3, sub16 = (tid & 7) * 16;
    f2v acc[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(value), 0, value_bytes, 0x00020000);
    const f4v* s_fv = reinterpret_cast<const f4v*>(smem);

#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int xmin = s_bb[l * 4 + 0], ymin = s_bb[l * 4 + 1], xmax = s_bb[l * 4 + 2], ymax = s_bb[l * 4 + 3];
        const bool any = xmax >= xmin;
        const int rw = xmax - xmin + 1, rows = any ? rw * (ymax - ymin + 1) : 0;
        const bool lds_path = rows <= region_cap;     // wave-uniform (also when the level has no sample at all)
        if (stats && tid == 0 && rows > 0) atomicAdd(&stats[lds_path ? 0 : 1], 1u);
        {
            const Dec& d = dec[l];
            const float hx = 1.f - d.lx, hy = 1.f - d.ly, a = d.in ? at[l] : 0.f;
            s_f[tid] = make_float4(hy * hx * a, hy * d.lx * a, d.ly * hx * a, d.ly * d.lx * a);
            s_o[tid] = corner_offsets(g, d, l, lds_path, xmin, ymin, rw, c.b, c.m);
        }
        if (lds_path && rows > 0) stage_region(value, smem, g, c.b, c.m, l, xmin, ymin, rw, rows, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // explicit 2-wide FMAs (v_pk_fma_f32 with the weight broadcast by op_sel): left to itself the compiler packs
        // ACROSS corners and pays a v_mov per operand
        auto accumulate = [&](int qq, const f4v& w, const f4v& v0, const f4v& v1, const f4v& v2, const f4v& v3) {
            acc[qq][0] = __builtin_elementwise_fma(f2v{w.x, w.x}, v0.xy, acc[qq][0]);
            acc[qq][1] = __builtin_elementwise_fma(f2v{w.x, w.x}, v0.zw, acc[qq][1]);
            acc[qq][0] = __builtin_elementwise_fma(f2v{w.y, w.y}, v1.xy, acc[qq][0]);
            acc[qq][1] = __builtin_elementwise_fma(f2v{w.y, w.y}, v1.zw, acc[qq][1]);
            acc[qq][0] = __builtin_elementwise_fma(f2v{w.z, w.z}, v2.xy, acc[qq][0]);
            acc[qq][1] = __builtin_elementwise_fma(f2v{w.z, w.z}, v2.zw, acc[qq][1]);
            acc[qq][0] = __builtin_elementwise_fma(f2v{w.w, w.w}, v3.xy, acc[qq][0]);
            acc[qq][1] = __builtin_elementwise_fma(f2v{w.w, w.w}, v3.zw, acc[qq][1]);
        };
        if (lds_path) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
                for (int p = 0; p < kP; ++p) {
                    const int s = (grp + 32 * qq) * kP + p;
                    const f4v w = s_fv[s];
                    const int4 o = s_o[s];
                    const f4v v0 = *reinterpret_cast<const f4v*>(smem + o.x + sub16), v1 = *reinterpret_cast<const f4v*>(smem + o.y + sub16);
                    const f4v v2 = *reinterpret_cast<const f4v*>(smem + o.z + sub16), v3 = *reinterpret_cast<const f4v*>(smem + o.w + sub16);
                    accumulate(qq, w, v0, v1, v2, v3);
                }
            }
        } else {
            // L2 gathers: two samples (8 rows) in flight per lane group, then a scheduling fence so that the compiler
            // does not hoist all 32 row loads (128 registers) to the top
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
                for (int p = 0; p < kP; p += 2) {
                    const int s = (grp + 32 * qq) * kP + p;
                    const f4v wa = s_fv[s], wb = s_fv[s + 1];
                    const int4 oa = s_o[s], ob = s_o[s + 1];
                    const float4 a0 = buf_row(vrs, oa.x + sub16), a1 = buf_row(vrs, oa.y + sub16);
                    const float4 a2 = buf_row(vrs, oa.z + sub16), a3 = buf_row(vrs, oa.w + sub16);
                    const float4 b0 = buf_row(vrs, ob.x + sub16), b1 = buf_row(vrs, ob.y + sub16);
                    const float4 b2 = buf_row(vrs, ob.z + sub16), b3 = buf_row(vrs, ob.w + sub16);
                    accumulate(qq, wa, f4v{a0.x, a0.y, a0.z, a0.w}, f4v{a1.x, a1.y, a1.z, a1.w}, f4v{a2.x, a2.y, a2.z, a2.w},
                               f4v{a3.x, a3.y, a3.z, a3.w});
                    accumulate(qq, wb, f4v{b0.x, b0.y, b0.z, b0.w}, f4v{b1.x, b1.y, b1.z, b1.w}, f4v{b2.x, b2.y, b2.z, b2.w},
                               f4v{b3.x, b3.y, b3.z, b3.w});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (l + 1 < NL) __syncthreads();
    }
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int q = query_of(g, c, grp + 32 * qq);
        if (q >= 0)
            *reinterpret_cast<float4*>(out + ((int64_t)(c.b * g.Lq + q) * g.M + c.m) * kD + (tid & 7) * 4) =
                make_float4(acc[qq][0].x, acc[qq][0].y, acc[qq][1].x, acc[qq][1].y);
    }
    if (!DG) break;
    __syncthreads();            // (the LDS tables are rebuilt by the next block of this workgroup)
  }
}

// --------------------------------------------------------------------------------------------------
// backward, kernel 1 ("push"): grad_attn / grad_loc (or their raw-projection form) + tile entry lists
// --------------------------------------------------------------------------------------------------
// LDS hash: slot of `key`, or -1 when the table is full
__device__ __forceinline__ int hash_slot(unsigned* keys, unsigned key)
{
    unsigned s = (key * 2654435761u) >> 24;          // 8 bits
#pragma unroll 1
    for (int probe = 0; probe < kSlots; ++probe) {
        const unsigned old = atomicCAS(&keys[s], kEmpty, key);
        if (old == kEmpty || old == key) return (int)s;
        s = (s + 1) & (kSlots - 1);
    }
    return -1;
}

// value rows in LDS (the tile kernel's grad_out rows): 128 B each (no padding — they arrive by DMA); the eight 16-byte pieces of LDS
// row r are stored at slot (piece ^ ((r >> 1) & 7)): the swizzle is applied on the SOURCE side of the DMA, and per-lane row reads
// (same piece, different rows) collide only for rows equal mod 16
__device__ __forceinline__ int swz16(int r) { return ((r >> 1) & 7) << 4; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTP = 256;          // tile-kernel workgroup: kWP waves
constexpr int kWP = kTP / 64;

// --------------------------------------------------------------------------------------------------
// backward, third generation: "bin" + "tile"  (round 4)
// --------------------------------------------------------------------------------------------------
// The push kernel above is a QUERY-centric pass: per 8x8 query block and level it stages the bounding box of the block's
// samples (halo re-reads: 246 MB for 137 algorithmic), reduces every sample against its four corner rows, and emits the
// tile entries; the pull kernel then gathers loc / attn / grad_out per entry a second time.  With scattered (trained)
// offsets the boxes no longer fit and push falls back to L2 gathers (146 -> 262 us).  Here the whole arithmetic moves to
// the DESTINATION side, where it is independent of how far the queries look:
//
//   bin   (query blocks, no value / no reduction): decode every sample, count it into its <= 4 destination tiles
//         (direct-mapped LDS counters, one global add per (workgroup, tile)), write the 4-byte entries; samples outside
//         the map get their zero gradients here.  Raw form: delta[q, m] = <grad_out[q, m, :], out[q, m, :]> — the
//         softmax-backward sum  sum_j a_j dA_j  of a (query, head) equals it, because out = sum_j a_j val_j — so the
//         per-sample gradients need nothing from the other samples of their query.
//   tile  (one wave per (4x4 tile, part), as pull): per chunk of 64 entries the grad_out rows of the chunk's queries are
//         copied global -> LDS by DMA ONCE and serve both roles: (a) the hat-function MFMA product that produces
//         grad_value (as pull), (b) for the samples the tile OWNS (top-left corner inside the tile; every sample has
//         exactly one owner) the four corner dot products against the tile's 5x5 value neighbourhood (also in LDS),
//         i.e. grad_attn / grad_loc (or their raw-projection form) — lane = sample walks the 32 channels, no cross-lane
//         reduction.  Value rows are read 25/16 x, nothing depends on a bounding box.
// per-wave LDS of the tile kernel: 64 grad_out rows | 25 value rows | per sample {a * hat_x(4 pixel columns), hat_y(4 pixel rows)}:
// 4 waves x 13 440 B = 53 760 B = 42 LDS granules of 1 280 B -> exactly three workgroups in a CU's 160 KB
constexpr int kTG = 8192, kTV = 3200, kTR = 2048;
constexpr int kTWave = kTG + kTV + kTR;
constexpr int kBDummy = 64;                              // lane-private dummy counters of the bin kernel

template <int NL, bool RAW, bool DG = false>
__global__ __launch_bounds__(kT) void msda_bwd_bin_kernel(
    const float* __restrict__ loc, const float* __restrict__ attn, const float* __restrict__ grad_out, const float* __restrict__ fwd_out,
    float* __restrict__ grad_loc, float* __restrict__ grad_attn, float* __restrict__ grad_raw, float* __restrict__ delta,
    int* __restrict__ tile_count, unsigned* __restrict__ entries, int* __restrict__ ovf_count, uint2* __restrict__ ovf, GeomB gk,
    int nblocks_k, unsigned* __restrict__ stats, float* __restrict__ graw_amax, int reverse, const GeomB* __restrict__ gd = nullptr)
{
    __shared__ int s_bb[kMaxL * 4];
    __shared__ unsigned s_keys[kSlots];
    __shared__ int s_cnt[kSlots + kBDummy];
    __shared__ int s_base[kSlots];
    __shared__ float ared[4];
    constexpr int LP = NL * kP;
    const GeomB& g = DG ? *gd : gk;
    if (DG && !g.ok) return;
    const int nblocks = DG ? g.nblk : nblocks_k;
  for (int it_ = 0;; ++it_) {
    // (reverse: tests only — the query blocks in the opposite order, i.e. another arrival order of the entries in their runs)
    int blk0;
    if (!dg_next<DG>(it_, nblocks, blk0)) break;
    const int blk = reverse ? nblocks - 1 - blk0 : blk0;
    BlockCtx c;
    block_of(g, blk, c);
    const int tid = threadIdx.x + dg_zero<DG>(), lane = tid & 63;
    if (tid < kMaxL * 4) s_bb[tid] = (tid & 2) ? INT_MIN : INT_MAX;
    s_keys[tid] = kEmpty; s_cnt[tid] = 0;                      // kSlots == kT
    if (tid < kBDummy) s_cnt[kSlots + tid] = 0;
    const int qi_d = tid >> 2, p_d = tid & 3;
    const int q_d = query_of(g, c, qi_d);
    const int64_t gi0 = ((int64_t)(c.b * g.Lq + max(q_d, 0)) * g.M + c.m) * LP + p_d;
    float2 xy[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) xy[l] = reinterpret_cast<const float2*>(loc)[gi0 + l * kP];
    float dl = 0.f;
    if (RAW) {
        // delta of (query, head): the four lanes of a query take 8 channels each
        const int64_t ro = ((int64_t)(c.b * g.Lq + max(q_d, 0)) * g.M + c.m) * kD + p_d * 8;
        const f4v g0 = *reinterpret_cast<const f4v*>(grad_out + ro), g1 = *reinterpret_cast<const f4v*>(grad_out + ro + 4);
        const f4v o0 = *reinterpret_cast<const f4v*>(fwd_out + ro), o1 = *reinterpret_cast<const f4v*>(fwd_out + ro + 4);
        dl = g0.x * o0.x + g0.y * o0.y + g0.z * o0.z + g0.w * o0.w + g1.x * o1.x + g1.y * o1.y + g1.z * o1.z + g1.w * o1.w;
        dl = quad_sum(dl);
        if (p_d == 0 && q_d >= 0) delta[(int64_t)(c.b * g.Lq + q_d) * g.M + c.m] = dl;
    }
    __syncthreads();
    const int bm = c.b * g.M + c.m;
    int x0[NL], y0[NL];
    unsigned in_mask = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const Dec d = decode(xy[l], g.H[l], g.W[l], q_d >= 0);
        x0[l] = d.x0; y0[l] = d.y0;
        in_mask |= d.in ? (1u << l) : 0u;
        const int H = g.H[l], W = g.W[l];
        const int xa = d.in ? max(d.x0, 0) : INT_MAX, ya = d.in ? max(d.y0, 0) : INT_MAX;
        const int xb = d.in ? min(d.x0 + 1, W - 1) : INT_MIN, yb = d.in ? min(d.y0 + 1, H - 1) : INT_MIN;
        const int x_lo = wave_min(xa), y_lo = wave_min(ya), x_hi = wave_max(xb), y_hi = wave_max(yb);
        if (lane == 0) {
            atomicMin(&s_bb[l * 4 + 0], x_lo); atomicMin(&s_bb[l * 4 + 1], y_lo);
            atomicMax(&s_bb[l * 4 + 2], x_hi); atomicMax(&s_bb[l * 4 + 3], y_hi);
        }
    }
    __syncthreads();
    // destination tiles: as in the push kernel (direct-mapped counters over the tile grid under the block's boxes, the
    // compare-and-swap hash when that grid has more than kSlots tiles); inactive adds go to a LANE-PRIVATE dummy counter
    // (one shared dummy made ~60 % of a workgroup's 3 072 adds hit the same LDS address)
    int tpk[NL][4];
    int tbx[NL], tby[NL], tbw[NL], tof[NL + 1];
    tof[0] = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int xmin = s_bb[l * 4 + 0], ymin = s_bb[l * 4 + 1], xmax = s_bb[l * 4 + 2], ymax = s_bb[l * 4 + 3];
        const bool any = xmax >= xmin;
        tbx[l] = xmin >> 2; tby[l] = ymin >> 2;
        tbw[l] = any ? (xmax >> 2) - tbx[l] + 1 : 0;
        tof[l + 1] = tof[l] + (any ? tbw[l] * ((ymax >> 2) - tby[l] + 1) : 0);
    }
    const bool direct = tof[NL] <= kSlots;             // workgroup-uniform
    if (stats && tid == 0) atomicAdd(&stats[direct ? 4 : 5], 1u);
    const int dummy = kSlots + (tid & (kBDummy - 1));
    if (direct) {
        int ret[NL][4];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const bool in = (in_mask >> l) & 1;
            const int H = g.H[l], W = g.W[l];
            const int txa = max(x0[l], 0) >> 2, txb = min(x0[l] + 1, W - 1) >> 2, tya = max(y0[l], 0) >> 2, tyb = min(y0[l] + 1, H - 1) >> 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ty = (e >> 1) ? tyb : tya, tx = (e & 1) ? txb : txa;
                const bool act = in && !((e & 1) && txb == txa) && !((e >> 1) && tyb == tya);
                const int slot = act ? tof[l] + (ty - tby[l]) * tbw[l] + (tx - tbx[l]) : dummy;
                tpk[l][e] = act ? slot : -1;
                ret[l][e] = atomicAdd(&s_cnt[slot], act ? 1 : 0);
            }
        }
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int e = 0; e < 4; ++e) tpk[l][e] = tpk[l][e] >= 0 ? ((tpk[l][e] << 16) | ret[l][e]) : -1;
    } else {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const bool in = (in_mask >> l) & 1;
            const int H = g.H[l], W = g.W[l];
            const int txa = max(x0[l], 0) >> 2, txb = min(x0[l] + 1, W - 1) >> 2, tya = max(y0[l], 0) >> 2, tyb = min(y0[l] + 1, H - 1) >> 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ty = (e >> 1) ? tyb : tya, tx = (e & 1) ? txb : txa;
                const bool act = in && !((e & 1) && txb == txa) && !((e >> 1) && tyb == tya);
                tpk[l][e] = -1;
                if (act) {
                    const int key = bm * g.tiles_per_bm + g.tile_base[l] + ty * g.ntx[l] + tx;
                    const int slot = hash_slot(s_keys, (unsigned)key);
                    if (slot >= 0) {
                        tpk[l][e] = (slot << 16) | atomicAdd(&s_cnt[slot], 1);
                    } else {
                        const int pos = atomicAdd(&tile_count[key], 1);
                        const unsigned ent = ((unsigned)q_d << 2) | (unsigned)p_d;
                        if (pos < g.cap[l]) {
                            entries[(int64_t)bm * g.ent_per_bm + g.ent_base[l] + (int64_t)(ty * g.ntx[l] + tx) * g.cap[l] + pos] = ent;
                        } else {
                            const int k = atomicAdd(ovf_count, 1);
                            ovf[k] = make_uint2((unsigned)key, ent);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    {   // one returning add per touched tile reserves the workgroup's run
        int my_base = 0;
        unsigned key = s_keys[tid];
        if (direct) {
            int l = 0;
#pragma unroll
            for (int k = 1; k < NL; ++k) l = tid >= tof[k] ? k : l;
            int bx_ = tbx[0], by_ = tby[0], bw_ = tbw[0], of_ = 0, tb_ = g.tile_base[0], nt_ = g.ntx[0];
#pragma unroll
            for (int k = 1; k < NL; ++k)
                if (l == k) { bx_ = tbx[k]; by_ = tby[k]; bw_ = tbw[k]; of_ = tof[k]; tb_ = g.tile_base[k]; nt_ = g.ntx[k]; }
            const int rel = tid - of_, ry = rel / max(bw_, 1), rx = rel - ry * bw_;
            key = tid < tof[NL] && s_cnt[tid] > 0 ? (unsigned)(bm * g.tiles_per_bm + tb_ + (by_ + ry) * nt_ + bx_ + rx) : kEmpty;
        }
        if (key != kEmpty) my_base = atomicAdd(&tile_count[key], s_cnt[tid]);
        s_base[tid] = my_base;
    }
    __syncthreads();
    float wmax = 0.f;            // raw form: largest magnitude this thread writes into grad_raw (amax slot of the GEMMs that read it)
    const unsigned ent = ((unsigned)max(q_d, 0) << 2) | (unsigned)p_d;
    if (q_d >= 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int txa = max(x0[l], 0) >> 2, txb = min(x0[l] + 1, g.W[l] - 1) >> 2;
        const int tya = max(y0[l], 0) >> 2, tyb = min(y0[l] + 1, g.H[l] - 1) >> 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (tpk[l][e] < 0) continue;
            const int local = ((e >> 1) ? tyb : tya) * g.ntx[l] + ((e & 1) ? txb : txa);
            const int pos = s_base[tpk[l][e] >> 16] + (tpk[l][e] & 0xFFFF);
            if (pos < g.cap[l]) {
                entries[(int64_t)bm * g.ent_per_bm + g.ent_base[l] + (int64_t)local * g.cap[l] + pos] = ent;
            } else {
                const int k = atomicAdd(ovf_count, 1);
                ovf[k] = make_uint2((unsigned)(bm * g.tiles_per_bm + g.tile_base[l] + local), ent);
            }
        }
        // a sample outside the map has no owner tile: its gradients (zero; raw form: the softmax term -a delta) are written here
        if (!((in_mask >> l) & 1)) {
            if (RAW) {
                const int no = g.M * LP * 2, nr = g.M * LP * 3, lp = l * kP + p_d;
                float* row = grad_raw + (int64_t)(c.b * g.Lq + q_d) * nr;
                reinterpret_cast<float2*>(row + c.m * LP * 2)[lp] = make_float2(0.f, 0.f);
                const float dlg = -attn[gi0 + l * kP] * dl;
                row[no + c.m * LP + lp] = dlg;
                wmax = fmaxf(wmax, fabsf(dlg));
            } else {
                grad_attn[gi0 + l * kP] = 0.f;
                reinterpret_cast<float2*>(grad_loc)[gi0 + l * kP] = make_float2(0.f, 0.f);
            }
        }
    }
    }
    if (RAW && graw_amax) {      // (uniform) one atomic max per workgroup
        amax_commit(graw_amax, wmax, ared);
    }
    if (!DG) break;
    __syncthreads();
  }
}

template <int NL, bool RAW, bool DBG, bool DG = false>
__global__ __launch_bounds__(kTP, 3) void msda_bwd_tile_kernel(
    const float* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn, const float* __restrict__ grad_out,
    const float* __restrict__ delta, const int* __restrict__ tile_count, const unsigned* __restrict__ entries,
    float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn, float* __restrict__ grad_raw, GeomB gk,
    int nwg_k, unsigned loc_bytes, unsigned* __restrict__ stats, int ablate_, unsigned long long* __restrict__ dbg_,
    float* __restrict__ graw_amax, float* __restrict__ gv_amax, const GeomB* __restrict__ gd = nullptr)
{
    // DBG (benchmarking: mpf_set_option("msda_push_ablate2") / mpf_debug_set_buffer): the ablation switches and phase stamps
    // exist only in that instantiation; the production kernel carries neither their branches nor their registers
    const int ablate = DBG ? ablate_ : 0;
    unsigned long long* const dbg = DBG ? dbg_ : nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const GeomB& g = DG ? *gd : gk;
    if (DG && !g.ok) return;
    const int nwg = DG ? g.nwg : nwg_k;
  for (int it_ = 0;; ++it_) {
    int wg;
    if (!dg_next<DG>(it_, nwg, wg)) break;
    const int tid = threadIdx.x + dg_zero<DG>(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // benchmarking only (mpf_debug_set_buffer): per wave [start, loop entry, sum wait, sum dots, sum operands, sum next-chunk, sum mfma, chunks]
    unsigned long long t_prev = 0, t_acc[5] = {0, 0, 0, 0, 0}, t_start = 0, t_loop = 0;
    int t_chunks = 0;
    auto stamp = [&](int k) {
        if (dbg) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) t_acc[k] += t - t_prev;
            t_prev = t;
        }
    };
    if (dbg) { t_start = __builtin_amdgcn_s_memtime(); }
    unsigned char* s_g = smem + wave * kTWave;            // [64 rows][128 B], 16-byte pieces XOR-swizzled by swz16(row)
    unsigned char* s_v = s_g + kTG;                        // [25 rows][128 B]: the 5 x 5 value neighbourhood, same swizzle
    float* s_w = reinterpret_cast<float*>(s_v + kTV);     // [64][8]: a * hat(column 0..3 of the tile), hat(row 0..3)
    const int bm = wg / g.wg_per_bm;
    const int r = wg - bm * g.wg_per_bm;
    int band, l, slot_base;
    if (g.band_uniform) {
        // (arithmetic: one dependent memory round trip less before the wave can ask for its tile's count and entries)
        band = r / g.wg_per_band;
        const int rr_ = r - band * g.wg_per_band;
        l = 0;
#pragma unroll
        for (int k = 1; k < NL; ++k) l = rr_ >= g.lvl_wg_base[k] ? k : l;
        slot_base = band * g.wg_per_band + sel(g.lvl_wg_base, l);
    } else {
        const int nslot = g.nband * NL;
        const int base_k = g.band_wg_base[min(lane, nslot - 1)];
        const unsigned long long ge = __ballot(lane < nslot && r >= base_k);
        const int slot = __builtin_amdgcn_readfirstlane(__popcll(ge) - 1);
        slot_base = __builtin_amdgcn_readlane(base_k, slot);
        band = slot / NL; l = slot - band * NL;
    }
    const int wpt = sel(g.wpt, l);
    const int W = sel(g.W, l), H = sel(g.H, l);
    const int nty = (H + 3) >> 2, ntx = sel(g.ntx, l);
    const int tile_base = sel(g.tile_base, l), cap = sel(g.cap, l), ent_base = sel(g.ent_base, l), start = sel(g.start, l);
    const int row0 = band * nty / g.nband, row1 = (band + 1) * nty / g.nband;
    const int unit = (r - slot_base) * kWP + wave;
    const int part = unit % wpt, tb = unit / wpt;
    const bool live = tb < (row1 - row0) * ntx;
    const int ty = row0 + tb / ntx, tx = tb - (tb / ntx) * ntx;
    const int local = live ? ty * ntx + tx : 0;
    const int b = bm / g.M, m = bm - b * g.M;
    constexpr int LP = NL * kP;
    const int MLP = g.M * LP;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float wmax_r = 0.f;          // largest magnitude this lane writes into grad_raw (the amax slot of the GEMMs that read it)
    const int kgrp = lane >> 4, j = lane & 15;
    const unsigned* ent = entries + (int64_t)bm * g.ent_per_bm + ent_base + (int64_t)local * cap;   // this tile's run
    // the first 64 slots of the run (cap >= 64) are requested together with the tile's count (one round trip instead of two);
    // slots at or past the count hold stale entries and are replaced by the run's last entry once the count is known
    unsigned e_first = ent[part * 64 + lane < cap ? part * 64 + lane : 0];
    unsigned e_second = ent[(part + wpt) * 64 + lane < cap ? (part + wpt) * 64 + lane : 0];
    int n = min(tile_count[bm * g.tiles_per_bm + tile_base + local], cap);
    if (!live) n = 0;
    const int nchunks = (n + 63) >> 6;
    if (part < nchunks) {
        // value neighbourhood: rows (4 ty + dy, 4 tx + dx), dy, dx = 0..4 (clamped to the image: a clamped row is only ever read
        // for a corner that is masked out), LDS row = dy * 5 + dx; DMA, 8 rows per wave instruction, source-side swizzle
        {
            const float* vbase = value + ((int64_t)(b * g.S + start) * g.M + m) * kD;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = min(i * 8 + (lane >> 3), 24);
                const int dy = rr / 5, dx = rr - dy * 5;
                const int yy = min(ty * 4 + dy, H - 1), xx = min(tx * 4 + dx, W - 1);
                const int piece = (lane & 7) ^ (((i * 8 + (lane >> 3)) >> 1) & 7);
                const float* src = vbase + (int64_t)(yy * W + xx) * (g.M * kD) + piece * 4;
                if (i < 3 || (lane >> 3) == 0)          // 25 rows: of the fourth piece only row 24 (masked lanes copy nothing)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(s_v + i * 1024), 16, 0, 0);
            }
        }
        const float fx0 = (float)(tx * 4), fy0 = (float)(ty * 4);
        const float fW = (float)W, fH = (float)H;
        const float rW = 1.0f / fW, rH = 1.0f / fH;
        const __amdgpu_buffer_rsrc_t rs_loc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(loc), 0, loc_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_att = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(attn), 0, loc_bytes >> 1, 0x00020000);
        const int gi_base = (b * g.Lq * g.M + m) * LP + l * kP;              // sample index of (q = 0, p = 0)
        const float* w_x = s_w + kgrp * 8 + (j & 3);          // MFMA role: this lane's column / row weight of sample (4 s + kgrp)
        const float* w_y = s_w + kgrp * 8 + 4 + (j >> 2);
        const unsigned char* g_l = s_g + kgrp * 128 + (j & 1) * 8;
        const int so_loc = gi_base * 8, so_att = gi_base * 4;
        const float* go_base = grad_out + (int64_t)(b * g.Lq * g.M + m) * kD;
        const int MD = g.M * kD;
        // entries past the run's end repeat its last entry (with weight 0): every lane always has a real row to fetch
        auto load_entry = [&](int c) { return ent[min(c * 64 + lane, n - 1)]; };
        struct LA { float x, y, a, d; };
        auto gather_la = [&](unsigned e) {
            const int si = (int)(e >> 2) * MLP + (int)(e & 3);
            const auto v2 = __builtin_amdgcn_raw_buffer_load_b64(rs_loc, si * 8, so_loc, 0);
            LA t;
            t.x = __int_as_float(v2[0]); t.y = __int_as_float(v2[1]);
            t.a = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_att, si * 4, so_att, 0));
            t.d = RAW ? delta[(int64_t)(b * g.Lq + (int)(e >> 2)) * g.M + m] : 0.f;
            return t;
        };
        auto synth_la = [&](unsigned e) {          // benchmarking (ablate & 8): no gathers, a location inside the tile
            LA t;
            t.x = ((float)(tx * 4) + 1.3f + (float)(e & 3) * 0.5f) / fW; t.y = ((float)(ty * 4) + 1.6f) / fH; t.a = 0.08f; t.d = 0.f;
            return t;
        };
        // grad_out rows of a chunk -> LDS by DMA (row = entry index in the chunk; lane (row, slot) fetches piece slot ^ swizzle)
        // Lane (r8 = lane >> 3) fetches a piece of rows r8, 8 + r8, ..., 56 + r8: the eight query indices it needs are passed
        // through the row buffer itself (free at this point), transposed, so that they come back as two 16-byte reads.
        auto dma_rows = [&](unsigned e, int cnt) {                                  // cnt = entries in the chunk (1..64)
            int* s_q = reinterpret_cast<int*>(s_g);
            s_q[(lane & 7) * 8 + (lane >> 3)] = (int)(e >> 2) * MD;                // entry (i * 8 + r8) -> slot r8 * 8 + i
            const int4 qa = *reinterpret_cast<const int4*>(s_q + (lane >> 3) * 8), qb = *reinterpret_cast<const int4*>(s_q + (lane >> 3) * 8 + 4);
            int qo[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            // BOTH reads must have returned before the first piece is requested: piece 0 lands on the scratch slots, and the
            // compiler only waits for the operand it is about to use (lgkmcnt(3) with the second read still queued: measured
            // as a memory fault once the LDS queue was busy enough)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (DBG && dbg) {       // debugging: row offsets outside grad_out are counted (slot after the per-wave records) and clamped
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if ((unsigned)qo[i] > (unsigned)((g.Lq - 1) * MD) || (qo[i] % MD) != 0) {
                        atomicAdd(dbg + (size_t)nwg * kWP * 8, 1ull);
                        dbg[(size_t)nwg * kWP * 8 + 1] = ((unsigned long long)(unsigned)qo[i] << 32) | (unsigned)cnt;
                        dbg[(size_t)nwg * kWP * 8 + 2] = ((unsigned long long)(unsigned)e << 32) | (unsigned)(i * 64 + lane);
                        qo[i] = 0;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rr = i * 8 + (lane >> 3);
                const int piece = (lane & 7) ^ ((rr >> 1) & 7);
                const float* src = go_base + qo[i] + piece * 4;
                // piece i = rows 8 i .. 8 i + 7 — skipped when the run ends before it (its rows carry weight 0 and whatever
                // finite data the buffer holds; operand groups past the end are skipped as well): two pieces at a time
                if ((i & ~1) * 8 < cnt)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(s_g + i * 1024), 16, 0, 0);
            }
        };
        // sample role (lane = entry): footprint, ownership, and the hat weights of the tile's four pixel columns / rows
        // (zero outside the 2 x 2 footprint and for padding lanes) -> s_w
        struct SR { int x0, y0; float lx, ly, a, d; unsigned e; bool owned; };
        auto sample_role = [&](int c, unsigned e, const LA& t) {
            SR s;
            const bool valid = c * 64 + lane < n;
            const float px = pix(t.x, W), py = pix(t.y, H);
            const float xf = floorf(px), yf = floorf(py);
            s.x0 = (int)xf; s.y0 = (int)yf; s.lx = px - xf; s.ly = py - yf; s.a = t.a; s.d = t.d; s.e = e;
            s.owned = valid && ((max(s.x0, 0) >> 2) == tx) && ((max(s.y0, 0) >> 2) == ty);
            const float av = valid ? t.a : 0.f;
            const float dxp = fx0 - px, dyp = fy0 - py;
            f4v wxa, wyv;
            wxa.x = av * fmaxf(0.f, 1.f - fabsf(dxp)); wxa.y = av * fmaxf(0.f, 1.f - fabsf(dxp + 1.f));
            wxa.z = av * fmaxf(0.f, 1.f - fabsf(dxp + 2.f)); wxa.w = av * fmaxf(0.f, 1.f - fabsf(dxp + 3.f));
            wyv.x = fmaxf(0.f, 1.f - fabsf(dyp)); wyv.y = fmaxf(0.f, 1.f - fabsf(dyp + 1.f));
            wyv.z = fmaxf(0.f, 1.f - fabsf(dyp + 2.f)); wyv.w = fmaxf(0.f, 1.f - fabsf(dyp + 3.f));
            *reinterpret_cast<f4v*>(s_w + lane * 8) = wxa;
            *reinterpret_cast<f4v*>(s_w + lane * 8 + 4) = wyv;
            return s;
        };
        int c = part;
        unsigned e_cur;
        {   // padding slots of the speculative load -> the run's last entry (it is inside this wave's first 64 slots or later)
            const int last = n - 1 - part * 64;                              // >= 0: part < nchunks
            const unsigned e_last = __builtin_amdgcn_readlane(e_first, min(last, 63));
            e_cur = lane <= last ? e_first : e_last;
        }
        unsigned e_nxt;
        {   // same for the second chunk's slots (all of them padding when the run ends before it)
            const int last = n - 1 - (part + wpt) * 64;
            const unsigned e_last2 = last >= 0 ? __builtin_amdgcn_readlane(e_second, min(max(last, 0), 63))
                                               : __builtin_amdgcn_readlane(e_cur, 63);
            e_nxt = lane <= last ? e_second : e_last2;
        }
        LA la = (ablate & 8) ? synth_la(e_cur) : gather_la(e_cur);
        if (!(ablate & 4)) dma_rows(e_cur, n - c * 64);
        SR sr = sample_role(c, e_cur, la);
        LA la_n = (ablate & 8) ? synth_la(e_nxt) : gather_la(e_nxt);
        unsigned e_nn = load_entry(c + 2 * wpt);
        if (dbg) { t_loop = __builtin_amdgcn_s_memtime(); t_prev = t_loop; }
#pragma unroll 1
        for (; c < nchunks; c += wpt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this chunk's rows (and the value neighbourhood) have landed
            __builtin_amdgcn_wave_barrier();
            stamp(0);
            ++t_chunks;
            if (sr.owned && !(ablate & 1)) {
                const int x0 = sr.x0, y0 = sr.y0;
                const bool x0v = x0 >= 0, x1v = x0 + 1 <= W - 1, y0v = y0 >= 0, y1v = y0 + 1 <= H - 1;
                const int dx = x0 - tx * 4, dy = y0 - ty * 4;            // -1 .. 3
                const int r0 = max(dy, 0) * 5 + max(dx, 0), r1 = max(dy, 0) * 5 + dx + 1, r2 = (dy + 1) * 5 + max(dx, 0), r3 = (dy + 1) * 5 + dx + 1;
                const unsigned char* pg = s_g + lane * 128;
                const unsigned char *p0 = s_v + r0 * 128, *p1 = s_v + r1 * 128, *p2 = s_v + r2 * 128, *p3 = s_v + r3 * 128;
                const int sg = swz16(lane), s0 = swz16(r0), s1 = swz16(r1), s2 = swz16(r2), s3 = swz16(r3);
                f2v t0 = {0.f, 0.f}, t1 = {0.f, 0.f}, t2 = {0.f, 0.f}, t3 = {0.f, 0.f};
                // the five 16-byte pieces of channel octet k + 1 are requested before the FMAs of octet k (two register sets):
                // a lane's waits for the LDS overlap its arithmetic instead of following each other
                f4v gA, vA[4], gB, vB[4];
                auto fetch1 = [&](int k, f4v& gg, f4v (&vv)[4]) {
                    gg = *reinterpret_cast<const f4v*>(pg + ((k * 16) ^ sg));
                    vv[0] = *reinterpret_cast<const f4v*>(p0 + ((k * 16) ^ s0));
                    vv[1] = *reinterpret_cast<const f4v*>(p1 + ((k * 16) ^ s1));
                    vv[2] = *reinterpret_cast<const f4v*>(p2 + ((k * 16) ^ s2));
                    vv[3] = *reinterpret_cast<const f4v*>(p3 + ((k * 16) ^ s3));
                };
                auto fma1 = [&](const f4v& gg, const f4v (&vv)[4]) {
                    t0 = __builtin_elementwise_fma(gg.xy, vv[0].xy, t0); t0 = __builtin_elementwise_fma(gg.zw, vv[0].zw, t0);
                    t1 = __builtin_elementwise_fma(gg.xy, vv[1].xy, t1); t1 = __builtin_elementwise_fma(gg.zw, vv[1].zw, t1);
                    t2 = __builtin_elementwise_fma(gg.xy, vv[2].xy, t2); t2 = __builtin_elementwise_fma(gg.zw, vv[2].zw, t2);
                    t3 = __builtin_elementwise_fma(gg.xy, vv[3].xy, t3); t3 = __builtin_elementwise_fma(gg.zw, vv[3].zw, t3);
                };
                fetch1(0, gA, vA);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    fetch1(k + 1, gB, vB);
                    __builtin_amdgcn_sched_barrier(0);
                    fma1(gA, vA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 2 < 8) fetch1(k + 2, gA, vA);
                    __builtin_amdgcn_sched_barrier(0);
                    fma1(gB, vB);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float s0f = (y0v && x0v) ? t0.x + t0.y : 0.f, s1f = (y0v && x1v) ? t1.x + t1.y : 0.f;
                const float s2f = (y1v && x0v) ? t2.x + t2.y : 0.f, s3f = (y1v && x1v) ? t3.x + t3.y : 0.f;
                const float lx = sr.lx, ly = sr.ly, hx = 1.f - lx, hy = 1.f - ly, a = sr.a;
                const float ra = hy * (hx * s0f + lx * s1f) + ly * (hx * s2f + lx * s3f);
                const float rx = fW * a * (hy * (s1f - s0f) + ly * (s3f - s2f));
                const float ry = fH * a * (hx * (s2f - s0f) + lx * (s3f - s1f));
                const int q = (int)(sr.e >> 2), p = (int)(sr.e & 3);
                if (RAW) {
                    const int no = g.M * LP * 2, nr = g.M * LP * 3, lp = l * kP + p;
                    float* row = grad_raw + (int64_t)(b * g.Lq + q) * nr;
                    // d offset = d loc / (W, H): the reciprocals are formed once per wave (exact for power-of-two maps, else
                    // within an ulp of the division the push kernel did)
                    const float ox = rx * rW, oy = ry * rH, dlg = a * (ra - sr.d);
                    reinterpret_cast<float2*>(row + m * LP * 2)[lp] = make_float2(ox, oy);
                    row[no + m * LP + lp] = dlg;
                    wmax_r = fmaxf(wmax_r, fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(dlg)));
                } else {
                    const int64_t gi = ((int64_t)(b * g.Lq + q) * g.M + m) * LP + l * kP + p;
                    grad_attn[gi] = ra;
                    reinterpret_cast<float2*>(grad_loc)[gi] = make_float2(rx, ry);
                }
            }
            // MFMA role: lane = (pixel j, sample kgrp of the step).  The chunk's weights and the lane's 8 bytes of every row go to
            // REGISTERS in two halves of eight steps (groups of four; a group past the run's end is skipped, steps past it inside
            // a group carry weight 0): half A's MFMAs run first, then half B's operands are taken, which frees the row buffer and
            // the weight table for the NEXT chunk — its sample role runs and its rows are requested before half B's 16 MFMAs,
            // which then cover part of the copy's round trip.  (All 16 steps in registers at once cost 24 more registers across
            // the next-chunk preparation: the raw form spilled, i.e. paid scratch round trips per chunk.)
            stamp(1);
            const int nst = (ablate & 2) ? 0 : min(16, (n - c * 64 + 3) >> 2);
            float wv[8];
            float2 gq[8];
            auto take = [&](int half) {
#pragma unroll
                for (int s4 = 0; s4 < 2; ++s4) {
                    if ((half * 2 + s4) * 4 < nst) {
                        float wa[4], wb[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int st = (half * 2 + s4) * 4 + u;                       // rows 4 st + kgrp
                            wa[u] = w_x[st * 32];
                            wb[u] = w_y[st * 32];
                            gq[s4 * 4 + u] = *reinterpret_cast<const float2*>(g_l + st * 512 + (((j >> 1) << 4) ^ swz16(st * 4 + kgrp)));
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 4; ++u) wv[s4 * 4 + u] = wa[u] * wb[u];
                    }
                }
            };
            auto mfmas = [&](int half) {
#pragma unroll
                for (int s4 = 0; s4 < 2; ++s4) {
                    if ((half * 2 + s4) * 4 < nst) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s4 * 4 + u], gq[s4 * 4 + u].x, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s4 * 4 + u], gq[s4 * 4 + u].y, acc1, 0, 0, 0);
                        }
                    }
                }
            };
            take(0);
            mfmas(0);
            take(1);
            if (dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(2); }
            if (c + wpt < nchunks) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every LDS read of this chunk has returned
                __builtin_amdgcn_wave_barrier();
                // (sample role BEFORE the copy is requested: hipcc waits vmcnt(0) in front of any LDS access that follows an
                // LDS-DMA, which would expose the copy's round trip right here instead of under the MFMAs)
                sr = sample_role(c + wpt, e_nxt, la_n);
                if (!(ablate & 4)) dma_rows(e_nxt, n - (c + wpt) * 64);
                e_cur = e_nxt; e_nxt = e_nn;
                la_n = (ablate & 8) ? synth_la(e_nxt) : gather_la(e_nxt);
                e_nn = load_entry(c + 3 * wpt);
            }
            stamp(3);
            mfmas(1);
            if (dbg) { asm volatile("s_nop 0" :: "v"(acc0), "v"(acc1)); stamp(4); }
        }
    }
    if (dbg && lane == 0) {
        unsigned long long* o = dbg + (size_t)(wg * kWP + wave) * 8;
        o[0] = t_start; o[1] = t_loop; o[2] = t_acc[0]; o[3] = t_acc[1]; o[4] = t_acc[2]; o[5] = t_acc[3]; o[6] = t_acc[4]; o[7] = (unsigned long long)t_chunks;
    }
    if (stats && lane == 0 && live && part == 0 && n > 0) atomicAdd(&stats[wpt > 1 ? 7 : 8], 1u);
    if (wpt > 1) {
        // tiles of coarse levels are split over the waves of the workgroup: partial tiles go through the wave's own row buffer
        float* red = reinterpret_cast<float*>(s_g);
        if (part > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { red[q * 64 + lane] = acc0[q]; red[(4 + q) * 64 + lane] = acc1[q]; }
        }
        __syncthreads();
        if (part == 0) {
            for (int k = 1; k < wpt; ++k) {
                const float* o = reinterpret_cast<const float*>(smem + (wave + k) * kTWave);
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc0[q] += o[q * 64 + lane]; acc1[q] += o[(4 + q) * 64 + lane]; }
            }
        }
    }
    float wmax_v = 0.f;
    if (live && part == 0) {
        const int py_ = ty * 4 + kgrp;
        if (py_ < H) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int px_ = tx * 4 + q;
                if (px_ < W) {
                    *reinterpret_cast<float2*>(grad_value + ((int64_t)(b * g.S + start + py_ * W + px_) * g.M + m) * kD + j * 2) =
                        make_float2(acc0[q], acc1[q]);
                    wmax_v = fmaxf(wmax_v, fmaxf(fabsf(acc0[q]), fabsf(acc1[q])));
                }
            }
        }
    }
    // amax slots of the two outputs (the fp16 x 2 GEMMs that consume grad_value / grad_raw scale by them): the workgroup's
    // largest magnitudes -> one atomic max each, instead of a pass over the tensors afterwards
    if (gv_amax || (RAW && graw_amax)) {           // (uniform)
        float* red = reinterpret_cast<float*>(smem);   // (one buffer for the workgroup: wave 0's rows)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { wmax_v = fmaxf(wmax_v, __shfl_xor(wmax_v, o)); wmax_r = fmaxf(wmax_r, __shfl_xor(wmax_r, o)); }
        __syncthreads();                            // every wave is done with its buffers
        if (lane == 0) { red[wave * 2] = wmax_v; red[wave * 2 + 1] = wmax_r; }
        __syncthreads();
        if (tid < 2) {                              // thread 0: grad_value, thread 1: grad_raw — one atomic max each per workgroup
            const float mx = fmaxf(fmaxf(red[tid], red[2 + tid]), fmaxf(red[4 + tid], red[6 + tid]));
            float* slot = tid == 0 ? gv_amax : (RAW ? graw_amax : nullptr);
            if (slot) atomicMax(reinterpret_cast<unsigned*>(slot) + (blockIdx.x % kAmaxSub) * kAmaxStride, __float_as_uint(mx));
        }
    }
    if (!DG) break;
    __syncthreads();            // (every wave is done with its row buffers before the next slot's copies land in them)
  }
}

// spill entries of the third generation: as msda_bwd_spill_kernel, plus the per-sample gradients of the entries whose tile OWNS
// the sample (the tile kernel never saw them).  32 lanes per entry (lane = channel); correctness path.
template <int NL, bool RAW, bool DG = false>
__global__ __launch_bounds__(kT) void msda_bwd_spill3_kernel(const float* __restrict__ value, const float* __restrict__ loc,
                                                             const float* __restrict__ attn, const float* __restrict__ grad_out,
                                                             const float* __restrict__ delta, const int* __restrict__ ovf_count,
                                                             const uint2* __restrict__ ovf, float* __restrict__ grad_value,
                                                             float* __restrict__ grad_loc, float* __restrict__ grad_attn,
                                                             float* __restrict__ grad_raw, GeomB gk, unsigned* __restrict__ stats,
                                                             float* __restrict__ graw_amax, float* __restrict__ gv_amax,
                                                             const GeomB* __restrict__ gd = nullptr)
{
    constexpr int LP = NL * kP;
    const GeomB& g = DG ? *gd : gk;
    if (DG && !g.ok) return;
    const int n = *ovf_count;
    const int c = threadIdx.x & 31;
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[6], (unsigned)n);
    float gvmax = 0.f;
    for (int i = blockIdx.x * (kT / 32) + (threadIdx.x >> 5); i < n; i += gridDim.x * (kT / 32)) {
        const uint2 o = ovf[i];
        const int key = (int)o.x, q = (int)(o.y >> 2), p = (int)(o.y & 3);
        const int bm = key / g.tiles_per_bm;
        int r = key - bm * g.tiles_per_bm, l = 0;
#pragma unroll
        for (int k = 1; k < NL; ++k) if (r >= g.tile_base[k]) l = k;
        r -= g.tile_base[l];
        const int ty = r / g.ntx[l], tx = r - ty * g.ntx[l];
        const int b = bm / g.M, m = bm - b * g.M;
        const int W = g.W[l], H = g.H[l];
        const int64_t gi = ((int64_t)(b * g.Lq + q) * g.M + m) * LP + l * kP + p;
        const float2 xy = reinterpret_cast<const float2*>(loc)[gi];
        const float a = attn[gi];
        const float x = pix(xy.x, W), y = pix(xy.y, H);
        if (!(y > -1.f && x > -1.f && y < (float)H && x < (float)W)) continue;
        const float gq = grad_out[((int64_t)(b * g.Lq + q) * g.M + m) * kD + c];
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf;
        const bool owned = ((max(x0, 0) >> 2) == tx) && ((max(y0, 0) >> 2) == ty);
        float sc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int px = x0 + (k & 1), py = y0 + (k >> 1);
            const bool inimg = px >= 0 && py >= 0 && px <= W - 1 && py <= H - 1;
            const int64_t vo = ((int64_t)(b * g.S + g.start[l] + (inimg ? py * W + px : 0)) * g.M + m) * kD + c;
            float d = (owned && inimg) ? gq * value[vo] : 0.f;
#pragma unroll
            for (int s = 16; s >= 1; s >>= 1) d += __shfl_xor(d, s, 32);
            sc[k] = d;
            if (!inimg || px < tx * 4 || px > tx * 4 + 3 || py < ty * 4 || py > ty * 4 + 3) continue;
            const float wgt = fmaxf(0.f, 1.f - fabsf((float)px - x)) * fmaxf(0.f, 1.f - fabsf((float)py - y)) * a;
            const float add = wgt * gq;
            const float before = atomicAdd(grad_value + vo, add);
            // the tile kernel recorded max |grad_value| BEFORE these adds: |before| + |add| bounds every intermediate
            // and the final magnitude of the element, and the slot may hold an upper bound
            gvmax = fmaxf(gvmax, fabsf(before) + fabsf(add));
        }
        if (owned && c == 0) {
            const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
            const float ra = hy * (hx * sc[0] + lx * sc[1]) + ly * (hx * sc[2] + lx * sc[3]);
            const float rx = (float)W * a * (hy * (sc[1] - sc[0]) + ly * (sc[3] - sc[2]));
            const float ry = (float)H * a * (hx * (sc[2] - sc[0]) + lx * (sc[3] - sc[1]));
            if (RAW) {
                const int no = g.M * LP * 2, nr = g.M * LP * 3, lp = l * kP + p;
                float* row = grad_raw + (int64_t)(b * g.Lq + q) * nr;
                const float ox = rx / (float)W, oy = ry / (float)H, dlg = a * (ra - delta[(int64_t)(b * g.Lq + q) * g.M + m]);
                reinterpret_cast<float2*>(row + m * LP * 2)[lp] = make_float2(ox, oy);
                row[no + m * LP + lp] = dlg;
                if (graw_amax)
                    atomicMax(reinterpret_cast<unsigned*>(graw_amax) + (blockIdx.x % kAmaxSub) * kAmaxStride,
                              __float_as_uint(fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(dlg))));
            } else {
                grad_attn[gi] = ra;
                reinterpret_cast<float2*>(grad_loc)[gi] = make_float2(rx, ry);
            }
        }
    }
    if (gv_amax && gvmax > 0.f)      // (n == 0: nobody gets here with gvmax > 0; a non-negative float's bit pattern orders like the float)
        atomicMax(reinterpret_cast<unsigned*>(gv_amax) + (blockIdx.x % kAmaxSub) * kAmaxStride, __float_as_uint(gvmax));
}

// --------------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------------
int g_region_rows = 217;      // forward: usable rows of the staged box (the buffer is rounded up to whole 32-row stage passes)
int g_block_disable = 0;
int g_fuse_prep = 1;        // mpf_set_option("msda_fuse_prep"): softmax / location arithmetic inside the forward kernel
unsigned long long* g_dbg = nullptr;   // benchmarking only: s_memtime phase stamps of the push kernel, [workgroup][16]
int g_push_ablate = 0;       // benchmarking only: 1 = no tile entries, 2 = no reduction, 4 = no box staging
int g_bwd_sorted = 0;        // mpf_set_option("msda_bwd_sorted"): sort every tile run before the tile kernel (bit-reproducible grad_value)
int g_bin_reverse = 0;       // tests: bin kernel walks the query blocks backwards (another arrival order)
// tests only (mpf_set_option("msda_stats", 1) / mpf_msda_stats): which route every (workgroup, level) took.
//   [0] forward: boxes staged in LDS   [1] forward: L2-gather fallback (box larger than the region)
//   [2] push: boxes staged in LDS      [3] push: L2-gather fallback
//   [4] push: workgroups with the direct-mapped tile counters   [5] push: workgroups on the compare-and-swap hash
//   [6] spill: entries applied with atomics (run overflow)
//   [7] pull: non-empty tiles split over several waves (merged through LDS)   [8] pull: non-empty single-wave tiles
constexpr int kNStats = 16;
unsigned* g_stats = nullptr;

// `hs` = [L][2] (H, W) and, for the device form, `lsi` = [L] level starts (nullptr: the running sum).  Host and device run the
// same code (IEEE double arithmetic, no contraction-sensitive expression), so a geometry built by msda_geom_kernel equals the
// one the host would have built from a copy of the shapes.  ent_budget / tile_budget (device form; 0 = unlimited): capacity per
// (image, head) of the entry runs / tile counters the caller allocated from S, L, Lq alone — run capacities are halved until they
// fit (what does not fit a run spills: correct, slower), a tile count over budget cannot happen (see dev_budgets).
__host__ __device__ inline bool build_geom(GeomB& g, const int64_t* hs, const int64_t* lsi, int N, int S, int M, int L, int Lq,
                                           int64_t ent_budget = 0, int64_t tile_budget = 0)
{
    g.ok = 0; g.contiguous = 1; g.nblk = 0; g.nwg = 0;
    if (L < 1 || L > kMaxL) return false;
    g.L = L; g.M = M; g.Lq = Lq; g.S = S; g.N = N;
    int64_t start = 0;
    for (int l = 0; l < kMaxL; ++l) {
        g.H[l] = g.W[l] = 1; g.start[l] = 0; g.ntx[l] = 1; g.tile_base[l] = 0; g.cap[l] = 1; g.ent_base[l] = 0; g.wpt[l] = 1;
        g.qH[l] = g.qW[l] = 1; g.qstart[l] = 0; g.qnbx[l] = 1; g.qblk_base[l] = 0;
    }
    double expect_l[kMaxL] = {0, 0, 0, 0};
    for (int l = 0; l < L; ++l) {
        const int64_t H = hs[2 * l], W = hs[2 * l + 1];
        if (H <= 0 || W <= 0 || H > 16384 || W > 16384) return false;
        const int64_t st = lsi ? lsi[l] : start;
        if (st != start) g.contiguous = 0;
        if (st < 0 || st + H * W > (int64_t)S) return false;
        g.H[l] = (int)H; g.W[l] = (int)W; g.start[l] = (int)st;
        start += H * W;
        g.ntx[l] = (int)((W + 3) / 4);
        // expected entries of a tile when the samples follow the queries: Lq * P * 16 / (H W) samples, x (5/4)^2 for
        // footprints straddling tile borders; run capacity = twice that + slack, the rest spills
        const double expect = (double)Lq * kP * 16.0 / ((double)H * (double)W) * 1.5625;
        expect_l[l] = expect;
        int64_t cap = (int64_t)(2.0 * expect) + 64;
        cap = (cap + 3) & ~(int64_t)3;
        if (cap > (1 << 28)) cap = 1 << 28;
        g.cap[l] = (int)cap;
        // waves per tile: aim at ~2 chunks (128 entries) per wave
        const int w_ = expect > 1400.0 ? 16 : (expect > 700.0 ? 4 : (expect > 350.0 ? 2 : 1));
        g.wpt[l] = w_ < kWP ? w_ : kWP;
    }
    (void)expect_l;
    if (!lsi && start != S) return false;
    if (lsi) {      // level ranges must not overlap (the tile kernel writes every pixel's grad_value row exactly once)
        for (int a = 0; a < L; ++a)
            for (int b = a + 1; b < L; ++b) {
                const int64_t a0 = g.start[a], a1 = a0 + (int64_t)g.H[a] * g.W[a], b0 = g.start[b], b1 = b0 + (int64_t)g.H[b] * g.W[b];
                if (a0 < b1 && b0 < a1) return false;
            }
        if (start != S) g.contiguous = 0;
    }
    for (int round = 0;; ++round) {          // tile / entry bases; with a budget: halve the run capacities until the runs fit
        int64_t tbase = 0, ebase = 0;
        bool fits = true;
        for (int l = 0; l < L; ++l) {
            const int64_t ntiles = (int64_t)g.ntx[l] * ((g.H[l] + 3) / 4);
            g.tile_base[l] = (int)tbase; tbase += ntiles;
            if (ebase + ntiles * g.cap[l] >= (1ll << 31)) return false;
            g.ent_base[l] = (int)ebase; ebase += ntiles * g.cap[l];
        }
        if (tile_budget > 0 && tbase > tile_budget) return false;
        if (ent_budget > 0 && ebase > ent_budget) fits = false;
        g.tiles_per_bm = (int)tbase; g.ent_per_bm = (int)ebase;
        if (fits) break;
        bool shrunk = false;
        for (int l = 0; l < L; ++l) {
            const int c2 = ((g.cap[l] / 2) + 3) & ~3;
            if (c2 >= 4 && c2 < g.cap[l]) { g.cap[l] = c2; shrunk = true; }
        }
        if (!shrunk || round > 40) return false;
    }
    {   // bands = tile rows of the level with the fewest tile rows (at most kMaxBand)
        int nband = kMaxBand;
        for (int l = 0; l < L; ++l) { const int nty = (g.H[l] + 3) / 4; nband = nty < nband ? nty : nband; }
        g.nband = nband;
        int base = 0;
        for (int k = 0; k < kMaxBand * kMaxL; ++k) g.band_wg_base[k] = 0;
        for (int bnd = 0; bnd < nband; ++bnd)
            for (int l = 0; l < L; ++l) {
                const int nty = (g.H[l] + 3) / 4;
                const int rows = (bnd + 1) * nty / nband - bnd * nty / nband;
                g.band_wg_base[bnd * L + l] = base;
                base += (rows * g.ntx[l] * g.wpt[l] + kWP - 1) / kWP;
            }
        g.wg_per_bm = base;
        g.band_uniform = 1;
        for (int l = 0; l < L; ++l) if (((g.H[l] + 3) / 4) % nband != 0) g.band_uniform = 0;
        g.wg_per_band = g.band_uniform ? base / nband : 0;
        for (int l = 0; l < kMaxL; ++l) g.lvl_wg_base[l] = (g.band_uniform && l < L) ? g.band_wg_base[l] : (1 << 30);
    }
    if ((int64_t)N * M * g.tiles_per_bm >= (1ll << 31)) return false;
    // query blocks: the queries ARE the pixels of the levels (stored back to back) when their count says so, else one strip
    int bbase = 0;
    if ((int64_t)Lq == start) {
        g.nql = L; g.bw_log2 = 3; g.bh = 8;
        int64_t qs = 0;
        for (int l = 0; l < L; ++l) {
            g.qH[l] = g.H[l]; g.qW[l] = g.W[l]; g.qstart[l] = (int)qs;
            qs += (int64_t)g.H[l] * g.W[l];
            g.qnbx[l] = (g.W[l] + 7) / 8;
            g.qblk_base[l] = bbase;
            bbase += g.qnbx[l] * ((g.H[l] + 7) / 8);
        }
    } else {
        g.nql = 1; g.bw_log2 = 6; g.bh = 1;
        g.qH[0] = 1; g.qW[0] = Lq; g.qstart[0] = 0; g.qnbx[0] = (Lq + 63) / 64; g.qblk_base[0] = 0;
        bbase = g.qnbx[0];
    }
    g.blocks_per_b = bbase;
    if ((int64_t)N * M * bbase >= (1ll << 31) || (int64_t)N * M * g.wg_per_bm >= (1ll << 31)) return false;
    g.nblk = N * M * bbase;
    g.nwg = N * M * g.wg_per_bm;
    g.ok = 1;
    return true;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool block_ok(int N, int S, int M, int D, int L, int Lq, int P, int dtype)
{
    if (g_block_disable) return false;
    if (dtype != MPF_F32 || D != kD || P != kP || L < 1 || L > kMaxL) return false;
    if ((int64_t)N * S * M * D * 4 >= (1ll << 31) || (int64_t)N * Lq * M * L * P * 8 >= (1ll << 31) ||
        (int64_t)N * Lq * M * D * 4 >= (1ll << 31))
        return false;
    if ((int64_t)Lq >= (1 << 29)) return false;
    return true;
}

int region_bytes() { return ((g_region_rows + 31) / 32) * 32 * 128; }     // whole stage passes of 32 rows

template <int NL>
hipError_t launch_fwd(const float* value, const float* loc, const float* attn, float* out, const GeomB& g, hipStream_t st,
                      const float* raw = nullptr, const float* ref = nullptr, const GeomB* gd = nullptr, int dev_grid = 0)
{
    const int nblocks = g.N * g.M * g.blocks_per_b;
    const int grid = ((nblocks + 7) / 8) * 8;
    const size_t lds = kOffReg + region_bytes();
    const unsigned vb = (unsigned)((size_t)g.N * g.S * g.M * kD * 4);
    if (gd) {          // geometry on the device: g carries only N, S, M (host arguments)
        hipLaunchKernelGGL((msda_fwd_block_kernel<NL, false, true>), dim3(dev_grid), dim3(kT), lds, st, value, loc, attn, out, g, 0,
                           g_region_rows, vb, nullptr, nullptr, g_stats, gd);
        return hipGetLastError();
    }
    if (raw)
        hipLaunchKernelGGL((msda_fwd_block_kernel<NL, true>), dim3(grid), dim3(kT), lds, st, value, loc, attn, out, g, nblocks,
                           g_region_rows, vb, raw, ref, g_stats);
    else
        hipLaunchKernelGGL((msda_fwd_block_kernel<NL, false>), dim3(grid), dim3(kT), lds, st, value, loc, attn, out, g, nblocks,
                           g_region_rows, vb, raw, ref, g_stats);
    return hipGetLastError();
}

// zero-fill of the tile counters as a KERNEL (not hipMemsetAsync): a captured HIP graph replays a kernel node faithfully; the
// runtime's memset node of this odd byte count did not (replays of the captured pixel-decoder backward ran the bin / tile /
// spill kernels on stale counters: memory faults) — and the launch is cheaper for the launch thread than the memset call
// Deterministic mode (mpf_set_option("msda_bwd_sorted", 1)): the entries of a tile's run arrive in the order in which the bin
// kernel's workgroups won their atomic adds, and grad_value sums them in that order — fp32 reassociation from run to run.
// Sorting every run (entries are distinct (query, point) codes) fixes the order: one workgroup per tile, bitonic sort in LDS.
// Runs that overflowed their capacity keep a spill part whose atomics stay unordered (msda_bwd_spill3_kernel).
// Cost at config B, N = 2: ~300 us for this plain kernel (off by default).  Side finding: on sorted runs the tile kernel is
// 9-14 % FASTER (190 -> 171 us init offsets, 215 -> 186 us scattered), and sorting inside the 64-entry chunks alone gives most
// of it (175 / 196) while a point-major order is slower than arrival order (215 / 230): what counts is that the four points
// of a query sit next to each other in a chunk (their grad_out row, delta and loc / attn words coalesce in the gathers).
__global__ __launch_bounds__(256) void msda_sort_runs_kernel(const int* __restrict__ tile_count, unsigned* __restrict__ entries, GeomB g)
{
    extern __shared__ unsigned s_sort[];
    const int key = (int)blockIdx.x;
    const int bm = key / g.tiles_per_bm, r = key - bm * g.tiles_per_bm;
    int l = 0;
    for (int k = 1; k < g.L; ++k) l = r >= g.tile_base[k] ? k : l;
    const int cap = g.cap[l], local = r - g.tile_base[l];
    const int n = min(tile_count[key], cap);
    if (n < 2) return;
    unsigned* run = entries + (int64_t)bm * g.ent_per_bm + g.ent_base[l] + (int64_t)local * cap;
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += 256) s_sort[i] = i < n ? run[i] : 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n2 >> 1); t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;         // comparator (i, i + j)
                const unsigned a = s_sort[i], b = s_sort[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { s_sort[i] = b; s_sort[p] = a; }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 256) run[i] = s_sort[i];
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ p, int nwords)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nwords) p[i] = 0u;
}
inline hipError_t zero_counters(void* ws, size_t bytes, hipStream_t st)
{
    const int nwords = (int)((bytes + 3) / 4);
    hipLaunchKernelGGL(zero_words_kernel, dim3((nwords + 255) / 256), dim3(256), 0, st, (unsigned*)ws, nwords);
    return hipGetLastError();
}

struct WsLayout {
    size_t off_count, off_ovf_count, off_entries, off_ovf, off_delta, total;
    int ntiles;
};

WsLayout ws_layout(const GeomB& g)
{
    WsLayout w;
    w.ntiles = g.N * g.M * g.tiles_per_bm;
    w.off_count = 0;
    w.off_ovf_count = (size_t)w.ntiles * 4;
    w.off_entries = align256(w.off_ovf_count + 4);
    w.off_ovf = align256(w.off_entries + (size_t)g.N * g.M * g.ent_per_bm * 4);
    w.off_delta = align256(w.off_ovf + (size_t)g.N * g.Lq * g.M * g.L * kP * 4 * 8);
    w.total = w.off_delta + (size_t)g.N * g.Lq * g.M * 4;          // delta[q, m] of the raw form (third generation)
    return w;
}

// backward: zero counters -> bin -> tile -> spill3.  fwd_out (the forward result, [N, Lq, M * 32]) is needed by the raw form only.
template <int NL>
hipError_t launch_bwd3(const float* value, const float* loc, const float* attn, const float* go, const float* fwd_out, float* gv,
                       float* gl, float* ga, float* graw, const GeomB& g, void* workspace, hipStream_t st, float* graw_amax,
                       float* gv_amax)
{
    const WsLayout w = ws_layout(g);
    char* ws = (char*)workspace;
    int* tile_count = (int*)(ws + w.off_count);
    int* ovf_count = (int*)(ws + w.off_ovf_count);
    unsigned* entries = (unsigned*)(ws + w.off_entries);
    uint2* ovf = (uint2*)(ws + w.off_ovf);
    float* delta = (float*)(ws + w.off_delta);
    hipError_t err = zero_counters(ws, w.off_ovf_count + 4, st);
    if (err != hipSuccess) return err;
    const int nblocks = g.N * g.M * g.blocks_per_b;
    const int grid = ((nblocks + 7) / 8) * 8;
    constexpr int LP = NL * kP;
    const double esz = 4.0;
    const double n_samp = (double)g.N * g.Lq * g.M * LP, n_row = (double)g.N * g.Lq * g.M * kD;
    mpf::prof_begin(st);
    if (graw)
        hipLaunchKernelGGL((msda_bwd_bin_kernel<NL, true>), dim3(grid), dim3(kT), 0, st, loc, attn, go, fwd_out, gl, ga, graw, delta, tile_count,
                           entries, ovf_count, ovf, g, nblocks, g_stats, graw_amax, g_bin_reverse);
    else
        hipLaunchKernelGGL((msda_bwd_bin_kernel<NL, false>), dim3(grid), dim3(kT), 0, st, loc, attn, go, fwd_out, gl, ga, graw, delta, tile_count,
                           entries, ovf_count, ovf, g, nblocks, g_stats, graw_amax, g_bin_reverse);
    // algorithmic bytes of the pair (1344 e S N) are split as: bin = loc in; tile = the rest
    mpf::prof_end("msda_bwd_bin_kernel", st, esz * n_samp * 2);
    if (g_bwd_sorted) {
        int cap_max = 2;
        for (int l = 0; l < g.L; ++l) cap_max = std::max(cap_max, g.cap[l]);
        int n2 = 2;
        while (n2 < cap_max) n2 <<= 1;
        const size_t sort_lds = (size_t)n2 * sizeof(unsigned);
        if (sort_lds > 160 * 1024 - 1024) return hipErrorInvalidValue;          // (a run of > 40 k entries: not a shape this path sees)
        hipError_t ea = hipFuncSetAttribute((const void*)msda_sort_runs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds);
        if (ea != hipSuccess) return ea;
        mpf::prof_begin(st);
        hipLaunchKernelGGL(msda_sort_runs_kernel, dim3(g.N * g.M * g.tiles_per_bm), dim3(256), sort_lds, st, tile_count, entries, g);
        mpf::prof_end("msda_sort_runs_kernel", st, esz * n_samp * 1.5625 * 2);
    }
    const int nwg = g.N * g.M * g.wg_per_bm;
    const size_t lds = (size_t)kWP * kTWave;
    const unsigned loc_bytes = (unsigned)((size_t)g.N * g.Lq * g.M * LP * 8);
    mpf::prof_begin(st);
    const dim3 tgrid(((nwg + 7) / 8) * 8);
    if (g_push_ablate || g_dbg) {
        if (graw)
            hipLaunchKernelGGL((msda_bwd_tile_kernel<NL, true, true>), tgrid, dim3(kTP), lds, st, value, loc, attn, go, delta, tile_count,
                               entries, gv, gl, ga, graw, g, nwg, loc_bytes, g_stats, g_push_ablate, g_dbg, graw_amax, gv_amax);
        else
            hipLaunchKernelGGL((msda_bwd_tile_kernel<NL, false, true>), tgrid, dim3(kTP), lds, st, value, loc, attn, go, delta, tile_count,
                               entries, gv, gl, ga, graw, g, nwg, loc_bytes, g_stats, g_push_ablate, g_dbg, graw_amax, gv_amax);
    } else if (graw)
        hipLaunchKernelGGL((msda_bwd_tile_kernel<NL, true, false>), tgrid, dim3(kTP), lds, st, value, loc, attn, go, delta, tile_count,
                           entries, gv, gl, ga, graw, g, nwg, loc_bytes, g_stats, 0, nullptr, graw_amax, gv_amax);
    else
        hipLaunchKernelGGL((msda_bwd_tile_kernel<NL, false, false>), tgrid, dim3(kTP), lds, st, value, loc, attn, go, delta, tile_count,
                           entries, gv, gl, ga, graw, g, nwg, loc_bytes, g_stats, 0, nullptr, graw_amax, gv_amax);
    mpf::prof_end("msda_bwd_tile_kernel", st, esz * ((double)g.N * g.S * g.M * kD * 2 + n_row + n_samp * 4));
    if (graw)
        hipLaunchKernelGGL((msda_bwd_spill3_kernel<NL, true>), dim3(64), dim3(kT), 0, st, value, loc, attn, go, delta, ovf_count, ovf, gv, gl, ga, graw, g, g_stats, graw_amax, gv_amax);
    else
        hipLaunchKernelGGL((msda_bwd_spill3_kernel<NL, false>), dim3(64), dim3(kT), 0, st, value, loc, attn, go, delta, ovf_count, ovf, gv, gl, ga, graw, g, g_stats, graw_amax, gv_amax);
    return hipGetLastError();
}

// ---- geometry on the device (the reference's all-device op signature: ms_deform_attn.h:25-66 hands over spatial_shapes and
// level_start_index as device tensors; nothing here copies them to the host) ------------------------------------------------------
struct DevBudget {
    int64_t tiles_bm, ent_bm;          // capacity per (image, head) of the tile counters / entry runs
    int blk_grid, tile_grid;           // launch sizes (multiples of 8) of the query-block kernels / the tile kernel
    size_t off_geom, off_count, off_ovf_count, off_entries, off_ovf, total;
};

// Everything the host can size from (N, S, M, L, Lq) alone.  A level of H x W pixels has ceil(H/4) ceil(W/4) <= H W / 4 + 1
// tiles (thin maps are the worst case), so S / 4 + L counters per (image, head) always suffice; run capacities are what
// build_geom derives for maps with sides that are multiples of 4 (12.5 Lq + 67 H W / 16 entries per level) + 25 %, and the
// device shrinks them when an odd pyramid needs more (the spill list covers every sample).  Launch sizes: the counts of a
// pyramid with sides that are multiples of 8 + slack; the kernels stride when the real count is larger.
DevBudget dev_budget(int N, int S, int M, int L, int Lq, bool backward)
{
    DevBudget b;
    b.tiles_bm = (int64_t)S / 4 + L;
    b.ent_bm = (int64_t)(1.25 * (12.5 * (double)L * (double)Lq + 67.0 * ((double)S / 16.0 + 8.0 * L))) + 1024;
    if (b.ent_bm < 4 * b.tiles_bm) b.ent_bm = 4 * b.tiles_bm;
    const int64_t bm = (int64_t)N * M;
    const int64_t blk = bm * (((int64_t)Lq + 63) / 64 + 2 * L);
    // (tile workgroups of a pyramid whose levels shrink 4x: S / 16 tiles x 1/3 — 1, 2, 4 waves per tile from the finest level
    // down, four waves per workgroup — estimated at 3/8 + slack)
    const int64_t tw = bm * (((int64_t)S / 16 * 3 + 7) / 8 + 4 * L);
    b.blk_grid = (int)std::min<int64_t>(((blk + 7) / 8) * 8, 1 << 30);
    b.tile_grid = (int)std::min<int64_t>(((tw + 7) / 8) * 8, 1 << 30);
    b.off_geom = 0;
    b.off_count = 1024;
    b.off_ovf_count = b.off_count + (size_t)(bm * b.tiles_bm) * 4;
    b.off_entries = align256(b.off_ovf_count + 4);
    b.off_ovf = align256(b.off_entries + (backward ? (size_t)(bm * b.ent_bm) * 4 : 0));
    b.total = b.off_ovf + (backward ? (size_t)N * Lq * M * L * kP * 4 * 8 : 0);
    return b;
}
static_assert(sizeof(GeomB) <= 1024, "GeomB must fit the workspace header");

// The prologue of a *_dev call, ONE workgroup:
//   1. the geometry record at the head of the workspace is kept if it was built from exactly these inputs (a key of every
//      number it depends on: sizes, budgets, the 2 L shape words and the L level starts) — a training loop calls with the same
//      pyramid every step and the forward's record serves the backward, so the usual prologue is a key comparison (~2 us)
//      instead of ~10 us of single-thread integer / double arithmetic; any other workspace content fails the comparison;
//   2. shapes the blocked kernels cannot serve -> every output element NaN (the kernels all return at once; a result that merely
//      looked plausible would be worse than none) — one workgroup writing tens of MB is slow, and only on that path;
//   3. a backward whose level ranges leave gaps in value: zeros in grad_value before the tile kernel writes the rows the levels own.
struct GeomKey {
    unsigned magic;
    int N, S, M, L, Lq;
    long long ent_budget, tile_budget;
    long long shapes[2 * kMaxL], lsi[kMaxL];
};
constexpr unsigned kGeomMagic = 0x4d504647u;          // "MPFG"
constexpr int kGeomKeyOff = 768;                       // byte offset of the key inside the 1 KB workspace header
static_assert(sizeof(GeomB) <= kGeomKeyOff && kGeomKeyOff + sizeof(GeomKey) <= 1024, "workspace header layout");

__global__ __launch_bounds__(256) void msda_geom_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi, unsigned char* header,
                                                        int N, int S, int M, int L, int Lq, long long ent_budget, long long tile_budget,
                                                        float* __restrict__ o0, int64_t n0, float* __restrict__ o1, int64_t n1,
                                                        float* __restrict__ o2, int64_t n2, float* __restrict__ grad_value, int64_t nv)
{
    __shared__ int s_flags[2];
    GeomB* out = reinterpret_cast<GeomB*>(header);
    if (threadIdx.x == 0) {
        GeomKey* key = reinterpret_cast<GeomKey*>(header + kGeomKeyOff);
        GeomKey k;
        k.magic = kGeomMagic; k.N = N; k.S = S; k.M = M; k.L = L; k.Lq = Lq; k.ent_budget = ent_budget; k.tile_budget = tile_budget;
        for (int l = 0; l < kMaxL; ++l) {
            k.shapes[2 * l] = l < L ? shapes[2 * l] : 0; k.shapes[2 * l + 1] = l < L ? shapes[2 * l + 1] : 0;
            k.lsi[l] = l < L ? lsi[l] : 0;
        }
        bool same = key->magic == k.magic && key->N == N && key->S == S && key->M == M && key->L == L && key->Lq == Lq &&
                    key->ent_budget == ent_budget && key->tile_budget == tile_budget;
        for (int l = 0; l < kMaxL; ++l)
            same = same && key->shapes[2 * l] == k.shapes[2 * l] && key->shapes[2 * l + 1] == k.shapes[2 * l + 1] && key->lsi[l] == k.lsi[l];
        if (!same) {
            GeomB g;
            const bool ok = build_geom(g, shapes, lsi, N, S, M, L, Lq, ent_budget, tile_budget);
            g.ok = ok ? 1 : 0;
            g.L = L; g.M = M; g.Lq = Lq; g.S = S; g.N = N;
            *out = g;
            *key = k;
        }
        s_flags[0] = out->ok; s_flags[1] = out->contiguous;
    }
    __syncthreads();
    const int ok = s_flags[0], contiguous = s_flags[1];
    if (ok && (contiguous || !grad_value)) return;
    if (!ok) {
        const float nan = __int_as_float(0x7fc00000);
        for (int64_t i = threadIdx.x; i < n0; i += 256) o0[i] = nan;
        for (int64_t i = threadIdx.x; i < n1; i += 256) o1[i] = nan;
        for (int64_t i = threadIdx.x; i < n2; i += 256) o2[i] = nan;
    } else {
        for (int64_t i = threadIdx.x; i < nv; i += 256) grad_value[i] = 0.f;
    }
}

template <int NL>
hipError_t launch_bwd3_dev(const float* value, const float* loc, const float* attn, const float* go, float* gv, float* gl, float* ga,
                           const GeomB& gh, const GeomB* gd, const DevBudget& b, char* ws, hipStream_t st)
{
    int* tile_count = (int*)(ws + b.off_count);
    int* ovf_count = (int*)(ws + b.off_ovf_count);
    unsigned* entries = (unsigned*)(ws + b.off_entries);
    uint2* ovf = (uint2*)(ws + b.off_ovf);
    hipError_t err = zero_counters(ws + b.off_count, b.off_ovf_count + 4 - b.off_count, st);
    if (err != hipSuccess) return err;
    constexpr int LP = NL * kP;
    const double n_samp = (double)gh.N * gh.Lq * gh.M * LP, n_row = (double)gh.N * gh.Lq * gh.M * kD;
    mpf::prof_begin(st);
    hipLaunchKernelGGL((msda_bwd_bin_kernel<NL, false, true>), dim3(b.blk_grid), dim3(kT), 0, st, loc, attn, go, nullptr, gl, ga, nullptr, nullptr,
                       tile_count, entries, ovf_count, ovf, gh, 0, g_stats, nullptr, 0, gd);
    mpf::prof_end("msda_bwd_bin_kernel", st, 4.0 * n_samp * 2);
    const size_t lds = (size_t)kWP * kTWave;
    const unsigned loc_bytes = (unsigned)((size_t)gh.N * gh.Lq * gh.M * LP * 8);
    mpf::prof_begin(st);
    hipLaunchKernelGGL((msda_bwd_tile_kernel<NL, false, false, true>), dim3(b.tile_grid), dim3(kTP), lds, st, value, loc, attn, go, nullptr,
                       tile_count, entries, gv, gl, ga, nullptr, gh, 0, loc_bytes, g_stats, 0, nullptr, nullptr, nullptr, gd);
    mpf::prof_end("msda_bwd_tile_kernel", st, 4.0 * ((double)gh.N * gh.S * gh.M * kD * 2 + n_row + n_samp * 4));
    hipLaunchKernelGGL((msda_bwd_spill3_kernel<NL, false, true>), dim3(64), dim3(kT), 0, st, value, loc, attn, go, nullptr, ovf_count, ovf, gv, gl,
                       ga, nullptr, gh, g_stats, nullptr, nullptr, gd);
    return hipGetLastError();
}

}  // namespace

extern "C" int mpf_debug_set_buffer(void* p)
{
    g_dbg = (unsigned long long*)p;
    return 0;
}

// tests only: route counters of the blocked kernels (see g_stats).  Synchronises the device.
extern "C" int mpf_msda_stats(unsigned long long* out, int n, int reset)
{
    if (!out || n < 1) return MPF_E_NULL;
    for (int i = 0; i < n; ++i) out[i] = 0;
    if (!g_stats) return mpf::fail(MPF_E_SHAPE, "mpf_msda_stats: enable with mpf_set_option(\"msda_stats\", 1) first");
    unsigned h[kNStats];
    hipError_t err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMemcpy(h, g_stats, sizeof(h), hipMemcpyDeviceToHost);
    if (err == hipSuccess && reset) err = hipMemset(g_stats, 0, sizeof(h));
    if (err != hipSuccess) return mpf::check(err, "mpf_msda_stats");
    for (int i = 0; i < n && i < kNStats; ++i) out[i] = h[i];
    return 0;
}

namespace mpf {

// forward through the blocked kernel; returns -1000 when the problem is outside its shapes (caller falls back)
int msda_block_forward(const void* value, const int64_t* host_shapes, const void* loc, const void* attn, void* out, int N, int S, int M,
                       int D, int L, int Lq, int P, int dtype, hipStream_t st, const void* raw, const void* ref)
{
    if (!host_shapes || !block_ok(N, S, M, D, L, Lq, P, dtype)) return -1000;
    if (raw && (g_fuse_prep == 0 || (int64_t)N * Lq * M * L * P * 12 >= (1ll << 31))) return -1000;
    GeomB g;
    if (!build_geom(g, host_shapes, nullptr, N, S, M, L, Lq)) return -1000;
    mpf::prof_begin(st);
    mpf::set_kernel(raw ? "msda_fwd_block_kernel<raw>" : "msda_fwd_block_kernel");
    hipError_t err;
    const float *rw = (const float*)raw, *rf = (const float*)ref;
    switch (L) {
        case 1: err = launch_fwd<1>((const float*)value, (const float*)loc, (const float*)attn, (float*)out, g, st, rw, rf); break;
        case 2: err = launch_fwd<2>((const float*)value, (const float*)loc, (const float*)attn, (float*)out, g, st, rw, rf); break;
        case 3: err = launch_fwd<3>((const float*)value, (const float*)loc, (const float*)attn, (float*)out, g, st, rw, rf); break;
        default: err = launch_fwd<4>((const float*)value, (const float*)loc, (const float*)attn, (float*)out, g, st, rw, rf); break;
    }
    // value + loc/attn (read) + out; the raw form also WRITES loc / attn for the backward (the traffic of
    // the msda_prep launch it replaces)
    mpf::prof_end("msda_fwd_block_kernel", st,
                  4.0 * ((double)N * S * M * D + (double)N * Lq * M * L * P * 3 * (raw ? 2 : 1) + (double)N * Lq * M * D));
    return mpf::check(err, "msda_fwd_block_kernel");
}

size_t msda_block_workspace_bytes(const int64_t* host_shapes, int N, int M, int L, int Lq, int P)
{
    if (!host_shapes || P != kP || L < 1 || L > kMaxL) return 0;
    int64_t S = 0;
    for (int l = 0; l < L; ++l) S += host_shapes[2 * l] * host_shapes[2 * l + 1];
    GeomB g;
    if (S >= (1ll << 31) || !build_geom(g, host_shapes, nullptr, N, (int)S, M, L, Lq)) return 0;
    return ws_layout(g).total;
}

int msda_block_backward(const void* value, const int64_t* host_shapes, const void* loc, const void* attn, const void* go, void* gv,
                        void* gl, void* ga, void* graw, int N, int S, int M, int D, int L, int Lq, int P, int dtype, void* workspace,
                        size_t workspace_bytes, hipStream_t st, const void* fwd_out, float* graw_amax, float* gv_amax,
                        bool* amax_recorded)
{
    if (amax_recorded) *amax_recorded = false;
    if (!host_shapes || !block_ok(N, S, M, D, L, Lq, P, dtype)) return -1000;
    GeomB g;
    if (!build_geom(g, host_shapes, nullptr, N, S, M, L, Lq)) return -1000;
    if (workspace_bytes < ws_layout(g).total) return mpf::fail(MPF_E_SHAPE, "msda_backward_ws: workspace too small");
    // the amax slots are filled by the bin + tile (+ spill) kernels of the raw form; on every other route the caller runs
    // the amax passes itself (a switch or a geometry outside these kernels must not make the training step fail)
    // the raw form needs the forward result (delta = <grad_out, out>): without it the caller's round-1 kernels run
    if (graw && !fwd_out) return -1000;
    if (!graw) graw_amax = gv_amax = nullptr;
    else if (amax_recorded) *amax_recorded = true;
    mpf::set_kernel("msda_bwd_block(bin+tile)");
    hipError_t e3;
    const float *v_ = (const float*)value, *l_ = (const float*)loc, *a_ = (const float*)attn, *g_ = (const float*)go, *o_ = (const float*)fwd_out;
    switch (L) {
        case 1: e3 = launch_bwd3<1>(v_, l_, a_, g_, o_, (float*)gv, (float*)gl, (float*)ga, (float*)graw, g, workspace, st, graw_amax, gv_amax); break;
        case 2: e3 = launch_bwd3<2>(v_, l_, a_, g_, o_, (float*)gv, (float*)gl, (float*)ga, (float*)graw, g, workspace, st, graw_amax, gv_amax); break;
        case 3: e3 = launch_bwd3<3>(v_, l_, a_, g_, o_, (float*)gv, (float*)gl, (float*)ga, (float*)graw, g, workspace, st, graw_amax, gv_amax); break;
        default: e3 = launch_bwd3<4>(v_, l_, a_, g_, o_, (float*)gv, (float*)gl, (float*)ga, (float*)graw, g, workspace, st, graw_amax, gv_amax); break;
    }
    return mpf::check(e3, "msda_bwd_block(bin+tile)");
}

int set_block_option(const char* key, int v)
{
    if (!strcmp(key, "msda_region_rows")) {
        if (v < 8 || v > 1000) return MPF_E_SHAPE;
        g_region_rows = v;
        return 0;
    }
    if (!strcmp(key, "msda_block_disable")) { g_block_disable = v; return 0; }
    if (!strcmp(key, "msda_fuse_prep")) { g_fuse_prep = v; return 0; }
    if (!strcmp(key, "msda_push_ablate2")) { g_push_ablate = v; return 0; }
    if (!strcmp(key, "msda_bwd_sorted")) { g_bwd_sorted = v != 0; return 0; }
    if (!strcmp(key, "msda_bin_reverse")) { g_bin_reverse = v != 0; return 0; }
    if (!strcmp(key, "msda_stats")) {
        if (v && !g_stats) {
            if (hipMalloc((void**)&g_stats, kNStats * sizeof(unsigned)) != hipSuccess) { g_stats = nullptr; return MPF_E_SHAPE; }
            (void)hipMemset(g_stats, 0, kNStats * sizeof(unsigned));
        } else if (!v && g_stats) {
            (void)hipDeviceSynchronize();
            (void)hipFree(g_stats);
            g_stats = nullptr;
        }
        return 0;
    }
    return 1;
}

}  // namespace mpf

// ---- the op with the reference's all-device signature on the blocked kernels -------------------------------------------------
extern "C" size_t mpf_msda_dev_workspace_bytes(int batch, int spatial_size, int num_heads, int num_levels, int num_query, int num_point,
                                               int backward)
{
    if (batch <= 0 || spatial_size <= 0 || num_heads <= 0 || num_levels < 1 || num_levels > kMaxL || num_query <= 0 || num_point != kP) return 0;
    const DevBudget b = dev_budget(batch, spatial_size, num_heads, num_levels, num_query, backward != 0);
    if ((int64_t)batch * num_heads * b.ent_bm >= (1ll << 31) || (int64_t)batch * num_heads * b.tiles_bm >= (1ll << 31)) return 0;
    return b.total;
}

namespace {
int dev_prologue(const int64_t* shapes, const int64_t* lsi, int N, int S, int M, int L, int Lq, bool backward, void* workspace,
                 size_t workspace_bytes, DevBudget& b, GeomB& gh, hipStream_t st, const char* who, float* o0, int64_t n0, float* o1 = nullptr,
                 int64_t n1 = 0, float* o2 = nullptr, int64_t n2 = 0, float* grad_value = nullptr, int64_t nv = 0)
{
    b = dev_budget(N, S, M, L, Lq, backward);
    if ((int64_t)N * M * b.ent_bm >= (1ll << 31) || (int64_t)N * M * b.tiles_bm >= (1ll << 31)) return -1000;
    if (!workspace || workspace_bytes < b.total) return mpf::fail(MPF_E_SHAPE, who);
    if ((uintptr_t)workspace & 255) return mpf::fail(MPF_E_SHAPE, "msda_*_dev: workspace must be 256-byte aligned");
    memset(&gh, 0, sizeof(gh));
    gh.N = N; gh.S = S; gh.M = M; gh.L = L; gh.Lq = Lq;
    hipLaunchKernelGGL(msda_geom_kernel, dim3(1), dim3(256), 0, st, shapes, lsi, (unsigned char*)workspace, N, S, M, L, Lq, (long long)b.ent_bm,
                       (long long)b.tiles_bm, o0, n0, o1, n1, o2, n2, grad_value, nv);
    return mpf::check(hipGetLastError(), "msda_geom_kernel");
}
}  // namespace

extern "C" int mpf_msda_forward_dev(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const void* sampling_loc,
                                    const void* attn_weight, void* output, int batch, int spatial_size, int num_heads, int channels,
                                    int num_levels, int num_query, int num_point, int dtype, void* workspace, size_t workspace_bytes,
                                    void* stream)
{
    const int N = batch, S = spatial_size, M = num_heads, D = channels, L = num_levels, Lq = num_query, P = num_point;
    hipStream_t st = (hipStream_t)stream;
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !output)
        return mpf::fail(MPF_E_NULL, "msda_forward_dev: NULL buffer");
    DevBudget b;
    GeomB gh;
    int r = block_ok(N, S, M, D, L, Lq, P, dtype) ? dev_prologue(spatial_shapes, level_start_index, N, S, M, L, Lq, false, workspace, workspace_bytes,
                                                                 b, gh, st, "msda_forward_dev: workspace too small (mpf_msda_dev_workspace_bytes)",
                                                                 (float*)output, (int64_t)N * Lq * M * D)
                                                  : -1000;
    if (r == -1000)          // other dtypes / head widths / point counts: the kernels that read the shapes themselves
        return mpf_msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, output, N, S, M, D, L, Lq, P, dtype, stream);
    if (r) return r;
    const GeomB* gd = (const GeomB*)workspace;
    mpf::prof_begin(st);
    mpf::set_kernel("msda_fwd_block_kernel<dev>");
    hipError_t err;
    const float *v_ = (const float*)value, *l_ = (const float*)sampling_loc, *a_ = (const float*)attn_weight;
    switch (L) {
        case 1: err = launch_fwd<1>(v_, l_, a_, (float*)output, gh, st, nullptr, nullptr, gd, b.blk_grid); break;
        case 2: err = launch_fwd<2>(v_, l_, a_, (float*)output, gh, st, nullptr, nullptr, gd, b.blk_grid); break;
        case 3: err = launch_fwd<3>(v_, l_, a_, (float*)output, gh, st, nullptr, nullptr, gd, b.blk_grid); break;
        default: err = launch_fwd<4>(v_, l_, a_, (float*)output, gh, st, nullptr, nullptr, gd, b.blk_grid); break;
    }
    mpf::prof_end("msda_fwd_block_kernel", st, 4.0 * ((double)N * S * M * D + (double)N * Lq * M * L * P * 3 + (double)N * Lq * M * D));
    return mpf::check(err, "msda_fwd_block_kernel<dev>");
}

extern "C" int mpf_msda_backward_dev(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const void* sampling_loc,
                                     const void* attn_weight, const void* grad_output, void* grad_value, void* grad_sampling_loc,
                                     void* grad_attn_weight, int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                     int num_query, int num_point, int dtype, void* workspace, size_t workspace_bytes, void* stream)
{
    const int N = batch, S = spatial_size, M = num_heads, D = channels, L = num_levels, Lq = num_query, P = num_point;
    hipStream_t st = (hipStream_t)stream;
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_output || !grad_value || !grad_sampling_loc ||
        !grad_attn_weight)
        return mpf::fail(MPF_E_NULL, "msda_backward_dev: NULL buffer");
    DevBudget b;
    GeomB gh;
    const int64_t n_s = (int64_t)N * Lq * M * L * P, n_v = (int64_t)N * S * M * D;
    int r = block_ok(N, S, M, D, L, Lq, P, dtype) ? dev_prologue(spatial_shapes, level_start_index, N, S, M, L, Lq, true, workspace, workspace_bytes,
                                                                 b, gh, st, "msda_backward_dev: workspace too small (mpf_msda_dev_workspace_bytes)",
                                                                 (float*)grad_value, n_v, (float*)grad_sampling_loc, n_s * 2,
                                                                 (float*)grad_attn_weight, n_s, (float*)grad_value, n_v)
                                                  : -1000;
    if (r == -1000)
        return mpf_msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                                 grad_attn_weight, N, S, M, D, L, Lq, P, dtype, stream);
    if (r) return r;
    const GeomB* gd = (const GeomB*)workspace;
    mpf::set_kernel("msda_bwd_block(bin+tile)<dev>");
    hipError_t e3;
    const float *v_ = (const float*)value, *l_ = (const float*)sampling_loc, *a_ = (const float*)attn_weight, *g_ = (const float*)grad_output;
    float *gv = (float*)grad_value, *gl = (float*)grad_sampling_loc, *ga = (float*)grad_attn_weight;
    switch (L) {
        case 1: e3 = launch_bwd3_dev<1>(v_, l_, a_, g_, gv, gl, ga, gh, gd, b, (char*)workspace, st); break;
        case 2: e3 = launch_bwd3_dev<2>(v_, l_, a_, g_, gv, gl, ga, gh, gd, b, (char*)workspace, st); break;
        case 3: e3 = launch_bwd3_dev<3>(v_, l_, a_, g_, gv, gl, ga, gh, gd, b, (char*)workspace, st); break;
        default: e3 = launch_bwd3_dev<4>(v_, l_, a_, g_, gv, gl, ga, gh, gd, b, (char*)workspace, st); break;
    }
    return mpf::check(e3, "msda_bwd_block(bin+tile)<dev>");
}

// tests / diagnostics: the geometry record a *_dev call left in its workspace (synchronises): out[0] = ok, [1] = contiguous,
// [2] = query-block workgroups, [3] = tile workgroups, [4] = entries per (image, head), [5] = tiles per (image, head), [6..9] = run capacities
extern "C" int mpf_msda_dev_geometry(const void* workspace, int* out, int n)
{
    if (!workspace || !out || n < 1) return MPF_E_NULL;
    GeomB g;
    hipError_t err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMemcpy(&g, workspace, sizeof(g), hipMemcpyDeviceToHost);
    if (err != hipSuccess) return mpf::check(err, "mpf_msda_dev_geometry");
    const int v[10] = {g.ok, g.contiguous, g.nblk, g.nwg, g.ent_per_bm, g.tiles_per_bm, g.cap[0], g.cap[1], g.cap[2], g.cap[3]};
    for (int i = 0; i < n && i < 10; ++i) out[i] = v[i];
    return 0;
}
