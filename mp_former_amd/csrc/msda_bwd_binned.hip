// MSDA backward without floating-point atomics: "bin by destination tile, reduce with exclusive
// ownership".  fp32, D = 32 channels per head.
//
// Why: grad_value is a scatter-add with ~48 contributions per 128-B row.  On MI355X a global fp32
// atomic costs one L2/fabric request per (instruction, 128-B row) at ~10 G requests/s chip-wide
// (tools/ubench/atomics.hip), so the reference's formulation (4 atomics per sample per channel,
// ms_deform_im2col_cuda.cuh:130-157) is pinned at ~0.8 ms per 1024x1024 image however it is tiled,
// and LDS fp32 atomics are slower still (0.33 lanes/clk/CU, tools/ubench/lds_atomics.hip).
//
// Three kernels + a scan, all on the caller's stream:
//   K1 push  : one workgroup = 64 consecutive queries of one head.  Decodes its 64*L*P samples once
//              into LDS, gathers the 4 corner rows of `value` (lane = channel, 2 samples per wave
//              instruction) to produce grad_attn / grad_loc (wave shuffles for the 32-channel
//              reductions), and COUNTS, per destination tile, the (sample, row) entries it will
//              emit (LDS hash + one global integer add per touched tile).
//   scan     : exclusive prefix sum of the per-tile counts.
//   K2 fill  : same decomposition, geometry only: reserves a contiguous run per (workgroup, tile)
//              with one returning integer add and writes 16-byte entries {grad_out row, w_left,
//              w_right, packed (y, x) within the tile}.
//   K3 pull  : one workgroup per destination tile (ts x ts pixels of one level, one head, one
//              image).  The entries of a tile are binned by pixel row mod 4, and wave w of the
//              workgroup streams bin w on its own (coalesced 16-B entries, one per lane, broadcast
//              with v_readlane; no queues, no barriers in the loop), accumulating
//              w * grad_out[row, c] into the LDS tile with plain read-modify-write — rows are
//              wave-exclusive and the two half-waves of an instruction hit neighbouring pixels —
//              then the tile is written to grad_value with coalesced stores.  Every element of grad_value is written exactly
//              once: no zero-fill pass, no float atomics, run-to-run deterministic up to the order
//              of entries inside a tile.
// K1 and K2 must derive identical entries, so the pixel coordinates are computed with explicitly
// rounded operations (no FMA contraction) in one shared function.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kD = 32;
constexpr int kQB = 32;          // queries per push/fill workgroup
constexpr int kMaxL = 8;
constexpr int kSlots = 256;      // LDS hash slots (distinct destination tiles per workgroup)
constexpr unsigned kEmpty = 0xFFFFFFFFu;

struct Geom {
    int L, P, M, Lq, S, N;
    int H[kMaxL], W[kMaxL], start[kMaxL];
    int ts[kMaxL], ntx[kMaxL], tile_base[kMaxL];
    int tiles_per_bm;
};

struct Entry {          // 16 bytes; everything the pull kernel needs is precomputed by the fill kernel
    unsigned g_off;     // BYTE offset of the grad_out row: ((b*Lq + q)*M + m) * 128
    float w0, w1;       // weight (bilinear * attention) of pixel (y, x) and (y, x+1)
    unsigned packed;    // bits 0..30: byte offset of pixel (y, x) in the bin's LDS image, bit 31: has1
};

// pixel coordinate of a sampling location; identical in every kernel (no contraction)
__device__ __forceinline__ float pix(float loc, int size) { return __fsub_rn(__fmul_rn(loc, (float)size), 0.5f); }

struct Sample {
    bool in_range;
    int x0, y0;
    float lx, ly;
};

__device__ __forceinline__ Sample decode(float2 xy, int H, int W)
{
    Sample s;
    const float x = pix(xy.x, W), y = pix(xy.y, H);
    s.in_range = (y > -1.f && x > -1.f && y < (float)H && x < (float)W);
    const float xf = floorf(x), yf = floorf(y);
    s.x0 = (int)xf; s.y0 = (int)yf; s.lx = x - xf; s.ly = y - yf;
    return s;
}

__device__ __forceinline__ int tile_of(const Geom& g, int bm, int l, int y, int x)
{
    const int ts = g.ts[l];
    return bm * g.tiles_per_bm + g.tile_base[l] + (y / ts) * g.ntx[l] + (x / ts);
}

// LDS hash: returns slot of `key`, or -1 if the table is full (caller falls back to a global op)
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

// "add 1 to the counter of `bin`": every lane does its own LDS compare-and-swap + add (measured 5-20x
// cheaper than aggregating equal bins with wave votes: the vote loop serialises LDS round trips).
// Returns the lane's index inside its (workgroup, bin) run; *slot_out = hash slot, or -1 if the table
// was full and the GLOBAL counter was used instead (then the index is relative to that counter).
__device__ __forceinline__ int lane_bin_add(bool active, int bin, unsigned* keys, int* cnt, int* gcount, int* slot_out)
{
    int idx = 0, slot = -2;
    if (active) {
        slot = hash_slot(keys, (unsigned)bin);
        idx = slot >= 0 ? atomicAdd(&cnt[slot], 1) : atomicAdd(&gcount[bin], 1);
    }
    *slot_out = slot;
    return idx;
}

// Entry e (0..3) of a sample: e>>1 selects the pixel row (y0 or y0+1), e&1 the first / second entry of
// that row (a row has two entries only when x0 and x0+1 fall into different tiles).
//   wsel = 0: the entry starts at x0 (w0 = left weight) [and also covers x0+1 if has1]
//   wsel = 1: the entry covers x0+1 only (w0 = right weight)
__device__ __forceinline__ bool entry_desc(const Geom& g, int bm, int l, const Sample& sm, int e,
                                           int* tile, int* y_out, int* xs, int* has1, int* wsel)
{
    const int H = g.H[l], W = g.W[l];
    const int y = sm.y0 + (e >> 1), k = e & 1;
    *y_out = y;
    *tile = 0; *xs = 0; *has1 = 0; *wsel = 0;
    if (!sm.in_range || y < 0 || y > H - 1) return false;
    const bool lv = sm.x0 >= 0, rv = sm.x0 + 1 <= W - 1;
    if (lv && rv) {
        const int ta = tile_of(g, bm, l, y, sm.x0), tb = tile_of(g, bm, l, y, sm.x0 + 1);
        if (ta == tb) {
            if (k) return false;
            *tile = ta; *xs = sm.x0; *has1 = 1;
            return true;
        }
        *tile = k ? tb : ta; *xs = sm.x0 + k; *wsel = k;
        return true;
    }
    if (k) return false;
    if (lv) { *tile = tile_of(g, bm, l, y, sm.x0); *xs = sm.x0; return true; }
    if (rv) { *tile = tile_of(g, bm, l, y, sm.x0 + 1); *xs = sm.x0 + 1; *wsel = 1; return true; }
    return false;
}

// sum over each 32-lane half of the wave with DPP (VALU cross-lane moves: no LDS traffic, unlike
// __shfl_xor which lowers to ds_bpermute).  The result is valid in lanes 16..31 and 48..63.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}

__device__ __forceinline__ float half_sum32(float v)
{
    v = dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);         // row_half_mirror
    v = dpp_add<0x140>(v);         // row_mirror      -> every lane: sum of its row of 16
    v = dpp_add<0x142, 0xa>(v);    // row_bcast15 into rows 1 and 3 -> sum of 32 lanes
    return v;
}

// ------------------------------------------------------------------------------------------------
// K1: grad_attn / grad_loc + per-tile entry counts
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void msda_bwd_push_kernel(
    const float* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn,
    const float* __restrict__ grad_out, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
    int* __restrict__ tile_count, Geom g, int nchunks, int nblocks, int ablate, unsigned value_bytes,
    float* __restrict__ grad_raw)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LP = g.L * g.P;
    const int cap = kQB * LP;
    int4* s_off = reinterpret_cast<int4*>(smem);                         // [cap] corner element offsets
    float4* s_f = reinterpret_cast<float4*>(smem + (size_t)cap * 16);    // [cap] lx, ly, a, -
    float* s_ga = reinterpret_cast<float*>(smem + (size_t)cap * 32);     // [cap]
    float2* s_gl = reinterpret_cast<float2*>(smem + (size_t)cap * 36);   // [cap]
    unsigned* s_keys = reinterpret_cast<unsigned*>(smem + (size_t)cap * 44);   // [kSlots]
    int* s_cnt = reinterpret_cast<int*>(s_keys + kSlots);                       // [kSlots]

    // XCD-aware: each XCD walks a contiguous range of (image, query chunk, head)
    const int per_xcd = (nblocks + 7) >> 3;
    const int blk = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (blk >= nblocks) return;
    const int m = blk % g.M;
    const int chunk = (blk / g.M) % nchunks;
    const int b = blk / (g.M * nchunks);
    const int q0 = chunk * kQB;
    const int nq = min(kQB, g.Lq - q0);
    const int tid = threadIdx.x;
    const int bm = b * g.M + m;

    for (int i = tid; i < kSlots; i += kThreads) { s_keys[i] = kEmpty; s_cnt[i] = 0; }
    __syncthreads();

    // ---- phase 1: decode + count (wave-uniform loop: the counting uses wave-wide votes) -----------
    for (int sbase = 0; sbase < nq * LP; sbase += kThreads) {
        const int s = sbase + tid;                      // query-major: coalesced loc / attn reads
        const bool valid = s < nq * LP;
        const int sc = valid ? s : 0;
        const int ql = sc / LP, lp = sc - ql * LP, l = lp / g.P;
        const int64_t gi = ((int64_t)(b * g.Lq + q0 + ql) * g.M + m) * LP + lp;
        const float2 xy = reinterpret_cast<const float2*>(loc)[gi];
        const float a = attn[gi];
        const int H = g.H[l], W = g.W[l];
        Sample sm = decode(xy, H, W);
        sm.in_range = sm.in_range && valid;
        int4 off = make_int4(-1, -1, -1, -1);
        float4 f = make_float4(0.f, 0.f, a, 0.f);
        if (sm.in_range) {
            f.x = sm.lx; f.y = sm.ly;
            const int sx = g.M * kD, sy = W * sx;
            const int base = ((b * g.S + g.start[l]) * g.M + m) * kD + sm.y0 * sy + sm.x0 * sx;
            const bool y0v = sm.y0 >= 0, y1v = sm.y0 + 1 <= H - 1, x0v = sm.x0 >= 0, x1v = sm.x0 + 1 <= W - 1;
            if (y0v && x0v) off.x = base;
            if (y0v && x1v) off.y = base + sx;
            if (y1v && x0v) off.z = base + sy;
            if (y1v && x1v) off.w = base + sy + sx;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int tile, y, xs, has1, wsel, slot;
            const bool act = entry_desc(g, bm, l, sm, e, &tile, &y, &xs, &has1, &wsel);
            if (!(ablate & 1)) lane_bin_add(act, tile * 4 + (y & 3), s_keys, s_cnt, tile_count, &slot);
        }
        if (valid) { s_off[s] = off; s_f[s] = f; }
    }
    __syncthreads();
    for (int i = tid; i < kSlots; i += kThreads)
        if (s_keys[i] != kEmpty) atomicAdd(&tile_count[s_keys[i]], s_cnt[i]);

    // ---- phase 2: gather value rows, per-sample reductions -----------------------------------------
    // lane = channel; each 32-lane half-wave owns whole queries (grad_out row loaded once per query)
    // and walks their L*P points in batches of 4: all 4 descriptors are read and all 16 corner-row
    // loads are issued before the first reduction, and the results go to LDS after the batch (an LDS
    // store between two descriptor reads would serialise the batches: same LDS array to the compiler).
    const int lane = tid & 63, c = lane & 31;
    const int sub = tid >> 5;                         // 0..7: half-wave index in the workgroup
    const int nq2 = (ablate & 2) ? 0 : nq;
    // corner offsets of -1 turn into an out-of-range byte offset: the buffer bounds check returns 0
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(value), 0, value_bytes, 0x00020000);
    auto ldv = [&](int elem_off) {
        const unsigned vo = elem_off < 0 ? 0x80000000u : (unsigned)elem_off * 4u + (unsigned)c * 4u;
        return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(vrs, (int)vo, 0, 0));
    };
    for (int ql = sub; ql < nq2; ql += 8) {
        const float go = grad_out[((int64_t)(b * g.Lq + q0 + ql) * g.M + m) * kD + c];
        for (int lp0 = 0; lp0 < LP; lp0 += 4) {
            int4 off[4];
            float4 f[4];
            float v[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lp = min(lp0 + k, LP - 1);
                off[k] = s_off[ql * LP + lp];
                f[k] = s_f[ql * LP + lp];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k][0] = ldv(off[k].x);
                v[k][1] = ldv(off[k].y);
                v[k][2] = ldv(off[k].z);
                v[k][3] = ldv(off[k].w);
            }
            float ra[4], rx[4], ry[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lx = f[k].x, ly = f[k].y, hx = 1.f - lx, hy = 1.f - ly;
                ra[k] = half_sum32(go * (hy * (hx * v[k][0] + lx * v[k][1]) + ly * (hx * v[k][2] + lx * v[k][3])));
                rx[k] = half_sum32(go * (hy * (v[k][1] - v[k][0]) + ly * (v[k][3] - v[k][2])));
                ry[k] = half_sum32(go * (hx * (v[k][2] - v[k][0]) + lx * (v[k][3] - v[k][1])));
            }
            if (c == 31) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int lp = lp0 + k;
                    if (lp < LP) {
                        const int l = lp / g.P;
                        s_ga[ql * LP + lp] = ra[k];
                        s_gl[ql * LP + lp] = make_float2((float)g.W[l] * f[k].z * rx[k], (float)g.H[l] * f[k].z * ry[k]);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (grad_raw) {
        // fused epilogue (module-level op): gradients wrt the RAW projection outputs instead of wrt
        // (loc, attn):  loc = ref + off / (W_l, H_l)  =>  d off = d loc / (W_l, H_l);
        // attn = softmax over the L*P logits of (q, m)  =>  d logit = a (dA - sum_j a_j dA_j).
        // Row layout of raw: [M*L*P*2 offsets | M*L*P logits].
        const int no = g.M * LP * 2, nr = g.M * LP * 3;
        for (int s = tid; s < nq * LP; s += kThreads) {
            const int ql = s / LP, lp = s - ql * LP, l = lp / g.P;
            float dot = 0.f;
            for (int j = 0; j < LP; ++j) dot += s_f[ql * LP + j].z * s_ga[ql * LP + j];
            float* row = grad_raw + (int64_t)(b * g.Lq + q0 + ql) * nr;
            const float2 gl = s_gl[s];
            reinterpret_cast<float2*>(row + m * LP * 2)[lp] = make_float2(gl.x / (float)g.W[l], gl.y / (float)g.H[l]);
            row[no + m * LP + lp] = s_f[s].z * (s_ga[s] - dot);
        }
        return;
    }
    for (int s = tid; s < nq * LP; s += kThreads) {
        const int ql = s / LP, lp = s - ql * LP;
        const int64_t gi = ((int64_t)(b * g.Lq + q0 + ql) * g.M + m) * LP + lp;
        grad_attn[gi] = s_ga[s];
        reinterpret_cast<float2*>(grad_loc)[gi] = s_gl[s];
    }
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of tile_count -> tile_start (single workgroup)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void tile_scan_kernel(const int* __restrict__ count, int* __restrict__ start,
                                                         int* __restrict__ total, int T)
{
    __shared__ int red[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < T ? count[i] : 0;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) red[wave] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { if (k < wave) wbase += red[k]; tot += red[k]; }
        const int carry = carry_s;
        if (i < T) start[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

// ------------------------------------------------------------------------------------------------
// K2: write the entries
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void msda_bwd_fill_kernel(
    const float* __restrict__ loc, const float* __restrict__ attn, const int* __restrict__ tile_start,
    int* __restrict__ tile_cursor, Entry* __restrict__ entries, Geom g, int nchunks, int nblocks)
{
    __shared__ unsigned s_keys[kSlots];
    __shared__ int s_cnt[kSlots];
    __shared__ int s_base[kSlots];   // (3 KB)
    const int LP = g.L * g.P;
    const int per_xcd = (nblocks + 7) >> 3;
    const int blk = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (blk >= nblocks) return;
    const int m = blk % g.M;
    const int chunk = (blk / g.M) % nchunks;
    const int b = blk / (g.M * nchunks);
    const int q0 = chunk * kQB;
    const int nq = min(kQB, g.Lq - q0);
    const int tid = threadIdx.x;
    const int bm = b * g.M + m;
    for (int i = tid; i < kSlots; i += kThreads) { s_keys[i] = kEmpty; s_cnt[i] = 0; }
    __syncthreads();

    // pass A: index of every entry inside its (workgroup, tile) run; kept in registers
    // (a thread owns at most kMaxOwn samples x 4 entries).
    constexpr int kMaxOwn = (kQB * 32 + kThreads - 1) / kThreads;   // supports L*P <= 32
    int my_slot[kMaxOwn][4], my_idx[kMaxOwn][4];
#pragma unroll
    for (int it = 0; it < kMaxOwn; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { my_slot[it][e] = -2; my_idx[it][e] = 0; }
        if (it * kThreads >= nq * LP) continue;          // uniform
        const int sp = tid + it * kThreads;
        const bool valid = sp < nq * LP;
        const int spc = valid ? sp : 0;
        const int ql = spc / LP, lp = spc - ql * LP, l = lp / g.P;
        const int64_t gi = ((int64_t)(b * g.Lq + q0 + ql) * g.M + m) * LP + lp;
        Sample sm = decode(reinterpret_cast<const float2*>(loc)[gi], g.H[l], g.W[l]);
        sm.in_range = sm.in_range && valid;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int tile, y, xs, has1, wsel, slot;
            const bool act = entry_desc(g, bm, l, sm, e, &tile, &y, &xs, &has1, &wsel);
            const int idx = lane_bin_add(act, tile * 4 + (y & 3), s_keys, s_cnt, tile_cursor, &slot);
            my_slot[it][e] = slot; my_idx[it][e] = idx;
        }
    }
    __syncthreads();
    for (int i = tid; i < kSlots; i += kThreads)
        if (s_keys[i] != kEmpty) s_base[i] = atomicAdd(&tile_cursor[s_keys[i]], s_cnt[i]);
    __syncthreads();

    // pass B: recompute the payloads and store
#pragma unroll
    for (int it = 0; it < kMaxOwn; ++it) {
        const int sp = tid + it * kThreads;
        if (sp >= nq * LP) continue;
        const int ql = sp / LP, lp = sp - ql * LP, l = lp / g.P;
        const int64_t gi = ((int64_t)(b * g.Lq + q0 + ql) * g.M + m) * LP + lp;
        const Sample sm = decode(reinterpret_cast<const float2*>(loc)[gi], g.H[l], g.W[l]);
        if (!sm.in_range) continue;
        const float a = attn[gi];
        const float hx = 1.f - sm.lx;
        const int g_row = (b * g.Lq + q0 + ql) * g.M + m;
        const int ts = g.ts[l];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int tile, y, xs, has1, wsel;
            if (!entry_desc(g, bm, l, sm, e, &tile, &y, &xs, &has1, &wsel)) continue;
            const int slot = my_slot[it][e];
            const int pos = tile_start[tile * 4 + (y & 3)] + (slot >= 0 ? s_base[slot] : 0) + my_idx[it][e];
            const float wy = ((e >> 1) ? sm.ly : 1.f - sm.ly) * a;
            Entry en;
            en.g_off = (unsigned)g_row * (unsigned)(kD * 4);
            en.w0 = wy * (wsel ? sm.lx : hx);
            en.w1 = has1 ? wy * sm.lx : 0.f;
            // pixel (y, xs) inside the bin's LDS image [(ts/4) rows][ts pixels][32 ch] (rows of class y&3)
            en.packed = (unsigned)((((y % ts) >> 2) * ts + (xs % ts)) * (kD * 4)) | ((unsigned)has1 << 31);
            entries[pos] = en;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3: one workgroup per destination tile
// ------------------------------------------------------------------------------------------------
constexpr int kPF = 16;      // grad_out rows per load group

// load the grad_out rows of entries [i0, i0+kPF) of the wave's current chunk (one entry per lane):
// buffer loads with the row's byte offset in an SGPR (soffset) and the channel in the VGPR offset —
// no per-entry vector address arithmetic
__device__ __forceinline__ void pull_load(float (&gv)[kPF], __amdgpu_buffer_rsrc_t rsrc, unsigned g_off_lane, int i0, int c4)
{
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        const unsigned go = (unsigned)__builtin_amdgcn_readlane((int)g_off_lane, (i0 + k) & 63);   // padding lanes name row 0
        gv[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, c4, (int)go, 0));
    }
}

// ordered LDS read-modify-writes of entries [i0, i0+kPF) (only those < cntw).  Lanes 0-31 update pixel
// (y, x) with w0, lanes 32-63 pixel (y, x+1) with w1; when the entry has no second pixel the upper half
// is steered to a per-lane dummy slot (address select instead of an exec-mask branch).
__device__ __forceinline__ void pull_rmw(const float (&gv)[kPF], char* acc_bytes, int dummy_off, int lane_off,
                                         const Entry& mine, int i0, int cntw, bool upper)
{
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        const int e = (i0 + k) & 63;
        if (i0 + k < cntw) {                                          // wave-uniform
            const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.w0), e));
            const float w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.w1), e));
            const unsigned p = (unsigned)__builtin_amdgcn_readlane((int)mine.packed, e);
            const int real = (int)(p & 0x7FFFFFFFu) + lane_off;
            const int off = (upper && !(p >> 31)) ? dummy_off : real;
            float* a = reinterpret_cast<float*>(acc_bytes + off);
            *a += (upper ? w1 : w0) * gv[k];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");        // keep the read-modify-writes in program order
    }
}

__global__ __launch_bounds__(64) void msda_bwd_pull_kernel(
    const float* __restrict__ grad_out, const int* __restrict__ bin_start, const int* __restrict__ bin_count,
    const Entry* __restrict__ entries, float* __restrict__ grad_value, Geom g, unsigned grad_out_bytes)
{
    // One single-wave workgroup per bin (tile x pixel-row class): the wave owns (ts/4) rows x ts pixels
    // x 32 channels in LDS and is the only writer, so plain read-modify-write needs no atomics and no
    // barriers; lanes 0-31 / 32-63 hit the two neighbouring pixels of an entry.
    extern __shared__ __attribute__((aligned(16))) float acc[];      // [(ts/4)*ts][32] + 64 dummy floats
    const int bin = blockIdx.x;
    const int tile = bin >> 2, wave = bin & 3;         // pixel rows of this bin: y mod 4 == wave
    const int bm = tile / g.tiles_per_bm;
    int r = tile - bm * g.tiles_per_bm;
    int l = 0;
#pragma unroll 1
    for (int k = 1; k < g.L; ++k) if (r >= g.tile_base[k]) l = k;
    r -= g.tile_base[l];
    const int ts = g.ts[l];
    const int ty0 = (r / g.ntx[l]) * ts, tx0 = (r % g.ntx[l]) * ts;
    const int b = bm / g.M, m = bm % g.M;
    const int lane = threadIdx.x, c = lane & 31, half = lane >> 5;
    const int npix = (ts >> 2) * ts;
    for (int i = lane; i < npix * kD + 64; i += 64) acc[i] = 0.f;
    const int e0 = bin_start[bin], n = bin_count[bin];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(grad_out), 0, grad_out_bytes, 0x00020000);
    const int c4 = c * 4;
    const int lane_off = half * (kD * 4) + c4;         // this lane's byte offset inside a pixel pair
    const int dummy_off = (npix * kD + lane) * 4;
    const bool upper = half != 0;
    char* acc_bytes = reinterpret_cast<char*>(acc);
    Entry nxt;
    nxt.g_off = 0; nxt.w0 = 0.f; nxt.w1 = 0.f; nxt.packed = 0u;
    if (lane < n) nxt = entries[e0 + lane];
    for (int base = 0; base < n; base += 64) {
        const Entry mine = nxt;                        // 64 entries, one per lane; next chunk prefetched
        nxt.g_off = 0; nxt.w0 = 0.f; nxt.w1 = 0.f; nxt.packed = 0u;
        if (base + 64 + lane < n) nxt = entries[e0 + base + 64 + lane];
        const int cntw = min(64, n - base);
        // software pipeline over the 4 groups of 16: the loads of group j+1 fly during the RMWs of group j
        float ga[kPF], gb[kPF];
        pull_load(ga, rsrc, mine.g_off, 0, c4);
        pull_load(gb, rsrc, mine.g_off, 16, c4);
        pull_rmw(ga, acc_bytes, dummy_off, lane_off, mine, 0, cntw, upper);
        if (cntw > 16) {
            pull_load(ga, rsrc, mine.g_off, 32, c4);
            pull_rmw(gb, acc_bytes, dummy_off, lane_off, mine, 16, cntw, upper);
            if (cntw > 32) {
                pull_load(gb, rsrc, mine.g_off, 48, c4);
                pull_rmw(ga, acc_bytes, dummy_off, lane_off, mine, 32, cntw, upper);
                pull_rmw(gb, acc_bytes, dummy_off, lane_off, mine, 48, cntw, upper);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    // write this bin's rows (y = wave, wave+4, ...): 2 pixels x 32 channels per pass, 128-B rows
    const int H = g.H[l], W = g.W[l];
    for (int p = half; p < npix; p += 2) {
        const int y = ty0 + wave + 4 * (p / ts), x = tx0 + p % ts;
        if (y < H && x < W)
            grad_value[(((int64_t)b * g.S + g.start[l] + (int64_t)y * W + x) * g.M + m) * kD + c] = acc[p * kD + c];
    }
}

int g_push_ablate = 0;

bool build_geom(Geom& g, const int64_t* hs, int N, int S, int M, int L, int Lq, int P)
{
    if (L > kMaxL || L * P > 32) return false;
    g.L = L; g.P = P; g.M = M; g.Lq = Lq; g.S = S; g.N = N;
    int64_t wmax = 0, start = 0;
    for (int l = 0; l < L; ++l) wmax = hs[2 * l + 1] > wmax ? hs[2 * l + 1] : wmax;
    int tbase = 0;
    for (int l = 0; l < L; ++l) {
        const int H = (int)hs[2 * l], W = (int)hs[2 * l + 1];
        if (H <= 0 || W <= 0) return false;
        g.H[l] = H; g.W[l] = W; g.start[l] = (int)start;
        start += (int64_t)H * W;
        // tile edge scales with the level so that every tile receives a similar number of samples
        int ratio = (int)((wmax + W / 2) / W);
        int ts = 16;      // finest level: 16x16 pixels per tile, a bin (4 rows x 16 px x 128 B) is 8 KB of LDS
        while (ratio > 1 && ts > 4) { ts >>= 1; ratio >>= 1; }
        g.ts[l] = ts;
        g.ntx[l] = (W + ts - 1) / ts;
        const int nty = (H + ts - 1) / ts;
        g.tile_base[l] = tbase;
        tbase += g.ntx[l] * nty;
    }
    if (start != S) return false;
    g.tiles_per_bm = tbase;
    return true;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t mpf_msda_backward_workspace_bytes(int batch, int num_heads, int num_levels, int num_query,
                                                    int num_point, const int64_t* host_spatial_shapes)
{
    if (!host_spatial_shapes || batch <= 0 || num_heads <= 0 || num_levels <= 0 || num_query <= 0 || num_point <= 0)
        return 0;
    Geom g;
    int64_t S = 0;
    for (int l = 0; l < num_levels; ++l) S += host_spatial_shapes[2 * l] * host_spatial_shapes[2 * l + 1];
    if (!build_geom(g, host_spatial_shapes, batch, (int)S, num_heads, num_levels, num_query, num_point)) return 0;
    const size_t T = (size_t)batch * num_heads * g.tiles_per_bm * 4;   // bins: tile x (pixel row mod 4)
    const size_t max_entries = (size_t)batch * num_query * num_heads * num_levels * num_point * 4;
    const size_t binned = align256((3 * T + 1) * sizeof(int)) + max_entries * sizeof(Entry);
    const size_t blocked = mpf::msda_block_workspace_bytes(host_spatial_shapes, batch, num_heads, num_levels, num_query, num_point);
    return binned > blocked ? binned : blocked;
}

// the amax slots of grad_value / grad_raw by a pass over the data: every route whose kernels do not record them themselves
static int amax_after(const void* grad_value, const void* grad_raw, int N, int S, int M, int D, int L, int Lq, int P, float* graw_amax,
                      float* gv_amax, void* stream)
{
    if (gv_amax)
        if (int r = mpf_amax_f32((const float*)grad_value, (int64_t)N * S * M * D, gv_amax, stream)) return r;
    if (graw_amax && grad_raw)
        if (int r = mpf_amax_f32((const float*)grad_raw, (int64_t)N * Lq * M * L * P * 3, graw_amax, stream)) return r;
    return 0;
}

static int backward_ws_impl(const void* value, const int64_t* host_spatial_shapes,
                            const void* sampling_loc, const void* attn_weight, const void* grad_output,
                            void* grad_value, void* grad_sampling_loc, void* grad_attn_weight, void* grad_raw,
                            int batch, int spatial_size, int num_heads, int channels,
                            int num_levels, int num_query, int num_point, int dtype,
                            void* workspace, size_t workspace_bytes, void* stream, const void* fwd_out = nullptr,
                            float* graw_amax = nullptr, float* gv_amax = nullptr, bool skip_block = false)
{
    const int N = batch, S = spatial_size, M = num_heads, D = channels, L = num_levels, Lq = num_query, P = num_point;
    if (!value || !host_spatial_shapes || !sampling_loc || !attn_weight || !grad_output || !grad_value ||
        (!grad_raw && (!grad_sampling_loc || !grad_attn_weight)) || !workspace)
        return mpf::fail(MPF_E_NULL, "msda_backward_ws: NULL buffer");
    if (dtype != MPF_F32 || D != kD) return mpf::fail(MPF_E_DTYPE, "msda_backward_ws: fp32 with 32 channels per head only");
    if (N <= 0 || S <= 0 || M <= 0 || L <= 0 || Lq <= 0 || P <= 0) return mpf::fail(MPF_E_SHAPE, "msda_backward_ws: bad sizes");
    if (!skip_block) {   // production path: destination-side bin + tile kernels (msda_block.hip); -1000 = not its shapes
        bool recorded = false;
        const int r = mpf::msda_block_backward(value, host_spatial_shapes, sampling_loc, attn_weight, grad_output, grad_value,
                                               grad_sampling_loc, grad_attn_weight, grad_raw, N, S, M, D, L, Lq, P, dtype, workspace,
                                               workspace_bytes, (hipStream_t)stream, fwd_out, graw_amax, gv_amax, &recorded);
        if (r != -1000) return (r != 0 || recorded) ? r : amax_after(grad_value, grad_raw, N, S, M, D, L, Lq, P, graw_amax, gv_amax, stream);
    }
    if (graw_amax || gv_amax) {      // round-1 route: same results, then the two amax passes
        const int r = backward_ws_impl(value, host_spatial_shapes, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                                       grad_attn_weight, grad_raw, batch, spatial_size, num_heads, channels, num_levels, num_query,
                                       num_point, dtype, workspace, workspace_bytes, stream, fwd_out, nullptr, nullptr, true);
        return r ? r : amax_after(grad_value, grad_raw, N, S, M, D, L, Lq, P, graw_amax, gv_amax, stream);
    }
    (void)skip_block;
    Geom g;
    if (!build_geom(g, host_spatial_shapes, N, S, M, L, Lq, P))
        return mpf::fail(MPF_E_SHAPE, "msda_backward_ws: unsupported level geometry (L <= 8, L*P <= 32, sum HW == S)");
    if ((int64_t)N * S * M * D * 4 >= (1ll << 31) || (int64_t)N * Lq * M * L * P * 4 >= (1ll << 31) ||
        (int64_t)N * Lq * M * D * 4 >= (1ll << 32))
        return mpf::fail(MPF_E_TOO_LARGE, "msda_backward_ws: tensor too large for 32-bit indexing");
    const size_t need = mpf_msda_backward_workspace_bytes(N, M, L, Lq, P, host_spatial_shapes);
    if (workspace_bytes < need) return mpf::fail(MPF_E_SHAPE, "msda_backward_ws: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = N * M * g.tiles_per_bm;
    const int T = ntiles * 4;                                     // bins
    int* tile_count = (int*)workspace;
    int* tile_start = tile_count + T;
    int* tile_cursor = tile_start + T;
    int* total = tile_cursor + T;
    Entry* entries = (Entry*)((char*)workspace + align256((3 * (size_t)T + 1) * sizeof(int)));
    hipError_t err = hipMemsetAsync(workspace, 0, (3 * (size_t)T + 1) * sizeof(int), st);
    if (err != hipSuccess) return mpf::check(err, "mpf_msda_backward_ws(memset)");

    const int nchunks = (Lq + kQB - 1) / kQB;
    const int nblocks = N * M * nchunks;
    const int grid = ((nblocks + 7) / 8) * 8;
    const size_t lds_push = (size_t)kQB * L * P * 44 + kSlots * 8;   // descriptors + results + hash table
    const double esz = 4.0;
    const double bytes_push = esz * ((double)N * S * M * D + (double)N * Lq * M * L * P * 6 + (double)N * Lq * M * D);
    mpf::prof_begin(st);
    mpf::set_kernel("msda_bwd_push_kernel");
    hipLaunchKernelGGL(msda_bwd_push_kernel, dim3(grid), dim3(kThreads), lds_push, st,
                       (const float*)value, (const float*)sampling_loc, (const float*)attn_weight,
                       (const float*)grad_output, (float*)grad_sampling_loc, (float*)grad_attn_weight,
                       tile_count, g, nchunks, nblocks, g_push_ablate, (unsigned)((size_t)N * S * M * D * 4), (float*)grad_raw);
    mpf::prof_end("msda_bwd_push_kernel", st, bytes_push);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, tile_count, tile_start, total, T);
    mpf::prof_begin(st);
    hipLaunchKernelGGL(msda_bwd_fill_kernel, dim3(grid), dim3(kThreads), 0, st,
                       (const float*)sampling_loc, (const float*)attn_weight, tile_start, tile_cursor, entries,
                       g, nchunks, nblocks);
    mpf::prof_end("msda_bwd_fill_kernel", st, esz * (double)N * Lq * M * L * P * 3);
    mpf::prof_begin(st);
    mpf::set_kernel("msda_bwd_pull_kernel");
    int ts_max = 4;
    for (int l = 0; l < L; ++l) ts_max = g.ts[l] > ts_max ? g.ts[l] : ts_max;
    hipLaunchKernelGGL(msda_bwd_pull_kernel, dim3(T), dim3(64), (size_t)(ts_max / 4) * ts_max * kD * 4 + 256, st,
                       (const float*)grad_output, tile_start, tile_count, entries, (float*)grad_value, g,
                       (unsigned)((size_t)N * Lq * M * D * 4));
    mpf::prof_end("msda_bwd_pull_kernel", st, esz * ((double)N * Lq * M * D + (double)N * S * M * D));
    mpf::set_kernel("msda_bwd_binned(push+fill+pull)");
    return mpf::check(hipGetLastError(), "mpf_msda_backward_ws");
}

extern "C" int mpf_msda_backward_ws(const void* value, const int64_t* host_spatial_shapes,
                                    const void* sampling_loc, const void* attn_weight, const void* grad_output,
                                    void* grad_value, void* grad_sampling_loc, void* grad_attn_weight,
                                    int batch, int spatial_size, int num_heads, int channels,
                                    int num_levels, int num_query, int num_point, int dtype,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    return backward_ws_impl(value, host_spatial_shapes, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                            grad_attn_weight, nullptr, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point,
                            dtype, workspace, workspace_bytes, stream);
}

extern "C" int mpf_msda_backward_ws_raw(const void* value, const int64_t* host_spatial_shapes,
                                        const void* sampling_loc, const void* attn_weight, const void* grad_output,
                                        void* grad_value, void* grad_raw,
                                        int batch, int spatial_size, int num_heads, int channels,
                                        int num_levels, int num_query, int num_point, int dtype,
                                        void* workspace, size_t workspace_bytes, void* stream)
{
    if (!grad_raw) return mpf::fail(MPF_E_NULL, "msda_backward_ws_raw: NULL grad_raw");
    return backward_ws_impl(value, host_spatial_shapes, sampling_loc, attn_weight, grad_output, grad_value, nullptr, nullptr,
                            grad_raw, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, dtype,
                            workspace, workspace_bytes, stream);
}

// raw form with the forward result at hand (the module keeps it): the softmax-backward sum of a (query, head) is
// <grad_output, output> of that (query, head), so the destination-side kernels (bin + tile, msda_block.hip) apply
extern "C" int mpf_msda_backward_ws_raw_o(const void* value, const int64_t* host_spatial_shapes,
                                          const void* sampling_loc, const void* attn_weight, const void* grad_output,
                                          const void* output, void* grad_value, void* grad_raw,
                                          int batch, int spatial_size, int num_heads, int channels,
                                          int num_levels, int num_query, int num_point, int dtype,
                                          void* workspace, size_t workspace_bytes, float* grad_raw_amax, float* grad_value_amax,
                                          void* stream)
{
    if (!grad_raw || !output) return mpf::fail(MPF_E_NULL, "msda_backward_ws_raw_o: NULL grad_raw / output");
    return backward_ws_impl(value, host_spatial_shapes, sampling_loc, attn_weight, grad_output, grad_value, nullptr, nullptr,
                            grad_raw, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, dtype,
                            workspace, workspace_bytes, stream, output, grad_raw_amax, grad_value_amax);
}

namespace mpf {
int set_binned_option(const char* key, int v)
{
    if (!strcmp(key, "msda_push_ablate")) { g_push_ablate = v; return 0; }
    return 1;
}
}  // namespace mpf
