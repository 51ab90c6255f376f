// GroupNorm of the pixel decoder on channel-last planes (msdeformattn.py:245-281: nn.GroupNorm(32, 256) after
// every 1x1 / 3x3 convolution), forward and backward, without leaving the layout the convolutions produce.
//
// MIOpen's fp32 convolutions hand back [N, H*W, C] planes (channels_last); aten's GroupNorm (and round 1's
// statistics kernel) want NCHW, so every norm was: transpose -> 2 statistics launches -> 4 element-wise launches
// building y = x*a + b, and in the backward aten's NCHW kernels plus the transposes back.  Here a plane stays
// [pixel][channel]:
//   forward   gn_cl_stats (chunked (count, mean, M2), Chan merge: the numerics of the NCHW version) ->
//             gn_merge (shared with elementwise.hip's layout) -> gn_cl_apply: y = x*a + b in one pass, optionally
//             fused with the ReLU that follows the FPN output convolution (msdeformattn.py:268-270) or with the
//             top-down FPN sum y = norm(lateral) + upsample2x(top) (msdeformattn.py:349-351; bilinear,
//             align_corners=False, exact 2x only);
//   backward  gn_cl_bwd_stats (per (image, channel) sums of gy and gy*x over pixel chunks) -> gn_cl_bwd_reduce
//             (chunk sums in fp64, group sums, dgamma / dbeta, the three per-(image, channel) coefficients) ->
//             gn_cl_bwd_apply: dx = A*gy + B*x + D;  upsample2x_cl_bwd gathers the gradient of the FPN top map.
// Every pass reads / writes whole 1 KB pixel rows (C = 256 fp32): HBM-bound, no LDS staging needed.
//
// Thread layout: C/4 lanes cover one pixel (a float4 of channels each), 256/(C/4) pixels in flight per workgroup.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

constexpr int kT = 256;            // threads per workgroup
constexpr int kStatPix = 128;      // pixels per statistics chunk (HW/128 x N workgroups: 1024 at 256x256, batch 2)
constexpr int kApplyPix = 64;      // pixels per apply workgroup
constexpr int kSub = 8;            // pixels a thread holds in registers between Chan merges

struct Plane {
    int HW, C, G, chunks;
    int tpc;         // threads per pixel = C / 4
    int pr;          // pixel rows per workgroup = 256 / tpc
};

__device__ __forceinline__ void chan_merge(float& n, float& mu, float& m2, float nb, float mb, float qb)
{
    if (nb == 0.f) return;
    const float nn = n + nb, d = mb - mu;
    mu += d * (nb / nn);
    m2 += qb + d * d * (n * nb / nn);
    n = nn;
}

__global__ __launch_bounds__(kT) void gn_cl_stats_kernel(const float* __restrict__ x, int64_t x_bs, float* __restrict__ part,
                                                         const Plane P)
{
    __shared__ float s_st[kT * 3];
    const int n = blockIdx.y, ch = blockIdx.x;
    const int c4 = threadIdx.x % P.tpc, prow = threadIdx.x / P.tpc;
    const int p0 = ch * kStatPix, p1 = min(P.HW, p0 + kStatPix);
    const float* xb = x + (int64_t)n * x_bs + c4 * 4;
    float cn = 0.f, mu = 0.f, m2 = 0.f;
    for (int base = p0 + prow; base < p1; base += P.pr * kSub) {
        float4 v[kSub];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
            const int p = base + j * P.pr;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < p1) { v[j] = *reinterpret_cast<const float4*>(xb + (int64_t)p * P.C); ++cnt; }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < kSub; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        const float nb = 4.f * cnt, mb = s / nb;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < kSub; ++j)
            if (j < cnt) {
                const float a = v[j].x - mb, b = v[j].y - mb, c = v[j].z - mb, d = v[j].w - mb;
                q += (a * a + b * b) + (c * c + d * d);
            }
        chan_merge(cn, mu, m2, nb, mb, q);
    }
    s_st[threadIdx.x * 3] = cn; s_st[threadIdx.x * 3 + 1] = mu; s_st[threadIdx.x * 3 + 2] = m2;
    __syncthreads();
    if (threadIdx.x < P.G) {
        const int g = threadIdx.x, tpg = P.tpc / P.G;      // channel-threads per group
        float an = 0.f, am = 0.f, aq = 0.f;
        for (int r = 0; r < P.pr; ++r)
            for (int k = 0; k < tpg; ++k) {
                const int t = r * P.tpc + g * tpg + k;
                chan_merge(an, am, aq, s_st[t * 3], s_st[t * 3 + 1], s_st[t * 3 + 2]);
            }
        float* o = part + (((int64_t)n * P.G + g) * P.chunks + ch) * 3;
        o[0] = an; o[1] = am; o[2] = aq;
    }
}

// one wave per (image, group) row: lanes stride over the chunks, then a butterfly Chan merge (fp64)
__global__ __launch_bounds__(kT) void gn_cl_merge_kernel(const float* __restrict__ part, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int rows, int chunks, float eps)
{
    const int row = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    double n = 0.0, mu = 0.0, m2 = 0.0;
    for (int c = lane; c < chunks; c += 64) {
        const float* o = part + ((int64_t)row * chunks + c) * 3;
        const double nb = o[0], mb = o[1], qb = o[2];
        if (nb == 0.0) continue;
        const double d = mb - mu, nn = n + nb;
        mu += d * nb / nn;
        m2 += qb + d * d * n * nb / nn;
        n = nn;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mu, o, 64), qb = __shfl_xor(m2, o, 64);
        const double nn = n + nb;
        if (nn > 0.0) {
            // symmetric form: both lanes of a pair compute the same merged triple
            const double d = mb - mu;
            const double mu_new = (n * mu + nb * mb) / nn;
            m2 = m2 + qb + d * d * n * nb / nn;
            mu = mu_new;
            n = nn;
        }
    }
    if (lane == 0) {
        mean[row] = (float)mu;
        rstd[row] = (float)(1.0 / sqrt(m2 / n + (double)eps));
    }
}

// bilinear 2x upsampling, align_corners=False (aten area_pixel_compute_source_index with scale 0.5): output row o
// reads rows i0, i1 of the source with weights w0, w1
__device__ __forceinline__ void up2_taps(int o, int in, int& i0, int& i1, float& w0, float& w1)
{
    const float src = fmaxf(0.5f * (o + 0.5f) - 0.5f, 0.f);
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    w1 = src - (float)i0;
    w0 = 1.f - w1;
}

struct ApplyArgs {
    const float* x; int64_t x_bs;
    const float* gamma; const float* beta; const float* mean; const float* rstd;
    const float* top; int64_t top_bs; int W;        // W of the OUTPUT plane (top is W/2 wide)
    float* y; int64_t y_bs;
};

template <bool RELU, bool TOP>
__global__ __launch_bounds__(kT) void gn_cl_apply_kernel(const ApplyArgs A, const Plane P)
{
    const int n = blockIdx.y;
    const int c4 = threadIdx.x % P.tpc, prow = threadIdx.x / P.tpc;
    const int c = c4 * 4, g = c / (P.C / P.G);
    const float rs = A.rstd[n * P.G + g], mu = A.mean[n * P.G + g];
    const float4 gm = *reinterpret_cast<const float4*>(A.gamma + c), bt = *reinterpret_cast<const float4*>(A.beta + c);
    const float4 a = make_float4(rs * gm.x, rs * gm.y, rs * gm.z, rs * gm.w);
    const float4 b = make_float4(bt.x - mu * a.x, bt.y - mu * a.y, bt.z - mu * a.z, bt.w - mu * a.w);
    const float* xb = A.x + (int64_t)n * A.x_bs + c;
    float* yb = A.y + (int64_t)n * A.y_bs + c;
    const float* tb = TOP ? A.top + (int64_t)n * A.top_bs + c : nullptr;
    const int p0 = blockIdx.x * kApplyPix, p1 = min(P.HW, p0 + kApplyPix);
    const int Wt = A.W >> 1, Ht = (P.HW / A.W) >> 1;
#pragma unroll 4
    for (int p = p0 + prow; p < p1; p += P.pr) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)p * P.C);
        float4 r = make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
        if (TOP) {
            const int oy = p / A.W, ox = p - oy * A.W;
            int y0, y1, x0, x1;
            float wy0, wy1, wx0, wx1;
            up2_taps(oy, Ht, y0, y1, wy0, wy1);
            up2_taps(ox, Wt, x0, x1, wx0, wx1);
            const float4 t00 = *reinterpret_cast<const float4*>(tb + ((int64_t)y0 * Wt + x0) * P.C);
            const float4 t01 = *reinterpret_cast<const float4*>(tb + ((int64_t)y0 * Wt + x1) * P.C);
            const float4 t10 = *reinterpret_cast<const float4*>(tb + ((int64_t)y1 * Wt + x0) * P.C);
            const float4 t11 = *reinterpret_cast<const float4*>(tb + ((int64_t)y1 * Wt + x1) * P.C);
            // aten: w0y * (w0x * t00 + w1x * t01) + w1y * (w0x * t10 + w1x * t11)
            r.x += wy0 * (wx0 * t00.x + wx1 * t01.x) + wy1 * (wx0 * t10.x + wx1 * t11.x);
            r.y += wy0 * (wx0 * t00.y + wx1 * t01.y) + wy1 * (wx0 * t10.y + wx1 * t11.y);
            r.z += wy0 * (wx0 * t00.z + wx1 * t01.z) + wy1 * (wx0 * t10.z + wx1 * t11.z);
            r.w += wy0 * (wx0 * t00.w + wx1 * t01.w) + wy1 * (wx0 * t10.w + wx1 * t11.w);
        }
        if (RELU) r = make_float4(fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f));
        *reinterpret_cast<float4*>(yb + (int64_t)p * P.C) = r;
    }
}

struct BwdArgs {
    const float* gy; int64_t gy_bs;
    const float* x; int64_t x_bs;
    const float* gamma; const float* beta; const float* mean; const float* rstd;
    float* part;       // [N][chunks][2][C]
    float* coef;       // [N][3][C]
    float* dx; int64_t dx_bs;
    float* dgamma; float* dbeta;
    int N;
};

// sums over the pixels of one chunk of gy and gy*x per channel (gy gated by the fused ReLU: y = x*a + b > 0)
template <bool RELU>
__global__ __launch_bounds__(kT) void gn_cl_bwd_stats_kernel(const BwdArgs A, const Plane P)
{
    __shared__ float4 s_a[kT], s_b[kT];
    const int n = blockIdx.y, ch = blockIdx.x;
    const int c4 = threadIdx.x % P.tpc, prow = threadIdx.x / P.tpc;
    const int c = c4 * 4, g = c / (P.C / P.G);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (RELU) {
        const float rs = A.rstd[n * P.G + g], mu = A.mean[n * P.G + g];
        const float4 gm = *reinterpret_cast<const float4*>(A.gamma + c), bt = *reinterpret_cast<const float4*>(A.beta + c);
        a = make_float4(rs * gm.x, rs * gm.y, rs * gm.z, rs * gm.w);
        b = make_float4(bt.x - mu * a.x, bt.y - mu * a.y, bt.z - mu * a.z, bt.w - mu * a.w);
    }
    const float* xb = A.x + (int64_t)n * A.x_bs + c;
    const float* gb = A.gy + (int64_t)n * A.gy_bs + c;
    const int p0 = ch * kStatPix, p1 = min(P.HW, p0 + kStatPix);
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sx = sg;
#pragma unroll 4
    for (int p = p0 + prow; p < p1; p += P.pr) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)p * P.C);
        float4 q = *reinterpret_cast<const float4*>(gb + (int64_t)p * P.C);
        if (RELU) {
            q.x = fmaf(v.x, a.x, b.x) > 0.f ? q.x : 0.f;
            q.y = fmaf(v.y, a.y, b.y) > 0.f ? q.y : 0.f;
            q.z = fmaf(v.z, a.z, b.z) > 0.f ? q.z : 0.f;
            q.w = fmaf(v.w, a.w, b.w) > 0.f ? q.w : 0.f;
        }
        sg.x += q.x; sg.y += q.y; sg.z += q.z; sg.w += q.w;
        sx.x = fmaf(q.x, v.x, sx.x); sx.y = fmaf(q.y, v.y, sx.y); sx.z = fmaf(q.z, v.z, sx.z); sx.w = fmaf(q.w, v.w, sx.w);
    }
    s_a[threadIdx.x] = sg; s_b[threadIdx.x] = sx;
    __syncthreads();
    if (prow == 0) {
        for (int r = 1; r < P.pr; ++r) {
            const float4 u = s_a[r * P.tpc + c4], w = s_b[r * P.tpc + c4];
            sg.x += u.x; sg.y += u.y; sg.z += u.z; sg.w += u.w;
            sx.x += w.x; sx.y += w.y; sx.z += w.z; sx.w += w.w;
        }
        float* o = A.part + ((int64_t)n * P.chunks + ch) * 2 * P.C;
        *reinterpret_cast<float4*>(o + c) = sg;
        *reinterpret_cast<float4*>(o + P.C + c) = sx;
    }
}

// workgroup = (image, 32 channels), 1024 threads = 32 channels x 32 chunk slices: chunk sums (fp64) -> group sums -> the
// coefficients of dx = A*gy + B*x + D and this image's share of the parameter gradients (dgb[n][2][C]; summed over the
// images by the first workgroup of the apply pass)
constexpr int kRedCh = 32;
__global__ __launch_bounds__(1024) void gn_cl_bwd_reduce_kernel(const BwdArgs A, const Plane P, float* __restrict__ dgb)
{
    __shared__ double s_1[1024], s_2[1024];
    const int n = blockIdx.y;
    const int cl = threadIdx.x % kRedCh, sl = threadIdx.x / kRedCh;
    const int c = blockIdx.x * kRedCh + cl;
    const int cpg = P.C / P.G, g = c / cpg;
    double sgy = 0.0, sgx = 0.0;
    for (int ch = sl; ch < P.chunks; ch += 1024 / kRedCh) {
        const float* o = A.part + ((int64_t)n * P.chunks + ch) * 2 * P.C;
        sgy += (double)o[c];
        sgx += (double)o[P.C + c];
    }
    s_1[threadIdx.x] = sgy; s_2[threadIdx.x] = sgx;
    __syncthreads();
    // tree over the 32 slices (stride = kRedCh threads)
    for (int h = 512; h >= kRedCh; h >>= 1) {
        if (threadIdx.x < h) { s_1[threadIdx.x] += s_1[threadIdx.x + h]; s_2[threadIdx.x] += s_2[threadIdx.x + h]; }
        __syncthreads();
    }
    if (sl != 0) return;
    sgy = s_1[cl]; sgx = s_2[cl];
    const float gm = A.gamma[c];
    const double mu = A.mean[n * P.G + g], rs = A.rstd[n * P.G + g];
    const double t2 = (sgx - mu * sgy) * rs;       // sum gy * xhat
    // group sums: the cpg channels of a group are adjacent lanes of this half-wave (cpg divides 32)
    double s1 = (double)gm * sgy, s2 = (double)gm * t2;
    for (int o = 1; o < cpg; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const double m = (double)cpg * (double)P.HW;
    // dx = rs * (gm*gy - s1/m - xhat * s2/m),  xhat = (x - mu) * rs
    const double Bc = -rs * rs * s2 / m;
    float* o = A.coef + (int64_t)n * 3 * P.C;
    o[c] = (float)(rs * (double)gm);
    o[P.C + c] = (float)Bc;
    o[2 * P.C + c] = (float)(-rs * s1 / m - Bc * mu);
    dgb[((int64_t)n * 2) * P.C + c] = (float)t2;
    dgb[((int64_t)n * 2 + 1) * P.C + c] = (float)sgy;
}

template <bool RELU>
__global__ __launch_bounds__(kT) void gn_cl_bwd_apply_kernel(const BwdArgs A, const Plane P, const float* __restrict__ dgb)
{
    const int n = blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int c = threadIdx.x; c < P.C; c += kT) {
            float dg = 0.f, db = 0.f;
            for (int i = 0; i < A.N; ++i) { dg += dgb[((int64_t)i * 2) * P.C + c]; db += dgb[((int64_t)i * 2 + 1) * P.C + c]; }
            if (A.dgamma) A.dgamma[c] = dg;
            if (A.dbeta) A.dbeta[c] = db;
        }
    const int c4 = threadIdx.x % P.tpc, prow = threadIdx.x / P.tpc;
    const int c = c4 * 4, g = c / (P.C / P.G);
    const float* co = A.coef + (int64_t)n * 3 * P.C + c;
    const float4 ca = *reinterpret_cast<const float4*>(co), cb = *reinterpret_cast<const float4*>(co + P.C),
                 cd = *reinterpret_cast<const float4*>(co + 2 * P.C);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (RELU) {
        const float rs = A.rstd[n * P.G + g], mu = A.mean[n * P.G + g];
        const float4 gm = *reinterpret_cast<const float4*>(A.gamma + c), bt = *reinterpret_cast<const float4*>(A.beta + c);
        a = make_float4(rs * gm.x, rs * gm.y, rs * gm.z, rs * gm.w);
        b = make_float4(bt.x - mu * a.x, bt.y - mu * a.y, bt.z - mu * a.z, bt.w - mu * a.w);
    }
    const float* xb = A.x + (int64_t)n * A.x_bs + c;
    const float* gb = A.gy + (int64_t)n * A.gy_bs + c;
    float* db = A.dx + (int64_t)n * A.dx_bs + c;
    const int p0 = blockIdx.x * kApplyPix, p1 = min(P.HW, p0 + kApplyPix);
#pragma unroll 4
    for (int p = p0 + prow; p < p1; p += P.pr) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)p * P.C);
        float4 q = *reinterpret_cast<const float4*>(gb + (int64_t)p * P.C);
        if (RELU) {
            q.x = fmaf(v.x, a.x, b.x) > 0.f ? q.x : 0.f;
            q.y = fmaf(v.y, a.y, b.y) > 0.f ? q.y : 0.f;
            q.z = fmaf(v.z, a.z, b.z) > 0.f ? q.z : 0.f;
            q.w = fmaf(v.w, a.w, b.w) > 0.f ? q.w : 0.f;
        }
        float4 r;
        r.x = fmaf(ca.x, q.x, fmaf(cb.x, v.x, cd.x));
        r.y = fmaf(ca.y, q.y, fmaf(cb.y, v.y, cd.y));
        r.z = fmaf(ca.z, q.z, fmaf(cb.z, v.z, cd.z));
        r.w = fmaf(ca.w, q.w, fmaf(cb.w, v.w, cd.w));
        *reinterpret_cast<float4*>(db + (int64_t)p * P.C) = r;
    }
}

// adjoint of the 2x bilinear upsampling: source pixel (ky, kx) gathers from output rows 2ky-1 .. 2ky+2 (weights
// .25 .75 .75 .25; the clamped border rows carry weight 1 instead of .75) and the same along x
__device__ __forceinline__ void up2_adj(int k, int in, int o[4], float w[4])
{
    o[0] = 2 * k - 1; o[1] = 2 * k; o[2] = 2 * k + 1; o[3] = 2 * k + 2;
    w[0] = k > 0 ? 0.25f : 0.f;
    w[1] = k > 0 ? 0.75f : 1.f;
    w[2] = k < in - 1 ? 0.75f : 1.f;
    w[3] = k < in - 1 ? 0.25f : 0.f;
    if (k == 0) o[0] = 0;
    if (k == in - 1) o[3] = 2 * k + 1;
}

__global__ __launch_bounds__(kT) void upsample2x_cl_bwd_kernel(const float* __restrict__ gy, int64_t gy_bs, float* __restrict__ dt,
                                                               int64_t dt_bs, int Ht, int Wt, int C, int tpc, int pr)
{
    const int n = blockIdx.y;
    const int c4 = threadIdx.x % tpc, prow = threadIdx.x / tpc;
    const int p = blockIdx.x * pr + prow;
    if (p >= Ht * Wt) return;
    const int ky = p / Wt, kx = p - ky * Wt;
    int oy[4], ox[4];
    float wy[4], wx[4];
    up2_adj(ky, Ht, oy, wy);
    up2_adj(kx, Wt, ox, wx);
    const float* gb = gy + (int64_t)n * gy_bs + c4 * 4;
    const int W = 2 * Wt;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 q = *reinterpret_cast<const float4*>(gb + ((int64_t)oy[i] * W + ox[j]) * C);
            row.x = fmaf(wx[j], q.x, row.x); row.y = fmaf(wx[j], q.y, row.y);
            row.z = fmaf(wx[j], q.z, row.z); row.w = fmaf(wx[j], q.w, row.w);
        }
        acc.x = fmaf(wy[i], row.x, acc.x); acc.y = fmaf(wy[i], row.y, acc.y);
        acc.z = fmaf(wy[i], row.z, acc.z); acc.w = fmaf(wy[i], row.w, acc.w);
    }
    *reinterpret_cast<float4*>(dt + (int64_t)n * dt_bs + (int64_t)p * C + c4 * 4) = acc;
}

bool make_plane(int HW, int C, int G, Plane& P)
{
    if (HW <= 0 || C <= 0 || G <= 0 || C % G != 0 || C % 4 != 0 || C > 1024) return false;
    const int cpg = C / G, tpc = C / 4;
    if (cpg % 4 != 0 || kT % tpc != 0 || G > kT || C % 32 != 0 || cpg > 32 || (cpg & (cpg - 1)) != 0) return false;
    P.HW = HW; P.C = C; P.G = G; P.tpc = tpc; P.pr = kT / tpc;
    P.chunks = (HW + kStatPix - 1) / kStatPix;
    return true;
}

size_t fwd_ws(int N, const Plane& P) { return (size_t)N * P.G * P.chunks * 3 * sizeof(float); }
size_t bwd_ws(int N, const Plane& P) { return ((size_t)N * P.chunks * 2 * P.C + (size_t)N * 5 * P.C) * sizeof(float); }

}  // namespace

extern "C" int mpf_gn_cl_supported(int HW, int C, int G)
{
    Plane P;
    return make_plane(HW, C, G, P) ? 1 : 0;
}

extern "C" size_t mpf_gn_cl_workspace_bytes(int N, int HW, int C, int G)
{
    Plane P;
    if (N <= 0 || !make_plane(HW, C, G, P)) return 0;
    const size_t a = fwd_ws(N, P), b = bwd_ws(N, P);
    return a > b ? a : b;
}

extern "C" int mpf_gn_cl_forward(const float* x, int64_t x_bs, const float* gamma, const float* beta, int N, int HW, int C, int G,
                                 float eps, int relu, const float* top, int64_t top_bs, int W, float* y, int64_t y_bs, float* mean,
                                 float* rstd, void* workspace, size_t workspace_bytes, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!x || !gamma || !beta || !y || !mean || !rstd || !workspace) return mpf::fail(MPF_E_NULL, "gn_cl_forward: NULL buffer");
    Plane P;
    if (N <= 0 || !make_plane(HW, C, G, P))
        return mpf::fail(MPF_E_SHAPE, "gn_cl_forward: needs C % 32 == 0, C/G a power of two in 4..32, 256 % (C/4) == 0");
    if (x_bs % 4 != 0 || y_bs % 4 != 0 || top_bs % 4 != 0) return mpf::fail(MPF_E_SHAPE, "gn_cl_forward: batch strides must be multiples of 4");
    if (top && (W <= 0 || W % 2 != 0 || HW % W != 0 || (HW / W) % 2 != 0))
        return mpf::fail(MPF_E_SHAPE, "gn_cl_forward: the fused top-down sum needs an exact 2x upsampling (even H and W)");
    if (top && relu) return mpf::fail(MPF_E_SHAPE, "gn_cl_forward: relu and top are exclusive");
    if (workspace_bytes < fwd_ws(N, P)) return mpf::fail(MPF_E_SHAPE, "gn_cl_forward: workspace too small");
    float* part = (float*)workspace;
    mpf::prof_begin(st);
    mpf::set_kernel("gn_cl_stats_kernel");
    hipLaunchKernelGGL(gn_cl_stats_kernel, dim3(P.chunks, N), dim3(kT), 0, st, x, x_bs, part, P);
    mpf::prof_end("gn_cl_stats_kernel", st, 4.0 * (double)N * HW * C);
    hipLaunchKernelGGL(gn_cl_merge_kernel, dim3((N * G + kT / 64 - 1) / (kT / 64)), dim3(kT), 0, st, (const float*)part, mean, rstd, N * G, P.chunks, eps);
    ApplyArgs A{x, x_bs, gamma, beta, mean, rstd, top, top_bs, W, y, y_bs};
    const dim3 grid((HW + kApplyPix - 1) / kApplyPix, N);
    mpf::prof_begin(st);
    mpf::set_kernel("gn_cl_apply_kernel");
    if (top) hipLaunchKernelGGL((gn_cl_apply_kernel<false, true>), grid, dim3(kT), 0, st, A, P);
    else if (relu) hipLaunchKernelGGL((gn_cl_apply_kernel<true, false>), grid, dim3(kT), 0, st, A, P);
    else hipLaunchKernelGGL((gn_cl_apply_kernel<false, false>), grid, dim3(kT), 0, st, A, P);
    mpf::prof_end("gn_cl_apply_kernel", st, (top ? 9.0 : 8.0) * (double)N * HW * C);
    return mpf::check(hipGetLastError(), "mpf_gn_cl_forward");
}

extern "C" int mpf_gn_cl_backward(const float* gy, int64_t gy_bs, const float* x, int64_t x_bs, const float* gamma, const float* beta,
                                  const float* mean, const float* rstd, int N, int HW, int C, int G, int relu, float* dx,
                                  int64_t dx_bs, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!gy || !x || !gamma || !beta || !mean || !rstd || !dx || !workspace) return mpf::fail(MPF_E_NULL, "gn_cl_backward: NULL buffer");
    Plane P;
    if (N <= 0 || !make_plane(HW, C, G, P))
        return mpf::fail(MPF_E_SHAPE, "gn_cl_backward: needs C % 32 == 0, C/G a power of two in 4..32, 256 % (C/4) == 0");
    if (x_bs % 4 != 0 || gy_bs % 4 != 0 || dx_bs % 4 != 0) return mpf::fail(MPF_E_SHAPE, "gn_cl_backward: batch strides must be multiples of 4");
    if (workspace_bytes < bwd_ws(N, P)) return mpf::fail(MPF_E_SHAPE, "gn_cl_backward: workspace too small");
    float* part = (float*)workspace;
    BwdArgs A{gy, gy_bs, x, x_bs, gamma, beta, mean, rstd, part, part + (size_t)N * P.chunks * 2 * C, dx, dx_bs, dgamma, dbeta, N};
    mpf::prof_begin(st);
    mpf::set_kernel("gn_cl_bwd_stats_kernel");
    if (relu) hipLaunchKernelGGL(gn_cl_bwd_stats_kernel<true>, dim3(P.chunks, N), dim3(kT), 0, st, A, P);
    else hipLaunchKernelGGL(gn_cl_bwd_stats_kernel<false>, dim3(P.chunks, N), dim3(kT), 0, st, A, P);
    mpf::prof_end("gn_cl_bwd_stats_kernel", st, 8.0 * (double)N * HW * C);
    float* dgb = A.coef + (size_t)N * 3 * C;
    hipLaunchKernelGGL(gn_cl_bwd_reduce_kernel, dim3(C / kRedCh, N), dim3(1024), 0, st, A, P, dgb);
    const dim3 grid((HW + kApplyPix - 1) / kApplyPix, N);
    mpf::prof_begin(st);
    mpf::set_kernel("gn_cl_bwd_apply_kernel");
    if (relu) hipLaunchKernelGGL(gn_cl_bwd_apply_kernel<true>, grid, dim3(kT), 0, st, A, P, (const float*)dgb);
    else hipLaunchKernelGGL(gn_cl_bwd_apply_kernel<false>, grid, dim3(kT), 0, st, A, P, (const float*)dgb);
    mpf::prof_end("gn_cl_bwd_apply_kernel", st, 12.0 * (double)N * HW * C);
    return mpf::check(hipGetLastError(), "mpf_gn_cl_backward");
}

extern "C" int mpf_upsample2x_cl_backward(const float* gy, int64_t gy_bs, int N, int Ht, int Wt, int C, float* dtop, int64_t dtop_bs,
                                          void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!gy || !dtop) return mpf::fail(MPF_E_NULL, "upsample2x_cl_backward: NULL buffer");
    if (N <= 0 || Ht <= 0 || Wt <= 0 || C <= 0 || C % 4 != 0 || C > 1024 || kT % (C / 4) != 0 || gy_bs % 4 != 0 || dtop_bs % 4 != 0)
        return mpf::fail(MPF_E_SHAPE, "upsample2x_cl_backward: needs C % 4 == 0 and 256 % (C/4) == 0");
    const int tpc = C / 4, pr = kT / tpc;
    mpf::prof_begin(st);
    mpf::set_kernel("upsample2x_cl_bwd_kernel");
    hipLaunchKernelGGL(upsample2x_cl_bwd_kernel, dim3((Ht * Wt + pr - 1) / pr, N), dim3(kT), 0, st, gy, gy_bs, dtop, dtop_bs, Ht, Wt, C,
                       tpc, pr);
    mpf::prof_end("upsample2x_cl_bwd_kernel", st, 20.0 * (double)N * Ht * Wt * C);
    return mpf::check(hipGetLastError(), "mpf_upsample2x_cl_backward");
}
