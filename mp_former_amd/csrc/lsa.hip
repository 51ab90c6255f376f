// Linear sum assignment of the Hungarian matcher on the device — no device->host copy, no host solve.
//
// The reference matcher moves every cost matrix to the host and calls SciPy (matcher.py:149-151:
// `C.cpu()`, `linear_sum_assignment`), which drains the stream once per step (once per image per decoder
// output in the reference) and leaves the GPU idle while the host solves and then issues the loss
// kernels.  This kernel solves all the problems of a step — one wavefront per problem — and writes the
// index arrays the loss kernels consume, so the criterion never synchronises.
//
// Algorithm: SciPy's own (scipy/optimize/rectangular_lsap: Crouse's shortest-augmenting-path variant
// of Jonker-Volgenant, fp64 duals), restated step for step so that the assignment — including how ties
// are broken — is the one SciPy returns:
//   * a tall matrix (more rows than columns; here rows = queries, columns = targets) is solved
//     transposed and reported sorted by row, as SciPy does;
//   * the scan over the remaining columns visits them in SciPy's order (a list filled in reverse and
//     compacted by moving the last entry into the hole); among equal reduced costs it keeps the first
//     one seen unless an unassigned column is seen later — here: the minimum over lanes, then the LAST
//     position holding an unassigned column with that minimum, else the FIRST position with it;
//   * r = ((minVal + c) - u_i) - v_j in fp64, in that order.
// The scan is spread over the 64 lanes (positions it = lane, lane + 64, ...); everything else is as
// sequential as the original.  tests/test_lsa_gpu.py compares with SciPy on random, integer (tie-heavy),
// constant and rectangular matrices.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

constexpr int kMaxDim = 512;            // rows / columns of one problem
constexpr int kCostLds = 10240;         // cost matrices up to this many entries are staged in LDS (fp32)

struct LsaOut {
    int32_t* row_out;
    int32_t* col_out;
    int64_t* aff_a;
    int64_t* aff_b;
    int64_t* scatter_dst;
    const int64_t* scatter_src;
    int32_t* status;          // OR-ed with 1 when a problem has no finite assignment (SciPy raises ValueError there)
};

__device__ __forceinline__ double wave_min_f64(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o));
    return x;
}

__device__ __forceinline__ int wave_min_i32(int x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = min(x, __shfl_xor(x, o));
    return x;
}

__device__ __forceinline__ int wave_max_i32(int x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = max(x, __shfl_xor(x, o));
    return x;
}

template <bool LDS_COST>
__global__ __launch_bounds__(64) void lsa_kernel(const float* __restrict__ cost_all, const MpfLsaProblem* __restrict__ probs,
                                                 const LsaOut out)
{
    __shared__ double u[kMaxDim], v[kMaxDim], spc[kMaxDim];
    __shared__ int col4row[kMaxDim], row4col[kMaxDim], path[kMaxDim], remaining[kMaxDim];
    __shared__ unsigned char SR[kMaxDim], SC[kMaxDim];
    __shared__ float cst[LDS_COST ? kCostLds : 1];
    __shared__ int sh_i, sh_sink, sh_nrem;

    const MpfLsaProblem P = probs[blockIdx.x];
    const int lane = threadIdx.x;
    const int R0 = (int)P.n_rows, C0 = (int)P.n_cols;
    if (R0 <= 0 || C0 <= 0) return;
    const bool tr = C0 < R0;                     // SciPy: "tall rectangular cost matrix must be transposed"
    const int nr = tr ? C0 : R0, nc = tr ? R0 : C0;
    const float* __restrict__ cg = cost_all + P.cost_off;
    // internal cost(i, j): i < nr, j < nc
    const int64_t si = tr ? 1 : P.row_stride, sj = tr ? P.row_stride : 1;
    // SciPy refuses a matrix with NaN or -inf entries up front ("matrix contains invalid numeric entries"): scan it (the
    // LDS staging pass reads every entry anyway) and report through the status word; the solve below still runs and simply
    // never takes such an entry
    bool invalid = false;
    if (LDS_COST) {
        // stage the matrix once (in solver orientation): every row scan then reads LDS instead of L2
        for (int e = lane; e < nr * nc; e += 64) {
            const int i = e / nc, j = e - i * nc;
            const float c = cg[i * si + j * sj];
            invalid |= (c != c) || c == -INFINITY;
            cst[e] = c;
        }
    } else if (out.status) {
        for (int e = lane; e < nr * nc; e += 64) {
            const int i = e / nc, j = e - i * nc;
            const float c = cg[i * si + j * sj];
            invalid |= (c != c) || c == -INFINITY;
        }
    }
    if (invalid && out.status) atomicOr(out.status, 1);
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    __syncthreads();

    for (int cur = 0; cur < nr; ++cur) {
        // ---- augmenting_path ----
        for (int it = lane; it < nc; it += 64) { remaining[it] = nc - it - 1; SC[it] = 0; spc[it] = INFINITY; }
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        if (lane == 0) { sh_i = cur; sh_sink = -1; sh_nrem = nc; }
        __syncthreads();
        double minVal = 0.0;
        while (true) {
            const int i = sh_i, nrem = sh_nrem;
            const double ui = u[i];
            double lowest = INFINITY;
            int first = 0x7fffffff, lastun = -1;
            for (int it = lane; it < nrem; it += 64) {
                const int j = remaining[it];
                const double c = LDS_COST ? (double)cst[i * nc + j] : (double)cg[i * si + j * sj];
                const double r = ((minVal + c) - ui) - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const bool un = row4col[j] == -1;
                if (s < lowest) { lowest = s; first = it; lastun = un ? it : -1; }
                else if (s == lowest && un) lastun = it;
            }
            const double m = wave_min_f64(lowest);
            const bool mine = lowest == m && first != 0x7fffffff;
            const int f = wave_min_i32(mine ? first : 0x7fffffff);
            const int lu = wave_max_i32(mine ? lastun : -1);
            const int index = lu >= 0 ? lu : f;
            minVal = m;
            if (!(m < INFINITY)) {           // infeasible (SciPy raises); leave the rows unassigned
                if (lane == 0) {
                    sh_sink = -2;
                    if (out.status) atomicOr(out.status, 1);
                }
                __syncthreads();
                break;
            }
            __syncthreads();                 // all reads of remaining / row4col done before lane 0 edits them
            if (lane == 0) {
                SR[i] = 1;
                const int j = remaining[index];
                if (row4col[j] == -1) sh_sink = j; else sh_i = row4col[j];
                SC[j] = 1;
                remaining[index] = remaining[nrem - 1];
                sh_nrem = nrem - 1;
            }
            __syncthreads();
            if (sh_sink != -1) break;
        }
        const int sink = sh_sink;
        if (sink < 0) break;
        // ---- update dual variables ----
        for (int i = lane; i < nr; i += 64) {
            if (i == cur) u[i] += minVal;
            else if (SR[i]) u[i] += minVal - spc[col4row[i]];
        }
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc[j];
        __syncthreads();
        // ---- augment previous solution ----
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int i = path[j];
                row4col[j] = i;
                const int t = col4row[i];
                col4row[i] = j;
                j = t;
                if (i == cur) break;
            }
        }
        __syncthreads();
    }

    // ---- report: SciPy order (sorted by row of the matrix as given) ----
    for (int i = lane; i < nr; i += 64) {
        const int c4r = col4row[i];
        if (c4r < 0) continue;
        int row, col, rank;
        if (tr) {
            row = c4r; col = i; rank = 0;
            for (int k = 0; k < nr; ++k) rank += (col4row[k] >= 0 && col4row[k] < c4r) ? 1 : 0;
        } else {
            row = i; col = c4r; rank = i;
        }
        const int64_t slot = P.out_pos + rank;
        if (out.row_out) out.row_out[slot] = row;
        if (out.col_out) out.col_out[slot] = (int32_t)(P.col_base + col);
        if (out.aff_a) out.aff_a[slot] = P.a_base + (int64_t)row * P.a_stride;
        if (out.aff_b) out.aff_b[slot] = P.b_base + (int64_t)row * P.b_stride;
        if (out.scatter_dst) out.scatter_dst[P.scatter_base + row] = out.scatter_src[P.col_base + col];
    }
}

}  // namespace

extern "C" int mpf_lsa_assign_status(const float* cost, const MpfLsaProblem* problems, int n_problems, int max_dim,
                                     int64_t max_entries, int32_t* row_out, int32_t* col_out, int64_t* aff_a, int64_t* aff_b,
                                     int64_t* scatter_dst, const int64_t* scatter_src, int32_t* status, void* stream);

extern "C" int mpf_lsa_assign(const float* cost, const MpfLsaProblem* problems, int n_problems, int max_dim, int64_t max_entries,
                              int32_t* row_out, int32_t* col_out, int64_t* aff_a, int64_t* aff_b, int64_t* scatter_dst,
                              const int64_t* scatter_src, void* stream)
{
    return mpf_lsa_assign_status(cost, problems, n_problems, max_dim, max_entries, row_out, col_out, aff_a, aff_b, scatter_dst,
                                 scatter_src, nullptr, stream);
}

extern "C" int mpf_lsa_assign_status(const float* cost, const MpfLsaProblem* problems, int n_problems, int max_dim,
                                     int64_t max_entries, int32_t* row_out, int32_t* col_out, int64_t* aff_a, int64_t* aff_b,
                                     int64_t* scatter_dst, const int64_t* scatter_src, int32_t* status, void* stream)
{
    if (n_problems == 0) return 0;
    if (!cost || !problems) return mpf::fail(MPF_E_NULL, "lsa_assign: NULL buffer");
    if (n_problems < 0 || max_dim <= 0) return mpf::fail(MPF_E_SHAPE, "lsa_assign: bad sizes");
    if (max_dim > kMaxDim) return mpf::fail(MPF_E_TOO_LARGE, "lsa_assign: a problem has more than 512 rows or columns");
    if (scatter_dst && !scatter_src) return mpf::fail(MPF_E_NULL, "lsa_assign: scatter_dst without scatter_src");
    hipStream_t st = (hipStream_t)stream;
    LsaOut o{row_out, col_out, aff_a, aff_b, scatter_dst, scatter_src, status};
    mpf::prof_begin(st);
    mpf::set_kernel("lsa_kernel");
    if (max_entries <= kCostLds)
        hipLaunchKernelGGL(lsa_kernel<true>, dim3(n_problems), dim3(64), 0, st, cost, problems, o);
    else
        hipLaunchKernelGGL(lsa_kernel<false>, dim3(n_problems), dim3(64), 0, st, cost, problems, o);
    mpf::prof_end("lsa_kernel", st, 0.0);
    return mpf::check(hipGetLastError(), "mpf_lsa_assign");
}
