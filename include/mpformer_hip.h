/*
 * mpformer_hip.h — C ABI of libmpformer_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the native hot path of IDEA-Research/MP-Former.  Every entry point takes
 * plain device pointers and sizes (no torch types), runs asynchronously on the HIP stream passed
 * as `stream` (a hipStream_t cast to void*; NULL = the null stream), never synchronises, never
 * allocates, and returns 0 on success, a positive hipError_t value if the launch failed, or a
 * negative MPF_E_* code if an argument was rejected.  Unlike the reference, which only printf()s
 * kernel-launch errors (ops/src/cuda/ms_deform_im2col_cuda.cuh:953-957,1326-1330), errors are
 * returned to the caller; mpf_last_error() gives a human-readable message for the calling thread.
 *
 * All tensors are contiguous, row-major, in the reference's layouts.
 */
#ifndef MPFORMER_HIP_H
#define MPFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of the floating-point buffers */
enum { MPF_F32 = 0, MPF_F64 = 1, MPF_BF16 = 2, MPF_U8 = 3, MPF_BITS = 4 /* byte masks packed 32 pixels per word */ };

/* argument errors */
enum {
    MPF_OK = 0,
    MPF_E_DTYPE = -1,    /* unsupported dtype code */
    MPF_E_SHAPE = -2,    /* non-positive / inconsistent sizes */
    MPF_E_NULL = -3,     /* NULL pointer for a required buffer */
    MPF_E_TOO_LARGE = -4 /* tensor exceeds the 2^31-element indexing limit of the kernels */
};

/* ABI version of this header (bumped on any signature change). */
int mpf_abi_version(void);
/* Message for the last non-zero return value on the calling thread ("" if none). */
const char* mpf_last_error(void);

/*
 * Multi-scale deformable attention, forward.
 * Replaces ms_deform_attn_forward (pybind: ops/src/vision.cpp:19; dispatcher
 * ops/src/ms_deform_attn.h:25-44; host ops/src/cuda/ms_deform_attn_cuda.cu:25-85; kernel
 * ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304).
 *
 *   value        [batch, spatial_size, num_heads, channels]            dtype
 *   spatial_shapes [num_levels, 2] int64 (H_l, W_l)                     DEVICE memory
 *   level_start_index [num_levels] int64                                DEVICE memory
 *   sampling_loc [batch, num_query, num_heads, num_levels, num_point, 2] dtype, (x, y) in [0,1]
 *   attn_weight  [batch, num_query, num_heads, num_levels, num_point]   dtype
 *   output       [batch, num_query, num_heads*channels]                 dtype, fully overwritten
 *
 * The reference's `im2col_step` batch chunking (ms_deform_attn_cuda.cu:53-80) has no numerical
 * effect and is not part of this ABI: the whole batch is one launch.
 */
int mpf_msda_forward(const void* value, const int64_t* spatial_shapes,
                     const int64_t* level_start_index, const void* sampling_loc,
                     const void* attn_weight, void* output,
                     int batch, int spatial_size, int num_heads, int channels,
                     int num_levels, int num_query, int num_point,
                     int dtype, void* stream);

/*
 * The same operation with spatial_shapes ALSO given from host memory (host_spatial_shapes [num_levels, 2] int64,
 * may be NULL): the production kernel (mp_former_amd/csrc/msda_block.hip) tiles the queries in 2-D blocks of their level
 * and sizes its launch from the level geometry, which a pure device-pointer signature cannot provide without a
 * device->host copy.  Levels must be stored back to back (level_start_index = running sum of H_l*W_l; the caller
 * checks).  With host_spatial_shapes == NULL, or shapes outside fp32 / 32 channels / 4 points / <= 4 levels, this is
 * mpf_msda_forward.
 */
int mpf_msda_forward_hs(const void* value, const int64_t* spatial_shapes,
                        const int64_t* level_start_index, const int64_t* host_spatial_shapes,
                        const void* sampling_loc, const void* attn_weight, void* output,
                        int batch, int spatial_size, int num_heads, int channels,
                        int num_levels, int num_query, int num_point,
                        int dtype, void* stream);

/*
 * Multi-scale deformable attention, backward.
 * Replaces ms_deform_attn_backward (pybind: ops/src/vision.cpp:20; host
 * ops/src/cuda/ms_deform_attn_cuda.cu:88-158; kernels ms_deform_im2col_cuda.cuh:306-925,
 * dispatch :961-1331).
 *
 *   grad_output      [batch, num_query, num_heads*channels]
 *   grad_value       like value         (zero-filled by this call, then accumulated with atomics)
 *   grad_sampling_loc like sampling_loc (fully overwritten)
 *   grad_attn_weight like attn_weight   (fully overwritten)
 */
int mpf_msda_backward(const void* value, const int64_t* spatial_shapes,
                      const int64_t* level_start_index, const void* sampling_loc,
                      const void* attn_weight, const void* grad_output,
                      void* grad_value, void* grad_sampling_loc, void* grad_attn_weight,
                      int batch, int spatial_size, int num_heads, int channels,
                      int num_levels, int num_query, int num_point,
                      int dtype, void* stream);

/*
 * The production kernels behind the reference's UNCHANGED all-device signature (ops/src/ms_deform_attn.h:25-66: spatial_shapes
 * and level_start_index are device tensors; ops/functions/ms_deform_attn_func.py:36,46 passes nothing else).  No device->host
 * copy, no synchronisation: a one-thread prologue kernel derives the launch geometry of the blocked forward / the bin + tile
 * backward (csrc/msda_block.hip) from the two device arrays into the first KB of `workspace`; the main kernels are launched on
 * an upper-bound estimate sized from (batch, spatial_size, num_heads, num_levels, num_query) alone, workgroups beyond the real
 * count return at once, and they stride when an odd pyramid needs more than the estimate.  level_start_index need not be the
 * running sum of H*W (the level ranges must lie inside value and not overlap; rows of grad_value no level owns are zero).
 * Shapes the prologue cannot serve (a side > 16384 or <= 0, a level outside value, overlapping levels) leave every output
 * element NaN — there is no host-visible error without a synchronisation; mpf_msda_dev_geometry reads the record back.
 *   workspace: mpf_msda_dev_workspace_bytes(..., backward) bytes, 256-byte aligned, owned by the caller for the duration of the
 *              kernels (forward: 1 KB; backward: tile counters, entry runs and the spill list, ~60 MB at 1024 x 1024, batch 2).
 * Other dtypes / head widths / point counts / more than 4 levels: mpf_msda_forward / mpf_msda_backward (which read the same two
 * device arrays themselves), also without a copy.
 */
size_t mpf_msda_dev_workspace_bytes(int batch, int spatial_size, int num_heads, int num_levels, int num_query, int num_point,
                                    int backward);
int mpf_msda_forward_dev(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const void* sampling_loc,
                         const void* attn_weight, void* output, int batch, int spatial_size, int num_heads, int channels,
                         int num_levels, int num_query, int num_point, int dtype, void* workspace, size_t workspace_bytes,
                         void* stream);
int mpf_msda_backward_dev(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const void* sampling_loc,
                          const void* attn_weight, const void* grad_output, void* grad_value, void* grad_sampling_loc,
                          void* grad_attn_weight, int batch, int spatial_size, int num_heads, int channels, int num_levels,
                          int num_query, int num_point, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* tests / diagnostics (synchronises): the geometry record a *_dev call left at the start of its workspace.  out[0] = ok,
 * [1] = level_start_index is the running sum, [2] / [3] = workgroups of the query-block kernels / the tile kernel,
 * [4] = entries per (image, head), [5] = tiles per (image, head), [6..9] = run capacity per level */
int mpf_msda_dev_geometry(const void* workspace, int* out, int n);

/*
 * Multi-scale deformable attention, backward, atomics-free ("binned") formulation for fp32 with 32
 * channels per head — the production path of the pixel decoder.  Same results as mpf_msda_backward
 * (grad_value sums are reassociated; no floating-point atomics are used), different contract:
 *   - spatial_shapes is passed from HOST memory (the launch geometry depends on it) and
 *     level_start_index is implied (levels are stored back to back);
 *   - the caller provides a device workspace of at least mpf_msda_backward_workspace_bytes(...);
 *   - grad_value needs no zero-fill: every element is written exactly once.
 * Restrictions: dtype MPF_F32, channels == 32, num_levels <= 8, num_levels*num_point <= 32.
 * See mp_former_amd/csrc/msda_bwd_binned.hip for the algorithm.
 */
size_t mpf_msda_backward_workspace_bytes(int batch, int num_heads, int num_levels, int num_query,
                                         int num_point, const int64_t* host_spatial_shapes);
int mpf_msda_backward_ws(const void* value, const int64_t* host_spatial_shapes,
                         const void* sampling_loc, const void* attn_weight, const void* grad_output,
                         void* grad_value, void* grad_sampling_loc, void* grad_attn_weight,
                         int batch, int spatial_size, int num_heads, int channels,
                         int num_levels, int num_query, int num_point, int dtype,
                         void* workspace, size_t workspace_bytes, void* stream);

/*
 * Tuning / introspection knobs (process-wide; for benchmarks and tests).
 *   mpf_set_option("msda_fwd_variant", v): 0 = auto, 1 = generic kernel, 2 = tiled V4 (8 lanes x 16 B per row), 3 = tiled V1 (32 lanes x 4 B),
 *                                           4 = tiled V2 (16 lanes x 8 B)
 *   mpf_set_option("msda_bwd_variant", v): same numbering
 *   mpf_set_option("gemm3_two_pass", n): mpf_gemm3_tn(_ex) / mpf_gemm3_conv3x3 use the 128 x 256 / 96 x 256 two-pass tile when the
 *                                           output width is a multiple of 256 and >= n (default 256; 0 = the 128 x 128 tiles only;
 *                                           same bits either way); "gemm3_two_pass_rows" 0 / 128 / 96 = rows per tile (0 = by rounds);
 *                                           "gemm3_mixed_tiles" 1 / 0 = 128 x 64 tiles for the last partial round of the one-pass kernel
 * Returns 0, or MPF_E_SHAPE for an unknown key/value.
 */
int mpf_set_option(const char* key, int value);
/* Name of the kernel the most recent native call in this process launched (any thread). */
const char* mpf_last_kernel(void);
/*
 * Tests only: which route the spatially blocked MSDA kernels (mp_former_amd/csrc/msda_block.hip) took since
 * mpf_set_option("msda_stats", 1) / the last reset.  Unlike every other entry point this one SYNCHRONISES the device.
 *   out[0] forward: (workgroup, level) boxes staged in LDS     out[1] forward: L2-gather fallback (box > region)
 *   out[2] push: boxes staged in LDS                           out[3] push: L2-gather fallback
 *   out[4] push: workgroups on direct-mapped tile counters     out[5] push: workgroups on the LDS hash
 *   out[6] spill: entries applied with atomics (run overflow)
 *   out[7] pull: non-empty tiles split over several waves      out[8] pull: non-empty single-wave tiles
 */
int mpf_msda_stats(unsigned long long* out, int n, int reset);

/*
 * Bilinear point sampling: out[i, p] = bilinear(src[rows[i]], coords[coord_rows[i], p]) with zero
 * padding and align_corners=False — detectron2 `point_sample` as called at
 * mask2former/modeling/matcher.py:122-132 (matching cost: one point set per image shared by all of
 * its masks -> coord_rows[i] = image of row i) and mask2former/modeling/criterion.py:164-170
 * (importance sampling of the loss points: coord_rows = NULL, i.e. one point set per row).
 *   src   base pointer of dense [h, w] maps of dtype MPF_F32 / MPF_BF16 / MPF_U8 (bool ground-truth
 *         masks, 0/1 bytes) / MPF_BITS (the same masks packed by mpf_pack_mask_bits); rows [n] int64 =
 *         ELEMENT (pixel, for MPF_BITS) offset of each row's map from `src` (maps of several tensors
 *         can be addressed in one launch from a common base)
 *   coords [C, P, 2] f32 (x, y) in [0,1]            coord_rows [n] int32 or NULL (= identity)
 *   out   [n, P] f32, fully overwritten
 */
int mpf_point_sample(const void* src, int src_dtype, int h, int w, const int64_t* rows,
                     const float* coords, const int32_t* coord_rows, float* out, int n, int P, void* stream);

/*
 * Fused point-sampled mask loss, forward: for every (prediction, target) pair i and point p,
 *   x = bilinear(pred[pred_rows[i]], c), t = bilinear(gt[gt_rows[i]], c),
 * and partial[i, chunk, :] = sums over the chunk's points of
 *   { BCEWithLogits(x, t), sigmoid(x)*t, sigmoid(x), t }
 * from which loss_mask and loss_dice follow (mask2former/modeling/criterion.py:21-65, :172-191:
 * sigmoid_ce_loss = mean_p BCE, dice_loss = 1 - (2*sum(s*t)+1)/(sum(s)+sum(t)+1)).  Replaces the
 * gather of src_masks, the float copy of the GT masks, two point_sample calls and the element-wise
 * loss tensors of the reference.
 *   pred  base pointer, MPF_F32 / MPF_BF16; pred_rows [n] int64 element offsets of the [h,w] maps
 *   gt [Rt, H, W] bytes (0/1) (gt_dtype MPF_U8) or their bit-packed form [Rt, H*W/32] words (MPF_BITS,
 *   H*W % 32 == 0: 8x smaller, L2-resident under the random gathers), gt_rows [n] int32
 *   coords [n, P, 2] f32      partial [n, chunks, 4] f32, fully overwritten
 */
int mpf_mask_loss_forward(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_rows,
                          const void* gt, int gt_dtype, int H, int W, const int32_t* gt_rows,
                          const float* coords, float* partial, int n, int P, int chunks, void* stream);

/* bits[i] bit j = masks[32 i + j] != 0 (n_pixels % 32 == 0): the MPF_BITS form of byte masks */
int mpf_pack_mask_bits(const uint8_t* masks, void* bits, int64_t n_pixels, void* stream);

/*
 * Backward of the above wrt pred: the f32 map at grad_pred + grad_offs[i] (caller zero-fills;
 * accumulated with atomics) += d/dx of  grad_sums[i,0]*sum BCE + grad_sums[i,1]*sum(s*t) +
 * grad_sums[i,2]*sum(s), scattered to the four bilinear corners (autograd of point_sample + the
 * losses, criterion.py:178-187).  grad_sums [n, 4] f32 (column 3 ignored: sum t has no gradient).
 */
int mpf_mask_loss_backward(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_rows,
                           const uint8_t* gt, int H, int W, const int32_t* gt_rows,
                           const float* coords, const float* grad_sums, float* grad_pred,
                           const int64_t* grad_offs, int n, int P, void* stream);

/*
 * Importance selection of the loss points (detectron2 get_uncertain_point_coords_with_randomness
 * with uncertainty = -|logit|, mask2former/modeling/criterion.py:73-87,164-170): for each of the n
 * rows keep the k entries of vals[row, 0:M] with the smallest |value| and write their coordinates to
 * coords_out[row, 0:k] (P_out >= k points per output row; the caller fills [k:P_out] with fresh
 * uniform points).  Replaces torch.topk + gather; ties at the threshold are broken by index.
 */
int mpf_select_uncertain(const float* vals, const float* coords_in, float* coords_out,
                         int n, int M, int k, int P_out, void* stream);

/*
 * mpf_point_sample + mpf_select_uncertain in one launch for bf16 maps whose [h, w] plane fits 128 KiB:
 * coords_out[i, :k] = the k candidate points (of the M in coords_in[i]) with the smallest |logit| of
 * the plane at pred + pred_offs[i], in candidate order (ties at the threshold by index).  The plane is
 * staged in LDS and the logits never reach memory.  M <= 40960.
 */
int mpf_sample_select_uncertain(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_offs,
                                const float* coords_in, float* coords_out, int n, int M, int k, int P_out,
                                void* stream);

/*
 * Matching cost, mask + dice part (mask2former/modeling/matcher.py:15-62,122-148), for n_rows
 * (layer, image, query) prediction maps against the ground-truth masks of their image:
 *   cost[row, t] = w_mask*(sum softplus(x) - sum x*t)/P + w_dice*(1 - (2 sum s*t + 1)/(sum s + sum t + 1))
 * x = prediction sampled at coords[coord_rows[row]] on the fly, t = tsamp[t_first[row] + t, :]
 * (ground-truth masks pre-sampled at the same points with mpf_point_sample), t < t_count[row].
 *   cost [n_rows, Tmax] f32; entries t >= t_count[row] are left untouched.
 * rows_per_group: the caller guarantees that every run of that many consecutive rows (starting at a
 * multiple of it) shares coord_rows / t_first / t_count — the queries of one (output, image) — which
 * lets one workgroup stage the planes in LDS and stream the target samples once per 4 rows; pass 1
 * for no guarantee.
 */
int mpf_match_cost(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_offs,
                   const float* coords, const int32_t* coord_rows, const float* tsamp,
                   const int32_t* t_first, const int32_t* t_count, float* cost,
                   int n_rows, int Tmax, int P, float w_mask, float w_dice, int rows_per_group, void* stream);

/*
 * Attention mask of the next decoder layer, fused (mask2former_transformer_decoder.py:1869-1875 +
 * :1814-1816 + :1780): out[n, q, i] (bytes, 1 = do not attend) =
 *   q <  pad : mp_rows[n, q, i]                       (mask-piloted queries: ground-truth rows)
 *   q >= pad : bilinear_resize(masks[n, q], hl x wl)[i] < 0   (align_corners=False; sigmoid(x) < 0.5)
 * and a row that is entirely 1 is written as all 0 ("attend everywhere").  One mask for all heads.
 *   masks: base pointer of [N, Q, h, w] logits (MPF_F32 / MPF_BF16), element strides stride_n /
 *          stride_q between images / queries, dense [h, w] planes
 *   mp_rows [N, pad, hl*wl] bytes or NULL when pad == 0       out [N, Q, hl*wl] bytes, fully written
 */
int mpf_attn_mask(const void* masks, int dtype, int64_t stride_n, int64_t stride_q, int h, int w,
                  const uint8_t* mp_rows, int pad, uint8_t* out, int N, int Q, int hl, int wl, void* stream);

/*
 * Masked multi-head attention core, forward (bf16 MFMA tiles; head_dim 32):
 *   out[q, n, h*32+d] = sum_k softmax_k(mask(scale * Q[q,n,h,:].K[k,n,h,:]))[k] * V[k,n,h,d]
 * — the softmax(QK^T)V of nn.MultiheadAttention as used by CrossAttentionLayer / SelfAttentionLayer
 * (mask2former_transformer_decoder.py:42-52, :100-112) between the packed in-projection and out_proj.
 *   q  [Lq, N, H*32] bf16      k [Lk, N, H*32] bf16      vt [N, H*32, Lk] bf16 (V transposed)
 *   mask: bytes, 1 = masked (-inf): [N, Lq, Lk] if mask_per_image else [Lq, Lk]; NULL = no mask.
 *         One mask for all heads.  A fully masked row yields zeros (the decoder never produces one).
 *   out [Lq, N, H*32] bf16     lse [N, H, Lq] f32 (log-sum-exp of the scaled scores; may be NULL)
 *   workspace: >= mpf_attn_workspace_bytes(Lq, Lk, N, H) bytes of device memory
 */
size_t mpf_attn_workspace_bytes(int Lq, int Lk, int N, int H);
int mpf_attn_forward(const void* q, const void* k, const void* vt, const uint8_t* mask, int mask_per_image,
                     void* out, float* lse, int Lq, int Lk, int N, int H, int head_dim, float scale,
                     void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward of mpf_attn_forward (autograd of the nn.MultiheadAttention core): dq [Lq,N,E], dk, dv
 * [Lk,N,E] bf16 (fully written) from the upstream gradient dout [Lq,N,E] bf16, the saved lse and
 * delta[n,h,q] = sum_d dout*out.  Besides the key-major q, k, v the kernels read the transposed
 * companions qT, doutT [N, E, LqP] (LqP = Lq rounded up to a multiple of 32, zero padded) and kT
 * [N, E, Lk] (the contraction index of an MFMA has to be contiguous in a lane's fragment).
 */
int mpf_attn_backward(const void* q, const void* k, const void* v, const void* kT, const void* qT,
                      const void* dout, const void* doutT, const uint8_t* mask, int mask_per_image,
                      const float* lse, const float* delta, void* dq, void* dk, void* dv,
                      int Lq, int LqP, int Lk, int N, int H, int head_dim, float scale,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * Layout helpers of the attention kernels (bf16): aT, bT [N, E, LP] = transposes of a, b [L, N, E],
 * zero padded to LP >= L (K^T and V^T for the forward / backward, Q^T and dO^T for the backward), and
 * delta[n, h, q] = sum_d dout[q, n, h*32+d] * out[q, n, h*32+d].
 */
int mpf_attn_transpose2(const void* a, const void* b, void* aT, void* bT, int L, int LP, int N, int E, void* stream);
/*
 * The same entry points for K / V (and dK / dV) that are COLUMN BLOCKS of a wider matrix: the key / value projections of
 * the three decoder layers that attend to one feature level (mask2former_transformer_decoder.py:1784-1789, level =
 * i % 3) are one [S * N, 256] x [256, 768] product, and a layer reads its 256 columns in place.  Strides in elements
 * between sequence positions (row) and between images (img); multiples of 8; 0, 0 = dense [L, N, E].
 */
int mpf_attn_transpose2_strided(const void* a, const void* b, int64_t in_row_stride, int64_t in_img_stride, void* aT, void* bT, int L,
                                int LP, int N, int E, void* stream);
int mpf_attn_forward_kv(const void* q, const void* k, int64_t k_row_stride, int64_t k_img_stride, const void* vt, const uint8_t* mask,
                        int mask_per_image, void* out, float* lse, int Lq, int Lk, int N, int H, int head_dim, float scale,
                        void* workspace, size_t workspace_bytes, void* stream);
int mpf_attn_backward_kv(const void* q, const void* k, const void* v, int64_t kv_row_stride, int64_t kv_img_stride, const void* kT,
                         const void* qT, const void* dout, const void* doutT, const uint8_t* mask, int mask_per_image,
                         const float* lse, const float* delta, void* dq, void* dk, void* dv, int64_t dkv_row_stride,
                         int64_t dkv_img_stride, int Lq, int LqP, int Lk, int N, int H, int head_dim, float scale, void* workspace,
                         size_t workspace_bytes, void* stream);
/*
 * The dK / dV kernel of the backward (keys on the lane axis) reads the mask TRANSPOSED ([N or 1, Lk, LqP] bytes) and
 * (lse, delta) as pairs padded to LqP — its "aux" operands, mpf_attn_bwd_aux_bytes(Lq, Lk, N, H, mask_images) bytes
 * (mask_images = N for a per-image mask, 1 for a shared one, 0 without mask).  mpf_attn_bwd_prep_aux makes them in
 * the launch that also makes qT / doT / delta (mpf_attn_bwd_prep below); mpf_attn_backward_kv_aux consumes them.  The
 * entry points without aux make them themselves (one more launch) in the workspace, which mpf_attn_workspace_bytes
 * sizes for that.
 */
size_t mpf_attn_bwd_aux_bytes(int Lq, int Lk, int N, int H, int mask_images);
int mpf_attn_bwd_prep_aux(const void* q, const void* dout, const void* out, const float* lse, const uint8_t* mask, int mask_per_image,
                          int Lk, void* qT, void* doT, float* delta, void* aux, size_t aux_bytes, int Lq, int LqP, int N, int H,
                          void* stream);
int mpf_attn_backward_kv_aux(const void* q, const void* k, const void* v, int64_t kv_row_stride, int64_t kv_img_stride, const void* kT,
                             const void* qT, const void* dout, const void* doutT, const uint8_t* mask, int mask_per_image,
                             const float* lse, const float* delta, void* dq, void* dk, void* dv, int64_t dkv_row_stride,
                             int64_t dkv_img_stride, int Lq, int LqP, int Lk, int N, int H, int head_dim, float scale, void* workspace,
                             size_t workspace_bytes, const void* aux, void* stream);
int mpf_attn_delta(const void* dout, const void* out, float* delta, int Lq, int N, int H, void* stream);
/* mpf_attn_transpose2(q, dout -> qT, doT; E = 32 * H) and mpf_attn_delta(dout, out) in ONE launch: everything the attention
 * backward derives from the query side (nn.MultiheadAttention backward, mask2former_transformer_decoder.py:42-52, :100-112) */
int mpf_attn_bwd_prep(const void* q, const void* dout, const void* out, void* qT, void* doT, float* delta, int Lq, int LqP, int N,
                      int H, void* stream);

/*
 * Module-level forms of the two ops above (ops/modules/ms_deform_attn.py:103-117 folded in): the
 * sampling locations and attention weights are derived inside the kernels from
 *   raw[N*Lq, M*L*P*3]  = [sampling_offsets output (M,L,P,2) | attention_weights output (M,L*P)] of a row
 *   ref_points[Lq, 2]    = the query's reference point, the same for every level (valid_ratios == 1)
 *   attn = softmax over the L*P logits of (query, head);  loc = ref + offset / (W_l, H_l).
 * Forward: loc_out / attn_out receive them (one preparation launch, then the gather kernel).  Backward (atomics-free
 * formulation, as mpf_msda_backward_ws): grad_raw[N*Lq, M*L*P*3] replaces (grad_loc, grad_attn).
 * fp32, 32 channels per head.
 */
int mpf_msda_forward_raw(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const void* raw, const void* ref_points, void* loc_out, void* attn_out, void* output,
                         int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                         int num_point, int dtype, void* stream);
int mpf_msda_forward_raw_hs(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                            const int64_t* host_spatial_shapes,
                            const void* raw, const void* ref_points, void* loc_out, void* attn_out, void* output,
                            int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                            int num_point, int dtype, void* stream);
int mpf_msda_backward_ws_raw(const void* value, const int64_t* host_spatial_shapes,
                             const void* sampling_loc, const void* attn_weight, const void* grad_output,
                             void* grad_value, void* grad_raw,
                             int batch, int spatial_size, int num_heads, int channels,
                             int num_levels, int num_query, int num_point, int dtype,
                             void* workspace, size_t workspace_bytes, void* stream);
/* the same with `output` = the forward result [N, Lq, M*32] of these inputs (MSDeformAttn keeps it for output_proj's
 * backward, ops/modules/ms_deform_attn.py:117-124): sum_j attn_j * d attn_j of a (query, head) equals
 * <grad_output, output>, which lets every sample's gradient be formed where its value rows are (msda_block.hip, "bin + tile").
 * grad_raw_amax / grad_value_amax (NULL or amax slots, MPF_AMAX_SLOT_FLOATS floats, zeroed by the caller): the largest
 * magnitudes of the two results, recorded by the kernels that write them (one atomic max per workgroup) for the fp16 x 2 GEMMs
 * that consume them — the separate amax passes over the two tensors are not needed */
int mpf_msda_backward_ws_raw_o(const void* value, const int64_t* host_spatial_shapes,
                               const void* sampling_loc, const void* attn_weight, const void* grad_output,
                               const void* output, void* grad_value, void* grad_raw,
                               int batch, int spatial_size, int num_heads, int channels,
                               int num_levels, int num_query, int num_point, int dtype,
                               void* workspace, size_t workspace_bytes, float* grad_raw_amax, float* grad_value_amax, void* stream);

/*
 * fp32 GEMM of the pixel-decoder encoder's Linear layers (reference: nn.Linear inside
 * ops/modules/ms_deform_attn.py:58-62,98-106 and msdeformattn.py:103-131, fp32 because
 * msdeformattn.py:314 disables autocast) on the bf16 matrix cores: every fp32 operand is split
 * error-free into three bf16 pieces and the six products of weight >= 2^-16 are accumulated in fp32
 * (fp32-accurate: the dropped terms are below one fp32 rounding of the product).
 *
 *   C[M,N] = (A[M,K] + A2[m % a2_rows, K]) . B[N,K]^T + bias[N] + Cin[M,N] + Cin2[M,N],
 *   then optional ReLU, then optional gate (C = gate[M,N] > 0 ? C : 0, the ReLU backward)
 *
 * B is passed pre-split: mpf_gemm3_split writes planes[3][rows][cols] bf16 of w[rows, cols]
 * (transpose = 0) or planes[3][cols][rows] of its transpose (transpose = 1, for dX = dY . W).
 * a2, bias, c_in, c_in2, gate may be NULL.  K % 32 == 0; N and all leading dimensions % 4 == 0; a2 rows
 * have stride K.  c_in / c_in2 may alias c.
 */
int mpf_gemm3_split(const float* w, int rows, int cols, int transpose, void* planes, void* stream);
int mpf_gemm3_tn(const float* a, int64_t lda, const float* a2, int a2_rows, const void* b_planes,
                 const float* bias, const float* c_in, int64_t ldcin, const float* c_in2, int64_t ldcin2,
                 const float* gate, int64_t ldgate, float* c, int64_t ldc, int M, int N, int K,
                 int relu, void* stream);

/*
 * Weight-gradient form of the same GEMM (contraction over the rows of two row-major activations):
 *   c_part[s][m][n] = sum_{r in split s} A[r, m] * (B[r, n] + B2[r % b2_rows, n]),   s = r / rows_per_split
 * (c_part[s][n][m] if transpose_out), and optionally the per-split column sums
 *   csum_a[s][m] = sum_r A[r, m],  csum_b[s][n] = sum_r (B + B2)[r, n]
 * (bias gradients; summed per level they give the level_embed gradient).  The caller sums the splits.
 * rows_per_split % 32 == 0; b2_rows >= 32 if b2 is given; Ndim % 4 == 0 unless transpose_out.
 */
int mpf_gemm3_nt(const float* a, int64_t lda, const float* b, int64_t ldb, const float* b2, int64_t ldb2, int b2_rows,
                 float* c_part, float* csum_a, float* csum_b, int R, int Mdim, int Ndim, int rows_per_split,
                 int transpose_out, void* stream);

/*
 * Small-row bf16 GEMM of the transformer decoder's query-side Linear layers (the nn.Linear /
 * nn.MultiheadAttention projections of mask2former_transformer_decoder.py:19-206 under autocast;
 * Qtot * N ~ 200-300 rows), fp32 accumulation:
 *   C[i][j] = sum_k A(i,k) * B(j,k) (+ bias[j]) (+ c_in[i][j]) (ReLU),  i < I, j < J, k < Kc
 * C and c_in (NULL or bf16, may alias c) row-major with row strides ldc / ldcin.
 * A(i,k) = a[i*a_rs + k*a_ks], B(j,k) = b[j*b_rs + k*b_ks] (element strides); for each operand one of
 * the two strides must be 1, so forward (x, W), input gradient (dY, W read along its rows) and weight
 * gradient (dY and x read along their rows) need no transposed copies.  gate (NULL or addressed like
 * a) keeps A(i,k) only where gate(i,k) > 0 (ReLU backward); rowsum_a (NULL or [I] bf16) receives
 * sum_k A(i,k) after the gate (the bias gradient in the weight-gradient form).  All buffers bf16.
 * Contraction-contiguous operands need 16-B aligned rows and Kc % 8 == 0; J % 4 == 0, ldc % 4 == 0.
 */
int mpf_small_gemm_bf16(const void* a, int64_t a_rs, int64_t a_ks, const void* gate, const void* b, int64_t b_rs,
                        int64_t b_ks, const void* bias, const void* c_in, int64_t ldcin, void* c, int64_t ldc,
                        void* rowsum_a, int I, int J, int Kc, int relu, void* stream);

/*
 * 'masked' rows of the mask-piloted queries (prepare_for_dn_v5, mask2former_transformer_decoder.py:
 * 986-987: F.interpolate(masks, size, mode='area') <= 1e-8): out[t, y, x] = 1 iff the byte mask t
 * [H, W] has no non-zero pixel inside block (y, x) of the h x w grid (H % h == 0, W % w == 0).
 */
int mpf_mask_block_empty(const uint8_t* masks, uint8_t* out, int T, int H, int W, int h, int w, void* stream);

/*
 * Decoder inputs of one feature level (mask2former_transformer_decoder.py:1756-1764: input_proj +
 * level_embed, flatten, permute; the key input adds the sine position embedding, :100-112):
 *   src[s, n, c] = x(n, c, s) + level_embed[c]        kin[s, n, c] = src[s, n, c] + pos[s, c]
 * x(n, c, s) = x[n * sx_n + s * sx_s + c] fp32 (unit channel stride: a channel-last view of the encoder
 * memory), pos [S, C] fp32; src / kin [S, N, C] row-contiguous, out_dtype MPF_BF16 (autocast) or MPF_F32.
 * Backward: dx(n, c, s) = g_src[s, n, c] + g_kin[s, n, c] (either may be NULL), written with x's strides.
 */
int mpf_decoder_inputs_forward(const float* x, int64_t sx_n, int64_t sx_s, const float* level_embed, const float* pos,
                               void* src, void* kin, int out_dtype, int S, int N, int C, void* stream);
int mpf_decoder_inputs_backward(const void* g_src, const void* g_kin, int g_dtype, float* dx, int64_t sx_n, int64_t sx_s, int S,
                                int N, int C, void* stream);

/*
 * Fused mask head -> next-layer attention mask (forward_prediction_heads, mask2former_transformer_decoder.py:1859-1877:
 * einsum("bqc,bchw->bqhw") + bilinear resize to the level grid + sigmoid < 0.5, detached), bf16.  The resize commutes
 * with the channel contraction, so the pixel-decoder features are resized once per step and level:
 *   mpf_pool_features: mask_features [N, 256, h, w] (MPF_F32 / MPF_BF16)  ->  out [N, hl*wl, 256] bf16 (pixel-major),
 *                      F.interpolate(mode="bilinear", align_corners=False) semantics;
 *   mpf_pool_features_cl: the same for channel-last features: image n = [h*w][256] at mask_features + n*batch_stride;
 * and each layer only multiplies its mask embeddings with that:
 *   mpf_mask_head_bits: mask_embed bf16, element (n, q, c) at n*stride_n + q*stride_q + c;  pooled [N, HW, 256] bf16;
 *                       out [N, Q, HW] bytes: 1 = do not attend (logit < 0), rows q < pad copied from mp_rows [N, pad, HW]
 *                       (mask-piloted rows, :1814-1816), rows that would be all 1 cleared to 0 (:1780);
 *                       flags [N*Q] int32, zero on entry, zero again on return (scratch for the all-masked rule).
 * The [N, Q, h, w] map of the reference is never formed.  HW must be a multiple of 16.
 */
int mpf_pool_features(const void* mask_features, int dtype, void* out_bf16, int N, int C, int h, int w, int hl, int wl, void* stream);
int mpf_pool_features_cl(const void* mask_features, int64_t batch_stride, int dtype, void* out_bf16, int N, int C, int h, int w,
                         int hl, int wl, void* stream);
int mpf_mask_head_bits(const void* mask_embed, int64_t stride_n, int64_t stride_q, const void* pooled, const uint8_t* mp_rows,
                       int pad, uint8_t* out, int32_t* flags, int N, int Q, int HW, void* stream);

/*
 * Linear sum assignment (the Hungarian step of HungarianMatcher.memory_efficient_forward, matcher.py:
 * 149-151, where the reference calls scipy.optimize.linear_sum_assignment on a host copy of the cost
 * matrix) solved on the device, one wavefront per problem, with SciPy's algorithm and tie-breaking
 * (csrc/lsa.hip) — the matcher needs no device->host copy.  Problem p is the n_rows x n_cols fp32
 * matrix at cost + cost_off (row stride row_stride); its min(n_rows, n_cols) assigned pairs (row, col)
 * are reported in SciPy's order (ascending row) at slots out_pos, out_pos + 1, ...:
 *   row_out[slot] = row           col_out[slot] = col_base + col
 *   aff_a[slot]   = a_base + row * a_stride        aff_b[slot] = b_base + row * b_stride
 *   scatter_dst[scatter_base + row] = scatter_src[col_base + col]
 * (any output pointer may be NULL).  The affine / scatter outputs are the index arrays of the loss
 * kernels: element offsets of the matched prediction planes, and the class target of a matched query.
 * max_dim = max over problems of max(n_rows, n_cols) (<= 512); max_entries = max n_rows * n_cols.
 * An infeasible problem (no finite assignment; SciPy raises) leaves its slots untouched.
 */
typedef struct MpfLsaProblem {
    int64_t cost_off, n_rows, n_cols, row_stride;
    int64_t out_pos, col_base;
    int64_t a_base, a_stride, b_base, b_stride;
    int64_t scatter_base;
} MpfLsaProblem;

int mpf_lsa_assign(const float* cost, const MpfLsaProblem* problems, int n_problems, int max_dim, int64_t max_entries,
                   int32_t* row_out, int32_t* col_out, int64_t* aff_a, int64_t* aff_b, int64_t* scatter_dst,
                   const int64_t* scatter_src, void* stream);
/*
 * The same with a device status word: `status` (may be NULL) is OR-ed with 1 when some problem has no finite assignment
 * (an all-infinite row, NaN costs) — the case in which scipy.optimize.linear_sum_assignment raises ValueError in the
 * reference.  The caller zeroes the word and reads it back when convenient (no synchronisation here).
 */
int mpf_lsa_assign_status(const float* cost, const MpfLsaProblem* problems, int n_problems, int max_dim, int64_t max_entries,
                          int32_t* row_out, int32_t* col_out, int64_t* aff_a, int64_t* aff_b, int64_t* scatter_dst,
                          const int64_t* scatter_src, int32_t* status, void* stream);

/*
 * One decoder layer of the masked-attention transformer decoder — cross-attention, self-attention, FFN,
 * each followed by its post-norm residual (mask2former_transformer_decoder.py:1784-1800 with
 * CrossAttentionLayer.forward_post :100-112, SelfAttentionLayer.forward_post :42-52, FFNLayer.forward_post
 * :165-169), under autocast — issued as ONE host call: the ~20 (forward) / ~45 (backward) kernels of the
 * layer are the entry points above (mpf_small_gemm_bf16, mpf_attn_*, mpf_res_ln256_*), launched back to
 * back on `stream` from native code instead of one Python-level autograd node each.  The key / value
 * projections of the cross-attention (k_c, v_c: [S, N, E] rows, library-sized GEMMs) stay outside.
 *
 * Shapes: R = Qt * N query rows, E = H * 32 = 256 channels, F = ffn_dim.  Activations marked bf16 are
 * the autocast operand copies; the residual stream (x0 .. x3, s1 .. s3) is fp32.  Masks: bytes, 1 =
 * masked; mask_c [N, Qt, S] per image, mask_s [Qt, Qt] shared or NULL.  All buffers are caller-owned
 * device memory; `scratch` must hold mpf_decoder_layer_scratch_bytes(...) bytes (contents undefined
 * afterwards), `attn_ws` max(mpf_attn_workspace_bytes(Qt, S, N, H), mpf_attn_workspace_bytes(Qt, Qt, N, H)).
 */
typedef struct MpfDecoderLayer {
    /* parameters: bf16 GEMM weights [out, in] and biases; fp32 LayerNorm affine */
    const void *ca_wq, *ca_bq, *ca_wo, *ca_bo;
    const float *ca_gamma, *ca_beta;
    const void *sa_wq, *sa_bq, *sa_wk, *sa_bk, *sa_wv, *sa_bv, *sa_wo, *sa_bo;
    const float *sa_gamma, *sa_beta;
    const void *ff_w1, *ff_b1, *ff_w2, *ff_b2;
    const float *ff_gamma, *ff_beta;
    /* forward inputs */
    const float* x0;       /* [R, E] fp32 residual stream */
    const void* xb0;       /* [R, E] bf16 copy of x0 */
    const void *k_c, *v_c; /* [S, N, E] bf16 projected keys / values of the level */
    const uint8_t *mask_c, *mask_s;
    /* written by forward, read by backward */
    void *q_c, *kT_c, *o_c; /* [R, E], [N, E, S], [R, E] bf16 */
    float* lse_c;           /* [N, H, Qt] */
    float *s1, *mean1, *rstd1;
    void* xb1;
    void *q_s, *k_s, *v_s, *kT_s, *o_s; /* [R, E] x3, [N, E, Qt], [R, E] bf16 */
    float* lse_s;
    float *s2, *mean2, *rstd2;
    void* xb2;
    void* h;                /* [R, F] bf16, after the ReLU */
    float *s3, *mean3, *rstd3;
    /* forward outputs */
    float* x3;              /* [R, E] fp32 */
    void* xb3;              /* [R, E] bf16 */
    /* workspaces */
    void *scratch, *attn_ws;
    uint64_t scratch_bytes, attn_ws_bytes;
    int32_t Qt, N, H, S, ffn_dim;
    float eps;
    /* element strides of k_c / v_c between sequence positions and between images; 0, 0 = dense [S, N, E].  A packed
       [S, N, 3E] projection shared by the three layers of a level is (N * 3E, 3E) with k_c pointing at the layer's columns */
    int64_t kv_row_stride, kv_img_stride;
} MpfDecoderLayer;

typedef struct MpfDecoderLayerGrad {
    /* upstream gradients of the outputs (either may be NULL, not both) */
    const float* g_x3;
    const void* g_xb3;
    /* gradients of the inputs */
    float* d_x0;            /* [R, E] fp32 */
    void* d_xb0;            /* [R, E] bf16 */
    void *d_k_c, *d_v_c;    /* [S, N, E] bf16 */
    /* gradients of the parameters: bf16 like the working copies; LayerNorm fp32 as one [6, 256] block
       (ca_gamma, ca_beta, sa_gamma, sa_beta, ff_gamma, ff_beta), zeroed by the call */
    void *d_ca_wq, *d_ca_bq, *d_ca_wo, *d_ca_bo;
    void *d_sa_wq, *d_sa_bq, *d_sa_wk, *d_sa_bk, *d_sa_wv, *d_sa_bv, *d_sa_wo, *d_sa_bo;
    void *d_ff_w1, *d_ff_b1, *d_ff_w2, *d_ff_b2;
    float* d_ln;
    int64_t dkv_row_stride, dkv_img_stride;   /* strides of d_k_c / d_v_c as above (0, 0 = dense) */
    /* a second upstream gradient of x3 (may be NULL; needs g_x3): x3 has two consumers — the next layer and the prediction
       heads (mask2former_transformer_decoder.py:1797-1800) — and their gradients are summed by the layer's first LayerNorm
       backward pass instead of by a separate add */
    const float* g_x3_plus;
} MpfDecoderLayerGrad;

/*
 * Row-local chains of the decoder's query side as ONE kernel each (csrc/row_chain.hip; a workgroup owns 16 rows):
 *   mpf_lin256_res_ln_forward: s = x + bf16(a W^T + b);  y = LayerNorm(s) — the output projection of an attention block with its
 *     post-norm residual (mask2former_transformer_decoder.py:42-52, :100-112); a [rows,256] bf16, W [256,256] bf16 (nn.Linear
 *     layout), b [256] bf16, x fp32; outputs as mpf_res_ln256_forward (s_out, mean, rstd required; y32 / y16 either may be NULL).
 *   mpf_ln256_mlp3_forward: out = W2 relu(W1 relu(W0 LayerNorm(x) + b0) + b1) + b2 in bf16 — decoder_norm followed by the
 *     mask_embed MLP (:1859-1866, :190-206), every intermediate rounded to bf16 as the separate Linear layers store it.
 * Bit-identical to the mpf_small_gemm_bf16 + mpf_res_ln256_forward launches they replace (same contraction order, same row sums).
 */
int mpf_lin256_res_ln_forward(const void* a, const void* w, const void* bias, const float* x, const float* gamma, const float* beta,
                              float* s_out, float* y32, void* y16, float* mean, float* rstd, int rows, float eps, void* stream);
int mpf_ln256_mlp3_forward(const float* x, const float* gamma, const float* beta, const void* w0, const void* b0, const void* w1,
                           const void* b1, const void* w2, const void* b2, void* out, int rows, float eps, void* stream);
uint64_t mpf_decoder_layer_struct_bytes(int which); /* 0: sizeof(MpfDecoderLayer), 1: sizeof(MpfDecoderLayerGrad) */
uint64_t mpf_decoder_layer_scratch_bytes(int Qt, int N, int H, int S, int ffn_dim, int backward);
int mpf_decoder_layer_forward(const MpfDecoderLayer* layer, void* stream);
/*
 * The attention mask the NEXT decoder layer needs, from this layer's residual stream — the detached part of
 * forward_prediction_heads (mask2former_transformer_decoder.py:1859-1875): decoder_norm -> mask_embed MLP -> (product with
 * mask_features resized to the level grid).sigmoid() < 0.5, MP rows (:1814-1816) and the all-masked-row rule (:1780) — as ONE call
 * (mpf_res_ln256_forward, 3 x mpf_small_gemm_bf16, mpf_mask_head_bits issued back to back).  x: fp32 [Q, N, 256] (sequence-first);
 * weights bf16 [256, 256] row-major, biases bf16 [256]; pooled: bf16 [N, HW, 256] (mpf_pool_features); mp_rows: [N, pad, HW] bytes
 * or NULL (pad = 0); out: bool bytes [N, Q, HW]; flags: int32 [N * Q], zero on entry (zeroed again by the kernel); scratch of
 * mpf_next_attn_mask_scratch_bytes(N, Q) bytes, 256-byte aligned.
 */
typedef struct MpfNextMask {
    const float* x;
    const float* ln_gamma;
    const float* ln_beta;
    const void* w0;
    const void* b0;
    const void* w1;
    const void* b1;
    const void* w2;
    const void* b2;
    const void* pooled;
    const uint8_t* mp_rows;
    uint8_t* out;
    int32_t* flags;
    void* scratch;
    size_t scratch_bytes;
    int N, Q, HW, pad;
    float eps;
} MpfNextMask;
size_t mpf_next_attn_mask_scratch_bytes(int N, int Q);
int mpf_next_attn_mask(const MpfNextMask* m, void* stream);
int mpf_decoder_layer_backward(const MpfDecoderLayer* layer, const MpfDecoderLayerGrad* grad, void* stream);

/*
 * dst_i[c, j] = cast(src_i[c, j] * scale_i[c]) for a list of tensors in ONE launch (the FrozenBatchNorm
 * scale folded into every convolution weight of the bench backbone — detectron2 FrozenBatchNorm2d under
 * configs/coco/instance-segmentation/Base-COCO-InstanceSegmentation.yaml:2-15 — and the matching
 * gradient rescale).  items_device: device array; item i covers workgroups [first_block, first_block +
 * ceil(numel / 2048)), first_block ascending from 0; total_blocks = their sum.  inner = elements per
 * channel (numel / C).  (src, dst) dtypes: (MPF_F32, MPF_BF16), (MPF_BF16, MPF_F32), (MPF_F32, MPF_F32).
 */
typedef struct MpfScaleCastItem {
    const void* src;
    void* dst;
    const float* scale;
    int64_t numel, inner, first_block;
} MpfScaleCastItem;

int mpf_grouped_scale_cast(const MpfScaleCastItem* items_device, int n_items, int64_t total_blocks, int src_dtype, int dst_dtype,
                           void* stream);

/*
 * The fp16 x 2 form of the same GEMMs ("_h2"): every fp32 operand x is scaled by a power of two s (chosen from the operand's
 * largest magnitude so that s * amax lies in [2^14, 2^15)) and split into TWO fp16 pieces h = fp16(s x), l = fp16(s x - h),
 * both rounded to nearest (|s x - h - l| <= 2^-23 |s x| while l is a normal fp16 number, i.e. for elements within 2^-18 of the
 * operand's largest; smaller elements keep >= 11 bits and lose one bit per further binade), and a product is the three
 * MFMA products l*h + h*l + h*h in the fp32 accumulator, rescaled by 1 / (s_a s_b) in the epilogue.  Half the matrix-core
 * work of the three-piece bf16 form; measured error against fp64 at or below the library fp32 GEMM's
 * (tests/test_gemm3_gpu.py).  The largest magnitudes live on the DEVICE in "amax slots" of MPF_AMAX_SLOT_FLOATS floats (16
 * sub-slots on different cache lines, so that the one atomic max per producing workgroup does not serialise on one address;
 * the value is the largest of the sub-slots): filled by mpf_amax_f32 or by the out_amax argument of the GEMM that produced
 * the operand, into a slot the caller has zeroed; every kernel derives the same scale from the same slot.
 *   mpf_amax_f32_grouped: item i (device table) = x_i[0 .. numel_i) -> *out_i, workgroups [first_block, + ceil(numel / 4096));
 *   mpf_gemm3_split_grouped_h2: mpf_gemm3_split_grouped writing planes[2][..] of fp16 with the scale from *amax;
 *   mpf_gemm3_tn_h2: mpf_gemm3_tn without the a2 rows; out_amax (may be NULL) receives max |C|.
 */
#define MPF_AMAX_SLOT_FLOATS 512
/* producers of amax slots outside the GEMMs (encoder LayerNorms, msdeformattn.py:123-131):
 *   mpf_res_ln256_forward_b: mpf_res_ln256_forward that also writes an UPPER BOUND of |y| into y_bound (16 max|gamma| +
 *     max|beta|: a normalised row of 256 values cannot exceed sqrt(255)) and of |y + padd| into yplus_bound (that + the value of
 *     the slot padd_amax) — no pass over the data; a bound a few binades above the true maximum costs the split nothing it needs;
 *   mpf_res_ln256_backward_ws_amax: mpf_res_ln256_backward_ws that records max |ds| in ds_amax (one atomic per workgroup). */
int mpf_res_ln256_forward_b(const float* x, const void* t, int t_dtype, const float* gamma, const float* beta, float* s_out, float* y32,
                            void* y16, float* mean, float* rstd, int rows, float eps, const float* padd, int padd_rows, float* y_plus,
                            float* y_bound, const float* padd_amax, float* yplus_bound, void* stream);
int mpf_res_ln256_backward_ws_amax(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                   const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma, float* dbeta, int rows,
                                   void* workspace, size_t workspace_bytes, float* ds_amax, void* stream);
typedef struct MpfAmaxItem {
    const float* src;
    float* out;
    int64_t numel, first_block;
} MpfAmaxItem;
typedef struct MpfSplitItemH2 {
    const float* src;
    void* dst;
    const float* amax;
    int64_t rows, cols, transpose, dst_ld, plane_stride, first_block;
} MpfSplitItemH2;
int mpf_amax_f32(const float* x, int64_t n, float* amax, void* stream);
int mpf_amax_f32_grouped(const MpfAmaxItem* items_device, int n_items, int64_t total_blocks, void* stream);
/* Run-time range guard of the fp16 x 2 form: the guarantee above is relative to the operand's largest magnitude, so rows far
 * below it lose bits.  For operand a [rows, cols] (row stride lda) and the slot its consumer scales by:
 * counters_device[0] += rows with a non-zero element, counters_device[1] += those whose largest magnitude is below
 * 2^-log2_below of the slot (18: where the second piece stops being a normal fp16 number).  One pass over the operand, no
 * synchronisation; the caller reads the two 64-bit counters when it wants them (mp_former_amd/encoder_fused.py: every 64th step). */
int mpf_h2_range_stats(const float* a, int rows, int cols, int64_t lda, const float* amax_slot, int log2_below,
                       unsigned long long* counters_device, void* stream);
int mpf_gemm3_split_grouped_h2(const MpfSplitItemH2* items_device, int n_items, int64_t total_blocks, void* stream);
/* the weight-gradient forms (mpf_gemm3_nt without b2, mpf_gemm3_nt_grouped, mpf_gemm3_conv3x3_wgrad) with both operands split
 * into fp16 pieces on the fly; column sums are taken from the unscaled values; mpf_gemm3_conv3x3_h2: mpf_gemm3_conv3x3 with
 * Cout % 256 == 0 */
typedef struct MpfNtItemH2 {
    const float* a;
    int64_t lda;
    const float* a_amax;
    const float* b;
    int64_t ldb;
    const float* b_amax;
    float* c_part;
    float* csum_a;
    int64_t Mdim, Ndim;
} MpfNtItemH2;
int mpf_gemm3_nt_h2(const float* a, int64_t lda, const float* a_amax, const float* b, int64_t ldb, const float* b_amax,
                    float* c_part, float* csum_a, float* csum_b, int R, int Mdim, int Ndim, int rows_per_split,
                    int transpose_out, void* stream);
int mpf_gemm3_nt_grouped_h2(const MpfNtItemH2* items, int n_items, int R, int rows_per_split, int64_t split_stride, void* stream);
int mpf_gemm3_conv3x3_h2(const float* x, const float* x_amax, const void* w_planes_h2, const float* w_amax, const float* bias,
                         float* y, float* out_amax, int n_img, int H, int W, int Cin, int Cout, int transposed, void* stream);
int mpf_gemm3_conv3x3_wgrad_h2(const float* dy, const float* dy_amax, const float* x, const float* x_amax, float* c_part,
                               float* csum_dy, int n_img, int H, int W, int Cin, int Cout, int rows_per_split, void* stream);
int mpf_gemm3_tn_h2(const float* a, int64_t lda, const float* a_amax, const void* b_planes_h2, const float* b_amax,
                    const float* bias, const float* c_in, int64_t ldcin, const float* c_in2, int64_t ldcin2,
                    const float* gate, int64_t ldgate, float* c, int64_t ldc, float* out_amax, int M, int N, int K,
                    int relu, void* stream);
/* mpf_gemm3_tn_h2 with the ReLU-backward gate as a BIT mask instead of the saved activation (linear1 / linear2 of an encoder
 * layer, msdeformattn.py:123-127: the backward of `activation(linear1(x))` needs the sign of 4 bytes per element only):
 *   gate_bits (may be NULL): [M][ldgbits bytes], bit (n & 7) of byte n / 8 of row m set <=> C[m][n] passes; otherwise 0;
 *   gate_bits_out (may be NULL): receives the mask of (C[m][n] > 0) of this product in the same layout (with relu = 1: the
 *     gate of its own backward).  N % 128 == 0, mask rows >= N / 8 bytes, 8-byte aligned.  Same arithmetic as
 *     mpf_gemm3_tn_h2 with gate = the activation (bit-identical C). */
int mpf_gemm3_tn_h2_bits(const float* a, int64_t lda, const float* a_amax, const void* b_planes_h2, const float* b_amax,
                         const float* bias, const float* c_in, int64_t ldcin, const float* c_in2, int64_t ldcin2,
                         const unsigned char* gate_bits, int64_t ldgbits, float* c, int64_t ldc, float* out_amax,
                         unsigned char* gate_bits_out, int64_t ldgbits_out, int M, int N, int K, int relu, void* stream);

/*
 * Several weight gradients over the SAME rows in one launch (the four Linear layers of an encoder layer whose operands are
 * alive at the end of the layer's backward, msdeformattn.py:103-131): item i is the problem of mpf_gemm3_nt without b2 /
 * csum_b / transpose_out,
 *   c_part_i[s * split_stride + m * Ndim_i + n] = sum_{r in split s} A_i[r, m] * B_i[r, n],   csum_a_i[s * split_stride + m],
 * i.e. the partial results of one split of all items can be laid out back to back (split_stride = their total size) and
 * summed over the splits by ONE mpf_gemm3_nt_reduce over split_stride elements.  Same tiles and the same order of products
 * as mpf_gemm3_nt with the same rows_per_split: bit-identical partial results.  items: HOST array, 1..8 entries;
 * csum_a may be NULL; Ndim % 4 == 0.
 */
typedef struct MpfNtItem {
    const float* a;
    int64_t lda;
    const float* b;
    int64_t ldb;
    float* c_part;
    float* csum_a;
    int64_t Mdim, Ndim;
} MpfNtItem;
int mpf_gemm3_nt_grouped(const MpfNtItem* items, int n_items, int R, int rows_per_split, int64_t split_stride, void* stream);

/*
 * 3x3 convolution (stride 1, zero padding 1, no groups / dilation) of channel-last fp32 images on the split-bf16 GEMM:
 * the FPN output convolution of the pixel decoder (msdeformattn.py:272-281 / :351), forward and input gradient.
 *   x [n_img][H][W][Cin] -> y [n_img][H][W][Cout] (+ bias[Cout]); one GEMM with K = 9*Cin whose A rows are read at the
 *   tap's (dy, dx) shift, taps off the image contributing zeros.
 *   w_planes = mpf_gemm3_split of the [Cout][9*Cin] matrix W2[co][(ky*3+kx)*Cin + ci] = W[co][ci][ky][kx].
 *   transposed != 0: the input gradient dx = conv_transpose(dy, W): call with x := dy (Cin := Cout of the forward),
 *   w_planes = split of W2t[ci][(ky*3+kx)*Cout + co] = W[co][ci][ky][kx]; the taps are then walked with the opposite sign.
 * Cin % 32 == 0, Cout % 4 == 0.
 */
int mpf_gemm3_conv3x3(const float* x, const void* w_planes, const float* bias, float* y, int n_img, int H, int W, int Cin,
                      int Cout, int transposed, void* stream);
/*
 * Its weight gradient (W % 8 == 0, Cin % 128 == 0): c_part[s][co][(ky*3+kx)*Cin + ci] = the contribution of the rows of split
 * s (rows_per_split % 32 == 0) to sum_r dy[r][co] * x[r shifted by the tap][ci]; csum_dy[s][co] (may be NULL) = the
 * column sums of dy (bias gradient).  Sum the splits with mpf_gemm3_nt_reduce.
 */
int mpf_gemm3_conv3x3_wgrad(const float* dy, const float* x, float* c_part, float* csum_dy, int n_img, int H, int W, int Cin, int Cout,
                            int rows_per_split, void* stream);

/*
 * mpf_gemm3_tn with a bf16 A operand and / or a bf16 result (a_dtype / c_dtype = MPF_F32 | MPF_BF16; no second addend / gate /
 * row-periodic addend): a bf16 activation (a backbone feature map under autocast, the bf16 gradient of mask_features) is
 * exactly its own first plane, so it enters the fp32 GEMM without a cast pass and at three products per K step; a bf16
 * result (rounded to nearest even) is what the bf16 producer of that activation expects as its gradient.
 */
int mpf_gemm3_tn_ex(const void* a, int a_dtype, int64_t lda, const void* b_planes, const float* bias, const float* c_in, int64_t ldcin,
                    void* c, int c_dtype, int64_t ldc, int M, int N, int K, int relu, void* stream);
/*
 * ... and the weight-gradient form with ONE bf16 operand (a = dY or b = x in bf16, the other fp32; Ndim % 128 == 0):
 * c_part / csum_a as mpf_gemm3_nt (sum the splits with mpf_gemm3_nt_reduce).
 */
int mpf_gemm3_nt_ex(const void* a, int a_dtype, int64_t lda, const void* b, int b_dtype, int64_t ldb, float* c_part, float* csum_a,
                    int R, int Mdim, int Ndim, int rows_per_split, void* stream);

/*
 * mpf_gemm3_split for a LIST of weight matrices in one launch (all Linear weights of the encoder, both
 * orientations, once per step).  Item i: src fp32 [rows, cols] contiguous -> three bf16 planes at dst,
 * dst + plane_stride, dst + 2 * plane_stride (elements), written as [rows][dst_ld] (transpose = 0) or
 * [cols][dst_ld] (transpose = 1) — so several sources can fill row / column ranges of one operand (the
 * 288-row sampling_offsets | attention_weights matrix).  Item i owns workgroups [first_block, first_block
 * + ceil(rows * cols / 1024)), first_block ascending from 0; total_blocks = their sum.
 */
typedef struct MpfSplitItem {
    const float* src;
    void* dst;
    int64_t rows, cols, transpose, dst_ld, plane_stride, first_block;
} MpfSplitItem;

int mpf_gemm3_split_grouped(const MpfSplitItem* items_device, int n_items, int64_t total_blocks, void* stream);

/*
 * Optimizer tail: full-model gradient-norm clipping + AdamW over ALL parameters in three launches — the
 * reference's FullModelGradientClippingOptimizer around torch.optim.AdamW (train_net.py:259-337,
 * :316-320: clip_grad_norm_(all params, CLIP_VALUE) then AdamW.step()).  Same arithmetic as
 * torch.nn.utils.clip_grad_norm_ (clip = min(1, max_norm / (norm + 1e-6))) followed by torch's AdamW
 * (decoupled weight decay; per-parameter bias corrections bc1 = 1 - beta1^step, bc2_sqrt = sqrt(1 -
 * beta2^step) from the parameter's own step count); the clipped gradients are applied on the fly, not
 * written back.  Item i = one parameter (fp32, dense): workgroups [first_block, first_block +
 * ceil(numel / 2048)), first_block ascending from 0.  partial: >= total_blocks floats of scratch;
 * norm_clip[2] receives {gradient norm, clip coefficient}.  max_norm <= 0 disables clipping.
 */
typedef struct MpfOptItem {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel, first_block;
    float lr, weight_decay, bc1, bc2_sqrt;
} MpfOptItem;

int mpf_clip_adamw_step(const MpfOptItem* items_device, int n_items, int64_t total_blocks, float max_norm, double beta1,
                        double beta2, double eps, float* partial, float* norm_clip, void* stream);

/*
 * out[b, c, r] = in[b, r, c] (fp32, R % 4 == 0, C % 4 == 0; batch strides in elements, the [R, C] / [C, R]
 * matrices themselves dense): the channels-last <-> NCHW relayout of the
 * pixel decoder's convolution outputs around nn.GroupNorm (msdeformattn.py:245-281), LDS-tiled.
 */
int mpf_transpose_f32(const float* in, int64_t in_batch_stride, float* out, int64_t out_batch_stride, int B, int R, int C,
                      void* stream);

/*
 * Sum of the split partials of mpf_gemm3_nt in one launch and a fixed order: c_out[j] = sum_s
 * c_part[s][j] (j < c_numel) and s_out[j] = sum_s s_part[s][j] (j < s_numel; s_numel may be 0).
 */
int mpf_gemm3_nt_reduce(const float* c_part, int64_t c_numel, const float* s_part, int64_t s_numel, int nsplit,
                        float* c_out, float* s_out, void* stream);
/*
 * The same with the column sums also grouped by a key per split (level_of_split[s] in [0, n_levels), n_levels <= 4):
 * lvl_out[l][j] = sum of s_part[s][j] over the splits of level l, s_out[j] = their total.  In the fused encoder the
 * splits of the 288-wide weight gradient never straddle a feature level, so lvl_out is the per-level column sum the
 * level_embed gradient needs (msdeformattn.py:74-77) and s_out the bias gradient.
 */
int mpf_gemm3_nt_reduce_levels(const float* c_part, int64_t c_numel, const float* s_part, int64_t s_numel, int nsplit,
                               const int64_t* level_of_split, int n_levels, float* c_out, float* lvl_out, float* s_out, void* stream);

/*
 * Weight gradient of a bf16 Linear with MANY rows (the key / value projections of the cross-attention,
 * nn.MultiheadAttention in_proj of mask2former_transformer_decoder.py:100-112 under autocast; rows =
 * S * N = 2 048 .. 65 536):  c_out[Mdim, Ndim] = a^T . b,  csum_out[Mdim] = column sums of a  (the bias
 * gradient; may be NULL),  a [R, Mdim] = dY, b [R, Ndim] = x, all bf16, fp32 accumulation.  Split over
 * rows like mpf_gemm3_nt (same kernel, one bf16 product instead of six), partials summed in a fixed
 * order by a second launch.  workspace: mpf_gemm_nt_bf16_workspace_bytes(...) bytes.
 */
/* Forward / input-gradient GEMM of the same layers (and of the batched prediction heads, :1859-1870): c[M, N] (bf16, row
 * stride ldc) = a[M, K] . b[N, K]^T (+ bias[N]), all bf16, fp32 accumulation, one rounding (the library's rounding points);
 * both operands contraction-contiguous (the input gradient passes the transposed weight).  K % 32 == 0; lda, ldb % 8 == 0,
 * ldc % 4 == 0; a, b 16-byte aligned, c 8-byte aligned; M, N arbitrary. */
int mpf_tall_gemm_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const void* bias, void* c, int64_t ldc, int M, int N,
                       int K, void* stream);
size_t mpf_gemm_nt_bf16_workspace_bytes(int R, int Mdim, int Ndim, int rows_per_split);
int mpf_gemm_nt_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* c_out, void* csum_out, int R, int Mdim,
                     int Ndim, int rows_per_split, void* workspace, size_t workspace_bytes, void* stream);

/*
 * mpf_small_gemm_bf16 with BLOCKED operands — three [R, 256] buffers standing for one [R, 768] matrix (the
 * q | k | v projections of nn.MultiheadAttention's packed in_proj, mask2former_transformer_decoder.py:42-52):
 *   a_blk > 0: the 768-long dimension of A is cut into blocks of a_blk, a_bs elements apart — the contraction
 *              index if A is contraction-contiguous (a_ks == 1; Kc % 32 == 0), the row index if it is
 *              row-contiguous (a_rs == 1); gate is addressed like a;
 *   c_blk > 0: output column j goes to c[(j / c_blk) * c_bs + i * ldc + j % c_blk] (c_in must be NULL).
 * a_blk % 32 == 0, c_blk % 64 == 0; 0 = not blocked.
 */
int mpf_small_gemm_bf16_blocked(const void* a, int64_t a_rs, int64_t a_ks, int a_blk, int64_t a_bs, const void* gate, const void* b,
                                int64_t b_rs, int64_t b_ks, const void* bias, const void* c_in, int64_t ldcin, void* c, int64_t ldc,
                                int c_blk, int64_t c_bs, void* rowsum_a, int I, int J, int Kc, int relu, void* stream);

/*
 * The end of SetCriterion (mask2former/modeling/criterion.py) as four launches instead of ~100 tensor-expression launches
 * over vectors of 10-500 elements.
 *
 * Class losses (loss_labels, :123-139) of L outputs at once:
 *   ce[l] = sum_r w[t_r] * (logsumexp(x_r) - x_r[t_r]) / sum_r w[t_r],   r over the N * Q rows of output l
 * logits: MPF_F32 or MPF_BF16, element (l, n, q, c) at l*sl + n*sn + q*sq + c (a strided view of the batched heads' output);
 * target int64 [L, N, Q] (target_per_output = 1) or [N, Q] shared by the outputs (0); weight fp32 [C] (empty_weight);
 * C <= 256.  The forward keeps lse (a buffer of 3 * L * N * Q floats: the rows' log-sum-exp [L, N*Q], then their (w * nll, w)
 * pairs — scratch) and wsum [L] for the backward, which writes the DENSE gradient [L, N, Q, C] in
 * the logits' dtype: grad_ce[l] / wsum[l] * w[t] * (softmax - onehot).  Fixed summation order (reproducible).
 */
int mpf_class_loss_forward(const void* logits, int dtype, int64_t sl, int64_t sn, int64_t sq, const int64_t* target,
                           int target_per_output, const float* weight, int L, int N, int Q, int C, float* lse, float* ce,
                           float* wsum, void* stream);
int mpf_class_loss_backward(const void* logits, int dtype, int64_t sl, int64_t sn, int64_t sq, const int64_t* target,
                            int target_per_output, const float* weight, int L, int N, int Q, int C, const float* lse,
                            const float* wsum, const float* grad_ce, void* dlogits, void* stream);
/*
 * Mask / dice losses (dice_loss :21-40, sigmoid_ce_loss :48-65, / num_masks :189-190) from the per-pair point sums of
 * mpf_mask_loss_forward: sums [n, 4] = (sum BCE, sum sigmoid*t, sum sigmoid, sum t); group g (an output's matched or MP
 * pairs) owns the pairs [runs[2g], runs[2g] + runs[2g+1]); out[g] = sum_i sums[i,0] / points / norm[g],
 * out[G + g] = sum_i (1 - (2 sums[i,1] + 1) / (sums[i,2] + sums[i,3] + 1)) / norm[g].  Backward: dsums [n, 4].
 */
int mpf_mask_loss_finalize(const float* sums, const int64_t* runs, const float* norm, int n, int G, int points, float* out,
                           void* stream);
int mpf_mask_loss_finalize_backward(const float* sums, const int64_t* runs, const float* norm, int n, int G, int points,
                                    const float* grad_out, float* dsums, void* stream);

/*
 * Up to 8 independent WEIGHT-GRADIENT problems of mpf_small_gemm_bf16 in one launch (both operands row-contiguous:
 * a_rs == 1 and b_rs == 1, i.e. dW[J_out, K_in] = dY^T . x with dY and x read along their rows; no bias / c_in / ReLU).
 * The six dW GEMMs of a decoder layer's backward (FFNLayer, SelfAttentionLayer, CrossAttentionLayer of
 * mask2former_transformer_decoder.py:42-52, :100-112, :165-169) are independent of the dX chain and of each other;
 * as separate launches each is mostly launch + memory latency.  Fields as the arguments of
 * mpf_small_gemm_bf16_blocked (a_blk / a_bs: row-blocked A).  Kc % 32 must agree across the items.
 */
typedef struct {
    const void* a;
    const void* gate;
    const void* b;
    void* c;
    void* rowsum_a;
    int64_t a_rs, a_ks, a_bs, b_rs, b_ks, ldc;
    int a_blk, I, J, Kc;
} MpfSmallGemmItem;
int mpf_small_gemm_bf16_group(const MpfSmallGemmItem* items, int n_items, void* stream);
/* Transposed, zero-padded bf16 copies of up to 16 matrices in one launch: src [R, C] (row stride ld elements) -> dst [C, Rp]
 * (Rp >= R, Rp % 8 == 0; the padding is zero); with `gate` (same addressing as src) elements whose gate is <= 0 become zero.
 * What turns the weight-gradient problems of mpf_small_gemm_bf16_group (contraction over the ROWS of dY and x) into its
 * contraction-contiguous form (a_ks == b_ks == 1, row strides Rp, Kc = Rp): 16-byte fragment loads instead of 2-byte ones. */
typedef struct MpfTransposeItem {
    const void* src;
    const void* gate;
    void* dst;
    int64_t ld;
    int R, C;
} MpfTransposeItem;
int mpf_transpose_group_bf16(const MpfTransposeItem* items, int n_items, int Rp, void* stream);

/*
 * y = relu?(x + bias[c] + res) over a dense channel-last activation (channel = fastest dimension,
 * C % 8 == 0): the folded FrozenBatchNorm shift, the residual add and the ReLU of a ResNet
 * bottleneck (detectron2 BottleneckBlock, used by configs/coco/instance-segmentation/Base-COCO-
 * InstanceSegmentation.yaml:2-15) in one pass.  x, res, y: dtype MPF_BF16 or MPF_F32; bias fp32;
 * res may be NULL; y may alias x.  Rounding points of the unfused bf16 ops are kept.
 */
int mpf_bias_act(const void* x, const float* bias, const void* res, void* y, int64_t numel, int C, int dtype,
                 int relu, void* stream);
/* Backward of the block-output ReLU of a residual block (detectron2 BottleneckBlock: out = relu(conv3 + shortcut)) with the sum
 * of the two gradients that reach the block's output folded in: out = y > 0 ? round_bf16(ga + gb) : 0 (gb may be NULL) — aten's
 * add + threshold_backward in one pass.  bf16, numel % 8 == 0, 16-byte aligned, same (dense) layout for all four tensors. */
int mpf_relu_bwd_add(const void* ga, const void* gb, const void* y, void* out, int64_t numel, int dtype, void* stream);
/* 3 x 3 / stride 2 / padding 1 max pooling of the ResNet stem (detectron2 BasicStem) on a channel-last bf16 activation x [N, H, W, C]
 * -> y [N, OH, OW, C] (OH = (H - 1) / 2 + 1) with torch.nn.functional.max_pool2d's tie rule (first maximum in window scan order);
 * code [N, OH, OW, C] bytes = the winner's window position, consumed by the backward (a gather: gx fully written). */
int mpf_maxpool3x3s2_forward(const void* x, void* y, void* code, int N, int H, int W, int C, void* stream);
int mpf_maxpool3x3s2_backward(const void* gy, const void* code, void* gx, int N, int H, int W, int C, void* stream);

/*
 * Small host -> device table (item lists of the grouped launches) through the kernel-argument segment: `nbytes` (multiple
 * of 4) from host memory to device memory on `stream`, 3 968 bytes per launch, no staging buffer — capturable in a HIP
 * graph (the node carries the bytes), and cheaper for the launch thread than a pinned-memory copy.  (No reference
 * counterpart: the reference builds such tensors with torch.tensor(..., device=...), a synchronising copy.)
 */
int mpf_upload_small(const void* host_src, void* device_dst, int64_t nbytes, void* stream);

/*
 * Post-norm residual block of the decoder layers (mask2former_transformer_decoder.py:42-52, :100-112,
 * :165-169: tgt = LayerNorm(tgt + tgt2)) for 256 channels, one pass:
 *   s = x + t;  y = (s - mean) * rstd * gamma + beta      x fp32 [rows,256]; t fp32/bf16 or NULL
 * y is written as fp32 (y32, the next residual) and/or bf16 (y16, the next GEMM operand); s_out
 * (optional), mean, rstd [rows] are what the backward needs.  y_plus (optional) = y + padd[row %
 * padd_rows]: the "src + pos" query input of the next encoder layer (msdeformattn.py:124).
 * Backward: g = gy32 + gy16 + gy_plus (any may be NULL) -> ds32 / ds16 (same values, fp32 / bf16;
 * either may be NULL) and dgamma / dbeta ACCUMULATED with float atomics (zero them before the first call).
 */
int mpf_res_ln256_forward(const float* x, const void* t, int t_dtype, const float* gamma, const float* beta,
                          float* s_out, float* y32, void* y16, float* mean, float* rstd, int rows, float eps,
                          const float* padd, int padd_rows, float* y_plus, void* stream);
int mpf_res_ln256_backward(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                           const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma, float* dbeta,
                           int rows, void* stream);
/* The same with dgamma / dbeta reduced through per-workgroup partials in a fixed order (no float atomics: deterministic, and
 * the two vectors need no zero-initialisation).  workspace: mpf_res_ln256_backward_workspace_bytes(rows) bytes. */
/* The two halves of mpf_res_ln256_backward_ws as separate calls (several LayerNorm backwards, ONE reduce launch): `partials`
 * of LayerNorm z at partials_base + z * stride_bytes, each mpf_res_ln256_backward_workspace_bytes(rows) bytes; out[z][2][256] =
 * (dgamma, dbeta) of LayerNorm z, summed in a fixed order. */
int mpf_res_ln256_backward_partial(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                   const void* gy16, const float* gy_plus, float* ds32, void* ds16, int rows, void* partials,
                                   size_t partials_bytes, void* stream);
int mpf_res_ln256_backward_partial_amax(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                                        const void* gy16, const float* gy_plus, float* ds32, void* ds16, int rows, void* partials,
                                        size_t partials_bytes, float* ds_amax, void* stream);
int mpf_ln_partial_reduce(const void* partials, size_t stride_bytes, int rows, int n_ln, float* out, void* stream);
/* One-launch deterministic form (the workgroup that arrives last sums the per-workgroup partials in a fixed order):
 * dgamma_dbeta [2][256] = (dgamma, dbeta), fully written.  The first 4 bytes of `workspace`
 * (mpf_res_ln256_backward_det_workspace_bytes(rows) bytes) must be zero on entry and are zero again on exit. */
size_t mpf_res_ln256_backward_det_workspace_bytes(int rows);
int mpf_res_ln256_backward_det(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                               const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma_dbeta, int rows,
                               void* workspace, size_t workspace_bytes, void* stream);
size_t mpf_res_ln256_backward_workspace_bytes(int rows);
int mpf_res_ln256_backward_ws(const float* s, const float* mean, const float* rstd, const float* gamma, const float* gy32,
                              const void* gy16, const float* gy_plus, float* ds32, void* ds16, float* dgamma, float* dbeta, int rows,
                              void* workspace, size_t workspace_bytes, void* stream);

/*
 * Same gradient as mpf_mask_loss_backward, written WITHOUT global atomics and without an fp32 image:
 * grad[grad_offs[i] + p] (element offsets into a gradient of the maps' own dtype) receives the whole
 * [h, w] plane of pair i (zeros where no point fell).  Every pair must name a distinct plane; planes
 * of maps that have no pair are left untouched (zero them beforehand).  Accumulation is fixed-point in
 * LDS (order-independent, deterministic; resolution 2^-23 of the per-pair gradient bound).
 */
int mpf_mask_loss_backward_dense(const void* pred, int pred_dtype, int h, int w, const int64_t* pred_rows,
                                 const void* gt, int gt_dtype, int H, int W, const int32_t* gt_rows, const float* coords,
                                 const float* grad_sums, void* grad, int grad_dtype, const int64_t* grad_offs,
                                 int n, int P, void* stream);

/*
 * GroupNorm statistics of the pixel decoder (nn.GroupNorm(32, conv_dim), msdeformattn.py:245-281):
 * mean[r], rstd[r] = 1/sqrt(var + eps) (biased variance) of `rows` = N*G contiguous runs of row_len =
 * (C/G)*H*W floats (NCHW).  Rows are cut into 8192-element chunks (count, mean, M2 per chunk, merged
 * with Chan's formula) so that 64 rows still fill the chip.  row_len % 4 == 0.
 */
size_t mpf_group_stats_workspace_bytes(int rows, int64_t row_len);
int mpf_group_stats(const float* x, int rows, int64_t row_len, float eps, float* mean, float* rstd,
                    void* workspace, size_t workspace_bytes, void* stream);

/*
 * GroupNorm of the pixel decoder on channel-last planes (the layout the fp32 convolutions produce), forward and
 * backward (replaces nn.GroupNorm(32, conv_dim) of msdeformattn.py:245-281 and, optionally, what follows it).
 * A plane is x[n] = [HW][C] floats at x + n * x_bs (dense pixel rows; the batch stride is free, in elements).
 * Needs C % 32 == 0, C/G a power of two in 4..32, 256 % (C/4) == 0 (mpf_gn_cl_supported).
 *   forward:  mean / rstd [N*G] (saved for the backward) and y = (x - mean) * rstd * gamma + beta, then
 *             relu != 0: y = max(y, 0)                 (detectron2 Conv2d(norm=GN, activation=relu), msdeformattn.py:268)
 *             top != NULL: y += bilinear 2x upsampling (align_corners=False) of top[n] = [HW/4][C] at
 *             top + n * top_bs; W = width of the OUTPUT plane   (the FPN top-down sum, msdeformattn.py:349-351)
 *   backward: dx, dgamma[C], dbeta[C] (either may be NULL) from gy; with relu != 0 gy is gated by y > 0 (recomputed).
 *             The gradient of `top` is mpf_upsample2x_cl_backward(gy): dtop[n] = [Ht*Wt][C].
 * Statistics: (count, mean, M2) per 128-pixel chunk merged with Chan's formula (fp64 across chunks).
 */
int mpf_gn_cl_supported(int HW, int C, int G);
size_t mpf_gn_cl_workspace_bytes(int N, int HW, int C, int G);
int mpf_gn_cl_forward(const float* x, int64_t x_bs, const float* gamma, const float* beta, int N, int HW, int C, int G,
                      float eps, int relu, const float* top, int64_t top_bs, int W, float* y, int64_t y_bs, float* mean,
                      float* rstd, void* workspace, size_t workspace_bytes, void* stream);
int mpf_gn_cl_backward(const float* gy, int64_t gy_bs, const float* x, int64_t x_bs, const float* gamma, const float* beta,
                       const float* mean, const float* rstd, int N, int HW, int C, int G, int relu, float* dx,
                       int64_t dx_bs, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
int mpf_upsample2x_cl_backward(const float* gy, int64_t gy_bs, int N, int Ht, int Wt, int C, float* dtop, int64_t dtop_bs,
                               void* stream);

/*
 * Fused mask product + point-sampled matching cost / loss planes (mp_former_amd/csrc/mask_fused.hip).
 * Replaces   outputs_mask = einsum("bqc,bchw->bqhw", mask_embed, mask_features)   (mask2former_transformer_decoder.py:1865-1870)
 * together with its only consumers on the training path: the matcher's point samples (matcher.py:95-147) and the
 * criterion's per-pair planes (criterion.py:141-191).  The [N, Qtot, H/4, W/4] maps are never formed.
 *   embed     bf16 mask embeddings, any row addressing: a row is 256 contiguous bf16 at `embed + offset`
 *   features  bf16 channel-last mask features [N][h*w][256] (image stride feat_img_stride elements)
 *
 * mpf_match_cost_fused: cost[g][q][t] (fp32, [G][Q][Tmax]; entries t >= t_count[g] untouched) of matcher.py:105-147
 *   (mask + dice part) for G groups = (decoder output, image): queries q = 0..Q-1 of group g are the rows
 *   embed + embed_first[g] + q * embed_row_stride; coords [G][P][2] (x, y) in [0,1]; tsamp [tsamp_rows][P] = the
 *   ground-truth masks sampled at the group's points, rows t_first[g] .. + t_count[g].  Tmax <= 128.  Deterministic.
 * mpf_pair_planes_forward: out[slot][h*w] (bf16) = embed row of slot . features of its image, for the slots
 *   slot_first[b] .. + slot_count[b] of image b; the row of a slot is embed + row_off[pair_of_slot ? pair_of_slot[slot] : slot].
 *   All index arrays are DEVICE memory (row_off is written by the device assignment solver).  h*w % 4 == 0.
 * mpf_pair_planes_backward: from grad_planes [total_slots][h*w] bf16:
 *   d_features[b][px][c] = sum_slot grad[slot][px] * embed_row(slot)[c]  (bf16, fully written; may be NULL) and
 *   d_embed (same row addressing as embed; dtype MPF_BF16 / MPF_F32; only paired rows are written; may be NULL)
 *   = sum_px grad[slot][px] * features[b][px][:]  (pixel-range partial sums reduced in a fixed order).  h*w % 128 == 0.
 *   PRECONDITION: the embedding rows of the valid slots are distinct (a query is matched at most once per output:
 *   matcher.py:149-156) — the d_embed row of a slot is WRITTEN, not accumulated; a row listed twice keeps the last write.
 */
size_t mpf_match_cost_fused_workspace_bytes(int G, int Q, int Tmax, int P, int tsamp_rows);
int mpf_match_cost_fused(const void* embed, const int64_t* embed_first, int64_t embed_row_stride, const void* features,
                         int64_t feat_img_stride, const int32_t* group_image, int h, int w, int channels, const float* coords,
                         const float* tsamp, int tsamp_rows, const int32_t* t_first, const int32_t* t_count, float* cost, int G,
                         int Q, int Tmax, int P, float w_mask, float w_dice, void* workspace, size_t workspace_bytes, void* stream);
int mpf_pair_planes_forward(const void* embed, const int64_t* row_off, const int32_t* pair_of_slot, const int32_t* slot_first,
                            const int32_t* slot_count, const void* features, int64_t feat_img_stride, void* out, int N, int HW,
                            int channels, void* stream);
size_t mpf_pair_planes_backward_workspace_bytes(int N, int HW, int total_slots, int max_count);
int mpf_pair_planes_backward(const void* grad_planes, const void* embed, const int64_t* row_off, const int32_t* pair_of_slot,
                             const int32_t* slot_first, const int32_t* slot_count, const void* features, int64_t feat_img_stride,
                             void* d_features, int64_t dfeat_img_stride, void* d_embed, int d_embed_dtype, int N, int HW, int channels,
                             int total_slots, int max_count, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Launch profiler.  While enabled, every kernel launch of this library is bracketed by two HIP
 * events recorded on the launch stream (kernel only: memsets and host work are outside the
 * bracket).  mpf_profile_enable(on) clears the log.  mpf_profile_get() waits for the logged events
 * and returns, for all launches whose kernel name contains `name_substr`, their number, summed
 * duration in milliseconds and summed ALGORITHMIC bytes (the figures of DESIGN.md).
 */
int mpf_profile_enable(int on);
int mpf_profile_get(const char* name_substr, int* count, double* total_ms, double* total_bytes);
/* summed ALGORITHMIC floating-point operations of the same launches (kernels that are priced against the MFMA peak) */
int mpf_profile_get_flops(const char* name_substr, double* total_flops);

/*
 * The forward of the pixel decoder's encoder layers (msdeformattn.py:92-161 x num_layers; ops/modules/ms_deform_attn.py:82-125
 * inside; dropout inactive, fp32, 8 heads x 32 channels) as ONE call: per layer the fixed sequence value_proj -> 288-wide
 * offsets | weights projection -> mpf_msda_forward_raw_hs -> output_proj + residual -> norm1 -> linear1 + ReLU (+ bit mask) ->
 * linear2 + residual -> norm2 (+ q of the next layer), on the entry points above with the arguments the python glue
 * (mp_former_amd/encoder_fused.py) used to pass one by one.  `layers` is a HOST table [n_layers][MPF_ENC_FIELDS] of device
 * addresses: weight planes / amax slots (mpf_gemm3_split_grouped_h2), fp32 vectors, the tensors the layer writes (all kept
 * for the backward) and its amax slots (zeroed by the caller).  Never allocates, never synchronises.
 */
enum {
    MPF_ENC_PV, MPF_ENC_PV_AM, MPF_ENC_PO, MPF_ENC_PO_AM, MPF_ENC_P1, MPF_ENC_P1_AM, MPF_ENC_P2, MPF_ENC_P2_AM, MPF_ENC_P288, MPF_ENC_P288_AM,
    MPF_ENC_BV, MPF_ENC_BO, MPF_ENC_BB1, MPF_ENC_BB2, MPF_ENC_B288, MPF_ENC_G1, MPF_ENC_B1, MPF_ENC_G2, MPF_ENC_B2,
    MPF_ENC_VALUE, MPF_ENC_RAW, MPF_ENC_LOC, MPF_ENC_ATTN, MPF_ENC_AO, MPF_ENC_S1, MPF_ENC_MEAN1, MPF_ENC_RSTD1, MPF_ENC_X1, MPF_ENC_H,
    MPF_ENC_HBITS, MPF_ENC_S2, MPF_ENC_MEAN2, MPF_ENC_RSTD2, MPF_ENC_X2, MPF_ENC_QN,
    MPF_ENC_AO_AM, MPF_ENC_X1_AM, MPF_ENC_H_AM, MPF_ENC_XN_AM, MPF_ENC_QN_AM,
    MPF_ENC_FIELDS
};
typedef struct MpfEncoderCall {
    int32_t N, S, M, L, P, nl, F, reserved;   /* images, tokens per image, heads (8), levels, points, layers, ffn width */
    float eps, pad_;
    const int64_t* host_shapes;               /* [L][2] (H, W) on the host */
    const void* shapes_dev;                   /* the same on the device (int64) */
    const void* lsi_dev;                      /* level_start_index (int64, device) */
    const float* ref;                         /* reference points [S][2] */
    const float* pos_full;                    /* [S][256] positional term (sine + level embedding); may be NULL for one layer */
    const float* pos_am;                      /* its amax slot */
    const float* x0;                          /* [N*S][256] input of layer 0 */
    const float* x0_am;
    const float* q0;                          /* x0 + pos */
    const float* q0_am;
    const uint64_t* layers;                   /* HOST table [nl][MPF_ENC_FIELDS] */
} MpfEncoderCall;
int mpf_encoder_fields(void);                 /* MPF_ENC_FIELDS of the library (layout check of the binding) */
int mpf_encoder_forward(const MpfEncoderCall* call, void* stream);

/*
 * The backward of the same layers as ONE call (last layer first).  `layers`: HOST table [n_layers][MPF_ENCB_FIELDS] of device
 * addresses — what the forward kept (the arena of mpf_encoder_forward), the planes of the TRANSPOSED weights, the layer's amax
 * slots (zeroed by the caller) and its results: WGRAD = [dW2 | db2 | dW1 | db1 | dWo | dbo | dWv | dbv] (group_stride floats),
 * DW288 [288][256], LVL [L][288] (per-level column sums of d raw), DB288 [288].  The temporaries are shared by all layers;
 * g[(layer) & 1] receives the gradient handed to the layer below (g[0] after the call: the gradient of the encoder's input),
 * dgb_out [2 n_layers][2][256] the LayerNorm parameter gradients in call order (norm2, norm1 of layers n-1 .. 0).
 * rps288 / rps_group: rows per split of the 288-wide / the grouped weight gradients (multiples of 32; rps288 must not straddle
 * a level: split_level[s] = level of split s, int64 on the device).
 */
enum {
    MPF_ENCB_X, MPF_ENCB_Q, MPF_ENCB_VALUE, MPF_ENCB_LOC, MPF_ENCB_ATTN, MPF_ENCB_AO, MPF_ENCB_S1, MPF_ENCB_MEAN1, MPF_ENCB_RSTD1, MPF_ENCB_X1,
    MPF_ENCB_H, MPF_ENCB_HBITS, MPF_ENCB_S2, MPF_ENCB_MEAN2, MPF_ENCB_RSTD2,
    MPF_ENCB_X_AM, MPF_ENCB_Q_AM, MPF_ENCB_AO_AM, MPF_ENCB_X1_AM, MPF_ENCB_H_AM,
    MPF_ENCB_G1, MPF_ENCB_G2,
    MPF_ENCB_TV, MPF_ENCB_TV_AM, MPF_ENCB_TO, MPF_ENCB_TO_AM, MPF_ENCB_T1, MPF_ENCB_T1_AM, MPF_ENCB_T2, MPF_ENCB_T2_AM, MPF_ENCB_T288, MPF_ENCB_T288_AM,
    MPF_ENCB_DS2_AM, MPF_ENCB_DH_AM, MPF_ENCB_DS1_AM, MPF_ENCB_DRAW_AM, MPF_ENCB_GV_AM,
    MPF_ENCB_WGRAD, MPF_ENCB_DW288, MPF_ENCB_LVL, MPF_ENCB_DB288,
    MPF_ENCB_FIELDS
};
typedef struct MpfEncoderBwdCall {
    int32_t N, S, M, L, P, nl, F, rps288, rps_group, reserved;
    int64_t group_stride;                     /* floats of one split of the grouped weight gradients = of WGRAD */
    const int64_t* host_shapes;
    const float* gout;                        /* [N*S][256] gradient of the encoder's output */
    const int64_t* split_level;               /* device, [ceil(N*S / rps288)] */
    float *ds2, *dh, *dx1, *ds1, *dao, *gv, *draw;    /* temporaries: [R][256] ([R][F] for dh, [R][288] for draw) */
    float* dq[2];
    float* g[2];
    float *cpart288, *cs288, *part_group;     /* [ns288][288*256], [ns288][288], [ns_group][group_stride] */
    void* ln_parts;                           /* 2 n_layers slots of ln_stride bytes (mpf_res_ln256_backward_workspace_bytes, 256-aligned) */
    uint64_t ln_stride;
    void* msda_ws;                            /* mpf_msda_backward_workspace_bytes */
    uint64_t msda_ws_bytes;
    float* dgb_out;
    const uint64_t* layers;                   /* HOST table [nl][MPF_ENCB_FIELDS] */
} MpfEncoderBwdCall;
int mpf_encoder_bwd_fields(void);
int mpf_encoder_backward(const MpfEncoderBwdCall* call, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MPFORMER_HIP_H */
