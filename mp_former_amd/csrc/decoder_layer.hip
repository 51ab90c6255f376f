// One transformer-decoder layer (cross-attention, self-attention, FFN, post-norm residuals) issued
// from native code: the host side of include/mpformer_hip.h MpfDecoderLayer.  No kernels here — the
// layer is a fixed sequence of the library's own entry points (small_gemm.hip, attn.hip,
// elementwise.hip); what this file removes is the per-kernel cost of a Python autograd node (~20 us of
// host time against 3-6 us of GPU time per launch), which made the decoder launch-bound.
// Reference: mask2former_transformer_decoder.py:1784-1800 (the layer loop), :42-52, :100-112, :165-169.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpf_common.h"

namespace {

constexpr int kE = 256;
constexpr int kMaxDw = 8;        // weight-gradient problems per layer (6 with packed in-projection gradients, 8 without)

struct Carve {
    char* p;
    size_t used = 0;
    explicit Carve(void* base) : p(static_cast<char*>(base)) {}
    template <typename T>
    T* take(size_t count)
    {
        T* r = reinterpret_cast<T*>(p + used);
        used += (count * sizeof(T) + 255) & ~size_t(255);
        return r;
    }
};

struct FwdScratch {
    void *vT_c, *vT_s, *t;
    float *x1, *x2;
    size_t bytes;
};

FwdScratch fwd_scratch(void* base, int Qt, int N, int S)
{
    Carve c(base);
    FwdScratch f;
    const size_t R = (size_t)Qt * N;
    f.vT_c = c.take<uint16_t>((size_t)N * kE * S);
    f.vT_s = c.take<uint16_t>((size_t)N * kE * Qt);
    f.t = c.take<uint16_t>(R * kE);
    f.x1 = c.take<float>(R * kE);
    f.x2 = c.take<float>(R * kE);
    f.bytes = c.used;
    return f;
}

struct BwdScratch {
    // dt3 / dt2 / dt1: the gradients entering the FFN / self-attention / cross-attention output projections, dq / dq_c the
    // in-projection gradients of the two attentions — each in its own buffer, because the weight gradients that read them
    // are issued together at the end of the layer (one grouped launch)
    void *dt3, *dt2, *dt1, *dxb, *dh, *dout, *dq, *dk_s, *dv_s, *dq_c, *qT, *doT;
    float *ds_a, *ds_b, *delta;
    void* aux;                   // aux operands of the attention dK / dV kernel (mpf_attn_bwd_aux_bytes: the larger of the two attentions)
    size_t aux_bytes;
    uint16_t* tbuf;              // transposed [C, Rp] copies of the weight-gradient operands (2 F + 13 * 256 columns)
    void* ln_ws;                 // per-workgroup partial sums of the three LayerNorms' parameter gradients, ln_ws_bytes each
    size_t ln_ws_bytes;
    size_t bytes;
};

BwdScratch bwd_scratch(void* base, int Qt, int N, int H, int F, int S)
{
    Carve c(base);
    BwdScratch b;
    const size_t R = (size_t)Qt * N;
    const size_t LqP = (size_t)((Qt + 31) / 32 * 32);
    b.dt3 = c.take<uint16_t>(R * kE);
    b.dt2 = c.take<uint16_t>(R * kE);
    b.dt1 = c.take<uint16_t>(R * kE);
    b.dxb = c.take<uint16_t>(R * kE);
    b.dh = c.take<uint16_t>(R * F);
    b.dout = c.take<uint16_t>(R * kE);
    b.dq = c.take<uint16_t>(R * kE);
    b.dk_s = c.take<uint16_t>(R * kE);
    b.dv_s = c.take<uint16_t>(R * kE);
    b.dq_c = c.take<uint16_t>(R * kE);
    b.qT = c.take<uint16_t>((size_t)N * kE * LqP);
    b.doT = c.take<uint16_t>((size_t)N * kE * LqP);
    b.ds_a = c.take<float>(R * kE);
    b.ds_b = c.take<float>(R * kE);
    b.delta = c.take<float>((size_t)N * H * Qt);
    {
        const size_t ca = mpf_attn_bwd_aux_bytes(Qt, S, N, H, N), sa = mpf_attn_bwd_aux_bytes(Qt, Qt, N, H, 1);
        b.aux_bytes = ca > sa ? ca : sa;
        b.aux = c.take<char>(b.aux_bytes);
    }
    b.tbuf = c.take<uint16_t>((size_t)(2 * F + 13 * kE) * ((R + 31) / 32 * 32));
    b.ln_ws_bytes = (mpf_res_ln256_backward_workspace_bytes((int)R) + 255) & ~size_t(255);
    b.ln_ws = c.take<char>(3 * b.ln_ws_bytes);
    b.bytes = c.used;
    return b;
}

int check_layer(const MpfDecoderLayer* L, const char* who)
{
    if (!L) return mpf::fail(MPF_E_NULL, who);
    if (L->Qt <= 0 || L->N <= 0 || L->S <= 0 || L->ffn_dim <= 0 || L->H * 32 != kE || L->ffn_dim % 32)
        return mpf::fail(MPF_E_SHAPE, "decoder_layer: needs H * 32 == 256 channels, positive sizes, ffn_dim % 32 == 0");
    const void* need[] = {L->ca_wq, L->ca_bq, L->ca_wo, L->ca_bo, L->ca_gamma, L->ca_beta, L->sa_wq, L->sa_bq, L->sa_wk, L->sa_bk,
                          L->sa_wv, L->sa_bv, L->sa_wo, L->sa_bo, L->sa_gamma, L->sa_beta, L->ff_w1, L->ff_b1, L->ff_w2, L->ff_b2,
                          L->ff_gamma, L->ff_beta, L->x0, L->xb0, L->k_c, L->v_c, L->mask_c, L->q_c, L->kT_c, L->o_c, L->lse_c,
                          L->s1, L->mean1, L->rstd1, L->xb1, L->q_s, L->k_s, L->v_s, L->kT_s, L->o_s, L->lse_s, L->s2, L->mean2,
                          L->rstd2, L->xb2, L->h, L->s3, L->mean3, L->rstd3, L->scratch, L->attn_ws};
    for (const void* q : need)
        if (!q) return mpf::fail(MPF_E_NULL, "decoder_layer: NULL buffer in MpfDecoderLayer");
    return 0;
}

// y[R, J] = x[R, Kc] . w[J, Kc]^T + b (ReLU)
int lin_fwd(const void* x, const void* w, const void* b, void* y, int R, int J, int Kc, int relu, void* st)
{
    return mpf_small_gemm_bf16(x, Kc, 1, nullptr, w, Kc, 1, b, nullptr, 0, y, J, nullptr, R, J, Kc, relu, st);
}

// dx[R, Kin] = (dy[R, J] gated) . w[J, Kin] (+ acc)
int lin_dx(const void* dy, const void* gate, const void* w, const void* acc, void* dx, int R, int J, int Kin, void* st)
{
    return mpf_small_gemm_bf16(dy, J, 1, gate, w, 1, Kin, nullptr, acc, Kin, dx, Kin, nullptr, R, Kin, J, 0, st);
}

// dw[J, Kin] = (dy gated)^T . x[R, Kin];  db[J] = column sums of (dy gated)
int lin_dw(const void* dy, const void* gate, const void* x, void* dw, void* db, int R, int J, int Kin, void* st)
{
    return mpf_small_gemm_bf16(dy, 1, J, gate, x, 1, Kin, nullptr, nullptr, 0, dw, Kin, db, J, Kin, R, 0, st);
}

// the same problem as a descriptor of the grouped launch
MpfSmallGemmItem dw_item(const void* dy, const void* gate, const void* x, void* dw, void* db, int R, int J, int Kin, int a_blk = 0,
                         int64_t a_bs = 0)
{
    MpfSmallGemmItem m;
    m.a = dy; m.gate = gate; m.b = x; m.c = dw; m.rowsum_a = db;
    m.a_rs = 1; m.a_ks = a_blk ? kE : J; m.a_bs = a_bs; m.b_rs = 1; m.b_ks = Kin; m.ldc = Kin;
    m.a_blk = a_blk; m.I = J; m.J = Kin; m.Kc = R;
    return m;
}

inline const char* at(const void* p, size_t bytes) { return static_cast<const char*>(p) + bytes; }

// q | k | v of the self-attention stand side by side in memory (packed in_proj weight / bias, the three
// activation buffers, the three gradient buffers): the three GEMMs of each kind become ONE blocked GEMM
bool packed_weights(const MpfDecoderLayer* L)
{
    const size_t w = (size_t)kE * kE * 2, b = (size_t)kE * 2;
    return L->sa_wk == at(L->sa_wq, w) && L->sa_wv == at(L->sa_wq, 2 * w) && L->sa_bk == at(L->sa_bq, b) && L->sa_bv == at(L->sa_bq, 2 * b);
}

}  // namespace

namespace {
// A/B switches (mpf_set_option): decoder_dw_group = 0 issues the weight gradients as separate launches (round-1 form), 1 = one grouped
// launch on the row-contiguous operands (round 2), 2 = grouped transposes + one grouped launch on contraction-contiguous copies
int g_dw_group = 2;
// decoder_row_chain = 0: output projection, residual LayerNorm, decoder_norm and the mask_embed layers as separate launches
// (the form of rounds 2-4); 1: the row-local chains of row_chain.hip
int g_row_chain = 1;
}  // namespace

namespace mpf {
int set_decoder_option(const char* key, int v)
{
    if (!strcmp(key, "decoder_dw_group")) { if (v < 0 || v > 2) return -1; g_dw_group = v; return 0; }
    if (!strcmp(key, "decoder_row_chain")) { if (v < 0 || v > 1) return -1; g_row_chain = v; return 0; }
    return 1;
}
}  // namespace mpf

#define MPF_TRY(expr)                \
    do {                             \
        const int rc_ = (expr);      \
        if (rc_ != 0) return rc_;    \
    } while (0)

extern "C" uint64_t mpf_decoder_layer_struct_bytes(int which)
{
    return which == 0 ? sizeof(MpfDecoderLayer) : sizeof(MpfDecoderLayerGrad);
}

extern "C" uint64_t mpf_decoder_layer_scratch_bytes(int Qt, int N, int H, int S, int ffn_dim, int backward)
{
    if (Qt <= 0 || N <= 0 || H <= 0 || S <= 0 || ffn_dim <= 0) return 0;
    return backward ? bwd_scratch(nullptr, Qt, N, H, ffn_dim, S).bytes : fwd_scratch(nullptr, Qt, N, S).bytes;
}

extern "C" int mpf_decoder_layer_forward(const MpfDecoderLayer* L, void* st)
{
    MPF_TRY(check_layer(L, "decoder_layer_forward: NULL layer"));
    if (!L->x3 && !L->xb3) return mpf::fail(MPF_E_NULL, "decoder_layer_forward: no output buffer");
    const int Qt = L->Qt, N = L->N, H = L->H, S = L->S, F = L->ffn_dim, R = Qt * N;
    if (L->scratch_bytes < fwd_scratch(nullptr, Qt, N, S).bytes)
        return mpf::fail(MPF_E_SHAPE, "decoder_layer_forward: scratch too small");
    const FwdScratch f = fwd_scratch(L->scratch, Qt, N, S);
    const float scale = 0.17677669529663687f;       // 1 / sqrt(32)
    // cross-attention (:1784-1789) + post-norm
    MPF_TRY(lin_fwd(L->xb0, L->ca_wq, L->ca_bq, L->q_c, R, kE, kE, 0, st));
    MPF_TRY(mpf_attn_transpose2_strided(L->k_c, L->v_c, L->kv_row_stride, L->kv_img_stride, L->kT_c, f.vT_c, S, S, N, kE, st));
    MPF_TRY(mpf_attn_forward_kv(L->q_c, L->k_c, L->kv_row_stride, L->kv_img_stride, f.vT_c, L->mask_c, 1, L->o_c, L->lse_c, Qt, S, N,
                                H, 32, scale, L->attn_ws, L->attn_ws_bytes, st));
    if (g_row_chain) {
        MPF_TRY(mpf_lin256_res_ln_forward(L->o_c, L->ca_wo, L->ca_bo, L->x0, L->ca_gamma, L->ca_beta, L->s1, f.x1, L->xb1, L->mean1, L->rstd1,
                                          R, L->eps, st));
    } else {
        MPF_TRY(lin_fwd(L->o_c, L->ca_wo, L->ca_bo, f.t, R, kE, kE, 0, st));
        MPF_TRY(mpf_res_ln256_forward(L->x0, f.t, MPF_BF16, L->ca_gamma, L->ca_beta, L->s1, f.x1, L->xb1, L->mean1, L->rstd1, R,
                                      L->eps, nullptr, 0, nullptr, st));
    }
    // self-attention (:1791-1795) + post-norm
    const size_t act = (size_t)R * kE * 2;
    if (packed_weights(L) && L->k_s == at(L->q_s, act) && L->v_s == at(L->q_s, 2 * act)) {
        MPF_TRY(mpf_small_gemm_bf16_blocked(L->xb1, kE, 1, 0, 0, nullptr, L->sa_wq, kE, 1, L->sa_bq, nullptr, 0, L->q_s, kE, kE,
                                            (int64_t)R * kE, nullptr, R, 3 * kE, kE, 0, st));
    } else {
        MPF_TRY(lin_fwd(L->xb1, L->sa_wq, L->sa_bq, L->q_s, R, kE, kE, 0, st));
        MPF_TRY(lin_fwd(L->xb1, L->sa_wk, L->sa_bk, L->k_s, R, kE, kE, 0, st));
        MPF_TRY(lin_fwd(L->xb1, L->sa_wv, L->sa_bv, L->v_s, R, kE, kE, 0, st));
    }
    MPF_TRY(mpf_attn_transpose2(L->k_s, L->v_s, L->kT_s, f.vT_s, Qt, Qt, N, kE, st));
    MPF_TRY(mpf_attn_forward(L->q_s, L->k_s, f.vT_s, L->mask_s, 0, L->o_s, L->lse_s, Qt, Qt, N, H, 32, scale, L->attn_ws,
                             L->attn_ws_bytes, st));
    if (g_row_chain) {
        MPF_TRY(mpf_lin256_res_ln_forward(L->o_s, L->sa_wo, L->sa_bo, f.x1, L->sa_gamma, L->sa_beta, L->s2, f.x2, L->xb2, L->mean2, L->rstd2,
                                          R, L->eps, st));
    } else {
        MPF_TRY(lin_fwd(L->o_s, L->sa_wo, L->sa_bo, f.t, R, kE, kE, 0, st));
        MPF_TRY(mpf_res_ln256_forward(f.x1, f.t, MPF_BF16, L->sa_gamma, L->sa_beta, L->s2, f.x2, L->xb2, L->mean2, L->rstd2, R,
                                      L->eps, nullptr, 0, nullptr, st));
    }
    // FFN (:1798-1800) + post-norm
    MPF_TRY(lin_fwd(L->xb2, L->ff_w1, L->ff_b1, L->h, R, F, kE, 1, st));
    MPF_TRY(lin_fwd(L->h, L->ff_w2, L->ff_b2, f.t, R, kE, F, 0, st));
    MPF_TRY(mpf_res_ln256_forward(f.x2, f.t, MPF_BF16, L->ff_gamma, L->ff_beta, L->s3, L->x3, L->xb3, L->mean3, L->rstd3, R,
                                  L->eps, nullptr, 0, nullptr, st));
    return 0;
}

extern "C" int mpf_decoder_layer_backward(const MpfDecoderLayer* L, const MpfDecoderLayerGrad* G, void* st)
{
    MPF_TRY(check_layer(L, "decoder_layer_backward: NULL layer"));
    if (!G) return mpf::fail(MPF_E_NULL, "decoder_layer_backward: NULL grad");
    if (!G->g_x3 && !G->g_xb3) return mpf::fail(MPF_E_NULL, "decoder_layer_backward: no upstream gradient");
    const void* need[] = {G->d_x0, G->d_xb0, G->d_k_c, G->d_v_c, G->d_ca_wq, G->d_ca_bq, G->d_ca_wo, G->d_ca_bo, G->d_sa_wq, G->d_sa_bq,
                          G->d_sa_wk, G->d_sa_bk, G->d_sa_wv, G->d_sa_bv, G->d_sa_wo, G->d_sa_bo, G->d_ff_w1, G->d_ff_b1, G->d_ff_w2,
                          G->d_ff_b2, G->d_ln};
    for (const void* q : need)
        if (!q) return mpf::fail(MPF_E_NULL, "decoder_layer_backward: NULL buffer in MpfDecoderLayerGrad");
    const int Qt = L->Qt, N = L->N, H = L->H, S = L->S, F = L->ffn_dim, R = Qt * N;
    const int LqP = (Qt + 31) / 32 * 32;
    if (L->scratch_bytes < bwd_scratch(nullptr, Qt, N, H, F, S).bytes)
        return mpf::fail(MPF_E_SHAPE, "decoder_layer_backward: scratch too small");
    const BwdScratch b = bwd_scratch(L->scratch, Qt, N, H, F, S);
    const float scale = 0.17677669529663687f;
    float* dln = G->d_ln;
    // LayerNorm parameter gradients: the three backward launches leave per-workgroup partial sums, ONE launch at the end of the
    // layer reduces them in a fixed order (no float atomics, nothing to zero)
    char* ln_part = static_cast<char*>(b.ln_ws);
    // The weight gradients depend only on the dY buffers of the chain below (each kept in its own scratch buffer), so they
    // are collected here and issued as ONE grouped launch after the chain.
    MpfSmallGemmItem dw[kMaxDw];
    int ndw = 0;
    // FFN block: x3 = LN(x2 + W2 relu(W1 xb2))
    if (G->g_x3_plus && !G->g_x3) return mpf::fail(MPF_E_NULL, "decoder_layer_backward: g_x3_plus without g_x3");
    MPF_TRY(mpf_res_ln256_backward_partial(L->s3, L->mean3, L->rstd3, L->ff_gamma, G->g_x3, G->g_xb3, G->g_x3_plus, b.ds_a, b.dt3, R,
                                           ln_part + 2 * b.ln_ws_bytes, b.ln_ws_bytes, st));
    MPF_TRY(lin_dx(b.dt3, nullptr, L->ff_w2, nullptr, b.dh, R, kE, F, st));
    dw[ndw++] = dw_item(b.dt3, nullptr, L->h, G->d_ff_w2, G->d_ff_b2, R, kE, F);
    MPF_TRY(lin_dx(b.dh, L->h, L->ff_w1, nullptr, b.dxb, R, F, kE, st));
    dw[ndw++] = dw_item(b.dh, L->h, L->xb2, G->d_ff_w1, G->d_ff_b1, R, F, kE);
    // self-attention block: x2 = LN(x1 + Wo attn(Wq xb1, Wk xb1, Wv xb1))
    MPF_TRY(mpf_res_ln256_backward_partial(L->s2, L->mean2, L->rstd2, L->sa_gamma, b.ds_a, b.dxb, nullptr, b.ds_b, b.dt2, R,
                                           ln_part + b.ln_ws_bytes, b.ln_ws_bytes, st));
    MPF_TRY(lin_dx(b.dt2, nullptr, L->sa_wo, nullptr, b.dout, R, kE, kE, st));
    dw[ndw++] = dw_item(b.dt2, nullptr, L->o_s, G->d_sa_wo, G->d_sa_bo, R, kE, kE);
    MPF_TRY(mpf_attn_bwd_prep_aux(L->q_s, b.dout, L->o_s, L->lse_s, L->mask_s, 0, Qt, b.qT, b.doT, b.delta, b.aux, b.aux_bytes, Qt, LqP,
                                  N, H, st));
    MPF_TRY(mpf_attn_backward_kv_aux(L->q_s, L->k_s, L->v_s, 0, 0, L->kT_s, b.qT, b.dout, b.doT, L->mask_s, 0, L->lse_s, b.delta, b.dq,
                                     b.dk_s, b.dv_s, 0, 0, Qt, LqP, Qt, N, H, 32, scale, L->attn_ws, L->attn_ws_bytes, b.aux, st));
    const size_t act = (size_t)R * kE * 2, wsz = (size_t)kE * kE * 2, bsz = (size_t)kE * 2;
    const bool grads_packed = b.dk_s == at(b.dq, act) && b.dv_s == at(b.dq, 2 * act);
    if (packed_weights(L) && grads_packed) {
        // dxb = [dq | dk | dv] . W_in (contraction over the 768 packed outputs, blocked over the three buffers)
        MPF_TRY(mpf_small_gemm_bf16_blocked(b.dq, kE, 1, kE, (int64_t)R * kE, nullptr, L->sa_wq, 1, kE, nullptr, nullptr, 0, b.dxb, kE,
                                            0, 0, nullptr, R, kE, 3 * kE, 0, st));
    } else {
        MPF_TRY(lin_dx(b.dq, nullptr, L->sa_wq, nullptr, b.dxb, R, kE, kE, st));
        MPF_TRY(lin_dx(b.dk_s, nullptr, L->sa_wk, b.dxb, b.dxb, R, kE, kE, st));
        MPF_TRY(lin_dx(b.dv_s, nullptr, L->sa_wv, b.dxb, b.dxb, R, kE, kE, st));
    }
    if (grads_packed && G->d_sa_wk == at(G->d_sa_wq, wsz) && G->d_sa_wv == at(G->d_sa_wq, 2 * wsz) &&
        G->d_sa_bk == at(G->d_sa_bq, bsz) && G->d_sa_bv == at(G->d_sa_bq, 2 * bsz)) {
        // dW_in [768, 256] = [dq | dk | dv]^T . xb1 and its 768 bias gradients as one problem (rows of A blocked)
        dw[ndw++] = dw_item(b.dq, nullptr, L->xb1, G->d_sa_wq, G->d_sa_bq, R, 3 * kE, kE, kE, (int64_t)R * kE);
    } else {
        dw[ndw++] = dw_item(b.dq, nullptr, L->xb1, G->d_sa_wq, G->d_sa_bq, R, kE, kE);
        dw[ndw++] = dw_item(b.dk_s, nullptr, L->xb1, G->d_sa_wk, G->d_sa_bk, R, kE, kE);
        dw[ndw++] = dw_item(b.dv_s, nullptr, L->xb1, G->d_sa_wv, G->d_sa_bv, R, kE, kE);
    }
    // cross-attention block: x1 = LN(x0 + Wo attn(Wq xb0, k_c, v_c))
    MPF_TRY(mpf_res_ln256_backward_partial(L->s1, L->mean1, L->rstd1, L->ca_gamma, b.ds_b, b.dxb, nullptr, G->d_x0, b.dt1, R, ln_part,
                                           b.ln_ws_bytes, st));
    MPF_TRY(mpf_ln_partial_reduce(ln_part, b.ln_ws_bytes, R, 3, dln, st));      // dln = [ca | sa | ff] x (dgamma, dbeta)
    MPF_TRY(lin_dx(b.dt1, nullptr, L->ca_wo, nullptr, b.dout, R, kE, kE, st));
    dw[ndw++] = dw_item(b.dt1, nullptr, L->o_c, G->d_ca_wo, G->d_ca_bo, R, kE, kE);
    MPF_TRY(mpf_attn_bwd_prep_aux(L->q_c, b.dout, L->o_c, L->lse_c, L->mask_c, 1, S, b.qT, b.doT, b.delta, b.aux, b.aux_bytes, Qt, LqP, N,
                                  H, st));
    MPF_TRY(mpf_attn_backward_kv_aux(L->q_c, L->k_c, L->v_c, L->kv_row_stride, L->kv_img_stride, L->kT_c, b.qT, b.dout, b.doT, L->mask_c,
                                     1, L->lse_c, b.delta, b.dq_c, G->d_k_c, G->d_v_c, G->dkv_row_stride, G->dkv_img_stride, Qt, LqP, S,
                                     N, H, 32, scale, L->attn_ws, L->attn_ws_bytes, b.aux, st));
    MPF_TRY(lin_dx(b.dq_c, nullptr, L->ca_wq, nullptr, G->d_xb0, R, kE, kE, st));
    dw[ndw++] = dw_item(b.dq_c, nullptr, L->xb0, G->d_ca_wq, G->d_ca_bq, R, kE, kE);
    if (g_dw_group == 2) {
        // the weight gradients contract over the ROWS of dY and x: on transposed [C, Rp] copies (one grouped launch) both operands
        // are contraction-contiguous, i.e. 16-byte fragment loads instead of eight 2-byte ones per fragment; same 32-row
        // contraction steps on the same waves, zero padding: bit-identical to the row-contiguous form
        const int Rp = (R + 31) / 32 * 32;
        uint16_t* tp = b.tbuf;
        MpfTransposeItem tr[16];
        int ntr = 0;
        auto tcopy = [&](const void* src, const void* gate, int C) {
            uint16_t* dst = tp;
            tr[ntr].src = src; tr[ntr].gate = gate; tr[ntr].dst = dst; tr[ntr].ld = C; tr[ntr].R = R; tr[ntr].C = C;
            ++ntr;
            tp += (size_t)C * Rp;
            return dst;
        };
        const uint16_t* dt3T = tcopy(b.dt3, nullptr, kE);
        const uint16_t* hT = tcopy(L->h, nullptr, F);
        const uint16_t* dhT = tcopy(b.dh, L->h, F);
        const uint16_t* xb2T = tcopy(L->xb2, nullptr, kE);
        const uint16_t* dt2T = tcopy(b.dt2, nullptr, kE);
        const uint16_t* osT = tcopy(L->o_s, nullptr, kE);
        const uint16_t* dqT = tcopy(b.dq, nullptr, kE);          // dq | dk | dv: three consecutive [256, Rp] blocks = one [768, Rp]
        tcopy(b.dk_s, nullptr, kE);
        tcopy(b.dv_s, nullptr, kE);
        const uint16_t* xb1T = tcopy(L->xb1, nullptr, kE);
        const uint16_t* dt1T = tcopy(b.dt1, nullptr, kE);
        const uint16_t* ocT = tcopy(L->o_c, nullptr, kE);
        const uint16_t* dqcT = tcopy(b.dq_c, nullptr, kE);
        const uint16_t* xb0T = tcopy(L->xb0, nullptr, kE);
        MPF_TRY(mpf_transpose_group_bf16(tr, ntr, Rp, st));
        auto titem = [&](const uint16_t* aT, const uint16_t* bT, void* dwp, void* dbp, int J, int Kin) {
            MpfSmallGemmItem m;
            m.a = aT; m.gate = nullptr; m.b = bT; m.c = dwp; m.rowsum_a = dbp;
            m.a_rs = Rp; m.a_ks = 1; m.a_bs = 0; m.b_rs = Rp; m.b_ks = 1; m.ldc = Kin;
            m.a_blk = 0; m.I = J; m.J = Kin; m.Kc = Rp;
            return m;
        };
        MpfSmallGemmItem tw[kMaxDw];
        int nt = 0;
        tw[nt++] = titem(dt3T, hT, G->d_ff_w2, G->d_ff_b2, kE, F);
        tw[nt++] = titem(dhT, xb2T, G->d_ff_w1, G->d_ff_b1, F, kE);
        tw[nt++] = titem(dt2T, osT, G->d_sa_wo, G->d_sa_bo, kE, kE);
        if (G->d_sa_wk == at(G->d_sa_wq, wsz) && G->d_sa_wv == at(G->d_sa_wq, 2 * wsz) && G->d_sa_bk == at(G->d_sa_bq, bsz) &&
            G->d_sa_bv == at(G->d_sa_bq, 2 * bsz)) {
            tw[nt++] = titem(dqT, xb1T, G->d_sa_wq, G->d_sa_bq, 3 * kE, kE);
        } else {
            tw[nt++] = titem(dqT, xb1T, G->d_sa_wq, G->d_sa_bq, kE, kE);
            tw[nt++] = titem(dqT + (size_t)kE * Rp, xb1T, G->d_sa_wk, G->d_sa_bk, kE, kE);
            tw[nt++] = titem(dqT + (size_t)2 * kE * Rp, xb1T, G->d_sa_wv, G->d_sa_bv, kE, kE);
        }
        tw[nt++] = titem(dt1T, ocT, G->d_ca_wo, G->d_ca_bo, kE, kE);
        tw[nt++] = titem(dqcT, xb0T, G->d_ca_wq, G->d_ca_bq, kE, kE);
        MPF_TRY(mpf_small_gemm_bf16_group(tw, nt, st));
    } else if (g_dw_group) {
        MPF_TRY(mpf_small_gemm_bf16_group(dw, ndw, st));
    } else {
        for (int t = 0; t < ndw; ++t)
            MPF_TRY(mpf_small_gemm_bf16_blocked(dw[t].a, dw[t].a_rs, dw[t].a_ks, dw[t].a_blk, dw[t].a_bs, dw[t].gate, dw[t].b, dw[t].b_rs,
                                                dw[t].b_ks, nullptr, nullptr, 0, dw[t].c, dw[t].ldc, 0, 0, dw[t].rowsum_a, dw[t].I,
                                                dw[t].J, dw[t].Kc, 0, st));
    }
    return 0;
}

// The attention mask of the NEXT layer from this layer's residual stream (mask2former_transformer_decoder.py:1869-1875 with the
// prediction-head part of :1859-1866, detached as the reference detaches it): decoder_norm -> mask_embed MLP (ReLU after the first
// two layers) -> sign of the product with the features pooled to the level grid, MP rows and the all-masked-row rule applied by
// the mask-head kernel.  Five launches issued from one call: the python glue around them (five autograd-free ops, ten times per
// step) was 0.5 ms of launch-thread time at the one place of the step where the device waits for every launch.
extern "C" size_t mpf_next_attn_mask_scratch_bytes(int N, int Q)
{
    if (N <= 0 || Q <= 0) return 0;
    const size_t rows = (size_t)N * Q;
    return 3 * ((rows * kE * 2 + 255) & ~(size_t)255) + 2 * ((rows * 4 + 255) & ~(size_t)255);
}

extern "C" int mpf_next_attn_mask(const MpfNextMask* m, void* st)
{
    if (!m || !m->x || !m->ln_gamma || !m->ln_beta || !m->w0 || !m->w1 || !m->w2 || !m->pooled || !m->out || !m->flags || !m->scratch)
        return mpf::fail(MPF_E_NULL, "next_attn_mask: NULL buffer");
    if (m->N <= 0 || m->Q <= 0 || m->HW <= 0 || m->pad < 0) return mpf::fail(MPF_E_SHAPE, "next_attn_mask: bad sizes");
    if (m->scratch_bytes < mpf_next_attn_mask_scratch_bytes(m->N, m->Q)) return mpf::fail(MPF_E_SHAPE, "next_attn_mask: scratch too small");
    const int rows = m->N * m->Q;
    const size_t act = ((size_t)rows * kE * 2 + 255) & ~(size_t)255, vec = ((size_t)rows * 4 + 255) & ~(size_t)255;
    char* sc = static_cast<char*>(m->scratch);
    void* d16 = sc;                 // decoder_norm(x) in bf16; reused for the third layer's result
    void* e1 = sc + act;
    void* e2 = sc + 2 * act;
    float* mean = reinterpret_cast<float*>(sc + 3 * act);
    float* rstd = reinterpret_cast<float*>(sc + 3 * act + vec);
    if (g_row_chain && m->b0 && m->b1 && m->b2) {
        MPF_TRY(mpf_ln256_mlp3_forward(m->x, m->ln_gamma, m->ln_beta, m->w0, m->b0, m->w1, m->b1, m->w2, m->b2, d16, rows, m->eps, st));
    } else {
        MPF_TRY(mpf_res_ln256_forward(m->x, nullptr, 0, m->ln_gamma, m->ln_beta, nullptr, nullptr, d16, mean, rstd, rows, m->eps, nullptr, 0,
                                      nullptr, st));
        MPF_TRY(mpf_small_gemm_bf16(d16, kE, 1, nullptr, m->w0, kE, 1, m->b0, nullptr, 0, e1, kE, nullptr, rows, kE, kE, 1, st));
        MPF_TRY(mpf_small_gemm_bf16(e1, kE, 1, nullptr, m->w1, kE, 1, m->b1, nullptr, 0, e2, kE, nullptr, rows, kE, kE, 1, st));
        MPF_TRY(mpf_small_gemm_bf16(e2, kE, 1, nullptr, m->w2, kE, 1, m->b2, nullptr, 0, d16, kE, nullptr, rows, kE, kE, 0, st));
    }
    // mask_embed is sequence-first [Q, N, 256]: image stride 256, query stride N * 256
    return mpf_mask_head_bits(d16, kE, (int64_t)m->N * kE, m->pooled, m->mp_rows, m->pad, m->out, m->flags, m->N, m->Q, m->HW, st);
}
