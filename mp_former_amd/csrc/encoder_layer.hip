// The six MSDeformAttn encoder layers of the pixel decoder issued from native code: the host side of include/mpformer_hip.h
// MpfEncoderCall (forward).  No kernels here — a layer is a fixed sequence of the library's own entry points (gemm3.hip,
// msda_block.hip, elementwise.hip); what this file removes is the per-launch cost of the python glue (ctypes marshalling of ~20
// arguments and one torch.empty per result: ~50 calls and ~100 allocations per forward).  Same kernels, same arguments and
// the same order as mp_former_amd/encoder_fused.py issued them one by one (tests/test_encoder_fused_gpu.py compares both).
// Reference: mask2former/modeling/pixel_decoder/msdeformattn.py:92-161 with ops/modules/ms_deform_attn.py:82-125 inside.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpf_common.h"

#define MPF_TRY(expr)                \
    do {                             \
        const int rc_ = (expr);      \
        if (rc_ != 0) return rc_;    \
    } while (0)

extern "C" int mpf_encoder_fields(void) { return MPF_ENC_FIELDS; }
extern "C" int mpf_encoder_bwd_fields(void) { return MPF_ENCB_FIELDS; }

extern "C" int mpf_encoder_forward(const MpfEncoderCall* E, void* st)
{
    if (!E || !E->layers || !E->host_shapes || !E->shapes_dev || !E->lsi_dev || !E->ref || !E->x0 || !E->q0 || !E->x0_am || !E->q0_am)
        return mpf::fail(MPF_E_NULL, "encoder_forward: NULL buffer");
    if (E->N <= 0 || E->S <= 0 || E->M != 8 || E->L <= 0 || E->P <= 0 || E->nl <= 0 || E->F <= 0 || E->F % 128 != 0)
        return mpf::fail(MPF_E_SHAPE, "encoder_forward: needs 8 heads (256 channels), ffn width a multiple of 128, positive sizes");
    if (E->nl > 1 && (!E->pos_full || !E->pos_am)) return mpf::fail(MPF_E_NULL, "encoder_forward: NULL positional term");
    constexpr int C = 256;
    const int R = E->N * E->S, F = E->F, NO = E->M * E->L * E->P * 3;
    const float* x = E->x0;
    const float* x_am = E->x0_am;
    const float* q = E->q0;
    const float* q_am = E->q0_am;
    for (int i = 0; i < E->nl; ++i) {
        const uint64_t* f = E->layers + (size_t)i * MPF_ENC_FIELDS;
        auto P = [&](int k) { return reinterpret_cast<void*>(f[k]); };
        auto PF = [&](int k) { return reinterpret_cast<float*>(f[k]); };
        for (int k = 0; k < MPF_ENC_FIELDS; ++k)
            if (!f[k] && k != MPF_ENC_QN && k != MPF_ENC_QN_AM) return mpf::fail(MPF_E_NULL, "encoder_forward: NULL field in a layer table");
        const bool last = i + 1 == E->nl;
        if (!last && (!f[MPF_ENC_QN] || !f[MPF_ENC_QN_AM])) return mpf::fail(MPF_E_NULL, "encoder_forward: NULL q of the next layer");
        // value = value_proj(x) (its epilogue records max |value|: the bound of the attention output)
        MPF_TRY(mpf_gemm3_tn_h2(x, C, x_am, P(MPF_ENC_PV), PF(MPF_ENC_PV_AM), PF(MPF_ENC_BV), nullptr, 0, nullptr, 0, nullptr, 0,
                                PF(MPF_ENC_VALUE), C, PF(MPF_ENC_AO_AM), R, C, C, 0, st));
        // sampling offsets | attention logits in one 288-wide product of q = x + pos
        MPF_TRY(mpf_gemm3_tn_h2(q, C, q_am, P(MPF_ENC_P288), PF(MPF_ENC_P288_AM), PF(MPF_ENC_B288), nullptr, 0, nullptr, 0, nullptr, 0,
                                PF(MPF_ENC_RAW), NO, nullptr, R, NO, C, 0, st));
        // softmax, loc = ref + offset / (W_l, H_l) and the sampling itself (ops/modules/ms_deform_attn.py:103-121)
        MPF_TRY(mpf_msda_forward_raw_hs(P(MPF_ENC_VALUE), (const int64_t*)E->shapes_dev, (const int64_t*)E->lsi_dev, E->host_shapes,
                                        P(MPF_ENC_RAW), E->ref, P(MPF_ENC_LOC), P(MPF_ENC_ATTN), P(MPF_ENC_AO), E->N, E->S, E->M, 32,
                                        E->L, E->S, E->P, MPF_F32, st));
        // s1 = output_proj(ao) + x;  x1 = norm1(s1)
        MPF_TRY(mpf_gemm3_tn_h2(PF(MPF_ENC_AO), C, PF(MPF_ENC_AO_AM), P(MPF_ENC_PO), PF(MPF_ENC_PO_AM), PF(MPF_ENC_BO), x, C, nullptr, 0,
                                nullptr, 0, PF(MPF_ENC_S1), C, nullptr, R, C, C, 0, st));
        MPF_TRY(mpf_res_ln256_forward_b(PF(MPF_ENC_S1), nullptr, 0, PF(MPF_ENC_G1), PF(MPF_ENC_B1), nullptr, PF(MPF_ENC_X1), nullptr,
                                        PF(MPF_ENC_MEAN1), PF(MPF_ENC_RSTD1), R, E->eps, nullptr, 0, nullptr, PF(MPF_ENC_X1_AM), nullptr,
                                        nullptr, st));
        // h = relu(linear1(x1)) (+ its gate as a bit mask);  s2 = linear2(h) + x1
        MPF_TRY(mpf_gemm3_tn_h2_bits(PF(MPF_ENC_X1), C, PF(MPF_ENC_X1_AM), P(MPF_ENC_P1), PF(MPF_ENC_P1_AM), PF(MPF_ENC_BB1), nullptr, 0,
                                     nullptr, 0, nullptr, 0, PF(MPF_ENC_H), F, PF(MPF_ENC_H_AM), (unsigned char*)P(MPF_ENC_HBITS), F / 8,
                                     R, F, C, 1, st));
        MPF_TRY(mpf_gemm3_tn_h2(PF(MPF_ENC_H), F, PF(MPF_ENC_H_AM), P(MPF_ENC_P2), PF(MPF_ENC_P2_AM), PF(MPF_ENC_BB2), PF(MPF_ENC_X1), C,
                                nullptr, 0, nullptr, 0, PF(MPF_ENC_S2), C, nullptr, R, C, F, 0, st));
        // x2 = norm2(s2); the next layer's q = x2 + pos from the same pass
        MPF_TRY(mpf_res_ln256_forward_b(PF(MPF_ENC_S2), nullptr, 0, PF(MPF_ENC_G2), PF(MPF_ENC_B2), nullptr, PF(MPF_ENC_X2), nullptr,
                                        PF(MPF_ENC_MEAN2), PF(MPF_ENC_RSTD2), R, E->eps, last ? nullptr : E->pos_full, last ? 0 : E->S,
                                        last ? nullptr : PF(MPF_ENC_QN), PF(MPF_ENC_XN_AM), last ? nullptr : E->pos_am,
                                        last ? nullptr : PF(MPF_ENC_QN_AM), st));
        x = PF(MPF_ENC_X2); x_am = PF(MPF_ENC_XN_AM);
        q = PF(MPF_ENC_QN); q_am = PF(MPF_ENC_QN_AM);
    }
    return 0;
}

// The backward of the same layers, last layer first: per layer the two LayerNorm backwards (parameter gradients as per-workgroup
// partials, reduced once at the end), the five input-gradient products, the MSDA backward (raw form with the forward result),
// the 288-wide weight gradient with its per-level column sums (level_embed / bias gradients) and the four plain weight gradients
// as one grouped launch + one reduction — the calls mp_former_amd/encoder_fused.py used to issue one by one, same arguments.
// Temporaries (d s2, d hidden, ...) are shared by all layers; the gradient handed to the layer below and d q alternate between
// two buffers each.
extern "C" int mpf_encoder_backward(const MpfEncoderBwdCall* E, void* st)
{
    if (!E || !E->layers || !E->host_shapes || !E->gout || !E->split_level || !E->ds2 || !E->dh || !E->dx1 || !E->ds1 || !E->dao || !E->gv ||
        !E->draw || !E->dq[0] || !E->dq[1] || !E->g[0] || !E->g[1] || !E->cpart288 || !E->cs288 || !E->part_group || !E->ln_parts ||
        !E->msda_ws || !E->dgb_out)
        return mpf::fail(MPF_E_NULL, "encoder_backward: NULL buffer");
    if (E->N <= 0 || E->S <= 0 || E->M != 8 || E->L <= 0 || E->L > 4 || E->P <= 0 || E->nl <= 0 || E->F <= 0 || E->F % 128 != 0 || E->rps288 <= 0 ||
        E->rps_group <= 0 || E->group_stride <= 0)
        return mpf::fail(MPF_E_SHAPE, "encoder_backward: bad sizes");
    constexpr int C = 256;
    const int R = E->N * E->S, F = E->F, NO = E->M * E->L * E->P * 3;
    const int ns288 = (R + E->rps288 - 1) / E->rps288, nsg = (R + E->rps_group - 1) / E->rps_group;
    const float* g_in = E->gout;
    const float* gq_in = nullptr;
    int kln = 0;
    char* parts = static_cast<char*>(E->ln_parts);
    for (int i = E->nl - 1; i >= 0; --i) {
        const uint64_t* f = E->layers + (size_t)i * MPF_ENCB_FIELDS;
        for (int k = 0; k < MPF_ENCB_FIELDS; ++k)
            if (!f[k]) return mpf::fail(MPF_E_NULL, "encoder_backward: NULL field in a layer table");
        auto P = [&](int k) { return reinterpret_cast<void*>(f[k]); };
        auto PF = [&](int k) { return reinterpret_cast<float*>(f[k]); };
        float* g_out = E->g[i & 1];
        float* dq = E->dq[i & 1];
        // norm2 <- ffn
        MPF_TRY(mpf_res_ln256_backward_partial_amax(PF(MPF_ENCB_S2), PF(MPF_ENCB_MEAN2), PF(MPF_ENCB_RSTD2), PF(MPF_ENCB_G2), g_in, nullptr,
                                                    gq_in, E->ds2, nullptr, R, parts + (size_t)kln * E->ln_stride, E->ln_stride,
                                                    PF(MPF_ENCB_DS2_AM), st));
        ++kln;
        MPF_TRY(mpf_gemm3_tn_h2_bits(E->ds2, C, PF(MPF_ENCB_DS2_AM), P(MPF_ENCB_T2), PF(MPF_ENCB_T2_AM), nullptr, nullptr, 0, nullptr, 0,
                                     (const unsigned char*)P(MPF_ENCB_HBITS), F / 8, E->dh, F, PF(MPF_ENCB_DH_AM), nullptr, 0, R, F, C, 0, st));
        MPF_TRY(mpf_gemm3_tn_h2(E->dh, F, PF(MPF_ENCB_DH_AM), P(MPF_ENCB_T1), PF(MPF_ENCB_T1_AM), nullptr, E->ds2, C, nullptr, 0, nullptr, 0,
                                E->dx1, C, nullptr, R, C, F, 0, st));
        // norm1 <- attention
        MPF_TRY(mpf_res_ln256_backward_partial_amax(PF(MPF_ENCB_S1), PF(MPF_ENCB_MEAN1), PF(MPF_ENCB_RSTD1), PF(MPF_ENCB_G1), E->dx1, nullptr,
                                                    nullptr, E->ds1, nullptr, R, parts + (size_t)kln * E->ln_stride, E->ln_stride,
                                                    PF(MPF_ENCB_DS1_AM), st));
        ++kln;
        MPF_TRY(mpf_gemm3_tn_h2(E->ds1, C, PF(MPF_ENCB_DS1_AM), P(MPF_ENCB_TO), PF(MPF_ENCB_TO_AM), nullptr, nullptr, 0, nullptr, 0, nullptr, 0,
                                E->dao, C, nullptr, R, C, C, 0, st));
        MPF_TRY(mpf_msda_backward_ws_raw_o(P(MPF_ENCB_VALUE), E->host_shapes, P(MPF_ENCB_LOC), P(MPF_ENCB_ATTN), E->dao, P(MPF_ENCB_AO), E->gv,
                                           E->draw, E->N, E->S, E->M, 32, E->L, E->S, E->P, MPF_F32, E->msda_ws, E->msda_ws_bytes,
                                           PF(MPF_ENCB_DRAW_AM), PF(MPF_ENCB_GV_AM), st));
        MPF_TRY(mpf_gemm3_tn_h2(E->draw, NO, PF(MPF_ENCB_DRAW_AM), P(MPF_ENCB_T288), PF(MPF_ENCB_T288_AM), nullptr, nullptr, 0, nullptr, 0,
                                nullptr, 0, dq, C, nullptr, R, C, NO, 0, st));
        // d W288^T = q^T . draw with the per-split column sums of draw: bias gradient and, per level, the level_embed gradient
        MPF_TRY(mpf_gemm3_nt_h2(PF(MPF_ENCB_Q), C, PF(MPF_ENCB_Q_AM), E->draw, NO, PF(MPF_ENCB_DRAW_AM), E->cpart288, nullptr, E->cs288, R, C, NO,
                                E->rps288, 1, st));
        MPF_TRY(mpf_gemm3_nt_reduce_levels(E->cpart288, (int64_t)C * NO, E->cs288, NO, ns288, E->split_level, E->L, PF(MPF_ENCB_DW288),
                                           PF(MPF_ENCB_LVL), PF(MPF_ENCB_DB288), st));
        // gradient of the layer's input: value_proj path + the residual; the (src + pos) path joins in the layer below's norm2
        MPF_TRY(mpf_gemm3_tn_h2(E->gv, C, PF(MPF_ENCB_GV_AM), P(MPF_ENCB_TV), PF(MPF_ENCB_TV_AM), nullptr, E->ds1, C, i == 0 ? dq : nullptr,
                                i == 0 ? C : 0, nullptr, 0, g_out, C, nullptr, R, C, C, 0, st));
        // the four plain weight gradients (operands all alive here): one grouped launch + one reduction
        {
            float* part = E->part_group;
            const int64_t o2 = 0, o1 = o2 + (int64_t)C * F + C, oo = o1 + (int64_t)F * C + F, ov = oo + (int64_t)C * C + C;
            MpfNtItemH2 it[4] = {
                {E->ds2, C, PF(MPF_ENCB_DS2_AM), PF(MPF_ENCB_H), F, PF(MPF_ENCB_H_AM), part + o2, part + o2 + (int64_t)C * F, C, F},
                {E->dh, F, PF(MPF_ENCB_DH_AM), PF(MPF_ENCB_X1), C, PF(MPF_ENCB_X1_AM), part + o1, part + o1 + (int64_t)F * C, F, C},
                {E->ds1, C, PF(MPF_ENCB_DS1_AM), PF(MPF_ENCB_AO), C, PF(MPF_ENCB_AO_AM), part + oo, part + oo + (int64_t)C * C, C, C},
                {E->gv, C, PF(MPF_ENCB_GV_AM), PF(MPF_ENCB_X), C, PF(MPF_ENCB_X_AM), part + ov, part + ov + (int64_t)C * C, C, C}};
            if (ov + (int64_t)C * C + C != E->group_stride) return mpf::fail(MPF_E_SHAPE, "encoder_backward: group_stride does not match the four gradients");
            MPF_TRY(mpf_gemm3_nt_grouped_h2(it, 4, R, E->rps_group, E->group_stride, st));
            MPF_TRY(mpf_gemm3_nt_reduce(part, E->group_stride, nullptr, 0, nsg, PF(MPF_ENCB_WGRAD), nullptr, st));
        }
        g_in = g_out;
        gq_in = dq;
    }
    return mpf_ln_partial_reduce(E->ln_parts, E->ln_stride, R, kln, E->dgb_out, st);
}
