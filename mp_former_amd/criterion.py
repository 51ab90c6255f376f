"""SetCriterion — mirror of mask2former/modeling/criterion.py (:21-65 dice / sigmoid-CE,
:73-87 uncertainty, :90-304 SetCriterion incl. the mask-piloted `*_dn` losses).  Same constructor,
same `forward(outputs, targets) -> dict` with the reference's 6 x (1 + #aux) keys, same `weight_dict`
attribute read by the caller (maskformer_model.py:226-231).

Scheduling (results identical for identical random draws): the reference walks the 10 outputs one
by one — matcher, CE, point-sampled BCE/dice, then the same for the MP (`_dn`) queries — with a
blocking `.cpu()` per image per output (matcher.py:149) and ~100 small kernels per output.  Here a
step is three batched stages over ALL outputs at once:
  1. matching: native GT sampling + native mask/dice cost + the native device solver (lsa.py: SciPy's
     algorithm, one wavefront per problem) — the step has NO device->host copy; MPF_DEVICE_LSA=0 takes
     the reference's route instead (one D2H copy + SciPy, matcher.py);
  2. mask losses: one importance-sampling pass (native sampling + native radix selection) and one
     fused native BCE/dice kernel over every matched / MP (prediction, target) pair of every output;
     its backward is one scatter kernel;
  3. class losses: one batched log-softmax over all outputs.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, _rng
from .dist import distributed, global_num_masks, global_num_masks_device
from .matcher import GTMasks
from ._h2d import upload
from .lsa import MAX_DIM as LSA_MAX_DIM, lsa_assign
from .mask_fused import FactoredMasks, PairPlanes
from .point_sample import MapSet, MaskLossSums, MaskLossSumsPlanes, sample_select_uncertain


def strided_stack(ts):
    """torch.stack for tensors that are equally-spaced views of one parent (the per-layer class logits
    of the batched prediction heads): returns (stacked [L, ...] VIEW of the parent, order) where
    ``order[i]`` is the index in ``ts`` of row i, or (torch.stack(ts), identity) when they are not.
    One as_strided node instead of L slice / transpose backward chains."""
    ident = list(range(len(ts)))
    base = ts[0]._base
    if base is None or len(ts) < 2:
        return torch.stack(ts), ident
    shape, stride = ts[0].shape, ts[0].stride()
    for t in ts:
        if t._base is not base or t.shape != shape or t.stride() != stride:
            return torch.stack(ts), ident
    offs = [t.storage_offset() for t in ts]
    order = sorted(ident, key=lambda i: offs[i])
    so = [offs[i] for i in order]
    delta = so[1] - so[0]
    if delta <= 0 or any(so[i + 1] - so[i] != delta for i in range(len(so) - 1)):
        return torch.stack(ts), ident
    return base.as_strided((len(ts),) + tuple(shape), (delta,) + tuple(stride), so[0]), order


_DT = {torch.float32: _lib.MPF_F32, torch.bfloat16: _lib.MPF_BF16}


def _native_tail():
    return os.environ.get("MPF_NATIVE_LOSS_TAIL", "1") == "1"


class _ClassLossFn(torch.autograd.Function):
    """ce[l] = F.cross_entropy(logits[l] (N*Q rows), target[l], weight) for all L outputs in one launch, and its gradient in
    one launch (csrc/criterion_tail.hip; criterion.py:123-139).  logits: [L, N, Q, C] view with unit class stride, fp32 or
    bf16 (the kernel reads it as it is and computes in fp32, like F.cross_entropy under autocast)."""

    @staticmethod
    def forward(ctx, logits, target, weight):
        L, N, Q, C = logits.shape
        dev = logits.device
        per_output = 1 if target.dim() == 3 else 0
        target = target.contiguous()
        weight = weight.float().contiguous()
        lse = torch.empty((3, L, N * Q), dtype=torch.float32, device=dev)    # log-sum-exp of the rows + the kernel's row scratch
        out = torch.empty((2, L), dtype=torch.float32, device=dev)           # ce, wsum
        with _lib.device_guard(dev):
            code = _lib.lib().mpf_class_loss_forward(
                logits.data_ptr(), _DT[logits.dtype], logits.stride(0), logits.stride(1), logits.stride(2), target.data_ptr(),
                per_output, weight.data_ptr(), L, N, Q, C, lse.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                _lib.stream_ptr(dev))
        _lib.check(code, "mpf_class_loss_forward")
        ctx.save_for_backward(logits, target, weight, lse, out)
        ctx.per_output = per_output
        return out[0]

    @staticmethod
    def backward(ctx, g):
        logits, target, weight, lse, out = ctx.saved_tensors
        L, N, Q, C = logits.shape
        dev = logits.device
        g = g.float().contiguous()
        d = torch.empty((L, N, Q, C), dtype=logits.dtype, device=dev)
        with _lib.device_guard(dev):
            code = _lib.lib().mpf_class_loss_backward(
                logits.data_ptr(), _DT[logits.dtype], logits.stride(0), logits.stride(1), logits.stride(2), target.data_ptr(),
                ctx.per_output, weight.data_ptr(), L, N, Q, C, lse.data_ptr(), out[1].data_ptr(), g.data_ptr(), d.data_ptr(),
                _lib.stream_ptr(dev))
        _lib.check(code, "mpf_class_loss_backward")
        return d, None, None


class _MaskLossFinalizeFn(torch.autograd.Function):
    """(loss_mask[g], loss_dice[g]) of every group from the per-pair point sums, one launch each way
    (csrc/criterion_tail.hip; criterion.py:21-40, :48-65, :189-190).  runs int64 [G, 2] = (first pair, count) per group."""

    @staticmethod
    def forward(ctx, sums, runs, norm, points):
        n, G = sums.shape[0], norm.shape[0]
        sums = sums.contiguous()
        out = torch.empty((2, G), dtype=torch.float32, device=sums.device)
        with _lib.device_guard(sums.device):
            code = _lib.lib().mpf_mask_loss_finalize(sums.data_ptr(), runs.data_ptr(), norm.data_ptr(), n, G, points, out.data_ptr(),
                                                     _lib.stream_ptr(sums.device))
        _lib.check(code, "mpf_mask_loss_finalize")
        ctx.save_for_backward(sums, runs, norm)
        ctx.points = points
        return out

    @staticmethod
    def backward(ctx, g):
        sums, runs, norm = ctx.saved_tensors
        n, G = sums.shape[0], norm.shape[0]
        g = g.float().contiguous()
        d = torch.empty_like(sums)
        with _lib.device_guard(sums.device):
            code = _lib.lib().mpf_mask_loss_finalize_backward(sums.data_ptr(), runs.data_ptr(), norm.data_ptr(), n, G, ctx.points,
                                                              g.data_ptr(), d.data_ptr(),
                                                              _lib.stream_ptr(sums.device))
        _lib.check(code, "mpf_mask_loss_finalize_backward")
        return d, None, None, None


class LossDict(dict):
    """The 60-entry loss dict of the reference interface; the entries are views of a few per-output
    vectors, which ``SetCriterion.weighted_total`` weights directly."""

    def __init__(self):
        super().__init__()
        self.groups = []

    def add_group(self, names, vec):
        self.groups.append((list(names), vec))
        for i, k in enumerate(names):
            self[k] = vec[i]


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses, num_points, oversample_ratio,
                 importance_sample_ratio, dn_no_lb=False):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict = num_classes, matcher, weight_dict
        self.eos_coef, self.losses, self.dn_no_lb = eos_coef, losses, dn_no_lb
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.num_points, self.oversample_ratio = num_points, oversample_ratio
        self.importance_sample_ratio = importance_sample_ratio
        for loss in losses:
            assert loss in ("labels", "masks"), f"do you really want to compute {loss} loss?"

    # ---------------------------------------------------------------------------------------------
    def _class_losses(self, logits, target_classes):
        """logits [L,N,Q,K+1], target_classes [L,N,Q] (or [N,Q], shared) -> per-output weighted CE [L]
        (F.cross_entropy with class weights = sum w_y nll / sum w_y, criterion.py:123-139)."""
        L = logits.shape[0]
        if (_native_tail() and logits.is_cuda and logits.dim() == 4 and logits.stride(3) == 1 and logits.dtype in _DT
                and logits.shape[3] <= 256):
            return _ClassLossFn.apply(logits, target_classes, self.empty_weight)
        if target_classes.dim() == 2:
            target_classes = target_classes[None].expand(L, -1, -1)
        lsm = F.log_softmax(logits.float(), -1)
        nll = -torch.gather(lsm, 3, target_classes[..., None]).squeeze(3)
        w = self.empty_weight[target_classes]
        return (nll * w).flatten(1).sum(1) / w.flatten(1).sum(1)

    @staticmethod
    def _slot_layout(bi, N):
        """Planes of the step's pairs back to back, image by image: -> (slot of every pair, first slot / count per image)."""
        slot = np.zeros(len(bi), dtype=np.int64)
        first, count, at = [], [], 0
        for b in range(N):
            idx = np.flatnonzero(bi == b)
            slot[idx] = at + np.arange(len(idx))
            first.append(at); count.append(len(idx))
            at += len(idx)
        return slot, np.asarray(first, dtype=np.int32), np.asarray(count, dtype=np.int32)

    def forward(self, outputs, targets):
        dn_out = outputs["dn_out"]
        outs = [{"pred_logits": outputs["pred_logits"], "pred_masks": outputs["pred_masks"]}] + list(outputs.get("aux_outputs", []))
        L = len(outs)
        suffixes = [""] + [f"_{i}" for i in range(L - 1)]
        N, Q = outs[0]["pred_logits"].shape[:2]
        dev = outs[0]["pred_logits"].device
        K = self.num_classes
        P = self.num_points
        n_local = sum(len(t["labels"]) for t in targets)
        # criterion.py:224-237; across ranks the value stays on the device (no .item(): no host synchronisation)
        num_masks = global_num_masks_device(n_local, dev) if distributed() else global_num_masks(n_local, dev)

        use_dn = bool(self.training and dn_out)
        dn_outs = []
        if use_dn:
            dn_outs = [{"pred_logits": dn_out["pred_logits"], "pred_masks": dn_out["pred_masks"]}] + list(dn_out["aux_outputs"])
            max_num = dn_out["dn_args"]["max_num"]
            scalar = dn_out["dn_args"]["pad_size"] // max_num
            pad = dn_out["dn_args"]["pad_size"]

        gt = GTMasks(targets)
        # every prediction-map tensor of the step in one address space: [main_0..main_{L-1}, dn_0..dn_{L-1}]
        map_tensors = [o["pred_masks"] for o in outs] + [o["pred_masks"] for o in dn_outs]
        # the decoder's training path hands the predictions over as their factors (mask_fused.FactoredMasks): matching cost
        # and loss planes are then produced from (mask_embed, mask_features) directly and no [N, Q, h, w] map exists
        fm = map_tensors[0] if isinstance(map_tensors[0], FactoredMasks) else None
        if fm is not None and not (all(fm.same_factors(t) for t in map_tensors) and gt.tmax <= 128):
            map_tensors = [t.materialize() if isinstance(t, FactoredMasks) else t for t in map_tensors]
            for o, t in zip(outs + dn_outs, map_tensors):
                o["pred_masks"] = t
            fm = None
        ms = MapSet(map_tensors) if fm is None else None

        # ---- stage 1: all matchings -----------------------------------------------------------------
        # device solver (csrc/lsa.hip): the matched query / target of every pair stays on the GPU and the
        # step has no device->host copy; MPF_DEVICE_LSA=0 (or a problem larger than the kernel takes)
        # selects the reference's route: cost matrices to the host, SciPy, indices back up.
        tags = ["match" + s for s in suffixes]
        dev_lsa = (os.environ.get("MPF_DEVICE_LSA", "1") == "1" and max(Q, gt.tmax) <= LSA_MAX_DIM)
        firsts = gt.offsets
        if dev_lsa:
            C = self.matcher.cost_matrices(outs, targets, gt=gt, tags=tags, mapset=ms, map_index=list(range(L)) if ms else None)
            indices = None
        else:
            indices = self.matcher.match_many(outs, targets, gt=gt, tags=tags, mapset=ms, map_index=list(range(L)) if ms else None)

        # ---- pair lists (host): order = for each output: matched pairs, then MP pairs ----------------
        ti, bi, qi, gr, gid, over_parts, rand_parts = [], [], [], [], [], [], []
        num_uncertain = int(self.importance_sample_ratio * P)
        num_sampled = int(P * self.oversample_ratio)
        logits_main, order_main = strided_stack([o["pred_logits"] for o in outs])
        slot_of_output = {l: i for i, l in enumerate(order_main)}       # row of output l in logits_main
        problems = []                                                    # device solver: one row per (output, image)
        n_pos = 0
        if dev_lsa:
            labels_dev = torch.cat([t["labels"] for t in targets]) if gt.total else None
            Tmax = gt.tmax
        else:
            tc_main = np.full((L, N, Q), K, dtype=np.int64)
            if gt.total:   # one D2H copy for all labels (the stream was just drained by the matcher's copy)
                lab = torch.cat([t["labels"] for t in targets]).cpu().numpy()
                labels_host = [lab[firsts[b]:firsts[b + 1]] for b in range(N)]
            else:
                labels_host = [np.zeros(0, np.int64)] * N
        for l in range(L):
            n_l = 0
            for b in range(N):
                if dev_lsa:
                    k = min(Q, gt.counts[b])
                    if k:
                        problems.append([(l * N + b) * Q * Tmax, Q, gt.counts[b], Tmax, n_pos, firsts[b],
                                         0, 0, 0, 0, (slot_of_output[l] * N + b) * Q])
                    src = np.zeros(k, np.int64)      # filled in on the device
                    tgt = np.zeros(k, np.int64)
                else:
                    src, tgt = (x.numpy() for x in indices[l][b])
                    tc_main[l, b, src] = labels_host[b][tgt]
                ti.append(np.full(len(src), l)); bi.append(np.full(len(src), b)); qi.append(src)
                gr.append(firsts[b] + tgt); gid.append(np.full(len(src), l))
                n_l += len(src)
                n_pos += len(src)
            over_parts.append(("loss" + suffixes[l] + "_over", (n_l, num_sampled, 2)))
            rand_parts.append(("loss" + suffixes[l] + "_rand", (n_l, P - num_uncertain, 2)))
            if use_dn:
                n_d = 0
                for b in range(N):
                    T = gt.counts[b]
                    j = np.tile(np.arange(T), scalar)
                    slot = np.repeat(np.arange(scalar) * max_num, T) + j
                    ti.append(np.full(len(j), L + l)); bi.append(np.full(len(j), b)); qi.append(slot)
                    gr.append(firsts[b] + j); gid.append(np.full(len(j), L + l))
                    n_d += len(j)
                n_pos += n_d
                over_parts.append(("loss_dn" + suffixes[l] + "_over", (n_d, num_sampled, 2)))
                rand_parts.append(("loss_dn" + suffixes[l] + "_rand", (n_d, P - num_uncertain, 2)))
        cat = lambda xs: np.concatenate(xs).astype(np.int64) if xs else np.zeros(0, np.int64)  # noqa: E731
        ti, bi, qi, gr, gid = cat(ti), cat(bi), cat(qi), cat(gr), cat(gid)
        n_pairs = len(ti)
        G = 2 * L if use_dn else L
        losses = LossDict()

        # ---- index arrays of the pairs -> device (one upload each); the solver fills the matched slots ----
        tc_main_d = torch.full((L, N, Q), K, dtype=torch.int64, device=dev) if dev_lsa else None
        planes = None
        if n_pairs:
            gt_rows = upload(gr.astype(np.int32), dev)
            if fm is not None:
                # factors: pair i = embedding row p_offs[i] (elements from mask_embed's base) -> plane slot[i] of a compact
                # [slots, h*w] tensor produced AFTER the assignment (mask_fused.pair_planes); the solver writes the row offsets
                hw = fm.mf.shape[2] * fm.mf.shape[3]
                q0s = np.array([t.q0 for t in map_tensors], dtype=np.int64)
                p_offs = bi * fm.me.stride(0) + (q0s[ti] + qi) * fm.me.stride(1)
                slot, s_first, s_count = self._slot_layout(bi, N)
                g_offs = slot * hw
                row_stride = np.full(n_pairs, fm.me.stride(1), dtype=np.int64)
                if not dev_lsa:
                    # mpf_pair_planes_backward WRITES the d_embed row of a slot (include/mpformer_hip.h): rows must be distinct
                    assert len(np.unique(p_offs)) == n_pairs, "an embedding row is paired twice in one step"
            else:
                g_offs = ms.grad_offsets(ti, bi, qi)
                p_offs = ms.offsets(ti, bi, qi)
                row_stride = ms.s1[ms.base_of[ti]]
                if not dev_lsa:
                    assert len(np.unique(g_offs)) == n_pairs, "a prediction plane is paired twice in one step"
            up = upload(np.concatenate([p_offs, g_offs, gid]), dev)
            pred_offs, grad_offs, gid_d = up[:n_pairs], up[n_pairs:2 * n_pairs], up[2 * n_pairs:]
            if dev_lsa and problems:
                # matched slots: the offsets of query 0 were uploaded; the solver writes base + q * stride,
                # the ground-truth row, and the class target of the matched query
                pr = np.asarray(problems, dtype=np.int64)
                pos = pr[:, 4]
                pr[:, 6], pr[:, 7] = p_offs[pos], row_stride[pos]
                pr[:, 8], pr[:, 9] = g_offs[pos], (0 if fm is not None else ms.h * ms.w)
                into = {"cols": gt_rows, "a": pred_offs}
                if fm is None:
                    into["b"] = grad_offs           # (a pair's slot in the compact planes does not depend on the matched query)
                lsa_assign(C, pr, n_pairs, want_rows=False, scatter_dst=tc_main_d, scatter_src=labels_dev, into=into)
            if fm is not None:
                inv = np.zeros(n_pairs, dtype=np.int32)
                inv[slot] = np.arange(n_pairs, dtype=np.int32)
                i32 = upload(np.concatenate([inv, s_first, s_count]), dev)
                planes = PairPlanes.apply(fm.me, fm.mf, pred_offs, i32[:n_pairs], i32[n_pairs:n_pairs + N], i32[n_pairs + N:],
                                          n_pairs, int(s_count.max()))
                ms = MapSet([planes.view(1, n_pairs, fm.mf.shape[2], fm.mf.shape[3])])
                pred_offs = grad_offs               # from here on a pair is addressed by its plane

        # ---- stage 2: mask losses ----------------------------------------------------------------------
        if "masks" in self.losses:
            if n_pairs:
                with torch.no_grad():   # criterion.py:162-176: point selection carries no gradient
                    coords_over = _rng.rand_cat(over_parts, dev)
                    coords = sample_select_uncertain(ms, pred_offs, coords_over, num_uncertain, P)
                    if P - num_uncertain > 0:
                        coords[:, num_uncertain:] = _rng.rand_cat(rand_parts, dev)
                if planes is not None:
                    sums = MaskLossSumsPlanes.apply(ms, pred_offs, gt, gt_rows, coords, planes)
                else:
                    sums = MaskLossSums.apply(ms, pred_offs, grad_offs, gt, gt_rows, coords, *ms.bases)
                mult = [1.0] * L + ([float(scalar)] * L if use_dn else [])
                if torch.is_tensor(num_masks):
                    norm = num_masks * upload(mult, dev, torch.float32)
                else:
                    norm = upload([float(num_masks) * m for m in mult], dev, torch.float32)
                # the pairs of a group are one run of the pair list (laid out above: per output, matched then MP pairs)
                change = np.flatnonzero(np.diff(gid)) + 1
                starts = np.concatenate([[0], change])
                one_run_each = len(np.unique(gid[starts])) == len(starts)
                if _native_tail() and one_run_each:
                    runs = np.zeros((G, 2), dtype=np.int64)
                    runs[gid[starts], 0] = starts
                    runs[gid[starts], 1] = np.diff(np.concatenate([starts, [n_pairs]]))
                    fin = _MaskLossFinalizeFn.apply(sums, upload(runs, dev), norm, P)
                    g_mask, g_dice = fin[0], fin[1]
                else:
                    per_mask = sums[:, 0] / P                                                  # mean_p BCE (criterion.py:48-65)
                    per_dice = 1 - (2 * sums[:, 1] + 1) / (sums[:, 2] + sums[:, 3] + 1)        # criterion.py:21-40
                    z = torch.zeros(G, dtype=torch.float32, device=dev)
                    g_mask = z.index_add(0, gid_d, per_mask) / norm
                    g_dice = z.index_add(0, gid_d, per_dice) / norm
            else:
                # no ground truth anywhere: sums over empty sets (still consume the draws for RNG parity)
                _rng.rand_cat(over_parts, dev)
                _rng.rand_cat(rand_parts, dev)
                if fm is not None:
                    zero = (fm.me.sum() * 0.0 + fm.mf.sum() * 0.0).float()
                else:
                    zero = sum(t.sum() * 0.0 for t in map_tensors).float()
                g_mask = g_dice = zero.expand(G)
            names_m = ["loss_mask" + s_ for s_ in suffixes] + (["loss_mask_dn" + s_ for s_ in suffixes] if use_dn else [])
            names_d = ["loss_dice" + s_ for s_ in suffixes] + (["loss_dice_dn" + s_ for s_ in suffixes] if use_dn else [])
            losses.add_group(names_m, g_mask)
            losses.add_group(names_d, g_dice)

        # ---- stage 3: class losses ---------------------------------------------------------------------
        if "labels" in self.losses:
            if dev_lsa:
                tc_d = tc_main_d                                     # already in the row order of logits_main
            else:
                tc_d = upload(tc_main[order_main], dev)
            ce = self._class_losses(logits_main, tc_d)
            losses.add_group(["loss_ce" + suffixes[i] for i in order_main], ce)
            if use_dn:
                # criterion.py:249-258: slot j of every group <-> GT j
                if dev_lsa:
                    tc_dn_d = torch.full((N * pad,), K, dtype=torch.int64, device=dev)
                    if gt.total:
                        dst = np.concatenate([b * pad + s * max_num + np.arange(gt.counts[b]) for b in range(N) for s in range(scalar)])
                        srcp = np.concatenate([firsts[b] + np.arange(gt.counts[b]) for b in range(N) for s in range(scalar)])
                        ix = upload(np.concatenate([dst, srcp]).astype(np.int64), dev)
                        tc_dn_d[ix[:len(dst)]] = labels_dev[ix[len(dst):]]
                    tc_dn_d = tc_dn_d.view(N, pad)
                else:
                    tc_dn = np.full((N, pad), K, dtype=np.int64)
                    for b in range(N):
                        T = gt.counts[b]
                        for s in range(scalar):
                            tc_dn[b, s * max_num:s * max_num + T] = labels_host[b]
                    tc_dn_d = upload(tc_dn, dev)
                logits_dn, order = strided_stack([o["pred_logits"] for o in dn_outs])
                ce_dn = self._class_losses(logits_dn, tc_dn_d)
                losses.add_group(["loss_ce_dn" + suffixes[i] for i in order], ce_dn)
        if not use_dn:
            z = torch.as_tensor(0.0, device=dev)
            for s in suffixes:
                losses.update({"loss_mask_dn" + s: z, "loss_dice_dn" + s: z, "loss_ce_dn" + s: z})
        if self.dn_no_lb:
            for k in [k for k in losses if k.startswith("loss_ce_dn")]:
                del losses[k]
        return losses

    def weighted_total(self, losses):
        """sum_k weight_dict[k] * losses[k] in two kernels (the caller-side loop of
        maskformer_model.py:226-231 costs one multiply and one add per key)."""
        groups = getattr(losses, "groups", None)
        if groups:
            # the dict entries are views of a few vectors: weight the vectors (the per-key route costs a
            # select-backward = zeros + copy + add per key, ~180 launches per step)
            covered = {k for names, _ in groups for k in names}
            ws = [self.weight_dict.get(k, 0.0) if k in losses else 0.0 for names, _ in groups for k in names]
            vec = torch.cat([v.float() for _, v in groups]) if len(groups) > 1 else groups[0][1].float()
            total = torch.dot(vec, upload(ws, vec.device, torch.float32))
            rest = [k for k in losses if k in self.weight_dict and k not in covered]
            for k in rest:
                total = total + losses[k] * self.weight_dict[k]
            return total
        keys = [k for k in losses if k in self.weight_dict]
        vec = torch.stack([losses[k] for k in keys])
        w = upload([self.weight_dict[k] for k in keys], vec.device, vec.dtype)
        return (vec * w).sum()

    def __repr__(self):
        body = [f"matcher: {self.matcher.__repr__(_repr_indent=8)}", f"losses: {self.losses}",
                f"weight_dict: {self.weight_dict}", f"num_classes: {self.num_classes}", f"eos_coef: {self.eos_coef}",
                f"num_points: {self.num_points}", f"oversample_ratio: {self.oversample_ratio}",
                f"importance_sample_ratio: {self.importance_sample_ratio}"]
        return "\n".join(["Criterion " + self.__class__.__name__] + [" " * 4 + line for line in body])
