"""torch.autograd glue around the HIP entry points that take part in training."""
import torch

from . import native as N


REDUCTIONS = {"mean": 0, "sum": 1, "none": 2}


def _contrastive_call(q_all, p_all, target, n_psg, reduction, row_grad, scale, q_local, q_row0, p_local, p_row0, want_grad):
    Qg, d = q_all.shape
    Pg = p_all.shape[0]
    dev = q_all.device
    loss = torch.empty((Qg,) if reduction == 2 else (), device=dev, dtype=torch.float32)
    scores = torch.empty(Qg, Pg, device=dev, dtype=torch.float32)
    ws = torch.empty(2 * Qg * Pg + Qg + 1, device=dev, dtype=torch.float32)
    dq = torch.empty_like(q_local, dtype=torch.float32) if want_grad else None
    dp = torch.empty_like(p_local, dtype=torch.float32) if want_grad else None
    with torch.cuda.device(dev):
        N.check(N.lib().om_contrastive_fwd_bwd_ex(
            N.ptr(q_all), N.ptr(p_all), Qg, Pg, d, N.ptr(target), int(n_psg), int(reduction), N.ptr(row_grad), float(scale),
            int(q_row0), q_local.shape[0], int(p_row0), p_local.shape[0], N.ptr(loss), N.ptr(scores),
            N.ptr(dq), N.ptr(dp), N.ptr(ws), N.stream_ptr(dev)))
    return loss, scores, dq, dp


class _ContrastiveLoss(torch.autograd.Function):
    """loss = scale * reduce_i CE(q_all[i] . p_all^T, target_i), forward AND backward in one
    `om_contrastive_fwd_bwd_ex` launch sequence; gradients exist only for this rank's rows
    (modeling/dense_retrieval_model.py:113-125 with the all_gather semantics of :247-258).
    reduction 'none' returns [Q] losses: their upstream gradients are only known in backward(), which then
    runs the gradient half of the same entry point."""

    @staticmethod
    def forward(ctx, q_local, p_local, q_all, p_all, n_psg, scale, q_row0, p_row0, target, reduction):
        q_all = q_all.detach().to(torch.float32).contiguous()
        p_all = p_all.detach().to(torch.float32).contiguous()
        N.require_device(q_all, p_all)
        if target is not None:
            target = target.to(device=q_all.device, dtype=torch.int64).contiguous()
            if target.shape != (q_all.shape[0],):
                raise ValueError("target must hold one class index per query row")
            # F.cross_entropy (reference loss.py:15 / modeling :122) raises on a class index outside [0, C) other than
            # ignore_index; the kernel indexes the score row with it, so check here (one host read, caller-supplied targets only)
            bad = (target != -100) & ((target < 0) | (target >= p_all.shape[0]))
            if bool(bad.any()):
                raise IndexError("Target {} is out of bounds.".format(int(target[bad][0])))
        need_grad = q_local.requires_grad or p_local.requires_grad
        fused = need_grad and reduction != 2
        loss, scores, dq, dp = _contrastive_call(q_all, p_all, target, n_psg, reduction, None, scale, q_local, q_row0,
                                                 p_local, p_row0, fused)
        ctx.deferred = None
        if need_grad and not fused:
            ctx.deferred = (q_all, p_all, target, n_psg, scale, q_row0, p_row0, q_local.shape, p_local.shape)
            ctx.save_for_backward()
        else:
            ctx.save_for_backward(dq, dp)
        ctx.mark_non_differentiable(scores)
        return loss, scores

    @staticmethod
    def backward(ctx, g_loss, _g_scores):
        if ctx.deferred is not None:
            q_all, p_all, target, n_psg, scale, q_row0, p_row0, qs, ps = ctx.deferred
            ql = torch.empty(qs, device=q_all.device)
            pl = torch.empty(ps, device=q_all.device)
            _, _, dq, dp = _contrastive_call(q_all, p_all, target, n_psg, 2, g_loss.to(torch.float32).contiguous(), scale,
                                             ql, q_row0, pl, p_row0, True)
        else:
            dq, dp = ctx.saved_tensors
            dq = dq * g_loss if dq is not None else None
            dp = dp * g_loss if dp is not None else None
        return (dq, dp, None, None, None, None, None, None, None, None)


def contrastive_loss(q_all, p_all, n_psg, scale, q_local, q_row0, p_local, p_row0, target=None, reduction="mean"):
    if reduction not in REDUCTIONS:
        raise ValueError("{} is not a valid value for reduction".format(reduction))      # torch's message
    return _ContrastiveLoss.apply(q_local, p_local, q_all, p_all, n_psg, scale, q_row0, p_row0, target, REDUCTIONS[reduction])


def encode_with_grad(model, head, items, pooling, normalize, code, training, packed_rows=None):
    from .train import encode_train  # HIP forward-with-saves + backward (train.hip)
    return encode_train(model, head, items, pooling, normalize, code, training, packed_rows=packed_rows)
