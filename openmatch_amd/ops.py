"""torch.autograd glue around the HIP entry points that take part in training."""
import torch

from . import native as N


class _ContrastiveLoss(torch.autograd.Function):
    """loss = scale * mean_i CE(q_all[i] . p_all^T, i * n_psg), forward AND backward in one
    `om_contrastive_fwd_bwd` launch sequence; gradients exist only for this rank's rows
    (modeling/dense_retrieval_model.py:113-125 with the all_gather semantics of :247-258)."""

    @staticmethod
    def forward(ctx, q_local, p_local, q_all, p_all, n_psg, scale, q_row0, p_row0):
        q_all = q_all.detach().to(torch.float32).contiguous()
        p_all = p_all.detach().to(torch.float32).contiguous()
        N.require_device(q_all, p_all)
        Qg, d = q_all.shape
        Pg = p_all.shape[0]
        dev = q_all.device
        need_grad = q_local.requires_grad or p_local.requires_grad
        loss = torch.empty((), device=dev, dtype=torch.float32)
        scores = torch.empty(Qg, Pg, device=dev, dtype=torch.float32)
        ws = torch.empty(2 * Qg * Pg + Qg, device=dev, dtype=torch.float32)
        dq = torch.empty_like(q_local, dtype=torch.float32) if need_grad else None
        dp = torch.empty_like(p_local, dtype=torch.float32) if need_grad else None
        with torch.cuda.device(dev):
            N.check(N.lib().om_contrastive_fwd_bwd(
                N.ptr(q_all), N.ptr(p_all), Qg, Pg, d, int(n_psg), float(scale), int(q_row0),
                q_local.shape[0], int(p_row0), p_local.shape[0], N.ptr(loss), N.ptr(scores),
                N.ptr(dq), N.ptr(dp), N.ptr(ws), N.stream_ptr(dev)))
        ctx.save_for_backward(dq, dp)
        ctx.mark_non_differentiable(scores)
        return loss, scores

    @staticmethod
    def backward(ctx, g_loss, _g_scores):
        dq, dp = ctx.saved_tensors
        return (dq * g_loss if dq is not None else None, dp * g_loss if dp is not None else None,
                None, None, None, None, None, None)


def contrastive_loss(q_all, p_all, n_psg, scale, q_local, q_row0, p_local, p_row0):
    return _ContrastiveLoss.apply(q_local, p_local, q_all, p_all, n_psg, scale, q_row0, p_row0)


def encode_with_grad(model, head, items, pooling, normalize, code, training):
    from .train import encode_train  # HIP forward-with-saves + backward (train.hip)
    return encode_train(model, head, items, pooling, normalize, code, training)
