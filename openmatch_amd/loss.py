"""Loss callables of the reference's GradCache path (src/openmatch/loss.py:7-38) on top of the
HIP contrastive kernel.  Same call signatures."""
import torch
from torch import Tensor
from torch import distributed as dist

from .ops import contrastive_loss


class SimpleContrastiveLoss:
    def __call__(self, x: Tensor, y: Tensor, target: Tensor = None, reduction: str = "mean"):
        """loss.py:9-15: `F.cross_entropy(x @ y.T, target, reduction=reduction)` with the in-batch target
        arange(0, Q * (P // Q), P // Q) when none is given."""
        n_psg = y.size(0) // x.size(0)
        loss, _ = contrastive_loss(x, y, n_psg, 1.0, x, 0, y, 0, target=target, reduction=reduction)
        return loss


class DistributedContrastiveLoss(SimpleContrastiveLoss):
    def __init__(self, n_target: int = 0, scale_loss: bool = True):
        assert dist.is_initialized(), "Distributed training has not been properly initialized."
        super().__init__()
        self.word_size = dist.get_world_size()
        self.rank = dist.get_rank()
        self.scale_loss = scale_loss

    def __call__(self, x: Tensor, y: Tensor, **kwargs):
        """loss.py:27-31: gather x and y over the ranks (own slot keeps its autograd), the simple loss on the
        gathered batch with the caller's `target=` / `reduction=`, times world_size when scale_loss."""
        gx, gy = self.gather_tensor(x), self.gather_tensor(y)
        n_psg = gy.size(0) // gx.size(0)
        scale = float(self.word_size) if self.scale_loss else 1.0
        loss, _ = contrastive_loss(gx, gy, n_psg, scale, x, self.rank * x.size(0), y, self.rank * y.size(0),
                                   target=kwargs.get("target"), reduction=kwargs.get("reduction", "mean"))
        return loss

    def gather_tensor(self, t):
        out = torch.empty((self.word_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.detach().contiguous())
        return out


# Pair losses of the re-ranker (reference loss.py:41-73).  They act on [B,1] score tensors: a few
# hundred scalar operations, plain torch.
import torch.nn.functional as _F


class MarginRankingLoss:
    def __init__(self, margin: float = 1.0):
        self.margin = margin

    def __call__(self, pos_scores: Tensor, neg_scores: Tensor):
        return torch.mean(_F.relu(self.margin - pos_scores + neg_scores))


class SoftMarginRankingLoss(MarginRankingLoss):
    def __call__(self, pos_scores: Tensor, neg_scores: Tensor):
        return torch.mean(_F.softplus(self.margin - pos_scores + neg_scores))


class BinaryCrossEntropyLoss:
    def __call__(self, pos_scores: Tensor, neg_scores: Tensor):
        bce = _F.binary_cross_entropy_with_logits
        return bce(pos_scores, torch.ones_like(pos_scores)) + bce(neg_scores, torch.zeros_like(neg_scores))


class CrossEntropyLoss:
    def __call__(self, pos_scores: Tensor, neg_scores: Tensor):
        ones = torch.ones(pos_scores.shape[0], dtype=torch.long, device=pos_scores.device)
        return _F.cross_entropy(pos_scores, ones) + _F.cross_entropy(neg_scores, torch.zeros_like(ones))


rr_loss_functions = {"mr": MarginRankingLoss, "smr": SoftMarginRankingLoss, "bce": BinaryCrossEntropyLoss,
                     "ce": CrossEntropyLoss}
