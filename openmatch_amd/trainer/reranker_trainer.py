"""Cross-encoder training loop (reference: trainer/reranker_trainer.py:16-73).

Same loop as DRTrainer (HIP forward + backward through `RRModel.encode`, one flat-bucket gradient
all-reduce per step when data-parallel, AdamW, linear schedule); the batch is the
(positive pairs, negative pairs) tuple of PairCollator and the model returns the pair loss."""
import torch

from .dense_trainer import DRTrainer


class RRTrainer(DRTrainer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._dist_loss_scale_factor = 1          # no cross-device negatives in pairwise re-ranking

    def compute_loss(self, model, inputs, return_outputs=False, **_unused):
        pos_pairs, neg_pairs = inputs
        outputs = model(pos_pairs=pos_pairs, neg_pairs=neg_pairs)
        return (outputs.loss, outputs) if return_outputs else outputs.loss

    @torch.no_grad()
    def prediction_step(self, model, inputs, prediction_loss_only, ignore_keys=None):
        """(loss, (pos scores, neg scores), None) for one evaluation batch."""
        inputs = self._prepare_inputs(inputs)
        model.eval()
        with self._autocast():
            loss, outputs = self.compute_loss(model, inputs, return_outputs=True)
        loss = loss.mean().detach()
        if prediction_loss_only:
            return loss, None, None
        return loss, (outputs.pos_pair_scores.detach(), outputs.neg_pair_scores.detach()), None
