"""Bi-encoder wrapper with the reference's public surface
(src/openmatch/modeling/dense_retrieval_model.py): `DROutput`, `DRModel`, `DRModelForInference`.

Same constructor, `forward(query, passage) -> DROutput`, `encode*`, `build`, `save`,
`dist_gather_tensor`, checkpoint files and error messages.  What differs is underneath:
`encode` never calls the HF module; it hands the token ids to `om_encoder_forward`
(HIP: MFMA GEMMs, fused attention, LDS LayerNorm, pooling/head/normalise tail), the in-batch
negatives loss runs in `om_contrastive_fwd_bwd`, and cross-device negatives are exchanged with
one all-gather over RCCL.  There is no CPU or eager fallback.
"""
import copy
import json
import inspect
import logging
import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn
from transformers import AutoModel, BatchEncoding, PreTrainedModel, T5EncoderModel
from transformers.modeling_outputs import ModelOutput

from ..arguments import DataArguments, ModelArguments
from ..arguments import DRTrainingArguments as TrainingArguments
from ..encoder import compute_dtype_code, hip_encode
from ..feed import is_packed, token_rows_bound, unpack_token_batch
from ..encoder import TOKEN_ROWS_KEY, rows_bound_of
from ..ops import contrastive_loss, encode_with_grad
from .linear import LinearHead

logger = logging.getLogger(__name__)


@dataclass
class DROutput(ModelOutput):
    q_reps: Tensor = None
    p_reps: Tensor = None
    loss: Tensor = None
    scores: Tensor = None


class DRModel(nn.Module):
    def __init__(
            self,
            lm_q: PreTrainedModel,
            lm_p: PreTrainedModel,
            tied: bool = True,
            feature: str = "last_hidden_state",
            pooling: str = "first",
            head_q: nn.Module = None,
            head_p: nn.Module = None,
            normalize: bool = False,
            model_args: ModelArguments = None,
            data_args: DataArguments = None,
            train_args: TrainingArguments = None,
    ):
        super().__init__()
        self.tied = tied
        self.lm_q, self.lm_p = lm_q, lm_p
        self.head_q, self.head_p = head_q, head_p
        self.feature, self.pooling, self.normalize = feature, pooling, normalize
        self.model_args, self.data_args, self.train_args = model_args, data_args, train_args

        if train_args is not None and train_args.negatives_x_device:
            if not dist.is_initialized():
                raise ValueError('Distributed training has not been initialized for representation all gather.')
            self.process_rank = dist.get_rank()
            self.world_size = dist.get_world_size()

    def _get_config_dict(self):
        return {
            "tied": self.tied,
            "plm_backbone": {"type": type(self.lm_q).__name__, "feature": self.feature},
            "pooling": self.pooling,
            "linear_head": bool(self.head_q),
            "normalize": self.normalize,
        }

    accepts_compact_batches = True     # encode() widens feed.py's compact batches itself (and reads their host-side lengths first)

    # ------------------------------------------------------------------ forward
    def forward(self, query: Dict[str, Tensor] = None, passage: Dict[str, Tensor] = None):
        if self._one_pass_ok(query, passage):
            q_reps, p_reps = self._encode_one_pass(query, passage)
        else:
            q_reps, p_reps = self._reps_only(self.encode_query, query), self._reps_only(self.encode_passage, passage)
        if q_reps is None or p_reps is None:
            return DROutput(q_reps=q_reps, p_reps=p_reps)

        xdev = self.train_args.negatives_x_device      # (the reference dereferences train_args too)
        if xdev:
            q_all, p_all = self.dist_gather_tensor(q_reps), self.dist_gather_tensor(p_reps)
            q_row0, p_row0 = self.process_rank * q_reps.shape[0], self.process_rank * p_reps.shape[0]
        else:
            q_all, p_all, q_row0, p_row0 = q_reps, p_reps, 0, 0
        # loss = mean_i CE(q_i . P^T, i * n_psg); x world_size in training to undo DDP's mean
        scale = float(self.world_size) if (self.training and xdev) else 1.0
        loss, scores = contrastive_loss(q_all, p_all, self.data_args.train_n_passages, scale,
                                        q_reps, q_row0, p_reps, p_row0)
        return DROutput(loss=loss, scores=scores, q_reps=q_all, p_reps=p_all)

    # ------------------------------------------------------------------ encode
    def encode(self, items, model, head, want_hidden=True):
        """(hidden, reps) as the reference's encode.  `want_hidden=False` (what forward() passes: it only uses the
        representations) skips materialising the [B, L, H] hidden states -- reps are then pooled from an f32 final
        LayerNorm of just the rows pooling needs, and hidden is None."""
        if items is None:
            return None, None
        rows = None
        if is_packed(items):            # a batch straight from DRInferenceCollator (16-bit ids + lengths, feed.py): widen it here,
            rows = None if want_hidden else token_rows_bound(items)     # (its lengths are on the host: the bound of the packed-rows encoder)
            items = unpack_token_batch(items, next(model.parameters()).device)     # so `model(passage=batch)` works as with the reference's collator
        token_rows = items.get(TOKEN_ROWS_KEY) if hasattr(items, "get") else None
        if token_rows is not None:
            items = {k: v for k, v in items.items() if k != TOKEN_ROWS_KEY}
        items = BatchEncoding(items)
        if "T5" in type(model).__name__ and not self.model_args.encoder_only:
            return self._encode_t5_decoder(items, model, head)
        if self.feature != "last_hidden_state":
            # the reference cannot run any other feature either: it indexes hidden[:, 0, :] / mean-pools over a sequence axis
            # (modeling/dense_retrieval_model.py:143-149), which fails on HF's 2-D pooler_output
            raise NotImplementedError("only feature='last_hidden_state' is produced by the HIP encoder (the reference's own "
                                      "encode() fails on 2-D features such as pooler_output)")
        if self.pooling not in ("first", "mean"):
            raise ValueError("Unknown pooling type: {}".format(self.pooling))
        code = compute_dtype_code(self.model_args)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
        cfg = getattr(model, "config", None)
        has_dropout = self.training and (getattr(cfg, "hidden_dropout_prob", 0.0) > 0
                                         or getattr(cfg, "attention_probs_dropout_prob", 0.0) > 0
                                         or getattr(cfg, "dropout_rate", 0.0) > 0)       # T5
        # train-mode forward: with autograd on, with dropout, and for ANY forward of a model in training mode -- the gradient-cache
        # trainer's first, tape-less pass must produce the representations its second pass differentiates (the inference kernels'
        # differ from the training forward's in the last 16-bit digits, which a contrastive loss over near-equal scores amplifies)
        if needs_grad or has_dropout or (self.training and any(p.requires_grad for p in model.parameters())):
            return encode_with_grad(model, head, items, self.pooling, self.normalize, code,
                                    self.training, packed_rows=rows_bound_of(token_rows))
        if rows is None and not want_hidden:
            rows = rows_bound_of(token_rows)        # (a mask that was still on the host when the trainer moved the batch)
        return hip_encode(model, items, self.pooling, head, self.normalize, code, want_hidden=want_hidden, packed_rows=rows)

    def _encode_t5_decoder(self, items, model, head):
        """T5 encoder-decoder pooling (reference :137-141): one decoder position fed token 0, reps = its hidden state,
        then head / normalize.  With autograd on (or dropout active) the training pair om_encoder_train_*_hidden +
        om_t5_decoder_train_* runs behind one autograd node (openmatch_amd/train.py), the head through a differentiable
        HIP linear; otherwise the inference kernels."""
        from ..encoder import hip_linear_f32, hip_linear_f32_autograd, hip_t5_decoder_step
        code = compute_dtype_code(self.model_args)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
        has_dropout = self.training and getattr(model.config, "dropout_rate", 0.0) > 0
        if needs_grad or has_dropout:
            from ..train import t5_decoder_state_train
            reps = t5_decoder_state_train(model, items, code, self.training)
            linear = hip_linear_f32_autograd
        else:
            reps = hip_t5_decoder_step(model, items, code)
            linear = hip_linear_f32
        hidden = reps.unsqueeze(1)                       # [B, 1, H]: the decoder's last_hidden_state
        if head is not None:
            reps = linear(reps, head.linear.weight)
        if self.normalize:
            reps = torch.nn.functional.normalize(reps, dim=1)
        return hidden, reps

    def encode_passage(self, psg, want_hidden=True):
        return self.encode(psg, self.lm_p, self.head_p, want_hidden=want_hidden)

    def encode_query(self, qry, want_hidden=True):
        return self.encode(qry, self.lm_q, self.head_q, want_hidden=want_hidden)

    @staticmethod
    def _reps_only(encode_fn, items):
        """forward() goes through encode_query / encode_passage as the reference's does (:89-93), so a subclass that
        overrides them is honoured; the hidden states are skipped where the method takes `want_hidden` (an override
        written against the reference's signature is called as it is)."""
        if "want_hidden" in inspect.signature(encode_fn).parameters:
            return encode_fn(items, want_hidden=False)[1]
        return encode_fn(items)[1]

    # A tied bi-encoder runs the SAME weights over the queries and the passages (reference :89-93: two calls of one
    # module).  In a training step the query batch is a few short rows -- 8 x 32 tokens beside 64 x 128 -- and a second
    # forward + backward over it costs a full set of launches for 3 % of the tokens, plus a second gradient arena that
    # autograd then adds to the first, parameter by parameter.  So the training forward pads the queries to the passage
    # length (mask 0) and encodes both in ONE pass; rows are independent and padded keys are masked, so the
    # representations are the two-call ones (up to the order of floating-point sums inside attention).
    def _one_pass_ok(self, query, passage):
        if query is None or passage is None or not self.training or not torch.is_grad_enabled():
            return False
        if self.lm_q is not self.lm_p or self.head_q is not self.head_p:
            return False
        if set(query.keys()) != set(passage.keys()) or "input_ids" not in query:
            return False
        (bq, lq), (bp, lp) = query["input_ids"].shape, passage["input_ids"].shape
        return lq <= lp and 4 * bq <= bp          # padding adds at most a quarter of the passage tokens

    def _encode_one_pass(self, query, passage):
        bq, lq = query["input_ids"].shape
        lp = passage["input_ids"].shape[1]
        pad_id = getattr(getattr(self.lm_p, "config", None), "pad_token_id", None) or 0
        merged = {}
        for key, q in query.items():
            p_ = passage[key]
            if key == TOKEN_ROWS_KEY:          # (host-side token counts, DRTrainer._prepare_inputs): the merged batch has both
                as_t = lambda v: v if torch.is_tensor(v) else torch.tensor([int(v)])
                merged[key] = torch.cat([as_t(q), as_t(p_)])
                continue
            q = q.to(p_.device)
            if q.dim() == 2 and q.shape[1] == lq and lq < lp:
                fill = pad_id if key == "input_ids" else 0
                q = torch.nn.functional.pad(q, (0, lp - lq), value=fill)
            merged[key] = torch.cat([q.to(p_.dtype), p_], dim=0)
        _, reps = self.encode(merged, self.lm_p, self.head_p)
        return reps[:bq], reps[bq:]

    # ------------------------------------------------------------------ build / save
    @classmethod
    def build(cls, model_args: ModelArguments, data_args: DataArguments = None,
              train_args: TrainingArguments = None, **hf_kwargs):
        path = model_args.model_name_or_path
        model_class = T5EncoderModel if model_args.encoder_only else AutoModel
        config = None
        head_q = head_p = None
        cfg_file = os.path.join(path, "openmatch_config.json")
        if os.path.exists(cfg_file):
            with open(cfg_file) as f:
                config = json.load(f)

        if os.path.isdir(path) and config is not None:      # an OpenMatch checkpoint directory
            tied = config["tied"]
            if tied:
                logger.info("loading model weight from %s", path)
                lm_q = lm_p = model_class.from_pretrained(path, **hf_kwargs)
                if config["linear_head"]:
                    head_q = head_p = LinearHead.load(path)
            else:
                sub = lambda name: os.path.join(path, name)
                logger.info("loading query model weight from %s", sub("query_model"))
                lm_q = model_class.from_pretrained(sub("query_model"), **hf_kwargs)
                logger.info("loading passage model weight from %s", sub("passage_model"))
                lm_p = model_class.from_pretrained(sub("passage_model"), **hf_kwargs)
                if config["linear_head"]:
                    head_q = LinearHead.load(sub("query_head"))
                    head_p = LinearHead.load(sub("passage_head"))
        else:                                                # a plain Hugging Face model
            tied = not model_args.untie_encoder
            lm_q = model_class.from_pretrained(path, **hf_kwargs)
            lm_p = lm_q if tied else copy.deepcopy(lm_q)
            if model_args.add_linear_head:
                head_q = LinearHead(model_args.projection_in_dim, model_args.projection_out_dim)
                head_p = head_q if tied else copy.deepcopy(head_q)

        return cls(
            lm_q=lm_q, lm_p=lm_p, tied=tied,
            feature=model_args.feature if config is None else config["plm_backbone"]["feature"],
            pooling=model_args.pooling if config is None else config["pooling"],
            head_q=head_q, head_p=head_p,
            normalize=model_args.normalize if config is None else config["normalize"],
            model_args=model_args, data_args=data_args, train_args=train_args,
        )

    def save(self, output_dir: str):
        if self.tied:
            self.lm_q.save_pretrained(output_dir)
            if self.head_q is not None:
                self.head_q.save(output_dir)
        else:
            for name, lm, head in (("query", self.lm_q, self.head_q), ("passage", self.lm_p, self.head_p)):
                os.makedirs(os.path.join(output_dir, name + "_model"))
                lm.save_pretrained(os.path.join(output_dir, name + "_model"))
            if self.head_q is not None:
                # (the reference writes the heads into directories it never creates; create them)
                for name, head in (("query_head", self.head_q), ("passage_head", self.head_p)):
                    os.makedirs(os.path.join(output_dir, name), exist_ok=True)
                    head.save(os.path.join(output_dir, name))
        with open(os.path.join(output_dir, "openmatch_config.json"), "w") as f:
            json.dump(self._get_config_dict(), f, indent=4)

    # ------------------------------------------------------------------ collectives
    def dist_gather_tensor(self, t: Optional[torch.Tensor]):
        """Rank-major concatenation of every rank's `t`; only this rank's slot keeps its autograd
        history (reference :247-258).  One all_gather into a single buffer over RCCL/xGMI."""
        if t is None:
            return None
        t = t.contiguous()
        from ..comm import native_comm
        comm = native_comm(t.device) if t.is_cuda else None
        if comm is not None:                     # OPENMATCH_AMD_COMM=native: om_allgather_rows behind the C ABI
            out = comm.allgather_rows(t.detach())
        else:
            out = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype,
                              device=t.device)
            dist.all_gather_into_tensor(out, t.detach())
        if t.requires_grad:
            parts = list(out.split(t.shape[0], dim=0))
            parts[self.process_rank] = t
            out = torch.cat(parts, dim=0)
        return out


class DRModelForInference(DRModel):
    """No-grad variant (reference :261-282): `forward` returns only the representations."""

    @torch.no_grad()
    def encode_passage(self, psg, want_hidden=True):
        return super().encode_passage(psg, want_hidden=want_hidden)

    @torch.no_grad()
    def encode_query(self, qry, want_hidden=True):
        return super().encode_query(qry, want_hidden=want_hidden)

    @torch.no_grad()
    def forward(self, query: Dict[str, Tensor] = None, passage: Dict[str, Tensor] = None):
        q_reps, p_reps = self._reps_only(self.encode_query, query), self._reps_only(self.encode_passage, passage)
        return DROutput(q_reps=q_reps, p_reps=p_reps)
