"""Cross-encoder re-ranker with the reference's `RRModel` surface
(src/openmatch/modeling/reranking_model.py:34-181): encoder over the concatenated (query, doc)
pair -> CLS / mean pooling -> `LinearHead(H, 1)`.  The whole scoring path is ONE
`om_encoder_forward` call (head_out = 1); BASELINE config 5 (bert-large, L = 162) runs the same
kernels as the bi-encoder at H = 1024.  The monoT5 encoder-decoder branch (:110-114) runs the encoder and one T5
decoder position (`om_t5_decoder_step`; in training `om_t5_decoder_train_*`) and reads two columns of the LM head.  Training is supported
for encoder-only backbones at sequence lengths the
HIP backward covers (L <= 256 in bfloat16, <= 192 in float32; the default pair length is 162)."""
import json
import logging
import os
from dataclasses import dataclass
from typing import Dict

import torch
from torch import Tensor, nn
from transformers import AutoModel, BatchEncoding, PreTrainedModel, T5EncoderModel
from transformers.modeling_outputs import ModelOutput

from ..encoder import TOKEN_ROWS_KEY, compute_dtype_code, hip_encode, rows_bound_of
from ..feed import is_packed, token_rows_bound, unpack_token_batch
from ..loss import rr_loss_functions
from ..ops import encode_with_grad
from .linear import LinearHead

logger = logging.getLogger(__name__)


@dataclass
class RROutput(ModelOutput):
    pos_pair_scores: Tensor = None
    neg_pair_scores: Tensor = None
    loss: Tensor = None


class RRModel(nn.Module):
    accepts_compact_batches = True     # encode() widens feed.py's compact batches itself (and reads their host-side lengths first)

    def __init__(self, lm: PreTrainedModel, head: nn.Module, feature: str = "last_hidden_state",
                 pooling: str = "first", pos_token: str = None, neg_token: str = None, tokenizer=None,
                 model_args=None, data_args=None, train_args=None):
        super().__init__()
        self.lm, self.head = lm, head
        self.feature, self.pooling = feature, pooling
        self.pos_token, self.neg_token, self.tokenizer = pos_token, neg_token, tokenizer
        self.pos_token_id = tokenizer.encode(pos_token, add_special_tokens=False)[0] if pos_token else None
        self.neg_token_id = tokenizer.encode(neg_token, add_special_tokens=False)[0] if neg_token else None
        self.model_args, self.data_args, self.train_args = model_args, data_args, train_args
        if train_args is not None:
            self.loss_fn_str = train_args.loss_fn
            self.loss_fn = rr_loss_functions[self.loss_fn_str]()
            self.margin = train_args.margin

    def _get_config_dict(self):
        return {"plm_backbone": {"type": type(self.lm).__name__, "feature": self.feature},
                "pooling": self.pooling, "pos_token": self.pos_token, "neg_token": self.neg_token}

    def forward(self, pos_pairs: Dict[str, Tensor] = None, neg_pairs: Dict[str, Tensor] = None):
        pos, neg = self.encode(pos_pairs), self.encode(neg_pairs)
        loss = self.loss_fn(pos, neg)           # (the reference passes margin= to losses that take none)
        return RROutput(loss=loss, pos_pair_scores=pos, neg_pair_scores=neg)

    def encode(self, items):
        if items is None:
            return None, None
        rows = None
        if is_packed(items):            # a batch straight from RRInferenceCollator (feed.py's compact wire format)
            rows = token_rows_bound(items)      # host-side lengths: the packed-rows encoder computes the real tokens only
            items = unpack_token_batch(items, next(self.lm.parameters()).device)
        items = BatchEncoding(items)
        if "T5" in type(self.lm).__name__ and not self.model_args.encoder_only:
            return self._encode_mono_t5(items)
        if self.pooling not in ("first", "mean"):
            raise ValueError("Unknown pooling type: {}".format(self.pooling))
        code = compute_dtype_code(self.model_args)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.lm.parameters()):
            tokens = items.get(TOKEN_ROWS_KEY) if hasattr(items, "get") else None      # (host-side token count: packed rows in training)
            return encode_with_grad(self.lm, self.head, items, self.pooling, False, code, self.training, packed_rows=rows_bound_of(tokens))[1]
        return hip_encode(self.lm, items, self.pooling, self.head, False, code, want_hidden=False, packed_rows=rows)[1]   # [B,1]

    def _encode_mono_t5(self, items):
        """monoT5 (reference :110-114): logits[:, 0, [neg_token, pos_token]] of a T5ForConditionalGeneration after one
        decoder position fed token 0 -> [B, 2]; the Reranker takes log_softmax(...)[:, 1].  With autograd on the training
        kernels run behind one autograd node and the two LM-head rows go through a differentiable HIP linear."""
        if self.pos_token_id is None or self.neg_token_id is None:
            raise ValueError("monoT5 scoring needs pos_token and neg_token")
        if not hasattr(self.lm, "lm_head"):
            raise ValueError("monoT5 scoring needs a T5ForConditionalGeneration (lm_head) model")
        from ..encoder import hip_linear_f32, hip_linear_f32_autograd, hip_t5_decoder_step
        code = compute_dtype_code(self.model_args)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.lm.parameters())
        has_dropout = self.training and getattr(self.lm.config, "dropout_rate", 0.0) > 0
        if needs_grad or has_dropout:
            from ..train import t5_decoder_state_train
            state = t5_decoder_state_train(self.lm, items, code, self.training)              # [B, H] f32, differentiable
            linear = hip_linear_f32_autograd
        else:
            state = hip_t5_decoder_step(self.lm, items, code)                                 # [B, H] f32
            linear = hip_linear_f32
        cfg = self.lm.config      # HF T5ForConditionalGeneration.forward: original T5 scales the state, v1.1 does not
        scale = cfg.scale_decoder_outputs if hasattr(cfg, "scale_decoder_outputs") else getattr(cfg, "tie_word_embeddings", True)
        if scale:                 # (transformers >= 5 keeps that bit in `scale_decoder_outputs`, 4.x in `tie_word_embeddings`)
            state = state * (cfg.d_model ** -0.5)
        cols = self.lm.lm_head.weight[[self.neg_token_id, self.pos_token_id]]
        return linear(state, cols)                                     # [B, 2]

    @classmethod
    def build(cls, model_args, data_args=None, train_args=None, tokenizer=None, **hf_kwargs):
        path = model_args.model_name_or_path
        # (reference :139-146: encoder-only T5 -> T5EncoderModel, any other T5 checkpoint -> T5ForConditionalGeneration
        #  (monoT5 reads its LM head), everything else -> AutoModel)
        if model_args.encoder_only:
            model_class = T5EncoderModel
        else:
            from transformers import AutoConfig, T5ForConditionalGeneration
            archs = getattr(AutoConfig.from_pretrained(path, **hf_kwargs), "architectures", None) or [""]
            model_class = T5ForConditionalGeneration if "T5" in archs[0] else AutoModel
        config = None
        if os.path.exists(os.path.join(path, "openmatch_config.json")):
            with open(os.path.join(path, "openmatch_config.json")) as f:
                config = json.load(f)
        lm = model_class.from_pretrained(path, **hf_kwargs)
        if os.path.isdir(path) and config is not None:
            head = LinearHead.load(ckpt_dir=path)
        else:
            head = LinearHead(model_args.projection_in_dim, 1)
        pick = lambda key, default: default if config is None else config[key]
        return cls(lm=lm, head=head,
                   feature=model_args.feature if config is None else config["plm_backbone"]["feature"],
                   pooling=pick("pooling", model_args.pooling), pos_token=pick("pos_token", model_args.pos_token),
                   neg_token=pick("neg_token", model_args.neg_token), tokenizer=tokenizer,
                   model_args=model_args, data_args=data_args, train_args=train_args)

    def save(self, output_dir: str):
        self.lm.save_pretrained(output_dir)
        self.head.save(output_dir)
        with open(os.path.join(output_dir, "openmatch_config.json"), "w") as f:
            json.dump(self._get_config_dict(), f, indent=4)
