"""Shared builders for the tests: HF models re-created from golden fixtures, synthetic inputs."""
import types

import numpy as np
import torch
from transformers import BertConfig, BertModel, T5Config, T5EncoderModel

NS = types.SimpleNamespace


def tiny_bert_config(**kw):
    return BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                      vocab_size=600, max_position_embeddings=160, **kw)


def tiny_t5_config(gated=False):
    return T5Config(d_model=128, d_ff=256, num_layers=2, num_heads=2, d_kv=64, vocab_size=600,
                    feed_forward_proj="gated-gelu" if gated else "relu")


def model_from_golden(g, arch, gated=False, **cfg_kw):
    """HF module whose parameters are the ones stored in the fixture (`w::<state_dict key>`)."""
    if arch == "bert":
        cfg = tiny_bert_config(**cfg_kw)
        model = BertModel(cfg)
    else:
        cfg = tiny_t5_config(gated)
        model = T5EncoderModel(cfg)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return cfg, model.eval()


def items_from_golden(g, kind, device="cpu"):
    items = {"input_ids": torch.from_numpy(g[kind + "_input_ids"]).to(device),
             "attention_mask": torch.from_numpy(g[kind + "_attention_mask"]).to(device)}
    if kind + "_token_type_ids" in g.files:
        items["token_type_ids"] = torch.from_numpy(g[kind + "_token_type_ids"]).to(device)
    return items


def synth_tokens(rng, n, L, vocab=30522, lo_len=16, lo_id=1000):
    """MS-MARCO-shaped BERT inputs: [CLS] body [SEP] pad, real length ~ U{lo_len..L}."""
    ids = np.zeros((n, L), np.int64)
    mask = np.zeros((n, L), np.int64)
    for i in range(n):
        ln = int(rng.integers(lo_len, L + 1))
        body = rng.integers(lo_id, vocab, size=ln)
        body[0], body[-1] = 101, 102
        ids[i, :ln] = body
        mask[i, :ln] = 1
    return ids, mask
