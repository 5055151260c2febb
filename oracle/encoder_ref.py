"""Plain-torch CPU restatement of the encoders the reference calls through
`lm(**items, return_dict=True)` (src/openmatch/modeling/dense_retrieval_model.py:143) and
of the pooling / head / normalise tail of `DRModel.encode` (:133-155).

Weights are passed as a HF `state_dict()` (name -> tensor) so the same tensors feed the HIP
path and this oracle.  `dtype` is torch.float32 (the reference's arithmetic) or torch.float64
(tie adjudication).  Test infrastructure only — see oracle/__init__.py.
"""
import math

import torch
import torch.nn.functional as F


def _lin(x, sd, name, dtype):
    w = sd[name + ".weight"].to(dtype)
    b = sd.get(name + ".bias")
    y = x @ w.t()
    return y + b.to(dtype) if b is not None else y


def _layer_norm(x, g, b, eps):
    # torch.nn.LayerNorm: biased variance over the last dim (HF:models/bert/modeling_bert.py:64)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _act(x, name):
    if name in ("gelu", "gelu_erf"):      # HF:activations.py GELUActivation -> erf form
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if name == "relu":
        return torch.relu(x)
    if name == "gelu_new":                # HF:activations.py NewGELUActivation
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))
    raise ValueError(name)


def _attention(q, k, v, n_heads, add_bias, scale):
    """softmax(q k^T * scale + add_bias) v ; q,k,v [B,L,H] -> [B,L,H]
    (HF:models/bert/modeling_bert.py:111-136 eager_attention_forward)."""
    B, L, H = q.shape
    dh = H // n_heads
    sp = lambda t: t.view(B, L, n_heads, dh).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-1, -2) * scale + add_bias
    p = torch.softmax(s, dim=-1)
    return (p @ sp(v)).transpose(1, 2).reshape(B, L, H)


def bert_forward(sd, cfg, input_ids, attention_mask, token_type_ids=None, dtype=torch.float32,
                 return_all=False):
    """HF BertModel.forward in eval mode -> last_hidden_state [B,L,H]
    (HF:models/bert/modeling_bert.py:68-108 embeddings, :374-418 layer, :623-684 model)."""
    B, L = input_ids.shape
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    g = lambda n: sd[n].to(dtype)
    x = (g("embeddings.word_embeddings.weight")[input_ids]
         + g("embeddings.token_type_embeddings.weight")[token_type_ids])
    x = x + g("embeddings.position_embeddings.weight")[torch.arange(L)]
    eps = cfg.layer_norm_eps
    x = _layer_norm(x, g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    # additive key mask (HF:modeling_attn_mask_utils / masking_utils): 0 keep, finfo.min drop
    ext = (1.0 - attention_mask[:, None, None, :].to(dtype)) * torch.finfo(dtype).min
    hs = [x]
    scale = 1.0 / math.sqrt(cfg.hidden_size // cfg.num_attention_heads)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        q = _lin(x, sd, p + "attention.self.query", dtype)
        k = _lin(x, sd, p + "attention.self.key", dtype)
        v = _lin(x, sd, p + "attention.self.value", dtype)
        ctx = _attention(q, k, v, cfg.num_attention_heads, ext, scale)
        a = _lin(ctx, sd, p + "attention.output.dense", dtype)
        x1 = _layer_norm(a + x, g(p + "attention.output.LayerNorm.weight"),
                         g(p + "attention.output.LayerNorm.bias"), eps)
        f = _act(_lin(x1, sd, p + "intermediate.dense", dtype), cfg.hidden_act)
        o = _lin(f, sd, p + "output.dense", dtype)
        x = _layer_norm(o + x1, g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)
        hs.append(x)
    return hs if return_all else x


def t5_relative_bucket(rel, num_buckets=32, max_distance=128):
    """T5Attention._relative_position_bucket, bidirectional (HF:models/t5/modeling_t5.py:217-262).
    `rel` = memory_position - query_position (LongTensor)."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(is_small, n, large)


def t5_forward(sd, cfg, input_ids, attention_mask, dtype=torch.float32):
    """HF T5EncoderModel.forward in eval mode -> last_hidden_state
    (HF:models/t5/modeling_t5.py: T5LayerNorm :50-72, FF :75-141, T5Attention :176-370, stack)."""
    B, L = input_ids.shape
    g = lambda n: sd[n].to(dtype)
    emb = "shared.weight" if "shared.weight" in sd else "encoder.embed_tokens.weight"
    x = g(emb)[input_ids]
    eps = cfg.layer_norm_epsilon
    rms = lambda t, w: t * torch.rsqrt((t ** 2).mean(-1, keepdim=True) + eps) * w
    pos = torch.arange(L)
    buckets = t5_relative_bucket(pos[None, :] - pos[:, None], cfg.relative_attention_num_buckets,
                                 cfg.relative_attention_max_distance)
    table = g("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight")  # [buckets, heads]
    bias = table[buckets].permute(2, 0, 1)[None]                                       # [1,h,L,L]
    bias = bias + (1.0 - attention_mask[:, None, None, :].to(dtype)) * torch.finfo(dtype).min
    gated = any(".wi_0." in k for k in sd)
    act = "gelu_new" if gated else ("relu" if cfg.dense_act_fn == "relu" else cfg.dense_act_fn)
    for l in range(cfg.num_layers):
        p = f"encoder.block.{l}.layer."
        n1 = rms(x, g(p + "0.layer_norm.weight"))
        q = n1 @ g(p + "0.SelfAttention.q.weight").t()
        k = n1 @ g(p + "0.SelfAttention.k.weight").t()
        v = n1 @ g(p + "0.SelfAttention.v.weight").t()
        ctx = _attention(q, k, v, cfg.num_heads, bias, 1.0)     # no 1/sqrt(d) in T5 (:197)
        x = x + ctx @ g(p + "0.SelfAttention.o.weight").t()
        n2 = rms(x, g(p + "1.layer_norm.weight"))
        if gated:
            h = _act(n2 @ g(p + "1.DenseReluDense.wi_0.weight").t(), act) * (
                n2 @ g(p + "1.DenseReluDense.wi_1.weight").t())
        else:
            h = _act(n2 @ g(p + "1.DenseReluDense.wi.weight").t(), act)
        x = x + h @ g(p + "1.DenseReluDense.wo.weight").t()
    return rms(x, g("encoder.final_layer_norm.weight"))


def mean_pooling(hidden, attention_mask):
    """src/openmatch/utils.py:233-235."""
    m = attention_mask.unsqueeze(-1).expand(hidden.size()).to(hidden.dtype)
    return torch.sum(hidden * m, 1) / torch.clamp(m.sum(1), min=1e-9)


def pool_head_normalize(hidden, attention_mask, pooling="first", head_weight=None, normalize=False):
    """Tail of DRModel.encode (modeling/dense_retrieval_model.py:145-154; linear.py:22-23)."""
    if pooling == "first":
        reps = hidden[:, 0, :]
    elif pooling == "mean":
        reps = mean_pooling(hidden, attention_mask)
    else:
        raise ValueError("Unknown pooling type: {}".format(pooling))
    if head_weight is not None:
        reps = reps @ head_weight.to(reps.dtype).t()
    if normalize:
        reps = F.normalize(reps, dim=1)
    return reps


def encode(sd, cfg, arch, items, pooling="first", head_weight=None, normalize=False,
           dtype=torch.float32):
    """(hidden, reps) exactly as DRModel.encode returns them (eval mode)."""
    if arch == "bert":
        hidden = bert_forward(sd, cfg, items["input_ids"], items["attention_mask"],
                              items.get("token_type_ids"), dtype)
    else:
        hidden = t5_forward(sd, cfg, items["input_ids"], items["attention_mask"], dtype)
    return hidden, pool_head_normalize(hidden, items["attention_mask"], pooling, head_weight, normalize)
