"""Host side of the HIP encoder: turns a HF `BertModel` / `T5EncoderModel` (the parameter
container the reference keeps in `DRModel.lm_q / lm_p`) into the packed device weights that
`om_encoder_forward` consumes, and launches it.

The HF module's own `forward` is never called on this path; it stays the owner of the
parameters so `state_dict()` / `save_pretrained()` keep the reference's checkpoint layout
(modeling/dense_retrieval_model.py:230-245).
"""
import ctypes as C
import os

import torch

from . import native as N

_ACT = {"gelu": N.ACT_GELU_ERF, "relu": N.ACT_RELU, "gelu_new": N.ACT_GELU_TANH,
        "gelu_pytorch_tanh": N.ACT_GELU_TANH}


def compute_dtype_code(model_args=None):
    """The compute format the caller asked for, the way the reference asks: `--fp16` -> torch autocast in
    retriever/dense_retriever.py:76 (float16 on a GPU), or ModelArguments.dtype.  float16 and bfloat16 both run on the
    16-bit MFMA path at the same rate; float32 on the exact f32 MFMA path."""
    if torch.is_autocast_enabled():
        return N.OM_F16 if torch.get_autocast_dtype("cuda") == torch.float16 else N.OM_BF16
    dt = getattr(model_args, "dtype", None) if model_args is not None else None
    if dt in ("float16", "fp16"):
        return N.OM_F16
    if dt in ("bfloat16", "bf16"):
        return N.OM_BF16
    return N.OM_F32


def torch_dtype_of(code):
    return {N.OM_BF16: torch.bfloat16, N.OM_F16: torch.float16}.get(code, torch.float32)


def inference_code(model, code, seq_len):
    """float16 is served where it is built: BERT-family encoders with erf-GELU (three more mantissa bits than bfloat16 in every
    stored activation, same speed) and (round 5) T5 ENCODER stacks with ReLU / tanh-GELU feed-forwards -- the reference's `--fp16`
    is float16 autocast for every backbone (retriever/dense_retriever.py:76).  As there, nothing clamps: a T5 checkpoint whose
    feed-forward activations leave the float16 range overflows in the reference and here alike; OM_T5_F16=0 (or `--bf16` /
    dtype="bfloat16") serves such a checkpoint with the bfloat16 kernels.  Other activations: a 16-bit request means bfloat16."""
    if code != N.OM_F16:
        return code
    cfg = getattr(model, "config", None)
    if _arch_of(model) == "t5":
        ok = _ACT.get(getattr(cfg, "dense_act_fn", None)) in (N.ACT_RELU, N.ACT_GELU_TANH) and os.environ.get("OM_T5_F16", "1") != "0"
        return code if ok else N.OM_BF16
    if _ACT.get(getattr(cfg, "hidden_act", None)) != N.ACT_GELU_ERF:
        return N.OM_BF16
    return code


def training_code(code, model=None):
    """The compute format of a TRAINING step.  float16 (the reference's `--fp16` training: HF Trainer's torch.cuda.amp autocast +
    GradScaler, trainer/dense_trainer.py:141-149) is served for BERT-family erf-GELU encoders (round 5; the trainer scales the
    loss, openmatch_amd/trainer/dense_trainer.py) and (round 6) for T5 stacks with ReLU / tanh-GELU feed-forwards, encoder and
    decoder position alike -- under the same conditions as float16 inference (inference_code: as under the reference's autocast nothing
    clamps, OM_T5_F16=0 keeps T5 in bfloat16).  Other activations train in bfloat16; OM_TRAIN_F16=0 sends every float16 request
    there.  Without a model (callers that only name a format): the conservative bfloat16."""
    if code != N.OM_F16:
        return code
    if model is None or os.environ.get("OM_TRAIN_F16", "1") == "0":
        return N.OM_BF16
    return inference_code(model, code, 0)


class _Packed:
    """Device copies of one encoder's weights in one compute dtype + the ctypes views of them."""

    def __init__(self):
        self.keep = []       # tensors that must outlive the ctypes pointers
        self.cfg = {}
        self.weights = N.OmEncoderWeights()
        self.layers = None

    def dev(self, t, dtype, device, sources=None):
        """Device copy of `t` in `dtype` (an alias of `t` itself when it already is that).  `sources`: the parameters the
        buffer is made of, in row order (default: `t` itself when it is a parameter) -- recorded so that an optimizer which
        updates the parameters in place can refresh the copy in the same pass (openmatch_amd/optim.py)."""
        src = t
        t = t.detach().to(device=device, dtype=dtype).contiguous()
        self.keep.append(t)
        if sources is None and isinstance(src, torch.nn.Parameter):
            sources = [src]
        if sources:
            off = 0
            for s_ in sources:
                if t.data_ptr() + off * t.element_size() != s_.data_ptr():       # not an alias of the parameter's own storage
                    _register_shadow(s_, self, t, off)
                off += s_.numel()
        return t.data_ptr()


# parameter -> the packed copies of it that live in some _Packed: [(weakref to the _Packed, buffer, element offset)]
_SHADOWS = {}


def _register_shadow(param, pk, buf, offset):
    import weakref
    key = id(param)
    lst = _SHADOWS.setdefault(key, [])
    lst[:] = [e for e in lst if e[0]() is not None and e[3]() is param]
    lst.append((weakref.ref(pk), buf, int(offset), weakref.ref(param)))


def shadows_of(param):
    """Live packed copies of `param`: [(packed object, buffer tensor, element offset)]; a _Packed that was replaced in its
    cache (or whose model is gone) no longer counts."""
    out = []
    for ref, buf, off, pref in _SHADOWS.get(id(param), ()):
        pk = ref()
        if pk is not None and pref() is param and not getattr(pk, "retired", False):
            out.append((pk, buf, off))
    return out


def _arch_of(model):
    name = type(model).__name__
    if "T5" in name:
        return "t5"
    if name.startswith("Bert") or "Bert" in name and "Roberta" not in name:
        return "bert"
    if "Roberta" in name:          # RobertaModel, XLMRobertaModel: the BERT stack behind offset position ids
        return "bert"
    raise NotImplementedError(
        f"openmatch_amd has HIP encoders for BERT / RoBERTa and T5-encoder backbones; got {name}")


def position_offset(model):
    """RoBERTa-family models number positions from padding_idx + 1 (HF:models/roberta/modeling_roberta.py
    create_position_ids_from_input_ids: cumsum over non-pad tokens + padding_idx): token t of a right-padded sequence
    reads row t + padding_idx + 1 of the position table.  The kernels index positions from 0, so the table (and its
    gradient) is handed over starting at that row.  Padded positions read other rows than HF's (row padding_idx) -- they
    are masked out of attention and pooling, so no output depends on them.  0 for BERT."""
    if "Roberta" in type(model).__name__:
        pad = getattr(model.config, "pad_token_id", None)
        return (1 if pad is None else int(pad)) + 1
    return 0


def check_position_layout(model, ids, mask):
    """RoBERTa-family models only: the kernels read position row t (+ position_offset) for token t, HF reads
    cumsum(input_ids != pad)[t] + padding_idx (modeling_roberta.py create_position_ids_from_input_ids).  The two agree on
    every attended token exactly when no pad id precedes an attended token (right-padded text without pad ids inside
    it).  Anything else -- left padding, pad ids inside the text -- would silently read other rows than the reference:
    rejected here (one small device reduction + host read per call, RoBERTa only)."""
    if position_offset(model) == 0:
        return
    pad = getattr(model.config, "pad_token_id", None)
    pad = 1 if pad is None else int(pad)
    L = ids.shape[1]
    pos = torch.arange(L, device=ids.device)
    first_pad = torch.where(ids == pad, pos, L).amin(dim=1)
    last_attended = torch.where(mask != 0, pos, -1).amax(dim=1)
    if bool((last_attended > first_pad).any()):
        raise ValueError("RoBERTa position ids: a pad token precedes an attended token (left-padded input, or pad ids "
                         "inside the text); the HIP encoder numbers positions for right-padded inputs only")


def _pack_bert(model, code, device):
    cfg = model.config
    if getattr(cfg, "position_embedding_type", "absolute") != "absolute":
        raise NotImplementedError("only absolute position embeddings are supported")
    wd = torch_dtype_of(code)
    f32 = torch.float32
    pk = _Packed()
    emb = model.embeddings
    w = pk.weights
    w.word_emb = pk.dev(emb.word_embeddings.weight, f32, device)
    off = position_offset(model)
    w.pos_emb = pk.dev(emb.position_embeddings.weight, f32, device) + off * cfg.hidden_size * 4
    w.type_emb = pk.dev(emb.token_type_embeddings.weight, f32, device)
    w.emb_ln_g = pk.dev(emb.LayerNorm.weight, f32, device)
    w.emb_ln_b = pk.dev(emb.LayerNorm.bias, f32, device)
    layers = (N.OmLayerWeights * cfg.num_hidden_layers)()
    for i, layer in enumerate(model.encoder.layer):
        at, lw = layer.attention, layers[i]
        qkv_w = torch.cat([at.self.query.weight, at.self.key.weight, at.self.value.weight], 0)
        qkv_b = torch.cat([at.self.query.bias, at.self.key.bias, at.self.value.bias], 0)
        lw.qkv_w = pk.dev(qkv_w, wd, device, [at.self.query.weight, at.self.key.weight, at.self.value.weight])
        lw.qkv_b = pk.dev(qkv_b, f32, device, [at.self.query.bias, at.self.key.bias, at.self.value.bias])
        lw.o_w = pk.dev(at.output.dense.weight, wd, device)
        lw.o_b = pk.dev(at.output.dense.bias, f32, device)
        lw.ln1_g = pk.dev(at.output.LayerNorm.weight, f32, device)
        lw.ln1_b = pk.dev(at.output.LayerNorm.bias, f32, device)
        lw.ffn1_w = pk.dev(layer.intermediate.dense.weight, wd, device)
        lw.ffn1_b = pk.dev(layer.intermediate.dense.bias, f32, device)
        lw.ffn2_w = pk.dev(layer.output.dense.weight, wd, device)
        lw.ffn2_b = pk.dev(layer.output.dense.bias, f32, device)
        lw.ln2_g = pk.dev(layer.output.LayerNorm.weight, f32, device)
        lw.ln2_b = pk.dev(layer.output.LayerNorm.bias, f32, device)
    pk.layers = layers
    w.layers_host = C.cast(layers, C.POINTER(N.OmLayerWeights))
    act = cfg.hidden_act if isinstance(cfg.hidden_act, str) else "gelu"
    if act not in _ACT:
        raise NotImplementedError(f"activation {act!r} has no HIP epilogue")
    pk.cfg = dict(arch=N.ARCH_BERT, dtype=code, hidden=cfg.hidden_size, n_layers=cfg.num_hidden_layers,
                  n_heads=cfg.num_attention_heads, head_dim=cfg.hidden_size // cfg.num_attention_heads,
                  ffn=cfg.intermediate_size, vocab=cfg.vocab_size, max_pos=cfg.max_position_embeddings - off,
                  type_vocab=cfg.type_vocab_size, act=_ACT[act], ln_eps=float(cfg.layer_norm_eps),
                  rel_buckets=0, rel_max_dist=0)
    return pk


def _pack_t5(model, code, device):
    cfg = model.config
    if cfg.num_heads * cfg.d_kv != cfg.d_model:
        raise NotImplementedError("T5 with inner_dim != d_model is not supported")
    wd = torch_dtype_of(code)
    f32 = torch.float32
    pk = _Packed()
    enc = model.encoder
    w = pk.weights
    w.word_emb = pk.dev(enc.embed_tokens.weight, f32, device)
    w.final_ln_g = pk.dev(enc.final_layer_norm.weight, f32, device)
    w.rel_bias = pk.dev(enc.block[0].layer[0].SelfAttention.relative_attention_bias.weight, f32, device)
    layers = (N.OmLayerWeights * cfg.num_layers)()
    gated = bool(getattr(cfg, "is_gated_act", False))
    for i, block in enumerate(enc.block):
        sa, ff, lw = block.layer[0].SelfAttention, block.layer[1].DenseReluDense, layers[i]
        lw.qkv_w = pk.dev(torch.cat([sa.q.weight, sa.k.weight, sa.v.weight], 0), wd, device, [sa.q.weight, sa.k.weight, sa.v.weight])
        lw.o_w = pk.dev(sa.o.weight, wd, device)
        lw.ln1_g = pk.dev(block.layer[0].layer_norm.weight, f32, device)
        lw.ln2_g = pk.dev(block.layer[1].layer_norm.weight, f32, device)
        if gated:
            lw.ffn1_w = pk.dev(ff.wi_0.weight, wd, device)
            lw.ffn1g_w = pk.dev(ff.wi_1.weight, wd, device)
        else:
            lw.ffn1_w = pk.dev(ff.wi.weight, wd, device)
        lw.ffn2_w = pk.dev(ff.wo.weight, wd, device)
    pk.layers = layers
    w.layers_host = C.cast(layers, C.POINTER(N.OmLayerWeights))
    act = cfg.dense_act_fn
    if act not in _ACT:
        raise NotImplementedError(f"activation {act!r} has no HIP epilogue")
    pk.cfg = dict(arch=N.ARCH_T5, dtype=code, hidden=cfg.d_model, n_layers=cfg.num_layers,
                  n_heads=cfg.num_heads, head_dim=cfg.d_kv, ffn=cfg.d_ff, vocab=cfg.vocab_size,
                  max_pos=0, type_vocab=0, act=_ACT[act], ln_eps=float(cfg.layer_norm_epsilon),
                  rel_buckets=cfg.relative_attention_num_buckets,
                  rel_max_dist=cfg.relative_attention_max_distance)
    return pk


def _pack_t5_decoder(model, code, device):
    """Decoder-side weights of a T5Model / T5ForConditionalGeneration for om_t5_decoder_step (one decoder position:
    self-attention needs only v, o; cross-attention k | v fused to [2H, H])."""
    cfg = model.config
    wd = torch_dtype_of(code)
    f32 = torch.float32
    pk = _Packed()
    dec = model.decoder
    # the reference feeds decoder_input_ids = zeros([B, 1]) (dense_retrieval_model.py:138): token 0, whatever the
    # config's decoder_start_token_id says
    w = N.OmT5DecoderWeights()
    w.start_emb = pk.dev(dec.embed_tokens.weight[0], f32, device)
    w.final_ln_g = pk.dev(dec.final_layer_norm.weight, f32, device)
    layers = (N.OmT5DecoderLayer * len(dec.block))()
    gated = bool(getattr(cfg, "is_gated_act", False))
    for i, block in enumerate(dec.block):
        sa, ca, ff, lw = block.layer[0].SelfAttention, block.layer[1].EncDecAttention, block.layer[2].DenseReluDense, layers[i]
        lw.sa_v_w = pk.dev(sa.v.weight, wd, device)
        lw.sa_o_w = pk.dev(sa.o.weight, wd, device)
        lw.sa_ln_g = pk.dev(block.layer[0].layer_norm.weight, f32, device)
        lw.ca_q_w = pk.dev(ca.q.weight, wd, device)
        lw.ca_kv_w = pk.dev(torch.cat([ca.k.weight, ca.v.weight], 0), wd, device, [ca.k.weight, ca.v.weight])
        lw.ca_o_w = pk.dev(ca.o.weight, wd, device)
        lw.ca_ln_g = pk.dev(block.layer[1].layer_norm.weight, f32, device)
        if gated:
            lw.ffn1_w = pk.dev(ff.wi_0.weight, wd, device)
            lw.ffn1g_w = pk.dev(ff.wi_1.weight, wd, device)
        else:
            lw.ffn1_w = pk.dev(ff.wi.weight, wd, device)
        lw.ffn2_w = pk.dev(ff.wo.weight, wd, device)
        lw.ffn_ln_g = pk.dev(block.layer[2].layer_norm.weight, f32, device)
    pk.layers = layers
    w.layers_host = C.cast(layers, C.POINTER(N.OmT5DecoderLayer))
    w.n_layers = len(dec.block)
    pk.weights = w
    return pk


_PACK_CACHE_ATTR = "_openmatch_amd_packed"


class _PackCache(dict):
    """Per-module cache of packed device weights (ctypes structs + device buffers).  It hangs off the module's __dict__,
    so `copy.deepcopy(model)` and `torch.save(model)` meet it: a copy / a pickle gets an EMPTY cache (the weights are
    re-packed on first use) instead of failing on the ctypes pointers."""
    def __deepcopy__(self, memo):
        return _PackCache()

    def __reduce__(self):
        return (_PackCache, ())


def _version_key(model, head):
    mods = [model] + ([head] if head is not None else [])
    return tuple((p.data_ptr(), p._version) for m in mods for p in m.parameters())


def packed_weights(model, head, code, device):
    """Packed weights for (model, head, dtype, device), rebuilt only when a parameter changed."""
    cache = model.__dict__.setdefault(_PACK_CACHE_ATTR, _PackCache())
    key = (code, str(device), id(head))
    ver = _version_key(model, head)
    hit = cache.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    if hit is not None:
        hit[1].retired = True                   # its buffers are no longer anyone's weights: optimizers stop refreshing them
    pk = _pack_bert(model, code, device) if _arch_of(model) == "bert" else _pack_t5(model, code, device)
    pk.owner_cache, pk.owner_key = cache, key
    if head is not None:
        lin = head.linear
        pk.weights.head_w = pk.dev(lin.weight, torch.float32, device)
        pk.cfg.update(head_in=lin.in_features, head_out=lin.out_features)
    else:
        pk.cfg.update(head_in=0, head_out=0)
    cache[key] = (ver, pk)
    return pk


def _ensure_folded(pk, device):
    """LayerNorm-folded weights of the fused 16-bit INFERENCE path, once per weight version, in a buffer the packed
    object owns (include/openmatch_hip.h: om_encoder_fold_weights; 0 bytes when the configuration has no fused path).
    Done on first inference use, not at packing time: training repacks every step and never reads them."""
    if getattr(pk, "fold_done", False):
        return
    pk.fold_done = True
    lib = N.lib()
    cfg = N.OmEncoderConfig(pooling=N.POOL_NONE, normalize=0, **pk.cfg)
    nfold = lib.om_encoder_fold_bytes(C.byref(cfg))
    if nfold:
        with torch.cuda.device(device):
            # ONE blob per packed object for its whole life: an optimizer that refreshes the packed copies in place
            # (after_inplace_update) invalidates the fold, and the next inference use folds again INTO THE SAME BUFFER
            # (a fresh blob per step/eval cycle appended to pk.keep grew by ~170 MB per cycle at bert-base)
            blob = getattr(pk, "fold_blob", None)
            if blob is None or blob.numel() < nfold + 256 or blob.device != torch.device(device):
                blob = torch.empty(nfold + 256, dtype=torch.uint8, device=device)
                pk.fold_blob = blob
            ptr = blob.data_ptr() + (-blob.data_ptr()) % 256
            N.check(lib.om_encoder_fold_weights(C.byref(cfg), C.byref(pk.weights), C.c_void_p(ptr), nfold, N.stream_ptr(device)))
        pk.weights.folded = ptr


def after_inplace_update(refreshed, stale):
    """An optimizer rewrote parameters through raw pointers (no version bump).  `refreshed`: packed objects whose copies it
    rewrote in the same pass -- they stay valid, only what was DERIVED from them (LayerNorm-folded inference weights) is
    dropped; `stale`: packed objects it could not refresh -- evicted from their cache, re-packed on next use."""
    for pk in refreshed:
        if getattr(pk, "fold_done", False):
            pk.fold_done = False
            pk.weights.folded = None
    for pk in stale:
        pk.retired = True
        cache, key = getattr(pk, "owner_cache", None), getattr(pk, "owner_key", None)
        if cache is not None and cache.get(key, (None, None))[1] is pk:
            del cache[key]


def invalidate_packed(root):
    """Drop every packed-weight cache under `root` (an nn.Module tree): the next HIP forward re-packs from the parameters.
    For optimizers that update parameters without bumping their version counters (torch's fused AdamW does not) -- the
    cache's freshness test cannot see those updates."""
    for mod in root.modules():
        for attr in (_PACK_CACHE_ATTR, _PACK_CACHE_ATTR + "_dec"):
            cache = mod.__dict__.get(attr)
            if cache:
                for _ver, pk in cache.values():
                    pk.retired = True
                cache.clear()


_POOL = {None: N.POOL_NONE, "first": N.POOL_FIRST, "mean": N.POOL_MEAN}


def packed_rows_bound(mask):
    """om_encoder_forward_packed's row bound for an attention mask held on the HOST (the collator's output, before it is
    moved to the device): sum over sequences of (1 + index of the last unmasked token; L for an all-masked row),
    rounded up to whole 256-row tiles.  None when packing cannot apply (fewer than 512 rows)."""
    m = mask.cpu() != 0
    L = m.shape[1]
    last = torch.where(m.any(1), L - m.flip(1).to(torch.int8).argmax(1), torch.full((m.shape[0],), L))
    rows = (int(last.sum()) + 255) // 256 * 256
    return rows if rows >= 512 else None


TOKEN_ROWS_KEY = "om_token_rows"      # batch-dict entry (a Python int): the batch's token count as the HOST knows it -- see token_rows_of


def token_rows_of(mask):
    """Per sequence: 1 + index of the last unmasked token (L for an all-masked row), as a HOST int64 tensor [B], for an attention mask
    that still lives on the host -- what the packed-rows entries need to size their row bound without a device synchronisation (the
    sum is the batch's token count; per sequence so that a batch that is split into chunks or merged keeps its counts).  None for a
    device tensor."""
    if not torch.is_tensor(mask) or mask.is_cuda or mask.dim() != 2:
        return None
    m = mask != 0
    L = m.shape[1]
    return torch.where(m.any(1), L - m.flip(1).to(torch.int8).argmax(1), torch.full((m.shape[0],), L)).to(torch.int64)


def rows_bound_of(tokens):
    """Row bound of the packed entries for a token count (an int, or token_rows_of's per-sequence tensor): whole 256-row tiles, None
    below 512 rows."""
    if tokens is None:
        return None
    total = int(tokens.sum()) if torch.is_tensor(tokens) else int(tokens)
    rows = (total + 255) // 256 * 256
    return rows if rows >= 512 else None


LAST_CALL = {}       # what the most recent hip_encode ran on: {"rows": token rows of the contractions, "packed": bool} (tests, bench)


def packed_rows_apply(cfg, B, L, rows, want_hidden, pooling, gated=False):
    """Whether om_encoder_forward_packed takes this call (include/openmatch_hip.h states the same conditions) and pays:
    a 16-bit encoder on its fused path (BERT-family erf-GELU; T5 without a gated feed-forward), representations only, and
    at least one 256-row tile saved.
    OM_ENCODER_PACKED=0 keeps every batch on the padded entry."""
    if os.environ.get("OM_ENCODER_PACKED", "1") == "0" or want_hidden or pooling is None:
        return False
    if cfg.dtype not in (N.OM_BF16, N.OM_F16) or cfg.hidden % 256 or cfg.ffn % 256 or cfg.n_layers < 1 or L > 1024:
        return False
    if cfg.arch == N.ARCH_BERT and cfg.act != N.ACT_GELU_ERF:
        return False
    if cfg.arch == N.ARCH_T5 and gated:
        return False
    if not (rows % 256 == 0 and 512 <= rows <= (B * L) // 256 * 256 - 256):
        return False
    # the library's own view under the current run-time switches (an A/B switch that takes the fused path away degrades the batch to
    # the padded entry instead of failing the call)
    if not isinstance(cfg, N.OmEncoderConfig):
        cfg = N.OmEncoderConfig(**{f: getattr(cfg, f) for f, _ in N.OmEncoderConfig._fields_ if hasattr(cfg, f)})
    return bool(N.lib().om_encoder_packed_supported(C.byref(cfg), int(bool(gated)), B, L, rows))


def hip_encode(model, items, pooling, head, normalize, code, want_hidden=True, packed_rows=None):
    """(hidden [B,L,H], reps [B,D] f32) through om_encoder_forward.  `items` holds
    input_ids / attention_mask / optional token_type_ids as int64 device tensors.
    packed_rows (with want_hidden=False): run om_encoder_forward_packed over that many rows (packed_rows_bound of the
    mask, computed where the mask still lives on the host) instead of B * L padded ones."""
    if pooling not in _POOL:
        raise ValueError("Unknown pooling type: {}".format(pooling))
    ids = items["input_ids"]
    mask = items["attention_mask"]
    tti = items.get("token_type_ids") if hasattr(items, "get") else None
    if ids.dim() != 2:
        raise ValueError("input_ids must be [batch, length]")
    ids = ids.to(torch.int64).contiguous()
    mask = mask.to(device=ids.device, dtype=torch.int64).contiguous()
    if tti is not None:
        tti = tti.to(device=ids.device, dtype=torch.int64).contiguous()
    N.require_device(ids, mask, tti)
    check_position_layout(model, ids, mask)
    device = ids.device
    code = inference_code(model, code, ids.shape[1])
    pk = packed_weights(model, head, code, device)
    _ensure_folded(pk, device)
    cfg = N.OmEncoderConfig(pooling=_POOL[pooling], normalize=int(bool(normalize)), **pk.cfg)
    B, L = ids.shape
    H = cfg.hidden
    D = cfg.head_out if cfg.head_in > 0 else H
    lib = N.lib()
    gated = bool(getattr(getattr(model, "config", None), "is_gated_act", False))
    if packed_rows and not packed_rows_apply(cfg, B, L, int(packed_rows), want_hidden, pooling, gated):
        packed_rows = None
    LAST_CALL.update(rows=int(packed_rows) if packed_rows else B * L, packed=bool(packed_rows))
    with torch.cuda.device(device):
        if packed_rows:
            nbytes = lib.om_encoder_workspace_bytes_packed(C.byref(cfg), B, L, int(packed_rows))
        else:
            nbytes = lib.om_encoder_workspace_bytes(C.byref(cfg), B, L)
        ws_buf, ws_ptr = N.Workspace.get(device, nbytes, "encoder")
        hidden = None
        if want_hidden:
            hidden = torch.empty(B, L, H, device=device,
                                 dtype=torch_dtype_of(code))
        reps = torch.empty(B, D, device=device, dtype=torch.float32) if pooling is not None else None
        if packed_rows:
            N.check(lib.om_encoder_forward_packed(C.byref(cfg), C.byref(pk.weights), N.ptr(ids), N.ptr(mask),
                                                  N.ptr(tti), B, L, int(packed_rows), N.ptr(reps),
                                                  C.c_void_p(ws_ptr), nbytes, N.stream_ptr(device)))
            return None, reps
        N.check(lib.om_encoder_forward(C.byref(cfg), C.byref(pk.weights), N.ptr(ids), N.ptr(mask),
                                       N.ptr(tti), B, L, N.ptr(hidden), N.ptr(reps),
                                       C.c_void_p(ws_ptr), nbytes, N.stream_ptr(device)))
    return hidden, reps


def packed_decoder_weights(model, code, device):
    """Packed decoder-side weights for (model, dtype, device), rebuilt only when a parameter changed."""
    cache = model.__dict__.setdefault(_PACK_CACHE_ATTR + "_dec", _PackCache())
    key = (code, str(device))
    ver = _version_key(model, None)
    hit = cache.get(key)
    if hit is None or hit[0] != ver:
        if hit is not None:
            hit[1].retired = True
        hit = (ver, _pack_t5_decoder(model, code, device))
        hit[1].owner_cache, hit[1].owner_key = cache, key
        cache[key] = hit
    return hit[1]


def hip_t5_decoder_step(model, items, code):
    """Decoder hidden state [B, H] (f32) of a T5 encoder-decoder after ONE decoder position fed token 0 -- the
    reference's `model(**items, decoder_input_ids=zeros([B, 1])).last_hidden_state[:, 0]`
    (modeling/dense_retrieval_model.py:137-141).  Encoder through om_encoder_forward, decoder through om_t5_decoder_step."""
    if not hasattr(model, "decoder") or not hasattr(model, "encoder"):
        raise ValueError("an encoder-decoder T5 model is required")
    code = inference_code(model, code, 0)  # float16 where the T5 stack takes it (round 6: the decoder position too), else bfloat16
    enc_hidden, _ = hip_encode(model, items, None, None, False, code, want_hidden=True)
    device = enc_hidden.device
    mask = items["attention_mask"].to(device=device, dtype=torch.int64).contiguous()
    dpk = packed_decoder_weights(model, code, device)
    epk = packed_weights(model, None, code, device)
    cfg = N.OmEncoderConfig(pooling=N.POOL_NONE, normalize=0, **epk.cfg)
    B, L = mask.shape
    lib = N.lib()
    with torch.cuda.device(device):
        nbytes = lib.om_t5_decoder_workspace_bytes(C.byref(cfg), B, L)
        _buf, ws_ptr = N.Workspace.get(device, nbytes, "decoder")
        out = torch.empty(B, cfg.hidden, device=device, dtype=torch.float32)
        N.check(lib.om_t5_decoder_step(C.byref(cfg), C.byref(dpk.weights), N.ptr(enc_hidden), N.ptr(mask), B, L,
                                       N.ptr(out), C.c_void_p(ws_ptr), nbytes, N.stream_ptr(device)))
    return out


class _LinearF32(torch.autograd.Function):
    """y = x W^T in f32 with its backward on the device (om_gemm_nt / om_linear_f32_backward): the LinearHead and the two
    LM-head rows of monoT5 behind the T5 decoder position when it is trained."""
    @staticmethod
    def forward(ctx, x, weight):
        x32 = x.to(torch.float32).contiguous()
        w32 = weight.to(device=x.device, dtype=torch.float32).contiguous()
        ctx.save_for_backward(x32, w32)
        return hip_linear_f32(x32, w32)

    @staticmethod
    def backward(ctx, dy):
        x32, w32 = ctx.saved_tensors
        dy = dy.to(torch.float32).contiguous()
        dx = torch.empty_like(x32)
        dw = torch.zeros_like(w32)
        with torch.cuda.device(x32.device):
            N.check(N.lib().om_linear_f32_backward(N.ptr(dy), N.ptr(x32), N.ptr(w32), N.ptr(dw), N.ptr(dx), x32.shape[0],
                                                  w32.shape[0], x32.shape[1], N.stream_ptr(x32.device)))
        return dx, dw


def hip_linear_f32_autograd(x, weight):
    return _LinearF32.apply(x, weight)


def hip_linear_f32(x, weight):
    """x [B, K] f32 @ weight[N, K]^T on the device through om_gemm_nt (LinearHead / LM-head columns)."""
    x = x.to(torch.float32).contiguous()
    w = weight.detach().to(device=x.device, dtype=torch.float32).contiguous()
    out = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        N.check(N.lib().om_gemm_nt(N.OM_F32, N.ptr(x), x.shape[1], N.ptr(w), w.shape[1], N.OM_F32, N.ptr(out), w.shape[0],
                                   x.shape[0], w.shape[0], x.shape[1], None, None, 0, 0, N.stream_ptr(x.device)))
    return out
