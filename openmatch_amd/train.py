"""Training path of the encoder: `om_encoder_train_forward` / `om_encoder_train_backward` behind a
`torch.autograd.Function`, so `DRModel.forward(...).loss.backward()` fills `param.grad` of the HF
module's parameters exactly as autograd through HF's own forward would (reference:
modeling/dense_retrieval_model.py:89-131, trainer/dense_trainer.py:102-108), without ever
running that forward."""
import ctypes as C
import os

import torch

from . import native as N
from .encoder import (_POOL, _arch_of, check_position_layout, packed_decoder_weights, packed_weights, position_offset,
                      torch_dtype_of, training_code)


def _bert_params(model, head):
    """Parameters in the order the backward returns their gradients."""
    emb = model.embeddings
    ps = [emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight,
          emb.LayerNorm.weight, emb.LayerNorm.bias]
    for layer in model.encoder.layer:
        at = layer.attention
        ps += [at.self.query.weight, at.self.key.weight, at.self.value.weight,
               at.self.query.bias, at.self.key.bias, at.self.value.bias,
               at.output.dense.weight, at.output.dense.bias, at.output.LayerNorm.weight, at.output.LayerNorm.bias,
               layer.intermediate.dense.weight, layer.intermediate.dense.bias,
               layer.output.dense.weight, layer.output.dense.bias,
               layer.output.LayerNorm.weight, layer.output.LayerNorm.bias]
    if head is not None:
        ps.append(head.linear.weight)
    return ps


def _t5_params(model, head):
    """T5EncoderModel parameters in the order the backward returns their gradients."""
    enc = model.encoder
    gated = bool(getattr(model.config, "is_gated_act", False))
    ps = [enc.embed_tokens.weight, enc.final_layer_norm.weight,
          enc.block[0].layer[0].SelfAttention.relative_attention_bias.weight]
    for block in enc.block:
        sa, ff = block.layer[0].SelfAttention, block.layer[1].DenseReluDense
        ps += [sa.q.weight, sa.k.weight, sa.v.weight, sa.o.weight, block.layer[0].layer_norm.weight]
        ps += [ff.wi_0.weight, ff.wi_1.weight] if gated else [ff.wi.weight]
        ps += [ff.wo.weight, block.layer[1].layer_norm.weight]
    if head is not None:
        ps.append(head.linear.weight)
    return ps


def _encoder_grad_arena(model, head, cfg, device, extra=0):
    """One zero-filled f32 arena holding every encoder (and head) gradient buffer the backward ADDS into, `extra` more
    floats behind them for the caller (the T5 decoder's gradients), the OmEncoderGrads struct pointing into it, and the
    gradient views in _bert_params / _t5_params order."""
    H, F, nl = cfg.hidden, cfg.ffn, cfg.n_layers
    # one zero-filled arena for every gradient (the backward ADDS into it with f32 atomics)
    t5 = _arch_of(model) != "bert"
    n_head = cfg.head_out * cfg.head_in if cfg.head_in > 0 else 0
    if t5:
        gated = bool(getattr(model.config, "is_gated_act", False))
        n_emb = model.encoder.embed_tokens.weight.numel() + H + cfg.rel_buckets * cfg.n_heads
        n_layer = 3 * H * H + H * H + 2 * H + (2 if gated else 1) * F * H + H * F
    else:
        emb = model.embeddings
        n_emb = sum(t.numel() for t in (emb.word_embeddings.weight, emb.position_embeddings.weight,
                                        emb.token_type_embeddings.weight)) + 2 * H
        n_layer = 3 * H * H + 3 * H + H * H + H + 2 * H + F * H + F + H * F + H + 2 * H
    arena = torch.zeros(n_emb + nl * n_layer + n_head + 64 * (8 + 13 * nl) + extra, device=device, dtype=torch.float32)
    cursor = [0]
    g = N.OmEncoderGrads()

    def buf(field_owner, name, *shape):
        n = 1
        for d_ in shape:
            n *= d_
        t = arena[cursor[0]:cursor[0] + n].view(*shape)
        cursor[0] += (n + 63) // 64 * 64            # keep every buffer 256-byte aligned
        setattr(field_owner, name, t.data_ptr())
        return t
    layers = (N.OmLayerGrads * nl)()
    per_layer = []
    layer_bounds = []                       # arena span of every layer (grad_sync buckets)
    if t5:
        gw = buf(g, "word_emb", *model.encoder.embed_tokens.weight.shape)
        gfin = buf(g, "final_ln_g", H)
        grel = buf(g, "rel_bias", cfg.rel_buckets, cfg.n_heads)
        for l in range(nl):
            lg = layers[l]
            lo_ = cursor[0]
            d = dict(qkv_w=buf(lg, "qkv_w", 3 * H, H), o_w=buf(lg, "o_w", H, H), ln1_g=buf(lg, "ln1_g", H),
                     ffn1_w=buf(lg, "ffn1_w", F, H), ffn2_w=buf(lg, "ffn2_w", H, F), ln2_g=buf(lg, "ln2_g", H))
            if gated:
                d["ffn1g_w"] = buf(lg, "ffn1g_w", F, H)
            per_layer.append(d)
            layer_bounds.append((lo_, cursor[0]))
    else:
        gw = buf(g, "word_emb", *emb.word_embeddings.weight.shape)
        gp = buf(g, "pos_emb", *emb.position_embeddings.weight.shape)
        g.pos_emb = gp[position_offset(model):].data_ptr()        # RoBERTa: positions start at padding_idx + 1
        gt = buf(g, "type_emb", *emb.token_type_embeddings.weight.shape)
        gg = buf(g, "emb_ln_g", H)
        gb = buf(g, "emb_ln_b", H)
        for l in range(nl):
            lg = layers[l]
            lo_ = cursor[0]
            per_layer.append(dict(
                qkv_w=buf(lg, "qkv_w", 3 * H, H), qkv_b=buf(lg, "qkv_b", 3 * H), o_w=buf(lg, "o_w", H, H),
                o_b=buf(lg, "o_b", H), ln1_g=buf(lg, "ln1_g", H), ln1_b=buf(lg, "ln1_b", H),
                ffn1_w=buf(lg, "ffn1_w", F, H), ffn1_b=buf(lg, "ffn1_b", F), ffn2_w=buf(lg, "ffn2_w", H, F),
                ffn2_b=buf(lg, "ffn2_b", H), ln2_g=buf(lg, "ln2_g", H), ln2_b=buf(lg, "ln2_b", H)))
            layer_bounds.append((lo_, cursor[0]))
    g.layers_host = C.cast(layers, C.POINTER(N.OmLayerGrads))
    ghead = buf(g, "head_w", cfg.head_out, cfg.head_in) if cfg.head_in > 0 else None
    if t5:
        grads = [gw, gfin, grel]
        for d in per_layer:
            q, k, v = d["qkv_w"].split(H, dim=0)
            grads += [q, k, v, d["o_w"], d["ln1_g"], d["ffn1_w"]]
            if "ffn1g_w" in d:
                grads.append(d["ffn1g_w"])
            grads += [d["ffn2_w"], d["ln2_g"]]
    else:
        grads = [gw, gp, gt, gg, gb]
        for d in per_layer:
            q, k, v = d["qkv_w"].split(H, dim=0)
            qb, kb, vb = d["qkv_b"].split(H, dim=0)
            grads += [q, k, v, qb, kb, vb, d["o_w"], d["o_b"], d["ln1_g"], d["ln1_b"], d["ffn1_w"], d["ffn1_b"],
                      d["ffn2_w"], d["ffn2_b"], d["ln2_g"], d["ln2_b"]]
    if ghead is not None:
        grads.append(ghead)
    keep = (layers, per_layer)                   # ctypes arrays referenced by g
    return arena, g, grads, layer_bounds, cursor, buf, keep


class _EncoderTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, head, ids, mask, tti, pooling, normalize, code, p_hidden, p_attn, seed, rows, *params):
        """rows > 0: the packed-rows pair (om_encoder_train_forward_packed / _backward_packed) over that many rows."""
        device = ids.device
        pk = packed_weights(model, head, code, device)
        cfg = N.OmEncoderConfig(pooling=_POOL[pooling], normalize=int(bool(normalize)), **pk.cfg)
        B, L = ids.shape
        D = cfg.head_out if cfg.head_in > 0 else cfg.hidden
        lib = N.lib()
        rows = int(rows or 0)
        if rows and not lib.om_encoder_train_packed_supported(C.byref(cfg), B, L, rows):
            rows = 0                     # (a configuration or a switch the packed pair does not take: the padded pair, same results)
        LAST_CALL.update(rows=rows if rows else B * L, packed=bool(rows))
        with torch.cuda.device(device):
            ntape = lib.om_encoder_tape_bytes_packed(C.byref(cfg), B, L, rows) if rows else lib.om_encoder_tape_bytes(C.byref(cfg), B, L)
            tape = torch.empty(ntape + 256, dtype=torch.uint8, device=device)
            tape_ptr = tape.data_ptr() + (-tape.data_ptr()) % 256
            nws = (lib.om_encoder_train_workspace_bytes_packed(C.byref(cfg), B, L, rows) if rows
                   else lib.om_encoder_train_workspace_bytes(C.byref(cfg), B, L))
            _buf, ws_ptr = N.Workspace.get(device, nws, "train")
            reps = torch.empty(B, D, device=device, dtype=torch.float32)
            if rows:
                N.check(lib.om_encoder_train_forward_packed(
                    C.byref(cfg), C.byref(pk.weights), N.ptr(ids), N.ptr(mask), N.ptr(tti), B, L, rows,
                    float(p_hidden), float(p_attn), int(seed), C.c_void_p(tape_ptr), tape.numel() - 256,
                    N.ptr(reps), C.c_void_p(ws_ptr), nws, N.stream_ptr(device)))
            else:
                N.check(lib.om_encoder_train_forward(
                    C.byref(cfg), C.byref(pk.weights), N.ptr(ids), N.ptr(mask), N.ptr(tti), B, L,
                    float(p_hidden), float(p_attn), int(seed), C.c_void_p(tape_ptr), tape.numel() - 256,
                    N.ptr(reps), C.c_void_p(ws_ptr), nws, N.stream_ptr(device)))
        ctx.model, ctx.head, ctx.cfg, ctx.pk = model, head, cfg, pk
        ctx.ids, ctx.mask, ctx.tti, ctx.tape, ctx.tape_ptr = ids, mask, tti, tape, tape_ptr
        ctx.drop = (float(p_hidden), float(p_attn), int(seed))
        ctx.rows = rows
        ctx.n_params = len(params)
        return reps

    @staticmethod
    def backward(ctx, d_reps):
        cfg, pk, model, head = ctx.cfg, ctx.pk, ctx.model, ctx.head
        device = ctx.ids.device
        B, L = ctx.ids.shape
        nl = cfg.n_layers
        rows = ctx.rows
        arena, g, grads, layer_bounds, _cursor, _buf_fn, _keep = _encoder_grad_arena(model, head, cfg, device)
        d_reps = d_reps.to(torch.float32).contiguous()
        lib = N.lib()
        # data-parallel training: the trainer's GradSync (if one is active for this step) all-reduces the arena in
        # layer-group buckets while the backward runs -- one event per layer tells it when a bucket is complete
        from . import grad_sync
        sync = grad_sync.active()
        events = None
        with torch.cuda.device(device):
            if sync is not None:
                events = [torch.cuda.Event() for _ in range(nl + 1)]
                for e in events:
                    e.record(torch.cuda.current_stream(device))           # materialises the hipEvent_t handle
                handles = (C.c_void_p * (nl + 1))(*[e.cuda_event for e in events])
                N.check(lib.om_encoder_train_set_layer_events(handles, nl + 1))
            nws = (lib.om_encoder_train_workspace_bytes_packed(C.byref(cfg), B, L, rows) if rows
                   else lib.om_encoder_train_workspace_bytes(C.byref(cfg), B, L))
            _buf, ws_ptr = N.Workspace.get(device, nws, "train")
            if rows:
                N.check(lib.om_encoder_train_backward_packed(
                    C.byref(cfg), C.byref(pk.weights), N.ptr(ctx.ids), N.ptr(ctx.mask), N.ptr(ctx.tti), B, L, rows,
                    ctx.drop[0], ctx.drop[1], ctx.drop[2], C.c_void_p(ctx.tape_ptr), N.ptr(d_reps), C.byref(g),
                    C.c_void_p(ws_ptr), nws, N.stream_ptr(device)))
            else:
                N.check(lib.om_encoder_train_backward(
                    C.byref(cfg), C.byref(pk.weights), N.ptr(ctx.ids), N.ptr(ctx.mask), N.ptr(ctx.tti), B, L,
                    ctx.drop[0], ctx.drop[1], ctx.drop[2], C.c_void_p(ctx.tape_ptr), N.ptr(d_reps), C.byref(g),
                    C.c_void_p(ws_ptr), nws, N.stream_ptr(device)))
        if sync is not None:
            sync.reduce_arena(arena, layer_bounds, events)
        ctx.tape = None
        return (None,) * 12 + tuple(grads)


LAST_CALL = {"rows": 0, "packed": False}       # what the last training forward ran over (tests, tools)
LAST_TRAIN_CODE = None                          # the compute format (native.OM_*) the last encode_train / t5_decoder_state_train ran in


def encode_train(model, head, items, pooling, normalize, code, training, packed_rows=None):
    """(None, reps) with an autograd edge from `reps` to every encoder / head parameter.
    Dropout follows the HF config only in training mode (model.train()).
    packed_rows: a bound on the batch's token count known on the HOST (encoder.packed_rows_bound of the collator's lengths) --
    the step then runs over that many rows instead of B x L where the packed pair takes the configuration (OM_TRAIN_PACKED=0: never)."""
    global LAST_TRAIN_CODE
    code = LAST_TRAIN_CODE = training_code(code, model)
    ids = items["input_ids"].to(torch.int64).contiguous()
    mask = items["attention_mask"].to(device=ids.device, dtype=torch.int64).contiguous()
    tti = items.get("token_type_ids") if hasattr(items, "get") else None
    if tti is not None:
        tti = tti.to(device=ids.device, dtype=torch.int64).contiguous()
    N.require_device(ids, mask, tti)
    check_position_layout(model, ids, mask)
    cfg = model.config
    bert = _arch_of(model) == "bert"
    if bert:
        p_hidden = float(cfg.hidden_dropout_prob) if training else 0.0
        p_attn = float(cfg.attention_probs_dropout_prob) if training else 0.0
    else:                                   # T5: one dropout_rate for every site
        p_hidden = p_attn = float(cfg.dropout_rate) if training else 0.0
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p_hidden > 0 or p_attn > 0) else 0
    params = _bert_params(model, head) if bert else _t5_params(model, head)
    rows = int(packed_rows) if (packed_rows and os.environ.get("OM_TRAIN_PACKED", "1") != "0") else 0      # (T5 stacks too since round 6; the library has the last word)
    reps = _EncoderTrain.apply(model, head, ids, mask, tti, pooling, normalize, code, p_hidden, p_attn, seed, rows,
                               *params)
    return None, reps


# ---------------------------------------------------------------------------------------------------------------------
# T5 encoder-decoder: the encoder stack + ONE decoder position (reference: modeling/dense_retrieval_model.py:137-141,
# modeling/reranking_model.py:110-114 in train mode)
def _t5_decoder_params(model):
    """(parameters whose gradients the decoder backward produces, in its order; parameters that take part in HF's graph
    with an identically zero gradient: q / k of the self-attention over a single key and its relative-position table)."""
    dec = model.decoder
    gated = bool(getattr(model.config, "is_gated_act", False))
    ps, zs = [dec.final_layer_norm.weight], []
    for block in dec.block:
        sa, ca, ff = block.layer[0].SelfAttention, block.layer[1].EncDecAttention, block.layer[2].DenseReluDense
        ps += [sa.v.weight, sa.o.weight, block.layer[0].layer_norm.weight,
               ca.q.weight, ca.k.weight, ca.v.weight, ca.o.weight, block.layer[1].layer_norm.weight]
        ps += [ff.wi_0.weight, ff.wi_1.weight] if gated else [ff.wi.weight]
        ps += [ff.wo.weight, block.layer[2].layer_norm.weight]
        zs += [sa.q.weight, sa.k.weight]
    rel = getattr(dec.block[0].layer[0].SelfAttention, "relative_attention_bias", None)
    if rel is not None:
        zs.append(rel.weight)
    return ps, zs


class _T5EncDecTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, ids, mask, code, p, seed, n_enc, n_dec, *params):
        device = ids.device
        epk = packed_weights(model, None, code, device)
        dpk = packed_decoder_weights(model, code, device)
        cfg = N.OmEncoderConfig(pooling=_POOL["first"], normalize=0, **epk.cfg)
        B, L = ids.shape
        H = cfg.hidden
        nld = dpk.weights.n_layers
        lib = N.lib()
        with torch.cuda.device(device):
            tape = torch.empty(lib.om_encoder_tape_bytes(C.byref(cfg), B, L) + 256, dtype=torch.uint8, device=device)
            tape_ptr = tape.data_ptr() + (-tape.data_ptr()) % 256
            nws = lib.om_encoder_train_workspace_bytes(C.byref(cfg), B, L)
            _buf, ws_ptr = N.Workspace.get(device, nws, "train")
            enc = torch.empty(B, L, H, device=device, dtype=torch_dtype_of(code))
            N.check(lib.om_encoder_train_forward_hidden(
                C.byref(cfg), C.byref(epk.weights), N.ptr(ids), N.ptr(mask), None, B, L, float(p), float(p), int(seed),
                C.c_void_p(tape_ptr), tape.numel() - 256, N.ptr(enc), C.c_void_p(ws_ptr), nws, N.stream_ptr(device)))
            dtape = torch.empty(lib.om_t5_decoder_tape_bytes(C.byref(cfg), nld, B, L) + 256, dtype=torch.uint8, device=device)
            dtape_ptr = dtape.data_ptr() + (-dtape.data_ptr()) % 256
            ndws = lib.om_t5_decoder_train_workspace_bytes(C.byref(cfg), nld, B, L)
            _buf2, dws_ptr = N.Workspace.get(device, ndws, "decoder_train")
            out = torch.empty(B, H, device=device, dtype=torch.float32)
            N.check(lib.om_t5_decoder_train_forward(
                C.byref(cfg), C.byref(dpk.weights), N.ptr(enc), N.ptr(mask), B, L, float(p), int(seed) + 1,
                C.c_void_p(dtape_ptr), dtape.numel() - 256, N.ptr(out), C.c_void_p(dws_ptr), ndws, N.stream_ptr(device)))
        ctx.model, ctx.cfg, ctx.epk, ctx.dpk = model, cfg, epk, dpk
        ctx.ids, ctx.mask, ctx.enc = ids, mask, enc
        ctx.tape, ctx.tape_ptr, ctx.dtape, ctx.dtape_ptr = tape, tape_ptr, dtape, dtape_ptr
        ctx.drop = (float(p), int(seed))
        ctx.counts = (n_enc, n_dec, len(params) - n_enc - n_dec)
        return out

    @staticmethod
    def backward(ctx, d_out):
        model, cfg, epk, dpk = ctx.model, ctx.cfg, ctx.epk, ctx.dpk
        device = ctx.ids.device
        B, L = ctx.ids.shape
        H, F = cfg.hidden, cfg.ffn
        nld = dpk.weights.n_layers
        dec_ps, zero_ps = _t5_decoder_params(model)
        gated = bool(getattr(model.config, "is_gated_act", False))
        pad = lambda n: (n + 63) // 64 * 64
        per_dec = 5 * pad(H * H) + pad(2 * H * H) + 3 * pad(H) + (3 if gated else 2) * pad(F * H)
        extra = pad(H) + nld * per_dec + sum(pad(z.numel()) for z in zero_ps)
        arena, g, enc_grads, _bounds, cursor, buf, _keep = _encoder_grad_arena(model, None, cfg, device, extra=extra)
        dg = N.OmT5DecoderGrads()
        dg.start_emb = enc_grads[0][0].data_ptr()              # row 0 of the shared table's gradient (decoder_input_ids = 0)
        dlayers = (N.OmT5DecoderLayerGrads * nld)()
        dec_grads = [buf(dg, "final_ln_g", H)]
        for l in range(nld):
            lg = dlayers[l]
            v, o, ln0 = buf(lg, "sa_v_w", H, H), buf(lg, "sa_o_w", H, H), buf(lg, "sa_ln_g", H)
            q, kv, co, ln1 = buf(lg, "ca_q_w", H, H), buf(lg, "ca_kv_w", 2 * H, H), buf(lg, "ca_o_w", H, H), buf(lg, "ca_ln_g", H)
            k_, v_ = kv.split(H, dim=0)
            dec_grads += [v, o, ln0, q, k_, v_, co, ln1]
            dec_grads += [buf(lg, "ffn1_w", F, H), buf(lg, "ffn1g_w", F, H)] if gated else [buf(lg, "ffn1_w", F, H)]
            dec_grads += [buf(lg, "ffn2_w", H, F), buf(lg, "ffn_ln_g", H)]
        dg.layers_host = C.cast(dlayers, C.POINTER(N.OmT5DecoderLayerGrads))
        holder = N.OmLayerGrads()                               # any struct with a settable field: `buf` needs an owner
        zero_grads = [buf(holder, "qkv_w", *z.shape) for z in zero_ps]
        assert cursor[0] <= arena.numel()
        d_out = d_out.to(torch.float32).contiguous()
        d_enc = torch.empty_like(ctx.enc)
        lib = N.lib()
        p, seed = ctx.drop
        with torch.cuda.device(device):
            ndws = lib.om_t5_decoder_train_workspace_bytes(C.byref(cfg), nld, B, L)
            _buf2, dws_ptr = N.Workspace.get(device, ndws, "decoder_train")
            N.check(lib.om_t5_decoder_train_backward(
                C.byref(cfg), C.byref(dpk.weights), N.ptr(ctx.enc), N.ptr(ctx.mask), B, L, p, seed + 1,
                C.c_void_p(ctx.dtape_ptr), N.ptr(d_out), C.byref(dg), N.ptr(d_enc), C.c_void_p(dws_ptr), ndws,
                N.stream_ptr(device)))
            nws = lib.om_encoder_train_workspace_bytes(C.byref(cfg), B, L)
            _buf, ws_ptr = N.Workspace.get(device, nws, "train")
            N.check(lib.om_encoder_train_backward_hidden(
                C.byref(cfg), C.byref(epk.weights), N.ptr(ctx.ids), N.ptr(ctx.mask), None, B, L, p, p, seed,
                C.c_void_p(ctx.tape_ptr), N.ptr(d_enc), C.byref(g), C.c_void_p(ws_ptr), nws, N.stream_ptr(device)))
        ctx.tape = ctx.dtape = ctx.enc = None
        n_enc, n_dec, n_zero = ctx.counts
        assert (len(enc_grads), len(dec_grads), len(zero_grads)) == (n_enc, n_dec, n_zero)
        return (None,) * 8 + tuple(enc_grads) + tuple(dec_grads) + tuple(zero_grads)


def t5_decoder_state_train(model, items, code, training):
    """Decoder hidden state [B, H] (f32) after one decoder position fed token 0, with an autograd edge to every encoder
    and decoder parameter -- the training-mode counterpart of encoder.hip_t5_decoder_step."""
    if not hasattr(model, "decoder") or not hasattr(model, "encoder"):
        raise ValueError("an encoder-decoder T5 model is required")
    global LAST_TRAIN_CODE
    code = LAST_TRAIN_CODE = training_code(code, model)
    ids = items["input_ids"].to(torch.int64).contiguous()
    mask = items["attention_mask"].to(device=ids.device, dtype=torch.int64).contiguous()
    N.require_device(ids, mask, None)
    p = float(model.config.dropout_rate) if training else 0.0
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
    enc_ps = _t5_params(model, None)
    dec_ps, zero_ps = _t5_decoder_params(model)
    return _T5EncDecTrain.apply(model, ids, mask, code, p, seed, len(enc_ps), len(dec_ps), *enc_ps, *dec_ps, *zero_ps)
