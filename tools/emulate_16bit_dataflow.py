#!/usr/bin/env python
"""CPU emulation of the rounding points of the fused 16-bit encoder dataflow (openmatch_amd/csrc/encoder.hip) in plain
torch: which stored tensors are rounded to the 16-bit format (embedding output, qkv, probabilities, context, the raw
pre-LayerNorm sums y1 / y2, the GELU output, the folded weights), with LayerNorm statistics taken from the f32 values as
the GEMM epilogues do.  Run for bfloat16 and float16, with and without an f32 residual stream, against HF BertModel in
fp32 and under torch.autocast(bfloat16) -- the numbers DESIGN.md 4.1 quotes (bf16 4.2e-5, f16 6e-7, reference autocast
1.6e-5).  Random-init bert-base, 8 ragged sequences of 128 tokens; ~1 minute on a few cores."""
import os, sys, torch, math
from transformers import BertConfig, BertModel
torch.manual_seed(0)
# EMU_INIT_RANGE=0.1: the spread-score model of tests/golden/config1_spread.npz (five-fold weights: larger pre-LayerNorm sums)
cfg = BertConfig(initializer_range=float(os.environ.get("EMU_INIT_RANGE", "0.02"))); m = BertModel(cfg).eval()
# make it less trivial than random init: scale some weights so activations have outliers
B, L = 8, 128
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30000, (B, L), generator=g); ids[:, 0] = 101
lens = torch.randint(40, 129, (B,), generator=g); mask = (torch.arange(L)[None] < lens[:, None]).long()
with torch.no_grad():
    ref = m(input_ids=ids, attention_mask=mask).last_hidden_state[:, 0]

def run(dt, resid_f32=False):
    q = (lambda x: x.to(dt).float()) if dt is not None else (lambda x: x)
    with torch.no_grad():
        e = m.embeddings
        x = e.LayerNorm(e.word_embeddings(ids) + e.position_embeddings(torch.arange(L))[None] + e.token_type_embeddings(torch.zeros_like(ids)))
        x = q(x)                      # embedding output stored 16-bit (post-LN)
        am = (1.0 - mask[:, None, None, :].float()) * -1e30
        prev_ln = None                # (gamma, beta) of the LN to apply to the raw stream y
        y = None
        def ln(v, w, b, eps=cfg.layer_norm_eps):
            mu = v.mean(-1, keepdim=True); var = v.var(-1, unbiased=False, keepdim=True)
            return (v - mu) / torch.sqrt(var + eps) * w + b
        inp_raw, inp_ln = x, None     # layer 0: input is x itself
        for l, layer in enumerate(m.encoder.layer):
            at = layer.attention
            def lin(a, mod): return a @ q(mod.weight).t() + mod.bias
            if l == 0:
                a_in = x; resid = x
            else:
                # A operand: raw 16-bit y2 normalised inside the GEMM (fold): stats from f32 values, operand rounded
                a_in = ln_apply(y2_16, st2, pl.output.LayerNorm)   # uses rounded y2, f32 stats
                resid = ln_apply(y2_res, st2, pl.output.LayerNorm)
            qh = q(lin(a_in, at.self.query)); kh = q(lin(a_in, at.self.key)); vh = q(lin(a_in, at.self.value))
            def sp(t): return t.view(B, L, 12, 64).transpose(1, 2)
            s = sp(qh) @ sp(kh).transpose(-1, -2) / 8.0 + am
            p = torch.softmax(s, -1)
            ctx = q((q(p) @ sp(vh)).transpose(1, 2).reshape(B, L, 768))
            y1 = lin(ctx, at.output.dense) + resid            # f32 in the epilogue
            st1 = (y1.mean(-1, keepdim=True), y1.var(-1, unbiased=False, keepdim=True))
            y1_16 = q(y1); y1_res = y1 if resid_f32 else y1_16
            def ln_apply(v, st, lnmod):
                return (v - st[0]) / torch.sqrt(st[1] + cfg.layer_norm_eps) * lnmod.weight + lnmod.bias
            a2 = ln_apply(y1_16, st1, at.output.LayerNorm)
            ff = q(torch.nn.functional.gelu(lin(a2, layer.intermediate.dense)))
            y2 = lin(ff, layer.output.dense) + ln_apply(y1_res, st1, at.output.LayerNorm)
            st2 = (y2.mean(-1, keepdim=True), y2.var(-1, unbiased=False, keepdim=True))
            y2_16 = q(y2); y2_res = y2 if resid_f32 else y2_16
            pl = layer
        out = ln_apply(y2_16, st2, pl.output.LayerNorm)
        return out[:, 0]

def report(name, o):
    cos = torch.nn.functional.cosine_similarity(o, ref, dim=-1)
    dd = (o @ o.t() - ref @ ref.t()).abs().max().item()
    print(f"{name:28s} 1-cos max {(1-cos).max().item():.2e} mean {(1-cos).mean().item():.2e}  max|ddot| {dd:.3e} (scale {float((ref@ref.t()).abs().max()):.0f})  max|emb err| {(o-ref).abs().max().item():.2e}")
report("f32 dataflow", run(None))
report("bf16", run(torch.bfloat16))
report("bf16 + f32 residual", run(torch.bfloat16, True))
report("f16", run(torch.float16))
report("f16 + f32 residual", run(torch.float16, True))
with torch.no_grad(), torch.autocast("cpu", torch.bfloat16):
    ab = m(input_ids=ids, attention_mask=mask).last_hidden_state[:, 0].float()
report("reference autocast bf16", ab)
try:
    with torch.no_grad(), torch.autocast("cpu", torch.float16):
        ah = m(input_ids=ids, attention_mask=mask).last_hidden_state[:, 0].float()
    report("reference autocast f16", ah)
except Exception as e:      # CPU autocast to float16 needs a recent torch
    print("reference autocast f16: not available here:", e)
