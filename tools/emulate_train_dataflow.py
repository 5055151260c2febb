#!/usr/bin/env python
"""CPU emulation of the ROUNDING POINTS of the 16-bit training step (round 5), to attribute its distance from the reference's fp32
gradients on tests/golden/train_base.npz: the BERT forward restated in plain torch with a round-to-bf16 (value AND gradient: a
cast's backward is a cast) at the places where the HIP path stores 16 bits, each class of points switchable.

  python tools/emulate_train_dataflow.py            # prints, per configuration, the per-tensor rel-L2 distance from the fixture's
                                                    # fp32 gradients over the reference's encoder-only bf16-autocast yardstick
Configurations: `hip` = every point the kernels round at; then one class at a time kept in f32.
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import BertConfig, BertModel  # noqa: E402

FMT = torch.float16 if os.environ.get("EMU_FMT", "bf16") == "f16" else torch.bfloat16


class _RoundGrad(torch.autograd.Function):
    """value unchanged, gradient rounded: a 16-bit gradient stream under an f32 forward"""
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.to(FMT).float()


class _GeluTape(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f):
        Phi = 0.5 * (1.0 + torch.erf(f / math.sqrt(2.0)))
        ctx.save_for_backward((Phi + f * torch.exp(-0.5 * f * f) * 0.3989422804014327).to(FMT))
        return f * Phi

    @staticmethod
    def backward(ctx, g):
        return g * ctx.saved_tensors[0].float()


def r(t, on=True):
    """on = True: value and gradient rounded; "grad": the gradient only; False: neither"""
    if on == "grad":
        return _RoundGrad.apply(t)
    return t.to(FMT).float() if on else t


def layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def forward(sd, cfg, ids, mask, R):
    """R: dict of switches -- x (LayerNorm outputs as stored), xres (the residual adds read the ROUNDED LayerNorm output), y (pre-LayerNorm
    sums), act (qkv / ctx / gelu output), probs (attention probabilities fed to P V), final (last hidden state before pooling)."""
    B, L = ids.shape
    lin = lambda x, n: x @ r(sd[n + ".weight"]).t() + sd[n + ".bias"]
    x = sd["embeddings.word_embeddings.weight"][ids] + sd["embeddings.token_type_embeddings.weight"][torch.zeros_like(ids)]
    x = x + sd["embeddings.position_embeddings.weight"][torch.arange(L)]
    eps = cfg.layer_norm_eps
    x = layer_norm(x, sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps)
    ext = (1.0 - mask[:, None, None, :].float()) * -1e30
    nh, dh = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    sp = lambda t: t.view(B, L, nh, dh).transpose(1, 2)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        xs = r(x, R["x"])                                   # the stored layer input: GEMM operand
        xr = xs if R["xres"] is True else r(x, R["xres"])   # what the residual add reads
        q, k, v = (r(lin(xs, p + "attention.self." + n), R["act"]) for n in ("query", "key", "value"))
        s = r(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dh), "grad" if R.get("bwd") in (True, "ds") else False) + ext      # dS is stored in 16 bits
        pr = r(torch.softmax(s, dim=-1), R["probs"])
        ctx = r((pr @ sp(v)).transpose(1, 2).reshape(B, L, -1), R["act"])
        y1 = r(lin(ctx, p + "attention.output.dense") + xr, R["y"])
        x1 = layer_norm(y1, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
        x1s = r(x1, R["x"])
        x1r = x1s if R["xres"] is True else r(x1, R["xres"])
        f = lin(x1s, p + "intermediate.dense")
        if R.get("bwd") in (True, "gelu"):                 # gelu'(f) sits on the tape in 16 bits and the backward multiplies by it
            g = r(_GeluTape.apply(f), R["act"])
        else:
            g = r(0.5 * f * (1.0 + torch.erf(f / math.sqrt(2.0))), R["act"])
        y2 = r(lin(g, p + "output.dense") + x1r, R["y"])
        x = layer_norm(y2, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return r(r(x, R["final"]), "grad" if R.get("bwd") in (True, "pool") else False)[:, 0]      # pool_bwd writes the hidden-state gradient in 16 bits


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "train_base.npz"))
    torch.manual_seed(3)
    cfg = BertConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    sd = dict(lm.named_parameters())

    def items(prefix, L):
        ids = torch.from_numpy(g[prefix + "input_ids"].astype(np.int64))
        lens = torch.from_numpy(g[prefix + "len"].astype(np.int64))
        return ids, (torch.arange(L)[None, :] < lens[:, None]).long()
    qi, qm = items("q_", 32)
    pi, pm = items("p_", 128)
    names = [str(n) for n in g["grad_names"]]
    yard = g["yardstick"]
    base = dict(x=True, xres=True, y=True, act=True, probs=True, final=False)
    configs = {"hip (all rounding points, f32 pooled rows)": base,
               "hip + the backward's own 16-bit points (dS, gelu' tape, pooled-row gradient)": dict(base, bwd=True),
               "hip + bwd: dS only": dict(base, bwd="ds"),
               "hip + bwd: gelu' tape only": dict(base, bwd="gelu"),
               "hip + bwd: pooled-row gradient only": dict(base, bwd="pool"),
               "f32 forward residual stream, 16-bit gradient stream + bwd points except the pooled-row gradient": dict(base, xres="grad", y="grad", bwd="ds"),
               "hip + final hidden state rounded (round 4)": dict(base, final=True),
               "residual adds read the unrounded LayerNorm output": dict(base, xres=False),
               "pre-LayerNorm sums y kept in f32": dict(base, y=False),
               "both (the residual stream in f32, as autocast keeps it)": dict(base, xres=False, y=False),
               "residual stream f32 in the FORWARD, its gradient stream still 16-bit": dict(base, xres="grad", y="grad"),
               "+ unrounded probabilities": dict(base, xres=False, y=False, probs=False),
               "only weights rounded": dict(x=False, xres=False, y=False, act=False, probs=False, final=False)}
    only = os.environ.get("EMU_ONLY")
    for tag, R in configs.items():
        if only and only not in tag:
            continue
        for p_ in sd.values():
            p_.grad = None
        q = forward(sd, cfg, qi, qm, R)
        p = forward(sd, cfg, pi, pm, R)
        scores = q @ p.t()
        loss = torch.nn.functional.cross_entropy(scores, torch.arange(q.shape[0]) * int(g["n_psg"]))
        loss.backward()
        fac_t, fac_w = [], []
        for i, n in enumerate(names):
            if yard[i, 4] < 1e-6:
                continue
            got = sd[n].grad.double()
            if "rows::" + n in g.files:
                got = got[torch.from_numpy(g["rows::" + n].astype(np.int64))]
            ref = torch.from_numpy(g["g::" + n]).double()
            rel = float((got - ref).norm() / ref.norm())
            fac_t.append(rel / yard[i, 5]); fac_w.append(rel / yard[i, 0])
        fac_t, fac_w = np.array(fac_t), np.array(fac_w)
        print("%-62s loss %.6f | vs encoder-only autocast: median %.2f max %.2f | vs whole-forward autocast: median %.2f max %.2f"
              % (tag, float(loss), np.median(fac_t), fac_t.max(), np.median(fac_w), fac_w.max()), flush=True)


if __name__ == "__main__":
    main()
