"""Generates tests/golden/train_t5_tiny_{relu,gated}.npz by EXECUTING THE REFERENCE: its
`DRModel.forward` in training mode (dropout 0) over HF `T5EncoderModel` (GTR shape: mean pooling,
LinearHead, normalise; `encoder_only=True`), loss.backward(), every parameter gradient.
Run:  cd /tmp && python /root/repo/oracle/make_golden_t5_train.py     (needs /root/reference)
Kept apart from make_golden.py so the committed fixtures of the other cases are not regenerated.
`--add-f16` (round 6): adds the reference's float16-autocast yardsticks to the committed fixtures (add_f16_yardsticks).
"""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from transformers import T5Config, T5EncoderModel  # noqa: E402

import datasets  # noqa: E402,F401
from oracle import flatip  # noqa: E402

faiss_stub = types.ModuleType("faiss")
faiss_stub.IndexFlatIP = flatip.IndexFlatIP
sys.modules["faiss"] = faiss_stub
import openmatch  # noqa: E402
assert openmatch.__file__.startswith(REF), openmatch.__file__
from openmatch.modeling import DRModel  # noqa: E402
from openmatch.modeling.linear import LinearHead  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
NS = types.SimpleNamespace
SEED = 20260925


def synth_t5(rng, n, L, vocab, lo_len):
    ids = np.zeros((n, L), np.int64)
    mask = np.zeros((n, L), np.int64)
    for i in range(n):
        ln = int(rng.integers(lo_len, L + 1))
        body = rng.integers(3, vocab, size=ln)
        body[-1] = 1                                    # </s>
        ids[i, :ln] = body
        mask[i, :ln] = 1
    return ids, mask


def case(tag, gated, rng):
    torch.manual_seed(SEED + (17 if gated else 13))
    cfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_heads=2, d_kv=64, vocab_size=600, dropout_rate=0.0,
                   feed_forward_proj="gated-gelu" if gated else "relu")
    lm = T5EncoderModel(cfg)
    with torch.no_grad():        # default T5 init leaves norms at 1 and the bias table tiny: make them matter
        for name, p in lm.named_parameters():
            if "layer_norm" in name:
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            elif "relative_attention_bias" in name:
                p.copy_(0.5 * torch.randn_like(p))
    head = LinearHead(128, 128)
    n_psg = 2
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True,
                    model_args=NS(encoder_only=True), data_args=NS(train_n_passages=n_psg),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=4))
    model.train()
    q_ids, q_mask = synth_t5(rng, 4, 32, cfg.vocab_size, 4)
    p_ids, p_mask = synth_t5(rng, 8, 100, cfg.vocab_size, 16)      # L = 100: not a multiple of 32
    q = {"input_ids": torch.from_numpy(q_ids), "attention_mask": torch.from_numpy(q_mask)}
    p = {"input_ids": torch.from_numpy(p_ids), "attention_mask": torch.from_numpy(p_mask)}
    o = model(query=q, passage=p)
    o.loss.backward()
    out = {"q_input_ids": q_ids, "q_attention_mask": q_mask, "p_input_ids": p_ids, "p_attention_mask": p_mask,
           "loss": o.loss.detach().numpy(), "scores": o.scores.detach().numpy(),
           "q_reps": o.q_reps.detach().numpy(), "p_reps": o.p_reps.detach().numpy(),
           "head_w": head.linear.weight.detach().numpy(), "g::head_w": head.linear.weight.grad.numpy(),
           "n_psg": np.array(n_psg)}
    for k, v in lm.state_dict().items():
        out["w::" + k] = v.detach().numpy()
    seen = set()
    for k, v in lm.named_parameters():                  # (shared.weight and embed_tokens.weight are one tensor)
        if v.grad is not None and id(v) not in seen:
            seen.add(id(v))
            out["g::" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
    print("wrote", tag, "loss", float(o.loss), "grads", sum(k.startswith("g::") for k in out))


def add_f16_yardsticks(tag, gated):
    """Round 6: the reference's OWN float16 mode on the committed fixture -- its training step under torch.autocast(float16) with
    a static loss scale of 1024 (what `--fp16` = torch.cuda.amp + GradScaler does between two scale updates,
    trainer/dense_trainer.py:141-149) -- recorded as per-tensor relative L2 deviations of its gradients from its fp32 gradients:
    the yardstick a float16 kernel path's gradients are held to (tests/test_gpu_parity.py).  The stored fp32 arrays are not
    regenerated: the model is rebuilt from the fixture's weights and its fp32 step re-checked against them first."""
    path = os.path.join(OUT, tag + ".npz")
    g = dict(np.load(path))
    cfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_heads=2, d_kv=64, vocab_size=600, dropout_rate=0.0,
                   feed_forward_proj="gated-gelu" if gated else "relu")

    def build():
        lm = T5EncoderModel(cfg)
        lm.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w::")})
        head = LinearHead(128, 128)
        head.linear.weight.data.copy_(torch.from_numpy(g["head_w"]))
        m = DRModel(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True,
                    model_args=NS(encoder_only=True), data_args=NS(train_n_passages=int(g["n_psg"])),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=4))
        return m.train(), lm, head
    q = {"input_ids": torch.from_numpy(g["q_input_ids"]), "attention_mask": torch.from_numpy(g["q_attention_mask"])}
    p = {"input_ids": torch.from_numpy(g["p_input_ids"]), "attention_mask": torch.from_numpy(g["p_attention_mask"])}

    def grads_of(lm, head):
        out, seen = {"head_w": head.linear.weight.grad.detach().clone()}, set()
        for k, v in lm.named_parameters():
            if v.grad is not None and id(v) not in seen:
                seen.add(id(v)); out[k] = v.grad.detach().clone()
        return out
    m, lm, head = build()
    o = m(query=q, passage=p); o.loss.backward()
    g32 = grads_of(lm, head)
    for k, v in g32.items():
        ref = g["g::" + k]
        assert np.allclose(v.numpy(), ref, rtol=1e-5, atol=1e-7), k
    scale = 1024.0
    m, lm, head = build()
    with torch.autocast("cpu", dtype=torch.float16):
        o16 = m(query=q, passage=p)
    (o16.loss.float() * scale).backward()
    g16 = {k: v / scale for k, v in grads_of(lm, head).items()}
    rel = {k: float((g16[k].double() - g32[k].double()).norm() / g32[k].double().norm().clamp_min(1e-30)) for k in g32}
    for k, r in rel.items():
        g["ac16rel::" + k] = np.array(r)
    g["ac16_loss"] = np.array(float(o16.loss))
    vals = sorted(rel.values())
    print(f"{tag}: reference float16-autocast gradients vs its fp32: rel-L2 median {vals[len(vals) // 2]:.2e} max {vals[-1]:.2e} "
          f"({max(rel, key=rel.get)}); loss {float(o16.loss):.5f} vs {float(g['loss']):.5f}")
    np.savez_compressed(path, **g)


if __name__ == "__main__":
    if "--add-f16" in sys.argv:
        add_f16_yardsticks("train_t5_tiny_relu", False)
        add_f16_yardsticks("train_t5_tiny_gated", True)
        sys.exit(0)
    rng = np.random.default_rng(SEED + 99)
    case("train_t5_tiny_relu", False, rng)
    case("train_t5_tiny_gated", True, rng)
