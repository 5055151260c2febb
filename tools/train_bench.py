#!/usr/bin/env python
"""Throughput of one contrastive training step at the reference's coCondenser shape
(BASELINE config 3 per GPU: 8 queries x 32 tok + 64 passages x 128 tok, bert-base, AdamW).
  python tools/train_bench.py [--precision bf16|f32] [--steps 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--ragged", action="store_true", help="ragged sequence lengths instead of full-length rows")
    ap.add_argument("--packed", type=int, default=1, help="with --ragged: 1 = the packed-rows training pair, 0 = the padded pair")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--optimizer", default="fused", help="fused: the product's FusedAdamW (clip + AdamW + weight refresh in one pass); "
                    "torch: clip_grad_norm_ + torch.optim.AdamW(fused=True) + a re-pack of the 16-bit weights")
    ap.add_argument("--passages", default="64x128", help="passages per step x tokens (e.g. 16x512: the same 8 192 passage tokens at 512 per passage, round 6)")
    ap.add_argument("--arch", default="bert", help="bert (bert-base) | t5 (GTR-base-shaped T5 encoder stack: d_model 768, d_ff 3072, 12 layers, ReLU)")
    ap.add_argument("--wgrad-wgs", type=int, default=0, help="weight-gradient workgroups per launch / 64 (0: the kernel's default, 5)")
    a = ap.parse_args()
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch.trainer import DRTrainer
    dev = "cuda:0"
    if a.wgrad_wgs:
        from openmatch_amd import native as N
        N.check(N.lib().om_debug_option(5, a.wgrad_wgs << 4))         # OM_OPT_WGRAD_DEBUG: bits 4.. = workgroups / 64
    torch.manual_seed(0)
    if a.arch == "t5":
        from transformers import T5Config, T5EncoderModel
        lm = T5EncoderModel(T5Config(d_model=768, d_ff=3072, num_layers=12, num_heads=12, d_kv=64, vocab_size=32128, feed_forward_proj="relu", dropout_rate=a.dropout))
    else:
        cfg = BertConfig(hidden_dropout_prob=a.dropout, attention_probs_dropout_prob=a.dropout)      # (max_position_embeddings 512)
        lm = BertModel(cfg)
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean" if a.arch == "t5" else "first",
                    model_args=NS(encoder_only=a.arch == "t5", dtype={"bf16": "bfloat16", "f16": "float16"}.get(a.precision, "float32")),
                    data_args=NS(train_n_passages=8 if a.passages == "64x128" else int(a.passages.split("x")[0]) // 8),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=8)).to(dev)
    g = torch.Generator().manual_seed(1)
    mk = lambda n, L: {"input_ids": torch.randint(1000, 30000, (n, L), generator=g), "attention_mask": torch.ones(n, L, dtype=torch.long)}
    if a.ragged:            # lengths ~ U{8..32} / U{16..128} (as bench.py's ragged encode batches), right-padded: what a collator hands over
        gl = torch.Generator().manual_seed(3)
        def rag(b, lo):
            L = b["input_ids"].shape[1]
            lens = torch.randint(lo, L + 1, (b["input_ids"].shape[0],), generator=gl)
            b["attention_mask"] = (torch.arange(L)[None, :] < lens[:, None]).long()
            return b
        mk_r = mk
        mk = lambda n, L: rag(mk_r(n, L), max(2, L // 8))
    ap_host = os.environ.get("TRAIN_BENCH_HOST_BATCH") == "1"       # 1: pageable host tensors, copied (synchronously) every step as before round 5
    pn, pl = (int(v) for v in a.passages.split("x"))
    batch = (mk(8, 32), mk(pn, pl))
    tokens = None
    if not ap_host:
        from openmatch_amd.encoder import TOKEN_ROWS_KEY, token_rows_of
        tokens = [token_rows_of(b["attention_mask"]) for b in batch]
        batch = tuple({k: v.to(dev) for k, v in b.items()} for b in batch)
        if a.ragged and a.packed:      # the token counts a trainer notes while the collator's batch is still on the host
            for b, n in zip(batch, tokens):
                b[TOKEN_ROWS_KEY] = n
    args = NS(device=dev, world_size=1, process_index=0, per_device_train_batch_size=8, negatives_x_device=False,
              learning_rate=5e-6, weight_decay=0.0, adam_beta1=0.9, adam_beta2=0.999, adam_epsilon=1e-8,
              gradient_accumulation_steps=1, max_grad_norm=1.0, fp16=a.precision == "f16", bf16=False)      # f16: float16 kernels + the dynamic loss scale
    trainer = DRTrainer(model=model, args=args)
    if a.optimizer == "torch":
        trainer.optimizer = torch.optim.AdamW(model.parameters(), lr=5e-6, weight_decay=0.0, fused=True)
    else:
        trainer.create_optimizer_and_scheduler(num_training_steps=10 ** 6)
    nap = float(os.environ.get("TRAIN_BENCH_HOST_SLEEP_US", "0")) * 1e-6      # probe: does the host have slack? (a busy wait per step)
    def step():
        if nap > 0:
            t_end = time.perf_counter() + nap
            while time.perf_counter() < t_end:
                pass
        loss = trainer.training_step(model, batch)
        trainer.optimizer_step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    flop = 3 * (8 * 5.474e9 + 64 * 22.347e9)
    print(json.dumps({"metric": "contrastive train steps/s (8 q x 32 + 64 p x 128, %s, fwd+bwd+AdamW)" % ("GTR-base-shaped T5 encoder" if a.arch == "t5" else "bert-base"), "steps_per_s": round(1 / dt, 2),
                      "ms_per_step": round(dt * 1e3, 2), "precision": a.precision, "dropout": a.dropout, "wgrad_wgs": a.wgrad_wgs, "optimizer": a.optimizer,
                      "algorithmic_tflops": round(flop / dt / 1e12, 1), "loss": float(loss), "ragged": bool(a.ragged),
                      "tokens": None if tokens is None else [int(n.sum()) for n in tokens], "rows": __import__("openmatch_amd.train", fromlist=["LAST_CALL"]).LAST_CALL}))


if __name__ == "__main__":
    main()
