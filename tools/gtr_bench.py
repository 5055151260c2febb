#!/usr/bin/env python
"""GTR-base-shaped encode throughput (BASELINE config 4: T5 encoder 12 x 768, relu FFN 3072, mean pooling,
768 -> 768 head, normalise; 128 tokens, bf16).   python tools/gtr_bench.py [--batch 1024] [--gated]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmatch_amd import native as N
from types import SimpleNamespace as NS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--gated", action="store_true")
    a = ap.parse_args()
    from transformers import T5Config, T5EncoderModel
    from openmatch.modeling import DRModelForInference, LinearHead
    dev = "cuda:0"
    torch.manual_seed(0)
    cfg = T5Config(d_model=768, d_ff=2048 if a.gated else 3072, num_layers=12, num_heads=12, d_kv=64,
                   feed_forward_proj="gated-gelu" if a.gated else "relu")
    lm = T5EncoderModel(cfg).eval()
    head = LinearHead(768, 768)
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", normalize=True, head_q=head, head_p=head,
                                model_args=NS(encoder_only=True, dtype="bfloat16")).to(dev).eval()
    ids = torch.randint(3, 32000, (a.batch, 128), device=dev)
    items = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    out = {}
    for fused in ("1", "0"):
        N.check(N.lib().om_debug_option(0, int(fused)))      # OM_OPT_ENCODER_FUSED_LN
        for _ in range(2):
            model(passage=items)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            model(passage=items)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
        out["fused_norm" if fused == "1" else "norm_kernels"] = {"passages_per_s": round(a.batch / dt, 1), "ms_per_step": round(dt * 1e3, 2)}
    print(json.dumps({"metric": "GTR-base-shaped encode passages/s (T5 encoder, 128 tokens, bf16)", "batch": a.batch,
                      "gated": a.gated, **out}))


if __name__ == "__main__":
    main()
