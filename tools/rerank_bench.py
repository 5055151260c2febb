#!/usr/bin/env python
"""Cross-encoder scoring throughput (BASELINE config 5: bert-large, pairs of 162 tokens, bf16).
  python tools/rerank_bench.py [--pairs 512] [--steps 5]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    from transformers import BertConfig, BertModel
    from openmatch.modeling import RRModel, LinearHead
    dev = "cuda:0"
    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    lm = BertModel(cfg).eval()
    model = RRModel(lm=lm, head=LinearHead(1024, 1), pooling="first",
                    model_args=NS(encoder_only=False, dtype="bfloat16")).to(dev).eval()
    L = 162
    ids = torch.randint(1000, 30000, (a.pairs, L), device=dev)
    items = {"input_ids": ids, "attention_mask": torch.ones_like(ids),
             "token_type_ids": (torch.arange(L, device=dev)[None, :] >= 34).long().expand(a.pairs, L).contiguous()}
    with torch.no_grad():
        for _ in range(2):
            model.encode(items)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            model.encode(items)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    gflop = 24 * (24 * L * 1024 * 1024 + 4 * L * L * 1024) / 1e9
    print(json.dumps({"metric": "cross-encoder pairs/s (bert-large, 162 tokens, bf16)", "pairs_per_s": round(a.pairs / dt, 1),
                      "ms_per_batch": round(dt * 1e3, 2), "pairs": a.pairs, "algorithmic_tflops": round(gflop * a.pairs / dt / 1e3, 1)}))


if __name__ == "__main__":
    main()
