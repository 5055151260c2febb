#!/usr/bin/env python
"""Encode throughput at long passage lengths (bert-base, float16): passages/s at B x L for the attention kernels in use
(OM_ATTENTION_FAST bit 2 selects the first online-softmax kernel for L > 256).   python tools/long_encode_bench.py [--shapes 256x512,128x1024]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="256x512,128x1024,512x256")
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    dev = "cuda:0"
    torch.manual_seed(0)
    lm = BertModel(BertConfig(max_position_embeddings=1024)).eval()
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype=a.dtype)).to(dev).eval()
    out = {"dtype": a.dtype, "attention_fast": os.environ.get("OM_ATTENTION_FAST", "1"), "rows": {}}
    for shp in a.shapes.split(","):
        B, L = (int(v) for v in shp.split("x"))
        ids = torch.randint(1000, 30000, (B, L), device=dev)
        items = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
        for _ in range(2):
            model(passage=items)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.iters):
            model(passage=items)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
        out["rows"][shp] = {"ms": round(dt * 1e3, 2), "passages_per_s": round(B / dt, 1), "tokens_per_s": round(B * L / dt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
