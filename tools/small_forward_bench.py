#!/usr/bin/env python
"""Latency of SMALL forwards (a served query, a handful of sequences): bert-base, float16 / bfloat16, B x L token blocks, for several
values of OM_OPT_GEMM_SKINNY_M (0 = the tile kernels only).   python tools/small_forward_bench.py [--dtype float16] [--iters 50]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from openmatch_amd import native as N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--limits", default="0,64,256,1024,4096")
    ap.add_argument("--fused", type=int, default=1, help="OM_OPT_ENCODER_FUSED_LN")
    ap.add_argument("--shapes", default="1x32,4x32,16x32,64x32,1x128,8x128,16x128,32x128")
    a = ap.parse_args()
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    dev = "cuda:0"
    torch.manual_seed(0)
    lm = BertModel(BertConfig()).eval()
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype=a.dtype)).to(dev).eval()
    shapes = [tuple(int(v) for v in x.split("x")) for x in a.shapes.split(",")]
    N.check(N.lib().om_debug_option(0, a.fused))
    out = {"metric": "ms per forward (bert-base, ids resident in HBM, representations out)", "dtype": a.dtype, "fused_ln": a.fused, "rows": {}}
    ref = {}
    for lim in [int(x) for x in a.limits.split(",")]:
        N.check(N.lib().om_debug_option(19, lim))
        for (B, L) in shapes:
            ids = torch.randint(1000, 30000, (B, L), device=dev)
            items = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
            for _ in range(5):
                reps = model(query=items).q_reps
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.iters):
                reps = model(query=items).q_reps
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
            out["rows"].setdefault(f"{B}x{L}", {})[str(lim)] = round(dt * 1e3, 3)
    N.check(N.lib().om_debug_option(19, 1024)); N.check(N.lib().om_debug_option(0, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
