#!/usr/bin/env python
"""GTR-base-shaped encode throughput (BASELINE config 4: T5 encoder 12 x 768, relu FFN 3072, mean pooling,
768 -> 768 head, normalise; 128 tokens).   python tools/gtr_bench.py [--batch 1024] [--gated] [--dtype bfloat16|float16]"""
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
    ap.add_argument("--dtype", default="bfloat16")
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
                                model_args=NS(encoder_only=True, dtype=a.dtype)).to(dev).eval()
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
    N.check(N.lib().om_debug_option(0, 1))
    if not a.gated:      # ragged batches (lengths ~ U{16..128}, as bench.py's) padded vs packed rows (om_encoder_forward_packed)
        import bench
        from openmatch_amd import encoder as E
        rid, rmask = bench.synth_ids(a.batch, 128, torch.device(dev), 7)
        ragged = {"input_ids": rid, "attention_mask": rmask}
        rows = E.packed_rows_bound(rmask)
        code = E.compute_dtype_code(model.model_args)
        run = lambda pr: E.hip_encode(model.lm_p, ragged, "mean", model.head_p, True, code, want_hidden=False, packed_rows=pr)[1]
        same = bool(torch.equal(run(rows), run(None)))
        for name, pr in (("ragged_padded", None), ("ragged_packed", rows)):
            for _ in range(2):
                run(pr)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.steps):
                run(pr)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
            out[name] = {"passages_per_s": round(a.batch / dt, 1), "ms_per_step": round(dt * 1e3, 2)}
        out["ragged_packed"].update(rows=rows, padded_rows=a.batch * 128, identical_to_padded=same)
    print(json.dumps({"metric": "GTR-base-shaped encode passages/s (T5 encoder, 128 tokens)", "dtype": a.dtype, "batch": a.batch,
                      "gated": a.gated, **out}))


if __name__ == "__main__":
    main()
