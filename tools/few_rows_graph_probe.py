#!/usr/bin/env python
"""Where a one-query forward's time goes (round 6): host time to ISSUE the launch chain against the device's time to run it, and the
same chain replayed from a captured HIP graph (torch.cuda.CUDAGraph around the library call: the library launches on torch's current
stream and allocates nothing after the first call of a shape).    python tools/few_rows_graph_probe.py [--dtype float16]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--shapes", default="1x32,4x32,8x32,1x128")
    a = ap.parse_args()
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    dev = "cuda:0"
    torch.manual_seed(0)
    lm = BertModel(BertConfig()).eval()
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype=a.dtype)).to(dev).eval()
    out = {"dtype": a.dtype, "rows": {}}
    for shp in a.shapes.split(","):
        B, L = (int(v) for v in shp.split("x"))
        ids = torch.randint(1000, 30000, (B, L), device=dev)
        items = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
        for _ in range(5):
            ref = model(query=items).q_reps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            model(query=items)
        t_issue = (time.perf_counter() - t0) / a.iters
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / a.iters
        row = {"host_issue_ms": round(t_issue * 1e3, 3), "wall_ms": round(t_all * 1e3, 3)}
        try:
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                for _ in range(3):
                    model(query=items)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                reps = model(query=items).q_reps
            g.replay(); torch.cuda.synchronize()
            row["graph_equal"] = bool(torch.equal(reps, ref))
            t0 = time.perf_counter()
            for _ in range(a.iters):
                g.replay()
            torch.cuda.synchronize()
            row["graph_ms"] = round((time.perf_counter() - t0) / a.iters * 1e3, 3)
        except Exception as e:  # noqa: BLE001
            row["graph_error"] = str(e)[:300]
        out["rows"][shp] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
