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
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16"], help="f16 = the reference's --fp16 (torch.cuda.amp float16)")
    a = ap.parse_args()
    from transformers import BertConfig, BertModel
    from openmatch.modeling import RRModel, LinearHead
    dev = "cuda:0"
    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    lm = BertModel(cfg).eval()
    model = RRModel(lm=lm, head=LinearHead(1024, 1), pooling="first",
                    model_args=NS(encoder_only=False, dtype={"bf16": "bfloat16", "f16": "float16"}[a.precision])).to(dev).eval()
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
    # ragged pairs (query + passage of ~U{40..162} tokens, padded to 162 by the collator): the padded entry vs the packed-rows
    # entry the model takes by itself when the batch arrives in the collator's compact form (feed.py: host-side lengths)
    from openmatch_amd import encoder as E
    from openmatch_amd.feed import pack_token_batch, token_rows_bound
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(40, L + 1, (a.pairs,), generator=g)
    hmask = (torch.arange(L)[None, :] < lens[:, None]).long()
    hids = torch.randint(1000, 30000, (a.pairs, L), generator=g) * hmask
    htt = ((torch.arange(L)[None, :] >= 34) & (hmask > 0)).long()
    compact = pack_token_batch({"input_ids": hids, "attention_mask": hmask, "token_type_ids": htt})
    compact = {k: (v.to(dev) if k != "lengths" and k != "_packed_tokens" else v) for k, v in compact.items()}
    padded_items = {"input_ids": hids.to(dev), "attention_mask": hmask.to(dev), "token_type_ids": htt.to(dev)}
    ragged = {}
    with torch.no_grad():
        same = bool(torch.equal(model.encode(compact), model.encode(padded_items)))
        took = dict(E.LAST_CALL)
        for name, items_r in (("ragged_padded", padded_items), ("ragged_packed", compact)):
            for _ in range(2):
                model.encode(items_r)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.steps):
                model.encode(items_r)
            torch.cuda.synchronize(); dtr = (time.perf_counter() - t0) / a.steps
            ragged[name] = {"pairs_per_s": round(a.pairs / dtr, 1), "ms_per_batch": round(dtr * 1e3, 2)}
    ragged["ragged_packed"].update(rows=token_rows_bound(compact), padded_rows=a.pairs * L, identical_to_padded=same, padded_call=took)
    print(json.dumps({"metric": "cross-encoder pairs/s (bert-large, 162 tokens, %s)" % a.precision, "pairs_per_s": round(a.pairs / dt, 1),
                      "ms_per_batch": round(dt * 1e3, 2), "pairs": a.pairs, "algorithmic_tflops": round(gflop * a.pairs / dt / 1e3, 1),
                      **ragged}))


if __name__ == "__main__":
    main()
