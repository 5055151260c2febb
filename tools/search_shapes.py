#!/usr/bin/env python
"""Exact top-k search latency / throughput across query-batch sizes (MFMA-bound for large
batches, HBM-bound -- one pass over the 13.6 GB f16 index -- for small ones).
  python tools/search_shapes.py [--rows 8841823] [--topk 1000]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8841823)
    ap.add_argument("--topk", type=int, default=1000)
    ap.add_argument("--queries", type=int, nargs="*", default=[1, 8, 64, 256, 1024, 6980])
    a = ap.parse_args()
    from openmatch_amd.index import FlatIPIndex
    dev = torch.device("cuda:0")
    index = FlatIPIndex(768, device=dev, precision="f16_rescore")
    g = torch.Generator(device=dev).manual_seed(77)
    shared = torch.randn(1, 768, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    index._reserve(a.rows)
    for s in range(0, a.rows, 1 << 20):
        n = min(1 << 20, a.rows - s)
        index.add(torch.randn(n, 768, device=dev, generator=g) * 0.05 + shared * 0.05)
    for nq in a.queries:
        q = torch.randn(nq, 768, device=dev, generator=g) * 0.05 + shared * 0.05
        index.search_device(q, a.topk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            index.search_device(q, a.topk)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"queries": nq, "topk": a.topk, "ms": round(dt * 1e3, 2), "queries_per_s": round(nq / dt, 1),
                          "index_stream_TBps_if_read_once": round(a.rows * 768 * 2 / dt / 1e12, 2),
                          "tflops": round(2.0 * a.rows * 768 * nq / dt / 1e12, 1), "info": index.last_search_info}))


if __name__ == "__main__":
    main()
