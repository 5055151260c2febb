#!/usr/bin/env python
"""Known-good reference on the same box: the vendor GEMM (hipBLASLt through torch.matmul) on the
encoder's four contraction shapes, random bf16 operands, no epilogue.  A measuring stick for
DESIGN.md only -- nothing in the product calls it."""
import json
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = "cuda:0"
out = []
for (n, k) in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (8192, 8192)]:
    m = M if n != 8192 else 8192
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.05
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    for name, fn in (("matmul", lambda: a @ w.t()), ("linear+bias", lambda: torch.nn.functional.linear(a, w, bias))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 20
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        r = {"op": name, "M": m, "N": n, "K": k, "ms": round(ms, 4), "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}
        print(json.dumps(r), flush=True)
        out.append(r)
