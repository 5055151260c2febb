#!/usr/bin/env python
"""Tile generation 1 (128 x 128) vs 2 (256 x 128) vs the automatic choice on the N = 768 contractions of a training step, at the row
counts of the padded (9 216) and a packed (5 120) batch: kernel durations through rocprofv3, or launch-to-launch time without it.
python tools/gemm_variant_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmatch_amd import native as N


def main():
    dev = torch.device("cuda:0")
    lib = N.lib()
    out = {}
    for M in (5120, 9216):
        for (Nn, K, odt) in ((768, 768, torch.float32), (768, 3072, torch.float32), (768, 3072, torch.float16), (768, 2304, torch.float16)):
            A = torch.randn(M, K).to(dev, torch.float16)
            W = (torch.randn(Nn, K) * 0.05).to(dev, torch.float16)
            C = torch.empty(M, Nn, device=dev, dtype=odt)
            R = torch.randn(M, Nn).to(dev, odt)
            bias = torch.randn(Nn).to(dev)
            ocode = N.OM_F32 if odt == torch.float32 else N.OM_F16
            row = {}
            for name, var in (("auto", 0), ("gen1_128x128", 1), ("gen2_256x128", 2)):
                N.check(lib.om_debug_option(12, var))           # OM_OPT_GEMM_VARIANT
                def go():
                    N.check(lib.om_gemm_nt(N.OM_F16, N.ptr(A), K, N.ptr(W), K, ocode, N.ptr(C), Nn, M, Nn, K, N.ptr(bias), N.ptr(R), Nn, 0, N.stream_ptr(dev)))
                for _ in range(5):
                    go()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(50):
                    go()
                e1.record(); torch.cuda.synchronize()
                row[name] = round(e0.elapsed_time(e1) * 1e3 / 50, 1)
            N.check(lib.om_debug_option(12, 0))
            out[f"M{M}_N{Nn}_K{K}_{'f32' if odt == torch.float32 else 'f16'}out"] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
