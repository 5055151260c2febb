#!/usr/bin/env python
"""The automatic tile choice against pinned generations (2: 256 x 128, 6: 256 x 256 persistent where a variant exists) on the WIDE
contractions of a training step (N = 2304, 3072) and the f32-output ones, at 9 216 and 5 120 rows, both 16-bit formats.
python tools/gemm_variant_probe2.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmatch_amd import native as N


def main():
    dev = torch.device("cuda:0")
    lib = N.lib()
    out = {}
    for dt, code in ((torch.float16, N.OM_F16), (torch.bfloat16, N.OM_BF16)):
        for M in (5120, 9216):
            for (Nn, K, odt, act, res) in ((2304, 768, dt, 0, False), (3072, 768, dt, N.ACT_MUL_RESID, True), (768, 3072, torch.float32, 0, True),
                                           (768, 768, torch.float32, 0, True)):
                A = torch.randn(M, K).to(dev, dt)
                W = (torch.randn(Nn, K) * 0.05).to(dev, dt)
                C = torch.empty(M, Nn, device=dev, dtype=odt)
                R = torch.randn(M, Nn).to(dev, odt)
                bias = torch.randn(Nn).to(dev)
                ocode = N.OM_F32 if odt == torch.float32 else code
                row = {}
                for name, var in (("auto", 0), ("gen2", 2), ("gen6", 6)):
                    N.check(lib.om_debug_option(12, var))
                    def go():
                        N.check(lib.om_gemm_nt(code, N.ptr(A), K, N.ptr(W), K, ocode, N.ptr(C), Nn, M, Nn, K, N.ptr(bias),
                                               N.ptr(R) if res else None, Nn, act, N.stream_ptr(dev)))
                    try:
                        for _ in range(5):
                            go()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        torch.cuda.synchronize(); e0.record()
                        for _ in range(50):
                            go()
                        e1.record(); torch.cuda.synchronize()
                        row[name] = round(e0.elapsed_time(e1) * 1e3 / 50, 1)
                    except N.NativeError as e:
                        row[name] = "n/a"
                N.check(lib.om_debug_option(12, 0))
                out[f"{'f16' if dt == torch.float16 else 'bf16'}_M{M}_N{Nn}_K{K}_{'f32out' if odt == torch.float32 else '16out'}{'_mulresid' if act else ''}"] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
