#!/usr/bin/env python
"""Launch-to-launch time of om_gemm_nt on few-row shapes for every (MT, NT, NW) of gemm_skinny.hip and for the tile kernels
(OM_OPT_GEMM_SKINNY_M = 0).   python tools/skinny_sweep.py [--dtype float16]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmatch_amd import native as N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--rows", default="1,32,64,128,256,512,1024")
    ap.add_argument("--shapes", default="2304x768,768x768,3072x768,768x3072", help="N x K list")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    td = getattr(torch, a.dtype)
    code = N.OM_F16 if a.dtype == "float16" else N.OM_BF16
    lib = N.lib()
    out = {"metric": "us per om_gemm_nt launch (dependent chain on one stream)", "dtype": a.dtype, "shapes": {}}
    N.check(lib.om_debug_option(19, 1 << 20))
    for (Nn, K) in [tuple(int(v) for v in x.split('x')) for x in a.shapes.split(',')]:
        W = (torch.randn(Nn, K) * 0.05).to(dev, td)
        bias = torch.randn(Nn).to(dev)
        for M in [int(x) for x in a.rows.split(",")]:
            A = torch.randn(M, K).to(dev, td)
            C = torch.empty(M, Nn, device=dev, dtype=td)
            mts = [1] if M <= 16 else ([1, 2] if M <= 32 else [1, 2, 4])
            cfgs = [("tiles", None)] + [(f"{mt}/{nt}/{nw}", mt * 10000 + nt * 100 + nw) for mt in mts for nt in (1, 2, 4) for nw in (4, 8, 16)
                                        if mt * nt * nw <= 64 and not (nt == 4 and nw == 16) and K % (32 * nw) == 0]
            row = {}
            for name, cfg in cfgs:
                N.check(lib.om_debug_option(19, 0 if cfg is None else 1 << 20))
                N.check(lib.om_debug_option(20, cfg or 0))
                def go():
                    N.check(lib.om_gemm_nt(code, N.ptr(A), K, N.ptr(W), K, code, N.ptr(C), Nn, M, Nn, K, N.ptr(bias), None, 0, 1, N.stream_ptr(dev)))
                for _ in range(10):
                    go()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(a.iters):
                    go()
                e1.record(); torch.cuda.synchronize()
                row[name] = round(e0.elapsed_time(e1) * 1e3 / a.iters, 2)
            out["shapes"].setdefault(f"N{Nn}_K{K}", {})[str(M)] = row
    N.check(lib.om_debug_option(19, 1024)); N.check(lib.om_debug_option(20, 0))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
