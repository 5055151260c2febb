"""Developer probe: one 1024-passage encode step as TWO half batches on two HIP streams, the persistent GEMM of each
capped to half the CUs (OM_OPT_GEMM_MAX_GRID), against the plain one-stream step.  What it asks: do two kernel chains
that drift out of phase (one in its epilogue / attention / normalisation while the other is in a K loop) use the chip
better than one chain whose 256 workgroups reach every store burst and every memory-bound kernel together?

    python tools/two_stream_probe.py [--precision f16|bf16] [--steps 10]
"""
import argparse
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024)
    a = ap.parse_args()
    from transformers import BertConfig, BertModel
    from openmatch_amd import native as N
    from openmatch_amd.encoder import hip_encode
    import bench

    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    lib = N.lib()
    torch.manual_seed(0)
    lm = BertModel(BertConfig()).eval().to(device)
    code = {"f16": N.OM_F16, "bf16": N.OM_BF16}[a.precision]
    L = 128
    batches = [bench.synth_ids(a.batch, L, device, i) for i in range(4)]
    batches = [{"input_ids": i, "attention_mask": m} for i, m in batches]

    orig_get = N.Workspace.get.__func__

    def get(cls, dev, nbytes, tag="default"):      # one workspace per stream
        return orig_get(cls, dev, nbytes, tag + ":" + str(torch.cuda.current_stream(dev).cuda_stream))
    N.Workspace.get = classmethod(get)

    def enc(items):
        return hip_encode(lm, items, "first", None, False, code, want_hidden=False)[1]

    def one_stream(i):
        return enc(batches[i % 4])

    streams = [torch.cuda.Stream(device) for _ in range(4)]

    def split(i, parts):
        b = batches[i % 4]
        n = a.batch // parts
        out = []
        cur = torch.cuda.current_stream(device)
        for p in range(parts):
            s = streams[p]
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                out.append(enc({k: v[p * n:(p + 1) * n] for k, v in b.items()}))
        for p in range(parts):
            cur.wait_stream(streams[p])
        return torch.cat(out)

    def timed(fn, label):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            fn(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print("%-44s %8.3f ms per 1024-passage step  %9.0f passages/s" % (label, dt * 1e3, a.batch / dt), flush=True)
        return dt

    ref = one_stream(0).clone()
    for rnd in range(2):
        lib.om_debug_option(N.OPT_GEMM_MAX_GRID, 0)
        timed(one_stream, "one stream, 256 workgroups")
        for parts, cap in ((2, 128), (2, 0), (2, 160), (4, 64), (2, 96)):
            lib.om_debug_option(N.OPT_GEMM_MAX_GRID, cap)
            got = split(0, parts)
            torch.cuda.synchronize()
            same = bool(torch.equal(got, ref))
            timed(lambda i: split(i, parts), "%d streams, GEMM grid cap %3d (identical=%s)" % (parts, cap, same))
        lib.om_debug_option(N.OPT_GEMM_MAX_GRID, 0)
        # (the staggered-start variant measured with this probe was removed from the kernels: +-1 %, profiles/r04_probe1_*)


if __name__ == "__main__":
    main()
