#!/usr/bin/env python
"""Per-tile phase stamps of the encoder's LAST residual GEMM (FFN2 of the last layer) on the benchmark batch, through the
library's own trace facility (om_debug_gemm_trace: 32 shader-clock stamps per tile; the last launch that covers a tile id wins --
after one forward the ids below its tile count belong to that FFN2).  Kernel 7r16 stamps: [0] K loop start, [15] K loop end,
[16] first epilogue wait done, [17 + p] start of patch iteration p, [28] last store issued.

    python tools/epilogue_trace.py            (OM_ENCODER_TWO_PLANE=3 | 1 selects the two- / one-plane float16 stream)
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS  # noqa: E402

from openmatch_amd import native as N  # noqa: E402


def main():
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lm = BertModel(BertConfig()).eval()
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="float16")).to(dev).eval()
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(1000, 30522, (1024, 128), generator=g)
    mask = torch.ones_like(ids)
    batch = {"input_ids": ids.to(dev), "attention_mask": mask.to(dev)}
    for _ in range(3):
        model(passage=batch)
    torch.cuda.synchronize()
    ntile = 8192
    buf = torch.zeros(ntile * 32, dtype=torch.int64, device=dev)
    N.lib().om_debug_gemm_trace(C.c_void_p(buf.data_ptr()))
    model(passage=batch)
    torch.cuda.synchronize()
    N.lib().om_debug_gemm_trace(None)
    t = buf.cpu().numpy().reshape(ntile, 32).astype(np.int64)
    nt = (1024 * 128 // 256) * 3                       # FFN2 / out-proj: 512 x 3 tiles
    t = t[:nt]
    ok = (t[:, 28] > t[:, 0]) & (t[:, 0] > 0)
    t = t[ok]
    out = {"tiles": int(ok.sum()), "two_plane": os.environ.get("OM_ENCODER_TWO_PLANE", "3")}
    med = lambda x: float(np.median(x))
    out["k_loop"] = med(t[:, 15] - t[:, 0])
    out["k_end_to_first_wait"] = med(t[:, 16] - t[:, 15])
    out["patch_iter"] = [med(t[:, 18 + p] - t[:, 17 + p]) for p in range(7)]
    out["last_iter_to_end"] = med(t[:, 28] - t[:, 24])
    out["epilogue_total"] = med(t[:, 28] - t[:, 15])
    # tile to tile on one CU: stamp [29] holds the workgroup id
    by_wg = {}
    for row in t:
        by_wg.setdefault(int(row[29]), []).append(row)
    gaps = []
    for rows in by_wg.values():
        rows.sort(key=lambda r: r[0])
        gaps += [int(b[0] - a[0]) for a, b in zip(rows, rows[1:])]
    out["tile_period"] = med(np.array(gaps)) if gaps else None
    # FFN1 + GELU of the last layer (kernel 7c16): tile ids [nt, 4 nt) survive the later FFN2 launch; stamps [16] epilogue start,
    # [17 + p] start of patch iteration p (p < 8), [28] last store issued
    t1 = buf.cpu().numpy().reshape(ntile, 32).astype(np.int64)[nt:4 * nt]
    ok1 = (t1[:, 28] > t1[:, 15]) & (t1[:, 15] > t1[:, 0]) & (t1[:, 0] > 0)
    t1 = t1[ok1]
    if len(t1):
        f1 = {"tiles": int(ok1.sum()), "k_loop": med(t1[:, 15] - t1[:, 0]), "epilogue_total": med(t1[:, 28] - t1[:, 15]),
              "k_end_to_epilogue_start": med(t1[:, 16] - t1[:, 15]),
              "patch_iter": [med(t1[:, 18 + p] - t1[:, 17 + p]) for p in range(7)], "last_iter_to_end": med(t1[:, 28] - t1[:, 24])}
        by = {}
        for row in t1:
            by.setdefault(int(row[29]), []).append(row)
        g1 = []
        for rows in by.values():
            rows.sort(key=lambda r: r[0])
            g1 += [int(b[0] - a[0]) for a, b in zip(rows, rows[1:])]
        f1["tile_period"] = med(np.array(g1)) if g1 else None
        out["ffn1_gelu"] = f1
    out["epi_skew"] = os.environ.get("OM_GEMM_EPI_SKEW", "0")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
