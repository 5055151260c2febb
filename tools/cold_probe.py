#!/usr/bin/env python
"""Cold-start probe of the encode leg: the bench's model and batches, N steps from a
process that has just started on an idle GPU, every step bracketed by HIP events (no host sync inside the
loop) -- prints the per-step series so that a clock ramp / allocator / first-touch effect shows as a shape.

  python tools/cold_probe.py [steps] [idle_seconds_before]
"""
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def sclk():
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))[:1]:
        try:
            out = [l.strip() for l in open(f).read().splitlines() if "*" in l]
        except Exception:
            pass
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    from types import SimpleNamespace as NS
    import bench
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    lm = BertModel(BertConfig()).eval()
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="bfloat16")).to(dev).eval()
    batches = [bench.synth_ids(1024, 128, dev, i) for i in range(4)]
    batches = [{"input_ids": i, "attention_mask": m} for i, m in batches]
    torch.cuda.synchronize()
    if idle > 0:
        time.sleep(idle)
    clk0 = sclk()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        model(passage=batches[i % 4])
        ev[i + 1].record()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    clk1 = sclk()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    print(json.dumps({"steps": steps, "idle_before_s": idle, "wall_ms_per_step": round(wall / steps * 1e3, 3),
                      "first5": [round(x, 2) for x in ms[:5]], "series_ms": [round(x, 2) for x in ms],
                      "host_issue_ms": [round(h * 1e3, 1) for h in host[:10]],
                      "min": round(min(ms), 3), "median": round(sorted(ms)[len(ms) // 2], 3), "sclk_before": clk0, "sclk_after": clk1}))


if __name__ == "__main__":
    main()
