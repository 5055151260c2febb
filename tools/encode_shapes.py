"""Encode throughput across batch shapes (queries L=32, passages L=128; bert-base, bf16) -- shows where
the launch-bound regime starts.  python tools/encode_shapes.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
from transformers import BertConfig, BertModel
from openmatch.modeling import DRModelForInference
torch.manual_seed(0)
lm = BertModel(BertConfig()).eval()
m = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="bfloat16")).to("cuda").eval()
for B, L in ((6980, 32), (1024, 32), (256, 32), (64, 32), (256, 128), (64, 128), (8, 128)):
    ids = torch.randint(1000, 30000, (B, L), device="cuda"); mask = torch.ones_like(ids)
    for _ in range(3): m(query={"input_ids": ids, "attention_mask": mask})
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): m(query={"input_ids": ids, "attention_mask": mask})
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    gf = 12 * (24 * L * 768 * 768 + 4 * L * L * 768) * B / 1e12
    print(f"B={B:5d} L={L:4d}: {dt*1e3:8.3f} ms  {B/dt:10.0f} seq/s  {gf/dt:7.1f} TFLOP/s")
