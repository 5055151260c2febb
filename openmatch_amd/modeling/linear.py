"""Projection head of the bi-encoder — same surface and files as the reference's
`openmatch.modeling.linear.LinearHead` (src/openmatch/modeling/linear.py:12-38): a bias-free
`nn.Linear`, saved as `linear.pt` + `head_config.json`.  On the hot path the matmul runs
inside `om_encoder_forward` (exact-f32 MFMA); `forward` here is the stand-alone entry."""
import json
import logging
import os

import torch
from torch import Tensor, nn

from .. import native as N

logger = logging.getLogger(__name__)


class LinearHead(nn.Module):
    def __init__(self, input_dim: int = 768, output_dim: int = 768):
        super().__init__()
        self.linear = nn.Linear(input_dim, output_dim, bias=False)
        self.config = {"input_dim": input_dim, "output_dim": output_dim}

    def forward(self, rep: Tensor = None):
        x = rep.to(torch.float32).contiguous()
        w = self.linear.weight.detach().to(torch.float32).contiguous()
        N.require_device(x, w)
        out = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            N.check(N.lib().om_gemm_nt(N.OM_F32, N.ptr(x), x.shape[1], N.ptr(w), w.shape[1], N.OM_F32,
                                       N.ptr(out), out.shape[1], x.shape[0], w.shape[0], x.shape[1],
                                       None, None, 0, N.ACT_NONE, N.stream_ptr(x.device)))
        return out

    @classmethod
    def load(cls, ckpt_dir: str):
        logger.info("Loading linear head from %s", ckpt_dir)
        with open(os.path.join(ckpt_dir, "head_config.json")) as f:
            cfg = json.load(f)
        head = cls(**cfg)
        head.load_state_dict(torch.load(os.path.join(ckpt_dir, "linear.pt")))
        return head

    def save(self, save_path):
        torch.save(self.state_dict(), os.path.join(save_path, "linear.pt"))
        with open(os.path.join(save_path, "head_config.json"), "w") as f:
            json.dump(self.config, f, indent=4)
