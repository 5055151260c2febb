"""One RCCL communicator per process behind the C ABI (include/openmatch_hip.h: om_comm_*, om_allgather_rows,
om_allreduce_grads, om_exchange_topk), bootstrapped from the torch.distributed rendezvous the launcher set up:
rank 0 draws the unique id, `broadcast_object_list` hands it to every rank.  The model / trainer / retriever use
torch.distributed's own collectives by default (backend "nccl" IS RCCL on ROCm, same wires); setting
OPENMATCH_AMD_COMM=native routes the three data-path collectives through this object instead -- the boundary a
non-Python host would bind."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import native as N

_comm = None


class RcclComm:
    def __init__(self, world: int, rank: int, device, unique_id: bytes):
        self.world, self.rank, self.device = world, rank, torch.device(device)
        self._h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, 128)
        with torch.cuda.device(self.device):
            N.check(N.lib().om_comm_init(buf, world, rank, C.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        N.check(N.lib().om_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, device):
        world, rank = dist.get_world_size(), dist.get_rank()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(world, rank, device, box[0])

    def count(self) -> int:
        """Ranks of the communicator as RCCL reports them (ncclCommCount)."""
        n = C.c_int(0)
        N.check(N.lib().om_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def close(self):
        if self._h:
            N.check(N.lib().om_comm_destroy(self._h))
            self._h = C.c_void_p()

    def allgather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """[n, ...] on every rank -> [world * n, ...], rank-major (DRModel.dist_gather_tensor's layout)."""
        t = t.contiguous()
        N.require_device(t)
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        rows = t.shape[0]
        row_bytes = t[0].numel() * t.element_size() if rows else t.element_size()
        with torch.cuda.device(t.device):
            N.check(N.lib().om_allgather_rows(self._h, N.ptr(t), N.ptr(out), rows, row_bytes, N.stream_ptr(t.device)))
        return out

    def allreduce_grads_(self, flat: torch.Tensor, average: bool = True) -> torch.Tensor:
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        N.require_device(flat)
        with torch.cuda.device(flat.device):
            N.check(N.lib().om_allreduce_grads(self._h, N.ptr(flat), flat.numel(), int(average), N.stream_ptr(flat.device)))
        return flat

    def exchange_topk(self, D: torch.Tensor, I: torch.Tensor):
        """D, I [world * blk, k] (this shard's candidates for all queries) -> [world, blk, k] candidates of every shard
        for this rank's query block."""
        D, I = D.to(torch.float32).contiguous(), I.to(torch.int64).contiguous()
        N.require_device(D, I)
        blk, k = D.shape[0] // self.world, D.shape[1]
        rD, rI = torch.empty_like(D), torch.empty_like(I)
        with torch.cuda.device(D.device):
            N.check(N.lib().om_exchange_topk(self._h, self.world, N.ptr(D), N.ptr(I), blk, k, N.ptr(rD), N.ptr(rI),
                                             N.stream_ptr(D.device)))
        return rD.view(self.world, blk, k), rI.view(self.world, blk, k)


def native_comm(device=None):
    """The process-wide communicator when OPENMATCH_AMD_COMM=native and torch.distributed is initialised, else None."""
    global _comm
    if os.environ.get("OPENMATCH_AMD_COMM", "") != "native" or not dist.is_available() or not dist.is_initialized():
        return None
    if _comm is None:
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        _comm = RcclComm.from_torch_distributed(dev)
    return _comm
