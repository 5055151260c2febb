"""Gradient all-reduce overlapped with the backward pass (data-parallel training over RCCL / xGMI).

What DistributedDataParallel does for the reference (HF Trainer wraps the model in DDP; trainer/dense_trainer.py:102-108):
gradients are averaged across ranks in buckets WHILE the backward still runs.  Here the HIP backward writes every
gradient of an encoder into one f32 arena, layer by layer from the top (openmatch_amd/train.py), and records one event per
layer on its stream (om_encoder_train_set_layer_events).  `GradSync.reduce_arena` hands each bucket -- a contiguous arena
slice of `bucket_layers` layers (default 4: the layer group of the backward's deferred weight-gradient launches,
OM_OPT_TRAIN_WGRAD_BATCH -- a group's progress events are recorded together, behind that launch); xGMI rings are per-link
bound, so few large collectives -- to the collective as soon as
its lowest layer's event has fired, on a side stream; `finish()` makes the caller's stream wait for all of them.  A flat
copy of nothing: the bucket IS the slice.

The arithmetic equals one all-reduce over the whole arena (every element is reduced exactly once, with the same
operation): tests/test_distributed_cpu.py checks bucketed == single-shot over gloo."""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_active = None          # the GradSync that the next _EncoderTrain.backward reports its arena to (set by the trainer)


def active():
    return _active


class _SideStreamWork:
    """What dist.all_reduce(async_op=True) returns, for a stand-in collective: wait() orders the current stream behind it."""

    def __init__(self, stream):
        self.event = torch.cuda.Event()
        self.event.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class GradSync:
    def __init__(self, world_size: int, bucket_layers: int = 4, collective=None):
        """collective (tests): a callable run on the side stream IN PLACE of the all-reduce, on each bucket -- a visible
        operation (e.g. `flat.mul_(2)`) shows on ONE device whether a bucket was handed over before its last gradient
        had been written, which an identity all-reduce on a one-rank group cannot."""
        self.world = int(world_size)
        self._collective = collective
        self.bucket_layers = max(1, int(bucket_layers))
        self.works = []
        self.reduced = set()          # storage pointers whose gradients are already averaged
        self._side = None
        self._keep = []

    # ---- trainer side ---------------------------------------------------------------------------------
    def begin(self):
        global _active
        self.works, self._keep = [], []
        self.reduced = set()
        _active = self

    def _wait_works(self):
        for w, flat in self.works:
            w.wait()                             # the CURRENT stream (not the host) waits for the collective
            if flat is not None:                 # gloo: sum -> mean
                flat /= self.world
        self.works = []

    def finish(self):
        """Block the current stream until every bucket's collective has completed."""
        global _active
        _active = None
        self._wait_works()
        self._keep = []

    # ---- backward side --------------------------------------------------------------------------------
    def buckets(self, layer_bounds: Sequence[Tuple[int, int]], arena_len: int) -> List[Tuple[int, int, int]]:
        """(lo, hi, event index) in COMPLETION order.  layer_bounds[l] = arena span of layer l; what precedes layer 0
        (embeddings) is the last bucket, what follows the top layer (the head) rides with the first."""
        nl = len(layer_bounds)
        out = []
        top = nl
        while top > 0:
            bot = max(0, top - self.bucket_layers)
            lo = layer_bounds[bot][0]
            hi = arena_len if top == nl else layer_bounds[top - 1][1]
            out.append((lo, hi, bot))            # ready once layer `bot` has been differentiated
            top = bot
        first = layer_bounds[0][0] if nl else arena_len
        if first > 0:
            out.append((0, first, nl))           # embeddings (+ T5 tables): event n_layers
        return out

    def reduce_arena(self, arena: torch.Tensor, layer_bounds, events: Optional[list] = None):
        """Launch the bucketed all-reduce(mean) of `arena` (1-D f32).  events[i]: torch.cuda.Event recorded by the backward
        (None on CPU: the buckets are reduced right away)."""
        # A step with several backward passes over shared weights (an untied or two-call bi-encoder: queries, then
        # passages) has autograd ADD the second arena's views into the first's: from the second arena on, everything
        # launched so far -- and this arena's own collectives, below -- must be complete on the caller's stream before the
        # backward returns to autograd.  (The one-arena step keeps its collectives in flight until finish().)
        later = bool(self.reduced)
        if later:
            self._wait_works()
        nccl = self._collective is None and dist.get_backend() == "nccl"
        if arena.is_cuda and self._side is None:
            self._side = torch.cuda.Stream(device=arena.device)
        for lo, hi, ev in self.buckets(layer_bounds, arena.numel()):
            if hi <= lo:
                continue
            flat = arena[lo:hi]
            if arena.is_cuda:
                if events is not None and events[ev] is not None:
                    self._side.wait_event(events[ev])
                else:
                    self._side.wait_stream(torch.cuda.current_stream(arena.device))
                with torch.cuda.stream(self._side):
                    if self._collective is not None:
                        self._collective(flat)
                        w = _SideStreamWork(self._side)
                    else:
                        w = dist.all_reduce(flat, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, async_op=True)
                flat.record_stream(self._side)
            else:
                w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            self.works.append((w, None if ((arena.is_cuda and nccl) or self._collective is not None) else flat))
        self._keep.append((arena, events))
        self.reduced.add(arena.untyped_storage().data_ptr())
        if later:
            self._wait_works()
