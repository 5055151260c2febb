"""`DRTrainer` with the reference's constructor and public methods
(src/openmatch/trainer/dense_trainer.py:27-108) but its OWN training loop: the reference
subclasses HF `Trainer`, whose 4.10-era hooks (`tokenizer=` kwarg, `compute_loss(model, inputs)`,
`self.scaler`) no longer exist in the installed transformers, and whose DDP wrapper would put an
NCCL bucket hook on every HIP backward.  Here one process per MI355X runs

    forward (HIP encoder, tape) -> all-gather of embeddings as cross-device negatives (RCCL) ->
    fused contrastive loss fwd+bwd -> HIP encoder backward -> ONE flat-bucket gradient all-reduce
    (mean, over xGMI) -> clip -> AdamW -> linear warm-up/decay schedule

with the reference's loss conventions: the model multiplies the loss by world_size when
negatives are shared (modeling :124-125), the mean all-reduce divides gradients back, and the
logged loss is divided by `_dist_loss_scale_factor` (:30,107-108).
"""
import logging
import math
import os
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, IterableDataset
from transformers.trainer import TRAINING_ARGS_NAME
from transformers.trainer_pt_utils import IterableDatasetShard

logger = logging.getLogger(__name__)


def linear_schedule_factor(step: int, warmup_steps: int, total_steps: int) -> float:
    """HF `get_linear_schedule_with_warmup`: 0 -> 1 over the warm-up, then 1 -> 0 at total_steps."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))


def parameter_groups(model, weight_decay: float):
    """HF Trainer's default: no weight decay on biases and LayerNorm weights."""
    decay, no_decay = [], []
    seen = set()
    for name, p in model.named_parameters():
        if not p.requires_grad or id(p) in seen:
            continue
        seen.add(id(p))
        (no_decay if (name.endswith("bias") or "LayerNorm" in name or "layer_norm" in name) else decay).append(p)
    return [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]


_ctl = {"group": None, "made_for": None}


def _control_group():
    """A host-side (gloo) process group for tiny control collectives that must not touch the GPU stream; the default group itself
    when that is already gloo.  Cached per DEFAULT process group: after destroy_process_group() + a new init (tests, several trainers
    in one process) a fresh one is made.  `DRTrainer.train` calls this once up front, where every rank is at the same point
    (new_group is itself a collective); a lazy first call from allreduce_mean_ is safe only because every rank makes it in step."""
    from torch.distributed import distributed_c10d as c10d
    default = c10d._get_default_group()
    if _ctl["made_for"] is not default:
        grp = None
        if dist.get_backend() != "gloo":
            try:
                grp = dist.new_group(backend="gloo")
            except Exception as e:            # noqa: BLE001  (no gloo in this build: the check runs over the default group instead)
                logger.warning("no gloo control group (%s): gradient-pattern check runs over the default group", e)
        _ctl["group"], _ctl["made_for"] = grp, default
    return _ctl["group"]


def allreduce_mean_(params: List[torch.nn.Parameter], world_size: int, bucket_bytes: int = 256 << 20, skip_storages=()):
    """Gradient averaging across ranks = what DistributedDataParallel does for the reference.

    The HIP backward writes every parameter gradient of an encoder into ONE zero-initialised f32 arena and hands
    autograd views of it (openmatch_amd/train.py), so `p.grad` of a whole model normally share one storage: that
    storage span is all-reduced IN PLACE with a single collective -- no flatten copy, no copy back.  (xGMI is
    point-to-point: a ring all-reduce is per-link bound, so one large call beats many small ones.)  Gradients that
    live elsewhere (a head trained by plain autograd, CPU tests) go through flat copy buckets of `bucket_bytes`."""
    # (skip_storages: arenas that openmatch_amd/grad_sync.py already averaged while the backward ran)
    grads = [p.grad for p in params if p.grad is not None and p.grad.untyped_storage().data_ptr() not in skip_storages]

    def reduce_(flat):
        from ..comm import native_comm
        comm = native_comm(flat.device) if flat.is_cuda and flat.dtype == torch.float32 else None
        if comm is not None:                     # OPENMATCH_AMD_COMM=native: om_allreduce_grads behind the C ABI
            comm.allreduce_grads_(flat, average=True)
        elif dist.get_backend() == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(flat)
            flat /= world_size

    by_store = {}
    for g in grads:
        by_store.setdefault((g.untyped_storage().data_ptr(), g.dtype, g.device), []).append(g)
    loose, spans = [], []
    for (_, dtype, _dev), gs in by_store.items():
        lo = min(g.storage_offset() for g in gs)
        hi = max(g.storage_offset() + g.numel() for g in gs)
        dense = all(g.is_contiguous() for g in gs) and sum(g.numel() for g in gs) * 2 >= (hi - lo)
        if len(gs) > 1 and dense:
            spans.append((dtype, gs[0], lo, hi, len(gs)))
        else:
            loose.extend(gs)
    # Every rank must issue the same collectives: the span sizes depend on which parameters received a gradient and on
    # the arena layout.  EVERY step a 56-bit hash of this rank's pattern goes through one 16-byte host-side all-reduce
    # (max of h and of -h: all ranks see the same two numbers, so they all raise or none does); a mismatch (a parameter
    # unused on one rank only) raises everywhere instead of hanging in collectives of unequal sizes.  (Round 3 compared
    # a pattern only the first time a rank saw it: a rank with a NEW pattern then gathered while the others all-reduced.)
    sig = (tuple((str(dt), hi - lo, n) for dt, _g, lo, hi, n in spans), tuple(g.numel() for g in loose))
    if world_size > 1 and dist.is_initialized():
        import hashlib
        h = int.from_bytes(hashlib.blake2b(repr(sig).encode(), digest_size=7).digest(), "little")
        grp = _control_group()
        t = torch.tensor([h, -h], dtype=torch.int64, device=grads[0].device if (grp is None and dist.get_backend() == "nccl" and grads) else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
        if int(t[0]) != -int(t[1]):
            raise RuntimeError("allreduce_mean_: the ranks hold different gradient patterns (a parameter that received "
                               f"no gradient on some ranks only?); this rank: {sig}")
    for dtype, g0, lo, hi, _n in spans:
        span = torch.empty(0, dtype=dtype, device=g0.device).set_(g0.untyped_storage(), lo, (hi - lo,))
        reduce_(span)                                       # the views of the arena see the averaged values
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        reduce_(flat)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    for g in loose:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
            bucket, size = [], 0
    flush()


class DRTrainer:
    def __init__(self, model=None, args=None, data_collator=None, train_dataset=None, eval_dataset=None,
                 tokenizer=None, callbacks=None, optimizers=(None, None), processing_class=None, **kwargs):
        self.model = model
        self.args = args
        self.data_collator = data_collator
        self.train_dataset = train_dataset
        self.eval_dataset = eval_dataset
        self.tokenizer = tokenizer if tokenizer is not None else processing_class
        self.callbacks = list(callbacks or [])
        self.optimizer, self.lr_scheduler = optimizers
        self.state = SimpleNamespace(epoch=0.0, global_step=0, max_steps=0, log_history=[])
        xdev = bool(getattr(args, "negatives_x_device", False))
        self._dist_loss_scale_factor = dist.get_world_size() if xdev else 1

    # ------------------------------------------------------------------ helpers
    def is_world_process_zero(self) -> bool:
        return getattr(self.args, "process_index", 0) == 0

    def _world(self):
        return getattr(self.args, "world_size", 1), getattr(self.args, "process_index", 0)

    def _save(self, output_dir: Optional[str] = None):
        output_dir = output_dir if output_dir is not None else self.args.output_dir
        os.makedirs(output_dir, exist_ok=True)
        logger.info("Saving model checkpoint to %s", output_dir)
        self.model.save(output_dir)
        if self.tokenizer is not None:
            self.tokenizer.save_pretrained(output_dir)
        torch.save(self.args, os.path.join(output_dir, TRAINING_ARGS_NAME))

    def save_model(self, output_dir: Optional[str] = None):
        if self.is_world_process_zero():
            self._save(output_dir)

    def _prepare_inputs(self, inputs: Tuple[Dict[str, Union[torch.Tensor, Any]], ...]) -> List[Dict[str, Any]]:
        from ..encoder import TOKEN_ROWS_KEY, token_rows_of
        dev = self.args.device
        out = []
        for x in inputs:
            if isinstance(x, torch.Tensor):
                out.append(x.to(dev, non_blocking=True))
            else:
                moved = {k: (v.to(dev, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in x.items()}
                # packed rows in training (round 5): while the collator's mask is still on the HOST, note how many token rows the
                # batch really has -- the training forward then runs over those instead of B x L (openmatch_amd/train.py)
                if hasattr(x, "keys") and TOKEN_ROWS_KEY not in x and "attention_mask" in x:
                    tokens = token_rows_of(x["attention_mask"])
                    if tokens is not None:
                        moved[TOKEN_ROWS_KEY] = tokens
                out.append(moved)
        return out

    def get_train_dataloader(self) -> DataLoader:
        if self.train_dataset is None:
            raise ValueError("Trainer: training requires a train_dataset.")
        ds = self.train_dataset
        W, r = self._world()
        per_dev = self.args.per_device_train_batch_size
        common = dict(collate_fn=self.data_collator, drop_last=False,
                      num_workers=getattr(self.args, "dataloader_num_workers", 0),
                      pin_memory=getattr(self.args, "dataloader_pin_memory", True))
        if isinstance(ds, IterableDataset):
            if W > 1:      # every rank walks the same stream and keeps its slice of each global batch
                ds = IterableDatasetShard(ds, batch_size=per_dev, drop_last=False, num_processes=W, process_index=r)
            return DataLoader(ds, batch_size=per_dev, **common)
        sampler = None
        if W > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=W, rank=r,
                                                                      seed=getattr(self.args, "seed", 42))
        return DataLoader(ds, batch_size=per_dev, sampler=sampler, shuffle=sampler is None, **common)

    def get_eval_dataloader(self, eval_dataset=None) -> DataLoader:
        ds = eval_dataset if eval_dataset is not None else self.eval_dataset
        if ds is None:
            raise ValueError("Trainer: evaluation requires an eval_dataset.")
        W, r = self._world()
        per_dev = getattr(self.args, "per_device_eval_batch_size", None) or self.args.per_device_train_batch_size
        common = dict(collate_fn=self.data_collator, drop_last=False,
                      num_workers=getattr(self.args, "dataloader_num_workers", 0),
                      pin_memory=getattr(self.args, "dataloader_pin_memory", True))
        if isinstance(ds, IterableDataset):
            if W > 1:
                ds = IterableDatasetShard(ds, batch_size=per_dev, drop_last=False, num_processes=W, process_index=r)
            return DataLoader(ds, batch_size=per_dev, **common)
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=W, rank=r, shuffle=False) if W > 1 else None
        return DataLoader(ds, batch_size=per_dev, sampler=sampler, shuffle=False, **common)

    def evaluate(self, eval_dataset=None, ignore_keys=None, metric_key_prefix: str = "eval") -> Dict[str, float]:
        """Validation loss, what the reference's evaluation is (docs/dr-msmarco-passage.md:85: "evaluation is just
        calculating the loss on the validation set"; driver/train_dr.py:84-97 hands `eval_dataset` to the HF Trainer,
        whose evaluate() averages compute_loss over the eval batches, each weighted by its size, across all ranks).
        The model runs in eval mode (no dropout, no x world_size on the loss: modeling :124-125) without gradients."""
        loader = self.get_eval_dataloader(eval_dataset)
        W, _ = self._world()
        was_training = self.model.training
        self.model.eval()
        tot = torch.zeros(2, dtype=torch.float64, device=self.args.device)       # (sum of loss * queries, queries)
        try:
            with torch.no_grad():
                for batch in loader:
                    inputs = self._prepare_inputs(batch)
                    with self._autocast():
                        loss = self.compute_loss(self.model, inputs)
                    n = next(iter(inputs[0].values())).shape[0] if isinstance(inputs[0], dict) else inputs[0].shape[0]
                    tot[0] += loss.detach().double() * n
                    tot[1] += n
        finally:
            self.model.train(was_training)
        if W > 1:
            dist.all_reduce(tot)
        n_all = float(tot[1])
        metrics = {f"{metric_key_prefix}_loss": float(tot[0]) / n_all if n_all else float("nan"),
                   f"{metric_key_prefix}_samples": int(n_all), "epoch": self.state.epoch, "step": self.state.global_step}
        self.state.log_history.append(metrics)
        if self.is_world_process_zero():
            logger.info("%s", metrics)
        for cb in self.callbacks:                       # HF callbacks that implement on_evaluate (TensorBoardCallback logs via on_log)
            fn = getattr(cb, "on_evaluate", None)
            if callable(fn):
                try:
                    fn(self.args, self.state, None, metrics=metrics)
                except Exception as e:                  # a callback written against HF's TrainerControl must not stop training
                    logger.debug("callback %r.on_evaluate failed: %s", cb, e)
        return metrics

    # ------------------------------------------------------------------ one step
    def compute_loss(self, model, inputs, return_outputs=False, **_unused):
        query, passage = inputs
        outputs = model(query=query, passage=passage)
        return (outputs.loss, outputs) if return_outputs else outputs.loss

    def _autocast(self):
        """--fp16: float16 autocast, as HF Trainer's torch.cuda.amp (trainer/dense_trainer.py:141-149 hands `fp16=`, `scaler=` on):
        BERT-family encoders then train on the float16 kernels with a dynamic loss scale (round 5), T5 on the bfloat16 ones
        (openmatch_amd/encoder.py:training_code).  --bf16: the bfloat16 kernels."""
        from contextlib import nullcontext
        if getattr(self.args, "fp16", False):
            return torch.autocast("cuda", dtype=torch.float16)
        if getattr(self.args, "bf16", False):
            return torch.autocast("cuda", dtype=torch.bfloat16)
        return nullcontext()

    def _loss_scaler(self):
        """The dynamic loss scale of --fp16 runs (None otherwise): created on first use, on the model's device."""
        if not getattr(self.args, "fp16", False) or str(getattr(self.args, "device", "cpu")).startswith("cpu"):
            return None
        if getattr(self, "_scaler", None) is None:
            from ..optim import LossScaler
            self._scaler = LossScaler(self.args.device, init_scale=float(getattr(self.args, "fp16_init_scale", 65536.0)),
                                      growth_interval=int(getattr(self.args, "fp16_growth_interval", 2000)))
        return self._scaler

    def _scaled(self, loss):
        sc = self._loss_scaler()
        return loss if sc is None else loss * sc.scale

    def training_step(self, model, inputs, *_unused) -> torch.Tensor:
        model.train()
        inputs = self._prepare_inputs(inputs)
        with self._autocast():
            loss = self.compute_loss(model, inputs)
        accum = max(1, getattr(self.args, "gradient_accumulation_steps", 1))
        self._scaled(loss / accum).backward()
        return loss.detach() / self._dist_loss_scale_factor

    # ------------------------------------------------------------------ loop
    def create_optimizer_and_scheduler(self, num_training_steps: int):
        a = self.args
        if self.optimizer is None:
            groups = parameter_groups(self.model, getattr(a, "weight_decay", 0.0))
            kw = dict(lr=a.learning_rate, betas=(getattr(a, "adam_beta1", 0.9), getattr(a, "adam_beta2", 0.999)),
                      eps=getattr(a, "adam_epsilon", 1e-8))
            on_gpu = all(p.is_cuda and p.dtype == torch.float32 for g in groups for p in g["params"])
            if on_gpu and any(g["params"] for g in groups):
                # HF Trainer's AdamW + clip_grad_norm_(max_grad_norm) as ONE pass that also refreshes the packed 16-bit weights
                from ..optim import FusedAdamW
                self.optimizer = FusedAdamW(groups, max_grad_norm=float(getattr(a, "max_grad_norm", 0.0) or 0.0), **kw)
            else:
                self.optimizer = torch.optim.AdamW(groups, **kw)
        if self.lr_scheduler is None:
            warm = getattr(a, "warmup_steps", 0) or math.ceil(num_training_steps * getattr(a, "warmup_ratio", 0.0))
            self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(
                self.optimizer, lambda s: linear_schedule_factor(s, warm, num_training_steps))

    def optimizer_step(self, params=None, skip_storages=()):
        """Everything between the last backward of a step and the next forward (HF Trainer.train's inner tail): gradient
        averaging across ranks, clip_grad_norm_(max_grad_norm), optimizer.step, lr_scheduler.step, zero_grad.  With the default
        optimizer (openmatch_amd.optim.FusedAdamW) clipping and the refresh of the packed 16-bit weights are part of its one
        pass; any other optimizer is followed by a re-pack (torch's fused optimizers do not bump parameter versions, so the
        packed-weight cache cannot see their updates)."""
        a = self.args
        W, _ = self._world()
        if params is None:
            params = [p for p in self.model.parameters() if p.requires_grad]
        if W > 1:
            allreduce_mean_(params, W, skip_storages=skip_storages)
        from ..optim import FusedAdamW
        max_norm = getattr(a, "max_grad_norm", 0.0)
        scaler = self._loss_scaler()
        if isinstance(self.optimizer, FusedAdamW):
            self.optimizer.max_grad_norm = float(max_norm or 0.0)
            if scaler is not None:               # unscale, skip a non-finite step, update the scale: all on the device
                scaler.attach(self.optimizer)
            self.optimizer.step()
            if scaler is not None:
                scaler.update(self.optimizer)
        else:
            skip = False
            if scaler is not None:               # a foreign optimizer under --fp16: unscale in place, skip on inf / nan (one host read)
                grads = [p.grad for p in params if p.grad is not None]
                torch._foreach_mul_(grads, scaler.inv_scale[0])
                total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g_) for g_ in grads]))
                skip = not bool(torch.isfinite(total))
                st = scaler.state
                if skip:
                    st[0] = torch.clamp(st[0] * 0.5, min=1.0); st[2] = 0; st[3] += 1
                else:
                    st[2] += 1
                    if float(st[2]) >= scaler.growth_interval:
                        st[0] = torch.clamp(st[0] * 2.0, max=16777216.0); st[2] = 0
                st[1] = 1.0 / st[0]
            if not skip:
                if max_norm and max_norm > 0:
                    torch.nn.utils.clip_grad_norm_(params, max_norm)
                self.optimizer.step()
            from ..encoder import invalidate_packed
            invalidate_packed(self.model)
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        self.optimizer.zero_grad(set_to_none=True)

    def _rewind_scheduler_for_skipped_steps(self):
        """--fp16: HF Trainer steps the LR scheduler only when GradScaler let the optimizer step run.  Here the skip happens on the
        device (no host read per step), so `optimizer_step` always steps the scheduler; at every log line -- where the host reads the
        running loss anyway -- the scheduler is set back by the steps the scaler skipped since the last look, so the schedule lags
        the reference's by at most `logging_steps` steps after an overflow."""
        scaler = getattr(self, "_scaler", None)
        if scaler is None or self.lr_scheduler is None:
            return
        skipped = scaler.skipped_steps()
        new = skipped - getattr(self, "_skipped_seen", 0)
        self._skipped_seen = skipped
        if new > 0 and hasattr(self.lr_scheduler, "last_epoch"):
            self.lr_scheduler.last_epoch = max(-1, self.lr_scheduler.last_epoch - new - 1)
            self.lr_scheduler.step()               # recomputes the groups' lr at the rewound position

    def _num_steps(self, loader):
        a = self.args
        accum = max(1, getattr(a, "gradient_accumulation_steps", 1))
        try:                                  # as HF Trainer: the epoch length is known whenever the loader has one,
            per_epoch = max(1, len(loader) // accum)      # max_steps or not (it feeds the fractional state.epoch)
        except TypeError:
            per_epoch = None
        if getattr(a, "max_steps", -1) and a.max_steps > 0:
            return a.max_steps, per_epoch
        if per_epoch is None:
            raise ValueError("args.max_steps must be set for a dataset without a length")
        return int(math.ceil(a.num_train_epochs * per_epoch)), per_epoch

    def train(self, resume_from_checkpoint=None, **_unused):
        a = self.args
        W, _ = self._world()
        self.model.to(a.device)
        loader = self.get_train_dataloader()
        total, per_epoch = self._num_steps(loader)
        self.create_optimizer_and_scheduler(total)
        self.state.max_steps = total
        params = [p for p in self.model.parameters() if p.requires_grad]
        accum = max(1, getattr(a, "gradient_accumulation_steps", 1))
        log_every = max(1, int(getattr(a, "logging_steps", 500) or 500))
        save_every = int(getattr(a, "save_steps", 0) or 0)
        # the running loss stays on the device between log lines: float(loss) every step is a stream synchronisation per step,
        # and the host then cannot enqueue step n + 1 while step n still runs
        running, micro, epoch = torch.zeros((), dtype=torch.float32, device=a.device), 0, 0
        self.optimizer.zero_grad(set_to_none=True)
        # multi-GPU: gradient all-reduce in layer-group buckets from inside the backward (grad_sync.py) -- with gradient
        # accumulation or the gradient cache the per-parameter gradients are sums over several backward passes, which are
        # reduced once after the last one instead
        from ..grad_sync import GradSync
        sync = GradSync(W, getattr(a, "grad_bucket_layers", 4)) if (
            W > 1 and accum == 1 and type(self).training_step is DRTrainer.training_step
            and getattr(a, "overlap_grad_allreduce", True)) else None
        if W > 1 and dist.is_initialized():
            _control_group()                  # made here, where every rank is at the same point
        if resume_from_checkpoint:
            logger.warning("resume_from_checkpoint=%r is not supported by this trainer (optimizer state is not "
                           "checkpointed): training starts from the model's current weights", resume_from_checkpoint)
        # --evaluation_strategy steps|epoch with --eval_steps (docs/dr-msmarco-passage.md:71-85); HF renamed the field to eval_strategy
        strategy = str(getattr(a, "evaluation_strategy", None) or getattr(a, "eval_strategy", "no")).lower().split(".")[-1]
        eval_every = int(getattr(a, "eval_steps", 0) or 0) or log_every
        do_eval = self.eval_dataset is not None and strategy in ("steps", "epoch")
        if strategy in ("steps", "epoch") and self.eval_dataset is None:
            raise ValueError("Trainer: evaluation requires an eval_dataset.")
        while self.state.global_step < total:
            # the datasets read int(trainer.state.epoch) when their iterator is created (train_dataset.py:115-119):
            # the epoch counter must already say `epoch` here, as HF Trainer's does, not after the first step
            self.state.epoch = float(epoch)
            if hasattr(self.train_dataset, "set_epoch"):
                self.train_dataset.set_epoch(epoch)
            if hasattr(getattr(loader, "sampler", None), "set_epoch"):
                loader.sampler.set_epoch(epoch)
            stepped, in_epoch = False, 0
            for batch in loader:
                if sync is not None:
                    sync.begin()
                try:
                    loss_t = self.training_step(self.model, batch)
                finally:                          # a raising step must not leave the bucket hand-off armed for later backwards
                    if sync is not None:
                        sync.finish()
                running += loss_t.detach().to(running.dtype)
                micro += 1
                if micro % accum:
                    continue
                self.optimizer_step(params, skip_storages=sync.reduced if sync is not None else ())
                self.state.global_step += 1
                stepped = True
                in_epoch += 1
                if per_epoch:                 # HF: epoch + (steps done in this epoch) / (steps per epoch)
                    self.state.epoch = epoch + min(1.0, in_epoch / per_epoch)
                if self.state.global_step % log_every == 0:
                    self._rewind_scheduler_for_skipped_steps()
                    entry = {"loss": float(running) / (log_every * accum), "learning_rate": self.lr_scheduler.get_last_lr()[0],
                             "epoch": self.state.epoch, "step": self.state.global_step}
                    self.state.log_history.append(entry)
                    if self.is_world_process_zero():
                        logger.info("%s", entry)
                    running.zero_()
                if do_eval and strategy == "steps" and self.state.global_step % eval_every == 0:
                    self.evaluate()
                if save_every and self.state.global_step % save_every == 0 and self.is_world_process_zero():
                    self._save(os.path.join(a.output_dir, f"checkpoint-{self.state.global_step}"))
                if self.state.global_step >= total:
                    break
            if do_eval and strategy == "epoch" and stepped:
                self.evaluate()
            epoch += 1
            if not stepped:
                raise ValueError("the training dataloader produced no batches")
        return SimpleNamespace(global_step=self.state.global_step, training_loss=float(running))


def split_dense_inputs(model_input: dict, chunk_size: int):
    """{"query": {k: [B,...]}} -> list of {"query": {k: [chunk,...]}} (reference :111-120)."""
    assert len(model_input) == 1
    (arg_key, arg_val), = model_input.items()
    keys = [k for k in arg_val.keys() if torch.is_tensor(arg_val[k])]     # (the host-side per-sequence token counts split with the rows)
    pieces = zip(*[arg_val[k].split(chunk_size, dim=0) for k in keys])
    return [{arg_key: dict(zip(keys, piece))} for piece in pieces]


def get_dense_rep(x):
    return x.p_reps if x.q_reps is None else x.q_reps


class GCDenseTrainer(DRTrainer):
    """Gradient-cache variant (reference :130-160, via the un-vendored `grad_cache` package):
    representations are computed chunk by chunk without a tape, the contrastive loss and its
    gradient w.r.t. every representation are computed once on the full (all-gathered) batch, then
    each chunk is re-encoded WITH the tape and back-propagated with its cached representation
    gradient.  Mathematically identical to the full-batch step, peak activation memory is one chunk."""

    def training_step(self, model, inputs, *_unused) -> torch.Tensor:
        from ..ops import contrastive_loss
        model.train()
        queries, passages = self._prepare_inputs(inputs)
        a = self.args
        q_chunks = split_dense_inputs({"query": queries}, a.gc_q_chunk_size)
        p_chunks = split_dense_inputs({"passage": passages}, a.gc_p_chunk_size)
        xdev = bool(getattr(a, "negatives_x_device", False))
        with self._autocast():
            # dropout must repeat between the two passes: fix the generator state per chunk
            states = []
            with torch.no_grad():
                reps = []
                for ch in q_chunks + p_chunks:
                    states.append(torch.random.get_rng_state())
                    reps.append(get_dense_rep(model(**ch)))
            nq = len(q_chunks)
            q_reps = torch.cat(reps[:nq]).requires_grad_()
            p_reps = torch.cat(reps[nq:]).requires_grad_()
            if xdev:
                q_all, p_all = model.dist_gather_tensor(q_reps), model.dist_gather_tensor(p_reps)
                q0, p0 = model.process_rank * q_reps.shape[0], model.process_rank * p_reps.shape[0]
                scale = float(model.world_size)
            else:
                q_all, p_all, q0, p0, scale = q_reps, p_reps, 0, 0, 1.0
            n_psg = p_all.shape[0] // q_all.shape[0]
            loss, _ = contrastive_loss(q_all, p_all, n_psg, scale, q_reps, q0, p_reps, p0)
            self._scaled(loss).backward()          # --fp16: the cached representation gradients carry the loss scale into every chunk's backward
            grads = list(q_reps.grad.split(a.gc_q_chunk_size)) + list(p_reps.grad.split(a.gc_p_chunk_size))
            for ch, st, gr in zip(q_chunks + p_chunks, states, grads):
                torch.random.set_rng_state(st)
                get_dense_rep(model(**ch)).backward(gr)
        return loss.detach() / self._dist_loss_scale_factor
