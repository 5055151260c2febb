"""AdamW with global-norm clipping for models whose forward runs on packed HIP weights (`om_grad_sqnorm`, `om_adamw_step`).

What the reference's training loop does per optimizer step through HF Trainer (trainer/dense_trainer.py:27-108 inherits
`Trainer.train`: `clip_grad_norm_(max_grad_norm)` -> `AdamW.step()`), in two passes over memory: one kernel reads every
gradient once for the norm, one kernel reads g, p, m, v and writes p, m, v AND the packed compute-dtype copies of the weights
the next HIP forward reads (openmatch_amd/encoder.py keeps Q|K|V fused and the matrices in bf16 / f16; re-packing them with
torch ops after every step is ~100 launches).  Same arithmetic as `torch.optim.AdamW` (tests/test_gpu_parity.py compares).

It is a `torch.optim.Optimizer`: `param_groups` (lr, betas, eps, weight_decay) drive it, LR schedulers step it, `state_dict()`
holds `step`, `exp_avg`, `exp_avg_sq` per parameter.  CUDA float32 parameters only -- anything else raises (no CPU fallback).
"""
import ctypes as C
import math
import struct

import torch

from . import native as N

_DT = {torch.float32: N.OM_F32, torch.bfloat16: N.OM_BF16, torch.float16: N.OM_F16}


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = float(max_grad_norm or 0.0)
        self.grad_scale = 1.0                # gradients are multiplied by it (a static 1 / loss scale)
        self.scale_state = None              # LossScaler.state: gradients are divided by the dynamic loss scale, skipped steps are not counted
        self.skip_nonfinite = False          # True: an inf / nan gradient norm skips the step (GradScaler semantics)
        self._plan = None
        self._norm_sq = None                 # device scalar of the last step: sum of squares of the raw gradients (before grad_scale and clipping)

    # ------------------------------------------------------------------------------------------------------------
    def _state_of(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def _build_plan(self, entries, device):
        """entries: [(param, grad, weight_decay)] -> device tables (one record per tensor, one pair per chunk)."""
        from . import encoder as enc
        recs, chunks, keep, refreshed, stale = [], [], [], [], []
        for t_idx, (p, g, wd) in enumerate(entries):
            st = self._state_of(p)
            shadows = []
            for pk, buf, off in enc.shadows_of(p):
                if buf.dtype in _DT and buf.device == p.device and len(shadows) < 2:
                    shadows.append((buf.data_ptr() + off * buf.element_size(), _DT[buf.dtype]))
                    refreshed.append(pk)
                else:
                    stale.append(pk)
            while len(shadows) < 2:
                shadows.append((0, 0))
            recs.append(struct.pack("<6Qqf3i", p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                    st["exp_avg_sq"].data_ptr(), shadows[0][0], shadows[1][0], p.numel(), float(wd),
                                    shadows[0][1], shadows[1][1], 0))
            chunks += [(t_idx, c) for c in range((p.numel() + N.ADAM_CHUNK - 1) // N.ADAM_CHUNK)]
            keep.append((p, g))
        assert len(recs[0]) == C.sizeof(N.OmAdamTensor)
        tab = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(device)
        ch = torch.tensor(chunks, dtype=torch.int32).reshape(-1).to(device)
        partial = torch.empty(max(1, len(chunks)), dtype=torch.float32, device=device)
        refreshed = list({id(x): x for x in refreshed}.values())
        stale = [x for x in {id(x): x for x in stale}.values() if all(x is not r for r in refreshed)]
        return dict(tab=tab, chunks=ch, n_chunks=len(chunks), partial=partial, refreshed=refreshed, stale=stale)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import encoder as enc
        lib = N.lib()
        # one launch per distinct (lr, betas, eps, step count) -- normally one: HF's two groups differ in weight decay only, and
        # every parameter that has a gradient has had one in every step.  Parameters without a gradient are left alone
        # (torch.optim.AdamW skips them too: no decay, no step count).
        launches = {}
        for group in self.param_groups:
            for p in group["params"]:
                g = p.grad
                if not p.requires_grad or g is None:
                    continue
                if g.is_sparse or g.dtype != torch.float32 or not g.is_contiguous():
                    raise N.NativeError("FusedAdamW takes contiguous float32 gradients")
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or g.device != p.device:
                    raise N.NativeError("FusedAdamW updates contiguous float32 parameters on an MI355X device; there is no CPU fallback")
                key = (float(group["lr"]), tuple(group["betas"]), float(group["eps"]), self._state_of(p)["step"] + 1)
                launches.setdefault(key, []).append((p, g, float(group["weight_decay"])))
        self._last_step_ran = False              # LossScaler.update judges a step only if one ran
        if not launches:
            return loss
        all_entries = [e for es in launches.values() for e in es]
        device = all_entries[0][0].device
        if any(e[0].device != device for e in all_entries):
            raise N.NativeError("FusedAdamW: all parameters must live on one device (one process per GPU)")
        sig = tuple((p.data_ptr(), g.data_ptr(), wd, tuple((id(pk), b.data_ptr()) for pk, b, _ in enc.shadows_of(p)))
                    for p, g, wd in all_entries) + tuple(len(es) for es in launches.values())
        if self._plan is None or self._plan["sig"] != sig:
            groups = [self._build_plan(es, device) for es in launches.values()]
            self._plan = dict(sig=sig, groups=groups, all=groups[0] if len(groups) == 1 else self._build_plan(all_entries, device))
        if self._norm_sq is None or self._norm_sq.device != device:
            self._norm_sq = torch.zeros(1, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            stream = N.stream_ptr(device)
            need_norm = self.max_grad_norm > 0 or self.skip_nonfinite
            if need_norm:
                pl = self._plan["all"]
                N.check(lib.om_grad_sqnorm(N.ptr(pl["tab"]), N.ptr(pl["chunks"]), pl["n_chunks"], N.ptr(pl["partial"]),
                                           N.ptr(self._norm_sq), stream))
            for (key, entries), pl in zip(launches.items(), self._plan["groups"]):
                lr, (b1, b2), eps, step_no = key
                for p, _, _ in entries:
                    self.state[p]["step"] = step_no
                N.check(lib.om_adamw_step(N.ptr(pl["tab"]), N.ptr(pl["chunks"]), pl["n_chunks"], lr, b1, b2, eps, step_no,
                                          N.ptr(self._norm_sq) if need_norm else None, self.max_grad_norm,
                                          float(self.grad_scale), int(self.skip_nonfinite),
                                          N.ptr(self.scale_state) if self.scale_state is not None else None, stream))
                enc.after_inplace_update(pl["refreshed"], pl["stale"])
        self._last_step_ran = True
        return loss

    def grad_norm(self):
        """|g * grad_scale|_2 of the last step (before clipping), as a device scalar -- what clip_grad_norm_ returns."""
        if self._norm_sq is None:
            return None
        n = self._norm_sq[0].sqrt() * abs(self.grad_scale)
        return n * self.scale_state[1] if self.scale_state is not None else n


class LossScaler:
    """The dynamic loss scale of float16 training, on the device (what torch.cuda.amp.GradScaler is to HF Trainer's --fp16, which
    the reference inherits: trainer/dense_trainer.py:141-149): `scale` multiplies the loss before backward; FusedAdamW divides
    it out of the gradients (`scale_state`), skips a step whose gradients are not finite, and `update()` halves the scale after
    such a step / doubles it after `growth_interval` clean ones -- without a host synchronisation."""

    def __init__(self, device, init_scale=65536.0, growth_interval=2000):
        self.state = torch.tensor([init_scale, 1.0 / init_scale, 0.0, 0.0], dtype=torch.float32, device=device)
        self.growth_interval = int(growth_interval)

    @property
    def scale(self):                 # device scalar (view)
        return self.state[0]

    @property
    def inv_scale(self):
        return self.state[1:2]

    def attach(self, optimizer):
        """The skipped-step count (state[3]) corrects the ATTACHED optimizer's bias corrections (its `state[p]['step']` counts
        attempted steps, the kernel subtracts the skipped ones): a different optimizer starts from a count of zero."""
        import weakref
        prev = getattr(self, "_attached", None)
        if prev is not None and prev() is not None and prev() is not optimizer:
            self._skipped_before = getattr(self, "_skipped_before", 0) + int(self.state[3].item())
            self.state[3].zero_()
        self._attached = weakref.ref(optimizer)
        optimizer.scale_state = self.state
        optimizer.skip_nonfinite = True

    def update(self, optimizer):
        """after optimizer.step(): reads the step's squared gradient norm where the optimizer left it"""
        if optimizer._norm_sq is None or not getattr(optimizer, "_last_step_ran", True):
            return                           # step() returned early (no parameter had a gradient): nothing to judge, the scale stands
        with torch.cuda.device(self.state.device):
            N.check(N.lib().om_loss_scale_update(N.ptr(optimizer._norm_sq), N.ptr(self.state), self.growth_interval,
                                                 N.stream_ptr(self.state.device)))

    def skipped_steps(self):
        return int(self.state[3].item()) + getattr(self, "_skipped_before", 0)
