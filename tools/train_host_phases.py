#!/usr/bin/env python
"""Host-side time of the phases of one training step (no synchronisation inside a step: what the HOST needs to enqueue each phase)
next to the same phases with a device synchronisation after each (what the DEVICE needs).  A phase whose host time exceeds the
device time of the phase before it leaves the device idle.   python tools/train_host_phases.py [--precision f16]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--fine", action="store_true", help="also time the pieces of the forward on the host (wrappers around the functions)")
    a = ap.parse_args()
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch.trainer import DRTrainer
    dev = "cuda:0"
    torch.manual_seed(0)
    lm = BertModel(BertConfig())
    model = DRModel(lm_q=lm, lm_p=lm, pooling="first",
                    model_args=NS(encoder_only=False, dtype={"bf16": "bfloat16", "f16": "float16"}[a.precision]),
                    data_args=NS(train_n_passages=8), train_args=NS(negatives_x_device=False, per_device_train_batch_size=8)).to(dev)
    g = torch.Generator().manual_seed(1)
    mk = lambda n, L: {"input_ids": torch.randint(1000, 30000, (n, L), generator=g).to(dev), "attention_mask": torch.ones(n, L, dtype=torch.long).to(dev)}
    batch = (mk(8, 32), mk(64, 128))
    args = NS(device=dev, world_size=1, process_index=0, per_device_train_batch_size=8, negatives_x_device=False, learning_rate=5e-6,
              weight_decay=0.0, gradient_accumulation_steps=1, max_grad_norm=1.0, fp16=a.precision == "f16", bf16=False)
    trainer = DRTrainer(model=model, args=args)
    trainer.create_optimizer_and_scheduler(num_training_steps=10 ** 6)
    model.train()
    fine = {}
    if a.fine:
        import functools
        from openmatch_amd import train as T, native as N, ops as O, encoder as E
        def wrap(owner, name, label):
            f = getattr(owner, name)
            @functools.wraps(f)
            def g(*x, **k):
                t0 = time.perf_counter()
                try:
                    return f(*x, **k)
                finally:
                    fine[label] = fine.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
            setattr(owner, name, g)
        lib = N.lib()
        for nm in ("om_encoder_train_forward", "om_encoder_train_backward", "om_contrastive_fwd_bwd_ex", "om_encoder_tape_bytes",
                   "om_encoder_train_workspace_bytes", "om_adamw_step", "om_grad_sqnorm"):
            if hasattr(lib, nm):
                wrap(lib, nm, "C:" + nm)
        wrap(T, "packed_weights", "packed_weights"); wrap(T, "_bert_params", "_bert_params"); wrap(T, "encode_train", "encode_train (all)")
        wrap(T, "_encoder_grad_arena", "_encoder_grad_arena"); wrap(N.Workspace, "get", "Workspace.get")
        wrap(type(model), "_encode_one_pass", "DRModel._encode_one_pass"); wrap(O, "contrastive_loss", "ops.contrastive_loss")
        wrap(torch, "empty", "torch.empty")

    def step(sync, acc):
        t = [time.perf_counter()]
        def mark():
            if sync:
                torch.cuda.synchronize()
            t.append(time.perf_counter())
        with trainer._autocast():
            loss = trainer.compute_loss(model, batch)
        mark()
        scaled = trainer._scaled(loss)
        mark()
        scaled.backward()
        mark()
        trainer.optimizer_step()
        mark()
        for i, k in enumerate(("forward+loss", "scale", "backward", "optimizer")):
            acc[k] = acc.get(k, 0.0) + (t[i + 1] - t[i]) * 1e3
    out = {}
    for sync in (False, True):
        for _ in range(3):
            step(sync, {})
        acc = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            step(sync, acc)
        torch.cuda.synchronize(); total = (time.perf_counter() - t0) / a.steps * 1e3
        out["device_ms" if sync else "host_enqueue_ms"] = {**{k: round(v / a.steps, 3) for k, v in acc.items()}, "step": round(total, 3)}
        if a.fine:
            out["fine_" + ("sync" if sync else "async")] = {k: round(v / (a.steps + 3), 3) for k, v in sorted(fine.items(), key=lambda kv: -kv[1])}
            fine.clear()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
