"""Worker of tests/test_multigpu.py: one process per GPU under torch.distributed.run (backend nccl = RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from types import SimpleNamespace as NS
    from openmatch_amd.comm import RcclComm
    from openmatch_amd.index import FlatIPIndex, sharded_topk

    # 1. the C-ABI collectives against torch.distributed's
    comm = RcclComm.from_torch_distributed(dev)
    x = torch.full((3, 5), float(rank), device=dev) + torch.arange(5, device=dev)
    ref = torch.empty(world * 3, 5, device=dev)
    dist.all_gather_into_tensor(ref, x)
    assert torch.equal(comm.allgather_rows(x), ref)
    g = torch.arange(1000, device=dev, dtype=torch.float32) * (rank + 1)
    want = g.clone(); dist.all_reduce(want, op=dist.ReduceOp.AVG)
    comm.allreduce_grads_(g, average=True)
    assert torch.allclose(g, want)

    # 2. sharded exact search (real FlatIPIndex + om_topk_merge, candidates exchanged by query range) == one index
    rng = np.random.default_rng(0)
    n, d, nq, k = 40000, 128, 37, 100
    P = rng.standard_normal((n, d)).astype(np.float32)
    Q = torch.from_numpy(rng.standard_normal((nq, d)).astype(np.float32)).to(dev)
    lo, hi = n * rank // world, n * (rank + 1) // world
    shard = FlatIPIndex(d, device=dev, precision="f32"); shard.add(P[lo:hi])
    whole = FlatIPIndex(d, device=dev, precision="f32"); whole.add(P)
    Dw, Iw = whole.search_device(Q, k)
    for mode in ("", "native"):
        os.environ["OPENMATCH_AMD_COMM"] = mode
        Dm, Im, blk = sharded_topk(shard, Q, k, lo)
        rows = slice(rank * blk, min((rank + 1) * blk, nq))
        m = rows.stop - rows.start
        if m > 0:
            assert torch.equal(Im[:m], Iw[rows]) and torch.equal(Dm[:m], Dw[rows]), (mode, rank)
    os.environ["OPENMATCH_AMD_COMM"] = ""

    # 3. contrastive step with cross-device negatives: every rank's loss and the averaged gradients equal the single-process
    #    full-batch step (reference semantics: modeling :105-125, trainer/dense_trainer.py:107-108)
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch_amd.trainer.dense_trainer import allreduce_mean_
    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=600,
                     max_position_embeddings=160, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    ref_lm = BertModel(cfg); ref_lm.load_state_dict(lm.state_dict())
    g_ = torch.Generator().manual_seed(3)
    q_all = {"input_ids": torch.randint(300, 600, (2 * world, 32), generator=g_), "attention_mask": torch.ones(2 * world, 32, dtype=torch.long)}
    p_all = {"input_ids": torch.randint(300, 600, (4 * world, 128), generator=g_), "attention_mask": torch.ones(4 * world, 128, dtype=torch.long)}
    part = lambda t, n_: {k_: v[rank * n_:(rank + 1) * n_].to(dev) for k_, v in t.items()}
    model = DRModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="float32"),
                    data_args=NS(train_n_passages=2),
                    train_args=NS(negatives_x_device=True, per_device_train_batch_size=2)).to(dev).train()
    # the gradient all-reduce runs in layer buckets from inside the backward (openmatch_amd/grad_sync.py), as DRTrainer does
    from openmatch_amd.grad_sync import GradSync
    sync = GradSync(world, bucket_layers=1)
    out = model(query=part(q_all, 2), passage=part(p_all, 4))
    sync.begin()
    out.loss.backward()
    sync.finish()
    assert len(sync.reduced) >= 1                      # the encoder's arena(s) went through the bucketed path
    params = [p for p in model.parameters() if p.grad is not None]
    allreduce_mean_(params, world, skip_storages=sync.reduced)
    full = DRModel(lm_q=ref_lm, lm_p=ref_lm, pooling="first", model_args=NS(encoder_only=False, dtype="float32"),
                   data_args=NS(train_n_passages=2),
                   train_args=NS(negatives_x_device=False, per_device_train_batch_size=2 * world)).to(dev).train()
    fo = full(query={k_: v.to(dev) for k_, v in q_all.items()}, passage={k_: v.to(dev) for k_, v in p_all.items()})
    fo.loss.backward()
    assert abs(out.loss.item() / world - fo.loss.item()) < 1e-4, (out.loss.item(), fo.loss.item())
    for (name, a), (_, b) in zip(lm.named_parameters(), ref_lm.named_parameters()):
        if b.grad is None:
            continue
        rel = ((a.grad - b.grad).norm() / (b.grad.norm() + 1e-20)).item()
        assert rel < 2e-3 or (a.grad - b.grad).abs().max().item() < 1e-7, (name, rel)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("MGPU-OK")


if __name__ == "__main__":
    main()
