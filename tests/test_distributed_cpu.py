"""CPU, 2 processes over gloo: the multi-rank host logic (query all-gather, shard offsets, per-shard
top-k gather + merge, cross-device negatives, gradient averaging).  The per-shard search and the
merge are HIP on a GPU box; here they are swapped for the CPU oracle so the ORCHESTRATION is what
is under test (the kernels themselves are covered by the -m gpu tests)."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import NS


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleShard:
    """FlatIPIndex stand-in with the same methods, backed by oracle/flatip.py."""

    def __init__(self, d, device=None, precision=None):
        from oracle import flatip
        self.d, self._idx = d, flatip.IndexFlatIP(d)

    ntotal = property(lambda self: self._idx.ntotal)

    def add(self, x):
        self._idx.add(x.cpu().numpy() if isinstance(x, torch.Tensor) else x)

    def search_device(self, queries, k, id_offset=0):
        D, I = self._idx.search(queries.cpu().numpy(), k)
        I = np.where(I >= 0, I + id_offset, -1)
        return torch.from_numpy(D), torch.from_numpy(I)

    def search(self, x, k):
        return self._idx.search(x, k)


def _merge_cpu(ps, pi, k_out):
    W, Q, k = ps.shape
    flat_s = ps.permute(1, 0, 2).reshape(Q, W * k)
    flat_i = pi.permute(1, 0, 2).reshape(Q, W * k)
    flat_s = torch.where(flat_i >= 0, flat_s, torch.full_like(flat_s, -3.4e38))
    order = torch.sort(flat_s, dim=1, descending=True, stable=True).indices[:, :k_out]
    return torch.gather(flat_s, 1, order), torch.gather(flat_i, 1, order)


def _search_worker(rank, world, port, tmp, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import openmatch_amd.retriever.dense_retriever as R
    R.FlatIPIndex, R.merge_topk = _OracleShard, _merge_cpu
    args = NS(device="cpu", output_dir=tmp, world_size=world, process_index=rank, local_process_index=rank, fp16=False)
    retriever = R.Retriever.from_embeddings(torch.nn.Linear(1, 1), args)        # sharded: rank r loads file r
    assert retriever._sharded and retriever.index.ntotal == 500
    result = retriever.search(100)
    if rank == 0:
        out_q.put(result)
    else:
        assert result == {}
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_equals_single_index(tmp_path, golden):
    from oracle import flatip, retrieval_ref
    g = golden("retrieval_1k")
    doc_ids, qry_ids = list(g["doc_ids"]), list(g["qry_ids"])
    for r in range(2):
        with open(tmp_path / f"embeddings.corpus.rank.{r}", "wb") as f:
            pickle.dump((g["P"][r * 500:(r + 1) * 500], doc_ids[r * 500:(r + 1) * 500]), f, protocol=4)
        with open(tmp_path / f"embeddings.query.rank.{r}", "wb") as f:
            pickle.dump((g["Q"][r * 50:(r + 1) * 50], qry_ids[r * 50:(r + 1) * 50]), f, protocol=4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_search_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    [p.start() for p in procs]
    result = q.get(timeout=180)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    single = flatip.IndexFlatIP(768); single.add(g["P"])
    D, I = single.search(g["Q"], 100)
    want = retrieval_ref.search_to_dict(D, I, doc_ids, qry_ids)
    assert list(result) == list(want)
    for qid in want:
        assert list(result[qid]) == list(want[qid])                      # same ids, same order
        assert np.allclose(list(result[qid].values()), list(want[qid].values()), atol=1e-6)


def _train_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openmatch.modeling import DRModel
    from openmatch_amd.trainer.dense_trainer import allreduce_mean_
    from oracle import retrieval_ref
    model = DRModel(lm_q=torch.nn.Linear(1, 1), lm_p=torch.nn.Linear(1, 1),
                    train_args=NS(negatives_x_device=True, per_device_train_batch_size=2))
    assert (model.process_rank, model.world_size) == (rank, world)
    torch.manual_seed(rank)
    t = torch.randn(2, 8, requires_grad=True)
    gathered = model.dist_gather_tensor(t)
    assert gathered.shape == (2 * world, 8) and torch.equal(gathered[rank * 2:(rank + 1) * 2], t)
    gathered.sum().backward()
    assert torch.equal(t.grad, torch.ones_like(t))                          # gradient only through the local slot
    parts = [torch.randn(2, 8, generator=torch.Generator().manual_seed(r)) for r in range(world)]
    assert torch.allclose(gathered.detach(), torch.cat(parts))              # rank-major order
    # gradient averaging == DDP mean
    p = torch.nn.Parameter(torch.zeros(3)); p.grad = torch.full((3,), float(rank + 1))
    mean_rank = (world + 1) / 2.0                       # mean over ranks of (rank + 1)
    allreduce_mean_([p], world)
    assert torch.allclose(p.grad, torch.full((3,), mean_rank))
    # gradients that are views of ONE arena (what the HIP backward hands autograd): averaged in place with a single
    # collective over the arena span -- the views keep their storage, padding between them stays zero
    arena = torch.zeros(64)
    shapes, views, off = [(3, 4), (5,), (2, 3)], [], 0
    for shp in shapes:
        n = int(np.prod(shp))
        views.append(arena[off:off + n].view(*shp)); off += (n + 15) // 16 * 16
    params = [torch.nn.Parameter(torch.zeros(*shp)) for shp in shapes]
    for i, (q_, v) in enumerate(zip(params, views)):
        v.fill_(float((rank + 1) * (i + 1))); q_.grad = v
    ptr = arena.data_ptr()
    allreduce_mean_(params, world)
    for i, q_ in enumerate(params):
        assert q_.grad.data_ptr() >= ptr and q_.grad.untyped_storage().data_ptr() == arena.untyped_storage().data_ptr()
        assert torch.allclose(q_.grad, torch.full_like(q_.grad, mean_rank * (i + 1)))
    assert float(arena[12:16].abs().sum()) == 0.0
    # bucketed all-reduce from inside the backward (openmatch_amd/grad_sync.py) == one all-reduce over the whole arena:
    # an arena laid out as train.py lays it out (embeddings | layers 0..4 | head), buckets of two layers in completion order
    from openmatch_amd.grad_sync import GradSync
    gen = torch.Generator().manual_seed(100 + rank)
    sizes = [192, 128, 128, 128, 128, 128, 64]                # embeddings, five layers, head
    arena2 = torch.randn(sum(sizes), generator=gen)
    starts = np.cumsum([0] + sizes)
    bounds = [(int(starts[i + 1]), int(starts[i + 2])) for i in range(5)]
    sync = GradSync(world, bucket_layers=2)
    bk = sync.buckets(bounds, arena2.numel())
    assert bk == [(576, 896, 3), (320, 576, 1), (192, 320, 0), (0, 192, 5)], bk      # top bucket carries the head; embeddings last
    covered = torch.zeros(arena2.numel(), dtype=torch.int32)
    for lo, hi, _ in bk:
        covered[lo:hi] += 1
    assert bool((covered == 1).all())
    single = arena2.clone()
    dist.all_reduce(single); single /= world
    sync.begin()
    from openmatch_amd import grad_sync
    assert grad_sync.active() is sync
    sync.reduce_arena(arena2, bounds, None)
    sync.finish()
    # (two ranks: a sum of two terms has one order; eight: the bucketed segments ride other ring chunks than the whole arena's, so the
    # last bits may differ)
    assert grad_sync.active() is None and (torch.equal(arena2, single) if world == 2 else torch.allclose(arena2, single, rtol=1e-5, atol=1e-6))
    # ... and the trainer's post-hoc reduction leaves such an arena alone
    pz = torch.nn.Parameter(torch.zeros(4)); pz.grad = arena2[:4]
    before = pz.grad.clone()
    allreduce_mean_([pz], world, skip_storages=sync.reduced)
    assert torch.equal(pz.grad, before)
    # ranks that disagree on which parameters hold a gradient fail loudly on EVERY rank (no hang in a collective of
    # unequal sizes)
    pa_, pb_ = torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(7))
    pa_.grad = torch.ones(5)
    if rank == 0:
        pb_.grad = torch.ones(7)
    try:
        allreduce_mean_([pa_, pb_], world)
        raise AssertionError("mismatched gradient patterns were not detected")
    except RuntimeError as e:
        assert "different gradient patterns" in str(e)
    # loss convention: every rank computes world * mean-CE over the gathered batch; after the mean
    # all-reduce the parameter gradient equals that of the plain global mean-CE
    qs = [torch.randn(2, 8, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    ps = [torch.randn(4, 8, generator=torch.Generator().manual_seed(20 + r)) for r in range(world)]
    ql, pl = qs[rank].clone().requires_grad_(), ps[rank].clone().requires_grad_()
    qa = retrieval_ref.gather_with_local_grad([ql if r == rank else qs[r] for r in range(world)], rank)
    pa = retrieval_ref.gather_with_local_grad([pl if r == rank else ps[r] for r in range(world)], rank)
    loss, _ = retrieval_ref.contrastive_loss(qa, pa, 2, scale=float(world))
    loss.backward()
    qf, pf = torch.cat(qs).requires_grad_(), torch.cat(ps).requires_grad_()
    full, _ = retrieval_ref.contrastive_loss(qf, pf, 2)
    full.backward()
    assert torch.allclose(ql.grad / world, qf.grad[rank * 2:(rank + 1) * 2], atol=1e-6)
    assert torch.allclose(pl.grad / world, pf.grad[rank * 4:(rank + 1) * 4], atol=1e-6)
    out_q.put((rank, float(loss)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_cross_device_negatives_and_gradient_averaging(world):
    """World size 2, and (round 6) the driver's 8: gather order, local-slot gradients, arena / bucketed / loose all-reduce, the
    gradient-pattern check and the loss convention -- so that the first 8-GPU run does not spend its lease on host-side bugs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=300) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert all(got[r] == pytest.approx(got[0], abs=1e-6) for r in range(world))      # identical loss on every rank


class _ToyData(torch.utils.data.Dataset):
    """43 samples of (x [6], y [2]) from one seeded generator: 43 = 8 * 5 + 3, so the distributed sampler pads three ranks and the
    per-rank share (6 samples) ends in a batch of 2 behind a batch of 4."""
    def __init__(self, n=43):
        g = torch.Generator().manual_seed(77)
        self.x, self.y = torch.randn(n, 6, generator=g), torch.randn(n, 2, generator=g)

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], self.y[i]


def _toy_args(world, rank, tmp):
    return NS(device="cpu", world_size=world, process_index=rank, per_device_train_batch_size=4, dataloader_num_workers=0,
              dataloader_pin_memory=False, negatives_x_device=False, learning_rate=5e-2, weight_decay=0.01, adam_beta1=0.9,
              adam_beta2=0.999, adam_epsilon=1e-8, warmup_ratio=0.0, warmup_steps=1, max_steps=-1, num_train_epochs=2,
              gradient_accumulation_steps=1, max_grad_norm=1.0, logging_steps=2, save_steps=0, output_dir=tmp, fp16=False, bf16=False,
              seed=5)


def _toy_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))


def _loop_worker(rank, world, port, tmp, out_q):
    """DRTrainer.train -- the product's loop: sampler sharding, epochs, gradient averaging with the per-step pattern check over the
    gloo control group, clipping, AdamW, the linear schedule, logging -- around a stand-in for the HIP forward + backward."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openmatch_amd.trainer.dense_trainer import DRTrainer

    class Toy(DRTrainer):
        def training_step(self, model, batch):
            x, y = batch
            loss = ((model(x) - y) ** 2).mean()
            loss.backward()
            return loss.detach()

    model = _toy_model()
    t = Toy(model=model, args=_toy_args(world, rank, tmp), train_dataset=_ToyData())
    res = t.train()
    out_q.put((rank, res.global_step, [p.detach().numpy().copy() for p in model.parameters()], [h["step"] for h in t.state.log_history]))
    dist.barrier()
    dist.destroy_process_group()


def test_training_loop_eight_ranks_uneven_last_batch(tmp_path):
    """World size 8 over gloo, two epochs of 43 samples at 4 per device: every rank takes 2 steps per epoch (4 + 2 samples), ends on
    the same parameters, and those equal a single-process replay of the same schedule (mean over ranks of the per-rank batch-mean
    gradients -> clip -> AdamW -> linear decay)."""
    from openmatch_amd.trainer.dense_trainer import linear_schedule_factor, parameter_groups
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    [p.start() for p in procs]
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert all(g[1] == 4 and g[3] == [2, 4] for g in got), [(g[1], g[3]) for g in got]
    for g in got[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(g[2], got[0][2]))        # replicas stay identical
    # single-process replay
    ds, model = _ToyData(), _toy_model()
    a = _toy_args(1, 0, str(tmp_path))
    opt = torch.optim.AdamW(parameter_groups(model, a.weight_decay), lr=a.learning_rate, betas=(0.9, 0.999), eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: linear_schedule_factor(s, 1, 4))
    params = list(model.parameters())
    for epoch in range(2):
        shares = []
        for r in range(world):
            sm = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=r, seed=a.seed)
            sm.set_epoch(epoch)
            shares.append(list(sm))
        assert all(len(sh) == 6 for sh in shares)
        for lo, hi in ((0, 4), (4, 6)):
            total = [torch.zeros_like(p) for p in params]
            for sh in shares:
                idx = sh[lo:hi]
                model.zero_grad(set_to_none=True)
                ((model(ds.x[idx]) - ds.y[idx]) ** 2).mean().backward()
                for tsum, p in zip(total, params):
                    tsum += p.grad
            for tsum, p in zip(total, params):
                p.grad = tsum / world
            torch.nn.utils.clip_grad_norm_(params, a.max_grad_norm)
            opt.step(); sched.step()
    for mine, theirs in zip(params, got[0][2]):
        assert np.allclose(mine.detach().numpy(), theirs, atol=1e-6), np.abs(mine.detach().numpy() - theirs).max()


def _uneven_worker(rank, world, port, n_rows, n_q, k, out_q):
    """bench.py's multi-rank search branch on CPU: rows and queries split by bench.shard_range (uneven: the benchmark's
    8 841 823 rows leave 7 over 8 ranks, its 6 980 queries 4), query all-gather with padding, per-shard search, candidates
    exchanged by query range (all-to-all) and merged per slice."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from openmatch_amd.index import sharded_topk
    d = 32
    rng = np.random.default_rng(123)
    P = rng.standard_normal((n_rows, d)).astype(np.float32)
    Q = rng.standard_normal((n_q, d)).astype(np.float32)
    rows, offset = bench.shard_range(n_rows, world, rank)
    nq_local, q_off = bench.shard_range(n_q, world, rank)
    shard = _OracleShard(d)
    shard.add(P[offset:offset + rows])
    # the query all-gather of bench.search_once (padded to the largest share, then trimmed)
    nmax = (n_q + world - 1) // world
    pad = torch.zeros(nmax, d)
    pad[:nq_local] = torch.from_numpy(Q[q_off:q_off + nq_local])
    allq = torch.empty(world * nmax, d)
    dist.all_gather_into_tensor(allq, pad)
    sizes = [bench.shard_range(n_q, world, r)[0] for r in range(world)]
    queries = torch.cat([allq[r * nmax:r * nmax + sizes[r]] for r in range(world)])
    assert torch.equal(queries, torch.from_numpy(Q))
    Dm, Im, blk = sharded_topk(shard, queries, k, offset, merge=_merge_cpu)
    lo, hi = rank * blk, min((rank + 1) * blk, n_q)
    out_q.put((rank, rows, offset, lo, hi, Dm[:max(hi - lo, 0)].numpy(), Im[:max(hi - lo, 0)].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_uneven_shards_equal_single_index():
    """World size 8 with the benchmark's remainders in miniature: rows % 8 == 7, queries % 8 == 4 (8 841 823 and 6 980),
    and a k larger than the smallest shard's useful rows for some queries.  Union of the rank slices == one index."""
    from oracle import flatip
    import bench
    world, n_rows, n_q, k = 8, 8 * 53 + 7, 8 * 5 + 4, 60
    assert 8_841_823 % 8 == n_rows % 8 and 6_980 % 8 == n_q % 8
    assert sum(bench.shard_range(8_841_823, 8, r)[0] for r in range(8)) == 8_841_823
    assert [bench.shard_range(8_841_823, 8, r)[1] for r in range(1, 8)] == [sum(bench.shard_range(8_841_823, 8, q)[0] for q in range(r)) for r in range(1, 8)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, n_rows, n_q, k, q)) for r in range(world)]
    [p.start() for p in procs]
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert sorted(t[1] for t in got) == [53] * 1 + [54] * 7 and sum(t[1] for t in got) == n_rows
    rng = np.random.default_rng(123)
    P = rng.standard_normal((n_rows, 32)).astype(np.float32)
    Q = rng.standard_normal((n_q, 32)).astype(np.float32)
    single = flatip.IndexFlatIP(32); single.add(P)
    D, I = single.search(Q, k)
    covered = 0
    for rank, rows, offset, lo, hi, Dm, Im in got:
        if hi > lo:
            assert np.array_equal(Im, I[lo:hi]), rank
            assert np.allclose(Dm, D[lo:hi], atol=1e-6)
            covered += hi - lo
    assert covered == n_q
