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
    assert gathered.shape == (4, 8) and torch.equal(gathered[rank * 2:(rank + 1) * 2], t)
    gathered.sum().backward()
    assert torch.equal(t.grad, torch.ones_like(t))                          # gradient only through the local slot
    parts = [torch.randn(2, 8, generator=torch.Generator().manual_seed(r)) for r in range(world)]
    assert torch.allclose(gathered.detach(), torch.cat(parts))              # rank-major order
    # gradient averaging == DDP mean
    p = torch.nn.Parameter(torch.zeros(3)); p.grad = torch.full((3,), float(rank + 1))
    allreduce_mean_([p], world)
    assert torch.allclose(p.grad, torch.full((3,), 1.5))
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
        assert torch.allclose(q_.grad, torch.full_like(q_.grad, 1.5 * (i + 1)))
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
    assert grad_sync.active() is None and torch.equal(arena2, single)
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


def test_cross_device_negatives_and_gradient_averaging():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got[0] == pytest.approx(got[1], abs=1e-6)                         # identical loss on every rank


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
