#!/usr/bin/env python
"""Benchmark of the hot path on BASELINE.json's metric:
   passages/sec encode (bert-base, 128 tok -> 768-d) + queries/sec exact top-1000 over 8.8M x 768.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A *step* = one pass of the encoder hot path over one batch of synthetic MS-MARCO-shaped token
ids already resident in HBM: ids -> embeddings+LN -> 12 x (QKV GEMM, fused attention, out-proj
GEMM + residual, LN, FFN GEMMs + GELU, LN) -> CLS pooling -> 768-d embedding.  float16 MFMA with f32
accumulation -- the reference's own 16-bit format (`--fp16` = torch.cuda.amp float16, retriever/dense_retriever.py:76,151;
docs/dr-msmarco-passage.md:74) and the 16-bit mode whose dot products stay within `north_star`'s 1e-4 of the fp32 chain;
`--precision bf16` runs the two-plane bfloat16 path (reported as the `bf16` sub-object of the default line).
Random-init bert-base weights.
`value` = passages/s over all ranks (weak scaling: each rank encodes its own batches).
The search leg (same process, after the encode leg) serves Q queries against an 8 841 823 x 768
index sharded over the ranks: all-gather of query vectors -> per-shard filtered MFMA scan +
top-1000 -> candidates exchanged by query range (all-to-all) and merged per slice (ids identical to the exact f32 scan).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CORPUS_ROWS = 8_841_823          # MS MARCO passages
N_QUERIES = 6_980                # MS MARCO dev queries
GFLOP_PER_PASSAGE = 22.347       # 12*(24*L*H^2 + 4*L^2*H), L=128, H=768
GEMM_GFLOP_PER_PASSAGE = 12 * 24 * 128 * 768 ** 2 / 1e9
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
# rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md) of this command, committed per round and per format
TRAFFIC_FILES = {"f16": "r06_hbm_traffic_f16.json", "bf16": "r03_hbm_traffic.json"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="passages per step per GPU")
    ap.add_argument("--index-rows", type=int, default=CORPUS_ROWS, help="total index rows (all ranks)")
    ap.add_argument("--queries", type=int, default=N_QUERIES)
    ap.add_argument("--topk", type=int, default=1000)
    ap.add_argument("--no-search", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the strided-subsample parity leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the f32 and training sub-objects")
    ap.add_argument("--precision", default="f16", choices=["bf16", "f16", "f32"],
                    help="compute format of the headline encode leg (default float16: the reference's --fp16 format)")
    return ap.parse_args()


def synth_ids(batch, L, device, seed):
    """[CLS] body [SEP] pad with real length ~ U{16..L}, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    ids = torch.randint(1000, 30522, (batch, L), device=device, generator=g)
    lens = torch.randint(16, L + 1, (batch,), device=device, generator=g)
    pos = torch.arange(L, device=device)[None, :]
    mask = (pos < lens[:, None]).long()
    ids = ids * mask
    ids[:, 0] = 101
    ids.scatter_(1, (lens - 1)[:, None], 102)
    return ids.contiguous(), mask.contiguous()


def shard_range(total, world, rank):
    """Contiguous near-even split of `total` items over the ranks (the first total % world ranks take one more):
    (count, offset) of rank's share.  8 841 823 rows over 8 ranks: 7 shards of 1 105 228 and one of 1 105 227."""
    base, rem = divmod(total, world)
    return base + (1 if rank < rem else 0), rank * base + min(rank, rem)


def barrier_sync(world):
    if world > 1 or FORCE_DIST:
        dist.barrier()
    torch.cuda.synchronize()


# OM_BENCH_FORCE_DIST=1 with one process: the N > 1 code path (process group, query all-gather, candidate all-to-all, max over
# ranks, RCCL rank count) on a one-rank RCCL group -- all a one-GPU box can execute of it; the extras of the N = 1 line are skipped
FORCE_DIST = os.environ.get("OM_BENCH_FORCE_DIST", "0") == "1"


def max_over_ranks(x, world, device):
    if world == 1 and not FORCE_DIST:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline(no_search, sample_batches):
    """The reference's own CPU path, timed on this box's host cores on a bounded sample.

    What the reference runs per batch (retriever/dense_retriever.py:60-92): DRModelForInference(passage=batch) ->
    HF BertModel.forward -> CLS pooling -> `.cpu().numpy()`, then one faiss IndexFlatIP.search over the index
    (:166-192).  When /root/reference is importable (the build container) exactly those classes are timed, kind
    "reference".  On the GPU box the reference tree does not exist: the SAME loop is run over the SAME third-party
    module the reference calls (HF BertModel, installed in the image) with the same ragged masks as the GPU
    batches -- kind "port"; oracle.flatip stands in for faiss in both cases (faiss is not installable here)."""
    import sys as _sys
    from transformers import BertConfig, BertModel
    from oracle import flatip
    cores = os.cpu_count() or 1
    threads = min(cores, 64)                      # torch's CPU GEMMs stop scaling beyond ~64 threads
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    lm = BertModel(BertConfig()).eval()
    kind, model = "port", None
    ref_src = "/root/reference/src"
    if os.path.isdir(os.path.join(ref_src, "openmatch")):
        try:
            saved = list(_sys.path)
            mods = {k: v for k, v in _sys.modules.items() if k == "openmatch" or k.startswith("openmatch.")}
            for k in mods:
                del _sys.modules[k]
            _sys.path.insert(0, ref_src)
            import types as _types
            import datasets  # noqa: F401  (probes for a real faiss at import time: must precede the stub)
            if "faiss" not in _sys.modules:
                stub = _types.ModuleType("faiss"); stub.IndexFlatIP = flatip.IndexFlatIP; _sys.modules["faiss"] = stub
            from openmatch.modeling import DRModelForInference as RefModel      # the reference's class
            from types import SimpleNamespace as NS
            model = RefModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False))
            kind = "reference"
        except Exception:
            model = None
        finally:
            for k in [k for k in _sys.modules if k == "openmatch" or k.startswith("openmatch.")]:
                del _sys.modules[k]
            _sys.modules.update(mods)
            _sys.path[:] = saved

    def encode(batch):
        with torch.no_grad():
            if model is not None:
                return model(passage=batch).p_reps.cpu().numpy()
            return lm(**batch, return_dict=True).last_hidden_state[:, 0, :].cpu().numpy()     # DRModel.encode, pooling "first"

    ids = torch.cat([b["input_ids"].cpu() for b in sample_batches])
    msk = torch.cat([b["attention_mask"].cpu() for b in sample_batches])

    def run(n, bs=64):
        t0 = time.perf_counter()
        for s0 in range(0, n, bs):
            encode({"input_ids": ids[s0:s0 + bs], "attention_mask": msk[s0:s0 + bs]})
        return time.perf_counter() - t0
    run(64)                                        # warm-up (thread pool, allocator)
    probe = run(128)
    n = int(min(ids.shape[0], max(128, 64 * round(15.0 * 128 / max(probe, 1e-3) / 64))))    # ~15 s of CPU work
    t = run(n) if n > 128 else probe
    out = {"value": round(n / t, 2), "unit": "passages/s", "cores": threads, "host_cores": cores, "kind": kind,
           "sample": f"{n} passages x 128 tok (the GPU leg's ragged synthetic batches), bert-base fp32, batch 64, "
                     + ("reference DRModelForInference loop" if kind == "reference" else
                        "the reference's loop over HF BertModel (reference tree absent on this box)")
                     + f", torch {threads} threads"}
    if not no_search:
        d, rows, nq = 768, 1_000_000, 64
        rng = np.random.default_rng(0)
        P = rng.standard_normal((rows, d), dtype=np.float32)
        Q = rng.standard_normal((nq, d), dtype=np.float32)
        idx = flatip.IndexFlatIP(d)
        idx.add(P)
        t0 = time.perf_counter()
        idx.search(Q, 1000)
        ts = time.perf_counter() - t0
        out["search"] = {"value": round(nq / ts * rows / CORPUS_ROWS, 2), "unit": "queries/s",
                         "sample": f"{nq} queries x {rows} rows x 768 fp32 top-1000 (oracle/flatip.py in place of faiss.IndexFlatIP), "
                                   f"scaled linearly to {CORPUS_ROWS} rows"}
    return out


def parity_leg(model_16, lm, batches, device, index=None, queries=None, topk=1000, headline="f16"):
    """parity of THIS run's configuration on a strided subsample against the CPU oracle.
    Encode: every 16th passage of the first two timed batches through (a) the bf16 path as benchmarked (inside its
    1024-passage batch) and (b) the exact-f32 path, against oracle/encoder_ref.py in fp32.  Search: a strided
    65 536-row slice of the benchmark index x 256 queries, both scan modes against oracle/flatip.py with fp64
    adjudication of boundary ties."""
    from oracle import encoder_ref, flatip
    from openmatch.modeling import DRModelForInference
    from openmatch_amd.index import FlatIPIndex
    from types import SimpleNamespace as NS
    out = {}
    stride = 16
    sel = torch.arange(0, batches[0]["input_ids"].shape[0], stride)
    ids = torch.cat([b["input_ids"][sel] for b in batches[:2]]).cpu()
    msk = torch.cat([b["attention_mask"][sel] for b in batches[:2]]).cpu()
    sd = {k: v.detach().cpu() for k, v in lm.state_dict().items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        _, ref = encoder_ref.encode(sd, lm.config, "bert", {"input_ids": ids, "attention_mask": msk}, "first")
    ref = ref.double()
    other = "bfloat16" if headline == "f16" else "float16"
    got_head = torch.cat([model_16(passage=b).p_reps[sel] for b in batches[:2]]).double().cpu()
    m32 = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="float32")).to(device).eval()
    got32 = m32(passage={"input_ids": ids.to(device), "attention_mask": msk.to(device)}).p_reps.double().cpu()
    m_other = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype=other)).to(device).eval()
    got_other = torch.cat([m_other(passage=b).p_reps[sel] for b in batches[:2]]).double().cpu()
    del m_other
    got16, gotf16 = (got_other, got_head) if headline == "f16" else (got_head, got_other)
    cos = torch.nn.functional.cosine_similarity(got16, ref, dim=1)
    cosf = torch.nn.functional.cosine_similarity(gotf16, ref, dim=1)
    dots_ref = ref[:16] @ ref.t()
    out["encode"] = {
        "sample": f"{ids.shape[0]} passages (every {stride}th of two timed batches) vs oracle/encoder_ref.py fp32, {time.perf_counter() - t0:.1f} s of CPU",
        "bf16_min_cosine": round(float(cos.min()), 6), "bf16_mean_cosine": round(float(cos.mean()), 6),
        "bf16_max_abs_ddot": round(float(((got16[:16] @ got16.t()) - dots_ref).abs().max()), 4),
        "bf16_max_rel_ddot": float((((got16[:16] @ got16.t()) - dots_ref).abs() / dots_ref.abs().clamp_min(1.0)).max()),
        "f16_min_cosine": round(float(cosf.min()), 8), "f16_mean_cosine": round(float(cosf.mean()), 8),
        "f16_max_abs_ddot": round(float(((gotf16[:16] @ gotf16.t()) - dots_ref).abs().max()), 4),
        "f16_max_rel_ddot": float((((gotf16[:16] @ gotf16.t()) - dots_ref).abs() / dots_ref.abs().clamp_min(1.0)).max()),
        "f32_max_abs_emb_err": float((got32 - ref).abs().max()),
        "f32_max_rel_ddot": float((((got32[:16] @ got32.t()) - dots_ref).abs() / dots_ref.abs().clamp_min(1.0)).max()),
        "dot_scale": round(float(dots_ref.abs().max()), 1),
    }
    if index is not None:
        n = index.ntotal
        step = max(1, n // 65536)
        rows = index._f32[:n][::step][:65536, :index.d].contiguous()
        q = queries[:256].contiguous()
        P, Q = rows.cpu().numpy(), q.cpu().numpy()
        o = flatip.IndexFlatIP(index.d); o.add(P)
        Dr, Ir = o.search(Q, topk)
        P64, Q64 = torch.from_numpy(P).double(), torch.from_numpy(Q).double()

        def full(qi, disputed):
            sc = P64 @ Q64[qi]
            return sc[torch.tensor(disputed)].numpy(), torch.topk(sc, topk).values[-1].item()
        res = {"sample": f"{rows.shape[0]} index rows (stride {step}) x {q.shape[0]} queries, top-{topk}, vs oracle/flatip.py"}
        for precision in ("f16_rescore", "f32"):
            sub = FlatIPIndex(index.d, device=device, precision=precision)
            sub.add(rows)
            D, I = sub.search_device(q, topk)
            n_exact, n_tie, n_bad, _ = flatip.topk_sets_equal(I.cpu().numpy(), Ir, full, rel_tol=2e-6)
            res[precision] = {"id_sets_identical": n_exact, "fp64_near_tie_only": n_tie, "wrong": n_bad,
                              "max_abs_score_err": float(np.abs(D.cpu().numpy() - Dr).max())}
        out["search"] = res
        # the benchmark's OWN index and k: om_sim_topk over all rows for 256 queries against an independent device-side
        # check (chunked torch fp32 matmul + topk, fp64 adjudication of differing sets: oracle/device_check.py)
        from oracle import device_check
        t0 = time.perf_counter()
        qf = queries[:256].contiguous()
        rows_all = index._f32[:n, :index.d]
        D, I = index.search_device(qf, topk)
        info = dict(index.last_search_info)
        Dr, Ir = device_check.reference_topk(rows_all, qf, topk)
        n_exact, n_tie, n_bad, detail = device_check.adjudicate(rows_all, qf, I, Ir, topk)
        torch.cuda.synchronize()
        out["search_full"] = {"rows": int(n), "queries": int(qf.shape[0]), "k": int(topk), "precision": index.precision,
                              "id_sets_identical": n_exact, "fp64_near_tie_only": n_tie, "wrong": n_bad,
                              "max_abs_score_err": float((D - Dr).abs().max()), "scan_info": info,
                              "checker": "chunked torch.matmul fp32 + torch.topk on the device, fp64 adjudication of differing id sets "
                                         "(oracle/device_check.py); %.1f s" % (time.perf_counter() - t0)}
    return out


def train_leg(device, steps=8):
    """BASELINE config 3 per GPU: one contrastive training step (8 queries x 32 tok + 64 passages x 128 tok, bert-base, dropout 0.1,
    forward + backward + clip + AdamW) through the product's own loop (DRTrainer.training_step + DRTrainer.optimizer_step).
    `value` is the reference's documented mode: `--fp16` (docs/dr-msmarco-passage.md:74) = float16 kernels + dynamic loss scale, with
    the forward's residual stream in f32 as autocast keeps it.  Beside it: the same step in bfloat16, and in bfloat16 with the 16-bit
    residual stream of rounds 1-4 (OM_TRAIN_RES32 = 0) for continuity with earlier rounds' numbers."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch.trainer import DRTrainer
    from openmatch_amd import native as N
    from types import SimpleNamespace as NS
    g = torch.Generator().manual_seed(1)
    mk = lambda n, L: {"input_ids": torch.randint(1000, 30000, (n, L), generator=g), "attention_mask": torch.ones(n, L, dtype=torch.long)}
    # the batch is resident in HBM before the timed region (the bench contract; a training run gets there through the DataLoader's
    # pinned buffers and non-blocking copies -- a pageable host tensor would make every step's copy a stream synchronisation)
    batch = tuple({k: v.to(device) for k, v in b.items()} for b in (mk(8, 32), mk(64, 128)))
    flop = 3 * (8 * 5.474e9 + 64 * GFLOP_PER_PASSAGE * 1e9)

    # the same step on RAGGED lengths (queries U{4..32}, passages U{16..128} tokens, right-padded: what a collator hands over) --
    # padded pair vs the packed-rows pair of round 5 (om_encoder_train_forward_packed: the step runs over the tokens, not over B x L)
    from openmatch_amd.encoder import TOKEN_ROWS_KEY, token_rows_of
    from openmatch_amd import train as T
    gl = torch.Generator().manual_seed(3)
    def ragged_of(b, lo):
        L = b["input_ids"].shape[1]
        lens = torch.randint(lo, L + 1, (b["input_ids"].shape[0],), generator=gl)
        m = (torch.arange(L)[None, :] < lens[:, None]).long()
        return {"input_ids": b["input_ids"], "attention_mask": m.to(device)}, token_rows_of(m)
    ragged = [ragged_of(batch[0], 4), ragged_of(batch[1], 16)]

    def run(fp16, dtype, res32, rag=None):
        torch.manual_seed(0)
        lm = BertModel(BertConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1))
        model = DRModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype=dtype),
                        data_args=NS(train_n_passages=8),
                        train_args=NS(negatives_x_device=False, per_device_train_batch_size=8)).to(device)
        args = NS(device=device, world_size=1, process_index=0, per_device_train_batch_size=8, negatives_x_device=False,
                  learning_rate=5e-6, weight_decay=0.0, max_grad_norm=1.0, gradient_accumulation_steps=1, fp16=fp16, bf16=False)
        trainer = DRTrainer(model=model, args=args)
        # HF Trainer's defaults of the reference's command line (docs/dr-msmarco-passage.md:62-80): AdamW, clip_grad_norm_(1.0) -- here
        # openmatch_amd.optim.FusedAdamW: clip + AdamW + the refresh of the packed 16-bit weights in one pass
        trainer.create_optimizer_and_scheduler(num_training_steps=10 ** 6)
        N.check(N.lib().om_debug_option(18, int(res32)))              # OM_OPT_TRAIN_RES32
        use = batch
        if rag is not None:         # "packed": the token counts ride along, as DRTrainer._prepare_inputs notes them for a host-side batch
            use = tuple(dict(b, **({TOKEN_ROWS_KEY: n} if rag == "packed" else {})) for b, n in ragged)
        try:
            def step():
                loss = trainer.training_step(model, use)
                trainer.optimizer_step()
                return loss
            for _ in range(3):
                step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                loss = step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        finally:
            N.check(N.lib().om_debug_option(18, 1))
        out = {"value": round(1 / dt, 2), "unit": "steps/s", "ms_per_step": round(dt * 1e3, 2),
               "algorithmic_tflops": round(flop / dt / 1e12, 1), "frac_of_mfma_peak": round(flop / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
               "loss": float(loss), "optimizer": type(trainer.optimizer).__name__}
        sc = trainer._loss_scaler()
        if sc is not None:
            out["loss_scale"] = float(sc.state[0]); out["skipped_steps"] = sc.skipped_steps()
        if rag is not None:
            out["rows"] = dict(T.LAST_CALL); out["tokens"] = [int(n.sum()) for _, n in ragged]
            out.pop("algorithmic_tflops"); out.pop("frac_of_mfma_peak")          # (priced for full-length rows)
        del model, trainer
        torch.cuda.empty_cache()
        return out

    out = {"metric": "contrastive training steps/s per GPU (8 q x 32 tok + 64 p x 128 tok, bert-base, fwd + bwd + clip + AdamW, dropout 0.1)",
           "dtype": "f16", "mode": "--fp16: float16 kernels + dynamic loss scale (the reference's documented training mode), f32 residual stream"}
    out.update(run(True, "float16", 1))
    out["bf16"] = dict(run(False, "bfloat16", 1), mode="bfloat16 kernels, f32 residual stream")
    out["bf16_res16"] = dict(run(False, "bfloat16", 0), mode="bfloat16 kernels, 16-bit residual stream (OM_TRAIN_RES32=0): the data flow of rounds 1-4")
    out["ragged"] = {"note": "the --fp16 step on ragged lengths (queries U{4..32}, passages U{16..128} tokens): the padded pair computes over B x L rows as the "
                             "reference does, the packed pair (round 5) over the tokens up to each sequence's last unmasked one -- same gradients (tests), and since "
                             "round 6 the same dropout masks (keyed on the token, not on the packed row): the two `loss` values below agree to 16-bit noise",
                     "padded": run(True, "float16", 1, "padded"), "packed": run(True, "float16", 1, "packed")}
    out["ragged"]["speedup"] = round(out["ragged"]["packed"]["value"] / out["ragged"]["padded"]["value"], 3)
    return out


def few_rows_leg(model, device):
    """Latency of SMALL forwards through the product path (a served query: 1 x 32 tokens; a few sequences), ids resident in HBM: the
    few-rows path of round 5 (gemm_skinny.hip: weight-streaming contractions, <= OM_OPT_GEMM_SKINNY_M token rows) against the tile
    kernels (the path switched off).  Roofline of such a forward: one pass over the 16-bit weights (170 MB for bert-base)."""
    from openmatch_amd import native as N
    out = {"metric": "ms per forward (bert-base, representations out, ids resident in HBM)", "weights_MB": 170.3, "rows": {}}
    for (B, L) in ((1, 32), (8, 32), (16, 32), (1, 128), (8, 128)):
        ids = torch.randint(1000, 30000, (B, L), device=device, generator=torch.Generator(device=device).manual_seed(B * L))
        items = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
        row = {}
        for name, lim in (("few_rows_path", 1024), ("tile_kernels", 0)):
            N.check(N.lib().om_debug_option(19, lim))           # OM_OPT_GEMM_SKINNY_M
            try:
                for _ in range(5):
                    model(query=items)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(30):
                    model(query=items)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
            finally:
                N.check(N.lib().om_debug_option(19, 1024))
            row[name + "_ms"] = round(dt * 1e3, 3)
        row["weight_stream_TBps"] = round(170.3e6 / (row["few_rows_path_ms"] * 1e-3) / 1e12, 3)
        out["rows"]["%dx%d" % (B, L)] = row
    return out


def long_passages_leg(model, device):
    """Encode at the lengths document corpora use (round 6): 256 x 512-token passages per batch -- the same 131 072 tokens as the headline
    batch -- full length and ragged through the packed-rows entry.  Beyond 256 tokens attention runs on attention_fwd16c_kernel (the fast
    kernel's body in an online-softmax loop over 128-key chunks); the contractions are the headline's."""
    from openmatch_amd import encoder as enc_mod
    from openmatch_amd.encoder import TOKEN_ROWS_KEY, token_rows_of
    B, L = 256, 512
    g = torch.Generator(device=device).manual_seed(512)
    ids = torch.randint(1000, 30000, (B, L), device=device, generator=g)
    full = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    lens = torch.randint(64, L + 1, (B,), generator=torch.Generator().manual_seed(7))
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    ragged = {"input_ids": ids, "attention_mask": mask.to(device)}
    ragged_packed = dict(ragged); ragged_packed[TOKEN_ROWS_KEY] = token_rows_of(mask)      # the token counts a collator's host-side mask gives
    out = {"metric": "passages/s encode at 512 tokens per passage (bert-base, float16, 256 passages per batch)", "tokens_per_batch": B * L}
    for name, items in (("full_length", full), ("ragged_padded", ragged), ("ragged_packed", ragged_packed)):
        for _ in range(2):
            model(passage=items)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            model(passage=items)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        out[name] = {"passages_per_s": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 2), "rows": dict(enc_mod.LAST_CALL)}
    out["full_length"]["tokens_per_s"] = round(B * L / (out["full_length"]["ms_per_batch"] * 1e-3))
    out["ragged_tokens"] = int(lens.sum())
    return out


def packed_leg(model, batches, a, L):
    """The SAME timed batches through om_encoder_forward_packed: only the rows up to each sequence's last token (lengths
    ~ U{16..128}) enter the embedding, the contractions and the normalisations; attention runs per sequence.  The
    reference pads to 128 and computes over the padding (dataset/data_collator.py:27-38), so the headline `value` stays
    the padded computation; this is what the product's own collator path (feed.py: host-side lengths) runs.  The row bound
    is computed from the lengths on the host BEFORE the timed region (in production it comes with the collator's batch)."""
    from openmatch_amd import encoder as E
    from openmatch_amd.encoder import compute_dtype_code, hip_encode, packed_rows_bound
    code = compute_dtype_code(model.model_args)
    bounds = [packed_rows_bound(b["attention_mask"]) for b in batches]
    tokens = [int(b["attention_mask"].sum()) for b in batches]

    def run(i, packed):
        b = batches[i % len(batches)]
        return hip_encode(model.lm_p, b, "first", None, False, code, want_hidden=False,
                          packed_rows=bounds[i % len(batches)] if packed else None)[1]
    same, took = True, None
    for i in range(len(batches)):
        got = run(i, True)
        took = dict(E.LAST_CALL)
        same = same and bool(torch.equal(got, run(i, False)))
    res = {}
    for packed in (False, True, False, True):          # interleaved: padded, packed, padded, packed
        for i in range(a.warmup):
            run(i, packed)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(a.steps):
            run(i, packed)
        torch.cuda.synchronize()
        res.setdefault(packed, []).append((time.perf_counter() - t0) / a.steps)
    dt_p, dt_d = min(res[True]), min(res[False])
    # the packed leg's own roofline: EXECUTED flops of its GEMM launches (2 M N K with M = the packed rows) over their hipEvent time
    from openmatch_amd import native as N
    lib = N.lib()
    lib.om_kernel_timing_read(0, None, None, None)
    lib.om_kernel_timing_enable(1)
    for i in range(a.steps):
        run(i, True)
    torch.cuda.synchronize()
    ms, n_launch, flops = C.c_double(), C.c_int64(), C.c_double()
    N.check(lib.om_kernel_timing_read(0, C.byref(ms), C.byref(n_launch), C.byref(flops)))
    lib.om_kernel_timing_enable(0)
    tf = flops.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    roof = {"kernel": "the same GEMM kernels over the packed rows", "bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "launches": int(n_launch.value),
            "flops_per_launch": flops.value / max(n_launch.value, 1), "avg_launch_us": round(ms.value * 1e3 / max(n_launch.value, 1), 2),
            "measured": "hipEvents around every GEMM launch of a.steps packed steps; executed flops (M = packed rows), not the padded shape's"}
    return {"metric": "passages/s encode over packed rows (om_encoder_forward_packed; same batches, same representations)",
            "roofline": roof,
            "value": round(a.batch / dt_p, 1), "unit": "passages/s", "ms_per_step": round(dt_p * 1e3, 3),
            "padded_same_loop_ms_per_step": round(dt_d * 1e3, 3), "speedup_vs_padded": round(dt_d / dt_p, 3),
            "rows_per_step": bounds, "tokens_per_step": tokens, "padded_rows_per_step": a.batch * L,
            "row_fraction": round(sum(bounds) / (len(bounds) * a.batch * L), 4),
            "identical_to_padded": same, "last_call": took,
            "algorithmic_tflops_on_packed_rows": round(sum(bounds) / len(bounds) / L * GFLOP_PER_PASSAGE / 1e3 / dt_p, 1)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or FORCE_DIST
    json_out = sys.stdout
    if dist_on:
        # RCCL prints a version banner to the C-level stdout (buffered: it lands BEHIND the JSON line at exit).  The one line on
        # stdout must be the JSON: file descriptor 1 is pointed at stderr for everything else, the line goes to the saved one.
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        json_out = os.fdopen(saved_fd, "w")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29553"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    rccl_ranks = None
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    from openmatch_amd import native as N
    from openmatch_amd.index import FlatIPIndex, sharded_topk
    from types import SimpleNamespace as NS

    lib = N.lib()
    if dist_on:      # ranks as RCCL itself counts them (ncclCommCount on a communicator built from this rendezvous)
        try:           # (a diagnostic: it must not cost the run its measurements)
            from openmatch_amd.comm import RcclComm
            comm = RcclComm.from_torch_distributed(device)
            rccl_ranks = comm.count()
            comm.close()
        except Exception as e:      # noqa: BLE001
            rccl_ranks = "unavailable: %s" % (str(e)[:120],)
    torch.manual_seed(0)
    lm = BertModel(BertConfig()).eval()
    half = a.precision in ("bf16", "f16")
    dtype_name = {"bf16": "bfloat16", "f16": "float16"}.get(a.precision, "float32")
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first",
                                model_args=NS(encoder_only=False, dtype=dtype_name)).to(device).eval()
    L = 128
    batches = [synth_ids(a.batch, L, device, 1000 * rank + i) for i in range(4)]
    batches = [{"input_ids": i, "attention_mask": m} for i, m in batches]

    def step(i):
        return model(passage=batches[i % len(batches)]).p_reps

    # ---------------- encode leg: W warm-up, K timed steps -------------------------------
    # Time-based pre-warm ahead of the counted warm-up: the first pass of a process carries the
    # one-off work (weight packing, folded-LayerNorm cache, workspace allocation) and a fresh box may still be raising
    # its clocks -- encode until three consecutive step times agree within 2 % (at least 0.3 s, capped at 3 s).
    # profiles/r03_cold_step_series.json: on a fresh box only the first pass differs (40 ms, then 24.1-24.3 ms).
    prewarm_ms = []
    t_pw = time.perf_counter()
    while True:
        torch.cuda.synchronize(); t1 = time.perf_counter()
        step(len(prewarm_ms))
        torch.cuda.synchronize()
        prewarm_ms.append((time.perf_counter() - t1) * 1e3)
        spent = time.perf_counter() - t_pw
        last = prewarm_ms[-3:]
        if spent >= 3.0 or (spent >= 0.3 and len(last) == 3 and max(last) <= 1.02 * min(last)):
            break
    for i in range(a.warmup):
        step(i)
    barrier_sync(world)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    barrier_sync(world)
    t_enc = max_over_ranks(time.perf_counter() - t0, world, device)
    passages_per_s = world * a.batch * a.steps / t_enc

    # second pass of the same K steps with every GEMM launch bracketed by HIP events
    lib.om_kernel_timing_enable(1)
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    ms, n_launch, flops = C.c_double(), C.c_int64(), C.c_double()
    cls = 0 if half else 1
    N.check(lib.om_kernel_timing_read(cls, C.byref(ms), C.byref(n_launch), C.byref(flops)))
    lib.om_kernel_timing_enable(0)
    gemm_tflops = flops.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    peak = PEAK_BF16_TFLOPS if half else 157.3
    traffic, tsrc = None, None   # HBM bytes per launch of the dominant kernel: the round's committed rocprofv3 --pmc passes of this command
    tfile = os.path.join(REPO, "profiles", TRAFFIC_FILES.get(a.precision, ""))
    if half and a.batch == 1024 and os.path.isfile(tfile):      # named per precision and round: a stale file is never picked up by sort order
        try:
            tj = json.load(open(tfile))
            traffic, tsrc = tj["hbm_bytes_per_launch"], "profiles/" + os.path.basename(tfile) + ": " + tj.get("note", "")
        except Exception:
            traffic = None
    roofline = {
        "kernel": ("gemm_nt_kernel7c16 / gemm_nt_kernel7r16<%s>" % a.precision) if half else "gemm_nt_kernel6<f32>",
        "kernel_note": (("persistent 256x256 tiles on a continuous five-unit LDS ring, 128-byte K steps, 16x16x32 MFMAs; the encoder's QKV / "
                         "out-proj / FFN contractions with plain / GELU / LayerNorm-residual epilogues%s"
                         % ("; bf16: the two-plane residual epilogue runs on gemm_nt_kernel7" if a.precision == "bf16" else ""))
                        if half else "exact-f32 MFMA (32x32x2), k-ordered fmaf chain"),
        "bound": "mfma", "achieved": round(gemm_tflops, 1), "peak": peak, "unit": "TFLOP/s",
        "frac": round(gemm_tflops / peak, 4), "traffic": traffic,
        "traffic_source": tsrc,
        "launches": int(n_launch.value), "avg_launch_us": round(ms.value * 1e3 / max(n_launch.value, 1), 2),
        "flops_per_launch": flops.value / max(n_launch.value, 1),
        "measured": "hipEvents around every launch of the kernel on its stream, second pass of the same K steps",
        "end_to_end_frac_of_peak": round(passages_per_s / world * GFLOP_PER_PASSAGE / 1e3 / peak, 4),
    }

    # ---------------- search leg -----------------------------------------------------------
    search, parity = None, None
    if not a.no_search:
        rows, offset = shard_range(a.index_rows, world, rank)
        index = FlatIPIndex(768, device=device, precision="f16_rescore" if half else "f32")
        g = torch.Generator(device=device).manual_seed(77 + rank)
        shared = torch.randn(1, 768, device=device, generator=torch.Generator(device=device).manual_seed(5))
        index._reserve(rows)
        for s in range(0, rows, 1 << 20):                         # anisotropic, CLS-like: mean + noise
            n = min(1 << 20, rows - s)
            index.add(torch.randn(n, 768, device=device, generator=g) * 0.05 + shared * 0.05)
        nq_local, _ = shard_range(a.queries, world, rank)
        q_local = torch.randn(nq_local, 768, device=device, generator=g) * 0.05 + shared * 0.05

        def search_once():
            if dist_on:
                nmax = (a.queries + world - 1) // world
                pad = torch.zeros(nmax, 768, device=device)
                pad[:nq_local] = q_local
                allq = torch.empty(world * nmax, 768, device=device)
                dist.all_gather_into_tensor(allq, pad)
                sizes = [shard_range(a.queries, world, r)[0] for r in range(world)]
                queries = torch.cat([allq[r * nmax:r * nmax + sizes[r]] for r in range(world)])
            else:
                queries = q_local
            if dist_on:      # candidates exchanged by query range (all-to-all), merged per slice: openmatch_amd/index.py
                D, I, _ = sharded_topk(index, queries, a.topk, offset)
            else:
                D, I = index.search_device(queries, a.topk, id_offset=offset)
            return D, I

        search_once()                                            # warm-up
        barrier_sync(world)
        reps = 2
        lib.om_kernel_timing_enable(1)
        t0 = time.perf_counter()
        for _ in range(reps):
            D, I = search_once()
        barrier_sync(world)
        t_s = max_over_ranks(time.perf_counter() - t0, world, device) / reps
        sms, sl, sf = C.c_double(), C.c_int64(), C.c_double()
        N.check(lib.om_kernel_timing_read(2, C.byref(sms), C.byref(sl), C.byref(sf)))
        lib.om_kernel_timing_read(0, None, None, None)
        lib.om_kernel_timing_read(1, None, None, None)
        lib.om_kernel_timing_enable(0)
        es = 2 if half else 4
        scan_bytes = rows * 768 * es
        search = {
            "metric": "queries/sec exact top-%d over %d x 768" % (a.topk, a.index_rows),
            "value": round(a.queries / t_s, 1), "unit": "queries/s", "queries": a.queries,
            "seconds_per_batch": round(t_s, 4),
            "scaling": "strong (index rows fixed, sharded by rank)" if world > 1 else "single shard",
            "precision": ("f16 MFMA candidate scan (certified margin) + exact f32 re-score; ids == f32 scan" if half else "exact f32 MFMA scan"),
            "scan_info": index.last_search_info,
            "algorithmic_tflops": round(2.0 * a.index_rows * 768 * a.queries / t_s / 1e12, 1),
            "frac_of_mfma_peak": round(2.0 * a.index_rows * 768 * a.queries / t_s / 1e12 / (peak * world), 4),
            "scan_kernel": {"tflops": round(sf.value / max(sms.value, 1e-9) / 1e9, 1),
                            "ms_per_search": round(sms.value / reps, 2), "launches_per_search": int(sl.value // reps),
                            "index_stream_GBps_if_read_once": round(scan_bytes / max(sms.value / reps, 1e-9) / 1e6, 1)},
        }
        if not dist_on and not a.no_extra:
            # small batches: the scan is one pass over the f16 index -- latency and the implied HBM stream rate
            small = {}
            for nq_s in (1, 64, 128):
                qs = q_local[:nq_s]
                index.search_device(qs, a.topk)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(3):
                    index.search_device(qs, a.topk)
                torch.cuda.synchronize(); dt_s = (time.perf_counter() - t1) / 3
                small["q%d" % nq_s] = {"ms": round(dt_s * 1e3, 2), "index_stream_TBps": round(scan_bytes / dt_s / 1e12, 2),
                                       "frac_of_hbm_peak": round(scan_bytes / dt_s / 1e9 / PEAK_HBM_GBS, 3)}
            search["small_batch_latency"] = small
        if rank == 0 and not dist_on and not a.no_parity and half:
            parity = parity_leg(model, lm, batches, device, index, q_local, a.topk, headline=a.precision)
        del index
        torch.cuda.empty_cache()
    if parity is None and rank == 0 and not dist_on and not a.no_parity and half:
        parity = parity_leg(model, lm, batches, device, headline=a.precision)

    # ---------------- exact-f32 mode and the training step (sub-objects; N = 1 only) ----------
    f32_mode, train, f16_mode, packed_mode, few_rows, long_passages = None, None, None, None, None, None
    other16 = "bf16" if a.precision == "f16" else "f16"
    if rank == 0 and not dist_on and not a.no_extra and half:
        # the OTHER 16-bit format on the SAME timed batches (same kernels and MFMA rate): bfloat16 carries its pre-LayerNorm
        # residual stream in two planes to stay inside the reference's 16-bit envelope, float16 needs one
        m16 = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first",
                                  model_args=NS(encoder_only=False, dtype={"bf16": "bfloat16", "f16": "float16"}[other16])).to(device).eval()
        for i in range(a.warmup):
            m16(passage=batches[i % len(batches)])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(a.steps):
            m16(passage=batches[i % len(batches)])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
        f16_mode = {"metric": ("passages/s encode in the bfloat16 MFMA mode (two-plane residual stream; parity in `parity.encode.bf16_*`)"
                               if other16 == "bf16" else
                               "passages/s encode in the float16 MFMA mode (the reference's --fp16 format; parity in `parity.encode.f16_*`)"),
                    "value": round(a.batch / dt, 1), "unit": "passages/s", "ms_per_step": round(dt * 1e3, 3),
                    "algorithmic_tflops": round(a.batch * GFLOP_PER_PASSAGE / 1e3 / dt, 1),
                    "frac_of_mfma_peak": round(a.batch * GFLOP_PER_PASSAGE / 1e3 / dt / PEAK_BF16_TFLOPS, 4)}
        del m16
        torch.cuda.empty_cache()
        packed_mode = packed_leg(model, batches, a, L)
        few_rows = few_rows_leg(model, device)
        long_passages = long_passages_leg(model, device)
        m32 = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first",
                                  model_args=NS(encoder_only=False, dtype="float32")).to(device).eval()
        sub = {k: v[:256] for k, v in batches[0].items()}
        for _ in range(2):
            m32(passage=sub)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            m32(passage=sub)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        f32_mode = {"metric": "passages/s encode in the exact-f32 MFMA mode (the mode that meets the 1e-4 parity bar)",
                    "value": round(256 / dt, 1), "unit": "passages/s", "passages_per_step": 256,
                    "algorithmic_tflops": round(256 * GFLOP_PER_PASSAGE / 1e3 / dt, 1),
                    "frac_of_f32_mfma_peak": round(256 * GFLOP_PER_PASSAGE / 1e3 / dt / 157.3, 4)}
        del m32
        torch.cuda.empty_cache()
        train = train_leg(device)

    cpu = None
    if rank == 0 and not dist_on and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.no_search, batches)

    if rank == 0:
        line = {
            "metric": "passages/sec encode (bert-base DPR bi-encoder, 128 tok -> 768-d) "
                      "[+ queries/sec exact top-1000 over 8.8M x 768 in `search`]",
            "value": round(passages_per_s, 1), "unit": "passages/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(t_enc / a.steps * 1e3, 3),
            "cold_first_pass_ms": round(prewarm_ms[0], 2),
            "prewarm": {"passes": len(prewarm_ms), "seconds": round(sum(prewarm_ms) / 1e3, 3), "last_ms": round(prewarm_ms[-1], 3),
                        "rule": "untimed passes until 3 consecutive agree within 2 % (>= 0.3 s, <= 3 s), before the W counted warm-up steps"},
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "bert-base DPR bi-encoder encode+search, MS MARCO 8.8M x 128-tok -> 768-d (BASELINE configs[1])",
                       "passages_per_step_per_gpu": a.batch, "seq_len": L, "global_batch": a.batch * world,
                       "index_rows": a.index_rows, "queries": a.queries, "topk": a.topk,
                       "weights": "random-init BertConfig() seed 0", "parallelism": f"shard{world}"},
            # what every rank did (round 5: so that the first real 8-GPU run needs no edits): encode is WEAK scaling (each rank its own
            # `passages_per_step_per_gpu` batches, no collective), search is STRONG scaling (the 8.84 M rows split by rank, queries
            # all-gathered, candidates exchanged by query range with one all-to-all + per-slice merge)
            "ranks": {"world": world, "rccl_ranks": rccl_ranks,
                      "encode": {"scaling": "weak", "passages_per_step": [a.batch] * world},
                      "search": (None if a.no_search else
                                 {"scaling": "strong" if world > 1 else "single shard",
                                  "index_rows": [shard_range(a.index_rows, world, r)[0] for r in range(world)],
                                  "query_slices": [shard_range(a.queries, world, r)[0] for r in range(world)]})},
            "roofline": roofline, "search": search, "parity": parity, other16: f16_mode, "packed": packed_mode, "few_rows": few_rows, "long_passages": long_passages, "f32": f32_mode, "train": train, "cpu_baseline": cpu,
        }
        print(json.dumps(line), file=json_out, flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
