#!/usr/bin/env python
"""Benchmark of the hot path on BASELINE.json's metric:
   passages/sec encode (bert-base, 128 tok -> 768-d) + queries/sec exact top-1000 over 8.8M x 768.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A *step* = one pass of the encoder hot path over one batch of synthetic MS-MARCO-shaped token
ids already resident in HBM: ids -> embeddings+LN -> 12 x (QKV GEMM, fused attention, out-proj
GEMM + residual, LN, FFN GEMMs + GELU, LN) -> CLS pooling -> 768-d embedding.  bf16 MFMA with f32
accumulation (the reference's documented `--fp16` mode), random-init bert-base weights.
`value` = passages/s over all ranks (weak scaling: each rank encodes its own batches).
The search leg (same process, after the encode leg) serves Q queries against an 8 841 823 x 768
index sharded over the ranks: all-gather of query vectors -> per-shard filtered MFMA scan +
top-1000 -> gather + merge on rank 0 (ids identical to the exact f32 scan).
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
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
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


def barrier_sync(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, device):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline(no_search):
    """The CPU oracle (a port of the reference's path: plain torch f32 restatement of HF BertModel +
    DRModel.encode, oracle/encoder_ref.py) timed on this box's host cores on a bounded sample."""
    from transformers import BertConfig, BertModel
    from oracle import encoder_ref, flatip
    cores = min(os.cpu_count() or 1, 64)     # beyond ~64 threads torch's CPU GEMMs stop scaling
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    cfg = BertConfig()
    lm = BertModel(cfg).eval()
    sd = lm.state_dict()
    rng = np.random.default_rng(0)

    def run(n, bs=16):
        ids = torch.from_numpy(rng.integers(1000, 30522, size=(n, 128)))
        items = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
        t0 = time.perf_counter()
        with torch.no_grad():
            for s in range(0, n, bs):
                encoder_ref.encode(sd, cfg, "bert", {k: v[s:s + bs] for k, v in items.items()}, "first")
        return time.perf_counter() - t0
    run(16)                                        # warm-up (thread pool, allocator)
    probe = run(32)
    n = int(min(4096, max(32, 16 * round(12.0 * 32 / max(probe, 1e-3) / 16))))    # ~12 s of CPU work
    t = run(n) if n > 32 else probe
    out = {"value": n / t, "unit": "passages/s", "cores": cores, "kind": "port",
           "sample": f"{n} passages x 128 tok, bert-base f32, oracle/encoder_ref.py, torch {cores} threads"}
    if not no_search:
        d, rows, nq = 768, 500_000, 64
        P = rng.standard_normal((rows, d), dtype=np.float32)
        Q = rng.standard_normal((nq, d), dtype=np.float32)
        idx = flatip.IndexFlatIP(d)
        idx.add(P)
        t0 = time.perf_counter()
        idx.search(Q, 1000)
        ts = time.perf_counter() - t0
        out["search"] = {"value": nq / ts * rows / CORPUS_ROWS, "unit": "queries/s",
                         "sample": f"{nq} queries x {rows} rows x 768 f32 top-1000 (oracle/flatip.py), "
                                   f"scaled linearly to {CORPUS_ROWS} rows"}
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    from openmatch_amd import native as N
    from openmatch_amd.index import FlatIPIndex, merge_topk
    from types import SimpleNamespace as NS

    lib = N.lib()
    torch.manual_seed(0)
    lm = BertModel(BertConfig()).eval()
    dtype_name = "bfloat16" if a.precision == "bf16" else "float32"
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first",
                                model_args=NS(encoder_only=False, dtype=dtype_name)).to(device).eval()
    L = 128
    batches = [synth_ids(a.batch, L, device, 1000 * rank + i) for i in range(4)]
    batches = [{"input_ids": i, "attention_mask": m} for i, m in batches]

    def step(i):
        return model(passage=batches[i % len(batches)]).p_reps

    # ---------------- encode leg: W warm-up, K timed steps -------------------------------
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
    cls = 0 if a.precision == "bf16" else 1
    N.check(lib.om_kernel_timing_read(cls, C.byref(ms), C.byref(n_launch), C.byref(flops)))
    lib.om_kernel_timing_enable(0)
    gemm_tflops = flops.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    peak = PEAK_BF16_TFLOPS if a.precision == "bf16" else 157.3
    traffic = None        # HBM bytes per launch of the dominant kernel, from the committed PMC passes
    tpath = os.path.join(REPO, "profiles", "r01_hbm_traffic.json")
    if a.precision == "bf16" and a.batch == 1024 and os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
    roofline = {
        "kernel": "gemm_nt_kernel6<%s> (encoder QKV / out-proj / FFN contractions; three epilogue variants)" % a.precision,
        "bound": "mfma", "achieved": round(gemm_tflops, 1), "peak": peak, "unit": "TFLOP/s",
        "frac": round(gemm_tflops / peak, 4), "traffic": traffic,
        "traffic_source": "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH doubled per the gfx950 note)" if traffic else None,
        "launches": int(n_launch.value), "avg_launch_us": round(ms.value * 1e3 / max(n_launch.value, 1), 2),
        "flops_per_launch": flops.value / max(n_launch.value, 1),
        "measured": "hipEvents around every launch of the kernel on its stream, second pass of the same K steps",
        "end_to_end_frac_of_peak": round(passages_per_s / world * GFLOP_PER_PASSAGE / 1e3 / peak, 4),
    }

    # ---------------- search leg -----------------------------------------------------------
    search = None
    if not a.no_search:
        rows = a.index_rows // world + (1 if rank < a.index_rows % world else 0)
        offset = rank * (a.index_rows // world) + min(rank, a.index_rows % world)
        index = FlatIPIndex(768, device=device, precision="f16_rescore" if a.precision == "bf16" else "f32")
        g = torch.Generator(device=device).manual_seed(77 + rank)
        shared = torch.randn(1, 768, device=device, generator=torch.Generator(device=device).manual_seed(5))
        index._reserve(rows)
        for s in range(0, rows, 1 << 20):                         # anisotropic, CLS-like: mean + noise
            n = min(1 << 20, rows - s)
            index.add(torch.randn(n, 768, device=device, generator=g) * 0.05 + shared * 0.05)
        nq_local = a.queries // world + (1 if rank < a.queries % world else 0)
        q_local = torch.randn(nq_local, 768, device=device, generator=g) * 0.05 + shared * 0.05

        def search_once():
            if world > 1:
                nmax = (a.queries + world - 1) // world
                pad = torch.zeros(nmax, 768, device=device)
                pad[:nq_local] = q_local
                allq = torch.empty(world * nmax, 768, device=device)
                dist.all_gather_into_tensor(allq, pad)
                sizes = [a.queries // world + (1 if r < a.queries % world else 0) for r in range(world)]
                queries = torch.cat([allq[r * nmax:r * nmax + sizes[r]] for r in range(world)])
            else:
                queries = q_local
            D, I = index.search_device(queries, a.topk, id_offset=offset)
            if world > 1:
                pd = [torch.empty_like(D) for _ in range(world)] if rank == 0 else None
                pi = [torch.empty_like(I) for _ in range(world)] if rank == 0 else None
                dist.gather(D, pd, dst=0)
                dist.gather(I, pi, dst=0)
                if rank == 0:
                    D, I = merge_topk(torch.stack(pd), torch.stack(pi), a.topk)
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
        es = 2 if a.precision == "bf16" else 4
        scan_bytes = rows * 768 * es
        search = {
            "metric": "queries/sec exact top-%d over %d x 768" % (a.topk, a.index_rows),
            "value": round(a.queries / t_s, 1), "unit": "queries/s", "queries": a.queries,
            "seconds_per_batch": round(t_s, 4),
            "scaling": "strong (index rows fixed, sharded by rank)" if world > 1 else "single shard",
            "precision": ("f16 MFMA candidate scan (certified margin) + exact f32 re-score; ids == f32 scan" if a.precision == "bf16" else "exact f32 MFMA scan"),
            "scan_info": index.last_search_info,
            "algorithmic_tflops": round(2.0 * a.index_rows * 768 * a.queries / t_s / 1e12, 1),
            "frac_of_mfma_peak": round(2.0 * a.index_rows * 768 * a.queries / t_s / 1e12 / (peak * world), 4),
            "scan_kernel": {"tflops": round(sf.value / max(sms.value, 1e-9) / 1e9, 1),
                            "ms_per_search": round(sms.value / reps, 2), "launches_per_search": int(sl.value // reps),
                            "index_stream_GBps_if_read_once": round(scan_bytes / max(sms.value / reps, 1e-9) / 1e6, 1)},
        }
        del index
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.no_search)

    if rank == 0:
        line = {
            "metric": "passages/sec encode (bert-base DPR bi-encoder, 128 tok -> 768-d) "
                      "[+ queries/sec exact top-1000 over 8.8M x 768 in `search`]",
            "value": round(passages_per_s, 1), "unit": "passages/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(t_enc / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "bert-base DPR bi-encoder encode+search, MS MARCO 8.8M x 128-tok -> 768-d (BASELINE configs[1])",
                       "passages_per_step_per_gpu": a.batch, "seq_len": L, "global_batch": a.batch * world,
                       "index_rows": a.index_rows, "queries": a.queries, "topk": a.topk,
                       "weights": "random-init BertConfig() seed 0", "parallelism": f"shard{world}"},
            "roofline": roofline, "search": search, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
