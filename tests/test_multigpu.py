"""Two processes on two GPUs over RCCL (torch.distributed.run): the multi-GPU path on real hardware -- the C-ABI
collectives, sharded FlatIPIndex search with the by-query-range merge against a single index, and a contrastive step with
cross-device negatives against the single-process full-batch step.  Skips on boxes with fewer than two GPUs (the
orchestration itself is covered on CPU by tests/test_distributed_cpu.py)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "mgpu_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "MGPU-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_multi_rank_branches_on_two_gpus():
    """bench.py's N > 1 branches (weak-scaling encode leg, query all-gather + sharded search + by-range merge, the
    max-over-ranks timing and rank 0's JSON line) on tiny sizes, launched exactly as the driver launches it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    repo = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(repo, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--index-rows", "40001",
                        "--queries", "37", "--topk", "100", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["search"]["value"] > 0 and line["search"]["queries"] == 37


def test_worker_on_a_one_rank_rccl_group():
    """The SAME worker under torch.distributed.run with ONE process: everything a one-GPU box can execute of the multi-GPU
    path -- the C-ABI collectives (om_comm_init from torch's rendezvous, om_allgather_rows, om_allreduce_grads,
    om_exchange_topk behind OPENMATCH_AMD_COMM=native), the sharded search with the fused all-to-all, the contrastive
    step with cross-device negatives and the bucketed all-reduce -- so that the two-GPU tests above do not meet it first."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "mgpu_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "MGPU-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_multi_rank_branches_on_a_one_rank_group():
    """bench.py's N > 1 branches on tiny sizes with OM_BENCH_FORCE_DIST=1 (one-rank RCCL group): exactly ONE line on stdout,
    the JSON, with rccl_ranks = 1 (RCCL's version banner goes to stderr)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OM_BENCH_FORCE_DIST="1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(s.getsockname()[1])
    repo = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--index-rows", "40001", "--queries", "37", "--topk", "100", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1
    assert line["value"] > 0 and line["search"]["value"] > 0 and line["search"]["queries"] == 37
