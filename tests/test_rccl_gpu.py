"""RCCL smoke test of the data-parallel path: one process per GPU, backend "nccl" (= RCCL on ROCm) over xGMI — what the
2-rank gloo tests cannot cover.  SKIPS on a box with fewer than 2 visible GPUs (every box this build has seen so far); it runs
the day a multi-GPU node exists: tests/dp_worker.py under torch.distributed.run with LNH_DIST_BACKEND=nccl — windowed fp16
all-reduce == single-process gradient, identical tables after DP steps, sharded table optimizer == replicated, sharded
evaluation == whole-frame evaluation — and `bench.py --gpus 2` with its `comm` block."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _free_port():
    """A rendezvous port nobody holds (a fixed one can still sit in TIME_WAIT from the test before)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs on one node (RCCL refuses two ranks on one device)")
def test_rccl_two_ranks_dp_worker():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(LNH_DIST_BACKEND="nccl", LNH_DP_WORKER_ONE_GPU_PER_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "tests", "dp_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("DP-OK") == 2, out[-3000:]


@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs on one node")
def test_rccl_bench_two_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "LNH_DIST_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-eval",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["comm"]["allreduce_exposed_ms"] is not None


def test_rccl_single_rank_runs_the_exchange_and_the_captured_dp_step():
    """ONE rank over RCCL on the one GPU every box has: a one-rank `nccl` process group with LNH_DP_SINGLE_RANK=1 takes the
    data-parallel code paths — windowed fp16 all-reduce (async, waited for at the optimizer), reduce-scatter -> sharded
    lnh_adam_table_step -> all-gather, sharded evaluation, collective-safe checkpoint — through RCCL itself, bit-identical
    to the step without a process group, and the DP step is captured in a hipGraph WITH its collectives and replayed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(LNH_DIST_BACKEND="nccl", LNH_DP_SINGLE_RANK="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("DP-OK") == 1 and "RCCL-GRAPH-OK" in out, out[-4000:]
    # ... and the collectives really went through RCCL: the worker reports the librccl it has mapped
    assert "RCCL-MAPPED" in out and "librccl" in out, out[-4000:]


def test_rccl_bench_single_rank_graph_line():
    """`bench.py --gpus 1` under a one-rank RCCL group: the line says the step (collectives included) was replayed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(LNH_DIST_BACKEND="nccl", LNH_DP_SINGLE_RANK="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-eval",
           "--no-cpu-baseline", "--no-mfma-states", "--rays", "1024"]
    # No retry: round 5 allowed one re-run here ("failed once in a dozen runs").  Round 6 looped this very command 50 times on
    # a box (tools/loop_rccl_bench.sh, profiles/r06_rccl_loop.txt) to find out what that was.
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(lines[0]) if lines else {}
    ok = (r.returncode == 0 and d.get("n_gpus") == 1 and "graph" in d and "error" not in d["graph"]
          and "collectives" in d["config"]["launch"])
    assert ok, f"rc {r.returncode} graph {d.get('graph')} launch {d.get('config', {}).get('launch')}\n" + \
        (r.stdout + r.stderr)[-4000:]


def test_rccl_refuses_two_ranks_on_one_gpu_loudly():
    """With fewer GPUs than ranks bench.py must refuse (no silent gloo / single-GPU fallback) unless the functional
    gloo mode is asked for explicitly."""
    if _n_gpus() >= 2:
        pytest.skip("a multi-GPU box runs the real test above")
    env = {k: v for k, v in os.environ.items() if k not in ("LNH_DIST_BACKEND",)}
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stdout + r.stderr) or "nccl" in (r.stdout + r.stderr).lower()
