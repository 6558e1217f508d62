"""Data-parallel path on real kernels: 2 and 4 ranks (gloo) sharing GPU 0 — see tests/dp_worker.py.  The collective backend
differs from production (RCCL needs one GPU per rank), the code path above it does not."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)


@pytest.mark.parametrize("ranks", [2, 4])
def test_data_parallel_training_step_matches_single_process(ranks):
    """2 and 4 ranks (4: shard boundaries inside levels, three or more addends per table row in the exchange)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LNH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", _free_port(), os.path.join(root, "tests", "dp_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("DP-OK") == ranks, out[-3000:]


def test_bench_launches_its_own_ranks_and_reports_the_exchange():
    """`python bench.py --gpus 2` outside torchrun re-launches itself with 2 ranks (here: gloo, both on GPU 0), prints ONE
    JSON line for the whole job with the all-reduce cost reported separately; a rank/`--gpus` mismatch is refused."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LNH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rays", "256",
           "--no-eval", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 256 * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]
    assert {"ms_per_step_inclusive", "ms_per_step_without_allreduce", "allreduce_exposed_ms", "payload_bytes",
            "window_plan"} <= set(d["comm"])
    assert d["comm"]["payload_bytes"]["table_gradient_fp16"] == 6837544 * 4 and len(d["comm"]["window_plan"]) == 3
    assert sum(w["rows"] for w in d["comm"]["window_plan"]) == 6837544
    # world size that does not match --gpus: refused
    env_bad = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run(cmd, env=env_bad, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)
