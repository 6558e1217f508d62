"""Data-parallel path on CPU: 2 processes, gloo.  DP invariant: after the gradient all-reduce every rank holds the
mean of the per-rank gradients == the gradient of the mean loss over the concatenated batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from lidarnerf import parallel
    r, l, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    big = torch.nn.Parameter(torch.zeros(1 << 20))           # "hash table": reduced on its own
    small = [torch.nn.Parameter(torch.zeros(64, 32)), torch.nn.Parameter(torch.zeros(16, 64))]  # coalesced
    if rank == 1:
        big.data.fill_(5.0)  # replicas start different; broadcast must fix that
    mod = torch.nn.ParameterList([big] + small)
    parallel.broadcast_parameters(mod)
    assert float(big.data.abs().max()) == 0.0
    x = torch.full((4,), float(rank + 1))
    loss = (big[:4] * x).sum() + sum((p * (rank + 1)).sum() for p in small)
    loss.backward()
    parallel.allreduce_gradients([big] + small, world)
    torch.testing.assert_close(big.grad[:4], torch.full((4,), 1.5))
    for p in small:
        torch.testing.assert_close(p.grad, torch.full_like(p, 1.5))
    # fp16 table-gradient path used by the fused backward
    tbl = torch.nn.Parameter(torch.zeros(8, 2))
    g16 = torch.full((8, 2), float(rank + 1) * 2.0, dtype=torch.float16)
    g16[0, 0] = 2.0 ** -24                                    # smallest fp16 subnormal: must survive the exchange
    h = parallel.allreduce_half_table(g16, tbl)
    h.wait()
    want = torch.full((8, 2), 6.0, dtype=torch.float16)       # the SUM over ranks; the consumer divides in fp32
    want[0, 0] = 2.0 ** -23
    torch.testing.assert_close(g16, want, rtol=0, atol=0)
    tbl.grad = g16.float() / world                            # what fused.backward does on the autograd path
    parallel.allreduce_gradients([tbl], world)  # must NOT reduce it a second time
    torch.testing.assert_close(tbl.grad, want.float() / 2)
    assert float(tbl.grad[0, 0]) == 2.0 ** -24                # (dividing the fp16 values first would have flushed it)
    assert tbl._lnh_grad_reduced is False
    a, b = parallel.shard_rays(67980, rank, world)
    assert (a, b) == ((0, 33990) if rank == 0 else (33990, 67980))
    assert parallel.max_over_ranks(float(rank), "cpu") == float(world - 1)
    # ---- second cut: reduce-scatter of padded fp16 windows, all-gather of the compute copy, sharded evaluation
    assert parallel.shard_rows(4920, 2) == 2460 and parallel.shard_rows(4921, 2) == 2464 and parallel.shard_rows(10, 8) == 4
    n_rows = 13                                               # a "window" of 13 rows: shards of 8 rows, 3 rows of padding
    s = parallel.shard_rows(n_rows, world)
    padded = torch.zeros((world * s, 2), dtype=torch.float16)
    padded[:n_rows] = torch.arange(n_rows * 2, dtype=torch.float16).view(n_rows, 2) * (rank + 1)
    mine = torch.empty((s, 2), dtype=torch.float16)
    parallel.reduce_scatter_half(padded, mine).wait()
    want = torch.zeros((world * s, 2), dtype=torch.float16)
    want[:n_rows] = torch.arange(n_rows * 2, dtype=torch.float16).view(n_rows, 2) * 3
    torch.testing.assert_close(mine, want[rank * s:(rank + 1) * s], rtol=0, atol=0)
    back = torch.empty((world * s, 2), dtype=torch.float16)
    parallel.all_gather_half(back, mine).wait()
    torch.testing.assert_close(back, want, rtol=0, atol=0)

    class Stub:                                               # render = a function of the ray alone
        def render(self, o, d, **kw):
            return {"depth_lidar": (o * d).sum(-1), "image_lidar": torch.stack([o[..., 0] + d[..., 1], d[..., 2]], -1)}
    g = torch.Generator().manual_seed(3)
    o, d = torch.rand(1, 1001, 3, generator=g), torch.rand(1, 1001, 3, generator=g)   # odd count: the last shard is short
    whole, parts = Stub().render(o, d), parallel.render_sharded(Stub(), o, d)
    for k in whole:
        assert parts[k].shape == whole[k].shape
        torch.testing.assert_close(parts[k], whole[k], rtol=0, atol=0)
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_dp_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def _worker_layout8(rank, world, port, out):
    """The sharded table optimizer's exchange on the REAL table layout (BASELINE configs 2/3/5: 16 levels, 6 837 544 rows,
    level windows 0-9 / 10-15) with 8 ranks: reduce-scatter of the padded fp16 windows, a stand-in for the shard step,
    all-gather of the compute copy (train_step.LidarTrainer._step_table_shards / fused._grid_bwd_sharded, minus the kernels)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import numpy as np
    from lidarnerf import parallel
    from lidarnerf.gridencoder.grid import level_offsets
    from lidarnerf.nerf.fused import _DP_LEVEL_WINDOWS
    parallel.init_from_env(backend="gloo")
    torch.set_num_threads(1)
    pls = np.exp2(np.log2(32768 / 16) / 15)
    off = level_offsets(3, 16, pls, 16, 19, False)
    rows = int(off[-1])
    assert rows == 6837544 and _DP_LEVEL_WINDOWS == ((0, 8), (8, 12), (12, 16))
    idx = torch.arange(rows, dtype=torch.int64)

    def grad_of(r):  # small integers: every partial sum is exact in fp16
        return torch.stack([(idx * 7 + r) % 13 - 6, (idx * 7 + 3 + r) % 13 - 6], -1).to(torch.float16)
    g = grad_of(rank)
    total = sum(grad_of(r).float() for r in range(world))                      # what the SUM over ranks must be
    table16 = torch.zeros((rows, 2), dtype=torch.float16)
    covered = 0
    for l0, l1 in _DP_LEVEL_WINDOWS:
        r0, r1 = int(off[l0]), int(off[l1])
        s = parallel.shard_rows(r1 - r0, world)
        assert s % 4 == 0 and world * s >= r1 - r0 and world * (s - 4) < r1 - r0   # 4-row aligned, minimal padding
        padded = torch.zeros((world * s, 2), dtype=torch.float16)
        padded[:r1 - r0] = g[r0:r1]
        mine = torch.empty((s, 2), dtype=torch.float16)
        parallel.reduce_scatter_half(padded, mine).wait()
        a = r0 + rank * s
        n = max(0, min(s, r1 - a))
        torch.testing.assert_close(mine[:n].float(), total[a:a + n], rtol=0, atol=0)
        assert float(mine[n:].abs().max()) == 0.0 if n < s else True            # beyond the window: padding
        covered += n
        stepped = (mine.float() * 0.25).to(torch.float16)                        # stand-in for lnh_adam_table_step on [a, a + n)
        full = torch.empty((world * s, 2), dtype=torch.float16)
        parallel.all_gather_half(full, stepped).wait()
        table16[r0:r1] = full[:r1 - r0]
    torch.testing.assert_close(table16.float(), (total * 0.25).to(torch.float16).float(), rtol=0, atol=0)
    cov = torch.tensor([covered], dtype=torch.int64)
    dist.all_reduce(cov)
    assert int(cov) == rows                                                       # every row has exactly one owner
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_sharded_table_exchange_on_the_real_layout_world8():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_layout8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(8)) == list(range(8))
