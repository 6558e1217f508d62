"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL (`backend="nccl"` on ROCm) over xGMI.

The reference's DDP hooks are dormant (lidarnerf/nerf/utils.py:589-593, never initialised — SURVEY.md §0.9), so this
is the path's own exchange step: rays are independent, the model (26-55 MB table + 86 KB of MLP weights) is
replicated, and the only collective is ONE gradient all-reduce per step:
  * the hash-table gradient (13.7 M fp32 = 54.7 MB) goes out as a single large message — on the fully connected
    xGMI mesh a few big collectives beat many small buckets (per-link bound, no NVSwitch);
  * all small MLP gradients are flattened into one coalesced buffer (one launch instead of ~8).
Sum-then-divide (never divide-then-sum: the fp16 table gradient would lose its low bits) keeps the DP mean equal to
the single-GPU mean over the concatenated batch when every rank draws the same number of rays
(lidarnerf/nerf/utils.py:746 uses .mean() over rays).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun-style env vars; returns (rank, local_rank, world_size).
    LNH_DIST_BACKEND overrides the backend (tests run two ranks on one GPU with gloo; RCCL refuses that)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or FORCE_SINGLE_RANK) and not dist.is_initialized():
        backend = backend or os.environ.get("LNH_DIST_BACKEND")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_rays(n_total, rank, world):
    """Contiguous [start, end) ray range of this rank (evaluation: shard a full range image)."""
    per = (n_total + world - 1) // world
    return min(rank * per, n_total), min((rank + 1) * per, n_total)


def allreduce_gradients(params, world=None, small_numel=1 << 20):
    """Average .grad over ranks.  Large tensors individually, small ones through one flat buffer."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world <= 1 and not dp_active():
        return
    big, small = [], []
    for p in params:
        if p.grad is None:
            continue
        if getattr(p, "_lnh_grad_reduced", False):  # already averaged inside backward (fp16 table gradient)
            p._lnh_grad_reduced = False
            continue
        (big if p.grad.numel() >= small_numel else small).append(p.grad)
    handles = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in big]
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for g in small:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
    for h, g in zip(handles, big):
        h.wait()
        g.div_(world)


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# A process group of ONE rank normally exchanges nothing (every `world > 1` test below is false).  LNH_DP_SINGLE_RANK=1 (or
# parallel.FORCE_SINGLE_RANK = True) makes an initialised one-rank group take the data-parallel code paths all the same —
# windowed fp16 all-reduce, reduce-scatter -> sharded optimizer -> all-gather, sharded evaluation — with collectives that
# are identities: the way to run the exchange through RCCL (backend "nccl") on a box with a single GPU, bit-identical to
# the step without a process group (tests/test_rccl_gpu.py).
FORCE_SINGLE_RANK = os.environ.get("LNH_DP_SINGLE_RANK", "") not in ("", "0")


def dp_active():
    """True when the gradient exchange runs: more than one rank, or a one-rank group with FORCE_SINGLE_RANK."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or FORCE_SINGLE_RANK


def backend():
    """Backend of the default process group ('nccl' = RCCL on ROCm, 'gloo'), or None without one."""
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def allreduce_half_table(g_table16, param):
    """SUM the hash-table gradient over ranks WHILE IT IS STILL fp16 (27 MB instead of 55 MB on the wire; the kernels
    produce it in fp16 anyway).  Returns a handle to wait on; marks `param` so allreduce_gradients skips it.
    The division by the world size happens AFTER the sum, in fp32, by the consumer (the fused table optimizer folds it
    into its inverse loss scale; the autograd path divides the fp32 copy): dividing the fp16 values first would drop
    log2(world) bits at the bottom of the fp16 range and flush small scaled gradients to zero.  An fp16 overflow of the
    sum surfaces as inf, which GradScaler turns into a skipped step + smaller scale, exactly as it does for a
    single-rank overflow."""
    if not dp_active():
        return None
    param._lnh_grad_reduced = True
    return dist.all_reduce(g_table16, op=dist.ReduceOp.SUM, async_op=True)


# ---- second cut: the table's optimizer state is SHARDED over the ranks (ZeRO-1 for the 13.7 M-parameter hash table) -------
# all-reduce + replicated Adam moves 2 x 27 MB per rank over xGMI and makes every rank read/write the whole 55 MB master
# table and its moments (383 MB of HBM traffic per step).  Sharded: reduce-scatter of the fp16 gradient (27 MB in, 27/N out),
# each rank steps ITS 1/N of the rows (lnh_adam_table_step on a row range), all-gather of the fp16 compute copy (27 MB) — the
# same bytes on the wire as an all-reduce (which IS reduce-scatter + all-gather), 1/N of the optimizer traffic per GPU, and
# the master table / moments of a rank only ever hold its own rows (SURVEY.md §8e, §5 "direct reduce-scatter + all-gather").
def shard_rows(n_rows, world, align=4):
    """Rows per rank when `n_rows` rows are dealt to `world` ranks in equal, `align`-row-aligned shards (the last may be
    partly or wholly padding): the collectives need equal shards, lnh_adam_table_step counts in multiples of 4 values."""
    per = (n_rows + world - 1) // world
    return (per + align - 1) // align * align


def reduce_scatter_half(padded, out_shard):
    """SUM-reduce `padded` [world * s, 2] fp16 over ranks and leave rows [rank * s, (rank + 1) * s) in `out_shard`
    (async handle).  fp16 on the wire, sum-then-divide like allreduce_half_table."""
    return dist.reduce_scatter_tensor(out_shard, padded, op=dist.ReduceOp.SUM, async_op=True)


def all_gather_half(out_padded, shard):
    """Inverse of the above for the fp16 compute copy of the table (async handle)."""
    return dist.all_gather_into_tensor(out_padded, shard, async_op=True)


def render_sharded(model, rays_o, rays_d, **render_kwargs):
    """Evaluation of a full range image on all ranks: each rank renders a contiguous range of the rays (shard_rays), the
    depth / image pieces are all-gathered, every rank returns the complete result (the reference's dormant hooks:
    lidarnerf/nerf/utils.py:1327-1350 gather `preds` the same way).  rays_o / rays_d [1, N, 3]."""
    world = world_size()
    if not dp_active():
        return model.render(rays_o, rays_d, **render_kwargs)
    rank = dist.get_rank()
    N = rays_o.shape[1]
    a, b = shard_rays(N, rank, world)
    per = (N + world - 1) // world
    part = model.render(rays_o[:, a:b].contiguous(), rays_d[:, a:b].contiguous(), **render_kwargs) if b > a else None
    out = {}
    for key, width in (("depth_lidar", 1), ("image_lidar", None)):
        if part is not None:
            piece = part[key].reshape(b - a, -1).float()
            width = piece.shape[1]
        if width is None:  # a rank without rays still has to know the channel count: the LiDAR image has 2
            width = 2
        mine = torch.zeros((per, width), dtype=torch.float32, device=rays_o.device)
        if part is not None:
            mine[:b - a] = piece
        full = torch.empty((world * per, width), dtype=torch.float32, device=rays_o.device)
        dist.all_gather_into_tensor(full, mine)
        full = full[:N]
        out[key] = full.reshape(1, N) if key == "depth_lidar" else full.reshape(1, N, width)
    return out


def broadcast_parameters(module, src=0):
    """Make every replica start from rank `src`'s weights."""
    if not dp_active():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.detach(), src=src)  # (detach shares the version counter: fused.table16_of sees the write)


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() <= 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
