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
    if world > 1 and not dist.is_initialized():
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
    if world <= 1:
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


def allreduce_half_table(g_table16, param):
    """SUM the hash-table gradient over ranks WHILE IT IS STILL fp16 (27 MB instead of 55 MB on the wire; the kernels
    produce it in fp16 anyway).  Returns a handle to wait on; marks `param` so allreduce_gradients skips it.
    The division by the world size happens AFTER the sum, in fp32, by the consumer (the fused table optimizer folds it
    into its inverse loss scale; the autograd path divides the fp32 copy): dividing the fp16 values first would drop
    log2(world) bits at the bottom of the fp16 range and flush small scaled gradients to zero.  An fp16 overflow of the
    sum surfaces as inf, which GradScaler turns into a skipped step + smaller scale, exactly as it does for a
    single-rank overflow."""
    w = world_size()
    if w <= 1:
        return None
    param._lnh_grad_reduced = True
    return dist.all_reduce(g_table16, op=dist.ReduceOp.SUM, async_op=True)


def broadcast_parameters(module, src=0):
    """Make every replica start from rank `src`'s weights."""
    if not dist.is_initialized() or dist.get_world_size() <= 1:
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
