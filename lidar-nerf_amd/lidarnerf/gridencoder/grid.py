"""GridEncoder — multi-resolution hash / tiled grid encoding on MI355X.

Python API of the reference module (lidarnerf/gridencoder/grid.py:141-235): same constructor arguments, same
`embeddings` Parameter [rows, level_dim] / `offsets` int32 buffer / `output_dim`, same forward(inputs, bound).
The arithmetic runs in liblidarnerf_hip.so (lnh_grid_encode_forward / _backward); there is no CPU path.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _hip

_GRIDTYPE = {"hash": 0, "tiled": 1}
_INTERP = {"linear": 0, "smoothstep": 1}


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """Rows per level: min(2^log2_hashmap_size, (res[+1])^D) rounded up to a multiple of 8 (grid.py:179-193)."""
    cap = 2 ** log2_hashmap_size
    offs = [0]
    for lvl in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** lvl))
        side = res if align_corners else res + 1
        rows = min(cap, side ** input_dim)
        offs.append(offs[-1] + int(math.ceil(rows / 8) * 8))
    return np.asarray(offs, dtype=np.int32)


def grid_forward_raw(inputs, embeddings, offsets_host, S, H, gridtype, align_corners, interp, want_dy_dx):
    """inputs [B,D] f32 in [0,1]; returns (outputs [L,B,C], dy_dx or None) — level-major like the kernel writes it."""
    _hip.require_cuda(inputs, embeddings)
    B, D = inputs.shape
    L, C = offsets_host.shape[0] - 1, embeddings.shape[1]
    out = torch.empty((L, B, C), device=inputs.device, dtype=embeddings.dtype)
    dy_dx = torch.empty((B, L * D * C), device=inputs.device, dtype=embeddings.dtype) if want_dy_dx else None
    _hip.call("lnh_grid_encode_forward", inputs.data_ptr(), embeddings.data_ptr(), offsets_host.data_ptr(),
              out.data_ptr(), B, D, C, L, float(S), int(H), _hip.ptr(dy_dx), gridtype, int(align_corners), interp,
              _hip.dtype_code(embeddings.dtype), tag=B)
    return out, dy_dx


_WORKSPACE = {}  # device -> uint8 scratch tensor for the bucketed backward (grow-only, reused every step)


def _workspace(device, nbytes):
    """Scratch of the bucketed table-gradient backward.  `nbytes` is what lnh_grid_backward_workspace_size prefers (one chunk
    of up to 4 M points: 3.1 GB at 4096 rays x 832 samples); LNH_BWD_WORKSPACE_MB caps it — the library then walks the batch
    in shorter chunks (1.6 GB: +2.7 % on the backward; it refuses, naming the minimum, below the plan of a 256 K-point chunk)."""
    cap = os.environ.get("LNH_BWD_WORKSPACE_MB")
    if cap:
        nbytes = min(int(nbytes), int(float(cap) * (1 << 20)))
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        # a step being captured in a hipGraph (LidarTrainer graph mode): the graph keeps the POINTER — a scratch from the
        # grow-only cache below would be freed under it the day a larger batch replaces it.  Take it from the graph's pool.
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    ws = _WORKSPACE.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _WORKSPACE.pop(device, None)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WORKSPACE[device] = ws
    return ws


def grid_backward_raw(grad_lbc, inputs, rows, offsets_host, S, H, gridtype, align_corners, interp, dy_dx):
    """grad [L,B,C] -> (grad_embeddings [rows,C] in grad's dtype, grad_inputs [B,D] or None)."""
    L, B, C = grad_lbc.shape
    D = inputs.shape[1]
    ge = torch.zeros((rows, C), device=grad_lbc.device, dtype=grad_lbc.dtype)
    if dy_dx is None and D == 3 and C == 2:
        # hot configuration: bucketed scatter-reduce, no atomic adds to HBM (lnh_grid_encode_backward_ws)
        code = _hip.dtype_code(grad_lbc.dtype)
        need = _hip.lib().lnh_grid_backward_workspace_size(offsets_host.data_ptr(), B, D, C, L, float(S), int(H),
                                                           gridtype, int(align_corners), code)
        if need > 0:
            ws = _workspace(grad_lbc.device, need)
            _hip.call("lnh_grid_encode_backward_ws", grad_lbc.data_ptr(), inputs.data_ptr(), offsets_host.data_ptr(),
                      ge.data_ptr(), B, D, C, L, float(S), int(H), gridtype, int(align_corners), interp, code,
                      ws.data_ptr(), ws.numel(), tag=B)
            return ge, None
    gi = torch.zeros((B, D), device=grad_lbc.device, dtype=grad_lbc.dtype) if dy_dx is not None else None
    _hip.call("lnh_grid_encode_backward", grad_lbc.data_ptr(), inputs.data_ptr(), None, offsets_host.data_ptr(),
              ge.data_ptr(), B, D, C, L, float(S), int(H), _hip.ptr(dy_dx), _hip.ptr(gi), gridtype,
              int(align_corners), interp, _hip.dtype_code(grad_lbc.dtype), tag=B)
    return ge, gi


class _GridEncode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets_host, S, H, calc_grad_inputs, gridtype, align_corners, interp):
        inputs = inputs.contiguous().float()
        # AMP rule of the reference (grid.py:54-57): fp16 tables under autocast when C is even, inputs stay fp32
        table = embeddings
        if torch.is_autocast_enabled() and embeddings.shape[1] % 2 == 0:
            table = embeddings.to(torch.half)
        table = table.contiguous()
        out, dy_dx = grid_forward_raw(inputs, table, offsets_host, S, H, gridtype, align_corners, interp,
                                      calc_grad_inputs)
        L, B, C = out.shape
        ctx.save_for_backward(inputs, dy_dx)
        ctx.meta = (offsets_host, S, H, gridtype, align_corners, interp, embeddings.shape[0], embeddings.dtype,
                    table.dtype)
        return out.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        offsets_host, S, H, gridtype, align_corners, interp, rows, param_dtype, table_dtype = ctx.meta
        B = inputs.shape[0]
        L = offsets_host.shape[0] - 1
        g = grad.to(table_dtype).view(B, L, -1).permute(1, 0, 2).contiguous()
        ge, gi = grid_backward_raw(g, inputs, rows, offsets_host, S, H, gridtype, align_corners, interp, dy_dx)
        if gi is not None:
            gi = gi.to(inputs.dtype)
        return gi, ge.to(param_dtype), None, None, None, None, None, None, None


def grid_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0):
    """Functional form with the reference's argument order (grid.py:24-41).  `offsets` may be a device or host
    int32 tensor; the C ABI wants it on the host (it never changes after construction)."""
    offsets_host = offsets if not offsets.is_cuda else offsets.cpu()
    return _GridEncode.apply(inputs, embeddings, offsets_host.contiguous(), float(np.log2(per_level_scale)),
                             base_resolution, calc_grad_inputs, gridtype, align_corners, interpolation)


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False,
                 interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:  # overrides per_level_scale (grid.py:158-161)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size, self.base_resolution = log2_hashmap_size, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype, self.gridtype_id = gridtype, _GRIDTYPE[gridtype]
        self.interpolation, self.interp_id = interpolation, _INTERP[interpolation]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size
        offs = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer("offsets", torch.from_numpy(offs))
        self._offsets_host = torch.from_numpy(offs.copy())  # host copy handed to the C ABI
        self.n_params = int(offs[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offs[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> "
                f"{int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} "
                f"gridtype={self.gridtype} align_corners={self.align_corners} interpolation={self.interpolation}")

    @property
    def log2_scale(self):
        return float(np.log2(self.per_level_scale))

    def _require_whole_table(self, what):
        # Sharded table optimizer (LidarTrainer(shard_table_optimizer=True)): between steps the fp32 parameter of a rank is
        # current on ITS rows only (the fused chain reads the all-gathered fp16 shadow instead).  Reading the parameter
        # here would silently use other ranks' stale rows.
        if getattr(self.embeddings, "_lnh_master_stale", False):
            raise RuntimeError(f"GridEncoder.{what}: the fp32 table is sharded over the ranks (sharded table optimizer); "
                               "call LidarTrainer.gather_table_state() on every rank first")

    def forward(self, inputs, bound=1):
        self._require_whole_table("forward")
        inputs = (inputs + bound) / (2 * bound)  # [-bound, bound] -> [0, 1]
        lead = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim)
        out = _GridEncode.apply(flat, self.embeddings, self._offsets_host, self.log2_scale, self.base_resolution,
                                flat.requires_grad, self.gridtype_id, self.align_corners, self.interp_id)
        return out.view(lead + [self.output_dim])

    @torch.autocast("cuda", enabled=False)
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """Adds the TV-regulariser gradient into embeddings.grad (grid.py:237-277)."""
        self._require_whole_table("grad_total_variation")
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
            B = inputs.shape[0]
        inputs = inputs.contiguous().to(self.embeddings.dtype)
        _hip.call("lnh_grad_total_variation", inputs.data_ptr(), self.embeddings.data_ptr(),
                  self.embeddings.grad.data_ptr(), self._offsets_host.data_ptr(), float(weight), B, self.input_dim,
                  self.level_dim, self.num_levels, self.log2_scale, self.base_resolution, self.gridtype_id,
                  int(self.align_corners), _hip.dtype_code(self.embeddings.dtype))
