"""Fused LiDAR render step: the whole `NeRFRenderer.run` + `NeRFNetwork.density/color` chain of the reference
(lidarnerf/nerf/renderer.py:99-298, lidarnerf/nerf/network.py:162-237) as ~16 kernel launches forward and ~10
backward, with explicit gradients.

Used by NeRFNetwork.render (both flavours: `network` and the tcnn-shaped `network_tcnn`, through `model.fused_spec()`)
for the configuration the reference trains (`cal_lidar_color=True`, hash-grid L=16 F=2, 64-wide sigma net with one
hidden layer, 64-wide LiDAR colour net with two) under fp16 autocast.  Anything else takes the modular path in
renderer.py / network.py, which computes the same values from the same kernels' unfused forms.

Forward                                                       kernels (liblidarnerf_hip.so)
  z = near + (far-near) linspace + perturb                    one torch.rand + lnh_lidar_coarse_samples
  x = (clip(o + d z) + b) / 2b                                lnh_lidar_sample_points
  feat = hashgrid(x)      [16, N(T+t), 2] fp16, row-mapped    lnh_grid_encode_forward_mapped
  h16, sigma = sigma_net(feat)  strided into [N,T+t,*]        lnh_density_mlp_forward   (weights: lnh_lidar_pack_weights)
  new_z, z_all, perm = resample(z, sigma)                     lnh_lidar_resample
  (same three steps for the t new samples, same buffers)
  sigma_m, w = merge + weights                                lnh_lidar_merge_weights
  cdir = W0[:, :kd] enc(d)      per RAY                       model's direction features + lnh_lidar_dir_term
  rgb = colour_net(h16[perm], cdir) * (w > 1e-4)              lnh_lidar_color_forward
  ws, depth, image = composite(sigma_m, rgb, z_all)           lnh_lidar_composite_forward
Backward runs the mirror image over all N(T+t) points at once; the hash-table gradient uses the bucketed
scatter-reduce (no HBM atomics) and, data-parallel, goes on the wire in fp16 window by window.
"""
import numpy as np
import torch
from torch.autograd import Function

from .. import _hip, parallel
from ..gridencoder.grid import _workspace


class FieldSpec:
    """What the fused chain needs from a field, independent of how the module stores it (`network.NeRFNetwork` keeps
    nn.Linear stacks + `encoder.embeddings`, `network_tcnn.NeRFNetwork` keeps flat tcnn-style `params` vectors):
    tensors that take part in autograd (possibly views of Parameters) + the per-ray direction features."""

    def __init__(self, grid, table, ws0, ws1, wc0, wc1, wc2, n_dir, dir_features, n_color_mats=3, table_param=None,
                 dir_freq_degree=None):
        self.grid, self.table, self.ws0, self.ws1 = grid, table, ws0, ws1
        self.table_param = table_param if table_param is not None else table  # the Parameter DP bookkeeping marks
        self.wc0, self.wc1, self.wc2, self.n_dir, self.dir_features = wc0, wc1, wc2, n_dir, dir_features
        self.n_color_mats = n_color_mats
        # set when dir_features is the frequency encoder of that degree on raw directions (n_dir = 3 + 6 * degree): the fused
        # chain then forms the features inside the direction-term kernel
        self.dir_freq_degree = dir_freq_degree


def _bucketed_backward_ok(enc):
    """The fused chain's table gradient takes the bucketed backward; tables with more than 64 buckets of 8192 rows per
    level (log2_hashmap_size > 19) are outside its plan (workspace size 0) and must take the modular path instead of
    failing in the middle of backward.  Cached on the encoder."""
    ok = getattr(enc, "_lnh_bucketed_ok", None)
    if ok is None:
        ok = _hip.lib().lnh_grid_backward_workspace_size(enc._offsets_host.data_ptr(), 1024, 3, 2, enc.num_levels,
                                                         enc.log2_scale, enc.base_resolution, 0, 0, _hip.LNH_F16) > 0
        enc._lnh_bucketed_ok = ok
    return ok


def supported(model, cal_lidar_color, num_steps, upsample_steps):
    """True when `model` has exactly the shapes the fused kernels are specialised for."""
    try:
        sp = model.fused_spec()
        enc = sp.grid
        ok = (cal_lidar_color and upsample_steps > 0 and model.bg_radius <= 0
              and enc.input_dim == 3 and enc.num_levels == 16
              and enc.level_dim == 2 and enc.gridtype_id == 0 and not enc.align_corners and enc.interp_id == 0
              and tuple(sp.ws0.shape) == (64, 32) and tuple(sp.ws1.shape) == (16, 64)
              and sp.n_color_mats == 3 and 1 <= sp.n_dir <= 128 and tuple(sp.wc0.shape) == (64, sp.n_dir + 15)
              and tuple(sp.wc1.shape) == (64, 64) and tuple(sp.wc2.shape) == (2, 64)
              and model.geo_feat_dim == 15 and (num_steps + upsample_steps) % 16 == 0
              and sp.table.is_cuda and _bucketed_backward_ok(enc))
        return bool(ok)
    except AttributeError:
        return False


def _grid_fwd(x01, table16, enc, B):
    L = enc.num_levels
    out = torch.empty((L, B, 2), dtype=torch.half, device=x01.device)
    _hip.call("lnh_grid_encode_forward", x01.data_ptr(), table16.data_ptr(), enc._offsets_host.data_ptr(),
              out.data_ptr(), B, 3, 2, L, enc.log2_scale, enc.base_resolution, None, 0, 0, 0, _hip.LNH_F16, tag=B)
    return out


def _grid_bwd_workspace(dev, enc, B):
    """(workspace tensor, bytes at its head the backward wants cleared) for a batch of B points — cached per encoder and B:
    the plan is host arithmetic, but the training step asks for it every step."""
    cache = enc.__dict__.setdefault("_lnh_bwd_ws_plan", {})
    ent = cache.get(B)
    off = enc._offsets_host
    if ent is None:
        L = enc.num_levels
        need = _hip.lib().lnh_grid_backward_workspace_size(off.data_ptr(), B, 3, 2, L, enc.log2_scale, enc.base_resolution, 0, 0,
                                                           _hip.LNH_F16)
        if len(cache) > 16:
            cache.clear()
        ent = cache[B] = [need, None]
    ws = _workspace(dev, ent[0])
    if ent[1] is None or ent[1][0] != ws.numel():
        clear = _hip.lib().lnh_grid_backward_workspace_clear_bytes(off.data_ptr(), B, 3, 2, enc.num_levels, enc.log2_scale,
                                                                   enc.base_resolution, 0, 0, 0, _hip.LNH_F16, ws.numel())
        ent[1] = (ws.numel(), int(clear))
    return ws, ent[1][1]


def _grid_bwd(g_feat, x01, g_table16, enc, B, ws=None, flags=0):
    """Table gradient of B points.  ws / flags: the caller has cleared the head of the workspace and / or the gradient table
    itself (lnh_grid_encode_backward_ws_ex: LNH_BWD_WS_CLEARED, LNH_BWD_TABLE_ZERO)."""
    L = enc.num_levels
    off = enc._offsets_host
    if ws is None:
        ws, _ = _grid_bwd_workspace(g_feat.device, enc, B)
    _hip.call("lnh_grid_encode_backward_ws_ex", g_feat.data_ptr(), x01.data_ptr(), off.data_ptr(), g_table16.data_ptr(),
              B, 3, 2, L, enc.log2_scale, enc.base_resolution, 0, 0, 0, _hip.LNH_F16, ws.data_ptr(), ws.numel(), 0, L, 0,
              int(flags), tag=B, timer="lnh_grid_encode_backward_ws")


# Data parallel: level windows of the table gradient.  The scatter pass of the backward runs once for all levels
# (lnh_grid_encode_backward_ws_begin); the reduce pass runs per window (_finish), and a window's rows are final as soon as its
# reduce has run, so its collective (fp16, RCCL's own stream) overlaps with the reduce of the next window.  What a window
# costs in compute, measured on one MI355X with the exchange a no-op (bench.py --dp-windows, 4096 rays, ms per step / backward
# us), round 4 — after the reduce pass stopped being as slow as its slowest bucket (rows of the dense levels dealt to 64
# buckets):   no windows 2.149 / 854    1 window 2.154 / 855    2 (0,10,16) 2.192 / 889    3 (0,8,12,16) 2.198 / 890
#             4 (0,7,10,13,16) 2.270 / 957    6 2.412 / 1093
# (round 3: 1 window 2.309 / 1013, 2 -> 2.498 / 1196, 3 -> 2.529 / 1221, 4 -> 2.603 / 1291; round 2 cut the SCATTER pass
# into windows as well: 2.683 / 1362.)  Three windows cost what two cost and cut the table into thirds by rows (39 % / 31 % / 31 %):
# only the last third's exchange has nothing left to hide behind, instead of 46 % with two.  Which count wins in the end
# depends on the all-reduce time, which no box of this build could measure (LNH_DP_WINDOWS overrides the cut for that).
import os as _os
_DP_LEVEL_WINDOWS = ((0, 8), (8, 12), (12, 16))
if _os.environ.get("LNH_DP_WINDOWS"):  # developer override for A/B runs: "0,10,16" = two windows
    _c = [int(v) for v in _os.environ["LNH_DP_WINDOWS"].split(",")]
    # (must be the same on every rank: the shard layout follows the windows)
    if len(_c) < 2 or _c[0] != 0 or _c[-1] != 16 or any(b <= a for a, b in zip(_c[:-1], _c[1:])):
        raise ValueError(f"LNH_DP_WINDOWS={_os.environ['LNH_DP_WINDOWS']!r}: need increasing level cuts from 0 to 16, "
                         "e.g. '0,10,16'")
    _DP_LEVEL_WINDOWS = tuple(zip(_c[:-1], _c[1:]))
FORCE_DP_WINDOWS = False  # bench.py --dp-windows: take the windowed backward on one GPU too (the exchange is a no-op)


def _grid_bwd_windows(g_feat, x01, g_table16, enc, B, ws=None, flags=0):
    """The table-gradient backward as a generator over level windows: `lnh_grid_encode_backward_ws_begin` once (everything
    but the last reduce pass: the scatter pass stays in ONE piece — cut into level windows it costs 0.33 ms more per 3.4 M
    points, profiles/r03_bench_dpwindows*.json of the first cut), then `..._finish` per window; yields (l0, l1) when the rows
    of that window are final, so that the caller can hand them to a collective while the next window is being reduced."""
    L = enc.num_levels
    off = enc._offsets_host
    if ws is None:
        ws, _ = _grid_bwd_workspace(g_feat.device, enc, B)
    args = (g_feat.data_ptr(), x01.data_ptr(), off.data_ptr(), g_table16.data_ptr(), B, 3, 2, L, enc.log2_scale,
            enc.base_resolution, 0, 0, 0, _hip.LNH_F16, ws.data_ptr(), ws.numel())
    _hip.call("lnh_grid_encode_backward_ws_ex", *args, 0, L, 1, int(flags), tag=B, timer="lnh_grid_encode_backward_ws_begin")
    for l0, l1 in (_DP_LEVEL_WINDOWS if L == 16 else ((0, L),)):
        _hip.call("lnh_grid_encode_backward_ws_ex", *args, l0, l1, 2, int(flags), timer="lnh_grid_encode_backward_ws_finish")
        yield l0, l1


def _grid_bwd_overlapped(g_feat, x01, g_table16, enc, B, table_param, ws=None, flags=0):
    """_grid_bwd + parallel.allreduce_half_table, pipelined over level windows.  Returns the handles to wait on."""
    off = enc._offsets_host
    handles = []
    for l0, l1 in _grid_bwd_windows(g_feat, x01, g_table16, enc, B, ws, flags):
        h = parallel.allreduce_half_table(g_table16[int(off[l0]):int(off[l1])], table_param)
        if h is not None:
            handles.append(h)
    return handles


def _grid_bwd_sharded(g_feat, x01, enc, B, table_param):
    """Data parallel, sharded table optimizer (parallel.py "second cut"): per level window, the finished rows are copied
    into a buffer padded to world * shard rows and REDUCE-SCATTERED (fp16, SUM) — each rank keeps only its rows.
    Returns [(r0, r1, shard tensor [s, 2] fp16, handle, padded buffer)] per window of table rows [r0, r1): this rank's shard
    holds rows [r0 + rank * s, r0 + (rank + 1) * s) (cut at r1).  Pipelined like the all-reduce: the collective of a window
    runs behind the reduce pass of the following windows."""
    import torch.distributed as dist
    off = enc._offsets_host
    world = dist.get_world_size()
    g_table16 = torch.zeros((int(off[-1]), 2), dtype=torch.half, device=g_feat.device)
    shards = []
    for l0, l1 in _grid_bwd_windows(g_feat, x01, g_table16, enc, B):
        r0, r1 = int(off[l0]), int(off[l1])
        s = parallel.shard_rows(r1 - r0, world)
        padded = torch.zeros((world * s, 2), dtype=torch.half, device=g_feat.device)
        padded[:r1 - r0] = g_table16[r0:r1]
        mine = torch.empty((s, 2), dtype=torch.half, device=g_feat.device)
        h = parallel.reduce_scatter_half(padded, mine)
        shards.append((r0, r1, mine, h, padded))
    table_param._lnh_grad_reduced = True
    return shards


def table16_of(param, embeddings=None, training=True):
    """fp16 compute copy of the hash table.  With the fused table optimizer (train_step.LidarTrainer) the copy is a
    persistent shadow that the optimizer kernel rewrites together with the fp32 master; it is tied to the master's
    torch version counter, so a write to the parameter through torch (load_state_dict, parallel.broadcast_parameters, a
    manual re-initialisation under no_grad) is noticed here and the shadow is re-cast.  Writes through `.data` (torch_ema's
    copy_to / restore) do not move the version counter: evaluation (`training=False`) therefore never trusts the shadow
    and casts the parameter as it is — 82 MB of traffic per render call, noise next to the render itself.  Without a
    shadow the table is cast per call (the autocast rule of grid.py:54-57)."""
    src = param if embeddings is None else embeddings
    shadow = getattr(param, "_lnh_table16", None)
    if shadow is not None and getattr(param, "_lnh_shard_optimizer", False):
        # sharded table optimizer: the fp32 master of this rank is current on its own rows only; the all-gathered shadow
        # IS the table (LidarTrainer.gather_table_state() completes the master for checkpoints).  A write to the parameter
        # through torch (load_state_dict, broadcast) moves its version counter — the optimizer kernel and the gather do not —
        # and then the parameter is whole and newer: re-cast.
        if getattr(param, "_lnh_table16_version", None) != param._version:
            if getattr(param, "_lnh_master_stale", False):
                # an in-place write that moved the version counter while only this rank's rows of the master are current:
                # re-casting would overwrite the other ranks' rows of the shadow with stale values
                raise RuntimeError("the fp32 hash table was written through torch while the sharded table optimizer holds "
                                   "only this rank's rows of it: call LidarTrainer.gather_table_state() (every rank) "
                                   "before modifying `embeddings`, or load through LidarTrainer.load_checkpoint")
            shadow.copy_(param.detach().reshape(shadow.shape))
            param._lnh_table16_version = param._version
        return shadow
    if shadow is None or not training:
        return src.detach().to(torch.half).contiguous()
    if getattr(param, "_lnh_table16_version", None) != param._version:
        shadow.copy_(param.detach().reshape(shadow.shape))
        param._lnh_table16_version = param._version
    return shadow


def _no_autocast(fn):
    """The kernel chain manages precision itself: run the glue ops (casts, the tiny per-ray GEMMs) with autocast off,
    otherwise fp32 operands of a matmul silently become fp16 tensors handed to kernels that expect fp32."""
    def wrapped(*args, **kwargs):
        with torch.autocast("cuda", enabled=False):
            return fn(*args, **kwargs)
    return wrapped


_SAMPLE_DIST = {}


def _sample_dist(N, value, dev):
    """[N] fp32 tensor filled with `value` — the same every step, so it is kept instead of re-filled (read-only)."""
    if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
        # (a captured step keeps the POINTER: never hand it a tensor this cache may drop later)
        return torch.full((N,), value, dtype=torch.float32, device=dev)
    key = (N, value, str(dev))
    t = _SAMPLE_DIST.get(key)
    if t is None:
        if len(_SAMPLE_DIST) > 16:
            _SAMPLE_DIST.clear()
        t = _SAMPLE_DIST[key] = torch.full((N,), value, dtype=torch.float32, device=dev)
    return t


class FusedLidarRender(Function):
    @staticmethod
    @_no_autocast
    def forward(ctx, rays_o, rays_d, noise, u, embeddings, ws0, ws1, wc0, wc1, wc2, model, density_scale, spec, mdt, near,
                far, T):
        # noise: [N*T] uniforms of the stratified perturbation, or None (evaluation)
        # mdt: element type of everything MLP-side (packed weights, sigma-net rows, their gradients): torch.half, or
        # torch.bfloat16 for the bf16-operand build of the kernels (config 5); hash features stay fp16 either way
        sfx = _hip.mlp_suffix(mdt)
        enc = spec.grid
        kd = spec.n_dir
        dev = rays_o.device
        N = rays_o.shape[0]
        t_new = u.shape[1]
        Ttot = T + t_new
        bound = float(model.bound)
        aabb = (model.aabb_train if model.training else model.aabb_infer).float().contiguous()
        # sample_dist = (fars - nears) / T with the fp32 roundings of renderer.py:129-156, formed on the host
        near32 = np.float32(model.min_near_lidar)
        sd = _sample_dist(N, float((np.float32(near32 * np.float32(81.0)) - near32) / np.float32(T)), dev)

        # fp16 copy of the table: maintained by the fused table optimizer when there is one (train_step.LidarTrainer),
        # otherwise cast here (the autocast rule of grid.py:54-57)
        table16 = table16_of(spec.table_param, embeddings, model.training)
        # fp32 master matrices (possibly strided views of flat parameter vectors) -> the flat fp16 vectors of the
        # kernels, one launch: wsig16 = [ws0 | ws1]; wcol16 = [(0 | wc0[:, kd:kd+15]) | wc1 | wc2 padded to 16 rows]
        wsig16 = torch.empty(64 * 32 + 16 * 64, dtype=mdt, device=dev)
        wcol16 = torch.empty(64 * 16 + 64 * 64 + 16 * 64, dtype=mdt, device=dev)
        mats = [m.detach() if m.dtype == torch.float32 and m.stride(-1) == 1 else m.detach().float().contiguous()
                for m in (ws0, ws1, wc0, wc1, wc2)]
        # direction features [N, kd] (constant along a ray), rounded to the MLP element type (what the MLP would see), and
        # the per-ray direction term of the colour head's first layer
        enc_d16 = torch.empty((N, kd), dtype=torch.float32, device=dev)
        cdir = torch.empty((N, 64), dtype=torch.float32, device=dev)
        z = torch.empty((N, T), dtype=torch.float32, device=dev)
        x01 = torch.empty((N * Ttot, 3), dtype=torch.float32, device=dev)
        deg = getattr(spec, "dir_freq_degree", None)
        prologue = deg is not None and 3 + 6 * deg == kd
        if prologue:
            # weight packing + direction term (frequency encoder folded in) + the coarse pass (stratified depths and their
            # grid coordinates) in ONE launch
            _hip.call("lnh_lidar_step_prologue" + sfx, mats[0].data_ptr(), mats[0].stride(0), mats[1].data_ptr(),
                      mats[1].stride(0), mats[2].data_ptr(), mats[2].stride(0), int(deg), mats[3].data_ptr(), mats[3].stride(0),
                      mats[4].data_ptr(), mats[4].stride(0), wsig16.data_ptr(), wcol16.data_ptr(),
                      None if noise is None else noise.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), aabb.data_ptr(), bound,
                      N, T, Ttot, float(near), float(far), z.data_ptr(), x01.data_ptr(), enc_d16.data_ptr(), cdir.data_ptr())
        else:
            _hip.call("lnh_lidar_pack_weights" + sfx, mats[0].data_ptr(), mats[0].stride(0), mats[1].data_ptr(),
                      mats[1].stride(0), mats[2].data_ptr(), mats[2].stride(0), kd, mats[3].data_ptr(), mats[3].stride(0),
                      mats[4].data_ptr(), mats[4].stride(0), wsig16.data_ptr(), wcol16.data_ptr())
            enc_d = spec.dir_features(rays_d).contiguous()
            _hip.call("lnh_lidar_dir_term" + sfx, enc_d.data_ptr(), mats[2].data_ptr(), mats[2].stride(0), N, kd,
                      enc_d16.data_ptr(), cdir.data_ptr())

        h16 = torch.empty((N * Ttot, 16), dtype=mdt, device=dev)
        sigma_pt = torch.empty((N, Ttot), dtype=torch.float32, device=dev)
        # coarse and importance samples of a ray live side by side (slots 0..T-1 | T..T+t-1) in ONE set of buffers, so
        # the backward pass is a single launch chain over all N*(T+t) points
        B_all = N * Ttot
        L = enc.num_levels
        feat = torch.empty((L, B_all, 2), dtype=torch.half, device=dev)

        def density(zz, Tc, off, have_points=False):
            B = N * Tc
            if have_points:  # (the prologue / the resample kernel has written the coordinates of these samples)
                pass
            elif zz is None:  # coarse pass: the stratified depths and their grid coordinates in one launch
                _hip.call("lnh_lidar_coarse_sample_points", None if noise is None else noise.data_ptr(), rays_o.data_ptr(),
                          rays_d.data_ptr(), aabb.data_ptr(), bound, N, Tc, Ttot, float(near), float(far), z.data_ptr(),
                          x01.data_ptr())
            else:
                _hip.call("lnh_lidar_sample_points", rays_o.data_ptr(), rays_d.data_ptr(), zz.data_ptr(), aabb.data_ptr(),
                          bound, N, Tc, Ttot, off, x01.data_ptr())
            _hip.call("lnh_grid_encode_forward_mapped", x01.data_ptr(), table16.data_ptr(),
                      enc._offsets_host.data_ptr(), feat.data_ptr(), B, Tc, Ttot, off, B_all, 2, L, enc.log2_scale,
                      enc.base_resolution, _hip.LNH_F16, tag=B)
            _hip.call("lnh_density_mlp_forward" + sfx, feat.data_ptr(), wsig16.data_ptr(), B, Tc, Ttot, off, B_all,
                      h16.data_ptr(), sigma_pt.data_ptr())

        density(None, T, 0, have_points=prologue)
        new_z = torch.empty((N, t_new), dtype=torch.float32, device=dev)
        z_all = torch.empty((N, Ttot), dtype=torch.float32, device=dev)
        perm = torch.empty((N, Ttot), dtype=torch.int32, device=dev)
        # (stage-1 densities = the first T columns of the [N, T+t] buffer: read in place, row stride T+t)
        # ... and the grid coordinates of the new samples are written by the same kernel
        _hip.call("lnh_lidar_resample_points", z.data_ptr(), sigma_pt.data_ptr(), Ttot, sd.data_ptr(), u.data_ptr(), N, T,
                  t_new, float(density_scale), new_z.data_ptr(), z_all.data_ptr(), perm.data_ptr(), rays_o.data_ptr(),
                  rays_d.data_ptr(), aabb.data_ptr(), bound, x01.data_ptr())
        density(new_z, t_new, T, have_points=True)
        if CAPTURE is not None:
            CAPTURE.update(z=z, new_z=new_z, z_all=z_all, perm=perm)

        sigma_m = torch.empty((N, Ttot), dtype=torch.float32, device=dev)
        weights = torch.empty((N, Ttot), dtype=torch.float32, device=dev)
        rgb = torch.empty((N, Ttot, 2), dtype=torch.float32, device=dev)
        ws = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty((N, 2), dtype=torch.float32, device=dev)
        if Ttot <= 2048:
            # merged densities + weights, colour head, compositing sums: one launch, one wave per ray
            _hip.call("lnh_lidar_color_composite_forward" + sfx, z_all.data_ptr(), sigma_pt.data_ptr(), perm.data_ptr(),
                      sd.data_ptr(), h16.data_ptr(), cdir.data_ptr(), wcol16.data_ptr(), N, Ttot, float(density_scale),
                      sigma_m.data_ptr(), weights.data_ptr(), rgb.data_ptr(), ws.data_ptr(), depth.data_ptr(),
                      image.data_ptr())
        else:
            _hip.call("lnh_lidar_merge_weights", z_all.data_ptr(), sigma_pt.data_ptr(), perm.data_ptr(), sd.data_ptr(), N,
                      Ttot, float(density_scale), sigma_m.data_ptr(), weights.data_ptr())
            _hip.call("lnh_lidar_color_forward" + sfx, h16.data_ptr(), perm.data_ptr(), weights.data_ptr(),
                      cdir.data_ptr(), wcol16.data_ptr(), N, Ttot, rgb.data_ptr())
            _hip.call("lnh_lidar_composite_forward", z_all.data_ptr(), sigma_m.data_ptr(), rgb.data_ptr(), sd.data_ptr(),
                      N, Ttot, 2, float(density_scale), None, ws.data_ptr(), depth.data_ptr(), image.data_ptr())

        ctx.save_for_backward(x01, feat, h16, perm, weights, z_all, sigma_m, rgb, sd, cdir, enc_d16, wsig16, wcol16)
        ctx.model, ctx.dims, ctx.density_scale, ctx.enc = model, (N, T, t_new), density_scale, enc
        ctx.table_param, ctx.mdt = spec.table_param, mdt
        ctx.param_dtypes = (embeddings.dtype, ws0.dtype, ws1.dtype, wc0.dtype, wc1.dtype, wc2.dtype)
        # the five small matrices as the Parameters they are (network.NeRFNetwork) — or None when the field hands out views of
        # flat parameter vectors (network_tcnn): LidarTrainer takes their gradients straight from this node's arena then
        small = (ws0, ws1, wc0, wc1, wc2)
        ctx.small_params = small if all(isinstance(w, torch.nn.Parameter) and w.dtype == torch.float32 and
                                        w.is_contiguous() for w in small) else None
        ctx.mark_non_differentiable(weights, z_all)
        # outputs the loss does not use (weights_sum in the LiDAR loss) arrive as None instead of a zero tensor filled
        # for the occasion; the compositing backward takes a null pointer for them
        ctx.set_materialize_grads(False)
        return ws, depth, image, weights, z_all

    @staticmethod
    @_no_autocast
    def backward(ctx, g_ws, g_depth, g_image, _gw, _gz):
        x01, feat, h16, perm, weights, z_all, sigma_m, rgb, sd, cdir, enc_d16, wsig16, wcol16 = ctx.saved_tensors
        model, (N, T, t_new), ds = ctx.model, ctx.dims, ctx.density_scale
        enc = ctx.enc
        dev = h16.device
        mdt, sfx = ctx.mdt, _hip.mlp_suffix(ctx.mdt)
        Ttot = T + t_new
        g_ws, g_depth, g_image = (None if g is None else g.contiguous().float() for g in (g_ws, g_depth, g_image))
        ptr = lambda t: None if t is None else t.data_ptr()

        g_sigma = torch.empty((N, Ttot), dtype=torch.float32, device=dev)
        # d loss / d rgb = weights (x) d loss / d image is formed inside the colour backward: the compositing backward
        # writes the density gradient only (grad_rgb = NULL)
        _hip.call("lnh_lidar_composite_backward", ptr(g_ws), ptr(g_depth), ptr(g_image),
                  z_all.data_ptr(), sigma_m.data_ptr(), rgb.data_ptr(), sd.data_ptr(), N, Ttot, 2, float(ds),
                  g_sigma.data_ptr(), None)
        if g_image is None:
            g_image = torch.zeros((N, 2), dtype=torch.float32, device=dev)

        g_h16 = torch.empty((N * Ttot, 16), dtype=mdt, device=dev)
        wws = _hip.wgrad_ws(dev)  # scratch of the fixed-order weight-gradient sums (one per device and stream, kept)
        kd = enc_d16.shape[1]
        n_col, n_sig, n_c0 = wcol16.numel(), wsig16.numel(), 64 * (kd + 15)
        # every gradient that is ACCUMULATED into (the small matrices' by the fixed-order sums, the table's by the reduce pass) is cleared
        # by one launch: the arena of all small gradients and — unless the sharded backward brings its own — the fp16 table
        zeros = torch.empty(n_col + n_sig + n_c0, dtype=torch.float32, device=dev)
        sharded = parallel.dp_active() and getattr(ctx.table_param, "_lnh_shard_optimizer", False)
        g_table16 = None if sharded else torch.empty((int(enc._offsets_host[-1]), 2), dtype=torch.half, device=dev)
        # ... and the cursors at the head of the table backward's workspace (its own clear launch is then skipped)
        bwd_ws, bwd_clear = _grid_bwd_workspace(dev, enc, N * Ttot)
        bwd_flags = 0 if sharded else (_hip.LNH_BWD_TABLE_ZERO | (_hip.LNH_BWD_WS_CLEARED if bwd_clear else 0))
        _hip.zero_regions((zeros, g_table16, None if sharded or not bwd_clear else bwd_ws[:bwd_clear]))
        g_wcol, g_wsig = zeros[:n_col], zeros[n_col:n_col + n_sig]
        ray_sum = torch.empty((N, 64), dtype=torch.float32, device=dev)
        _hip.call("lnh_lidar_color_backward_image" + sfx, g_image.data_ptr(), g_sigma.data_ptr(), h16.data_ptr(), perm.data_ptr(),
                  weights.data_ptr(), cdir.data_ptr(), wcol16.data_ptr(), N, Ttot, g_h16.data_ptr(), g_wcol.data_ptr(),
                  ray_sum.data_ptr(), *wws)
        g_w0g = g_wcol[:64 * 16].view(64, 16)
        g_wc0 = zeros[n_col + n_sig:].view(64, kd + 15)
        # the whole gradient of the colour head's first matrix in one launch: direction columns = S^T enc(d), geo-feature
        # columns copied out of the packed [64, 16] block the colour backward accumulated
        _hip.call("lnh_lidar_dir_term_backward", ray_sum.data_ptr(), enc_d16.data_ptr(), N, kd, g_w0g.data_ptr(),
                  g_wc0.data_ptr(), kd + 15, *wws)
        g_wc1 = g_wcol[64 * 16:64 * 16 + 64 * 64].view(64, 64)
        g_wc2 = g_wcol[64 * 16 + 64 * 64:].view(16, 64)[:2]
        dts = ctx.param_dtypes
        g_small = (g_wsig[:64 * 32].view(64, 32), g_wsig[64 * 32:].view(16, 64), g_wc0, g_wc1, g_wc2)
        if ctx.small_params is not None and getattr(ctx.table_param, "_lnh_direct_small_grads", False):
            # LidarTrainer's fused optimizer: the gradients are handed over as views of the arena (autograd's AccumulateGrad
            # would copy each view into a tensor of its own: five launches), and the arena goes on the wire as ONE tensor
            for p_, g_ in zip(ctx.small_params, g_small):
                p_.grad = g_
            ctx.table_param._lnh_small_arena = zeros
            g_small = (None,) * 5
        else:
            g_small = tuple(g_.to(dt_) for g_, dt_ in zip(g_small, dts[1:]))

        B_all = N * Ttot
        g_feat = torch.empty((enc.num_levels, B_all, 2), dtype=torch.half, device=dev)
        _hip.call("lnh_density_mlp_backward" + sfx, g_h16.data_ptr(), feat.data_ptr(), wsig16.data_ptr(), B_all, Ttot, Ttot, 0,
                  g_feat.data_ptr(), g_wsig.data_ptr(), *wws)
        if sharded:
            # data parallel, sharded table optimizer: reduce-scatter per window; the trainer steps this rank's rows
            ctx.table_param._lnh_grad16_shards = _grid_bwd_sharded(g_feat, x01, enc, B_all, ctx.table_param)
            ctx.table_param._lnh_grad16_div = parallel.world_size()
            ctx.table_param._lnh_grad16 = None
            return (None, None, None, None, None) + g_small + (None,) * 7
        if parallel.dp_active() or FORCE_DP_WINDOWS:
            # data parallel: the table gradient goes on the wire as fp16, window by window, behind the kernels of the
            # following windows
            handles = _grid_bwd_overlapped(g_feat, x01, g_table16, enc, B_all, ctx.table_param, bwd_ws, bwd_flags)
            if getattr(ctx.table_param, "_lnh_keep_grad16", False):
                # the fused table optimizer is the only consumer: it waits right before its kernels (train_step.py), so the
                # last window's bytes travel under the MLP gradients' all-reduce and the loss-scale bookkeeping instead of
                # being waited for here, inside backward
                ctx.table_param._lnh_grad16_handles = handles
            else:
                for handle in handles:
                    handle.wait()
        else:
            _grid_bwd(g_feat, x01, g_table16, enc, B_all, bwd_ws, bwd_flags)
        world = parallel.world_size()
        if getattr(ctx.table_param, "_lnh_keep_grad16", False):
            # the fused table optimizer consumes the fp16 gradient directly: no fp32 copy, no .grad on the table
            # (data parallel: the gradient is the SUM over ranks; the optimizer divides by `_lnh_grad16_div` in fp32)
            ctx.table_param._lnh_grad16 = g_table16
            ctx.table_param._lnh_grad16_div = world
            g_table = None
        else:
            g_table = g_table16.to(dts[0])
            if world > 1:
                g_table.div_(world)  # sum over ranks -> mean, after the widening
        return (None, None, None, None, g_table) + g_small + (None,) * 7


MASK_STATS = None  # bench.py: set to a list to collect, per render call, the fraction of samples with weight > 1e-4
CAPTURE = None  # parity tests: set to a dict to receive the sample depths (z, new_z, z_all, perm) of the next render call


def render_lidar(model, rays_o, rays_d, num_steps, upsample_steps, perturb, noise=None, u=None):
    """Drop-in for NeRFRenderer.run(cal_lidar_color=True) on a supported NeRFNetwork.  `noise` [N, num_steps] / `u`
    [N, upsample_steps] in [0, 1) replace the random draws (perturbation / sample_pdf), as in NeRFRenderer.run."""
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3).float()
    rays_d = rays_d.contiguous().view(-1, 3).float()
    N, dev = rays_o.shape[0], rays_o.device
    # near = min_near_lidar, far = 81 * near: hard-coded 1 m .. 81 m in scene units (renderer.py:129-138); the fp32
    # products are formed exactly as the tensor code of the reference forms them
    near = torch.tensor(float(model.min_near_lidar), dtype=torch.float32)
    far = near * 81.0
    # one random draw serves the stratified perturbation (first N*num_steps values) and, in training mode, the
    # importance-sampling positions u (the rest)
    n_noise = N * num_steps if perturb and noise is None else 0
    n_u = N * upsample_steps if model.training and u is None else 0
    rnd = torch.rand(n_noise + n_u, device=dev) if n_noise + n_u else None
    if not perturb:
        noise = None
    elif noise is None:
        noise = rnd[:n_noise]
    else:
        noise = noise.to(dev).float().reshape(N * num_steps).contiguous()
    if u is not None:
        u = u.to(dev).float().reshape(N, upsample_steps).contiguous()
    elif model.training:
        u = rnd[n_noise:].view(N, upsample_steps)
    else:
        u = torch.linspace(0.5 / upsample_steps, 1.0 - 0.5 / upsample_steps, upsample_steps,
                           device=dev).expand(N, upsample_steps).contiguous()
    sp = model.fused_spec()
    from ..ffmlp.ffmlp import mlp_dtype
    ws, depth, image, weights, _ = FusedLidarRender.apply(rays_o, rays_d, noise, u, sp.table, sp.ws0, sp.ws1, sp.wc0,
                                                          sp.wc1, sp.wc2, model, model.density_scale, sp, mlp_dtype(),
                                                          float(near), float(far), num_steps)
    if MASK_STATS is not None:
        MASK_STATS.append((weights > 1e-4).float().mean())
    return {"depth_lidar": depth.view(*prefix), "image_lidar": image.view(*prefix, 2), "weights_sum_lidar": ws}


# ------------------------------------------------------------------------------------------------ occupancy-grid chain
class FusedLidarRagged(Function):
    """The occupancy-grid render step (BASELINE config 4; renderer.run_cuda) as one autograd node over the marcher's ragged
    samples: hash-grid encode -> sigma net -> LiDAR colour head -> ragged compositing, and the mirror image backward, on the
    same kernels as the dense chain.  What the modular path does through four autograd nodes, a torch.cat of [M, 90] colour
    inputs in fp32 and per-module casts, runs here as ~12 launches forward / ~12 backward; the hash-table gradient comes out
    of the bucketed scatter-reduce in fp16 and feeds the fused table optimizer (LidarTrainer) like the dense chain's.

    Differences to the dense chain that are properties of the path, not of this implementation: every marched sample lies in
    an occupied cell, so the colour head runs on ALL of them (no weight mask); a ray owns `rays[i, 2]` consecutive samples
    from `rays[i, 1]` on instead of a row of T, so the colour kernels walk the marcher's ray table (one wave per ray group,
    lnh_ragged_color_forward / _backward: the direction term once per ray, K = 16 per sample — round 5; rounds 3-4 assembled
    a [M, 96] input per sample for the generic MLP kernel)."""

    @staticmethod
    @_no_autocast
    def forward(ctx, xyzs, dirs, deltas, rays, rays_o, rays_d, embeddings, ws0, ws1, wc0, wc1, wc2, model, spec, mdt,
                T_thresh):
        sfx = _hip.mlp_suffix(mdt)
        enc = spec.grid
        dev = xyzs.device
        M, N, L = xyzs.shape[0], rays.shape[0], enc.num_levels
        bound, ds = float(model.bound), float(model.density_scale)
        if M == 0:
            # The marcher found no sample (every ray misses the occupied cells).  The node still exists: its outputs are
            # zeros CONNECTED to the graph, and its backward hands out zero gradients and takes part in the gradient
            # exchange — under data parallel a rank that skipped the table all-reduce would leave the others waiting.
            ctx.meta = (model, enc, spec.table_param, mdt, float(T_thresh), spec.n_dir, ds,
                        (embeddings.dtype, ws0.dtype, ws1.dtype, wc0.dtype, wc1.dtype, wc2.dtype))
            ctx.empty = True
            ctx.shapes = (ws0.shape, ws1.shape, wc0.shape, wc1.shape, wc2.shape)
            ctx.set_materialize_grads(False)
            z = torch.zeros(N, dtype=torch.float32, device=dev)
            return z, z.clone(), torch.zeros((N, 2), dtype=torch.float32, device=dev)
        ctx.empty = False
        table16 = table16_of(spec.table_param, embeddings, model.training)
        kd, deg = spec.n_dir, int(spec.dir_freq_degree)  # 75, 12
        mats = [m.detach() if m.dtype == torch.float32 and m.stride(-1) == 1 else m.detach().float().contiguous()
                for m in (ws0, ws1, wc0, wc1, wc2)]
        # the dense chain's packing: wcol16 = [(0 | wc0[:, kd:kd+15]) | wc1 | wc2 padded to 16 rows] — the direction columns of
        # the first matrix act once per RAY (cdir), a sample's colour input is its own 16-wide sigma-net row
        wsig16 = torch.empty(64 * 32 + 16 * 64, dtype=mdt, device=dev)
        wcol16 = torch.empty(64 * 16 + 64 * 64 + 16 * 64, dtype=mdt, device=dev)
        _hip.call("lnh_lidar_pack_weights" + sfx, mats[0].data_ptr(), mats[0].stride(0), mats[1].data_ptr(),
                  mats[1].stride(0), mats[2].data_ptr(), mats[2].stride(0), kd, mats[3].data_ptr(), mats[3].stride(0),
                  mats[4].data_ptr(), mats[4].stride(0), wsig16.data_ptr(), wcol16.data_ptr())
        enc_d16 = torch.empty((N, kd), dtype=torch.float32, device=dev)
        cdir = torch.empty((N, 64), dtype=torch.float32, device=dev)
        _hip.call("lnh_lidar_dir_term_freq" + sfx, rays_d.data_ptr(), deg, mats[2].data_ptr(), mats[2].stride(0), N,
                  enc_d16.data_ptr(), cdir.data_ptr())
        x01 = torch.empty((M, 3), dtype=torch.float32, device=dev)
        _hip.call("lnh_ragged_points", xyzs.data_ptr(), bound, M, x01.data_ptr())
        feat = torch.empty((L, M, 2), dtype=torch.half, device=dev)
        _hip.call("lnh_grid_encode_forward", x01.data_ptr(), table16.data_ptr(), enc._offsets_host.data_ptr(),
                  feat.data_ptr(), M, 3, 2, L, enc.log2_scale, enc.base_resolution, None, 0, 0, 0, _hip.LNH_F16, tag=M)
        h16 = torch.empty((M, 16), dtype=mdt, device=dev)
        sigma = torch.empty(M, dtype=torch.float32, device=dev)
        _hip.call("lnh_density_mlp_forward" + sfx, feat.data_ptr(), wsig16.data_ptr(), M, M, M, 0, 0, h16.data_ptr(),
                  sigma.data_ptr())
        # colour head, one wave per ray of the marcher's table (round 5; before: a [M, 96] input assembled per sample, the
        # generic 96 -> 64 -> 64 -> 16 MLP kernel and a sigmoid pass)
        # (rows no ray owns stay 0.)  run_cuda hands over rows its prologue launch has already cleared
        rgb = getattr(model, "_lnh_rgb_rows", None)
        model._lnh_rgb_rows = None
        if rgb is None or rgb.shape != (M, 2) or rgb.device != dev or rgb.dtype != torch.float32:
            rgb = torch.zeros((M, 2), dtype=torch.float32, device=dev)
        _hip.call("lnh_ragged_color_forward" + sfx, h16.data_ptr(), rays.data_ptr(), cdir.data_ptr(), wcol16.data_ptr(), N, M,
                  rgb.data_ptr())
        sig_s = sigma * ds if ds != 1.0 else sigma
        ws = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty((N, 2), dtype=torch.float32, device=dev)
        _hip.call("lnh_lidar_composite_rays_train_forward", sig_s.data_ptr(), rgb.data_ptr(), deltas.data_ptr(),
                  xyzs.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), rays.data_ptr(), M, N, 2, float(T_thresh),
                  ws.data_ptr(), depth.data_ptr(), image.data_ptr())
        ctx.save_for_backward(x01, feat, h16, sig_s, cdir, enc_d16, rgb, deltas, xyzs, rays_o, rays_d, rays, ws, depth, image,
                              wsig16, wcol16)
        ctx.meta = (model, enc, spec.table_param, mdt, float(T_thresh), kd, ds,
                    (embeddings.dtype, ws0.dtype, ws1.dtype, wc0.dtype, wc1.dtype, wc2.dtype))
        small = (ws0, ws1, wc0, wc1, wc2)
        ctx.small_params = small if all(isinstance(w, torch.nn.Parameter) and w.dtype == torch.float32 and
                                        w.is_contiguous() for w in small) else None
        ctx.set_materialize_grads(False)
        return ws, depth, image

    @staticmethod
    @_no_autocast
    def backward(ctx, g_ws, g_depth, g_image):
        model, enc, table_param, mdt, T_thresh, kd, ds, dts = ctx.meta
        if ctx.empty:  # no sample: zero gradients, but the same collectives as every other rank
            dev = table_param.device
            g_table16 = torch.zeros((int(enc._offsets_host[-1]), 2), dtype=torch.half, device=dev)
            world = parallel.world_size()
            if parallel.dp_active():
                off = enc._offsets_host
                handles = []
                for l0, l1 in (_DP_LEVEL_WINDOWS if enc.num_levels == 16 else ((0, enc.num_levels),)):
                    h = parallel.allreduce_half_table(g_table16[int(off[l0]):int(off[l1])], table_param)
                    if h is not None:
                        handles.append(h)
                for h in handles:
                    h.wait()
            if getattr(table_param, "_lnh_keep_grad16", False):
                table_param._lnh_grad16, table_param._lnh_grad16_div = g_table16, world
                g_table = None
            else:
                g_table = g_table16.to(dts[0])
            zw = [torch.zeros(sh, dtype=dt, device=dev) for sh, dt in zip(ctx.shapes, dts[1:])]
            return (None, None, None, None, None, None, g_table, zw[0], zw[1], zw[2], zw[3], zw[4], None, None, None, None)
        (x01, feat, h16, sig_s, cdir, enc_d16, rgb, deltas, xyzs, rays_o, rays_d, rays, ws, depth, image, wsig16,
         wcol16) = ctx.saved_tensors
        sfx = _hip.mlp_suffix(mdt)
        dev = x01.device
        M, N, L = x01.shape[0], rays.shape[0], enc.num_levels
        zN = lambda g, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if g is None else g.contiguous().float()
        g_ws, g_depth, g_image = zN(g_ws, (N,)), zN(g_depth, (N,)), zN(g_image, (N, 2))
        # every buffer that is accumulated into or only partly written — both compositing gradients, the sigma-net rows'
        # gradient (rows no ray owns), the per-ray sums, the small matrices' gradients, the fp16 table gradient, the cursors
        # of the table backward — cleared by ONE launch
        gsf = torch.empty(M * 3, dtype=torch.float32, device=dev)
        n_col, n_sig, n_c0 = wcol16.numel(), wsig16.numel(), 64 * (kd + 15)
        zeros = torch.empty(n_col + n_sig + n_c0, dtype=torch.float32, device=dev)
        g_table16 = torch.empty((int(enc._offsets_host[-1]), 2), dtype=torch.half, device=dev)
        g_h16 = torch.empty((M, 16), dtype=mdt, device=dev)
        ray_sum = torch.empty((N, 64), dtype=torch.float32, device=dev)
        bwd_ws, bwd_clear = _grid_bwd_workspace(dev, enc, M)
        bwd_flags = _hip.LNH_BWD_TABLE_ZERO | (_hip.LNH_BWD_WS_CLEARED if bwd_clear else 0)
        _hip.zero_regions((gsf, zeros, g_table16, g_h16, ray_sum, bwd_ws[:bwd_clear] if bwd_clear else None))
        gs, gf = gsf[:M], gsf[M:].view(M, 2)
        _hip.call("lnh_lidar_composite_rays_train_backward", g_ws.data_ptr(), g_depth.data_ptr(), g_image.data_ptr(),
                  sig_s.data_ptr(), rgb.data_ptr(), deltas.data_ptr(), xyzs.data_ptr(), rays_o.data_ptr(),
                  rays_d.data_ptr(), rays.data_ptr(), ws.data_ptr(), depth.data_ptr(), image.data_ptr(), M, N, 2,
                  T_thresh, gs.data_ptr(), gf.data_ptr())
        g_wcol, g_wsig = zeros[:n_col], zeros[n_col:n_col + n_sig]
        wws = _hip.wgrad_ws(dev)
        # colour head backward ray by ray: sigmoid, the three layers, all weight gradients, the sigma-net rows' gradient
        # (col 0 = the density gradient through trunc_exp) and the per-ray sum for the direction columns — one launch
        _hip.call("lnh_ragged_color_backward" + sfx, gf.data_ptr(), gs.data_ptr(), ds, h16.data_ptr(), rays.data_ptr(),
                  cdir.data_ptr(), wcol16.data_ptr(), N, M, g_h16.data_ptr(), g_wcol.data_ptr(), ray_sum.data_ptr(), *wws)
        g_w0g = g_wcol[:64 * 16].view(64, 16)
        g_wc0 = zeros[n_col + n_sig:].view(64, kd + 15)
        _hip.call("lnh_lidar_dir_term_backward", ray_sum.data_ptr(), enc_d16.data_ptr(), N, kd, g_w0g.data_ptr(),
                  g_wc0.data_ptr(), kd + 15, *wws)
        g_wc1 = g_wcol[64 * 16:64 * 16 + 64 * 64].view(64, 64)
        g_wc2 = g_wcol[64 * 16 + 64 * 64:].view(16, 64)[:2]
        g_feat = torch.empty((L, M, 2), dtype=torch.half, device=dev)
        _hip.call("lnh_density_mlp_backward" + sfx, g_h16.data_ptr(), feat.data_ptr(), wsig16.data_ptr(), M, M, M, 0,
                  g_feat.data_ptr(), g_wsig.data_ptr(), *wws)
        world = parallel.world_size()
        if parallel.dp_active():
            for handle in _grid_bwd_overlapped(g_feat, x01, g_table16, enc, M, table_param, bwd_ws, bwd_flags):
                handle.wait()
        else:
            _grid_bwd(g_feat, x01, g_table16, enc, M, bwd_ws, bwd_flags)
        if getattr(table_param, "_lnh_keep_grad16", False):
            table_param._lnh_grad16, table_param._lnh_grad16_div = g_table16, world
            g_table = None
        else:
            g_table = g_table16.to(dts[0])
            if world > 1:
                g_table.div_(world)
        g_small = (g_wsig[:64 * 32].view(64, 32), g_wsig[64 * 32:].view(16, 64), g_wc0, g_wc1, g_wc2)
        if ctx.small_params is not None and getattr(table_param, "_lnh_direct_small_grads", False):
            for p_, g_ in zip(ctx.small_params, g_small):  # (see FusedLidarRender.backward)
                p_.grad = g_
            table_param._lnh_small_arena = zeros
            g_small = (None,) * 5
        else:
            g_small = tuple(g_.to(dt_) for g_, dt_ in zip(g_small, dts[1:]))
        return (None, None, None, None, None, None, g_table) + g_small + (None, None, None, None)


def ragged_supported(model):
    """The occupancy-grid chain needs the same field shapes as the dense chain (and the in-kernel frequency encoder)."""
    try:
        sp = model.fused_spec()
    except AttributeError:
        return False
    return supported(model, True, 16, 16) and getattr(sp, "dir_freq_degree", None) is not None \
        and 3 + 6 * sp.dir_freq_degree == sp.n_dir and sp.n_dir + 15 <= 96


def render_lidar_ragged(model, xyzs, dirs, deltas, rays, rays_o, rays_d, T_thresh):
    """Field + compositing of renderer.run_cuda on the marcher's samples -> (weights_sum [N], depth [N], image [N, 2])."""
    from ..ffmlp.ffmlp import mlp_dtype
    sp = model.fused_spec()
    return FusedLidarRagged.apply(xyzs.contiguous(), dirs.contiguous(), deltas.contiguous(), rays.contiguous(),
                                  rays_o.contiguous(), rays_d.contiguous(), sp.table, sp.ws0, sp.ws1, sp.wc0, sp.wc1, sp.wc2,
                                  model, sp, mlp_dtype(), T_thresh)
