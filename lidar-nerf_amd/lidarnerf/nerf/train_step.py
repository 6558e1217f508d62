"""LiDAR training step: loss of the reference Trainer.train_step (lidarnerf/nerf/utils.py:697-884) and the inner loop
of train_one_epoch (1206-1226): zero_grad -> autocast(render + loss) -> scaled backward -> [DP all-reduce] ->
scaler.step(optimizer) -> scaler.update() -> lr_scheduler.step().  Only the LiDAR branch exists in the reference
path (opt.enable_lidar is forced True, main_lidarnerf.py:229)."""
import os

import torch

from .. import parallel


def lidar_loss(outputs, images_lidar, alpha_d=1000.0, alpha_r=1.0, alpha_i=10.0):
    """utils.py:712-746 with the default criteria (L1 depth, MSE ray-drop, MSE intensity; main_lidarnerf.py:330-342).
    images_lidar [B,N,3] = (raydrop, intensity, depth).  Returns (loss, pred_depth, gt_depth)."""
    gt_raydrop = images_lidar[..., 0]
    gt_intensity = images_lidar[..., 1] * gt_raydrop
    gt_depth = images_lidar[..., 2] * gt_raydrop
    pred_raydrop = outputs["image_lidar"][..., 0]
    pred_intensity = outputs["image_lidar"][..., 1] * gt_raydrop
    pred_depth = outputs["depth_lidar"] * gt_raydrop
    per_ray = (alpha_d * (pred_depth - gt_depth).abs() + alpha_r * (pred_raydrop - gt_raydrop) ** 2
               + alpha_i * (pred_intensity - gt_intensity) ** 2)
    return per_ray.mean(), pred_depth, gt_depth


class _FusedLidarLoss(torch.autograd.Function):
    """lidar_loss as ONE kernel that also emits d loss / d (depth, image); backward only scales by the upstream scalar."""

    @staticmethod
    def forward(ctx, depth, image, gt, ad, ar, ai, patch=None, grad_scale=None):
        # patch = (px, py, scale, alpha_grad): the reference's patch epochs, structural-gradient term included
        # grad_scale (device scalar): the kernel multiplies the gradients it emits by it — the caller then starts
        # backward() from a gradient of ONE (LidarTrainer: the loss scale, without an element-wise launch in backward)
        from .. import _hip
        n = depth.numel()
        depth, image, gt = depth.reshape(n).float().contiguous(), image.reshape(n, 2).float().contiguous(), \
            gt.reshape(n, 3).float().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=depth.device)
        grads = torch.empty(3 * n, dtype=torch.float32, device=depth.device)  # [d/d depth (n) | d/d image (n, 2)]
        gs = None if grad_scale is None else grad_scale.data_ptr()
        if patch is None:
            _hip.call("lnh_lidar_loss", depth.data_ptr(), image.data_ptr(), gt.data_ptr(), n, float(ad), float(ar), float(ai),
                      gs, loss.data_ptr(), grads.data_ptr(), grads.data_ptr() + 4 * n)
        else:
            px, py, scale, ag = patch
            _hip.call("lnh_lidar_loss_patch", depth.data_ptr(), image.data_ptr(), gt.data_ptr(), n, int(px), int(py),
                      float(scale), float(ad), float(ar), float(ai), float(ag), gs, loss.data_ptr(), grads.data_ptr(),
                      grads.data_ptr() + 4 * n)
        ctx.save_for_backward(grads)
        ctx.n, ctx.prescaled = n, grad_scale is not None
        return loss

    @staticmethod
    def backward(ctx, g):
        (grads,) = ctx.saved_tensors
        # (pre-scaled gradients: the contract of grad_scale is that backward() starts from ONE)
        scaled = grads if ctx.prescaled else grads * g  # one launch for both
        return scaled[:ctx.n], scaled[ctx.n:].view(ctx.n, 2), None, None, None, None, None, None


class _ScaleGrad(torch.autograd.Function):
    """Identity whose gradient is multiplied by a device scalar (the loss scale, for the loss paths without a kernel)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.save_for_backward(scale)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.saved_tensors[0], None


def fused_lidar_loss(outputs, images_lidar, alpha_d=1000.0, alpha_r=1.0, alpha_i=10.0, patch=None, grad_scale=None):
    """lidar_loss (+ patch_gradient_loss when patch = (px, py, scale, alpha_grad)) through the single-launch kernel (GPU
    tensors only); same value and gradients.  grad_scale (device scalar): the gradients come out multiplied by it and
    backward() must then be started from a gradient of one."""
    depth, image = outputs["depth_lidar"], outputs["image_lidar"]
    loss = _FusedLidarLoss.apply(depth.reshape(-1), image.reshape(-1, 2), images_lidar, alpha_d, alpha_r, alpha_i, patch,
                                 grad_scale)
    return loss


def patch_gradient_loss(pred_depth, gt_depth, gt_raydrop, px, py, scale, alpha_grad=100.0):
    """utils.py:760-876 (grad_loss, non-sobel): |dx| of the prediction vs the SIGNED dx of the ground truth, masked to
    |gt dx| < 0.01 m and returned rays; only the x term enters the loss (the y terms are computed but unused)."""
    pred = pred_depth.reshape(-1, 1, px, py) / scale
    gt = gt_depth.reshape(-1, 1, px, py) / scale
    rd = gt_raydrop.reshape(-1, 1, px, py)
    pred_gx = (pred[..., :-1] - pred[..., 1:]).abs()
    gt_gx = gt[..., :-1] - gt[..., 1:]
    mask = rd[..., :-1] * (gt_gx.abs() < 0.01)
    return alpha_grad * (pred_gx * mask - gt_gx * mask).abs().mean()


def _hip_consts():
    from .. import _hip
    return _hip


# rungs per octave of the captured step's sample-capacity ladder (LidarTrainer._graph_capacity)
_LADDER_RUNGS_PER_OCTAVE = int(os.environ.get("LNH_GRAPH_LADDER", "8"))


class LidarTrainer:
    """The hot loop only (no logging / checkpoint / EMA: those are host glue outside the path).

    fused_table_optimizer (default on for fp16 + a fusable field on the GPU): the 13.7 M-parameter hash table — 99.8 %
    of all parameters — leaves torch.optim.Adam / GradScaler and is stepped by ONE kernel (lnh_adam_table_step) that
    reads the fp16 gradient the backward kernels produce and writes the fp32 master table, its moments and the fp16
    copy of the next step.  Same arithmetic as torch's fused Adam, same GradScaler contract (dynamic loss scale, inf/nan
    => the step is skipped for EVERY parameter and the scale backs off); the MLP parameters stay on torch's Adam."""

    def __init__(self, model, lr=1e-2, iters=30000, fp16=True, alpha_d=1000.0, alpha_r=1.0, alpha_i=10.0,
                 alpha_grad=100.0, scale=1.0, world_size=1, render_kwargs=None, fused_table_optimizer=True,
                 mlp_dtype=torch.float16, shard_table_optimizer=False, graph=False):
        # mlp_dtype: the autocast dtype — torch.float16 (the reference's --fp16) or torch.bfloat16 (BASELINE config 5:
        # bf16 MFMA MLPs; the hash table and its gradient stay fp16, so the dynamic loss scale is kept either way)
        # (the backward picks reduce-scatter or all-reduce from the process group, parallel.world_size(): a world_size
        #  argument that disagrees with it would silently run without the exchange and fail later, in gather_table_state)
        if world_size > 1 and parallel.world_size() != world_size:
            raise RuntimeError(f"LidarTrainer(world_size={world_size}) but the initialised process group has "
                               f"{parallel.world_size()} rank(s): call torch.distributed.init_process_group first")
        self.model, self.fp16, self.world, self.amp_dtype = model, fp16, world_size, mlp_dtype
        # the gradient exchange runs with more than one rank — or on a one-rank process group that was asked to exchange
        # all the same (parallel.FORCE_SINGLE_RANK: the way to run the collectives through RCCL on a one-GPU box)
        self.dp = world_size > 1 or parallel.dp_active()
        self.alpha = (alpha_d, alpha_r, alpha_i, alpha_grad)
        self.scale = scale
        self.render_kwargs = render_kwargs or {}
        # Adam(betas .9/.99, eps 1e-15) and lr * 0.1^(it/iters) (main_lidarnerf.py:389-391, 408-410)
        # get_params returns generators: materialise them; one fused kernel for all parameter groups on the GPU
        params = [dict(g, params=list(g["params"])) for g in model.get_params(lr)]
        # layout of the reference's optimizer (Adam over model.get_params(lr), main_lidarnerf.py:389-391): the order in
        # which its state_dict numbers the parameters — checkpoints are written / read in that layout
        self._ref_layout = [[p for p in g["params"]] for g in params]
        self.epoch, self.stats = 0, {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None}
        on_gpu = all(p.is_cuda for g in params for p in g["params"])
        self.table, self.sharded = None, False
        self.occupancy = bool(getattr(model, "cuda_ray", False))
        self.update_extra_interval, self.global_step = 16, 0
        if fused_table_optimizer and fp16 and on_gpu and hasattr(model, "fused_spec") and \
                (not self.occupancy or self._ragged_chain(model)):
            try:
                tp = model.fused_spec().table_param
            except AttributeError:
                tp = None
            others = [p for g in params for p in g["params"] if p is not tp]
            # the fused optimizer steps EVERY parameter: the table from its fp16 gradient and the other tensors (the MLP
            # weights: a handful of small fp32 matrices) in the same launch
            if tp is not None and tp.dtype == torch.float32 and tp.is_contiguous() and tp.numel() % 4 == 0 and \
                    len(others) <= _hip_consts().TRAIN_MAX_SMALL and \
                    all(p.dtype == torch.float32 and p.is_contiguous() and p.is_cuda for p in others):
                H = _hip_consts()
                self.table = tp
                params = [dict(g, params=[p for p in g["params"] if p is not tp]) for g in params]
                self.t_m, self.t_v = torch.zeros_like(tp), torch.zeros_like(tp)
                # every scalar of the optimizer in ONE device buffer (include/lidarnerf_hip.h LNH_TS_*): a captured step
                # needs no host value.  loss_scale / growth_tracker / t_steps are views of it (the names rounds 2-4 used).
                self.opt_state = torch.zeros(H.TRAIN_STATE_FLOATS, dtype=torch.float32, device=tp.device)
                self.opt_state[H.TS_SCALE] = 65536.0
                self.loss_scale = self.opt_state[H.TS_SCALE:H.TS_SCALE + 1].view(())
                self.growth_tracker = self.opt_state[H.TS_GROWTH:H.TS_GROWTH + 1].view(())
                self.t_steps, self.t_flip = [self.opt_state[H.TS_T_NEXT:H.TS_T_NEXT + 1].view(())] * 2, 0
                self._one = torch.ones((), dtype=torch.float32, device=tp.device)
                self._lr0, self._iters = float(lr), float(iters)
                self.small = others
                self.small_off = [0]
                for p in others:
                    self.small_off.append(self.small_off[-1] + p.numel())
                self.small_m = torch.zeros(max(self.small_off[-1], 1), dtype=torch.float32, device=tp.device)
                self.small_v = torch.zeros_like(self.small_m)
                self._small_stepped = set()  # ids of the small parameters that have taken a step (torch creates state lazily)
                tp._lnh_keep_grad16 = True
                tp._lnh_direct_small_grads = True
                tp._lnh_table16 = tp.detach().to(torch.half).reshape(-1, 2).contiguous()
                tp._lnh_table16_version = tp._version  # fused.table16_of re-casts when the parameter is written elsewhere
                # data parallel, second cut (parallel.py): reduce-scatter of the table gradient, every rank steps 1/N of
                # the rows, all-gather of the fp16 compute copy.  The fp32 master table and the Adam moments of a rank are
                # then current on ITS rows only: gather_table_state() completes them (checkpoints call it).
                self.sharded = bool(shard_table_optimizer and self.dp)
                tp._lnh_shard_optimizer = self.sharded
                tp._lnh_master_stale = False
        params = [g for g in params if len(g["params"])]
        # the reference's groups differ in nothing but their parameter lists (network.py get_params: every group at `lr`):
        # step them as ONE group — torch launches its fused Adam once per group — and keep the reference's grouping for
        # the checkpoint layout only (_optimizer_state_ref_layout)
        if len(params) > 1 and all({k: v for k, v in g.items() if k != "params"} ==
                                   {k: v for k, v in params[0].items() if k != "params"} for g in params):
            params = [dict(params[0], params=[p for g in params for p in g["params"]])]
        # graph=True (fused chain + fused table optimizer): the whole step — (march,) render chain, loss, backward, (the
        # gradient exchange,) both optimizers, loss-scale update — is captured in a hipGraph per (batch shape, sample
        # capacity) and replayed (_step_graphed).  Built for occupancy-grid sampling; the dense step (no data-dependent sizes
        # at all) captures the same way and then costs the host 0.05 ms instead of ~0.8 — it is GPU-bound either way on the
        # fast hosts of this build, a slower host is not (3.42 against 2.21 ms).  The step is ~45 launches over
        # ~0.4 M samples: eager, the host cannot issue them as fast as the GPU retires them (profiles/r04_bench_nerfmvl.json:
        # 1.0 ms of host time per 0.7 ms of kernels).  What a capture freezes — kernel arguments — must not change between
        # replays, so the learning rate becomes a device scalar (torch's capturable Adam, lnh_adam_table_step_dlr) and the
        # marcher's sample capacity comes from a ladder of sizes (_graph_capacity; the reference sizes it to the running
        # mean rounded to 128, raymarching.py:223-229: a larger buffer drops fewer rays on overflow, nothing else changes).
        # Data parallel (round 5): under RCCL (backend "nccl") the collectives of the step — the windowed fp16 all-reduce
        # or reduce-scatter / all-gather of the table, the MLP gradients, the found-inf MAX — are captured with it, each on
        # RCCL's own stream inside the graph, so N ranks replay N identical graphs and none of them is host-bound (eight
        # processes share the 16-CPU quota of a box).  gloo cannot be captured (its collectives synchronise with the host).
        self.graph = bool(graph and self.table is not None and on_gpu and (not self.dp or parallel.backend() == "nccl"))
        if graph and not self.graph:
            raise RuntimeError("LidarTrainer(graph=True): the captured step needs the fused chain with the fused table "
                               "optimizer (fp16, a fusable field on the GPU)" +
                               (f" and, data parallel, the 'nccl' (RCCL) backend — this process group runs "
                                f"'{parallel.backend()}', whose collectives cannot be captured" if self.dp else ""))
        self._graphs, self._graph_warm, self._graph_pool, self.graph_error = {}, set(), None, None
        self._capture_stream, self.capture_ms = None, []
        # With the fused optimizer (self.table is not None) this torch optimizer never steps: it holds the parameter groups
        # the scheduler and the checkpoint layout are written against (lr stays a host number; the kernels form the same
        # schedule on the device from their own step counter).
        self.optimizer = torch.optim.Adam(params, betas=(0.9, 0.99), eps=1e-15, fused=on_gpu)
        if self.table is not None:
            self.optimizer._opt_called = True  # (the scheduler's "step() before optimizer.step()" check: the kernels are the optimizer)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lambda it: 0.1 ** min(it / iters, 1))
        self.scaler = torch.amp.GradScaler("cuda", enabled=fp16)
        self.params = [p for g in self.optimizer.param_groups for p in g["params"]]

    @staticmethod
    def _ragged_chain(model):
        """Occupancy-grid sampling renders through the fused ragged chain (nerf/fused.py) when the field has the shapes it is
        built for: the table gradient then arrives in fp16 like the dense chain's; otherwise through the modular density() /
        color() path, where it is a normal .grad and the table stays in torch.optim.Adam."""
        from . import fused
        return bool(getattr(model, "fused_lidar", False)) and fused.ragged_supported(model)

    def loss(self, rays_o, rays_d, images_lidar, patch=(1, 1), grad_scale=None):
        """grad_scale (device scalar): the loss comes back unscaled, its gradient multiplied by it."""
        out = self.model.render(rays_o, rays_d, cal_lidar_color=True, staged=False, perturb=True,
                                **self.render_kwargs)
        ad, ar, ai, ag = self.alpha
        if out["depth_lidar"].is_cuda and (patch[0] <= 1 or (patch[1] >= 2 and out["depth_lidar"].numel() %
                                                               (patch[0] * patch[1]) == 0)):
            return fused_lidar_loss(out, images_lidar, ad, ar, ai,
                                    None if patch[0] <= 1 else (patch[0], patch[1], self.scale, ag), grad_scale)
        loss, pred_depth, gt_depth = lidar_loss(out, images_lidar, ad, ar, ai)
        if patch[0] > 1:
            loss = loss + patch_gradient_loss(pred_depth, gt_depth, images_lidar[..., 0], patch[0], patch[1],
                                              self.scale, ag)
        return loss if grad_scale is None else _ScaleGrad.apply(loss, grad_scale)

    def _forward_backward(self, rays_o, rays_d, images_lidar, patch):
        """Render + loss + backward of one batch with the fused optimizer's conventions: the gradients come out multiplied by
        the CURRENT loss scale — the table's in fp16 (`table._lnh_grad16`), the small tensors' as `.grad` views of one arena.
        Nothing is stepped (tests/test_patch_step_gpu.py compares exactly this state with the oracle)."""
        tp = self.table
        for p in self.small:
            p.grad = None
        tp._lnh_grad16 = None
        tp._lnh_grad_reduced = False
        tp._lnh_grad16_handles = None
        tp._lnh_small_arena = None
        with torch.autocast("cuda", dtype=self.amp_dtype):
            loss = self.loss(rays_o, rays_d, images_lidar, patch, grad_scale=self.loss_scale)
        loss.backward(gradient=self._one)  # (the loss kernel has multiplied its gradients by the loss scale)
        return loss

    def _step_fused_table(self, rays_o, rays_d, images_lidar, patch):
        """One iteration with the fused optimizer: render + loss + backward, (the gradient exchange,) and the optimizer as two
        launches — lnh_train_check (finite check of every gradient, 1 / scale, the learning rate of this step) and
        lnh_train_step (Adam on the table and on the small tensors, GradScaler's skip / scale update, the step counters)."""
        from .. import _hip
        from .fused import table16_of
        H = _hip
        tp, st = self.table, self.opt_state
        loss = self._forward_backward(rays_o, rays_d, images_lidar, patch)
        # --- data parallel: the small gradients.  The fused chain leaves all of them in ONE arena (views): that tensor goes
        # on the wire as it is, summed — the division by the world size is folded into 1 / scale like the table's
        div_small, small_handle = 1.0, None
        if self.dp:
            arena = getattr(tp, "_lnh_small_arena", None)
            if arena is not None:
                import torch.distributed as dist
                small_handle = dist.all_reduce(arena, op=dist.ReduceOp.SUM, async_op=True)
                div_small = float(self.world)
            else:
                parallel.allreduce_gradients(self.small, self.world)
        # data parallel: the fp16 table gradient arrives as the sum over ranks; its mean is taken in fp32 by the kernels
        div = float(getattr(tp, "_lnh_grad16_div", 1))
        shards = getattr(tp, "_lnh_grad16_shards", None) if self.sharded else None
        g16 = tp._lnh_grad16
        if g16 is None and not shards:
            raise RuntimeError("fused table optimizer: the backward pass produced no fp16 table gradient "
                               "(render did not go through the fused LiDAR chain)")
        grads = [p.grad for p in self.small]
        for p, g in zip(self.small, grads):
            if g is not None:
                if not (g.dtype == torch.float32 and g.is_contiguous()):
                    raise RuntimeError("fused optimizer: gradients of the small parameters must be contiguous fp32")
                self._small_stepped.add(id(p))
        n_small = len(self.small)
        gp = H.ptr_array([None if g is None else g.data_ptr() for g in grads])
        pp = H.ptr_array([p.data_ptr() for p in self.small])
        nn_ = H.u32_array([p.numel() for p in self.small])
        cast = lambda arr: H.C.cast(arr, H.C.c_void_p)
        if small_handle is not None:
            small_handle.wait()
        check = lambda ptr, n, first: _hip.call("lnh_train_check", st.data_ptr(), ptr, n, cast(gp), cast(nn_),
                                                n_small if first else 0, div, div_small, self._lr0, self._iters)
        step_args = (cast(pp), cast(gp), cast(nn_), n_small, self.small_m.data_ptr(), self.small_v.data_ptr(), 0.9, 0.99,
                     1e-15, 2.0, 0.5, 2000)
        if shards:
            import torch.distributed as dist
            rank = dist.get_rank()
            for i, (r0, r1, mine, handle, _padded) in enumerate(shards):
                handle.wait()
                rows = max(0, min(mine.shape[0], r1 - (r0 + rank * mine.shape[0])))
                check(mine.data_ptr() if rows else None, rows * 2, i == 0)
            # every rank has looked at its own rows only: the skip / back-off decision must be the same everywhere (the stamp
            # of a step is the same number on every rank, so MAX keeps it)
            dist.all_reduce(st[H.TS_FOUND:H.TS_FOUND + 1], op=dist.ReduceOp.MAX)
            shadow = table16_of(tp)  # (re-cast first if somebody wrote the parameter since the last step)
            _hip.call("lnh_train_step", st.data_ptr(), None, None, None, None, None, 0, *step_args)
            self._step_table_shards(shards, shadow)
        else:
            for handle in getattr(tp, "_lnh_grad16_handles", None) or ():
                handle.wait()  # the table windows' all-reduce (left in flight by the backward pass)
            tp._lnh_grad16_handles = None
            check(g16.data_ptr(), g16.numel(), True)
            shadow = table16_of(tp)  # (re-cast first if somebody wrote the parameter since the last step)
            _hip.call("lnh_train_step", st.data_ptr(), tp.data_ptr(), self.t_m.data_ptr(), self.t_v.data_ptr(), g16.data_ptr(),
                      shadow.data_ptr(), tp.numel(), *step_args)
        if not torch.cuda.is_current_stream_capturing():
            self.scheduler.step()  # (host bookkeeping only: the kernels form the schedule from their own counter)
        return loss

    def steps_taken(self):
        """Number of optimizer steps applied so far (skipped steps — inf / nan gradients — do not count).  Synchronises."""
        return int(self.opt_state[_hip_consts().TS_T_NEXT]) if self.table is not None else None

    def _sync_counters(self, steps=None):
        """After a load: the device-side counters follow the host's (scheduler position; optionally the Adam step count)."""
        H = _hip_consts()
        it = float(self.scheduler.last_epoch)
        self.opt_state[H.TS_IT], self.opt_state[H.TS_IT_NEXT] = it, it
        # the inf / nan stamp is `it + 1` of the step that saw it and relies on `it` only ever growing: after a rewind a stale
        # stamp would match again when training reaches that iteration (a finite step skipped, the loss scale halved)
        self.opt_state[H.TS_FOUND], self.opt_state[H.TS_SKIPPED] = 0.0, 0.0
        if steps is not None:
            self.opt_state[H.TS_T], self.opt_state[H.TS_T_NEXT] = float(steps), float(steps)

    # ---- the captured step (graph=True)
    def _graph_capacity(self):
        """Sample capacity of the marcher for a captured step: the running mean of the recent marches (renderer.py
        update_extra_state) rounded UP to the next of a geometric ladder of capacities (ratio 2^(1/8), multiples of 1024;
        LNH_GRAPH_LADDER = rungs per octave): while the occupancy grid is still settling the mean swings by tens of percent
        from one grid update to the next (measured on the NeRF-MVL-shaped bench: 107 K .. 393 K over 300 steps), and every
        distinct capacity is one capture (0.7 .. 0.9 ms since the trainer captures without emptying the allocator's cache) —
        the ladder has ~15 rungs over that range, each captured once and kept.  On average 4 % of the buffer is padding (zero
        samples the chain runs over; 9 % with the 2^(1/4) ladder of round 4, when a capture cost 70 ms: 0.500 against
        0.516 ms per step at 66 .. 69 samples per ray).  0 while there is no mean yet (the first 16 steps march into N x 1024
        buffers and read the count back)."""
        mc = int(self.model.mean_count)
        if mc <= 0:
            return 0
        import math
        per = _LADDER_RUNGS_PER_OCTAVE
        rung = math.ceil(per * math.log2(max(mc, 1024) / 1024.0) - 1e-9)
        return int(math.ceil(1024 * 2 ** (rung / per) / 1024.0)) * 1024

    def _step_graphed(self, rays_o, rays_d, images_lidar, patch):
        model = self.model
        # (the dense step has no sample buffers to size: one graph per batch shape)
        cap = self._graph_capacity() if self.occupancy else -1
        if cap == 0 or not self._graph_warm:
            # eager: no sample mean yet / the very first step (it takes every lazy initialisation — workspaces, kernel
            # attributes, optimizer state — out of the captures that follow).
            # (detached: a caller holding the loss would keep this step's autograd graph — and its AccumulateGrad nodes,
            #  bound to the eager stream — alive into the capture that follows)
            self._graph_warm.add("eager")
            model._static_march = None
            return self._step_fused_table(rays_o, rays_d, images_lidar, patch).detach()
        # what a capture bakes in as kernel arguments is part of the key: the loss weights, the scene scale, the render
        # arguments (a change of any of them captures a new step instead of silently replaying the old values)
        key = (tuple(rays_o.shape), tuple(images_lidar.shape), tuple(patch), tuple(self.alpha), float(self.scale),
               tuple(sorted((k, repr(v)) for k, v in self.render_kwargs.items())), cap)
        tp = self.table
        if getattr(tp, "_lnh_table16_version", None) != tp._version:
            # somebody wrote the fp32 table through torch since the last step (model.load_state_dict, a manual
            # re-initialisation): a replay never runs table16_of, so the fp16 compute copy is re-cast here — the captured
            # kernels read it in place
            from .fused import table16_of
            table16_of(tp)
        ent = self._graphs.get(key)
        if ent is None:
            dev = self.table.device
            if self._graph_pool is None:
                # one memory pool for all captured steps: they never run concurrently and none reads what another left
                # behind, so a later capture may reuse what an earlier one freed (and no capture after the largest pays
                # for fresh device allocations)
                self._graph_pool = torch.cuda.graph_pool_handle()
            ent = {"rays_o": torch.empty_like(rays_o), "rays_d": torch.empty_like(rays_d),
                   "gt": torch.empty_like(images_lidar), "counter": torch.zeros(2, dtype=torch.int32, device=dev),
                   "graph": torch.cuda.CUDAGraph()}
            for k, src in (("rays_o", rays_o), ("rays_d", rays_d), ("gt", images_lidar)):
                ent[k].copy_(src)
            if self.occupancy:
                model._static_march = (ent["counter"], cap - 128)  # (march_rays_train adds its 128-alignment on top)
            try:
                # capture_begin / capture_end by hand: torch.cuda.graph's context manager empties the allocator's cache
                # first (every cached block back to the driver: the workspaces of this step and of the evaluation pass
                # are re-allocated afterwards), which made a capture cost 70-80 ms — 140 steps of the occupancy-grid
                # workload, whose sample capacity moves to a new rung (a new capture) whenever the grid has changed enough
                torch.cuda.synchronize()
                import time
                if self.dp:
                    # Data parallel: ProcessGroupNCCL's watchdog thread keeps every EAGER collective in a list until one of
                    # its sweeps (every 100 ms) finds the work's end event complete.  RCCL's stream joins the capture below,
                    # and hipEventQuery on an event of a stream that is capturing fails with hipErrorCapturedEvent — in the
                    # watchdog thread, which takes the process down ("failed once in a dozen runs" in round 5, 1 of 50 in
                    # round 6's loop: the chance that a sweep falls into the 1-2 ms of a capture; with the capture stalled for
                    # 300 ms it is every run, profiles/r06_rccl_loop.txt).  The collectives of the eager steps have finished
                    # (the synchronize above): give the watchdog two sweeps to drop them, so that it has nothing to poll while
                    # this thread captures.  Collectives issued DURING a capture are never put on that list.
                    time.sleep(float(os.environ.get("LNH_CAPTURE_DRAIN_MS", "250")) * 1e-3)
                t_cap = time.perf_counter()
                if self._capture_stream is None:
                    self._capture_stream = torch.cuda.Stream()
                with torch.cuda.stream(self._capture_stream):
                    # (a capture that polices every thread of the process would trip over the watchdog's other HIP calls)
                    ent["graph"].capture_begin(self._graph_pool,
                                               capture_error_mode="thread_local" if self.dp else "global")
                    try:
                        ent["loss"] = self._step_fused_table(ent["rays_o"], ent["rays_d"], ent["gt"], patch).detach()
                        if os.environ.get("LNH_DEBUG_CAPTURE_STALL_MS"):  # (diagnosis only: widens the window above)
                            time.sleep(float(os.environ["LNH_DEBUG_CAPTURE_STALL_MS"]) * 1e-3)
                    finally:
                        ent["graph"].capture_end()
                # the gradient and the scale it carries live in THIS graph's buffers: table_grad() must see the ones of
                # the graph that was replayed last, not of the one that was captured last
                ent["g16"] = tp._lnh_grad16
                self.capture_ms.append(round((time.perf_counter() - t_cap) * 1e3, 2))  # (host time of this capture)
            except Exception as e:  # noqa: BLE001 — a capture that does not go through must not cost the run
                # (nothing of a captured step has executed: the state is what it was.)  Launch by launch from here on; the
                # reason stays readable (bench.py reports it).
                model._static_march = None
                self.graph, self.graph_error = False, f"{type(e).__name__}: {e}"
                return self._step_fused_table(rays_o, rays_d, images_lidar, patch).detach()
            finally:
                model._static_march = None
            try:
                ent["graph"].replay()
            except Exception as e:  # noqa: BLE001 — a graph the runtime captured and then refuses to launch (same rule)
                self.graph, self.graph_error = False, f"replay: {type(e).__name__}: {e}"
                return self._step_fused_table(rays_o, rays_d, images_lidar, patch).detach()
            self._graphs[key] = ent
        else:
            torch._foreach_copy_([ent["rays_o"], ent["rays_d"], ent["gt"]], [rays_o, rays_d, images_lidar])  # one launch
            ent["graph"].replay()
        tp._lnh_grad16 = ent["g16"]
        if self.occupancy:
            model.step_counter[model.local_step % 16].copy_(ent["counter"])
            model.local_step += 1
        self.scheduler.step()
        return ent["loss"].clone()  # (the graphs share a pool: the next replay of another one may reuse this memory)

    def _step_table_shards(self, shards, shadow):
        """Sharded table optimizer: Adam on this rank's rows of every level window (learning rate, 1 / scale, the skip flag
        and the step counter read from the optimizer's device scalars, which lnh_train_check / lnh_train_step have set), then
        the all-gather of the fp16 compute copy (the only part of the table the next forward pass reads)."""
        from .. import _hip
        import torch.distributed as dist
        H = _hip
        tp, st = self.table, self.opt_state
        sp = lambda i: st.data_ptr() + 4 * i
        rank = dist.get_rank()
        flat16, table16 = shadow.view(-1), shadow.view(-1, 2)
        gathers = []
        for r0, r1, mine, _h, padded in shards:
            s = mine.shape[0]
            row0 = r0 + rank * s
            rows = max(0, min(s, r1 - row0))
            if rows:
                o = row0 * 2  # element offset of the shard in the [rows, 2] table
                _hip.call("lnh_adam_table_step_dlr", tp.data_ptr() + 4 * o, self.t_m.data_ptr() + 4 * o,
                          self.t_v.data_ptr() + 4 * o, mine.data_ptr(), flat16.data_ptr() + 2 * o, rows * 2, sp(H.TS_LR), 0.9,
                          0.99, 1e-15, sp(H.TS_INV_TABLE), sp(H.TS_SKIPPED), sp(H.TS_T), sp(H.TS_T_NEXT))
            mine16 = torch.zeros_like(mine)
            if rows:
                mine16[:rows] = table16[row0:row0 + rows]
            # (the window's padded gradient buffer has served its purpose: it receives the gathered copy)
            gathers.append((dist.all_gather_into_tensor(padded, mine16, async_op=True), padded, r0, r1))
        for handle, out, r0, r1 in gathers:
            handle.wait()
            table16[r0:r1] = out[:r1 - r0]  # the shards back to back; what lies beyond r1 is padding
        tp._lnh_grad16_shards = None
        # from here on the fp32 master (and the moments) of this rank are current on ITS rows only: whoever reads
        # `embeddings` itself (GridEncoder.forward, grad_total_variation, a state_dict) must gather_table_state() first
        tp._lnh_master_stale = True

    def gather_table_state(self):
        """Sharded table optimizer: complete the fp32 master table and the Adam moments on every rank from their owners
        (checkpoints, evaluation through `embeddings`, a switch back to the replicated optimizer).  Collective."""
        if not self.sharded:
            return
        import torch.distributed as dist
        from .fused import _DP_LEVEL_WINDOWS
        enc = self.model.fused_spec().grid
        off, world, rank = enc._offsets_host, dist.get_world_size(), dist.get_rank()
        windows = _DP_LEVEL_WINDOWS if enc.num_levels == 16 else ((0, enc.num_levels),)
        for t in (self.table.data, self.t_m, self.t_v):
            rows = t.view(-1, 2)
            for l0, l1 in windows:
                r0, r1 = int(off[l0]), int(off[l1])
                s = parallel.shard_rows(r1 - r0, world)
                mine = torch.zeros((s, 2), dtype=t.dtype, device=t.device)
                a = r0 + rank * s
                n = max(0, min(s, r1 - a))
                if n:
                    mine[:n] = rows[a:a + n]
                full = torch.empty((world * s, 2), dtype=t.dtype, device=t.device)
                dist.all_gather_into_tensor(full, mine)
                rows[r0:r1] = full[:r1 - r0]
        self.table._lnh_master_stale = False

    # ---- what lives outside torch.optim / GradScaler when the table is stepped by the fused kernel
    def table_grad(self):
        """fp32, unscaled gradient of the hash table of the LAST step (the fused path keeps it in fp16 and never sets
        `.grad` on the parameter: code that wants `embeddings.grad` — gradient clipping, `grad_total_variation` — asks
        here).  None before the first step or without the fused table optimizer."""
        g16 = getattr(self.table, "_lnh_grad16", None) if self.table is not None else None
        if g16 is None:
            return None
        for handle in getattr(self.table, "_lnh_grad16_handles", None) or ():
            handle.wait()
        div = float(getattr(self.table, "_lnh_grad16_div", 1))
        # (the scale the backward ran with — the optimizer kernel has already moved loss_scale on growth / backoff steps)
        H = _hip_consts()
        return g16.float().reshape(self.table.shape) / (self.opt_state[H.TS_LAST_SCALE] * div)

    def state_dict(self):
        """Everything a resume needs: torch optimizer / scheduler / scaler state plus — fused table optimizer — the
        table's Adam moments, its device-side step counter and the dynamic loss scale (99.8 % of the optimizer state).
        Sharded table optimizer: COLLECTIVE (gather_table_state) — call it on every rank."""
        self.gather_table_state()
        sd = {"optimizer": self.optimizer.state_dict(), "scheduler": self.scheduler.state_dict(),
              "scaler": self.scaler.state_dict(), "fused_table": None}
        if self.table is not None:
            sd["fused_table"] = {"exp_avg": self.t_m, "exp_avg_sq": self.t_v, "step": self.t_steps[self.t_flip].clone(),
                                 "loss_scale": self.loss_scale.clone(), "growth_tracker": self.growth_tracker.clone(),
                                 # the small tensors' moments, back to back in the order of self.small
                                 "small_exp_avg": self.small_m, "small_exp_avg_sq": self.small_v,
                                 "small_stepped": [id(p) in self._small_stepped for p in self.small]}
        return sd

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd["optimizer"])
        self._after_optimizer_load()
        self.scheduler.load_state_dict(sd["scheduler"])
        self.scaler.load_state_dict(sd["scaler"])
        self._pending_steps = None
        ft = sd.get("fused_table")
        if (ft is None) != (self.table is None):
            raise RuntimeError("LidarTrainer.load_state_dict: checkpoint and trainer disagree on the fused table optimizer")
        if ft is not None:
            self.t_m.copy_(ft["exp_avg"])
            self.t_v.copy_(ft["exp_avg_sq"])
            self.loss_scale.copy_(ft["loss_scale"])
            self.growth_tracker.copy_(ft["growth_tracker"])
            if "small_exp_avg" in ft:
                self.small_m.copy_(ft["small_exp_avg"])
                self.small_v.copy_(ft["small_exp_avg_sq"])
                self._small_stepped = {id(p) for p, f in zip(self.small, ft["small_stepped"]) if f}
            self._sync_counters(steps=float(ft["step"]))

    # ---- checkpoints in the reference Trainer's format (lidarnerf/nerf/utils.py:1449-1568)
    def _optimizer_state_ref_layout(self):
        """torch.optim.Adam.state_dict() as the reference's optimizer would write it: one entry per parameter of
        model.get_params(lr) in order, the fused table optimizer's moments / step count included."""
        own = self.optimizer.state_dict()
        own_ids = {id(p): i for i, p in enumerate(p for g in self.optimizer.param_groups for p in g["params"])}
        template = {k: v for k, v in own["param_groups"][0].items() if k != "params"} if own["param_groups"] else {}
        # (graph mode keeps lr in a device scalar: the file carries the number, as the reference's does)
        template = {k: (float(v) if torch.is_tensor(v) and v.dim() == 0 else v) for k, v in template.items()}
        state, groups, idx = {}, [], 0
        for gi, group in enumerate(self._ref_layout):
            ids = []
            for p in group:
                if self.table is not None and p is self.table:
                    state[idx] = {"step": self.t_steps[self.t_flip].detach().clone().float().cpu(),
                                  "exp_avg": self.t_m.detach().clone(), "exp_avg_sq": self.t_v.detach().clone()}
                elif self.table is not None and id(p) in self._small_stepped:
                    # stepped by the fused optimizer: its moments are a slice of the flat buffers, the step count is the
                    # table's (one counter for all parameters: a skipped step skips every one of them)
                    k = next(i for i, q in enumerate(self.small) if q is p)
                    sl = slice(self.small_off[k], self.small_off[k + 1])
                    state[idx] = {"step": self.t_steps[self.t_flip].detach().clone().float().cpu(),
                                  "exp_avg": self.small_m[sl].detach().clone().view_as(p),
                                  "exp_avg_sq": self.small_v[sl].detach().clone().view_as(p)}
                elif id(p) in own_ids and own_ids[id(p)] in own["state"]:
                    state[idx] = own["state"][own_ids[id(p)]]
                ids.append(idx)
                idx += 1
            g = dict(template)
            g["params"] = ids
            groups.append(g)
        return {"state": state, "param_groups": groups}

    def _load_optimizer_state_ref_layout(self, sd):
        own_ids = {id(p): i for i, p in enumerate(p for g in self.optimizer.param_groups for p in g["params"])}
        own = self.optimizer.state_dict()
        idx, loaded_steps = 0, None
        for group in self._ref_layout:
            for p in group:
                st = sd["state"].get(idx)
                if st is not None:
                    if self.table is not None and p is self.table:
                        self.t_m.copy_(st["exp_avg"].to(self.t_m.device))
                        self.t_v.copy_(st["exp_avg_sq"].to(self.t_v.device))
                        loaded_steps = float(st["step"])
                    elif self.table is not None and any(q is p for q in self.small):
                        k = next(i for i, q in enumerate(self.small) if q is p)
                        sl = slice(self.small_off[k], self.small_off[k + 1])
                        self.small_m[sl].copy_(st["exp_avg"].to(self.small_m.device).reshape(-1))
                        self.small_v[sl].copy_(st["exp_avg_sq"].to(self.small_v.device).reshape(-1))
                        self._small_stepped.add(id(p))
                        if loaded_steps is None:
                            loaded_steps = float(st["step"])
                    elif id(p) in own_ids:
                        own["state"][own_ids[id(p)]] = st
                idx += 1
        # learning rates: the reference's groups that hold parameters stepped here, in order; when they are stepped as
        # one merged group (see __init__) they all carry the same value and the first one is taken
        lr_by_pos = [g.get("lr") for g in sd["param_groups"]]
        lrs = [l for l, grp in zip(lr_by_pos, self._ref_layout) if any(id(p) in own_ids for p in grp)]
        if len(own["param_groups"]) == 1:
            lrs = lrs[:1]
        for g, lr in zip(own["param_groups"], lrs):
            if lr is not None:
                g["lr"] = lr
        self.optimizer.load_state_dict(own)
        self._after_optimizer_load()
        if self.table is not None and loaded_steps is not None:
            self._pending_steps = loaded_steps  # (committed by load_checkpoint once the scheduler's position is loaded too)

    def _drop_graphs(self):
        """Forget every captured step: the next step runs launch by launch (taking every lazy initialisation and version
        check with it), the one after is captured afresh."""
        had = bool(self._graphs)
        self._graphs.clear()
        self._graph_warm.clear()
        # the graphs' memory pool goes with them: a pool none of whose graphs is alive any more cannot take a new capture
        # (the allocator asserts on it); the next capture opens a new one, and the blocks of the old one go back to the driver
        self._graph_pool = None
        if had and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            torch.cuda.empty_cache()

    def _after_optimizer_load(self):
        """Graph mode: every captured step is dropped after a load (the next step runs launch by launch and takes the
        version checks and lazy initialisations with it)."""
        if self.graph:
            self._drop_graphs()

    def _own_group_of_ref_group(self):
        """For every parameter group of the reference's optimizer: index of the group of self.optimizer that steps its
        parameters (the table's group, stepped by the fused kernel, and empty groups follow group 0: every group of
        model.get_params(lr) carries the same lr and the same lambda)."""
        own = {id(p): gi for gi, g in enumerate(self.optimizer.param_groups) for p in g["params"]}
        return [next((own[id(p)] for p in grp if id(p) in own), 0) for grp in self._ref_layout]

    def _scheduler_state_ref_layout(self):
        """LambdaLR.state_dict() as the reference's scheduler over Adam(model.get_params(lr)) writes it: `base_lrs`,
        `_last_lr` and `lr_lambdas` carry one entry per REFERENCE parameter group (6, or 8 with a background net), not per
        group of the merged optimizer stepped here — a stock scheduler loading the file zips them against its groups."""
        sd = dict(self.scheduler.state_dict())
        m = self._own_group_of_ref_group()
        for key in ("base_lrs", "_last_lr"):
            if key in sd:
                sd[key] = [float(sd[key][i]) if torch.is_tensor(sd[key][i]) else sd[key][i] for i in m]
        if "lr_lambdas" in sd:
            sd["lr_lambdas"] = [sd["lr_lambdas"][i] for i in m]
        return sd

    def _load_scheduler_state_ref_layout(self, sd):
        """Inverse of the above (also accepts a state written per own group, e.g. by LidarTrainer.state_dict)."""
        sd = dict(sd)
        n_own, m = len(self.optimizer.param_groups), self._own_group_of_ref_group()
        for key in ("base_lrs", "_last_lr", "lr_lambdas"):
            vals = sd.get(key)
            if vals is None or len(vals) == n_own:
                continue
            if len(vals) != len(m):
                raise RuntimeError(f"lr_scheduler state: {len(vals)} entries in '{key}' for {len(m)} reference parameter "
                                   f"groups / {n_own} groups stepped here")
            first = {}
            for ref_i, own_i in enumerate(m):
                first.setdefault(own_i, vals[ref_i])
            sd[key] = [first.get(i, vals[0]) for i in range(n_own)]
        self.scheduler.load_state_dict(sd)

    def save_checkpoint(self, path, full=True, gather=True):
        """Same dictionary as Trainer.save_checkpoint (utils.py:1449-1480): epoch, global_step, stats, model and — `full`
        — optimizer / lr_scheduler / scaler in the layout the reference's Trainer.load_checkpoint restores (a reference
        run can resume from it and vice versa: the state dict keys of the model are the reference's, see network.py).

        Sharded table optimizer (shard_table_optimizer=True): the master table and the moments live in pieces on the
        ranks, so completing them is a COLLECTIVE.  Two ways to write a checkpoint:
          * call save_checkpoint(path) on EVERY rank — all of them gather, rank 0 alone writes the file (the others
            return the path without touching it); a call on rank 0 only would wait for the others forever;
          * the reference's convention, save on local_rank 0 only (utils.py:1069-1074): call gather_table_state() on every
            rank first, then save_checkpoint(path, gather=False) where the reference saves.  Without the gather that
            raises instead of writing a table with other ranks' stale rows."""
        write = True
        if self.sharded:
            import torch.distributed as dist
            if gather:
                self.gather_table_state()
                write = dist.get_rank() == 0
            elif getattr(self.table, "_lnh_master_stale", False):
                raise RuntimeError("save_checkpoint(gather=False) with the sharded table optimizer: call "
                                   "gather_table_state() on every rank first (this rank holds only its own rows of the "
                                   "table and the Adam moments)")
        if not write:
            return path
        state = {"epoch": self.epoch, "global_step": self.global_step, "stats": self.stats}
        if full:
            state["optimizer"] = self._optimizer_state_ref_layout()
            state["lr_scheduler"] = self._scheduler_state_ref_layout()
            if self.table is not None:  # the dynamic loss scale lives with the fused table optimizer
                state["scaler"] = {"scale": float(self.loss_scale), "growth_factor": 2.0, "backoff_factor": 0.5,
                                   "growth_interval": 2000, "_growth_tracker": int(self.growth_tracker)}
            else:
                state["scaler"] = self.scaler.state_dict()
        if getattr(self.model, "cuda_ray", False):  # a reference loader ignores the extra keys
            state["mean_count"], state["mean_density"] = self.model.mean_count, self.model.mean_density
            state["iter_density"], state["local_step"] = self.model.iter_density, self.model.local_step
        state["model"] = self.model.state_dict()
        torch.save(state, path)
        return path

    def load_checkpoint(self, path, model_only=False):
        """Trainer.load_checkpoint (utils.py:1511-1568): a bare state dict or the dictionary above; strict=False."""
        ck = torch.load(path, map_location=next(self.model.parameters()).device, weights_only=False)
        # whatever the file holds, the model is about to change under the captured steps (a bare state dict and
        # model_only=True never reach _after_optimizer_load): drop them on every path
        self._drop_graphs()
        if "model" not in ck:
            self.model.load_state_dict(ck)
            return [], []
        missing, unexpected = self.model.load_state_dict(ck["model"], strict=False)
        if getattr(self.model, "cuda_ray", False):
            for key in ("mean_count", "mean_density", "iter_density", "local_step"):
                if key in ck:  # without them the next 16 grid updates are full sweeps and sample buffers are N * 1024
                    setattr(self.model, key, ck[key])
        if self.table is not None:
            self.table._lnh_master_stale = False  # a loaded table is whole
        if model_only:
            return missing, unexpected
        self.stats, self.epoch, self.global_step = ck["stats"], ck["epoch"], ck["global_step"]
        if "optimizer" in ck:
            self._load_optimizer_state_ref_layout(ck["optimizer"])
        if "lr_scheduler" in ck:
            self._load_scheduler_state_ref_layout(ck["lr_scheduler"])
        if self.table is not None:  # the device-side counters follow what was loaded
            self._sync_counters(steps=getattr(self, "_pending_steps", None))
            self._pending_steps = None
        if "scaler" in ck and ck["scaler"]:
            if self.table is not None:
                self.loss_scale.fill_(float(ck["scaler"]["scale"]))
                self.growth_tracker.fill_(int(ck["scaler"].get("_growth_tracker", 0)))
            else:
                self.scaler.load_state_dict(ck["scaler"])
        return missing, unexpected

    def step(self, rays_o, rays_d, images_lidar, patch=(1, 1)):
        if self.occupancy and self.global_step % self.update_extra_interval == 0:
            with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.fp16):
                self.model.update_extra_state()  # refresh the occupancy grid the marcher reads (every 16 steps)
        self.global_step += 1
        if self.graph:
            return self._step_graphed(rays_o, rays_d, images_lidar, patch)
        if self.table is not None:
            return self._step_fused_table(rays_o, rays_d, images_lidar, patch)
        self.optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.fp16):
            loss = self.loss(rays_o, rays_d, images_lidar, patch)
        self.scaler.scale(loss).backward()
        if self.dp:
            parallel.allreduce_gradients(self.params, self.world)
        self.scaler.step(self.optimizer)
        self.scaler.update()
        self.scheduler.step()
        return loss
