"""NeRFRenderer — ray sampling, importance resampling and volumetric compositing on MI355X.

Same public surface as the reference (lidarnerf/nerf/renderer.py:61-345): constructor arguments, `aabb_train` /
`aabb_infer` buffers, `run(...)`, `render(..., staged, max_ray_batch)`, result keys `depth_lidar`, `image_lidar`,
`weights_sum_lidar`; subclasses provide `density(x)` and `color(x, d, mask=..., geo_feat=...)`.
What the reference does with ~40 PyTorch launches per call (cumprod weights, sample_pdf, sort, gathers, weighted
sums) runs here in three wave-per-ray kernels: lnh_lidar_resample, lnh_lidar_weights, lnh_lidar_composite_*.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _hip, raymarching


def lidar_weights(z, sigma, sample_dist, density_scale=1.0):
    """w[N,T] of renderer.py:233-243 (no grad)."""
    N, T = z.shape
    w = torch.empty((N, T), dtype=torch.float32, device=z.device)
    _hip.call("lnh_lidar_weights", z.data_ptr(), sigma.data_ptr(), sample_dist.data_ptr(), N, T, float(density_scale),
              w.data_ptr())
    return w


def lidar_resample(z, sigma, sample_dist, u, density_scale=1.0):
    """renderer.py:180-231: (new_z [N,n], merged z [N,T+n], perm [N,T+n] int32 into concat([old, new]))."""
    N, T = z.shape
    n_new = u.shape[1]
    new_z = torch.empty((N, n_new), dtype=torch.float32, device=z.device)
    z_out = torch.empty((N, T + n_new), dtype=torch.float32, device=z.device)
    perm = torch.empty((N, T + n_new), dtype=torch.int32, device=z.device)
    _hip.call("lnh_lidar_resample", z.data_ptr(), sigma.data_ptr(), sample_dist.data_ptr(), u.data_ptr(), N, T, n_new,
              float(density_scale), 0, new_z.data_ptr(), z_out.data_ptr(), perm.data_ptr())
    return new_z, z_out, perm


class _LidarComposite(Function):
    """(sigma [N,T], rgb [N,T,K]) -> (weights_sum [N], depth [N], image [N,K]); z / sample_dist carry no grad."""

    @staticmethod
    def forward(ctx, sigma, rgb, z, sample_dist, density_scale):
        sigma, rgb = sigma.contiguous().float(), rgb.contiguous().float()
        N, T = z.shape
        K = rgb.shape[-1]
        ws = torch.empty(N, dtype=torch.float32, device=z.device)
        depth = torch.empty(N, dtype=torch.float32, device=z.device)
        image = torch.empty((N, K), dtype=torch.float32, device=z.device)
        _hip.call("lnh_lidar_composite_forward", z.data_ptr(), sigma.data_ptr(), rgb.data_ptr(),
                  sample_dist.data_ptr(), N, T, K, float(density_scale), None, ws.data_ptr(), depth.data_ptr(),
                  image.data_ptr())
        ctx.save_for_backward(sigma, rgb, z, sample_dist)
        ctx.density_scale = density_scale
        return ws, depth, image

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image):
        sigma, rgb, z, sample_dist = ctx.saved_tensors
        N, T = z.shape
        K = rgb.shape[-1]
        gs = torch.empty_like(sigma)
        gc = torch.empty_like(rgb)
        _hip.call("lnh_lidar_composite_backward", g_ws.contiguous().data_ptr(), g_depth.contiguous().data_ptr(),
                  g_image.contiguous().data_ptr(), z.data_ptr(), sigma.data_ptr(), rgb.data_ptr(),
                  sample_dist.data_ptr(), N, T, K, float(ctx.density_scale), gs.data_ptr(), gc.data_ptr())
        return gs, gc, None, None, None


lidar_composite = _LidarComposite.apply


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, density_scale=1, min_near=0.2, min_near_lidar=0.2, density_thresh=0.01, bg_radius=-1,
                 cuda_ray=False):
        super().__init__()
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.min_near_lidar = min_near_lidar
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        box = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb_train", box)
        self.register_buffer("aabb_infer", box.clone())
        # Occupancy-grid sampling (BASELINE config 4).  The reference kept the constructor arguments (`density_thresh`,
        # cascade / grid_size above) and the raymarching kernels of torch-ngp but dropped the caller (SURVEY.md §0.6:
        # no density_grid / run_cuda / update_extra_state in renderer.py); this is that caller, for the LiDAR outputs.
        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros(self.cascade, self.grid_size ** 3))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))  # 16 most recent marches
            self.mean_density, self.iter_density, self.mean_count, self.local_step = 0.0, 0, 0, 0

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def color(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    # -- sampling helpers ------------------------------------------------------------------------------------
    def _ray_bounds(self, rays_o, rays_d, aabb, cal_lidar_color):
        N = rays_o.shape[0]
        if cal_lidar_color:  # hard-coded 1 m .. 81 m in scene units (renderer.py:129-138)
            nears = torch.full((N,), float(self.min_near_lidar), dtype=rays_o.dtype, device=rays_o.device)
            return nears, nears * 81.0
        return raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)

    def run(self, rays_o, rays_d, cal_lidar_color=False, num_steps=128, upsample_steps=128, bg_color=None,
            perturb=False, noise=None, u=None, **kwargs):
        """renderer.py:99-298.  `noise` [N, num_steps] / `u` [N, upsample_steps] in [0,1) replace the two torch.rand
        draws of the reference (perturbation, sample_pdf) — parity tests replay the reference's own draws through them;
        the reference's **kwargs swallows both names, so passing them there is harmless."""
        self.out_dim = self.out_lidar_color_dim if cal_lidar_color else self.out_color_dim
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, dev = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer

        nears, fars = self._ray_bounds(rays_o, rays_d, aabb, cal_lidar_color)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
        z = nears + (fars - nears) * torch.linspace(0.0, 1.0, num_steps, device=dev).unsqueeze(0)
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z = z + ((torch.rand(z.shape, device=dev) if noise is None else noise.to(dev).view(z.shape)) - 0.5) * sample_dist
        z = z.contiguous()
        sd = sample_dist.reshape(-1).contiguous()

        def positions(zz):
            p = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * zz.unsqueeze(-1)
            return torch.min(torch.max(p, aabb[:3]), aabb[3:])

        xyzs = positions(z)
        dens = self.density(xyzs.reshape(-1, 3))
        sigma = dens["sigma"].view(N, num_steps)
        extras = {k: v.view(N, num_steps, -1) for k, v in dens.items() if k != "sigma"}

        if upsample_steps > 0:
            with torch.no_grad():
                if u is not None:
                    u = u.to(dev).view(N, upsample_steps).float().contiguous()
                elif self.training:
                    u = torch.rand((N, upsample_steps), device=dev)
                else:
                    u = torch.linspace(0.5 / upsample_steps, 1.0 - 0.5 / upsample_steps, upsample_steps,
                                       device=dev).expand(N, upsample_steps).contiguous()
                new_z, z_all, perm = lidar_resample(z, sigma.detach().float().contiguous(), sd, u, self.density_scale)
                new_xyzs = positions(new_z)
                index = perm.long()
            new_dens = self.density(new_xyzs.reshape(-1, 3))
            sigma = torch.gather(torch.cat([sigma, new_dens["sigma"].view(N, upsample_steps)], dim=1), 1, index)
            xyzs = torch.gather(torch.cat([xyzs, new_xyzs], dim=1), 1, index.unsqueeze(-1).expand(-1, -1, 3))
            for k in extras:
                both = torch.cat([extras[k], new_dens[k].view(N, upsample_steps, -1)], dim=1)
                extras[k] = torch.gather(both, 1, index.unsqueeze(-1).expand_as(both))
            z = z_all

        T = z.shape[1]
        with torch.no_grad():
            weights = lidar_weights(z, sigma.detach().float().contiguous(), sd, self.density_scale)
            mask = weights > 1e-4  # hard coded in the reference (renderer.py:249)
        dirs = rays_d.view(-1, 1, 3).expand(N, T, 3)
        flat_extras = {k: v.reshape(N * T, -1) for k, v in extras.items()}
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), cal_lidar_color=cal_lidar_color,
                          mask=mask.reshape(-1), **flat_extras)
        rgbs = rgbs.view(N, T, self.out_dim)

        weights_sum, depth, image = lidar_composite(sigma, rgbs, z, sd, self.density_scale)

        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(sph, rays_d.reshape(-1, 3))
        elif bg_color is None:
            bg_color = 1
        if not cal_lidar_color:
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color

        return {"depth_lidar": depth.view(*prefix), "image_lidar": image.view(*prefix, self.out_dim),
                "weights_sum_lidar": weights_sum}

    # -- occupancy-grid sampling ------------------------------------------------------------------------------
    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.step_counter.zero_()
        self.mean_density, self.iter_density, self.mean_count, self.local_step = 0.0, 0, 0, 0

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """Refresh the occupancy grid from the current density field (call every ~16 training steps): sample one
        jittered point per grid cell (every cell for the first 16 updates, then a random quarter of the cells plus a
        quarter's worth of currently occupied ones), grid = max(grid * decay, density), threshold at
        min(mean density, density_thresh) and pack to the bitfield the marcher reads.  Cell <-> flat index is the Morton
        code of raymarching.cu:71-95 (lnh_morton3D / lnh_morton3D_invert, bit-exact)."""
        if not self.cuda_ray:
            return
        dev, G = self.density_grid.device, self.grid_size
        tmp = -torch.ones_like(self.density_grid)

        def splat(cas, coords, indices):
            bound = min(2 ** cas, self.bound)
            half = bound / G
            xyzs = (2 * coords.float() / (G - 1) - 1) * (bound - half)
            xyzs = xyzs + (torch.rand_like(xyzs) * 2 - 1) * half
            sig = self.density(xyzs)["sigma"].reshape(-1).detach().float() * self.density_scale
            tmp[cas, indices] = sig

        if self.iter_density < 16:  # full sweep, S^3 cells at a time
            ar = torch.arange(G, dtype=torch.int32, device=dev)
            for xs in ar.split(S):
                for ys in ar.split(S):
                    for zs in ar.split(S):
                        xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                        coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).contiguous()
                        indices = raymarching.morton3D(coords).long()
                        for cas in range(self.cascade):
                            splat(cas, coords, indices)
        else:
            n = G ** 3 // 4
            for cas in range(self.cascade):
                coords = torch.randint(0, G, (n, 3), dtype=torch.int32, device=dev)
                indices = raymarching.morton3D(coords).long()
                occ = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                if occ.numel() > 0:
                    occ = occ[torch.randint(0, occ.shape[0], (n,), device=dev)]
                    coords = torch.cat([coords, raymarching.morton3D_invert(occ.int())], 0)
                    indices = torch.cat([indices, occ], 0)
                splat(cas, coords.contiguous(), indices)
        valid = (self.density_grid >= 0) & (tmp >= 0)
        # (renderer.py:517-519 writes this as a masked assignment; the same values as one element-wise pass — boolean-mask
        #  indexing costs two compactions, two gathers and a scatter over the 2 M cells)
        self.density_grid.copy_(torch.where(valid, torch.maximum(self.density_grid * decay, tmp), self.density_grid))
        self.mean_density = float(self.density_grid.clamp(min=0).mean())
        self.iter_density += 1
        raymarching.packbits(self.density_grid, min(self.mean_density, self.density_thresh), self.density_bitfield)
        steps = min(16, self.local_step)
        if steps > 0:  # average number of samples of the recent marches -> size of the next sample buffers
            self.mean_count = int(self.step_counter[:steps, 0].sum().item() / steps)
        self.local_step = 0

    def run_cuda(self, rays_o, rays_d, cal_lidar_color=True, dt_gamma=0, perturb=False, force_all_rays=False,
                 max_steps=1024, T_thresh=1e-4, **kwargs):
        """Occupancy-grid render of LiDAR rays: march through the occupied cells between 1 m and 81 m (scene units,
        renderer.py:129-138) with lnh_march_rays_train, evaluate density + LiDAR colour on the ragged samples, composite
        with the K = 2 / absolute-depth kernel.  Training mode keeps autograd; evaluation marches the same way (all rays)
        under no_grad.  Same result keys as run()."""
        if not cal_lidar_color:
            raise NotImplementedError("occupancy-grid rendering is built for the LiDAR outputs (cal_lidar_color=True)")
        self.out_dim = self.out_lidar_color_dim
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        _hip.require_cuda(rays_o, rays_d)
        dev = rays_o.device
        static = getattr(self, "_static_march", None)
        if self.training and static is not None:
            # LidarTrainer's captured step (train_step.py, graph=True): a fixed counter buffer and a fixed sample capacity —
            # this Python runs at capture only; the trainer copies the counter into the ring and advances local_step per replay
            counter, mean_count = static
        elif self.training:
            counter = self.step_counter[self.local_step % 16]
            self.local_step += 1
            mean_count = self.mean_count
        else:
            counter, mean_count, force_all_rays = None, -1, True
        # ONE launch in front of the marcher (lnh_lidar_march_prologue): the LiDAR range of every ray — 1 m .. 81 m, cut at the
        # ray's exit from the box: the marcher clamps sample POSITIONS to the box, and the compositing kernel recovers a
        # sample's depth from its position ((xyz - o) . d) — a sample marched past the box would enter the depth sum with a
        # shortened z (the dense path keeps the true z next to the clamped position, renderer.py:164-167) — and the clearing
        # of the marcher's zero-initialised sample buffers, of its counter and of the colour buffer the fused chain fills
        # (rows no ray owns stay 0).  Before: a fill, a multiply, lnh_near_far_from_aabb, a minimum and two more fills — five
        # launches of ~5 us each in a step of 0.5 ms.
        aabb = (self.aabb_train if self.training else self.aabb_infer).contiguous().float()
        M = raymarching.march_capacity(N, max_steps, mean_count, 128, force_all_rays)
        nears = torch.empty(N, dtype=torch.float32, device=dev)
        fars = torch.empty(N, dtype=torch.float32, device=dev)
        # the two colour columns are the fused ragged chain's (it fills the rows rays own, the others stay 0): evaluation and
        # the modular path march into M * 8 — with force_all_rays M is N * 1024, and a full LiDAR frame's extra quarter was
        # hundreds of MB of zero fill per render call
        from . import fused
        use_fused = (getattr(self, "fused_lidar", False) and rays_o.is_cuda and torch.is_autocast_enabled()
                     and fused.ragged_supported(self))
        cols = 10 if use_fused and torch.is_grad_enabled() else 8
        buf = torch.empty(M * cols, dtype=torch.float32, device=dev)  # xyzs 3 | dirs 3 | deltas 2 (| lidar colour 2)
        regions = [buf] + ([counter] if counter is not None else [])
        regions = [t for t in regions if t.numel()]
        zp = (_hip.C.c_void_p * max(len(regions), 1))(*[t.data_ptr() for t in regions])
        zb = (_hip.C.c_uint64 * max(len(regions), 1))(*[t.numel() * t.element_size() for t in regions])
        _hip.call("lnh_lidar_march_prologue", rays_o.data_ptr(), rays_d.data_ptr(), aabb.data_ptr(), N,
                  float(self.min_near_lidar), 81.0, nears.data_ptr(), fars.data_ptr(), _hip.C.cast(zp, _hip.C.c_void_p),
                  _hip.C.cast(zb, _hip.C.c_void_p), len(regions))
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(
            rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size, nears, fars, counter,
            mean_count, perturb, 128, force_all_rays, dt_gamma, max_steps, sample_buffer=buf)
        # the cleared colour rows of the samples that were kept (evaluation trims the buffers to the marched count)
        self._lnh_rgb_rows = buf[M * 8:M * 8 + 2 * xyzs.shape[0]].view(-1, 2) if cols == 10 else None
        if xyzs.shape[0] == 0 and not (use_fused and torch.is_grad_enabled()):
            # no sample on any ray (evaluation, or the modular path): plain zeros.  The fused training chain below runs its
            # node on zero samples instead — zeros connected to the graph, zero gradients, and under data parallel the same
            # collectives as the other ranks (a rank that skipped the table all-reduce would leave them waiting)
            z = torch.zeros(N, device=rays_o.device)
            return {"depth_lidar": z.view(*prefix), "image_lidar": torch.zeros(*prefix, self.out_dim, device=z.device),
                    "weights_sum_lidar": z}
        if use_fused:
            # one autograd node over the ragged samples (encode -> sigma net -> colour head -> compositing)
            ws, depth, image = fused.render_lidar_ragged(self, xyzs, dirs, deltas, rays, rays_o, rays_d, T_thresh)
            return {"depth_lidar": depth.view(*prefix), "image_lidar": image.view(*prefix, self.out_dim),
                    "weights_sum_lidar": ws}
        dens = self.density(xyzs)
        sigmas = dens["sigma"].float() * self.density_scale
        # every marched sample lies in an occupied cell: the colour head runs on all of them (no weight mask)
        feats = self.color(xyzs, dirs, cal_lidar_color=True, mask=None, geo_feat=dens["geo_feat"]).float()
        ws, depth, image = raymarching.composite_rays_train_lidar(sigmas, feats, deltas, xyzs, rays_o, rays_d, rays,
                                                                 T_thresh)
        return {"depth_lidar": depth.view(*prefix), "image_lidar": image.view(*prefix, self.out_dim),
                "weights_sum_lidar": ws}

    def render(self, rays_o, rays_d, cal_lidar_color=False, staged=False, max_ray_batch=4096, **kwargs):
        if self.cuda_ray and cal_lidar_color:
            run = self.run_cuda
        else:
            run = self.run
        if not staged:
            return run(rays_o, rays_d, cal_lidar_color=cal_lidar_color, **kwargs)
        B, N = rays_o.shape[:2]
        out_dim = self.out_lidar_color_dim if cal_lidar_color else self.out_color_dim
        depth = torch.empty((B, N), device=rays_o.device)
        image = torch.empty((B, N, out_dim), device=rays_o.device)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                part = run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail],
                           cal_lidar_color=cal_lidar_color, **kwargs)
                depth[b:b + 1, head:tail] = part["depth_lidar"]
                image[b:b + 1, head:tail] = part["image_lidar"]
        return {"depth_lidar": depth, "image_lidar": image}
