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
    def __init__(self, bound=1, density_scale=1, min_near=0.2, min_near_lidar=0.2, density_thresh=0.01, bg_radius=-1):
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

    def render(self, rays_o, rays_d, cal_lidar_color=False, staged=False, max_ray_batch=4096, **kwargs):
        if not staged:
            return self.run(rays_o, rays_d, cal_lidar_color=cal_lidar_color, **kwargs)
        B, N = rays_o.shape[:2]
        out_dim = self.out_lidar_color_dim if cal_lidar_color else self.out_color_dim
        depth = torch.empty((B, N), device=rays_o.device)
        image = torch.empty((B, N, out_dim), device=rays_o.device)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                part = self.run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail],
                                cal_lidar_color=cal_lidar_color, **kwargs)
                depth[b:b + 1, head:tail] = part["depth_lidar"]
                image[b:b + 1, head:tail] = part["image_lidar"]
        return {"depth_lidar": depth, "image_lidar": image}
