"""PyTorch-CPU fp32 restatement of the reference renderer / network / loss (the float part of the hot path).

TEST INFRASTRUCTURE ONLY (also the `cpu_baseline` leg of bench.py).  Follows /root/reference:
  lidarnerf/nerf/renderer.py:10-46     sample_pdf
  lidarnerf/nerf/renderer.py:99-298    NeRFRenderer.run  (LiDAR mode: cal_lidar_color=True, bg_radius<=0)
  lidarnerf/nerf/network.py:162-179    density  (encoder -> Linear/ReLU stack -> trunc_exp | geo_feat)
  lidarnerf/nerf/network.py:199-237    color    (masked; freq-encoded dir ++ geo_feat -> Linear stack -> sigmoid)
  lidarnerf/nerf/utils.py:697-746      train_step LiDAR loss (+ 760-876 patch gradient term)
Pinned against the imported reference by tests/golden (G1 sample_pdf, G2 run, G4 freq, G5 trunc_exp).
"""
import numpy as np
import torch

from . import c_oracle, encoders_ref, grid_ref


# ----------------------------------------------------------------------------- sampling
def sample_pdf(bins, weights, n_samples, det=False, u=None):
    """renderer.py:10-46.  `u` lets a test inject the uniform draws (the reference uses torch.rand)."""
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples)
            u = u.expand(list(cdf.shape[:-1]) + [n_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


def weights_from_sigma(z_vals, sigma, sample_dist, density_scale=1.0):
    """renderer.py:233-243 (same arithmetic as 180-194): deltas, alphas, exclusive cumprod of (1-alpha+1e-15)."""
    deltas = z_vals[..., 1:] - z_vals[..., :-1]
    deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
    alphas = 1 - torch.exp(-deltas * density_scale * sigma)
    shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
    weights = alphas * torch.cumprod(shifted, dim=-1)[..., :-1]
    return weights, deltas


def run_lidar(rays_o, rays_d, density_fn, color_fn, aabb, min_near_lidar, num_steps=768, upsample_steps=64,
              perturb=False, training=False, density_scale=1.0, noise=None, u=None, out_dim=2, z_override=None,
              new_z_override=None):
    """renderer.py:99-298 for cal_lidar_color=True.  density_fn(x[P,3]) -> (sigma[P], geo[P,G]);
    color_fn(x[P,3], d[P,3], mask[P], geo[P,G]) -> rgb[P,out_dim] (zeros where ~mask).
    `noise` [N,num_steps] in [0,1) replaces torch.rand for the perturbation; `u` [N,upsample] for sample_pdf.
    `z_override` [N,num_steps] / `new_z_override` [N,upsample] inject the sample depths themselves (a parity test that
    hands this function the depths the HIP chain drew compares everything downstream without the ill-conditioned
    dependence of the importance samples on an ulp of the cdf)."""
    rays_o = rays_o.reshape(-1, 3)
    rays_d = rays_d.reshape(-1, 3)
    N = rays_o.shape[0]
    nears = torch.full((N, 1), float(min_near_lidar), dtype=rays_o.dtype)
    fars = nears * 81.0
    z_vals = torch.linspace(0.0, 1.0, num_steps).unsqueeze(0).expand(N, num_steps)
    z_vals = nears + (fars - nears) * z_vals
    sample_dist = (fars - nears) / num_steps
    if perturb:
        if noise is None:
            noise = torch.rand(z_vals.shape)
        z_vals = z_vals + (noise - 0.5) * sample_dist
    if z_override is not None:
        z_vals = z_override.reshape(N, num_steps).to(rays_o.dtype)
    xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
    xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])
    sigma, geo = density_fn(xyzs.reshape(-1, 3))
    sigma, geo = sigma.view(N, num_steps), geo.view(N, num_steps, -1)
    if upsample_steps > 0:
        with torch.no_grad():
            weights, deltas = weights_from_sigma(z_vals, sigma, sample_dist, density_scale)
            z_mid = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
            new_z = sample_pdf(z_mid, weights[:, 1:-1], upsample_steps, det=not training, u=u).detach()
            if new_z_override is not None:
                new_z = new_z_override.reshape(N, upsample_steps).to(rays_o.dtype)
            new_xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * new_z.unsqueeze(-1)
            new_xyzs = torch.min(torch.max(new_xyzs, aabb[:3]), aabb[3:])
        new_sigma, new_geo = density_fn(new_xyzs.reshape(-1, 3))
        new_sigma, new_geo = new_sigma.view(N, upsample_steps), new_geo.view(N, upsample_steps, -1)
        z_vals = torch.cat([z_vals, new_z], dim=1)
        z_vals, z_index = torch.sort(z_vals, dim=1)
        xyzs = torch.gather(torch.cat([xyzs, new_xyzs], dim=1), 1, z_index.unsqueeze(-1).expand(-1, -1, 3))
        sigma = torch.gather(torch.cat([sigma, new_sigma], dim=1), 1, z_index)
        geo_all = torch.cat([geo, new_geo], dim=1)
        geo = torch.gather(geo_all, 1, z_index.unsqueeze(-1).expand_as(geo_all))
    weights, _ = weights_from_sigma(z_vals, sigma, sample_dist, density_scale)
    T = z_vals.shape[1]
    dirs = rays_d.view(-1, 1, 3).expand(N, T, 3)
    mask = weights > 1e-4
    rgbs = color_fn(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask.reshape(-1), geo.reshape(N * T, -1))
    rgbs = rgbs.view(N, T, out_dim)
    weights_sum = weights.sum(dim=-1)
    depth = torch.sum(weights * z_vals, dim=-1)
    image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
    return {"depth_lidar": depth, "image_lidar": image, "weights_sum_lidar": weights_sum,
            "weights": weights, "z_vals": z_vals, "mask": mask}


def composite_ragged(sigmas, feats, deltas, xyzs, rays_o, rays_d, rays, T_thresh=1e-4):
    """Ragged LiDAR compositing over the marcher's samples (the product's lnh_lidar_composite_rays_train_*): the weight
    arithmetic of raymarching.cu:577-655 (alpha = 1 - exp(-sigma dt), w = alpha T, T *= 1 - alpha, stop once T < T_thresh)
    with the LiDAR outputs of renderer.py:268-271 — absolute depth sum(w z), z = (xyz - o) . d, and K feature channels.
    Differentiable torch code (autograd supplies the reference gradients); small inputs only (Python loop over rays)."""
    N, K = rays.shape[0], feats.shape[-1]
    ws = torch.zeros(N, dtype=sigmas.dtype)
    depth = torch.zeros(N, dtype=sigmas.dtype)
    image = torch.zeros(N, K, dtype=sigmas.dtype)
    ws_l, d_l, im_l = [None] * N, [None] * N, [None] * N
    for n in range(N):
        idx, off, cnt = (int(v) for v in rays[n])
        zero = sigmas.sum() * 0
        if cnt == 0 or off + cnt > sigmas.shape[0]:
            ws_l[idx], d_l[idx], im_l[idx] = zero, zero, zero.expand(K)
            continue
        sl = slice(off, off + cnt)
        alpha = 1 - torch.exp(-sigmas[sl] * deltas[sl, 0])
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=sigmas.dtype), 1 - alpha]), 0)  # T[i] before sample i
        below = (T[1:] < T_thresh).nonzero()
        last = int(below[0]) if below.numel() else cnt - 1                                # last sample that counts
        keep = (torch.arange(cnt) <= last).to(sigmas.dtype)
        w = alpha * T[:-1] * keep
        z = ((xyzs[sl] - rays_o[idx]) * rays_d[idx]).sum(-1)
        ws_l[idx], d_l[idx], im_l[idx] = w.sum(), (w * z).sum(), (w[:, None] * feats[sl]).sum(0)
    for i in range(N):
        if ws_l[i] is None:
            z0 = sigmas.sum() * 0
            ws_l[i], d_l[i], im_l[i] = z0, z0, z0.expand(K)
    return torch.stack(ws_l), torch.stack(d_l), torch.stack(im_l)


# ----------------------------------------------------------------------------- network pieces
class _TruncExp(torch.autograd.Function):
    """activation.py:6-20"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


class _GridEncodeCPU(torch.autograd.Function):
    """grid.py:24-138 on the CPU through the C oracle: fp32 tables, or — `half` — the autocast mode of grid.py:54-57: fp16
    table, fp16 accumulation (gridencoder.cu:173-198) and an fp16 gradient whose w * g contributions are fp16 values
    (gridencoder.cu:335-350: __half2 products)."""

    @staticmethod
    def forward(ctx, x01, emb, offsets, S, H, half=False):
        table = emb.detach().numpy()
        out, _ = c_oracle.grid_forward(x01.detach().numpy(), table.astype(np.float16) if half else table, offsets, S, H)
        ctx.save_for_backward(x01)
        ctx.meta = (offsets, S, H, emb.shape[0], half)
        L, B, Cc = out.shape
        return torch.from_numpy(out.astype(np.float32)).permute(1, 0, 2).reshape(B, L * Cc)

    @staticmethod
    def backward(ctx, g):
        (x01,) = ctx.saved_tensors
        offsets, S, H, rows, half = ctx.meta
        L = len(offsets) - 1
        B = g.shape[0]
        gl = g.view(B, L, -1).permute(1, 0, 2).contiguous().numpy().astype(np.float16 if half else np.float32)
        ge = c_oracle.grid_backward(gl, x01.numpy(), offsets, rows, S, H)
        return None, torch.from_numpy(ge.astype(np.float32)), None, None, None, None


class _Stored(torch.autograd.Function):
    """A value that passes through a 16-bit buffer: rounded to `dt` on the way forward, and the gradient arriving for it
    rounded to `dt` on the way back (the kernels keep activations AND their gradient rows in the MLP element type;
    DESIGN.md §5b).  Model of where 16-bit storage sits in the fused chain — the reference's own fp16 mode (ffmlp: __half
    activations, forward_buffer / backward_buffer) has the same storage points."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def _stored(x, dt):
    return x if dt is None else _Stored.apply(x, dt)


def freq_encode_torch(d, degree):
    """encoding.py:35-47 layout ([x | sin(2^f x) | cos(2^f x)]_f)."""
    out = [d]
    for f in range(degree):
        out.append(torch.sin(d * (2.0 ** f)))
        out.append(torch.cos(d * (2.0 ** f)))
    return torch.cat(out, dim=-1)


class RefLidarField(torch.nn.Module):
    """network.py NeRFNetwork restricted to the LiDAR branch (density + lidar colour), fp32, CPU."""

    def __init__(self, desired_resolution=32768, log2_hashmap_size=19, num_levels=16, level_dim=2, base_resolution=16,
                 hidden_dim=64, geo_feat_dim=15, hidden_dim_color=64, num_layers_color=3, freq_degree=12, bound=1.0,
                 out_dim=2, storage=None):
        super().__init__()
        # storage = torch.float16 / torch.bfloat16: model of the 16-bit mode (the reference's --fp16; BASELINE config 5 for
        # bf16) — fp16 hash table / features / feature gradients, MLP activations and their gradient rows in `storage`,
        # every dot product exact-ish (fp32) and rounded ONCE where it is stored.  None: fp32 throughout.
        self.storage = storage
        self.bound = bound
        self.pls = grid_ref.per_level_scale(desired_resolution, base_resolution, num_levels)
        self.S = float(np.log2(self.pls))
        self.H = base_resolution
        self.offsets = grid_ref.make_offsets(3, num_levels, self.pls, base_resolution, log2_hashmap_size)
        self.embeddings = torch.nn.Parameter(torch.empty(int(self.offsets[-1]), level_dim).uniform_(-1e-4, 1e-4))
        in_dim = num_levels * level_dim
        self.sigma_net = torch.nn.ModuleList([torch.nn.Linear(in_dim, hidden_dim, bias=False),
                                              torch.nn.Linear(hidden_dim, 1 + geo_feat_dim, bias=False)])
        self.freq_degree = freq_degree
        cin = 3 + 6 * freq_degree + geo_feat_dim
        dims = [cin] + [hidden_dim_color] * (num_layers_color - 1) + [out_dim]
        self.lidar_color_net = torch.nn.ModuleList(
            [torch.nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(num_layers_color)])
        self.out_dim = out_dim

    def density(self, x):
        st = self.storage
        x01 = (x + self.bound) / (2 * self.bound)
        h = _GridEncodeCPU.apply(x01, self.embeddings, self.offsets, self.S, self.H, st is not None)
        if st is not None:
            h = _stored(_stored(h, torch.float16), st)  # fp16 feature buffer (and its fp16 gradient), MLP operand type
        h = _stored(torch.relu(self.sigma_net[0](h)), st)
        h = _stored(self.sigma_net[1](h), st)           # the 16-wide row: density pre-activation | geo_feat
        return trunc_exp(h[..., 0]), h[..., 1:]

    def color(self, x, d, mask, geo):
        st = self.storage
        rgbs = torch.zeros(mask.shape[0], self.out_dim, dtype=x.dtype)
        if not mask.any():
            return rgbs
        enc_d = freq_encode_torch(d[mask], self.freq_degree)
        if st is not None:
            enc_d = enc_d.to(st).to(enc_d.dtype)        # direction features as the MLP sees them (no gradient)
        h = torch.cat([enc_d, geo[mask]], dim=-1)
        for i, lin in enumerate(self.lidar_color_net):
            h = lin(h)
            if i != len(self.lidar_color_net) - 1:
                h = _stored(torch.relu(h), st)
        rgbs[mask] = torch.sigmoid(h)
        return rgbs


class RefFreqField(torch.nn.Module):
    """BASELINE config 1 (the reference's CPU-runnable path): NeRFNetwork with the PURE-TORCH positional encoder of
    encoding.py:6-47 in place of the hash grid — [x | sin(2^f x) | cos(2^f x)]_{f<6} on the raw position (FreqEncoder's
    forward ignores `bound`; degree = get_encoder's default multires, encoding.py:53) -> bias-free Linear 39->64->16
    (network.py:45-59), LiDAR colour head as in RefLidarField (network.py:83-99, 199-237), fp32, CPU."""

    def __init__(self, pos_degree=6, hidden_dim=64, geo_feat_dim=15, hidden_dim_color=64, num_layers_color=3,
                 freq_degree=12, bound=1.0, out_dim=2):
        super().__init__()
        self.bound, self.pos_degree, self.freq_degree, self.out_dim = bound, pos_degree, freq_degree, out_dim
        self.storage = None  # fp32 (config 1 is the reference's CPU path)
        self.sigma_net = torch.nn.ModuleList([torch.nn.Linear(3 + 6 * pos_degree, hidden_dim, bias=False),
                                              torch.nn.Linear(hidden_dim, 1 + geo_feat_dim, bias=False)])
        dims = [3 + 6 * freq_degree + geo_feat_dim] + [hidden_dim_color] * (num_layers_color - 1) + [out_dim]
        self.lidar_color_net = torch.nn.ModuleList(
            [torch.nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(num_layers_color)])

    def density(self, x):
        h = torch.relu(self.sigma_net[0](freq_encode_torch(x, self.pos_degree)))
        h = self.sigma_net[1](h)
        return trunc_exp(h[..., 0]), h[..., 1:]

    color = RefLidarField.color


# ----------------------------------------------------------------------------- loss
def lidar_loss(depth, image, gt, alpha_d=1000.0, alpha_r=1.0, alpha_i=10.0):
    """utils.py:712-746 with L1 depth, MSE raydrop, MSE intensity (main_lidarnerf.py:330-342 defaults)."""
    gt_raydrop = gt[..., 0]
    gt_intensity = gt[..., 1] * gt_raydrop
    gt_depth = gt[..., 2] * gt_raydrop
    pred_raydrop = image[..., 0]
    pred_intensity = image[..., 1] * gt_raydrop
    pred_depth = depth * gt_raydrop
    per_ray = (alpha_d * (pred_depth - gt_depth).abs() + alpha_r * (pred_raydrop - gt_raydrop) ** 2
               + alpha_i * (pred_intensity - gt_intensity) ** 2)
    return per_ray.mean()


def patch_grad_loss(depth, gt, px, py, scale, alpha_grad=100.0):
    """utils.py:760-876 (non-sobel, grad_loss only): pred uses |dx|, gt signed dx; only the x term enters."""
    gt_raydrop = gt[..., 0]
    gt_depth = (gt[..., 2] * gt_raydrop).reshape(-1, 1, px, py) / scale
    pred = (depth * gt_raydrop).reshape(-1, 1, px, py) / scale
    rd = gt_raydrop.reshape(-1, 1, px, py)
    pred_gx = (pred[:, :, :, :-1] - pred[:, :, :, 1:]).abs()
    gt_gx = gt_depth[:, :, :, :-1] - gt_depth[:, :, :, 1:]
    mask_dx = rd[:, :, :, :-1] * torch.where(gt_gx.abs() < 0.01, 1, 0)
    return alpha_grad * (pred_gx * mask_dx - gt_gx * mask_dx).abs().mean()
