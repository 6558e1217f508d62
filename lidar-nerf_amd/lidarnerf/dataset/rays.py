"""LiDAR range-image ray generation (the input generator of the hot path).

Same result as `get_lidar_rays` of the reference (lidarnerf/dataset/base_dataset.py:16-105): pixel (row j, col i)
of an H x W range image maps to azimuth beta = -(i - W/2)/W * 2pi and elevation alpha = (fov_up - j/H * fov) deg.
"""
import numpy as np
import torch


def patch_indices(H, W, N, patch_size, device):
    """Flattened pixel indices of N rays drawn as num_patch = N/(px*py) patches (base_dataset.py:50-70).
    Top-left corners: row ~ randint(0, H-px), col ~ randint(0, W-py) — so with px = py = 1 the last row / column is
    never sampled, exactly like the reference."""
    px, py = (patch_size, patch_size) if isinstance(patch_size, int) else \
        ((patch_size[0], patch_size[0]) if len(patch_size) == 1 else tuple(patch_size))
    if px <= 0:
        return torch.randint(0, H * W, size=[N], device=device)
    num_patch = N // (px * py)
    rows = torch.randint(0, H - px, size=[num_patch], device=device)
    cols = torch.randint(0, W - py, size=[num_patch], device=device)
    dr, dc = torch.meshgrid(torch.arange(px, device=device), torch.arange(py, device=device), indexing="ij")
    r = (rows[:, None] + dr.reshape(1, -1)).reshape(-1)
    c = (cols[:, None] + dc.reshape(1, -1)).reshape(-1)
    return r * W + c


@torch.autocast("cuda", enabled=False)
def get_lidar_rays(poses, intrinsics, H, W, N=-1, patch_size=1):
    """poses [B,4,4] (lidar->world), intrinsics (fov_up, fov) in degrees -> dict(rays_o, rays_d [B,N,3], inds [B,N])."""
    device = poses.device
    B = poses.shape[0]
    if N > 0:
        N = min(N, H * W)
        inds = patch_indices(H, W, N, patch_size, device)
    else:
        inds = torch.arange(H * W, device=device)
    col = (inds % W).to(torch.float32)
    row = torch.div(inds, W, rounding_mode="floor").to(torch.float32)
    fov_up, fov = intrinsics
    beta = -(col - W / 2) / W * 2 * np.pi
    alpha = (fov_up - row / H * fov) / 180 * np.pi
    d = torch.stack([torch.cos(alpha) * torch.cos(beta), torch.cos(alpha) * torch.sin(beta), torch.sin(alpha)], -1)
    d = d.unsqueeze(0).expand(B, -1, -1)
    rays_d = d @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    return {"rays_o": rays_o, "rays_d": rays_d, "inds": inds.expand(B, -1)}
