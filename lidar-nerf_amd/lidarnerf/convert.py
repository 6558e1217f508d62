"""Point cloud <-> range image on the GPU — same function names and argument meaning as lidarnerf/convert.py
(lidar_to_pano_with_intensities 99-160, lidar_to_pano 163-191, pano_to_lidar_with_intensities 194-237, pano_to_lidar
240-254).  The reference loops over points in Python; here one atomic-min pass + one resolve pass.

Inputs may be NumPy arrays (results come back as NumPy, like the reference) or CUDA tensors (results stay on the GPU).
Values are float32 (the reference stores the same float32 values in float64 arrays).  Not built: the bbox-mask and
z-buffer ("fpa") variants of convert.py:4-97, 257-361.
"""
import numpy as np
import torch

from . import _hip


def _to_gpu(a, cols=None):
    was_np = isinstance(a, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda() if was_np else a
    if not t.is_cuda:
        raise RuntimeError("lidarnerf.convert: tensors must live on the GPU (no CPU fallback)")
    t = t.float().contiguous()
    if cols is not None and (t.dim() != 2 or t.shape[1] != cols):
        raise ValueError(f"expected an [N, {cols}] array, got {tuple(t.shape)}")
    return t, was_np


def lidar_to_pano_with_intensities(local_points_with_intensities, lidar_H, lidar_W, lidar_K, max_depth=80):
    pts, was_np = _to_gpu(local_points_with_intensities, 4)
    fov_up, fov = float(lidar_K[0]), float(lidar_K[1])
    H, W = int(lidar_H), int(lidar_W)
    keys = torch.empty(H * W, dtype=torch.int64, device=pts.device)
    pano = torch.empty((H, W), dtype=torch.float32, device=pts.device)
    inten = torch.empty((H, W), dtype=torch.float32, device=pts.device)
    _hip.call("lnh_lidar_to_pano", pts.data_ptr(), pts.shape[0], H, W, fov_up, fov, float(max_depth), keys.data_ptr(),
              pano.data_ptr(), inten.data_ptr())
    if was_np:
        return pano.cpu().numpy().astype(np.float64), inten.cpu().numpy().astype(np.float64)
    return pano, inten


def lidar_to_pano(local_points, lidar_H, lidar_W, lidar_K, max_depth=80):
    if isinstance(local_points, np.ndarray):
        p4 = np.concatenate([local_points, np.zeros((local_points.shape[0], 1), local_points.dtype)], axis=1)
    else:
        p4 = torch.cat([local_points, torch.zeros_like(local_points[:, :1])], dim=1)
    return lidar_to_pano_with_intensities(p4, lidar_H, lidar_W, lidar_K, max_depth)[0]


def pano_to_lidar_with_intensities(pano, intensities, lidar_K):
    p, was_np = _to_gpu(pano)
    it = None if intensities is None else _to_gpu(intensities)[0]
    H, W = p.shape
    fov_up, fov = float(lidar_K[0]), float(lidar_K[1])
    pts = torch.empty((H * W, 4), dtype=torch.float32, device=p.device)
    valid = torch.empty(H * W, dtype=torch.uint8, device=p.device)
    _hip.call("lnh_pano_to_lidar", p.data_ptr(), None if it is None else it.data_ptr(), H, W, fov_up, fov,
              pts.data_ptr(), valid.data_ptr())
    out = pts[valid.bool()]  # pixel (row-major) order, like np.where
    return out.cpu().numpy() if was_np else out


def pano_to_lidar(pano, lidar_K):
    return pano_to_lidar_with_intensities(pano, None, lidar_K)[:, :3]
