"""Evaluation metrics on the device: the meters of lidarnerf/nerf/utils.py:226-427 that the LiDAR
evaluation uses, without the reference's round trip through NumPy/CPU, and the chamfer distance of
extern/chamfer3D (dist_chamfer_3D.py, chamfer3D.cu) + extern/fscore.py on the HIP nearest-neighbour kernel.

Same class names, constructor arguments, `update / measure / report / clear` protocol and numbers; inputs are CUDA
tensors (NumPy arrays are moved to the GPU).  Not here: PSNR/LPIPS image meters of the RGB branch.
"""
import numpy as np
import torch

from . import _hip
from .convert import pano_to_lidar


def _gpu(x):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not torch.is_tensor(x):
        x = torch.as_tensor(x)
    return x.detach().cuda().float()


class chamfer_3DDist(torch.nn.Module):
    """dist1 [B,n], dist2 [B,m] squared nearest-neighbour distances, idx1 / idx2 int32 (dist_chamfer_3D.py:37-84).
    Forward only (the reference's evaluation never back-propagates through it)."""

    def forward(self, xyz1, xyz2):
        xyz1, xyz2 = xyz1.float().contiguous(), xyz2.float().contiguous()
        if not (xyz1.is_cuda and xyz2.is_cuda):
            raise RuntimeError("chamfer_3DDist: inputs must live on the GPU (no CPU fallback)")
        B, n, d1 = xyz1.shape
        _, m, d2 = xyz2.shape
        assert d1 == 3 and d2 == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        dist1 = torch.zeros((B, n), device=xyz1.device)
        dist2 = torch.zeros((B, m), device=xyz1.device)
        idx1 = torch.zeros((B, n), dtype=torch.int32, device=xyz1.device)
        idx2 = torch.zeros((B, m), dtype=torch.int32, device=xyz1.device)
        for b in range(B):
            _hip.call("lnh_chamfer_nn", xyz1[b].data_ptr(), n, xyz2[b].data_ptr(), m, dist1[b].data_ptr(),
                      idx1[b].data_ptr())
            _hip.call("lnh_chamfer_nn", xyz2[b].data_ptr(), m, xyz1[b].data_ptr(), n, dist2[b].data_ptr(),
                      idx2[b].data_ptr())
        return dist1, dist2, idx1, idx2


def fscore(dist1, dist2, threshold=0.001):
    """extern/fscore.py:4-18 (distances are squared: adapt the threshold)."""
    precision_1 = torch.mean((dist1 < threshold).float(), dim=1)
    precision_2 = torch.mean((dist2 < threshold).float(), dim=1)
    f = 2 * precision_1 * precision_2 / (precision_1 + precision_2)
    f[torch.isnan(f)] = 0
    return f, precision_1, precision_2


class _Meter:
    def __init__(self):
        self.V, self.N = 0, 0

    def clear(self):
        self.V, self.N = 0, 0

    def measure(self):
        return self.V / self.N


class RMSEMeter(_Meter):
    """utils.py:226-260"""

    def update(self, preds, truths):
        preds, truths = _gpu(preds), _gpu(truths)
        self.V += float(torch.sqrt(((truths - preds) ** 2).mean()))
        self.N += 1

    def report(self):
        return f"RMSE = {self.measure():.6f}"


class MAEMeter(_Meter):
    """utils.py:263-301"""

    def __init__(self, intensity_inv_scale=1.0):
        super().__init__()
        self.intensity_inv_scale = intensity_inv_scale

    def update(self, preds, truths):
        preds, truths = _gpu(preds), _gpu(truths)
        self.V += float((truths * self.intensity_inv_scale - preds * self.intensity_inv_scale).abs().mean())
        self.N += 1

    def report(self):
        return f"MAE = {self.measure():.6f}"


def structural_similarity(im1, im2, data_range, win_size=7, K1=0.01, K2=0.03):
    """Mean SSIM with the defaults the reference relies on (skimage.metrics.structural_similarity: uniform 7x7 window,
    sample covariance, mean over the region the window covers completely)."""
    x, y = im1[None, None].double(), im2[None, None].double()
    pool = torch.nn.functional.avg_pool2d
    ux, uy = pool(x, win_size, 1), pool(y, win_size, 1)
    uxx, uyy, uxy = pool(x * x, win_size, 1), pool(y * y, win_size, 1), pool(x * y, win_size, 1)
    norm = win_size ** 2 / (win_size ** 2 - 1.0)
    vx, vy, vxy = norm * (uxx - ux * ux), norm * (uyy - uy * uy), norm * (uxy - ux * uy)
    c1, c2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return float(s.mean())


class DepthMeter:
    """utils.py:304-362: (rmse, a1, a2, a3, ssim) of a [1,H,W] depth image in metres."""

    def __init__(self, scale):
        self.V, self.N, self.scale = [], 0, scale

    def clear(self):
        self.V, self.N = [], 0

    def update(self, preds, truths):
        preds, truths = _gpu(preds) / self.scale, _gpu(truths) / self.scale
        self.V.append(list(self.compute_depth_errors(truths, preds)))
        self.N += 1

    def compute_depth_errors(self, gt, pred, min_depth=1e-3, max_depth=80, thresh_set=1.25):
        pred, gt = pred.clamp(min_depth, max_depth), gt.clamp(min_depth, max_depth)
        thresh = torch.maximum(gt / pred, pred / gt)
        a1 = float((thresh < thresh_set).float().mean())
        a2 = float((thresh < thresh_set ** 2).float().mean())
        a3 = float((thresh < thresh_set ** 3).float().mean())
        rmse = float(torch.sqrt(((gt - pred) ** 2).mean()))
        ssim = structural_similarity(pred.squeeze(0), gt.squeeze(0), data_range=float(gt.max() - gt.min()))
        return rmse, a1, a2, a3, ssim

    def measure(self):
        assert self.N == len(self.V)
        return np.array(self.V).mean(0)

    def report(self):
        return f"Depth_error(rmse, a1, a2, a3, ssim) = {self.measure()}"


class PointsMeter:
    """utils.py:365-413: chamfer distance + F-score (threshold 0.05 on squared distances) of the point clouds
    back-projected from the predicted and the ground-truth range image."""

    def __init__(self, scale, intrinsics):
        self.V, self.N, self.scale, self.intrinsics = [], 0, scale, intrinsics

    def clear(self):
        self.V, self.N = [], 0

    def update(self, preds, truths):
        preds, truths = _gpu(preds) / self.scale, _gpu(truths) / self.scale
        pred_lidar = pano_to_lidar(preds[0], self.intrinsics)
        gt_lidar = pano_to_lidar(truths[0], self.intrinsics)
        dist1, dist2, _, _ = chamfer_3DDist()(pred_lidar[None], gt_lidar[None])
        chamfer_dis = dist1.mean() + dist2.mean()
        f, _, _ = fscore(dist1, dist2, 0.05)
        self.V.append([float(chamfer_dis), float(f[0])])
        self.N += 1

    def measure(self):
        assert self.N == len(self.V)
        return np.array(self.V).mean(0)

    def report(self):
        return f"CD f-score = {self.measure()}"
