"""TEST INFRASTRUCTURE — CPU restatement of lidarnerf/convert.py (lidar_to_pano_with_intensities 99-160,
pano_to_lidar_with_intensities 194-237), vectorised.  Pinned bit-exactly against outputs of the imported reference
(tests/golden/g6_convert.npz, tests/test_oracle_golden.py).

The reference's per-point loop uses float32 NumPy scalars; under NumPy >= 2 promotion (NEP 50) the Python-float
constants (np.pi, fov/180*np.pi/H, ...) are computed in double and rounded to float32 at the operation.  Only the
nearest point per pixel survives, the first one on ties: a stable sort by (pixel, dist) reproduces the loop.
"""
import numpy as np


def lidar_to_pano_with_intensities(pts, H, W, lidar_K, max_depth=80):
    pts = np.asarray(pts, dtype=np.float32)
    fov_up, fov = lidar_K
    fov_down = fov - fov_up
    xyz, inten = pts[:, :3], pts[:, 3]
    dist = np.linalg.norm(xyz, axis=1)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    f32 = np.float32
    beta = f32(np.pi) - np.arctan2(y, x)
    alpha = np.arctan2(z, np.sqrt(x ** 2 + y ** 2)) + f32(fov_down / 180 * np.pi)
    c = np.rint(beta / f32(2 * np.pi / W))           # Python round() on a float32 scalar: half to even
    r = np.rint(f32(H) - alpha / f32(fov / 180 * np.pi / H))
    ok = (dist < max_depth) & (r >= 0) & (r < H) & (c >= 0) & (c < W)
    pano = np.zeros((H, W), np.float64)
    out_i = np.zeros((H, W), np.float64)
    idx = np.nonzero(ok)[0]
    pix = r[idx].astype(np.int64) * W + c[idx].astype(np.int64)
    order = np.lexsort((idx, dist[idx], pix))        # by pixel, then distance, then original position
    pix_s, idx_s = pix[order], idx[order]
    first = np.ones(len(order), bool)
    first[1:] = pix_s[1:] != pix_s[:-1]
    win = idx_s[first]
    pano.reshape(-1)[pix_s[first]] = dist[win]
    out_i.reshape(-1)[pix_s[first]] = inten[win]
    return pano, out_i


def pano_to_lidar_with_intensities(pano, intensities, lidar_K):
    fov_up, fov = lidar_K
    H, W = pano.shape
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    beta = -(i - W / 2) / W * 2 * np.pi
    alpha = (fov_up - j / H * fov) / 180 * np.pi
    dirs = np.stack([np.cos(alpha) * np.cos(beta), np.cos(alpha) * np.sin(beta), np.sin(alpha)], -1)
    pts = dirs * pano.reshape(H, W, 1)
    out = np.concatenate([pts, intensities.reshape(H, W, 1)], axis=2)
    return out[np.where(pano != 0.0)]
