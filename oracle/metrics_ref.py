"""TEST INFRASTRUCTURE — NumPy/SciPy restatement of the LiDAR evaluation meters (lidarnerf/nerf/utils.py:226-413,
extern/fscore.py:4-18) and of skimage.metrics.structural_similarity with the defaults the reference relies on
(uniform 7x7 window, sample covariance, K1 = 0.01, K2 = 0.03, mean over the fully covered region).
PINNED (round 6) for rmse / mae / depth_errors[:4] / fscore by tests/golden/g11_metrics.npz — the reference's OWN RMSEMeter,
MAEMeter, DepthMeter and extern/fscore.py, imported with placeholder modules for the third-party imports they never touch
(tests/golden/make_g11_metrics.py; tests/test_g11_metrics_cpu.py).  PARITY UNPINNED for `ssim` (scikit-image is absent here:
restated from its published algorithm) and for the chamfer nearest-neighbour search (a CUDA kernel: checked against the
definition, brute force)."""
import numpy as np
from scipy.ndimage import uniform_filter


def rmse(preds, truths):
    return np.sqrt(((truths - preds) ** 2).mean())


def mae(preds, truths, inv_scale=1.0):
    return np.abs(truths * inv_scale - preds * inv_scale).mean()


def ssim(im1, im2, data_range, win=7, K1=0.01, K2=0.03):
    x, y = im1.astype(np.float64), im2.astype(np.float64)
    f = lambda a: uniform_filter(a, size=win)
    ux, uy, uxx, uyy, uxy = f(x), f(y), f(x * x), f(y * y), f(x * y)
    norm = win * win / (win * win - 1.0)
    vx, vy, vxy = norm * (uxx - ux * ux), norm * (uyy - uy * uy), norm * (uxy - ux * uy)
    c1, c2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    p = (win - 1) // 2
    return s[p:-p, p:-p].mean()


def depth_errors(gt, pred, min_depth=1e-3, max_depth=80, thresh_set=1.25):
    pred, gt = np.clip(pred, min_depth, max_depth), np.clip(gt, min_depth, max_depth)
    thresh = np.maximum(gt / pred, pred / gt)
    a = [(thresh < thresh_set ** k).mean() for k in (1, 2, 3)]
    return (np.sqrt(((gt - pred) ** 2).mean()), a[0], a[1], a[2],
            ssim(pred.squeeze(0), gt.squeeze(0), gt.max() - gt.min()))


def fscore(dist1, dist2, threshold):
    p, r = (dist1 < threshold).mean(), (dist2 < threshold).mean()
    return 0.0 if p + r == 0 else 2 * p * r / (p + r)
