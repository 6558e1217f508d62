#!/usr/bin/env python3
"""G11 — the reference's OWN evaluation meters on synthetic range images (CPU, build container only: needs /root/reference):
RMSEMeter, MAEMeter, DepthMeter (lidarnerf/nerf/utils.py:226-372) and extern/fscore.py:4-18, imported and run as they are.
nerf/utils.py's module-level third-party imports that these classes never touch (cv2, imageio, lpips, mcubes, tensorboardX,
torch_ema, the chamfer CUDA extension) resolve to empty placeholder modules, as for G8.  scikit-image is not installed here:
`structural_similarity` is a placeholder that returns NaN, so of DepthMeter's five numbers (rmse, a1, a2, a3, ssim) the first
four are pinned and the SSIM stays a restatement of skimage's published algorithm (oracle/metrics_ref.py says so).

    python tests/golden/make_g11_metrics.py     # writes tests/golden/g11_metrics.npz
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


def stub(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        sys.modules[name] = m
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


for name in ("trimesh", "cv2", "imageio", "lpips", "mcubes", "tensorboardX"):
    stub(name)
stub("skimage").metrics = stub("skimage.metrics", structural_similarity=lambda a, b, data_range=None: float("nan"))
stub("torch_ema", ExponentialMovingAverage=None)
from extern.fscore import fscore  # noqa: E402  (pure torch: the real one)
stub("extern.chamfer3D")
stub("extern.chamfer3D.dist_chamfer_3D", chamfer_3DDist=None)
from lidarnerf.nerf.utils import DepthMeter, MAEMeter, RMSEMeter  # noqa: E402

SCALE = 0.010784853507573345


def main():
    rng = np.random.default_rng(11)
    out = {"scale": np.float64(SCALE)}
    frames = []
    for k, (H, W) in enumerate(((33, 515), (16, 256), (16, 256))):  # (the formulae do not depend on the image size)
        gt = (rng.uniform(2.0, 78.0, (1, H, W)) * (rng.uniform(size=(1, H, W)) > 0.2)).astype(np.float32)
        pred = (gt * rng.normal(1.0, 0.03 * (k + 1), gt.shape)
                + (gt == 0) * (rng.uniform(size=gt.shape) > 0.97) * 5.0).astype(np.float32)
        pred[0, 0, :4] = [0.0, 100.0, 1e-4, 80.0]  # below min_depth, above max_depth, at the clamps
        frames.append((pred, gt))
    for k, (p, g) in enumerate(frames):
        out[f"pred{k}"], out[f"gt{k}"] = p, g
    r, m, d = RMSEMeter(), MAEMeter(intensity_inv_scale=2.0), DepthMeter(scale=SCALE)
    per_frame = []
    for pred, gt in frames:
        r.update(torch.from_numpy(pred), torch.from_numpy(gt))
        m.update(torch.from_numpy(pred), torch.from_numpy(gt))
        # DepthMeter divides by the scene scale itself and clamps its (numpy) inputs in place: hand it scaled copies
        d.update(torch.from_numpy(pred * np.float32(SCALE)), torch.from_numpy(gt * np.float32(SCALE)))
        per_frame.append(d.V[-1][:4])
    out.update(rmse=np.float64(r.measure()), mae=np.float64(m.measure()), depth_per_frame=np.array(per_frame, dtype=np.float64),
               depth_measure=np.array(d.measure()[:4], dtype=np.float64))
    # extern/fscore.py on squared nearest-neighbour distances [B, N]
    d1 = torch.from_numpy(rng.exponential(0.03, (2, 5000)).astype(np.float32))
    d2 = torch.from_numpy(rng.exponential(0.06, (2, 4000)).astype(np.float32))
    d1[1] = 1.0
    d2[1] = 2.0  # nothing below the threshold: the 0 / 0 -> 0 rule
    f, p, q = fscore(d1, d2, 0.05)
    out.update(fs_d1=d1.numpy(), fs_d2=d2.numpy(), fs_threshold=np.float64(0.05), fs_f=f.numpy(), fs_p=p.numpy(), fs_r=q.numpy())
    path = os.path.join(OUT, "g11_metrics.npz")
    np.savez_compressed(path, **out)
    print(path, {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items() if not k.startswith(("pred", "gt"))})


if __name__ == "__main__":
    main()
